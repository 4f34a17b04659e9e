// TEST DOUBLES (tests only) for the two third-party types of the reference's public interface, line3D.h:69-79: cv::Mat and Eigen's 3x3 / 3x1
// double matrices.  Neither OpenCV nor Eigen exists in this image; these carry exactly what the drivers touch between cv::imread and
// Line3D::addImage (main_vsfm.cpp:225-273): image.cols / image.rows, K(i, j) = x, Matrix3d::Zero(), t(i).
// cv::imread does not decode anything: "<path>" holds the two numbers "<width> <height>" written by the test.
#pragma once
#include <cstdio>
#include <string>

namespace cv {
struct Mat { int rows = 0, cols = 0; };
inline Mat imread(const std::string& path)
{
    Mat m;
    if (FILE* f = fopen(path.c_str(), "r")) { if (fscanf(f, "%d %d", &m.cols, &m.rows) != 2) m = Mat(); fclose(f); }
    return m;
}
}  // namespace cv

namespace Eigen {
struct Matrix3d {
    double m[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    static Matrix3d Zero() { return Matrix3d(); }
    double& operator()(int i, int j) { return m[i * 3 + j]; }
    double operator()(int i, int j) const { return m[i * 3 + j]; }
};
struct Vector3d {
    double v[3] = { 0, 0, 0 };
    double& operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
};
}  // namespace Eigen
