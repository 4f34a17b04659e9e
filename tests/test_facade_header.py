"""include/line3D_amd.hpp (the C++ L3D::Line3D facade) compiles against the C ABI with plain g++ and links."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "line3D_amd.hpp"
// matrix types with the access the reference's drivers use (Eigen::Matrix3d / Eigen::Vector3d): K(i, j), t(i)
struct Mat3 { double m[9]; double operator()(int i, int j) const { return m[i * 3 + j]; } };
struct Vec3 { double v[3]; double operator()(int i) const { return v[i]; } };
int main() {
    L3D::Line3D l("dir", 10, 5.0f, 1.0f, 3.5f, 10.0f, 0.25f, true, false);
    std::list<L3D::L3DFinalLine3D> r;
    l.getResult(r);
    L3D::L3DSegment2D a(1, 2), b(1, 3);
    // every overload of line3D.h:69-79 instantiates: plain arrays and matrix types, with and without the two trailing defaults
    std::vector<L3D::float4> segs(3, L3D::float4{ 0.f, 0.f, 10.f, 10.f });
    std::list<unsigned int> wps{ 1, 2, 3 };
    std::map<unsigned int, float> sim{ { 1u, 0.5f } };
    const double K[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, t[3] = { 0, 0, 0 };
    Mat3 Km{ { 1, 0, 0, 0, 1, 0, 0, 0, 1 } };
    Vec3 tm{ { 0, 0, 0 } };
    l.addImage(0, 640, 480, segs, K, K, t, wps);
    l.addImage(1, 640, 480, segs, K, K, t, wps, 1920, false);
    l.addImage(2, 640, 480, segs, Km, Km, tm, wps);
    l.addImage(3, 640, 480, segs, Km, Km, tm, wps, 800, false);
    l.addImage_fixed_sim(4, 640, 480, segs, K, K, t, sim);
    l.addImage_fixed_sim(5, 640, 480, segs, Km, Km, tm, sim, 1920, false);
    return (a < b && !(a == b) && r.empty() && (l.numCameras() == 0 || l.numCameras() == 6)) ? 0 : 1;   // (without a GPU every call reports and returns)
}
'''


def test_facade_compiles_and_links():
    lib = os.path.join(ROOT, "line3d_amd")
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "t.cpp")
        open(src, "w").write(SRC)
        exe = os.path.join(td, "t")
        subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src, "-L" + lib, "-lline3d_amd",
                               "-Wl,-rpath," + lib, "-o", exe])
        # without a GPU the constructor reports and every call degrades to a no-op, like the reference's print-and-return
        assert subprocess.run([exe], stderr=subprocess.DEVNULL).returncode == 0
