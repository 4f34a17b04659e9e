// line3D_amd.hpp -- C++ facade with the reference's public interface (class L3D::Line3D, line3D.h:61-101)
// over the C ABI of include/line3d_amd.h.  Same method names, argument order and defaults (commons.h:42-61).
// addImage / addImage_fixed_sim come in two families: (1) the reference's own signatures -- `image` (anything with .cols / .rows: cv::Mat),
// K, R, t matrix-typed (anything with K(i, j) / t(i): Eigen's) -- whose segments come from the segment cache of the data directory, the
// reference's own side door for precomputed segments (line3D.cc:143-168); (2) width, height and the segments the detector would have
// produced (std::vector<float4>, the side door of L3DSegments(list<float4>&, bool), segments.h:60), cameras as plain row-major arrays or
// matrix types.  Neither OpenCV nor Eigen is needed to compile this header.
#pragma once

#include <array>
#include <cstdio>
#include <iostream>
#include <list>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "line3d_amd.h"

namespace L3D {

#ifndef L3D_AMD_HAVE_FLOAT4
struct float4 { float x, y, z, w; };      // the reference gets this type from the CUDA headers
#endif
typedef std::array<double, 3> Vec3d;

// commons.h:81-99
class L3DSegment2D {
public:
    L3DSegment2D() : camID_(0), segID_(0) {}
    L3DSegment2D(unsigned int camID, unsigned int segID) : camID_(camID), segID_(segID) {}
    unsigned int camID() const { return camID_; }
    unsigned int segID() const { return segID_; }
    bool operator==(const L3DSegment2D& r) const { return camID_ == r.camID_ && segID_ == r.segID_; }
    bool operator<(const L3DSegment2D& r) const { return camID_ < r.camID_ || (camID_ == r.camID_ && segID_ < r.segID_); }
    bool operator!=(const L3DSegment2D& r) const { return !(*this == r); }
private:
    unsigned int camID_, segID_;
};

// commons.h:215-238
class L3DFinalLine3D {
public:
    L3DFinalLine3D(std::list<L3DSegment2D> segments2D, std::list<std::pair<Vec3d, Vec3d> > segments3D)
        : segments3D_(std::move(segments3D)), segments2D_(std::move(segments2D)) {}
    std::list<std::pair<Vec3d, Vec3d> >* segments3D() { return &segments3D_; }
    std::list<L3DSegment2D>* segments2D() { return &segments2D_; }
private:
    std::list<std::pair<Vec3d, Vec3d> > segments3D_;
    std::list<L3DSegment2D> segments2D_;
};

class Line3D {
public:
    // line3D.h:61-66 (data_directory: where addImage keeps its segment caches, line3D.cc:143-150)
    Line3D(const std::string data_directory, const int matchingNeighbors = 10,
           const float uncertainty_t_upper_2D = 5.0f, const float uncertainty_t_lower_2D = 1.0f,
           const float sigma_p = 3.5f, const float sigma_a = 10.0f, const float min_baseline = 0.25f,
           bool useCollinearity = true, bool verbose = false, int device = 0)
        : h_(nullptr), prefix_("[L3D] "), data_directory_(data_directory), use_collinearity_(useCollinearity)
    {
        int rc = l3d_line3d_create(device, matchingNeighbors, uncertainty_t_upper_2D, uncertainty_t_lower_2D, sigma_p, sigma_a,
                                   min_baseline, useCollinearity ? 1 : 0, verbose ? 1 : 0, &h_);
        if (rc != L3D_OK) std::cerr << prefix_ << "no usable HIP device (code " << rc << "); this build has no CPU fallback" << std::endl;
    }
    ~Line3D() { l3d_line3d_destroy(h_); }
    Line3D(const Line3D&) = delete;
    Line3D& operator=(const Line3D&) = delete;
    bool valid() const { return h_ != nullptr; }

    // line3D.h:69-73; errors are printed and the call returns, like the reference (line3D.cc:101-127).  `image` is replaced by its size
    // and the segments the detector would have produced; maxImgWidth / loadAndStoreSegments keep their meaning: the segment cache
    // "<data_directory>/segments_<id>_<w'>x<h'>_coll<0|1>.bin" is removed, read INSTEAD of `segments`, or written (line3D.cc:128-199)
    void addImage(const unsigned int imageID, const unsigned int width, const unsigned int height,
                  const std::vector<float4>& segments, const double K[9], const double R[9], const double t[3],
                  std::list<unsigned int>& worldpointIDs, const int maxImgWidth = 1920, const bool loadAndStoreSegments = true)
    {
        std::vector<uint32_t> wps(worldpointIDs.begin(), worldpointIDs.end());
        report(l3d_line3d_add_image_ex(h_, imageID, width, height, segments.empty() ? nullptr : &segments[0].x, (int)segments.size(),
                                       K, R, t, wps.data(), (int)wps.size(), data_directory_.c_str(), maxImgWidth, loadAndStoreSegments ? 1 : 0));
    }
    // addImage when "<data_directory>/segments_<id>_<w>x<h>_coll<0|1>.bin" of an earlier run exists (line3D.cc:143-168):
    // the file's segments and collinearities stand in for the image (the detector is not part of this library).
    // false: no such file, or it is not a segment cache (message printed)
    bool addImageFromCache(const unsigned int imageID, const unsigned int width, const unsigned int height,
                           const double K[9], const double R[9], const double t[3], std::list<unsigned int>& worldpointIDs)
    {
        char name[128];
        if (l3d_segment_cache_filename(imageID, width, height, use_collinearity_ ? 1 : 0, name, sizeof(name)) != L3D_OK) return false;
        l3d_segment_cache* cache = nullptr;
        int rc = l3d_segment_cache_read((data_directory_ + name).c_str(), &cache);
        if (rc != L3D_OK) { std::cerr << prefix_ << l3d_segment_cache_last_error(cache) << std::endl; l3d_segment_cache_free(cache); return false; }
        std::vector<uint32_t> wps(worldpointIDs.begin(), worldpointIDs.end());
        rc = l3d_line3d_add_image_cached(h_, imageID, width, height, cache, K, R, t, wps.data(), (int)wps.size());
        l3d_segment_cache_free(cache);
        report(rc);
        return rc == L3D_OK;
    }
    // line3D.h:75-79
    void addImage_fixed_sim(const unsigned int imageID, const unsigned int width, const unsigned int height,
                            const std::vector<float4>& segments, const double K[9], const double R[9], const double t[3],
                            std::map<unsigned int, float>& viewSimilarity, const int maxImgWidth = 1920, const bool loadAndStoreSegments = true)
    {
        std::vector<uint32_t> ids;
        std::vector<float> sims;
        for (auto& kv : viewSimilarity) { ids.push_back(kv.first); sims.push_back(kv.second); }
        report(l3d_line3d_add_image_fixed_sim_ex(h_, imageID, width, height, segments.empty() ? nullptr : &segments[0].x, (int)segments.size(), K, R, t,
                                                 ids.data(), sims.data(), (int)ids.size(), data_directory_.c_str(), maxImgWidth, loadAndStoreSegments ? 1 : 0));
    }
    // The same two calls with matrix-typed cameras, as the reference's drivers pass them (Eigen::Matrix3d K, R; Eigen::Vector3d t,
    // main_vsfm.cpp:273-281): any type with K(i, j) / t(i) access -- Eigen is not a dependency of this header.
    template <class M3, class V3, class = decltype(std::declval<const M3&>()(0, 0)), class = decltype(std::declval<const V3&>()(0))>
    void addImage(const unsigned int imageID, const unsigned int width, const unsigned int height, const std::vector<float4>& segments,
                  const M3& K, const M3& R, const V3& t, std::list<unsigned int>& worldpointIDs, const int maxImgWidth = 1920,
                  const bool loadAndStoreSegments = true)
    {
        double k[9], r[9], tt[3];
        flatten(K, R, t, k, r, tt);
        addImage(imageID, width, height, segments, k, r, tt, worldpointIDs, maxImgWidth, loadAndStoreSegments);
    }
    template <class M3, class V3, class = decltype(std::declval<const M3&>()(0, 0)), class = decltype(std::declval<const V3&>()(0))>
    void addImage_fixed_sim(const unsigned int imageID, const unsigned int width, const unsigned int height, const std::vector<float4>& segments,
                            const M3& K, const M3& R, const V3& t, std::map<unsigned int, float>& viewSimilarity, const int maxImgWidth = 1920,
                            const bool loadAndStoreSegments = true)
    {
        double k[9], r[9], tt[3];
        flatten(K, R, t, k, r, tt);
        addImage_fixed_sim(imageID, width, height, segments, k, r, tt, viewSimilarity, maxImgWidth, loadAndStoreSegments);
    }
    // The reference's OWN signatures (line3D.h:69-79): `image` as the second parameter -- cv::Mat in the reference, here any type with
    // `.cols` / `.rows` (cv::Mat itself when OpenCV is there; OpenCV is not a dependency of this header), K / R / t matrix-typed as above.
    // main_vsfm.cpp:273 / main_bundler.cpp:287 compile against these unchanged.  The image gives the view its size; its PIXELS are not read:
    // line-segment detection is not part of this library, so the segments come from the reference's side door, the segment cache
    // "<data_directory>/segments_<id>_<w'>x<h'>_coll<0|1>.bin" of an earlier run (w' x h' after the maxImgWidth rule, line3D.cc:130-150).
    // No such file: the error is printed and the call returns without a view (where the reference would run LSD, line3D.cc:169-190).
    // loadAndStoreSegments = false removes the file like the reference does (line3D.cc:153-156) -- and then there is nothing to add.
    template <class Img, class M3, class V3, class = decltype(std::declval<const Img&>().cols), class = decltype(std::declval<const Img&>().rows),
              class = decltype(std::declval<const M3&>()(0, 0)), class = decltype(std::declval<const V3&>()(0))>
    void addImage(const unsigned int imageID, const Img& image, const M3& K, const M3& R, const V3& t, std::list<unsigned int>& worldpointIDs,
                  const int maxImgWidth = 1920, const bool loadAndStoreSegments = true)
    {
        double k[9], r[9], tt[3];
        flatten(K, R, t, k, r, tt);
        const unsigned int w = image.cols > 0 ? (unsigned int)image.cols : 0u, h = image.rows > 0 ? (unsigned int)image.rows : 0u;
        std::vector<uint32_t> wps(worldpointIDs.begin(), worldpointIDs.end());
        report(l3d_line3d_add_image_ex(h_, imageID, w, h, nullptr, 0, k, r, tt, wps.data(), (int)wps.size(), data_directory_.c_str(), maxImgWidth,
                                       loadAndStoreSegments ? 1 : 0));
    }
    template <class Img, class M3, class V3, class = decltype(std::declval<const Img&>().cols), class = decltype(std::declval<const Img&>().rows),
              class = decltype(std::declval<const M3&>()(0, 0)), class = decltype(std::declval<const V3&>()(0))>
    void addImage_fixed_sim(const unsigned int imageID, const Img& image, const M3& K, const M3& R, const V3& t, std::map<unsigned int, float>& viewSimilarity,
                            const int maxImgWidth = 1920, const bool loadAndStoreSegments = true)
    {
        double k[9], r[9], tt[3];
        flatten(K, R, t, k, r, tt);
        const unsigned int w = image.cols > 0 ? (unsigned int)image.cols : 0u, h = image.rows > 0 ? (unsigned int)image.rows : 0u;
        std::vector<uint32_t> ids;
        std::vector<float> sims;
        for (auto& kv : viewSimilarity) { ids.push_back(kv.first); sims.push_back(kv.second); }
        report(l3d_line3d_add_image_fixed_sim_ex(h_, imageID, w, h, nullptr, 0, k, r, tt, ids.data(), sims.data(), (int)ids.size(), data_directory_.c_str(),
                                                 maxImgWidth, loadAndStoreSegments ? 1 : 0));
    }
    // line3D.h:82
    void compute3Dmodel(bool perform_diffusion = false) { report(l3d_line3d_compute3Dmodel(h_, perform_diffusion ? 1 : 0)); }
    // line3D.h:85
    void getResult(std::list<L3DFinalLine3D>& result)
    {
        result.clear();
        int nl = 0, n3 = 0, n2 = 0;
        if (l3d_line3d_result_sizes(h_, &nl, &n3, &n2) != L3D_OK || nl == 0) return;
        std::vector<int> l3((size_t)nl), l2((size_t)nl);
        std::vector<double> s3((size_t)n3 * 6);
        std::vector<uint32_t> s2((size_t)n2 * 2);
        l3d_line3d_get_result(h_, l3.data(), l2.data(), s3.data(), s2.data());
        size_t a = 0, b = 0;
        for (int k = 0; k < nl; ++k) {
            std::list<std::pair<Vec3d, Vec3d> > seg3;
            std::list<L3DSegment2D> seg2;
            for (int i = 0; i < l3[(size_t)k]; ++i, a += 6)
                seg3.push_back({ Vec3d{ s3[a], s3[a + 1], s3[a + 2] }, Vec3d{ s3[a + 3], s3[a + 4], s3[a + 5] } });
            for (int i = 0; i < l2[(size_t)k]; ++i, b += 2) seg2.push_back(L3DSegment2D(s2[b], s2[b + 1]));
            result.push_back(L3DFinalLine3D(seg2, seg3));
        }
    }
    // line3D.h:88
    float4 getSegment2D(L3DSegment2D& seg2D)
    {
        float o[4] = { 0.0f, 0.0f, 0.0f, 0.0f };   // (stays zero when the handle is null: the constructor found no HIP device)
        if (l3d_line3d_get_segment2D(h_, seg2D.camID(), seg2D.segID(), o) != L3D_OK)
            std::cerr << prefix_ << "no view with ID " << seg2D.camID() << "!" << std::endl;
        return float4{ o[0], o[1], o[2], o[3] };
    }
    // line3D.h:91,94 -- `result` is written as given (the caller may have filtered it); formats as in line3D.cc:384-473
    void save3DLinesAsSTL(std::list<L3DFinalLine3D>& result, std::string filename)
    {
        FILE* f = fopen(filename.c_str(), "w");
        if (!f) return;
        fprintf(f, "solid lineModel\n");
        for (L3DFinalLine3D& l : result)
            for (auto& sg : *l.segments3D()) {
                fprintf(f, " facet normal 1.0e+000 0.0e+000 0.0e+000\n  outer loop\n");
                fprintf(f, "   vertex %e %e %e\n", (double)sg.first[0], (double)sg.first[1], (double)sg.first[2]);
                fprintf(f, "   vertex %e %e %e\n", (double)sg.second[0], (double)sg.second[1], (double)sg.second[2]);
                fprintf(f, "   vertex %e %e %e\n", (double)sg.first[0], (double)sg.first[1], (double)sg.first[2]);
                fprintf(f, "  endloop\n endfacet\n");
            }
        fprintf(f, "endsolid lineModel\n");
        fclose(f);
    }
    void save3DLinesAsTXT(std::list<L3DFinalLine3D>& result, std::string filename)
    {
        FILE* f = fopen(filename.c_str(), "w");
        if (!f) return;
        for (L3DFinalLine3D& l : result) {
            if (l.segments3D()->empty()) continue;
            fprintf(f, "%zu ", l.segments3D()->size());
            for (auto& sg : *l.segments3D())
                fprintf(f, "%g %g %g %g %g %g ", (double)sg.first[0], (double)sg.first[1], (double)sg.first[2], (double)sg.second[0], (double)sg.second[1], (double)sg.second[2]);
            fprintf(f, "%zu ", l.segments2D()->size());
            for (L3DSegment2D& s2 : *l.segments2D()) {
                const float4 c = getSegment2D(s2);
                fprintf(f, "%u %u %g %g %g %g ", s2.camID(), s2.segID(), (double)c.x, (double)c.y, (double)c.z, (double)c.w);
            }
            fprintf(f, "\n");
        }
        fclose(f);
    }
    unsigned int numCameras() { return (unsigned int)l3d_line3d_num_cameras(h_); }   // line3D.h:98
    void reset() { l3d_line3d_reset(h_); }                                            // line3D.h:101
    l3d_line3d* handle() { return h_; }

private:
    template <class M3, class V3> static void flatten(const M3& K, const M3& R, const V3& t, double* k, double* r, double* tt)
    {
        for (int i = 0; i < 3; ++i) { tt[i] = t(i); for (int j = 0; j < 3; ++j) { k[i * 3 + j] = K(i, j); r[i * 3 + j] = R(i, j); } }
    }
    void report(int rc) { if (rc != L3D_OK) std::cerr << prefix_ << l3d_line3d_last_error(h_) << std::endl; }
    l3d_line3d* h_;
    std::string prefix_, data_directory_;
    bool use_collinearity_;
};

}  // namespace L3D
