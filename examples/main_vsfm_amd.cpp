// main_vsfm_amd.cpp -- the flow of the reference's VisualSfM driver (main_vsfm.cpp:34-329) over this library, with the
// segment caches of an earlier Line3D run standing in for the images (no OpenCV, no tclap, no boost):
//
//   main_vsfm_amd <scene.nvm | bundle.rd.out> <data directory> [neighbors=10] [diffusion=0] [output folder=<data directory>]
//
// For every camera of the NVM file the data directory ("<image folder>/L3D_data" of the reference, main_vsfm.cpp:108-116)
// must hold "segments_<id>_<w>x<h>_coll1.bin" (line3D.cc:143-150) -- the image size is read off the file name, K is built
// from the focal length and that size the way the driver does it (main_vsfm.cpp:232-241), addImage uses the cached segments
// and collinearities (line3D.cc:160-168), compute3Dmodel runs on the GPU, the result goes to
// "<output folder>/line3D_result__W_..." as STL and TXT (main_vsfm.cpp:289-325).
//
// Build:  g++ -std=c++17 -Iinclude examples/main_vsfm_amd.cpp -Lline3d_amd -lline3d_amd -Wl,-rpath,$PWD/line3d_amd -o main_vsfm_amd
#include <dirent.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>

#include "line3D_amd.hpp"

namespace {

// "segments_<id>_<w>x<h>_coll1.bin" of camera `id` in `dir`: the image size from the name
bool find_cache(const std::string& dir, unsigned id, unsigned& w, unsigned& h)
{
    DIR* d = opendir(dir.c_str());
    if (!d) return false;
    bool found = false;
    while (dirent* e = readdir(d)) {
        unsigned fid = 0, fw = 0, fh = 0, coll = 0;
        char tail[8] = { 0 };
        if (sscanf(e->d_name, "segments_%u_%ux%u_coll%u.%3s", &fid, &fw, &fh, &coll, tail) == 5 && fid == id && coll == 1 && strcmp(tail, "bin") == 0) {
            w = fw; h = fh; found = true;
            break;
        }
    }
    closedir(d);
    return found;
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s <scene.nvm | bundle.rd.out> <data directory> [neighbors=10] [diffusion=0] [output folder]\n", argv[0]); return 2; }
    const std::string nvm = argv[1], data_dir = argv[2];
    const int neighbors = argc > 3 ? atoi(argv[3]) : 10;
    const bool diffusion = argc > 4 && atoi(argv[4]) != 0;
    const std::string out_dir = argc > 5 ? argv[5] : data_dir;

    l3d_sfm_scene* scene = nullptr;
    // a bundler file (bundle.rd.out, main_bundler.cpp:110-204) is read just as well: the rest of the two drivers is the same flow
    const bool is_nvm = nvm.size() >= 4 && nvm.compare(nvm.size() - 4, 4, ".nvm") == 0;
    if ((is_nvm ? l3d_sfm_read_nvm(nvm.c_str(), &scene) : l3d_sfm_read_bundler(nvm.c_str(), &scene)) != L3D_OK) {
        fprintf(stderr, "%s\n", l3d_sfm_last_error(scene));
        l3d_sfm_free(scene);
        return 1;
    }
    const int n = l3d_sfm_num_cameras(scene);
    // Line3D(data_directory, matchingNeighbors, ...) with the driver's defaults (main_vsfm.cpp:60-99)
    L3D::Line3D line3D(data_dir, neighbors, 5.0f, 1.0f, 3.5f, 10.0f, 0.25f, true, true);
    if (!line3D.valid()) { l3d_sfm_free(scene); return 1; }
    int added = 0;
    for (int i = 0; i < n; ++i) {
        double focal = 0, dist[2] = { 0, 0 }, R[9], t[3];
        int nwp = 0;
        l3d_sfm_camera(scene, i, &focal, dist, R, t, &nwp);
        if (dist[0] != 0.0 || dist[1] != 0.0) { fprintf(stderr, "camera %d has lens distortion: the cached segments must come from undistorted images\n", i); continue; }
        unsigned w = 0, h = 0;
        if (!find_cache(data_dir, (unsigned)i, w, h)) { fprintf(stderr, "camera %d: no segment cache in %s\n", i, data_dir.c_str()); continue; }
        double K[9];
        l3d_sfm_intrinsics(focal, w, h, K);
        std::vector<uint32_t> ids((size_t)nwp);
        l3d_sfm_camera_worldpoints(scene, i, ids.data());
        std::list<unsigned int> wps(ids.begin(), ids.end());
        if (line3D.addImageFromCache((unsigned)i, w, h, K, R, t, wps)) ++added;
    }
    l3d_sfm_free(scene);
    fprintf(stderr, "[L3D] %d of %d cameras added\n", added, n);
    line3D.compute3Dmodel(diffusion);
    std::list<L3D::L3DFinalLine3D> result;
    line3D.getResult(result);
    fprintf(stderr, "[L3D] %zu 3-D lines\n", result.size());

    // the driver's output name (main_vsfm.cpp:289-313), numbers in stream-default formatting
    std::stringstream name;
    name << out_dir << "/line3D_result__W_" << -1 << "__";
    if (neighbors < 0) name << "N_ALL__"; else name << "N_" << neighbors << "__";
    name << "tL_" << 1.0f << "__tU_" << 5.0f << "__sigmaP_" << 3.5f << "__sigmaA_" << 10.0f << "__COLLIN__" << (diffusion ? "DIFFUSION" : "NO_DIFFUSION");
    line3D.save3DLinesAsSTL(result, name.str() + ".stl");                       // line3D.h:88-91
    line3D.save3DLinesAsTXT(result, name.str() + ".txt");
    return result.empty() ? 3 : 0;
}
