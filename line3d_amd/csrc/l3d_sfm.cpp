// l3d_sfm.cpp -- SfM front ends of the reference's two drivers without tclap / OpenCV / boost:
// the VisualSfM NVM reader of main_vsfm.cpp:121-223 and the bundler reader of main_bundler.cpp:110-204, reduced to what
// feeds Line3D::addImage: per camera focal length, rotation, translation, distortion coefficients and the list of
// world points it observes (the similarity source of findVisualNeighbors, line3D.cc:1874-1935).  Image decoding,
// undistortion and the LSD detector stay outside (segments are inputs); a camera's K is built by the caller from the
// focal length and the image size the way the drivers do it (main_vsfm.cpp:232-241): [[f,0,w/2],[0,f,h/2],[0,0,1]].
//
// Parsing follows the drivers' own token order, including what they skip (header lines, the separator line before the
// point count in NVM files, colours, feature positions).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/line3d_amd.h"

struct l3d_sfm_scene {
    int n_cams = 0, n_points = 0;
    std::vector<double> focal, dist, R, t;          // n, 2n, 9n (row-major), 3n
    std::vector<std::string> names;                 // image file names (NVM) or "" (bundler: %08d.jpg by convention)
    std::vector<std::vector<uint32_t>> wps;         // per camera: world point ids in file order
    std::string err;
};

namespace {

int fail(l3d_sfm_scene* s, const std::string& m, l3d_sfm_scene** out)
{
    // the message survives in a scene object the caller can query and must free
    s->err = m;
    s->n_cams = 0;
    *out = s;
    return L3D_ERR_INVALID;
}

}  // namespace

extern "C" {

// main_vsfm.cpp:121-223
int l3d_sfm_read_nvm(const char* path, l3d_sfm_scene** out)
{
    if (!path || !out) return L3D_ERR_INVALID;
    l3d_sfm_scene* s = new l3d_sfm_scene();
    std::ifstream f(path);
    if (!f.is_open()) return fail(s, std::string("NVM file ") + path + " does not exist!", out);
    std::string line;
    std::getline(f, line);                                   // header ("NVM_V3 ...")
    std::getline(f, line);                                   // empty
    std::getline(f, line);
    {
        std::stringstream ss(line);
        unsigned int n = 0;
        ss >> n;
        if (n == 0) return fail(s, "No aligned cameras in NVM file!", out);
        s->n_cams = (int)n;
    }
    const size_t n = (size_t)s->n_cams;
    s->focal.assign(n, 0.0); s->dist.assign(2 * n, 0.0); s->R.assign(9 * n, 0.0); s->t.assign(3 * n, 0.0);
    s->names.assign(n, ""); s->wps.assign(n, {});
    for (size_t i = 0; i < n; ++i) {
        std::getline(f, line);
        std::stringstream ss(line);
        std::string name;
        double fl = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0, cx = 0, cy = 0, cz = 0, d = 0;
        ss >> name >> fl >> q3 >> q0 >> q1 >> q2;            // file order: w x y z
        ss >> cx >> cy >> cz >> d;
        s->names[i] = name;
        s->focal[i] = (double)(float)fl;                     // the driver keeps focal and distortion as float
        s->dist[2 * i] = (double)(float)d;
        double* R = &s->R[9 * i];
        R[0] = 1.0 - 2.0 * q1 * q1 - 2.0 * q2 * q2; R[1] = 2.0 * q0 * q1 - 2.0 * q2 * q3; R[2] = 2.0 * q0 * q2 + 2.0 * q1 * q3;
        R[3] = 2.0 * q0 * q1 + 2.0 * q2 * q3; R[4] = 1.0 - 2.0 * q0 * q0 - 2.0 * q2 * q2; R[5] = 2.0 * q1 * q2 - 2.0 * q0 * q3;
        R[6] = 2.0 * q0 * q2 - 2.0 * q1 * q3; R[7] = 2.0 * q1 * q2 + 2.0 * q0 * q3; R[8] = 1.0 - 2.0 * q0 * q0 - 2.0 * q1 * q1;
        double* t = &s->t[3 * i];                            // t = -R C, evaluated the way Eigen evaluates -R*C: (-R) * C
        for (int r = 0; r < 3; ++r) t[r] = (-R[3 * r]) * cx + (-R[3 * r + 1]) * cy + (-R[3 * r + 2]) * cz;
    }
    std::getline(f, line);                                   // separator
    std::getline(f, line);
    {
        std::stringstream ss(line);
        unsigned int np = 0;
        ss >> np;
        s->n_points = (int)np;
    }
    for (int i = 0; i < s->n_points; ++i) {
        if (!std::getline(f, line)) break;
        std::istringstream ss(line);
        double px, py, pz, cr, cg, cb;
        ss >> px >> py >> pz >> cr >> cg >> cb;
        unsigned int nv = 0;
        ss >> nv;
        for (unsigned int j = 0; j < nv; ++j) {
            unsigned int cam = 0, sift = 0;
            float x = 0, y = 0;
            ss >> cam >> sift >> x >> y;
            if (!ss) break;
            if (cam < (unsigned int)s->n_cams) s->wps[cam].push_back((uint32_t)i);
        }
    }
    *out = s;
    return L3D_OK;
}

// main_bundler.cpp:110-204 (bundle.rd.out)
int l3d_sfm_read_bundler(const char* path, l3d_sfm_scene** out)
{
    if (!path || !out) return L3D_ERR_INVALID;
    l3d_sfm_scene* s = new l3d_sfm_scene();
    std::ifstream f(path);
    if (!f.is_open()) return fail(s, std::string("bundle file ") + path + " does not exist!", out);
    std::string line;
    std::getline(f, line);                                   // "# Bundle file v0.3"
    std::getline(f, line);
    {
        std::stringstream ss(line);
        unsigned int nc = 0, np = 0;
        ss >> nc >> np;
        if (nc == 0 || np == 0) return fail(s, "No cameras and/or points in bundle file!", out);
        s->n_cams = (int)nc; s->n_points = (int)np;
    }
    const size_t n = (size_t)s->n_cams;
    s->focal.assign(n, 0.0); s->dist.assign(2 * n, 0.0); s->R.assign(9 * n, 0.0); s->t.assign(3 * n, 0.0);
    s->names.assign(n, ""); s->wps.assign(n, {});
    for (size_t i = 0; i < n; ++i) {
        double fl = 0, d1 = 0, d2 = 0;
        std::getline(f, line);
        { std::stringstream ss(line); ss >> fl >> d1 >> d2; }
        s->focal[i] = (double)(float)fl;
        s->dist[2 * i] = (double)(float)d1; s->dist[2 * i + 1] = (double)(float)d2;
        double* R = &s->R[9 * i];
        for (int r = 0; r < 3; ++r) {
            std::getline(f, line);
            std::stringstream ss(line);
            ss >> R[3 * r] >> R[3 * r + 1] >> R[3 * r + 2];
        }
        for (int k = 3; k < 9; ++k) R[k] *= -1.0;            // bundler looks down -z: flip the 2nd and 3rd row ...
        std::getline(f, line);
        double* t = &s->t[3 * i];
        { std::stringstream ss(line); ss >> t[0] >> t[1] >> t[2]; }
        t[1] *= -1.0; t[2] *= -1.0;                          // ... and y, z of the translation
        char nm[32];
        snprintf(nm, sizeof nm, "%08u", (unsigned)i);        // visualize/%08d.{jpg,png,...}, main_bundler.cpp:208-236
        s->names[i] = nm;
    }
    for (int i = 0; i < s->n_points; ++i) {
        std::getline(f, line);                               // position
        std::getline(f, line);                               // colour
        if (!std::getline(f, line)) break;                   // view list
        std::istringstream ss(line);
        unsigned int nv = 0;
        ss >> nv;
        for (unsigned int j = 0; j < nv; ++j) {
            unsigned int cam = 0, key = 0;
            float x = 0, y = 0;
            ss >> cam >> key >> x >> y;
            if (!ss) break;
            if (cam < (unsigned int)s->n_cams) s->wps[cam].push_back((uint32_t)i);
        }
    }
    *out = s;
    return L3D_OK;
}

void l3d_sfm_free(l3d_sfm_scene* s) { delete s; }
const char* l3d_sfm_last_error(const l3d_sfm_scene* s) { return s ? s->err.c_str() : "null scene"; }
int l3d_sfm_num_cameras(const l3d_sfm_scene* s) { return s ? s->n_cams : 0; }
int l3d_sfm_num_points(const l3d_sfm_scene* s) { return s ? s->n_points : 0; }

int l3d_sfm_camera(const l3d_sfm_scene* s, int i, double* focal, double dist[2], double R[9], double t[3], int* n_worldpoints)
{
    if (!s || i < 0 || i >= s->n_cams) return L3D_ERR_INVALID;
    if (focal) *focal = s->focal[(size_t)i];
    if (dist) { dist[0] = s->dist[2 * (size_t)i]; dist[1] = s->dist[2 * (size_t)i + 1]; }
    if (R) memcpy(R, &s->R[9 * (size_t)i], 72);
    if (t) memcpy(t, &s->t[3 * (size_t)i], 24);
    if (n_worldpoints) *n_worldpoints = (int)s->wps[(size_t)i].size();
    return L3D_OK;
}
const char* l3d_sfm_camera_name(const l3d_sfm_scene* s, int i) { return (s && i >= 0 && i < s->n_cams) ? s->names[(size_t)i].c_str() : ""; }
int l3d_sfm_camera_worldpoints(const l3d_sfm_scene* s, int i, uint32_t* ids)
{
    if (!s || i < 0 || i >= s->n_cams || !ids) return L3D_ERR_INVALID;
    const std::vector<uint32_t>& w = s->wps[(size_t)i];
    if (!w.empty()) memcpy(ids, w.data(), w.size() * 4);
    return L3D_OK;
}

// The drivers' camera matrix (main_vsfm.cpp:232-241, main_bundler.cpp:243-252): principal point = image size / 2 in float
void l3d_sfm_intrinsics(double focal, unsigned int width, unsigned int height, double K[9])
{
    const float px = float(width) / 2.0f, py = float(height) / 2.0f, f = (float)focal;
    for (int k = 0; k < 9; ++k) K[k] = 0.0;
    K[0] = f; K[4] = f; K[2] = px; K[5] = py; K[8] = 1.0;
}

}  // extern "C"
