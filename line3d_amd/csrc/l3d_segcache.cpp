// l3d_segcache.cpp -- the segment cache of Line3D::addImage without boost.
//
// The reference keeps the detected segments of an image in "<data dir>/segments_<id>_<w>x<h>_coll<0|1>.bin"
// (line3D.cc:143-150) and reads them back instead of running the detector (line3D.cc:160-168, 285-293).  The file is a
// boost::archive::binary_oarchive holding ONE object, `ar & make_nvp("data", L3DSegments const&)` (serialization.h:49-56);
// L3DSegments::serialize writes the collinearity map, then the DataArray<float> POINTER (segments.h:124-131);
// DataArray::serialize writes seven scalars and the raw padded rows (dataArray.h:296-318).
//
// boost is a third-party dependency of the reference that is absent here, and the reference pins no version
// (CMakeLists.txt: find_package(Boost COMPONENTS serialization filesystem ...)).  What follows restates the native
// binary archive layout of boost.serialization as its headers define it for archive library versions 9 and later
// (boost >= 1.44; the reference is from 2015: boost 1.54-1.58 write versions 10-12), x86-64 little endian:
//
//   header     u64 22, "serialization::archive"                      basic_binary_oarchive::init
//              u16 library version
//              u8 sizeof(int)=4, u8 sizeof(long)=8, u8 sizeof(float)=4, u8 sizeof(double)=8, i32 1   basic_binary_oprimitive::init
//   object     saved by reference, class-info level: the FIRST object of a class is preceded by
//              u8 tracking flag, u32 class version (class_id_optional is not written by binary archives); a tracked
//              object (one whose class is also saved through a pointer) is followed by u32 object id
//   std::map   (an object like any other: its first instance carries the 5-byte preamble) u64 count, u32 item version,
//              then the items, each a std::pair object (preamble on the first one of each pair type): first, second
//   pointer    i16 class id (-1 = null pointer); the first pointer of a class: u8 tracking flag, u32 class version;
//              u32 object id when tracked; then the object's data
//   scalars    native width; make_array(float*, n): the n floats back to back, no count
//
// NOT validated against a file written by boost (none exists here, the reference holds no sample): parity unpinned.  The
// reader therefore checks everything the layout determines -- signature, the native-size bytes, flags that must be 0/1,
// pitch/stride against the widths, and that the payload ends exactly at the end of the file -- and refuses a file that
// deviates instead of guessing.  The writer exists for the round-trip tests and for handing segments to a reference build.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <fstream>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/line3d_amd.h"

struct l3d_segment_cache {
    int library_version = 0;
    std::vector<float> segs;                    // n x 4
    std::vector<int32_t> ci, cj;                // directed collinearity entries, ascending (i, j) (std::map order)
    std::vector<float> cw;
    std::string err;
};

namespace {

const char kSignature[] = "serialization::archive";

struct Reader {
    const std::vector<unsigned char>& b;
    size_t pos = 0;
    bool ok = true;
    explicit Reader(const std::vector<unsigned char>& bytes) : b(bytes) {}
    template <typename T> T get()
    {
        T v{};
        if (!ok || b.size() - pos < sizeof(T)) { ok = false; return v; }
        memcpy(&v, b.data() + pos, sizeof(T));
        pos += sizeof(T);
        return v;
    }
    // the preamble in front of the first object of a class saved by reference; tracked: is an object id expected?
    bool class_preamble(bool& tracked)
    {
        const uint8_t t = get<uint8_t>();
        const uint32_t version = get<uint32_t>();
        if (!ok || t > 1 || version != 0) return false;     // none of the classes involved declares a version
        tracked = t == 1;
        return true;
    }
};

int fail(l3d_segment_cache* s, const std::string& m, l3d_segment_cache** out)
{
    s->err = m;
    s->segs.clear(); s->ci.clear(); s->cj.clear(); s->cw.clear();
    *out = s;
    return L3D_ERR_INVALID;
}

template <typename T> void put(std::vector<unsigned char>& o, T v)
{
    const unsigned char* p = reinterpret_cast<const unsigned char*>(&v);
    o.insert(o.end(), p, p + sizeof(T));
}

}  // namespace

extern "C" {

// line3D.cc:143-150
int l3d_segment_cache_filename(uint32_t image_id, unsigned int width, unsigned int height, int use_collinearity, char* out, size_t out_size)
{
    if (!out || out_size == 0) return L3D_ERR_INVALID;
    const int n = snprintf(out, out_size, "/segments_%u_%ux%u_coll%d.bin", image_id, width, height, use_collinearity ? 1 : 0);
    return n > 0 && (size_t)n < out_size ? L3D_OK : L3D_ERR_INVALID;
}

// serializeFromFile<L3DSegments> (serialization.h:58-69)
static int segment_cache_read_into(const char* path, l3d_segment_cache* s, l3d_segment_cache** out);
int l3d_segment_cache_read(const char* path, l3d_segment_cache** out)
{
    if (!out) return L3D_ERR_INVALID;
    *out = nullptr;
    l3d_segment_cache* s = new (std::nothrow) l3d_segment_cache();
    if (!s) return L3D_ERR_NOMEM;
    // no exception leaves an extern "C" function: a file whose size (or whose counts) cannot be allocated is refused with a message
    try {
        return segment_cache_read_into(path, s, out);
    } catch (const std::exception&) {           // (std::bad_alloc / std::length_error of a resize)
        s->segs.clear(); s->ci.clear(); s->cj.clear(); s->cw.clear();
        try { s->err = "segment cache: out of memory while reading the file"; } catch (...) {}
        *out = s;
        return L3D_ERR_NOMEM;
    }
}
static int segment_cache_read_into(const char* path, l3d_segment_cache* s, l3d_segment_cache** out)
{
    if (!path) return fail(s, "null path", out);
    std::vector<unsigned char> bytes;
    {
        std::ifstream f(path, std::ios::binary);
        if (!f.is_open()) return fail(s, std::string("segment cache ") + path + " could not be opened", out);
        f.seekg(0, std::ios::end);
        const std::streamoff len = f.tellg();
        f.seekg(0, std::ios::beg);
        if (len < 0) return fail(s, "segment cache: cannot determine the file size", out);
        bytes.resize((size_t)len);
        if (len > 0) f.read(reinterpret_cast<char*>(bytes.data()), len);
        if (!f) return fail(s, "segment cache: short read", out);
    }
    Reader r(bytes);
    // ---- archive header
    if (r.get<uint64_t>() != sizeof(kSignature) - 1 || !r.ok || bytes.size() - r.pos < sizeof(kSignature) - 1 ||
        memcmp(bytes.data() + r.pos, kSignature, sizeof(kSignature) - 1) != 0)
        return fail(s, "segment cache: not a boost binary archive (signature)", out);
    r.pos += sizeof(kSignature) - 1;
    const uint16_t lib = r.get<uint16_t>();
    if (!r.ok || lib < 9 || lib > 64)
        return fail(s, "segment cache: archive library version " + std::to_string(lib) + " is outside the supported range (9 and later: boost >= 1.44)", out);
    s->library_version = lib;
    const uint8_t si = r.get<uint8_t>(), sl = r.get<uint8_t>(), sf = r.get<uint8_t>(), sd = r.get<uint8_t>();
    const int32_t one = r.get<int32_t>();
    if (!r.ok || si != 4 || sl != 8 || sf != 4 || sd != 8 || one != 1)
        return fail(s, "segment cache: written on a platform with other native sizes or byte order (expected int 4, long 8, float 4, double 8, little endian)", out);
    // ---- L3DSegments (by reference, never through a pointer: untracked)
    bool tracked = false;
    if (!r.class_preamble(tracked) || tracked) return fail(s, "segment cache: unexpected class preamble of L3DSegments", out);
    // ---- segment2collinearities_: std::map<unsigned, std::map<unsigned, float>>
    if (!r.class_preamble(tracked) || tracked) return fail(s, "segment cache: unexpected class preamble of the collinearity map", out);
    const uint64_t n_outer = r.get<uint64_t>();
    if (r.get<uint32_t>() != 0 || !r.ok) return fail(s, "segment cache: unexpected item version of the collinearity map", out);
    if (n_outer > bytes.size()) return fail(s, "segment cache: collinearity map count exceeds the file size", out);
    bool seen_outer_pair = false, seen_inner_map = false, seen_inner_pair = false;
    for (uint64_t a = 0; a < n_outer; ++a) {
        if (!seen_outer_pair) { if (!r.class_preamble(tracked) || tracked) return fail(s, "segment cache: unexpected class preamble of a map item", out); seen_outer_pair = true; }
        const uint32_t i = r.get<uint32_t>();
        if (!seen_inner_map) { if (!r.class_preamble(tracked) || tracked) return fail(s, "segment cache: unexpected class preamble of an inner map", out); seen_inner_map = true; }
        const uint64_t n_inner = r.get<uint64_t>();
        if (r.get<uint32_t>() != 0 || !r.ok) return fail(s, "segment cache: unexpected item version of an inner map", out);
        if (n_inner > bytes.size()) return fail(s, "segment cache: inner map count exceeds the file size", out);
        for (uint64_t q = 0; q < n_inner; ++q) {
            if (!seen_inner_pair) { if (!r.class_preamble(tracked) || tracked) return fail(s, "segment cache: unexpected class preamble of an inner map item", out); seen_inner_pair = true; }
            const uint32_t j = r.get<uint32_t>();
            const float w = r.get<float>();
            if (!r.ok) return fail(s, "segment cache: truncated inside the collinearity map", out);
            if (i > 0x7fffffffu || j > 0x7fffffffu) return fail(s, "segment cache: segment index out of range in the collinearity map", out);
            s->ci.push_back((int32_t)i); s->cj.push_back((int32_t)j); s->cw.push_back(w);
        }
        if (!r.ok) return fail(s, "segment cache: truncated inside the collinearity map", out);
    }
    // ---- segments_: DataArray<float>* (null in a default-constructed L3DSegments, segments.h:62-64)
    const int16_t class_id = r.get<int16_t>();
    if (!r.ok) return fail(s, "segment cache: truncated in front of the segment array", out);
    size_t n_seg = 0;
    if (class_id != -1) {
        if (class_id < 0) return fail(s, "segment cache: negative class id of the segment array", out);
        if (!r.class_preamble(tracked)) return fail(s, "segment cache: unexpected class preamble of the segment array", out);
        if (tracked && r.get<uint32_t>() != 0) return fail(s, "segment cache: the segment array is not the first tracked object", out);
        const uint32_t width = r.get<uint32_t>(), height = r.get<uint32_t>(), real_width = r.get<uint32_t>();
        const uint64_t pitch_cpu = r.get<uint64_t>(), stride_cpu = r.get<uint64_t>();
        (void)r.get<uint64_t>(); (void)r.get<uint64_t>();                  // pitchGPU_, strideGPU_: reset on load (dataArray.h:309-315)
        if (!r.ok) return fail(s, "segment cache: truncated inside the segment array header", out);
        // DataArray<float>(4, n) (segments.h:70, dataArray.h:66-95): rows padded to a multiple of 32 bytes
        if (width != 4 || real_width < width || stride_cpu != real_width || pitch_cpu != (uint64_t)real_width * 4)
            return fail(s, "segment cache: segment array is not a 4-column float array (width " + std::to_string(width) + ", row " + std::to_string(real_width) + ")", out);
        const uint64_t left = bytes.size() - r.pos, row_bytes = (uint64_t)real_width * 4;
        const uint64_t payload = height <= left / row_bytes ? row_bytes * height : ~(uint64_t)0;      // (no overflow: rows that cannot be there)
        if (left != payload)
            return fail(s, "segment cache: " + std::to_string(bytes.size() - r.pos) + " bytes left for " + std::to_string(height) + " rows of " + std::to_string(real_width) + " floats", out);
        n_seg = height;
        s->segs.resize(n_seg * 4);
        for (size_t y = 0; y < n_seg; ++y) memcpy(&s->segs[4 * y], bytes.data() + r.pos + y * (size_t)real_width * 4, 16);
        r.pos += (size_t)payload;
    }
    if (r.pos != bytes.size()) return fail(s, "segment cache: " + std::to_string(bytes.size() - r.pos) + " trailing bytes", out);
    for (size_t k = 0; k < s->ci.size(); ++k)
        if ((size_t)s->ci[k] >= n_seg || (size_t)s->cj[k] >= n_seg) return fail(s, "segment cache: collinearity entry names a segment that does not exist", out);
    *out = s;
    return L3D_OK;
}

void l3d_segment_cache_free(l3d_segment_cache* s) { delete s; }
const char* l3d_segment_cache_last_error(const l3d_segment_cache* s) { return s ? s->err.c_str() : "null cache"; }
int l3d_segment_cache_num_segments(const l3d_segment_cache* s) { return s ? (int)(s->segs.size() / 4) : 0; }
int l3d_segment_cache_num_collinearities(const l3d_segment_cache* s) { return s ? (int)s->ci.size() : 0; }
int l3d_segment_cache_library_version(const l3d_segment_cache* s) { return s ? s->library_version : 0; }

int l3d_segment_cache_get(const l3d_segment_cache* s, float* segments, int32_t* ci, int32_t* cj, float* cw)
{
    if (!s) return L3D_ERR_INVALID;
    if (segments && !s->segs.empty()) memcpy(segments, s->segs.data(), s->segs.size() * 4);
    if (ci && !s->ci.empty()) memcpy(ci, s->ci.data(), s->ci.size() * 4);
    if (cj && !s->cj.empty()) memcpy(cj, s->cj.data(), s->cj.size() * 4);
    if (cw && !s->cw.empty()) memcpy(cw, s->cw.data(), s->cw.size() * 4);
    return L3D_OK;
}

// serializeToFile<L3DSegments> (serialization.h:49-56) of L3DSegments(list<float4>&, collin) (segments.h:67-101).
// ci/cj/cw: the DIRECTED entries of segment2collinearities_ (both (i,j) and (j,i) as the constructor inserts them), any order.
int l3d_segment_cache_write(const char* path, const float* segments, int n_segments, const int32_t* ci, const int32_t* cj, const float* cw,
                            int n_coll, int library_version)
{
    if (!path || n_segments < 0 || n_coll < 0 || (n_segments > 0 && !segments) || (n_coll > 0 && (!ci || !cj || !cw))) return L3D_ERR_INVALID;
    if (library_version < 9 || library_version > 64) return L3D_ERR_INVALID;
    std::vector<size_t> order((size_t)n_coll);
    for (size_t k = 0; k < order.size(); ++k) order[k] = k;
    for (int k = 0; k < n_coll; ++k) if (ci[k] < 0 || cj[k] < 0 || ci[k] >= n_segments || cj[k] >= n_segments) return L3D_ERR_INVALID;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return ci[a] != ci[b] ? ci[a] < ci[b] : cj[a] < cj[b]; });
    for (size_t k = 1; k < order.size(); ++k) if (ci[order[k]] == ci[order[k - 1]] && cj[order[k]] == cj[order[k - 1]]) return L3D_ERR_INVALID;   // a map holds a key once

    std::vector<unsigned char> o;
    put<uint64_t>(o, sizeof(kSignature) - 1);
    o.insert(o.end(), kSignature, kSignature + sizeof(kSignature) - 1);
    put<uint16_t>(o, (uint16_t)library_version);
    put<uint8_t>(o, 4); put<uint8_t>(o, 8); put<uint8_t>(o, 4); put<uint8_t>(o, 8); put<int32_t>(o, 1);
    auto preamble = [&](uint8_t tracked) { put<uint8_t>(o, tracked); put<uint32_t>(o, 0); };
    preamble(0);                                                            // L3DSegments
    preamble(0);                                                            // outer map
    size_t n_outer = 0;
    for (size_t k = 0; k < order.size(); ++k) n_outer += k == 0 || ci[order[k]] != ci[order[k - 1]];
    put<uint64_t>(o, n_outer); put<uint32_t>(o, 0);
    bool first_outer = true, first_inner = true;
    for (size_t k = 0; k < order.size();) {
        size_t e = k;
        while (e < order.size() && ci[order[e]] == ci[order[k]]) ++e;
        if (first_outer) preamble(0);                                       // pair<const unsigned, map>
        put<uint32_t>(o, (uint32_t)ci[order[k]]);
        if (first_outer) preamble(0);                                       // inner map
        first_outer = false;
        put<uint64_t>(o, e - k); put<uint32_t>(o, 0);
        for (size_t q = k; q < e; ++q) {
            if (first_inner) { preamble(0); first_inner = false; }          // pair<const unsigned, float>
            put<uint32_t>(o, (uint32_t)cj[order[q]]);
            put<float>(o, cw[order[q]]);
        }
        k = e;
    }
    // the pointer: class ids count the classes in the order the archive met them
    put<int16_t>(o, (int16_t)(order.empty() ? 2 : 5));
    preamble(1);                                                            // DataArray<float>: tracked (saved through a pointer)
    put<uint32_t>(o, 0);                                                    // object id
    const uint32_t real_width = 8;                                          // 4 floats = 16 bytes, padded to 32 (dataArray.h:74-84)
    put<uint32_t>(o, 4); put<uint32_t>(o, (uint32_t)n_segments); put<uint32_t>(o, real_width);
    put<uint64_t>(o, (uint64_t)real_width * 4); put<uint64_t>(o, real_width);
    put<uint64_t>(o, 0); put<uint64_t>(o, 0);                               // pitchGPU_, strideGPU_ of an array that is not on the GPU (segments.h:82)
    const size_t base = o.size();
    o.resize(base + (size_t)n_segments * real_width * 4, 0);
    for (int y = 0; y < n_segments; ++y) memcpy(&o[base + (size_t)y * real_width * 4], segments + 4 * (size_t)y, 16);

    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if (!f.is_open()) return L3D_ERR_INVALID;
    f.write(reinterpret_cast<const char*>(o.data()), (std::streamsize)o.size());
    f.close();
    return f ? L3D_OK : L3D_ERR_INVALID;
}

}  // extern "C"
