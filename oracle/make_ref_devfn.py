#!/usr/bin/env python3
"""oracle/make_ref_devfn.py -- TEST INFRASTRUCTURE.  Builds oracle/_ref/libdevfn_ref.so: the reference's own texture-free
device functions, compiled from the sources where they lie under /root/reference against the genuine NVIDIA runtime
headers bundled with this image's triton wheel (cuda_runtime.h, device_launch_parameters.h, math_constants.h; helper_math.h of the reference is used
unmodified).  The translation unit is assembled in memory from line ranges of cudawrapper.h / cudawrapper.cu and piped to
g++ -- no reference text is written into the repository, no stand-in header is involved; only the extern "C" door
(ref_devfn_door.cc) is ours.  Exit code 0 and a message when the reference checkout or the headers are absent (the GPU box
uses the prebuilt file)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("REF", "/root/reference")
# (file, first line, last line): cudawrapper.h:43-46 constants; cudawrapper.cu: D_distance_p2l_2D_f3, D_segment_length_2D_f3,
# D_angle_between_lines_deg_3D_f3 + D_point_on_segment_2D_f3, D_segment_overlap_2D + D_normalize_hom_coords_2D + D_get_ray_src,
# D_unproject_point_src; the two texture-free kernels of replicator_dynamics_diffusion: K_sparseMat_row_normalization,
# K_sparseMat_diffusion_step (their launch variables -- threadIdx & co., declared by the genuine <device_launch_parameters.h> -- get
# their storage from ref_devfn_launch.cc, a second translation unit of ours)
RANGES = [("cudawrapper.h", 43, 46), ("cudawrapper.cu", 56, 61), ("cudawrapper.cu", 93, 99), ("cudawrapper.cu", 116, 141),
          ("cudawrapper.cu", 165, 285), ("cudawrapper.cu", 337, 344), ("cudawrapper.cu", 716, 829),
          # K_collinearity's body for one pair of segments: the reference's own lines behind the four texture fetches (:492 `result`, :499 and
          # :506 the two lines, :508-529 distances, affinity, overlap check) inside a function whose parameters stand for the fetched points --
          # the two lines of text below are ours, everything between them is the reference's
          ("text", "static float l3dref_collinearity_body(const float3 p1, const float3 p2, const float3 q1, const float3 q2, const float coll_sigma_sqr)\n{"),
          ("cudawrapper.cu", 492, 492), ("cudawrapper.cu", 499, 499), ("cudawrapper.cu", 506, 506), ("cudawrapper.cu", 508, 529),
          ("text", "    return result;\n}"),
          # D_hypothesis_confidence (:380-427) without its one texture fetch (:407): the fetched target segment is a parameter
          ("text", "static float l3dref_hypothesis_confidence_body(const float3 p1, const float3 p2, const float3 P1, const float3 P2, const float3 Q1, const float3 Q2,\n"
                   "                                                 const float3 C, const float4 data, const float sigma_p, const float sigma_a, const float spatial_k)\n{"),
          ("cudawrapper.cu", 387, 406), ("cudawrapper.cu", 408, 427),
          # the middle of K_pairwise_matches (:548 `result`, :555 / :561 the two lines, :569-589 intersections, validity with the early exit,
          # overlaps, the threshold test and its opening brace): points and epipolar lines are parameters; `buffer`, `x`, `y`, `stride` exist
          # because the early exit (:579-580) stores `result` before it returns
          ("text", "static void l3dref_pairwise_overlap_body(const float3 p1, const float3 p2, const float3 q1, const float3 q2, const float3 epi_p1, const float3 epi_p2,\n"
                   "                                         const float3 epi_q1, const float3 epi_q2, float* out13)\n{\n"
                   "    float4 buffer[1]; const int x = 0, y = 0, stride = 0;\n    for (int k_ = 0; k_ < 13; ++k_) out13[k_] = 0.0f;"),
          ("cudawrapper.cu", 548, 548), ("cudawrapper.cu", 555, 555), ("cudawrapper.cu", 561, 561), ("cudawrapper.cu", 569, 589),
          ("text", "        out13[0] = 1.0f;\n        out13[1] = l2_p1.x; out13[2] = l2_p1.y; out13[3] = l2_p1.z; out13[4] = l2_p2.x; out13[5] = l2_p2.y; out13[6] = l2_p2.z;\n"
                   "        out13[7] = l1_q1.x; out13[8] = l1_q1.y; out13[9] = l1_q1.z; out13[10] = l1_q2.x; out13[11] = l1_q2.y; out13[12] = l1_q2.z;\n"
                   "    }\n    (void)buffer;\n}")]


def nvidia_include():
    try:
        import triton
        p = os.path.join(os.path.dirname(triton.__file__), "backends", "nvidia", "include")
    except Exception:       # noqa: BLE001
        p = "/usr/local/lib/python3.10/dist-packages/triton/backends/nvidia/include"
    return p if all(os.path.exists(os.path.join(p, f)) for f in ("cuda_runtime.h", "device_launch_parameters.h", "math_constants.h")) else None


def main():
    inc = nvidia_include()
    if not os.path.exists(os.path.join(REF, "cudawrapper.cu")) or inc is None:
        print("reference checkout or NVIDIA runtime headers absent: keeping prebuilt oracle/_ref/libdevfn_ref.so (if any)")
        return 0
    tu = ['#include <cuda_runtime.h>\n#include <device_launch_parameters.h>\n#include <math_constants.h>\n#include "helper_math.h"\nnamespace L3D {\n']
    for name, a, b in [(r + (None,))[:3] for r in RANGES]:
        if name == "text":
            tu.append(a + "\n")
            continue
        with open(os.path.join(REF, name)) as f:
            lines = f.read().split("\n")
        tu.append("\n".join(lines[a - 1:b]) + "\n")
    tu.append('}  // namespace L3D\n#include "ref_devfn_door.cc"\n')
    out_dir = os.path.join(HERE, "_ref")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libdevfn_ref.so")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-fPIC", "-shared", "-std=c++11", "-ffp-contract=off", "-fno-fast-math", "-I" + inc, "-I" + REF,
           "-I" + HERE, os.path.join(HERE, "ref_devfn_launch.cc"), "-x", "c++", "-", "-o", out]
    p = subprocess.run(cmd, input="".join(tu).encode(), capture_output=True)
    sys.stderr.write(p.stderr.decode())
    if p.returncode == 0:
        print("built oracle/_ref/libdevfn_ref.so")
    return p.returncode


if __name__ == "__main__":
    sys.exit(main())
