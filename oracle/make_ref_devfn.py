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
          ("cudawrapper.cu", 165, 285), ("cudawrapper.cu", 337, 344), ("cudawrapper.cu", 716, 829)]


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
    for name, a, b in RANGES:
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
