// l3d_unproject.hpp -- L3DView::unprojectSegment (view.cc:302-342) in double, one sequence of operations for the host pipeline
// (greedy selection on host lists) and the device (greedy selection on the resident products, l3d_products.hip): same bits.
#pragma once

#include "l3d_linalg.hpp"

namespace l3d {

L3D_LA_HD inline void unproject_segment_f64(const la::M3& RtKinv, la::V3 C, float x1, float y1, float x2, float y2, float d1, float d2,
                                            la::V3& P1, la::V3& P2, la::V3& dir)
{
    la::V3 r1 = la::mul(RtKinv, la::V3{ (double)x1, (double)y1, 1.0 });
    r1 = r1 / la::norm(r1);
    la::V3 r2 = la::mul(RtKinv, la::V3{ (double)x2, (double)y2, 1.0 });
    r2 = r2 / la::norm(r2);
    P1 = C + r1 * (double)d1;
    P2 = C + r2 * (double)d2;
    dir = P2 - P1;
    dir = dir / la::norm(dir);
}

}  // namespace l3d
