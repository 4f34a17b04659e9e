// l3d_runtable.hpp -- run tables of kept lists (round 6).  A view's kept list is ordered (segment, LOCAL camera, target) by construction; its run
// table rt[(N + 1)][S] (rows S apart) holds, per local camera q and segment s, the position in the list of the first record of (s, q) -- row N = the
// end of segment s -- and its side array qt one word per record: (local camera << 16) | target segment.  The chain's kept writer fills both
// (l3d_kept.hpp); lists that did not come out of it (views taken over from another rank, the sharded chain's retired arena) get them rebuilt here.
#pragma once

#include <hip/hip_runtime.h>

#include "l3d_kernels.hpp"

namespace l3d {

struct RtJob {                      // one view's list
    const Match* recs;
    unsigned* qt;                   // out: n words
    int* rt;                        // out: (N + 1) * S ints
    const unsigned* ids;            // the view's neighbours' global ids, ascending
    const int* qs;                  // ... and their local camera numbers
    int n, S, N, pad;
    unsigned* skey;                 // scratch, n words: (segment << 8 | local camera) per record -- what the run table's lower bounds search (4 contiguous bytes per probe
                                    // instead of a 32-byte record and a side word)
};

// qt of every job's records; *err counts records whose camera is not a neighbour or whose (segment, camera) order descends (the tables would be wrong)
void launch_qt_from_records(const RtJob* jobs_dev, int n_jobs, int max_n, int* err, hipStream_t st);
// rt of every job from its records and qt (lower bounds: one thread per (segment, camera) cell)
void launch_rt_from_qt(const RtJob* jobs_dev, int n_jobs, int max_cells, hipStream_t st);

}  // namespace l3d
