// l3d_options.hpp -- every diagnostic / A-B switch of the library in one place.  The environment is read ONCE, by
// l3d_ctx_create (l3d::options_from_env, l3d_capi.hip: the only getenv of the library); afterwards a switch changes only through
// l3d_set_option (tests, scripts).  Nothing here selects a CPU path for the arithmetic: the switches choose between device
// variants, turn on self-checks and timing prints, or force the small-capacity / many-pass code paths tests want to reach.
#pragma once

#include <atomic>
#include <cstring>

namespace l3d {

// X(field, environment variable, default, meaning)
#define L3D_OPTION_TABLE(X)                                                                                                          \
    X(timing, "L3D_TIMING", 0, "1: per-stage timing lines on stderr, 2: + per-view trace of the chain")                              \
    X(check_pot, "L3D_CHECK_POT", 0, "tests: compare the device products of matchViews with the plain host construction")            \
    X(chain_ring, "L3D_CHAIN_RING", 1, "0: triangulation on the chain's stream instead of the stage-1 ring (A/B)")                   \
    X(pretest, "L3D_PRETEST", 3, "stage-1 conservative filters: bit 0 wedge test, bit 1 overlap-bound test (A/B)")                   \
    X(mask_stream, "L3D_MASK_STREAM", 0, "1: k_pair_mask on a stream of its own, a view ahead of k_pair_fill (A/B, DESIGN.md section 6)")                       \
    X(prof_stride, "L3D_PROF_STRIDE", 1, "with one kernel bracketed (l3d_profile_only): HIP events around every n-th launch of it only -- a uniform sample of a timed region")   \
    X(event_fence, "L3D_EVENT_FENCE", 0, "1: the library's synchronisation events WITH system-scope fences (HIP's default; A/B: +0.27 ms per config-2 pass)")     \
    X(stream_prio, "L3D_STREAM_PRIO", 0, "1: the chain's stream at the highest priority (measured: no difference)")                  \
    X(pair_stats, "L3D_PAIR_STATS", 0, "device counters of k_pair_mask's levels, printed at destroy")                                \
    X(vw_debug, "L3D_VW_DEBUG", 0, "k_verify_window debug mode")                                                                     \
    X(vw_stamps, "L3D_VW_STAMPS", 0, "in-kernel clock stamps of k_verify_window's phases, printed at destroy")                       \
    X(vw_lds, "L3D_VW_LDS", 0, "dynamic LDS budget of k_verify_window in bytes (0: the measured default)")                           \
    X(vw_wide_max, "L3D_VW_WIDE_MAX", 640, "launches of up to this many segments use the 8-wave k_verify_window")                    \
    X(chain_serial, "L3D_CHAIN_SERIAL", 0, "1: the chain on one stream, kernels one at a time (isolated durations)")                 \
    X(tgt_rays, "L3D_TGT_RAYS", 1, "0: k_pair_fill normalises the target rays per candidate (A/B)")                                  \
    X(src_rays, "L3D_SRC_RAYS", 1, "0: k_pair_fill normalises the source rays per row (A/B)")                                        \
    X(depth_in_fill, "L3D_DEPTH_IN_FILL", 1, "0: depths triangulated in k_pair_mask (A/B)")                                          \
    X(fused_rows, "L3D_FUSED_ROWS", 1, "0: separate row-scan launch on the stage-1 stream (A/B)")                                    \
    X(pair_spb, "L3D_PAIR_SPB", 0, "source segments per k_pair_mask workgroup (0: by the size of the launch)")                       \
    X(wait_sleep_us, "L3D_WAIT_SLEEP_US", 20, "sleep of the sharded run's waiting host threads")                                     \
    X(aff_chunk, "L3D_AFF_CHUNK", 0, "tests: targets per pass of k_aff_groups (0: 64)")                                              \
    X(aff_per_view, "L3D_AFF_PER_VIEW", 0, "tests: one k_aff_groups launch per view (the schedule of one-way records)")              \
    X(aff_sym, "L3D_AFF_SYM", 1, "affinity fill: 1 = a symmetric collinearity table (checked) takes the short-list path, 0 = always the general path (A/B)") \
    X(aff_block, "L3D_AFF_BLOCK", 0, "candidate pairs per block of sources of the affinity fill (0: 2^27; tests: small values force many blocks)") \
    X(aff_word_block, "L3D_AFF_WORD_BLOCK", 0, "decision words per outer block of sources of the affinity fill (0: 2^26)")          \
    X(cc_max_rounds, "L3D_CC_MAX_ROUNDS", 0, "tests: rounds of the connected-components loop before it gives up (0: 64)")            \
    X(host_threads, "L3D_HOST_THREADS", 0, "worker threads of the host-side stages (0: min(16, usable CPUs))")                       \
    X(reserve_hint, "L3D_RESERVE_HINT", 1, "prepare(): 1 = the finishing stages' arenas are reserved ahead from the size of the scene (speed of the first finish), 0 = on demand (memory)") \
    X(handover_chunk_kb, "L3D_HANDOVER_CHUNK_KB", 262144, "views sharded in blocks: a missed block's sources travel in chunks of this many KB per all-gather slot (tests: small values force many chunks)") \
    X(kept_cams, "L3D_KEPT_CAMS", 1, "resident chain: 1 = the kept writer leaves every record's target camera in a side array and later views scan that (4 B per record) for their reverse matches, 0 = they scan the records (A/B)") \
    X(part_release, "L3D_PART_RELEASE", 1, "partitioned run: 1 = the chain's per-launch scratch is released before the products are built (memory before the speed of a second pass)") \
    X(block_recover, "L3D_BLOCK_RECOVER", 1, "views sharded in blocks: 1 = a block whose cold-started speculation failed is re-run warm from its predecessor's true lists, 0 = any miss ends the call with verdict 1 (round 4; A/B)") \
    X(slot_cams_min, "L3D_SLOT_CAMS_MIN", 65536, "sharded chain: slots of at least this many records carry a 4-byte side array of target cameras (0: always; tests)") \
    X(vw_gb, "L3D_VW_GB", 1, "k_verify_window at more than 16 neighbours: 1 = bucket starts in global memory during the rounds (four workgroups per CU at 24 neighbours), 0 = in LDS (A/B)") \
    X(exist_sort_apart, "L3D_EXIST_SORT_APART", -1, "sharded chain: reverse-match runs ranked by a launch of their own (one wave per run) instead of the segment's workgroup: 1 always, 0 never, -1 a rank's small launches on dense scenes") \
    X(vw_split, "L3D_VW_SPLIT", -1, "k_verify_window: long segments built by the first launch, verified in units by a second (k_vw_walk): 1 always, 0 never, -1 launches of few segments on dense scenes") \
    X(vw_unit, "L3D_VW_UNIT", 512, "hypotheses per unit of the split verification (a multiple of 256; emulated rank of eight at 64 x 4000 x 24: 256 -> 95.6, 512 -> 90.7, 1024 -> 92.4, 2048 -> 99.0 ms)")                               \
    X(vw_split_avg, "L3D_VW_SPLIT_AVG", 4096, "vw_split = -1: split when the candidate capacity per segment of the launch is at least this") \
    X(prod_block_keys, "L3D_PROD_BLOCK_KEYS", 0, "key slots per block of the products' construction (0: 2^30 entries transposed, 2^28 key slots sorted; tests: small values force many blocks)") \
    X(prod_transpose, "L3D_PROD_TRANSPOSE", 1, "matchViews' products: 1 = rows from run tables, per-pair LDS transposes and an LDS bitmap per row (round 6), 0 = radix sort of two 64-bit keys per record (A/B)") \
    X(prod_pair_stage, "L3D_PROD_PAIR_STAGE", 1, "transposed products: 1 = the pair transposes scatter in two levels (buckets of consecutive target segments in a staging region, then an LDS image written in whole lines), 0 = directly (A/B: 4 GB of partial-line write-backs per 0.6 GB of entries at 40 x 4000 x 24)") \
    X(prod_pair_g, "L3D_PROD_PAIR_G", -1, "transposed products: lanes sharing a run in the pair transposes (-1: by the average run, 0: a run per thread)") \
    X(prod_early, "L3D_PROD_EARLY", 1, "transposed products: the chain transposes its views' (view, camera) pairs on a side stream behind their kept writers; only the rows are left for the end of matchViews (1: for lists above 2^18 records per view; 2 / 3: always, a view / eight views per launch; 0: all at the end)") \
    X(retire_tables, "L3D_RETIRE_TABLES", 1, "sharded chain, ring mode: the retire kernel files the slots' side words and run tables with their records, the products transpose without rebuilding them (0: rebuilt from the records)") \
    X(retire_apart, "L3D_RETIRE_APART", 1, "sharded chain, ring mode: batches of views are retired into the compact arena on a side stream; the chain waits only before it overwrites their ring blocks (0: on the chain's stream)") \
    X(arena_guess, "L3D_ARENA_GUESS", 50, "resident chain: first guess of the kept arena in thousandths of the scene's segment pairs, within 35 % of the free HBM (4: the small guess of rounds 1-5, grown on overflow)") \
    X(prod_row_group, "L3D_PROD_ROW_GROUP", 1, "transposed products: bitmap words a group of touched views may fill together in the rows kernel (1: a view at a time, up to 512)") \
    X(slot_scan_grain, "L3D_SLOT_SCAN_GRAIN", 0, "sharded chain: records of a slot per workgroup of the two scans of the sources' slots (0: 4096)") \
    X(rt_place_lds, "L3D_RT_PLACE_LDS", 0, "resident chain with run tables: 1 = reverse matches placed by big workgroups with LDS cursors from the chunk bases (no global cursor; measured slower: too few workgroups), 0 = through the global row cursors (A/B)") \
    X(rt_g, "L3D_RT_G", -1, "resident chain with run tables: lanes sharing a run when a view collects its reverse matches (-1: by the average run, 0: a run per thread)") \
    X(part_vrank, "L3D_PART_VRANK", 0, "with part_vworld: the rank of the job whose block of views a world-1 partitioned run keeps") \
    X(part_vworld, "L3D_PART_VWORLD", 0, "a world-1 shard_run with commit 3 keeps the block of views rank part_vrank of a job of this many ranks would own: ONE rank's share of a job that does not fit one GPU, exercised on one GPU (scripts/run_rank_share.py)") \
    X(run_tables, "L3D_RUN_TABLES", 1, "resident chain: 1 = the kept writer fills a run table per view and packs (local camera, target) into the side array; later views and the products read runs instead of scanning lists, 0 = round 5's scans of the side array of global camera ids (A/B)") \
    X(slot_ring, "L3D_SLOT_RING", -1, "sharded run: 1 = always retire old gathered blocks into the compact arena (ring of window + 18 views), 0 = never, -1 = when all blocks exceed 8 GB") \
    X(defer_stats, "L3D_DEFER_STATS", 0, "sharded native run: 1 = no host wait for a view's stage-1 statistics (measured slower: 93 vs 84 us per view at 8 ranks, DESIGN 6)") \
    X(graph, "L3D_GRAPH", 0, "sharded native run with L3D_DEFER_STATS=1: 1 = passes 3.. replay a view's five launches as one captured graph (measured slower: 99 vs 93 us per view, DESIGN 6)")             \
    X(host_bookkeeping, "L3D_HOST_BOOKKEEPING", 0, "CROSS-CHECK BUILD ONLY (-DL3D_CROSSCHECKS, libline3d_amd_check.so): matchViews with the rounds-1-2 host bookkeeping") \
    X(host_clustering, "L3D_HOST_CLUSTERING", 0, "CROSS-CHECK BUILD ONLY: merge loop and grouping on the host threads")             \
    X(match_sync, "L3D_MATCH_SYNC", 0, "CROSS-CHECK BUILD ONLY: matchViews through the per-view seam call by default")

struct Options {
#define X(field, env, def, doc) int field = def;
    L3D_OPTION_TABLE(X)
#undef X
};

// The three switches that force a host-side stage where the device stage would run exist for A/B tests only: the shipped library
// (built without -DL3D_CROSSCHECKS) neither reads them from the environment nor lets l3d_set_option set them -- the host stages stay
// in it only as what they are in the product: the per-view seam path (the reference's control flow), the sharded run's host commit,
// and the fallbacks for inputs the device stages refuse.
inline bool option_is_crosscheck(const char* env)
{
    return strcmp(env, "L3D_HOST_BOOKKEEPING") == 0 || strcmp(env, "L3D_HOST_CLUSTERING") == 0 || strcmp(env, "L3D_MATCH_SYNC") == 0;
}
#ifdef L3D_CROSSCHECKS
constexpr bool kCrossChecks = true;
#else
constexpr bool kCrossChecks = false;
#endif

// name = the environment variable with or without its L3D_ prefix, any case of the prefix-less part as in the table
inline int* option_field(Options& o, const char* name)
{
    if (!name) return nullptr;
#define X(field, env, def, doc) if (strcmp(name, env) == 0 || strcmp(name, &env[4]) == 0 || strcmp(name, #field) == 0) return (kCrossChecks || !option_is_crosscheck(env)) ? &o.field : nullptr;
    L3D_OPTION_TABLE(X)
#undef X
    return nullptr;
}

Options options_from_env();             // l3d_capi.hip
}  // namespace l3d
struct l3d_ctx;
namespace l3d {
const Options& ctx_options(const l3d_ctx* c);   // l3d_capi.hip: the switches of a context, for code that sees the context as an opaque handle (line3d_host.cpp)

// the two switches read by code that is process-wide by nature (the ONE pool of host worker threads every context shares; the sleep of waiting
// host threads): set from the options of the last context created / the last l3d_set_option.  Everything that shapes a LAUNCH (vw_lds,
// vw_wide_max, pair_spb) is passed down from the launching context's own options -- a second context does not change the first one's launches.
struct ProcessTunables {
    std::atomic<int> host_threads{0}, wait_sleep_us{20};
};
inline ProcessTunables& tunables() { static ProcessTunables t; return t; }
inline void publish_tunables(const Options& o)
{
    ProcessTunables& t = tunables();
    t.host_threads = o.host_threads; t.wait_sleep_us = o.wait_sleep_us;
}

}  // namespace l3d
