// l3d_sort.hpp -- the library's device-wide sorts and scans (hipCUB / rocPRIM) behind four plain functions, instantiated ONCE, in
// l3d_sort.hip.  Every other translation unit used to instantiate hipCUB's radix sort itself: four multi-megabyte code objects
// (12 of the library's 14 MB) whose loading was most of the first prepare() of a process.  Same calls, same results: a stable LSD
// radix sort has one result.  tmp == nullptr: size query (bytes is set), as with hipCUB.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

namespace l3d {

hipError_t sort_pairs_u64_u32(void* tmp, size_t& bytes, const unsigned long long* keys_in, unsigned long long* keys_out, const unsigned* vals_in, unsigned* vals_out,
                              int n, int begin_bit, int end_bit, hipStream_t st);
hipError_t sort_pairs_u32_u32(void* tmp, size_t& bytes, const unsigned* keys_in, unsigned* keys_out, const unsigned* vals_in, unsigned* vals_out,
                              int n, int begin_bit, int end_bit, hipStream_t st);
hipError_t sort_keys_u64(void* tmp, size_t& bytes, const unsigned long long* keys_in, unsigned long long* keys_out, int n, int begin_bit, int end_bit, hipStream_t st);
hipError_t exclusive_sum_int(void* tmp, size_t& bytes, const int* in, int* out, int n, hipStream_t st);

}  // namespace l3d
