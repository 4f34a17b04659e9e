// l3d_sort.hip -- the one translation unit that instantiates hipCUB (see l3d_sort.hpp).
#include <hipcub/hipcub.hpp>

#include "l3d_sort.hpp"
#include "l3d_kernels.hpp"

namespace l3d {

hipError_t sort_pairs_u64_u32(void* tmp, size_t& bytes, const unsigned long long* keys_in, unsigned long long* keys_out, const unsigned* vals_in, unsigned* vals_out,
                              int n, int begin_bit, int end_bit, hipStream_t st)
{
    return hipcub::DeviceRadixSort::SortPairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, st);
}
hipError_t sort_pairs_u32_u32(void* tmp, size_t& bytes, const unsigned* keys_in, unsigned* keys_out, const unsigned* vals_in, unsigned* vals_out,
                              int n, int begin_bit, int end_bit, hipStream_t st)
{
    return hipcub::DeviceRadixSort::SortPairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, st);
}
hipError_t sort_keys_u64(void* tmp, size_t& bytes, const unsigned long long* keys_in, unsigned long long* keys_out, int n, int begin_bit, int end_bit, hipStream_t st)
{
    return hipcub::DeviceRadixSort::SortKeys(tmp, bytes, keys_in, keys_out, n, begin_bit, end_bit, st);
}
hipError_t exclusive_sum_int(void* tmp, size_t& bytes, const int* in, int* out, int n, hipStream_t st)
{
    return hipcub::DeviceScan::ExclusiveSum(tmp, bytes, in, out, n, st);
}

__global__ void k_sort_touch() {}
void warm_sort() { touch_kernel(reinterpret_cast<const void*>(&k_sort_touch)); }

}  // namespace l3d
