// oracle/ref_devfn_launch.cc -- TEST INFRASTRUCTURE.  Storage for the launch variables the genuine NVIDIA header
// <device_launch_parameters.h> DECLARES (`extern const uint3 threadIdx, blockIdx; extern const dim3 blockDim, gridDim`) and that nvcc
// provides as built-ins: the reference's texture-free __global__ kernels (cudawrapper.cu:717-829, compiled from the reference's own
// text by make_ref_devfn.py) read them like any other variable.  This translation unit does not include the header, so it may define
// them writable; the door (ref_devfn_door.cc) sets them through l3dref_set_launch before every call -- one "thread" at a time.
extern "C" {
struct l3dref_u3 { unsigned x, y, z; };      // the layout of uint3 / dim3: three 32-bit unsigned
l3dref_u3 threadIdx = { 0, 0, 0 }, blockIdx = { 0, 0, 0 }, blockDim = { 1, 1, 1 }, gridDim = { 1, 1, 1 };
int warpSize = 32;
void l3dref_set_launch(unsigned block_x, unsigned block_y, unsigned thread_x, unsigned thread_y, unsigned dim_x, unsigned dim_y)
{
    blockIdx.x = block_x; blockIdx.y = block_y; blockIdx.z = 0;
    threadIdx.x = thread_x; threadIdx.y = thread_y; threadIdx.z = 0;
    blockDim.x = dim_x; blockDim.y = dim_y; blockDim.z = 1;
}
}
