// TEST INFRASTRUCTURE (tests/test_gpu_rccl_coresidency.py): a stand-in for an RCCL all-reduce kernel on ONE GPU -- a grid of
// `blocks` workgroups of 256 threads (RCCL runs one workgroup per channel; distributed.rccl_footprint_defaults() caps it at 16)
// that streams over a buffer of the gradient's size and stamps, per workgroup, when it started and ended (100 MHz wall clock).
// Compiled by the test with hipcc; never part of the product library.
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void k_occupy(float* buf, long n, int iters, long long* stamps) {
    if (threadIdx.x == 0) stamps[2 * blockIdx.x] = (long long)wall_clock64();
    const long per = (n + gridDim.x - 1) / gridDim.x, lo = (long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (int it = 0; it < iters; ++it)
        for (long i = lo + threadIdx.x; i < hi; i += 256) buf[i] = buf[i] * 1.0f + 0.0f;
    __syncthreads();
    if (threadIdx.x == 0) stamps[2 * blockIdx.x + 1] = (long long)wall_clock64();
}
__global__ void k_stamp(long long* out) { out[0] = (long long)wall_clock64(); }

extern "C" int occupy_launch(float* buf, long n, int iters, long long* stamps, int blocks, void* stream) {
    hipLaunchKernelGGL(k_occupy, dim3(blocks), dim3(256), 0, (hipStream_t)stream, buf, n, iters, stamps);
    return (int)hipGetLastError();
}
extern "C" int stamp_launch(long long* out, void* stream) {
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, (hipStream_t)stream, out);
    return (int)hipGetLastError();
}
