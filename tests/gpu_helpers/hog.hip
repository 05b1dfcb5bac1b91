// TEST INFRASTRUCTURE (tests/test_gpu_decode_pins.py): workgroups that do nothing but HOLD a compute unit each (their dynamic LDS
// leaves no room for a second large workgroup) until a stop word is set or a time limit passes -- a stand-in for "another kernel has
// the CUs" when the persistent decode launch starts.  Compiled by the test with hipcc; never part of the product library.
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(64) void k_hog(const int* stop, long long ticks, long long* started) {
    extern __shared__ char lds[];
    lds[threadIdx.x] = 0;
    const long long t0 = (long long)wall_clock64();   // 100 MHz
    if (threadIdx.x == 0) atomicAdd((unsigned long long*)started, 1ull);
    while ((long long)wall_clock64() - t0 < ticks && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        __builtin_amdgcn_s_sleep(127);
}

extern "C" int hog_launch(int blocks, int lds_bytes, const int* stop, long long ticks, long long* started, void* stream) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_hog), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return -1;
        attr = true;
    }
    hipLaunchKernelGGL(k_hog, dim3(blocks), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, stop, ticks, started);
    return (int)hipGetLastError();
}
