// Micro-benchmark (gfx950): how do VALU instructions share a SIMD with a stream of v_mfma_f32_32x32x16_bf16?
//   mode 0: every wave issues NM independent MFMAs per iteration
//   mode 1: every wave issues NM MFMAs with NV independent v_fma_f32 after each MFMA
//   mode 2: waves 0-3 (one per SIMD) MFMAs only, waves 4-7 (their SIMD partners) VALU only (NV * NM v_fma per iteration)
//   mode 3: VALU only, all waves
// prints cycles per iteration of wave 0 (MFMA wave) and wave 4 (its partner).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip && ./mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE, int NV>
__global__ __launch_bounds__(512) void k(long long* out, float* sink, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    const bool do_mfma = MODE == 0 || MODE == 1 || (MODE == 2 && wave < 4);
    const bool do_valu = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= 4);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (do_mfma) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
            if (do_valu) {
#pragma unroll
                for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(v[(q + 1) & 7]), "v"(v[(q + 2) & 7]));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int NV>
static void run(const char* name, int blocks, int threads) {
    long long* out; float* sink;
    hipMalloc(&out, 4096 * 8); hipMalloc(&sink, 4);
    const int iters = 2000;
    hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(threads), 0, 0, out, sink, iters);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(threads), 0, 0, out, sink, iters);
    hipDeviceSynchronize();
    long long h[8];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-58s wave0 %7.1f  wave4 %7.1f cycles per 16-MFMA iteration (%.1f per MFMA slot)\n", name, (double)h[0] / iters,
           threads > 256 ? (double)h[4] / iters : 0.0, (double)h[0] / iters / 16);
    hipFree(out); hipFree(sink);
}

int main() {
    run<0, 0>("MFMA only, 1 wave/SIMD", 256, 256);
    run<0, 0>("MFMA only, 2 waves/SIMD", 256, 512);
    run<1, 1>("MFMA + 1 VALU each, 1 wave/SIMD", 256, 256);
    run<1, 2>("MFMA + 2 VALU each, 1 wave/SIMD", 256, 256);
    run<1, 4>("MFMA + 4 VALU each, 1 wave/SIMD", 256, 256);
    run<1, 6>("MFMA + 6 VALU each, 1 wave/SIMD", 256, 256);
    run<1, 8>("MFMA + 8 VALU each, 1 wave/SIMD", 256, 256);
    run<1, 4>("MFMA + 4 VALU each, 2 waves/SIMD", 256, 512);
    run<3, 4>("VALU only (4 per slot), 1 wave/SIMD", 256, 256);
    run<3, 4>("VALU only (4 per slot), 2 waves/SIMD", 256, 512);
    run<2, 1>("wave A MFMA, wave B VALU (1 per slot)", 256, 512);
    run<2, 2>("wave A MFMA, wave B VALU (2 per slot)", 256, 512);
    run<2, 4>("wave A MFMA, wave B VALU (4 per slot)", 256, 512);
    run<2, 8>("wave A MFMA, wave B VALU (8 per slot)", 256, 512);
    return 0;
}
