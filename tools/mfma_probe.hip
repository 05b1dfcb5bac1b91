// tools/mfma_probe.hip -- micro-benchmark of the f32-input MFMA issue rate on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_probe tools/mfma_probe.hip && ./gpurun_out/mfma_probe
// Variants: operands from registers vs LDS (ds_read + wait per k-step, like wn_gemm.hip),
// 1..3 workgroups (of 4 waves) per CU, 32x32x2 vs 16x16x4, 2x2 vs 4x1 accumulator tiles.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: registers, 32x32x2 2x2 ; 1: LDS-fed 2x2 ; 2: LDS-fed with next-step prefetch ; 3: 16x16x4 regs (16 acc)
__global__ __launch_bounds__(256) void probe(float* out, int iters, int lds_pad) {
    __shared__ float As[32 * 128 + 64], Bs[32 * 128 + 64];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hi = lane >> 5;
    for (int i = tid; i < 32 * 128; i += 256) { As[i] = 0.001f * (i & 15); Bs[i] = 0.002f * (i & 7); }
    __syncthreads();
    if (MODE == 3) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        float a = 0.001f * lane, b = 0.002f * lane;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
            }
        }
        float r = 0;
        for (int q = 0; q < 16; ++q) r += acc[q][0] + acc[q][3];
        out[blockIdx.x * 256 + tid] = r;
        return;
    }
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    float a0 = 0.001f * lane, a1 = 0.003f * lane, b0 = 0.002f * lane, b1 = 0.004f * lane;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        } else if (MODE == 1) {
#pragma unroll 4
            for (int s = 0; s < 16; ++s) {
                const int kk = 2 * s + hi;
                const float x0 = As[kk * 128 + (wm * 2 + 0) * 32 + li], x1 = As[kk * 128 + (wm * 2 + 1) * 32 + li];
                const float y0 = Bs[kk * 128 + (wn * 2 + 0) * 32 + li], y1 = Bs[kk * 128 + (wn * 2 + 1) * 32 + li];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, acc[1][1], 0, 0, 0);
            }
            if (lds_pad) __syncthreads();
        } else {
            // software pipelined: operands of step s+1 are read before the MFMAs of step s
            float x0 = As[hi * 128 + (wm * 2 + 0) * 32 + li], x1 = As[hi * 128 + (wm * 2 + 1) * 32 + li];
            float y0 = Bs[hi * 128 + (wn * 2 + 0) * 32 + li], y1 = Bs[hi * 128 + (wn * 2 + 1) * 32 + li];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int kn = (2 * (s + 1) + hi) & 31;
                const float nx0 = As[kn * 128 + (wm * 2 + 0) * 32 + li], nx1 = As[kn * 128 + (wm * 2 + 1) * 32 + li];
                const float ny0 = Bs[kn * 128 + (wn * 2 + 0) * 32 + li], ny1 = Bs[kn * 128 + (wn * 2 + 1) * 32 + li];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, acc[1][1], 0, 0, 0);
                x0 = nx0; x1 = nx1; y0 = ny0; y1 = ny1;
            }
            if (lds_pad) __syncthreads();
        }
    }
    float r = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) r += acc[i][j][0] + acc[i][j][15];
    out[blockIdx.x * 256 + tid] = r;
}

template <int MODE>
static void run(const char* name, int blocks_per_cu, int sync) {
    float* out;
    const int nblk = 256 * blocks_per_cu, iters = 2000;
    hipMalloc(&out, (size_t)nblk * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(nblk), dim3(256), 0, 0, out, 50, sync);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(nblk), dim3(256), 0, 0, out, iters, sync);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops_per_iter_wave = (MODE == 3) ? 64.0 * 2 * 16 * 16 * 4 : 64.0 * 2 * 32 * 32 * 2;
    const double tf = flops_per_iter_wave * iters * 4.0 * nblk / (ms * 1e-3) / 1e12;
    printf("%-34s blocks/CU=%d sync=%d : %8.3f ms  %7.1f TFLOP/s (%.0f%% of 157.3)\n", name, blocks_per_cu, sync, ms, tf, 100 * tf / 157.3);
    hipFree(out);
}

int main() {
    for (int b = 1; b <= 3; ++b) run<0>("regs 32x32x2 2x2", b, 0);
    for (int b = 1; b <= 3; ++b) run<3>("regs 16x16x4 x16", b, 0);
    for (int b = 1; b <= 3; ++b) run<1>("lds-fed 32x32x2 2x2", b, 0);
    for (int b = 1; b <= 3; ++b) run<1>("lds-fed 32x32x2 2x2 +barrier/16", b, 1);
    for (int b = 1; b <= 3; ++b) run<2>("lds-fed prefetch-next", b, 0);
    for (int b = 1; b <= 3; ++b) run<2>("lds-fed prefetch-next +barrier/16", b, 1);
    return 0;
}
