// Micro-benchmark (gfx950), round 6: how the weight-gradient kernels' operand loads stream.  A workgroup of 512 threads (one per CU,
// like k_dw_skipres8) reads 512 rows (row stride T floats) of its own k-chunk, 16-byte loads, a fixed number of loads in flight:
//   seg64 : 4 lanes cover 64 contiguous bytes of a row per step of 16 positions (a wave instruction = 16 rows x 64 B: half cache lines;
//           the other half of each line is requested one step later)  -- the mapping of k_gemm6_dw / k_dw_skipres8
//   seg128: 8 lanes cover a whole 128-byte line of a row per step of 32 positions (a wave instruction = 8 rows x 128 B)
//   seg256: 16 lanes cover 256 contiguous bytes per step of 64 positions
// hipcc --offload-arch=gfx950 -O3 -o stream_rows stream_rows.hip && ./stream_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int LPR, int DEPTH>   // lanes per row; row groups in flight
__global__ __launch_bounds__(512) void k_rows(const float* __restrict__ p, float* out, int T, int kchunk, int nrows) {
    constexpr int RPI = 512 / LPR;            // rows per block-wide load instruction
    constexpr int KSTEP = LPR * 4;            // positions per step
    const int tid = threadIdx.x;
    const int row0 = tid / LPR, kk = (tid % LPR) * 4;
    const float* base = p + ((long)blockIdx.x * nrows) * T + kk;
    const int ngroups = nrows / RPI;          // row groups per step
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int nsteps = kchunk / KSTEP;
    const int total = nsteps * ngroups;       // loads of this thread, in order: step-major, then row group
    float4 v[DEPTH];
    auto addr = [&](int i) {
        const int s = i / ngroups, g = i - s * ngroups;
        return reinterpret_cast<const float4*>(base + (long)(row0 + g * RPI) * T + s * KSTEP);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) v[d] = *addr(d < total ? d : total - 1);
    for (int i = 0; i < total; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const float4 u = v[d];
            const int nx = i + DEPTH + d;
            v[d] = *addr(nx < total ? nx : total - 1);
            acc.x += u.x; acc.y += u.y; acc.z += u.z; acc.w += u.w;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int LPR, int DEPTH>
static int run(const float* p, float* out, int T, int kchunk, int nrows, int nblocks, const char* name) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_rows<LPR, DEPTH>), dim3(nblocks), dim3(512), 0, 0, p, out, T, kchunk, nrows);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 1 && ms < best) best = ms;
    }
    const double bytes = (double)nblocks * nrows * kchunk * 4;
    printf("%-8s depth %2d (%5.1f KB in flight per CU): %7.1f us  %5.2f TB/s   (%.2f GB)\n", name, DEPTH, DEPTH * 512 * 16 / 1024.0, best * 1e3,
           bytes / (best * 1e-3) / 1e12, bytes / 1e9);
    return 0;
}

int main() {
    const int T = 23040, nrows = 512, nblocks = 240, kchunk = 11520;
    float *p, *out;
    const size_t n = (size_t)nblocks * nrows * T;
    CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4)); CK(hipMalloc(&out, 64));
    CK(hipDeviceSynchronize());
    if (run<4, 4>(p, out, T, kchunk, nrows, nblocks, "seg64")) return 1;
    if (run<4, 8>(p, out, T, kchunk, nrows, nblocks, "seg64")) return 1;
    if (run<4, 16>(p, out, T, kchunk, nrows, nblocks, "seg64")) return 1;
    if (run<8, 4>(p, out, T, kchunk, nrows, nblocks, "seg128")) return 1;
    if (run<8, 8>(p, out, T, kchunk, nrows, nblocks, "seg128")) return 1;
    if (run<8, 16>(p, out, T, kchunk, nrows, nblocks, "seg128")) return 1;
    if (run<16, 4>(p, out, T, kchunk, nrows, nblocks, "seg256")) return 1;
    if (run<16, 8>(p, out, T, kchunk, nrows, nblocks, "seg256")) return 1;
    if (run<16, 16>(p, out, T, kchunk, nrows, nblocks, "seg256")) return 1;
    return 0;
}
