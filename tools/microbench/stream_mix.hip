// Micro-benchmark (gfx950): what does the memory system sustain for a launch that READS nr and WRITES nw activation tensors
// [B][64][T] fp32 (53.5 MB each at B = 8, T = 26112) exactly once, with no arithmetic at all?  The fused forward block is 1 read :
// 3 writes (x -> s, z, x_next), the backward chain 5 reads : 2 writes; their achieved rates (3.5 and 4.7 TB/s of their own bytes)
// are compared with these numbers in DESIGN.md 8.
//   style 0: linear float4 per lane (the friendliest possible stream)
//   style 1: the kernels' tile pattern: a wave owns 64 channels x 32 samples, one dword per lane and channel row
//            (two 128-byte row segments per instruction, rows T*4 bytes apart), contiguous tile spans; "tile/8" = 8 waves per CU
//            as in the fused kernels (256 VGPRs each), "tile/32" = 32 waves per CU (what the pattern allows with small waves)
// Every launch works on a fresh set of buffers (ring of 6 sets), as a layer's tensors are fresh in the training step.
//   hipcc --offload-arch=gfx950 -O3 -o stream_mix stream_mix.hip && ./stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
static __device__ rsrc_t make_buf(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000); }
struct Ptrs { const float* r[6]; float* w[4]; };

template <int NR, int NW>
__global__ __launch_bounds__(512) void k_linear(Ptrs p, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll
        for (int a = 0; a < NR; ++a) {
            const float4 u = reinterpret_cast<const float4*>(p.r[a])[i];
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
#pragma unroll
        for (int a = 0; a < NW; ++a) reinterpret_cast<float4*>(p.w[a])[i] = v;
        if (NW == 0 && v.x + v.y + v.z + v.w == 12345.678f) p.w[0][0] = v.y;   // keeps the loads alive
    }
}

template <int NR, int NW>
__global__ __launch_bounds__(512) void k_tile(Ptrs p, int B, int T) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, hi = lane >> 5;
    const int tiles_per_b = T / 32, ntiles = B * tiles_per_b;
    const int nwv = gridDim.x * 8, w = blockIdx.x * 8 + wave;
    const int per = (ntiles + nwv - 1) / nwv;                       // contiguous span of tiles per wave
    const int T4 = T * 4;
    for (int tile = w * per; tile < ntiles && tile < (w + 1) * per; ++tile) {
        const int bb = tile / tiles_per_b, tt = (tile - bb * tiles_per_b) * 32;
        const int voff = (4 * hi * T + tt + li) * 4;
        float x[32];
#pragma unroll
        for (int s = 0; s < 32; ++s) x[s] = (float)s;
#pragma unroll
        for (int a = 0; a < NR; ++a) {
            const rsrc_t Xr = make_buf(p.r[a] + (long)bb * 64 * T, 64u * T4);
#pragma unroll
            for (int s = 0; s < 32; ++s)
                x[s] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(Xr, voff, (8 * (s >> 2) + (s & 3)) * T4, 0));
        }
#pragma unroll
        for (int a = 0; a < NW; ++a) {
            const rsrc_t Yr = make_buf(p.w[a] + (long)bb * 64 * T, 64u * T4);
#pragma unroll
            for (int s = 0; s < 32; ++s)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x[s] + (float)a), Yr, voff, (8 * (s >> 2) + (s & 3)) * T4, 0);
        }
        if (NW == 0) {
            float t = 0.f;
#pragma unroll
            for (int s = 0; s < 32; ++s) t += x[s];
            if (t == 12345.678f) p.w[0][0] = t;   // keeps all 32 loads alive
        }
    }
}


// the tile pattern as the fused kernels run it: the loads of a wave's NEXT tile are in flight while the current tile is stored
//   WALK 0: a wave owns a contiguous span of tiles (the kernels' tile_walk: the history tap of a tile is the same wave's)
//   WALK 1: the 8 waves of a workgroup take 8 CONSECUTIVE tiles at a time (1 KB of every channel row written at about the same
//           time instead of 128 bytes), workgroups own contiguous spans
template <int NR, int NW, int WALK>
__global__ __launch_bounds__(512) void k_tile_pf(Ptrs p, int B, int T) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, hi = lane >> 5;
    const int tiles_per_b = T / 32, ntiles = B * tiles_per_b;
    const int T4 = T * 4;
    int first, end, step;
    if (WALK == 0) {
        const int nwv = gridDim.x * 8, w = blockIdx.x * 8 + wave;
        const int per = (ntiles + nwv - 1) / nwv;
        first = w * per; end = min(ntiles, (w + 1) * per); step = 1;
    } else {
        const int perwg = ((ntiles + gridDim.x - 1) / gridDim.x + 7) & ~7;
        first = blockIdx.x * perwg + wave; end = min(ntiles, (blockIdx.x + 1) * perwg); step = 8;
    }
    float x[2][NR > 0 ? NR : 1][32];
    auto issue = [&](int tile, int buf) {
        const int bb = tile / tiles_per_b, tt = (tile - bb * tiles_per_b) * 32;
        const int voff = (4 * hi * T + tt + li) * 4;
#pragma unroll
        for (int a = 0; a < NR; ++a) {
            const rsrc_t Xr = make_buf(p.r[a] + (long)bb * 64 * T, 64u * T4);
#pragma unroll
            for (int s = 0; s < 32; ++s)
                x[buf][a][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(Xr, voff, (8 * (s >> 2) + (s & 3)) * T4, 0));
        }
    };
    auto body = [&](int tile, int buf, int next) {
        if (next < end) issue(next, buf ^ 1);
        const int bb = tile / tiles_per_b, tt = (tile - bb * tiles_per_b) * 32;
        const int voff = (4 * hi * T + tt + li) * 4;
        float v[32];
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            v[s] = (float)s;
#pragma unroll
            for (int a = 0; a < NR; ++a) v[s] += x[buf][a][s];
        }
#pragma unroll
        for (int a = 0; a < NW; ++a) {
            const rsrc_t Yr = make_buf(p.w[a] + (long)bb * 64 * T, 64u * T4);
#pragma unroll
            for (int s = 0; s < 32; ++s)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[s] + (float)a), Yr, voff, (8 * (s >> 2) + (s & 3)) * T4, 0);
        }
    };
    if (first < end) issue(first, 0);
    int tile = first;
    while (tile < end) {   // two tiles per trip: the register buffers are indexed statically
        body(tile, 0, tile + step);
        tile += step;
        if (tile < end) { body(tile, 1, tile + step); tile += step; }
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NR, int NW>
static int run(std::vector<float*>& bufs, int B, int T, int blocks_per_cu) {
    const long n = (long)B * 64 * T;
    const int SETS = 6, PER = 7;   // buffers per set
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int style = 0; style < 5; ++style) {
        if (style >= 3 && (NR > 2 || NW == 0)) continue;
        float best = 1e30f, sum = 0.f;
        const int reps = 18;
        for (int it = 0; it < reps + 3; ++it) {
            Ptrs p;
            float** set = &bufs[(it % SETS) * PER];
            for (int a = 0; a < 6; ++a) p.r[a] = set[a < NR ? a : 0];
            for (int a = 0; a < 4; ++a) p.w[a] = set[a < NW ? PER - NW + a : PER - 1];   // NR + NW <= 7: disjoint
            CK(hipEventRecord(e0, 0));
            if (style == 0) hipLaunchKernelGGL((k_linear<NR, NW>), dim3(256 * blocks_per_cu), dim3(512), 0, 0, p, n / 4);
            else if (style < 3) hipLaunchKernelGGL((k_tile<NR, NW>), dim3(style == 1 ? 256 : 1024), dim3(512), 0, 0, p, B, T);
            else if (style == 3) hipLaunchKernelGGL((k_tile_pf<(NR <= 2 ? NR : 1), NW, 0>), dim3(256), dim3(512), 0, 0, p, B, T);
            else hipLaunchKernelGGL((k_tile_pf<(NR <= 2 ? NR : 1), NW, 1>), dim3(256), dim3(512), 0, 0, p, B, T);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 3) { best = ms < best ? ms : best; sum += ms; }
        }
        const double bytes = (double)(NR + NW) * n * 4;
        printf("reads %d writes %d  %-7s  mean %7.1f us  best %7.1f us   %5.2f TB/s mean  %5.2f TB/s best   (%.0f MB per launch)\n", NR, NW,
               style == 0 ? "linear" : style == 1 ? "tile/8" : style == 2 ? "tile/32" : style == 3 ? "pf/span" : "pf/wg8", sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12, bytes / 1e6);
    }
    return 0;
}

int main() {
    const int B = 8, T = 26112;
    const long n = (long)B * 64 * T;
    std::vector<float*> bufs(6 * 7);
    for (auto& b : bufs) { CK(hipMalloc(&b, n * 4)); CK(hipMemset(b, 0, n * 4)); }
    CK(hipDeviceSynchronize());
    if (run<1, 0>(bufs, B, T, 8)) return 1;
    if (run<0, 1>(bufs, B, T, 8)) return 1;
    if (run<1, 1>(bufs, B, T, 8)) return 1;
    if (run<1, 3>(bufs, B, T, 8)) return 1;   // forward block
    if (run<2, 2>(bufs, B, T, 8)) return 1;
    if (run<3, 1>(bufs, B, T, 8)) return 1;
    if (run<5, 2>(bufs, B, T, 8)) return 1;   // backward chain
    return 0;
}
