// Micro-benchmark (gfx950), round 6: the chip's streaming plateau for the fused kernels' byte mixes, measured the way
// MI355X_MICROARCH.md measures its 6.29 TB/s (float4 copy) -- buffers far beyond the 256 MiB Infinity Cache -- so that the
// "box plateau" of profiles/r03/stream_mix.txt (53 - 374 MB per launch, partly cache-resident, 13 - 70 us launches) can be
// reconciled with it.  Every tensor is 256 MiB (64 Mi floats = [B = 32][64][T = 32768]); a launch reads NR and writes NW of them
// (0.5 - 1.8 GiB per launch), two disjoint sets alternate.
//   grids: 256 x 8 (many short workgroups), 256 and 240 persistent workgroups of 512 threads (the fused kernels' shape)
//   styles: linear  = float4 per lane, grid-stride (the guide's copy when NR = NW = 1)
//           linear-nt = the same with non-temporal loads and stores
//           tile    = the fused kernels' access pattern: a wave owns 64 channels x 32 samples, one dword per lane and channel row
//                     (rows T*4 bytes apart), next tile's loads in flight while the current one is stored (k_tile_pf of stream_mix.hip)
//   hipcc --offload-arch=gfx950 -O3 -o stream_big stream_big.hip && ./stream_big
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
static __device__ rsrc_t make_buf(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000); }
struct Ptrs { const float* r[6]; float* w[4]; };

template <int NR, int NW, bool NT>
__global__ __launch_bounds__(512) void k_linear(Ptrs p, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll
        for (int a = 0; a < NR; ++a) {
            const float4* q = reinterpret_cast<const float4*>(p.r[a]) + i;
            float4 u;
            if (NT) { u.x = __builtin_nontemporal_load(&q->x); u.y = __builtin_nontemporal_load(&q->y); u.z = __builtin_nontemporal_load(&q->z); u.w = __builtin_nontemporal_load(&q->w); }
            else u = *q;
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
#pragma unroll
        for (int a = 0; a < NW; ++a) {
            float4* q = reinterpret_cast<float4*>(p.w[a]) + i;
            if (NT) { __builtin_nontemporal_store(v.x, &q->x); __builtin_nontemporal_store(v.y, &q->y); __builtin_nontemporal_store(v.z, &q->z); __builtin_nontemporal_store(v.w, &q->w); }
            else *q = v;
        }
        if (NW == 0 && v.x + v.y + v.z + v.w == 12345.678f) p.w[0][0] = v.y;
    }
}

// contiguous chunk per workgroup (a persistent workgroup streams ITS span, like the fused kernels' tile spans), float4 per lane
template <int NR, int NW>
__global__ __launch_bounds__(512) void k_span(Ptrs p, long n4) {
    const long per = (n4 + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
    for (long i = lo + threadIdx.x; i < hi; i += 512) {
        float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll
        for (int a = 0; a < NR; ++a) { const float4 u = reinterpret_cast<const float4*>(p.r[a])[i]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
#pragma unroll
        for (int a = 0; a < NW; ++a) reinterpret_cast<float4*>(p.w[a])[i] = v;
        if (NW == 0 && v.x + v.y + v.z + v.w == 12345.678f) p.w[0][0] = v.y;
    }
}

template <int NR, int NW>
__global__ __launch_bounds__(512) void k_tile_pf(Ptrs p, int B, int T) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, hi = lane >> 5;
    const int tiles_per_b = T / 32, ntiles = B * tiles_per_b;
    const int T4 = T * 4;
    const int nwv = gridDim.x * 8, w = blockIdx.x * 8 + wave;
    const int per = (ntiles + nwv - 1) / nwv;
    const int first = w * per, end = min(ntiles, (w + 1) * per);
    constexpr int NRR = NR > 2 ? 2 : (NR > 0 ? NR : 1);   // register budget: at most two tensors prefetched a tile ahead, the rest loaded in the tile
    float x[2][NRR][32];
    auto issue = [&](int tile, int buf) {
        const int bb = tile / tiles_per_b, tt = (tile - bb * tiles_per_b) * 32;
        const int voff = (4 * hi * T + tt + li) * 4;
#pragma unroll
        for (int a = 0; a < NRR && a < NR; ++a) {
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
            for (int a = 0; a < NRR && a < NR; ++a) v[s] += x[buf][a][s];
        }
#pragma unroll
        for (int a = NRR; a < NR; ++a) {
            const rsrc_t Xr = make_buf(p.r[a] + (long)bb * 64 * T, 64u * T4);
#pragma unroll
            for (int s = 0; s < 32; ++s)
                v[s] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(Xr, voff, (8 * (s >> 2) + (s & 3)) * T4, 0));
        }
#pragma unroll
        for (int a = 0; a < NW; ++a) {
            const rsrc_t Yr = make_buf(p.w[a] + (long)bb * 64 * T, 64u * T4);
#pragma unroll
            for (int s = 0; s < 32; ++s)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[s] + (float)a), Yr, voff, (8 * (s >> 2) + (s & 3)) * T4, 0);
        }
        if (NW == 0) {
            float t = 0.f;
#pragma unroll
            for (int s = 0; s < 32; ++s) t += v[s];
            if (t == 12345.678f) p.w[0][0] = t;
        }
    };
    if (first < end) issue(first, 0);
    int tile = first;
    while (tile < end) {
        body(tile, 0, tile + 1);
        ++tile;
        if (tile < end) { body(tile, 1, tile + 1); ++tile; }
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NR, int NW>
static int run(std::vector<float*>& bufs, int B, int T) {
    const long n = (long)B * 64 * T;
    const int SETS = 2, PER = 7;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { const char* name; int style; int grid; };
    const V vs[] = {{"linear 2048wg", 0, 2048}, {"linear 256wg", 0, 256}, {"linear-nt 2048wg", 1, 2048}, {"span 256wg", 2, 256}, {"span 240wg", 2, 240},
                    {"tile 256wg", 3, 256}, {"tile 240wg", 3, 240}};
    for (const V& v : vs) {
        float best = 1e30f, sum = 0.f;
        const int reps = 8;
        for (int it = 0; it < reps + 2; ++it) {
            Ptrs p;
            float** set = &bufs[(it % SETS) * PER];
            for (int a = 0; a < 6; ++a) p.r[a] = set[a < NR ? a : 0];
            for (int a = 0; a < 4; ++a) p.w[a] = set[a < NW ? PER - NW + a : PER - 1];
            CK(hipEventRecord(e0, 0));
            if (v.style == 0) hipLaunchKernelGGL((k_linear<NR, NW, false>), dim3(v.grid), dim3(512), 0, 0, p, n / 4);
            else if (v.style == 1) hipLaunchKernelGGL((k_linear<NR, NW, true>), dim3(v.grid), dim3(512), 0, 0, p, n / 4);
            else if (v.style == 2) hipLaunchKernelGGL((k_span<NR, NW>), dim3(v.grid), dim3(512), 0, 0, p, n / 4);
            else hipLaunchKernelGGL((k_tile_pf<NR, NW>), dim3(v.grid), dim3(512), 0, 0, p, B, T);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 2) { best = ms < best ? ms : best; sum += ms; }
        }
        const double bytes = (double)(NR + NW) * n * 4;
        printf("reads %d writes %d  %-17s mean %8.1f us  best %8.1f us   %5.2f TB/s mean  %5.2f TB/s best   (%.0f MiB per launch)\n", NR, NW, v.name,
               sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12, bytes / 1048576.0);
    }
    return 0;
}

int main() {
    const int B = 32, T = 32768;                       // 64 Mi floats = 256 MiB per tensor
    const long n = (long)B * 64 * T;
    std::vector<float*> bufs(2 * 7);
    for (auto& b : bufs) { CK(hipMalloc(&b, n * 4)); CK(hipMemset(b, 0, n * 4)); }
    CK(hipDeviceSynchronize());
    if (run<1, 1>(bufs, B, T)) return 1;   // the guide's copy
    if (run<1, 0>(bufs, B, T)) return 1;
    if (run<0, 1>(bufs, B, T)) return 1;
    if (run<1, 3>(bufs, B, T)) return 1;   // forward block: x -> x', s, z
    if (run<5, 2>(bufs, B, T)) return 1;   // backward chain as built: dP (2), dZs, s, z -> dP' (2)... 5 reads : 2+ writes
    if (run<2, 2>(bufs, B, T)) return 1;
    return 0;
}
