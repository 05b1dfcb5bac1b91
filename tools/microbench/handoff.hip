// Micro-benchmark (gfx950): what does handing data from one workgroup to another cost INSIDE a launch?
//
// Two questions of the round-3 verdict are priced here instead of estimated:
//   (item 6) decode B = 1 over two CUs of one XCD: every layer would exchange 32-float vectors twice.  Mode A = the latency
//            form of the guide: G granules {float, tag} of 8 bytes, ONE sc1 store per lane, the consumer polls the granules
//            themselves with sc1 loads.  Reported: one-way hop time for G = 1, 32, 64 on an otherwise idle chip, same XCD and
//            cross XCD.
//   (item 5) a persistent multi-layer forward: layer l+1's tile waits for layer l's tiles of a neighbouring workgroup.  Mode B =
//            the bandwidth form: a tile of P bytes with plain stores from all 512 threads, drained (vmcnt(0)), barrier, flag;
//            the consumer polls the flag (relaxed), acquires (agent scope: drops its L1), barrier, reads the tile with plain
//            loads and checks EVERY word.  Variants: producer release fence (placement independent) or none (only correct if both
//            workgroups share an L2, i.e. sit on one XCD -- stale words are counted, not assumed away).
//   Beside them: the cost of a dependent kernel boundary for a 240 x 512-thread grid (empty kernel, and a kernel that only fills
//   120 KB of LDS from an L2-resident weight image the way the fused kernels' prologue does).
//
//   hipcc --offload-arch=gfx950 -O3 -o handoff handoff.hip && ./handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define SPIN_MAX (1 << 24)

static __device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; }  // HW_REG_XCC_ID
typedef unsigned long long u64;

struct Report {
    u64 ticks;        // wall_clock64 ticks (100 MHz) of the timed loop in block a
    unsigned xcc_a, xcc_b;
    unsigned stale;   // payload words that did not carry the expected value
    unsigned timeout; // a spin ran into SPIN_MAX
};

// ---- mode A: granule ping-pong, one wave per side -------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_granule(u64* ab, u64* ba, int G, int iters, int blk_a, int blk_b, Report* rep) {
    const int me = (int)blockIdx.x == blk_a ? 0 : ((int)blockIdx.x == blk_b ? 1 : -1);
    if (me < 0 || threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    if (lane == 0) (me == 0 ? rep->xcc_a : rep->xcc_b) = xcc_id();
    u64* out = me == 0 ? ab : ba;
    u64* in = me == 0 ? ba : ab;
    unsigned bad = 0, to = 0;
    u64 t0 = 0;
    for (int it = 1; it <= iters + 8; ++it) {
        if (it == 9 && me == 0) t0 = wall_clock64();
        if (me == 0 && lane < G)
            __hip_atomic_store(out + lane, ((u64)(unsigned)it << 32) | (unsigned)(it * 64 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < G) {
            int spin = 0;
            u64 v;
            do {
                v = __hip_atomic_load(in + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } while ((unsigned)(v >> 32) != (unsigned)it && ++spin < SPIN_MAX);
            if (spin >= SPIN_MAX) to = 1;
            if ((unsigned)v != (unsigned)(it * 64 + lane)) bad++;
        }
        if (to) break;
        if (me == 1 && lane < G)
            __hip_atomic_store(out + lane, ((u64)(unsigned)it << 32) | (unsigned)(it * 64 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (me == 0 && lane == 0) rep->ticks = wall_clock64() - t0;
    if (bad) atomicAdd(&rep->stale, bad);
    if (to) rep->timeout = 1;
}

// ---- mode B: tile + flag ----------------------------------------------------------------------------------------------------
// release: 0 = none (vmcnt(0) only), 1 = agent-scope release fence by lane 0 after the barrier, 2 = write-through (sc1) data
// stores and sc1 loads, no fence on either side
__global__ __launch_bounds__(512) void k_tile(float* ab, float* ba, unsigned* flag_ab, unsigned* flag_ba, int nfloat, int iters,
                                               int release, int blk_a, int blk_b, Report* rep) {
    const int me = (int)blockIdx.x == blk_a ? 0 : ((int)blockIdx.x == blk_b ? 1 : -1);
    if (me < 0) return;
    const int tid = threadIdx.x;
    if (tid == 0) (me == 0 ? rep->xcc_a : rep->xcc_b) = xcc_id();
    float* out = me == 0 ? ab : ba;
    const float* in = me == 0 ? ba : ab;
    unsigned* fout = me == 0 ? flag_ab : flag_ba;
    unsigned* fin = me == 0 ? flag_ba : flag_ab;
    __shared__ int s_to;
    if (tid == 0) s_to = 0;
    __syncthreads();
    unsigned bad = 0;
    u64 t0 = 0;
    auto publish = [&](int it) {
        if (release == 2) {   // write-through stores (agent scope: sc1), no fence: is "store acknowledged" == "visible to the other XCDs"?
            for (int i = tid; i < nfloat; i += 512) __hip_atomic_store(out + i, (float)(it * 7 + (i & 1023)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            for (int i = tid; i < nfloat; i += 512) out[i] = (float)(it * 7 + (i & 1023));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (release == 1) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_store(fout, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto consume = [&](int it) {
        if (tid == 0) {
            int spin = 0;
            while (__hip_atomic_load(fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)it && ++spin < SPIN_MAX) __builtin_amdgcn_s_sleep(1);
            if (spin >= SPIN_MAX) s_to = 1;
            if (release != 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (release == 2) {   // sc1 loads (they do not take a stale line of this XCD's L2), no invalidate
            for (int i = tid; i < nfloat; i += 512)
                if (__hip_atomic_load(in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (float)(it * 7 + (i & 1023))) bad++;
        } else {
            for (int i = tid; i < nfloat; i += 512)
                if (in[i] != (float)(it * 7 + (i & 1023))) bad++;
        }
    };
    for (int it = 1; it <= iters + 8; ++it) {
        if (it == 9 && me == 0 && tid == 0) t0 = wall_clock64();
        if (me == 0) publish(it);
        consume(it);
        if (s_to) break;
        if (me == 1) publish(it);
    }
    if (me == 0 && tid == 0) rep->ticks = wall_clock64() - t0;
    if (bad) atomicAdd(&rep->stale, bad);
    if (tid == 0 && s_to) rep->timeout = 1;
}

// ---- launch boundary pieces -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_empty(float* p) {
    if (p == nullptr && threadIdx.x == 9999) p[0] = 1.f;
}
// every workgroup copies `bytes` of an image (L2 resident after the first launches) into LDS, 16 bytes per lane, like the fused
// kernels' weight prologue, and waits for it
__global__ __launch_bounds__(512) void k_ldsfill(const float* img, int bytes, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), (short)0, bytes, 0x00020000);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int off = wave * 1024; off < bytes; off += 8 * 1024)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + off), 16, lane * 16, off, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (sink && reinterpret_cast<float*>(smem)[threadIdx.x] == 12345.678f) sink[0] = 1.f;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main() {
    Report* rep;
    CK(hipMalloc(&rep, sizeof(Report)));
    u64 *gab, *gba;
    CK(hipMalloc(&gab, 64 * 8)); CK(hipMalloc(&gba, 64 * 8));
    float *tab, *tba;
    const int maxf = 16384;
    CK(hipMalloc(&tab, maxf * 4)); CK(hipMalloc(&tba, maxf * 4));
    unsigned* flags;
    CK(hipMalloc(&flags, 256));
    const int iters = 2000;
    auto show = [&](const char* what, int hops_per_iter) {
        Report h;
        CK(hipMemcpy(&h, rep, sizeof(h), hipMemcpyDeviceToHost));
        printf("%-58s xcc %u -> %u : %7.3f us per one-way hop   stale words %u%s\n", what, h.xcc_a, h.xcc_b,
               (double)h.ticks * 0.01 / iters / hops_per_iter, h.stale, h.timeout ? "   TIMEOUT" : "");
    };
    const int pairs[2][2] = {{0, 8}, {0, 1}};   // block b runs on XCD b % 8 (observed, reported below): same XCD, neighbouring XCD
    for (int pi = 0; pi < 2; ++pi) {
        for (int G : {1, 32, 64}) {
            CK(hipMemset(rep, 0, sizeof(Report))); CK(hipMemset(gab, 0, 512)); CK(hipMemset(gba, 0, 512));
            hipLaunchKernelGGL(k_granule, dim3(256), dim3(512), 0, 0, gab, gba, G, iters, pairs[pi][0], pairs[pi][1], rep);
            CK(hipDeviceSynchronize());
            char w[128];
            snprintf(w, sizeof(w), "A granules: %2d x 8 B sc1 (blocks %d, %d)", G, pairs[pi][0], pairs[pi][1]);
            show(w, 2);
        }
        for (int nf : {64, 2048, 6144, 16384}) {
            for (int rel = 0; rel < 3; ++rel) {
                CK(hipMemset(rep, 0, sizeof(Report))); CK(hipMemset(flags, 0, 256));
                hipLaunchKernelGGL(k_tile, dim3(256), dim3(512), 0, 0, tab, tba, flags, flags + 32, nf, iters, rel, pairs[pi][0], pairs[pi][1], rep);
                CK(hipDeviceSynchronize());
                char w[128];
                snprintf(w, sizeof(w), "B tile %6d B %s (blocks %d, %d)", nf * 4, rel == 2 ? "sc1 stores + flag, sc1 loads, no fence" : (rel ? "plain stores + flag, release fence" : "plain stores + flag, no release   "), pairs[pi][0], pairs[pi][1]);
                show(w, 2);
            }
        }
    }
    // dependent launch boundary of a 240 x 512 grid: empty kernel, LDS-image fill of 120 KB
    float* img;
    CK(hipMalloc(&img, 30 * 122880));
    CK(hipMemset(img, 0, 30 * 122880));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ldsfill), hipFuncAttributeMaxDynamicSharedMemorySize, 122880));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int kind = 0; kind < 2; ++kind) {
        const int n = 3000;
        for (int rep_ = 0; rep_ < 2; ++rep_) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < n; ++i) {
                if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(240), dim3(512), 0, 0, img);
                else hipLaunchKernelGGL(k_ldsfill, dim3(240), dim3(512), 122880, 0, img + (size_t)(i % 30) * 30720, 122880, (float*)nullptr);
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep_ == 1) printf("%-58s %7.3f us per launch (back to back on one stream)\n", kind == 0 ? "empty kernel, 240 x 512 threads" : "120 KB LDS image fill (buffer_load ... lds), 240 x 512", ms * 1000.0 / n);
        }
    }
    return 0;
}
