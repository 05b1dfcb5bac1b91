// hip_emu.h -- TEST INFRASTRUCTURE ONLY.
//
// A minimal host-side emulation of the HIP execution model (grid of blocks, 64-lane
// wavefronts, __syncthreads, LDS, cross-lane shuffles and the gfx950 f32 MFMA lane maps)
// so that the kernel sources under pytorchwavenetvocoder_amd/csrc/ can be compiled with g++
// (-DWN_EMU) and their index arithmetic checked on a machine without a GPU.
//
// It is only ever built by tests/emu/build_emu.py into tests/emu/_build/libwavenet_emu.so and
// only ever loaded by tests (tests/test_emu_*.py).  The product package never loads it: the
// product loader (pytorchwavenetvocoder_amd/_lib.py) loads the gfx950 library only and raises
// if it is missing.  Nothing here is a CPU fallback for users.
//
// Model: every thread of a block is a ucontext fiber on ONE OS thread; blocks run one after
// another.  __syncthreads() and the wave-level rendezvous (shuffle / MFMA) yield to a
// round-robin scheduler until all participants arrived.  Divergent barriers deadlock and are
// reported.  MFMA lane maps follow /opt/skills/guides/cdna_hip_programming.md section 3:
//   v_mfma_f32_32x32x2_f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
//                            D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5), r in [0,16)
//   numerics: D = fma(a_k1,b_k1, fma(a_k0,b_k0, C))  (k-ordered f32 fma chain).
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {

struct State {
    dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;
    char* dyn_smem = nullptr;
    // scheduler
    ucontext_t sched_ctx;
    struct Fiber {
        ucontext_t ctx;
        char* stack = nullptr;
        bool done = false;
    };
    std::vector<Fiber> fibers;
    int cur = -1;
    int nthreads = 0;
    int alive = 0;
    // block barrier
    int bar_arrived = 0;
    long bar_gen = 0;
    // wave rendezvous
    std::vector<int> wv_arrived;
    std::vector<long> wv_gen;
    std::vector<int> wv_alive;
    long progress = 0;
    // exchange slots: [thread] 16 bytes x 2
    std::vector<uint64_t> slot_a, slot_b;
    std::function<void()> body;
};

// The state of the block that is executing.  Ordinary launches run their blocks one after another on ONE State;
// launch_coop() (below) keeps one State per block alive and points this at the block whose fibers are being scheduled.
inline State*& cur_state_() {
    static State base;
    static State* p = &base;
    return p;
}
inline State& S() { return *cur_state_(); }

inline void yield_() {
    State& s = S();
    swapcontext(&s.fibers[s.cur].ctx, &s.sched_ctx);
}

inline int tid_() {
    State& s = S();
    return (int)(s.threadIdx_.x + s.blockDim_.x * (s.threadIdx_.y + s.blockDim_.y * s.threadIdx_.z));
}

inline void sync_block() {
    State& s = S();
    long gen = s.bar_gen;
    if (++s.bar_arrived >= s.alive) {
        s.bar_arrived = 0;
        s.bar_gen++;
        s.progress++;
    } else {
        while (s.bar_gen == gen) yield_();
    }
}

inline void sync_wave() {
    State& s = S();
    int w = tid_() >> 6;
    long gen = s.wv_gen[w];
    if (++s.wv_arrived[w] >= s.wv_alive[w]) {
        s.wv_arrived[w] = 0;
        s.wv_gen[w]++;
        s.progress++;
    } else {
        while (s.wv_gen[w] == gen) yield_();
    }
}

inline void fiber_entry() {
    State& s = S();
    s.body();
    // thread finished
    int t = s.cur;
    s.fibers[t].done = true;
    s.alive--;
    s.wv_alive[t >> 6]--;
    s.progress++;
    if (s.alive > 0 && s.bar_arrived >= s.alive && s.bar_arrived > 0) {
        s.bar_arrived = 0;
        s.bar_gen++;
    }
    int w = t >> 6;
    if (s.wv_alive[w] > 0 && s.wv_arrived[w] >= s.wv_alive[w]) {
        s.wv_arrived[w] = 0;
        s.wv_gen[w]++;
    }
    swapcontext(&s.fibers[t].ctx, &s.sched_ctx);
}

static const size_t kStack = 256 * 1024;
static const size_t kStackCoop = 64 * 1024;   // cooperative launches keep EVERY block's fibers alive: smaller stacks

inline void init_block(size_t stack_bytes = kStack) {
    State& s = S();
    int n = s.nthreads;
    if ((int)s.fibers.size() < n) {
        size_t old = s.fibers.size();
        s.fibers.resize(n);
        for (size_t i = old; i < (size_t)n; ++i) s.fibers[i].stack = (char*)malloc(stack_bytes);
    }
    int nw = (n + 63) / 64;
    s.wv_arrived.assign(nw, 0);
    s.wv_gen.assign(nw, 0);
    s.wv_alive.assign(nw, 0);
    for (int t = 0; t < n; ++t) s.wv_alive[t >> 6]++;
    s.slot_a.assign((size_t)n * 2, 0);
    s.slot_b.assign((size_t)n * 2, 0);
    s.alive = n;
    s.bar_arrived = 0;
    s.bar_gen = 0;
    for (int t = 0; t < n; ++t) {
        State::Fiber& f = s.fibers[t];
        f.done = false;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = stack_bytes;
        f.ctx.uc_link = &s.sched_ctx;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    }
}

// one round-robin pass over the live fibers of the current block
inline void pass_block() {
    State& s = S();
    const int n = s.nthreads;
    unsigned bx = s.blockDim_.x, by = s.blockDim_.y;
    for (int t = 0; t < n; ++t) {
        if (s.fibers[t].done) continue;
        s.cur = t;
        s.threadIdx_ = dim3(t % bx, (t / bx) % by, t / (bx * by));
        swapcontext(&s.sched_ctx, &s.fibers[t].ctx);
    }
}

inline void run_block() {
    State& s = S();
    init_block();
    while (s.alive > 0) {
        long before = s.progress;
        pass_block();
        if (s.progress == before && s.alive > 0) {
            fprintf(stderr, "[hip_emu] DEADLOCK in block (%u,%u,%u): divergent barrier / wave op\n",
                    s.blockIdx_.x, s.blockIdx_.y, s.blockIdx_.z);
            abort();
        }
    }
}

inline void launch(dim3 grid, dim3 block, size_t smem, std::function<void()> body) {
    State& s = S();
    s.gridDim_ = grid;
    s.blockDim_ = block;
    s.nthreads = (int)(block.x * block.y * block.z);
    s.body = body;
    std::vector<char> dyn(smem + 64);
    s.dyn_smem = (char*)(((uintptr_t)dyn.data() + 15) & ~(uintptr_t)15);
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                s.blockIdx_ = dim3(x, y, z);
                run_block();
            }
}

// Cooperative launch: every block of the grid is alive at the same time, so blocks may wait for each other through global
// memory (the persistent any-size decode kernel hands vectors from workgroup to workgroup inside one launch).  Each block has
// its own State (fibers, barrier counters, dynamic LDS); the scheduler gives every block one pass over its fibers in turn.
// A fiber that polls memory must yield (WN_SLEEP / emu::yield_()).  Static `__shared__` arrays are ONE object for all blocks
// here: kernels launched this way use dynamic shared memory only.  A grid in which nothing progresses for a long time is
// reported as a deadlock.
inline void launch_coop(dim3 grid, dim3 block, size_t smem, std::function<void()> body) {
    State*& cur = cur_state_();
    State* const saved = cur;
    const int nb = (int)(grid.x * grid.y * grid.z);
    std::vector<State*> st(nb);
    std::vector<std::vector<char>> dyn(nb);
    for (int b = 0; b < nb; ++b) {
        State* s = new State();
        st[b] = s;
        s->gridDim_ = grid;
        s->blockDim_ = block;
        s->nthreads = (int)(block.x * block.y * block.z);
        s->body = body;
        dyn[b].assign(smem + 64, 0);
        s->dyn_smem = (char*)(((uintptr_t)dyn[b].data() + 15) & ~(uintptr_t)15);
        s->blockIdx_ = dim3(b % grid.x, (b / grid.x) % grid.y, b / (grid.x * grid.y));
        cur = s;
        init_block(kStackCoop);
    }
    long idle_passes = 0;
    for (;;) {
        int alive = 0;
        long progress = 0;
        for (int b = 0; b < nb; ++b) {
            if (st[b]->alive <= 0) continue;
            cur = st[b];
            const long before = st[b]->progress;
            pass_block();
            progress += st[b]->progress - before;
            alive += st[b]->alive > 0;
        }
        if (alive == 0) break;
        idle_passes = progress ? 0 : idle_passes + 1;
        if (idle_passes > 200000) {
            fprintf(stderr, "[hip_emu] DEADLOCK in a cooperative launch: no block made progress for %ld passes\n", idle_passes);
            abort();
        }
    }
    for (int b = 0; b < nb; ++b) {
        for (auto& f : st[b]->fibers) free(f.stack);
        delete st[b];
    }
    cur = saved;
}

// ---- cross-lane exchange ---------------------------------------------------------------
template <class T>
inline T shfl_idx(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "shfl payload");
    State& s = S();
    int t = tid_();
    int base = t & ~63;
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    s.slot_a[2 * (size_t)t] = raw;  // every thread owns slots [2t, 2t+1] (the bf16 MFMA payload is 16 bytes): no aliasing
    sync_wave();
    int src = base + (src_lane & 63);
    if (src >= s.nthreads) src = t;
    uint64_t got = s.slot_a[2 * (size_t)src];
    sync_wave();
    T r;
    memcpy(&r, &got, sizeof(T));
    return r;
}

typedef float f32x16_t __attribute__((vector_size(64)));
typedef float f32x4_t __attribute__((vector_size(16)));

inline f32x16_t mfma_f32_32x32x2f32(float a, float b, f32x16_t c) {
    State& s = S();
    int t = tid_();
    int base = t & ~63, l = t & 63;
    memcpy(&s.slot_a[2 * (size_t)t], &a, 4);
    memcpy(&s.slot_b[2 * (size_t)t], &b, 4);
    sync_wave();
    int col = l & 31, hi = l >> 5;
    f32x16_t d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float a0, a1, b0, b1;
        memcpy(&a0, &s.slot_a[2 * (size_t)(base + row)], 4);       // A[row][k=0] lives in lane row
        memcpy(&a1, &s.slot_a[2 * (size_t)(base + 32 + row)], 4);  // A[row][k=1] lives in lane 32+row
        memcpy(&b0, &s.slot_b[2 * (size_t)(base + col)], 4);       // B[k=0][col]
        memcpy(&b1, &s.slot_b[2 * (size_t)(base + 32 + col)], 4);  // B[k=1][col]
        d[r] = fmaf(a1, b1, fmaf(a0, b0, c[r]));
    }
    sync_wave();
    return d;
}

// v_mfma_f32_32x32x16_bf16: lane l holds A[i = l&31][k = 8*(l>>5) .. +7] and B[k = 8*(l>>5) .. +7][j = l&31]
// as 8 bf16 each; products are exact in f32, accumulation in f32 (k ascending); D layout as 32x32x2.
inline float bf16_bits_to_float(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// IEEE binary16 <-> binary32 (round to nearest even, overflow -> inf, subnormals kept): v_cvt_f16_f32 / v_cvt_f32_f16
inline float f16_bits_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const int e = (h >> 10) & 31;
    const uint32_t m = h & 0x3ffu;
    float f;
    if (e == 31) {
        const uint32_t u = sign | 0x7f800000u | (m << 13);
        memcpy(&f, &u, 4);
        return f;
    }
    if (e == 0) {
        f = ldexpf((float)m, -24);
        return sign ? -f : f;
    }
    const uint32_t u = sign | ((uint32_t)(e + 112) << 23) | (m << 13);
    memcpy(&f, &u, 4);
    return f;
}
inline uint16_t float_to_f16_bits(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const uint32_t a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (a > 0x7f800000u ? 0x200u : 0u));   // inf / nan
    float ax;
    memcpy(&ax, &a, 4);
    if (ax >= 65520.0f) return (uint16_t)(sign | 0x7c00u);     // rounds to inf
    if (ax < 6.103515625e-05f) {                               // below the smallest normal 2^-14: multiples of 2^-24
        const float q = ax * 16777216.0f;                      // exact
        const float r = nearbyintf(q);                         // to nearest even (default rounding mode)
        return (uint16_t)(sign | (uint16_t)r);                 // r = 1024 is the smallest normal: same bit pattern
    }
    const uint32_t r = a + 0xfffu + ((a >> 13) & 1u);          // to nearest even at bit 13
    return (uint16_t)(sign | (uint16_t)((r - 0x38000000u) >> 13));
}
inline f32x16_t mfma_f32_32x32x16bf16(const uint16_t* a8, const uint16_t* b8, f32x16_t c) {
    State& s = S();
    int t = tid_();
    int base = t & ~63, l = t & 63;
    memcpy(&s.slot_a[2 * (size_t)t], a8, 16);
    memcpy(&s.slot_b[2 * (size_t)t], b8, 16);
    sync_wave();
    int col = l & 31, hi = l >> 5;
    f32x16_t d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int kb = 0; kb < 2; ++kb) {
            uint16_t av[8], bv[8];
            memcpy(av, &s.slot_a[2 * (size_t)(base + 32 * kb + row)], 16);
            memcpy(bv, &s.slot_b[2 * (size_t)(base + 32 * kb + col)], 16);
            for (int e = 0; e < 8; ++e) acc = fmaf(bf16_bits_to_float(av[e]), bf16_bits_to_float(bv[e]), acc);
        }
        d[r] = acc;
    }
    sync_wave();
    return d;
}

inline f32x16_t mfma_f32_32x32x16f16(const uint16_t* a8, const uint16_t* b8, f32x16_t c) {
    State& s = S();
    int t = tid_();
    int base = t & ~63, l = t & 63;
    memcpy(&s.slot_a[2 * (size_t)t], a8, 16);
    memcpy(&s.slot_b[2 * (size_t)t], b8, 16);
    sync_wave();
    int col = l & 31, hi = l >> 5;
    f32x16_t d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int kb = 0; kb < 2; ++kb) {
            uint16_t av[8], bv[8];
            memcpy(av, &s.slot_a[2 * (size_t)(base + 32 * kb + row)], 16);
            memcpy(bv, &s.slot_b[2 * (size_t)(base + 32 * kb + col)], 16);
            for (int e = 0; e < 8; ++e) acc = fmaf(f16_bits_to_float(av[e]), f16_bits_to_float(bv[e]), acc);
        }
        d[r] = acc;
    }
    sync_wave();
    return d;
}

inline f32x4_t mfma_f32_16x16x4f32(float a, float b, f32x4_t c) {
    State& s = S();
    int t = tid_();
    int base = t & ~63, l = t & 63;
    memcpy(&s.slot_a[2 * (size_t)t], &a, 4);
    memcpy(&s.slot_b[2 * (size_t)t], &b, 4);
    sync_wave();
    int col = l & 15, q = l >> 4;
    f32x4_t d = c;
    for (int r = 0; r < 4; ++r) {
        int row = q * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            memcpy(&av, &s.slot_a[2 * (size_t)(base + 16 * k + row)], 4);  // A[row][k] in lane 16k+row
            memcpy(&bv, &s.slot_b[2 * (size_t)(base + 16 * k + col)], 4);  // B[k][col] in lane 16k+col
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    sync_wave();
    return d;
}

}  // namespace emu

// ---- HIP surface syntax ------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__
#endif
#define threadIdx (emu::S().threadIdx_)
#define blockIdx (emu::S().blockIdx_)
#define blockDim (emu::S().blockDim_)
#define gridDim (emu::S().gridDim_)

inline void __syncthreads() { emu::sync_block(); }
template <class T>
inline T __shfl_xor(T v, int mask, int width = 64) {
    (void)width;
    int l = emu::tid_() & 63;
    return emu::shfl_idx(v, l ^ mask);
}
template <class T>
inline T __shfl_down(T v, unsigned delta, int width = 64) {
    (void)width;
    int l = emu::tid_() & 63;
    int src = l + (int)delta;
    if (src > 63) src = l;
    return emu::shfl_idx(v, src);
}
template <class T>
inline T __shfl(T v, int src, int width = 64) {
    (void)width;
    return emu::shfl_idx(v, src);
}
inline float atomicAdd(float* p, float v) {
    float o = *p;
    *p = o + v;
    return o;
}
inline int atomicAdd(int* p, int v) {
    int o = *p;
    *p = o + v;
    return o;
}
inline unsigned atomicAdd(unsigned* p, unsigned v) {
    unsigned o = *p;
    *p = o + v;
    return o;
}
inline float __frcp_rn(float x) { return 1.0f / x; }
struct float4 {
    float x, y, z, w;
};
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
