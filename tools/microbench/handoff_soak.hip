// Soak of the fence-free hand-off of csrc/wn_dlpf.hip (gfx950), with the chip busy around it.
//
// The persistent decode hands a stage's vector from 64 workgroups ("units") to the same 64 workgroups without a fence:
//   producer   every thread: agent-scope (sc1, write-through) stores of its part; s_waitcnt vmcnt(0); workgroup barrier; ONE
//              thread: agent-scope store of the unit's flag (the stage's tag; flags only grow)
//   consumer   first wave: one lane per unit polls the flags with agent-scope loads; workgroup barrier; the WHOLE vector comes
//              in as buffer_load_dwordx4 ... lds with the sc1 cache policy; s_waitcnt vmcnt(0); barrier; tiles read the LDS
// (DESIGN.md 3.4 has the memory-model argument.)  tools/microbench/handoff.hip measured 0 stale words in 2000 trips between
// TWO workgroups on an idle chip.  This is the same protocol in the decode kernel's own shape -- 64 units spread over all 8
// XCDs, two vectors alive (stage parity), every unit reads everything every stage, EVERY word of EVERY stage checked -- for
// >= 1e6 stages, while the other 192 workgroups of the launch stream through 1 GiB (reads + writes: L2 evictions, HBM traffic,
// fabric contention on every XCD).
//
//   hipcc --offload-arch=gfx950 -O3 -o handoff_soak handoff_soak.hip && ./handoff_soak [stages=1000000] [traffic=1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64;
#define NU 64            // units (workgroups of the hand-off)
#define PER 256          // floats a unit contributes per stage (the decode kernel: 16 rows x 16 columns)
#define VEC (NU * PER)   // 16384 floats = 64 KB per stage
#define SPIN_MAX (1 << 26)

struct Report {
    u64 ticks;
    u64 stale;
    unsigned timeout;
    unsigned xcc[NU];
    u64 traffic_bytes;
};

static __device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; }
static __device__ __forceinline__ float expect(unsigned stage, int i) { return (float)((stage * 2654435761u + (unsigned)i * 40503u) & 0xffffffu); }

__global__ __launch_bounds__(512) void k_soak(float* vec /*[2][VEC]*/, u64* flags /*[NU]*/, unsigned* stop, float* big, size_t big_floats,
                                              unsigned stages, Report* rep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_v = reinterpret_cast<float*>(smem);
    __shared__ int s_to;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u = blockIdx.x;
    if (u >= NU) {   // ---- traffic: read-modify-write sweeps over `big` until the units are done ----
        const size_t nwg = gridDim.x - NU, w = u - NU;
        const size_t chunk = big_floats / nwg;
        float4* p = reinterpret_cast<float4*>(big + w * chunk);
        u64 moved = 0;
        while (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            for (size_t i = tid; i < chunk / 4; i += 512) {
                float4 v = p[i];
                v.x += 1.0f; v.y += 2.0f;
                p[i] = v;
            }
            moved += chunk * 8;
        }
        if (tid == 0) atomicAdd(&rep->traffic_bytes, moved);
        return;
    }
    if (tid == 0) { s_to = 0; rep->xcc[u] = xcc_id(); }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(vec, (short)0, 2 * VEC * 4, 0x00020000);
    u64 bad = 0;
    u64 t0 = 0;
    for (unsigned st = 1; st <= stages + 1; ++st) {
        if (st == 2 && u == 0 && tid == 0) t0 = wall_clock64();
        if (st >= 2) {
            // consumer of stage st-1: flags, barrier, the whole vector by LDS-DMA with sc1, every word checked
            if (wave == 0) {
                int spin = 0;
                while ((unsigned)(__hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < st - 1) {
                    if (++spin > SPIN_MAX || s_to) { s_to = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (s_to) break;
            const int par = (st - 1) & 1;
            for (int off = wave * 1024; off < VEC * 4; off += 8 * 1024)   // 16 bytes per lane, 1 KB per instruction and wave
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, (__attribute__((address_space(3))) void*)(smem + off), 16, lane * 16,
                                                         par * VEC * 4 + off, 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int i = tid; i < VEC; i += 512)
                if (s_v[i] != expect(st - 1, i)) bad++;
            __syncthreads();   // (the LDS image is overwritten by the next stage's transfers)
        }
        if (st <= stages) {
            // producer of stage st: this unit's PER floats, agent-scope stores, vmcnt(0), barrier, flag
            const int par = st & 1;
            if (tid < PER) __hip_atomic_store(vec + par * VEC + u * PER + tid, expect(st, u * PER + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flags + u, (u64)st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (u == 0 && tid == 0) rep->ticks = wall_clock64() - t0;
    if (bad) atomicAdd(&rep->stale, bad);
    if (tid == 0 && s_to) rep->timeout = 1;
    // the last unit out stops the traffic
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(flags + u, (u64)0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (u == 0) {
            for (int k = 0; k < NU; ++k) {
                int spin = 0;
                while (__hip_atomic_load(flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (u64)0xffffffffu && ++spin < SPIN_MAX) __builtin_amdgcn_s_sleep(4);
            }
            __hip_atomic_store(stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const unsigned stages = argc > 1 ? (unsigned)atol(argv[1]) : 1000000u;
    const int traffic = argc > 2 ? atoi(argv[2]) : 1;
    Report* rep;
    CK(hipMalloc(&rep, sizeof(Report)));
    float* vec; u64* flags; unsigned* stop; float* big;
    const size_t big_floats = (size_t)1 << 28;   // 1 GiB
    CK(hipMalloc(&vec, 2 * VEC * 4)); CK(hipMalloc(&flags, NU * 8)); CK(hipMalloc(&stop, 64)); CK(hipMalloc(&big, big_floats * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_soak), hipFuncAttributeMaxDynamicSharedMemorySize, VEC * 4));
    int cus = 0, per = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_soak, 512, VEC * 4));
    const int grid = traffic ? (cus * per < 256 ? cus * per : 256) : NU;
    printf("device: %d CUs x %d resident workgroups; grid %d (%d units + %d traffic workgroups)\n", cus, per, grid, NU, grid - NU);
    if (grid < NU) { printf("not enough resident workgroups\n"); return 1; }
    for (int round = 0; round < 2; ++round) {
        const unsigned n = round == 0 ? 2000u : stages;   // (a short warm-up round first)
        CK(hipMemset(rep, 0, sizeof(Report))); CK(hipMemset(vec, 0, 2 * VEC * 4)); CK(hipMemset(flags, 0, NU * 8)); CK(hipMemset(stop, 0, 64));
        CK(hipMemset(big, 0, big_floats * 4));
        hipLaunchKernelGGL(k_soak, dim3(grid), dim3(512), VEC * 4, 0, vec, flags, stop, big, big_floats, n, rep);
        CK(hipDeviceSynchronize());
        Report h;
        CK(hipMemcpy(&h, rep, sizeof(h), hipMemcpyDeviceToHost));
        unsigned hist[16] = {0};
        for (int k = 0; k < NU; ++k) hist[h.xcc[k] & 15]++;
        const double secs = (double)h.ticks * 1e-8;
        printf("%u stages of 64 KB over 64 units (units per XCD:", n);
        for (int x = 0; x < 8; ++x) printf(" %u", hist[x]);
        printf("): %.3f us per stage, %llu words checked, STALE %llu%s; traffic beside it %.1f GB = %.2f TB/s\n", secs * 1e6 / n,
               (unsigned long long)n * VEC * NU, (unsigned long long)h.stale, h.timeout ? "  TIMEOUT" : "", h.traffic_bytes * 1e-9,
               secs > 0 ? h.traffic_bytes * 1e-12 / secs : 0.0);
    }
    return 0;
}
