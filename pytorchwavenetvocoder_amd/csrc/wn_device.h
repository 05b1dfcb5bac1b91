// wn_device.h -- device-side common definitions for the gfx950 (CDNA4 / MI355X) kernels.
//
// The product build is hipcc --offload-arch=gfx950.  The same sources also compile with g++
// and -DWN_EMU against tests/emu/hip_emu.h (TEST INFRASTRUCTURE: index-math checks without a
// GPU).  Everything architecture specific is funnelled through this header:
//   * f32x16 / mfma32(): v_mfma_f32_32x32x2_f32 -- exact f32 (k-ordered fma chain) at the f32
//     vector rate; lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31]; the 16 accumulator
//     registers of lane l hold D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31].
//   * WN_LAUNCH(): kernel launch on an explicit stream.
#pragma once

#include <stddef.h>
#include <stdint.h>

#define WN_VOFF_DEAD 0x7ffffff0   // per-lane byte offset of a lane whose buffer store must not land
#ifdef WN_EMU
#include "hip_emu.h"
typedef emu::f32x16_t f32x16;
typedef void* wn_stream_t;
#define WN_LAUNCH(kernel, grid, block, smem, stream, ...) \
    emu::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })
// launch whose workgroups wait for each other inside the kernel (all of them resident at once; wn_dlp.hip)
#define WN_LAUNCH_COOP(kernel, grid, block, smem, stream, ...) \
    emu::launch_coop((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })
// workgroups of `kernel` the device keeps resident at once (the emulator keeps any grid alive); WN_COOP_CAPACITY: test knob
#include <stdlib.h>
static inline int wn_coop_capacity_override() {
    const char* e = getenv("WN_COOP_CAPACITY");
    return e && *e ? atoi(e) : -1;
}
template <typename Kn>
static inline int wn_coop_capacity(Kn, int, size_t) {
    const int o = wn_coop_capacity_override();
    return o >= 0 ? o : 0x7fffffff;
}
#define WN_COOP_MAXDEV 64
template <typename Kn>
static inline int wn_coop_capacity_cached(int (&)[WN_COOP_MAXDEV], bool&, Kn k, int b, size_t l) { return wn_coop_capacity(k, b, l); }
#define WN_DYN_SMEM(name) char* name = emu::S().dyn_smem
static inline f32x16 mfma32(float a, float b, f32x16 c) { return emu::mfma_f32_32x32x2f32(a, b, c); }
typedef emu::f32x4_t f32x4;
static inline f32x4 mfma16(float a, float b, f32x4 c) { return emu::mfma_f32_16x16x4f32(a, b, c); }
#define WN_UNROLL
#define WN_UNROLL_N(n)
#define WN_NOUNROLL
// buffer access: base (wave-uniform) + per-lane byte offset (voff) + wave-uniform byte offset (soff)
// like the hardware, accesses beyond num_records read 0 / are dropped
struct wn_rsrc_t {
    const char* base;
    unsigned bytes;
};
static inline wn_rsrc_t wn_make_buf(const void* p, unsigned bytes) { return wn_rsrc_t{(const char*)p, bytes}; }
static inline float wn_buf_load(wn_rsrc_t r, int voff, int soff) {
    const unsigned long off = (unsigned long)(unsigned)voff + (unsigned long)(unsigned)soff;
    return off + 4 <= r.bytes ? *(const float*)(r.base + off) : 0.0f;
}
static inline void wn_buf_store(wn_rsrc_t r, float v, int voff, int soff) {
    const unsigned long off = (unsigned long)(unsigned)voff + (unsigned long)(unsigned)soff;
    if (off + 4 <= r.bytes) *(float*)(r.base + off) = v;
}
static inline float wn_buf_load_once(wn_rsrc_t r, int voff, int soff) { return wn_buf_load(r, voff, soff); }
static inline float4 wn_buf_load4(wn_rsrc_t r, int voff, unsigned soff) {
    const unsigned long off = (unsigned long)(unsigned)voff + (unsigned long)soff;
    float4 v{0.f, 0.f, 0.f, 0.f};
    if (off + 16 <= r.bytes) memcpy(&v, r.base + off, 16);   // 4-byte aligned sources (shifted taps)
    return v;
}
// global -> LDS without registers: lane l of the wave writes 16 bytes at lds_wave_base + 16*l
static inline void wn_buf_load_lds16(wn_rsrc_t r, char* lds_wave_base, int voff, unsigned soff) {
    const float4 v = wn_buf_load4(r, voff, soff);
    memcpy(lds_wave_base + 16 * (threadIdx.x & 63), &v, 16);
}
static inline void wn_buf_load_lds16_coherent(wn_rsrc_t r, char* lds_wave_base, int voff, unsigned soff) { wn_buf_load_lds16(r, lds_wave_base, voff, soff); }
static inline void wn_store_coherent(float* p, float v) { *p = v; }
static inline void wn_store_coherent_int(int* p, int v) { *p = v; }
static inline int wn_load_coherent_int(const int* p) { return *p; }
// global -> LDS, 4 bytes per lane: lane l of the wave writes at lds_wave_base + 4*l (inactive lanes write nothing)
static inline void wn_buf_load_lds4(wn_rsrc_t r, char* lds_wave_base, int voff, unsigned soff) {
    const float v = wn_buf_load(r, voff, (int)soff);
    memcpy(lds_wave_base + 4 * (threadIdx.x & 63), &v, 4);
}
#define WN_WAIT_VMCNT(n)
#define WN_UNIFORM(x) (x)
#define WN_SCHED_BARRIER()
#define WN_SLEEP(n) emu::yield_()   // a polling wave lets the other fibers run
#define WN_HW_WAVE_SLOT() 0
#define WN_SGB_DS(n)
#define WN_SGB_MFMA(n)
#define WN_SGB_VALU(n)
#define WN_SCHED_FENCE_ALU()
#define WN_SGB_DSW(n)
#define WN_SGB_VMEM(n)
// workgroup barrier that orders LDS traffic only (outstanding global loads stay in flight)
#define WN_LDS_BARRIER() __syncthreads()
// load that must observe earlier stores of other waves of the same workgroup (bypasses the L1)
static inline float wn_ld_coherent(const float* p) { return *p; }
// 8-byte granule {value, tag} handed from one workgroup to another: ONE agent-scope store, ONE agent-scope load
static inline void wn_granule_store(unsigned long long* p, float v, unsigned tag) {
    unsigned b;
    memcpy(&b, &v, 4);
    *p = ((unsigned long long)tag << 32) | b;
}
static inline unsigned long long wn_granule_load(const unsigned long long* p) { return *p; }
// producer / consumer flag of a hand-off of PLAIN data (wn_dlpf.hip): release store after the workgroup's data stores (and a
// workgroup barrier), acquire fence before the consumer's data loads
static inline void wn_flag_release(unsigned long long* p, unsigned long long v) { *p = v; }
static inline void wn_fence_acquire() {}
// v + the value of lane (l ^ m); all lanes of an aligned group of 2m hold the same partial sums
static inline float wn_xor_add(float v, int m) { return v + __shfl_xor(v, m, 64); }
// bf16 matrix-core step (v_mfma_f32_32x32x16_bf16): a / b = 8 bf16 per lane packed in a float4
typedef float4 wn_f4;  // 16-byte register quad
static inline wn_f4 wn_ld4_unaligned(const float* p) { return wn_f4{p[0], p[1], p[2], p[3]}; }
static inline wn_f4 wn_ld4_stream(const wn_f4* p) { return *p; }
static inline f32x16 mfma_bf16(const wn_f4& a, const wn_f4& b, f32x16 c) {
    return emu::mfma_f32_32x32x16bf16(reinterpret_cast<const uint16_t*>(&a), reinterpret_cast<const uint16_t*>(&b), c);
}
// two fp32 -> two bf16 (round to nearest even) packed lo | hi << 16
static inline unsigned wn_pk_bf16(float a, float b) {
    auto one = [](float x) -> unsigned {
        unsigned u;
        memcpy(&u, &x, 4);
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    };
    return one(a) | (one(b) << 16);
}
// fp16 matrix-core step (v_mfma_f32_32x32x16_f16) and the fp16 pair split of the weight gradients (wn_gemm6.hip, F16)
static inline f32x16 mfma_f16(const wn_f4& a, const wn_f4& b, f32x16 c) {
    return emu::mfma_f32_32x32x16f16(reinterpret_cast<const uint16_t*>(&a), reinterpret_cast<const uint16_t*>(&b), c);
}
// two fp32 -> two fp16 (round to nearest even, overflow -> inf) packed lo | hi << 16
static inline unsigned wn_pk_f16(float a, float b) { return (unsigned)emu::float_to_f16_bits(a) | ((unsigned)emu::float_to_f16_bits(b) << 16); }
static inline float wn_f16lo_f32(unsigned u) { return emu::f16_bits_to_float((uint16_t)(u & 0xffffu)); }
static inline float wn_f16hi_f32(unsigned u) { return emu::f16_bits_to_float((uint16_t)(u >> 16)); }
static inline float wn_bits_f32(unsigned u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline unsigned wn_f32_bits(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return u;
}
// two-lane fp32 vector for v_pk_fma_f32
struct f32x2 {
    float x, y;
};
static inline f32x2 wn_pk_fma(f32x2 a, f32x2 b, f32x2 c) { return f32x2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
#else
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef hipStream_t wn_stream_t;
#define WN_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), (smem), (stream), __VA_ARGS__)
// A launch whose workgroups wait for each other inside the kernel: a plain launch of a grid that is resident as a whole (at
// most one workgroup per CU here; MI355X_MICROARCH.md: plain, cooperative and graph launches give the same residency)
#define WN_LAUNCH_COOP(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), (smem), (stream), __VA_ARGS__)
// Workgroups of `kernel` (block threads, `lds` bytes of dynamic LDS) the current device keeps resident at once = occupancy per
// CU x CUs: the bound a WN_LAUNCH_COOP grid is checked against BEFORE it is launched (a partitioned GPU -- CPX / DPX --, a part
// with fewer CUs).  0 when the query fails.  The caller caches the value per kernel (the devices of a node are alike).
// WN_COOP_CAPACITY=<n> in the environment overrides the query (test knob: the error / fall-back paths on a full chip).
#include <stdlib.h>
static inline int wn_coop_capacity_override() {
    const char* e = getenv("WN_COOP_CAPACITY");
    return e && *e ? atoi(e) : -1;
}
template <typename Kn>
static inline int wn_coop_capacity(Kn kernel, int block, size_t lds) {
    const int o = wn_coop_capacity_override();
    if (o >= 0) return o;
    int dev = 0, cus = 0, per = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kernel, block, lds) != hipSuccess) {
        (void)hipGetLastError();   // a failed query must not leave a sticky error for the caller's next launch check
        return 0;
    }
    return per * cus;
}
// The same, cached PER DEVICE (one process may decode on several devices, and a node's devices need not be alike:
// partitioned GPUs).  `cache` = a static array of WN_COOP_MAXDEV ints initialised to -1 by the caller's first use.
#define WN_COOP_MAXDEV 64
template <typename Kn>
static inline int wn_coop_capacity_cached(int (&cache)[WN_COOP_MAXDEV], bool& init, Kn kernel, int block, size_t lds) {
    const int o = wn_coop_capacity_override();
    if (o >= 0) return o;
    if (!init) {
        for (int i = 0; i < WN_COOP_MAXDEV; ++i) cache[i] = -1;
        init = true;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (dev < 0 || dev >= WN_COOP_MAXDEV) return wn_coop_capacity(kernel, block, lds);
    if (cache[dev] < 0) cache[dev] = wn_coop_capacity(kernel, block, lds);
    return cache[dev];
}
#define WN_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
static __device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x4_f32 (exact f32, 8 passes): lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; its 4
// accumulator registers hold D[row = 4 (l >> 4) + r][col = l & 15]
typedef float f32x4 __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// Buffer access (CDNA "MUBUF"): the 128-bit resource descriptor and the scalar offset live in SGPRs,
// only the per-lane byte offset needs a VGPR -> a tile's 32 channel rows cost ONE address VGPR
// (voff = time) plus an SGPR per row (soff = channel * T * 4) instead of 32 64-bit address pairs.
// Loads of dead lanes clamp voff to 0 (a valid dummy address).  STORES of dead lanes carry WN_VOFF_DEAD instead of sitting
// under a lane-conditional branch: an offset past num_records is dropped by the hardware range check (and by the host
// emulation), and the store stays in the straight-line instruction stream.
typedef __amdgpu_buffer_rsrc_t wn_rsrc_t;
static __device__ __forceinline__ wn_rsrc_t wn_make_buf(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
static __device__ __forceinline__ float wn_buf_load(wn_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
static __device__ __forceinline__ void wn_buf_store(wn_rsrc_t r, float v, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, voff, soff, 0);
}
// load of data that is read exactly once per launch: non-temporal (does not displace the lines a later launch re-reads)
static __device__ __forceinline__ float wn_buf_load_once(wn_rsrc_t r, int voff, int soff) {
#ifndef WN_NO_NT_LOADS   // (-DWN_NO_NT_LOADS: A/B builds)
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 2));
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
#endif
}
// global -> LDS without registers (buffer_load_dwordx4 ... lds): lane l of the wave writes 16 bytes at
// lds_wave_base + 16*l; lds_wave_base must be wave-uniform.  Completion is counted by vmcnt.
static __device__ __forceinline__ void wn_buf_load_lds16(wn_rsrc_t r, char* lds_wave_base, int voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, (int)soff, 0, 0);
}
// the same with the agent-scope cache policy (sc1): never served from a stale line of this XCD's L2 -- the read side of a
// hand-off whose data was written with wn_store_coherent (write-through) by a workgroup on another XCD
static __device__ __forceinline__ void wn_buf_load_lds16_coherent(wn_rsrc_t r, char* lds_wave_base, int voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, (int)soff, 0, 16);
}
static __device__ __forceinline__ void wn_store_coherent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the error word of a persistent launch: written by workgroups that gave up, read by workgroups that start later -- possibly on
// another XCD, whose L2 would not see a plain store before the kernel ends
static __device__ __forceinline__ void wn_store_coherent_int(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ int wn_load_coherent_int(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the same with 4 bytes per lane: lane l writes at lds_wave_base + 4*l
static __device__ __forceinline__ void wn_buf_load_lds4(wn_rsrc_t r, char* lds_wave_base, int voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 4, voff, (int)soff, 0, 0);
}
// s_waitcnt vmcnt(n) only (n <= 15)
#define WN_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | (n))
// 16-byte load: per-lane byte offset in a VGPR, wave-uniform byte offset in an SGPR (no 64-bit
// per-lane address registers -- a register ring of weights needs none)
static __device__ __forceinline__ float4 wn_buf_load4(wn_rsrc_t r, int voff, unsigned soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, (int)soff, 0));
}
// make a value the compiler can prove wave-uniform (it IS uniform: derived from the wave id)
#define WN_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
// scheduling fence: hipcc may not move instructions across it (pins software-pipeline issue order)
#define WN_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#define WN_SLEEP(n) __builtin_amdgcn_s_sleep(n)
// wave slot of this wave on its SIMD: HW_REG_HW_ID (4), bits [3:0]
#define WN_HW_WAVE_SLOT() (__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 15)
// fence for VALU, MFMA and global-memory instructions: only LDS and scalar instructions may still be moved across it
#define WN_SCHED_FENCE_ALU() __builtin_amdgcn_sched_barrier(0x384)
// scheduling groups: "the next n DS reads" / "the next n MFMAs" are emitted as a block in this order
#define WN_SGB_DS(n) __builtin_amdgcn_sched_group_barrier(0x100, (n), 0)
#define WN_SGB_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, (n), 0)
#define WN_SGB_VALU(n) __builtin_amdgcn_sched_group_barrier(0x002, (n), 0)
#define WN_SGB_DSW(n) __builtin_amdgcn_sched_group_barrier(0x200, (n), 0)
#define WN_SGB_VMEM(n) __builtin_amdgcn_sched_group_barrier(0x020, (n), 0)
// s_barrier preceded by lgkmcnt(0) only: LDS writes are visible afterwards, while prefetched
// global loads stay outstanding (a __syncthreads() would also wait for vmcnt(0)).
#define WN_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
static __device__ __forceinline__ float wn_ld_coherent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 8-byte granule {value, tag} handed from one workgroup to another (MI355X_MICROARCH.md, hand-off price list): ONE relaxed
// agent-scope store (global_store_dwordx2 sc1) per lane, the consumer polls the granule itself with relaxed agent-scope loads
// -- valid across XCDs without fences (measured: 0 stale words, 0.7 - 0.8 us per hop, profiles/r04/handoff_microbench.txt)
static __device__ __forceinline__ void wn_granule_store(unsigned long long* p, float v, unsigned tag) {
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ unsigned long long wn_granule_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// producer / consumer flag of a hand-off of PLAIN data (wn_dlpf.hip).  Release at agent scope = write back this XCD's L2 and
// wait for the wave's stores before the flag store; acquire = invalidate the non-coherent lines of L1 / L2 after the flag load:
// the memory model's own sequences, 1.6 us per hop across XCDs with 0 stale words (profiles/r04/handoff_microbench.txt)
// (the sequence tools/microbench/handoff.hip validated: every wave has waited for its own stores and passed a workgroup barrier
// before ONE thread calls wn_flag_release; the consumer polls with wn_granule_load, then wn_fence_acquire, then a barrier)
static __device__ __forceinline__ void wn_flag_release(unsigned long long* p, unsigned long long v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ void wn_fence_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
// v + partner value, partner in the other half of the aligned 2m-lane group.  m = 1,2,4,8 are DPP
// moves (quad_perm / row_half_mirror / row_mirror: valid because after the previous steps every
// lane of an m-lane group holds the same partial sum); larger m goes through ds_bpermute.
static __device__ __forceinline__ float wn_xor_add(float v, int m) {
    const int iv = __builtin_bit_cast(int, v);
    int o;
    switch (m) {
        case 1: o = __builtin_amdgcn_update_dpp(0, iv, 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]
        case 2: o = __builtin_amdgcn_update_dpp(0, iv, 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]
        case 4: o = __builtin_amdgcn_update_dpp(0, iv, 0x141, 0xF, 0xF, true); break;  // row_half_mirror
        case 8: o = __builtin_amdgcn_update_dpp(0, iv, 0x140, 0xF, 0xF, true); break;  // row_mirror
        default: return v + __shfl_xor(v, m, 64);
    }
    return v + __builtin_bit_cast(float, o);
}
// bf16 matrix-core step (v_mfma_f32_32x32x16_bf16, 16x the f32 MFMA rate): a / b = 8 bf16 per lane
typedef __bf16 wn_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wn_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wn_f4 __attribute__((ext_vector_type(4)));  // 16-byte register quad (first-class vector: stays in VGPRs)
// one global_load_dwordx4 from a 4-byte aligned address (shifted conv taps)
static __device__ __forceinline__ wn_f4 wn_ld4_unaligned(const float* p) {
    typedef float v4u __attribute__((ext_vector_type(4), aligned(4)));
    return *reinterpret_cast<const v4u*>(p);
}
// 16-byte load of data that is read once (a weight stream): non-temporal
static __device__ __forceinline__ wn_f4 wn_ld4_stream(const wn_f4* p) { return __builtin_nontemporal_load(p); }
static __device__ __forceinline__ f32x16 mfma_bf16(wn_f4 a, wn_f4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wn_bf16x8, a), __builtin_bit_cast(wn_bf16x8, b), c, 0, 0, 0);
}
// two fp32 -> two bf16 (round to nearest even) packed lo | hi << 16: one v_cvt_pk_bf16_f32
static __device__ __forceinline__ unsigned wn_pk_bf16(float a, float b) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, wn_bf16x2));
}
// fp16 matrix-core step (v_mfma_f32_32x32x16_f16, the bf16 step's rate and fragment layout) and the conversions of the fp16 pair
// split of the weight gradients (wn_gemm6.hip, F16): v_cvt_pk_f16_f32 rounds to nearest even, overflow gives inf
typedef _Float16 wn_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 wn_f16x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ f32x16 mfma_f16(wn_f4 a, wn_f4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wn_f16x8, a), __builtin_bit_cast(wn_f16x8, b), c, 0, 0, 0);
}
static __device__ __forceinline__ unsigned wn_pk_f16(float a, float b) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, wn_f16x2));
}
static __device__ __forceinline__ float wn_f16lo_f32(unsigned u) { return (float)__builtin_bit_cast(wn_f16x2, u).x; }
static __device__ __forceinline__ float wn_f16hi_f32(unsigned u) { return (float)__builtin_bit_cast(wn_f16x2, u).y; }
static __device__ __forceinline__ float wn_bits_f32(unsigned u) { return __builtin_bit_cast(float, u); }
static __device__ __forceinline__ unsigned wn_f32_bits(float f) { return __builtin_bit_cast(unsigned, f); }
// two-lane fp32 vector: fma on it is one v_pk_fma_f32
typedef float f32x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ f32x2 wn_pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
#define WN_UNROLL _Pragma("unroll")
#define WN_NOUNROLL _Pragma("unroll 1")
#define WN_PRAGMA(x) _Pragma(#x)
#define WN_UNROLL_N(n) WN_PRAGMA(unroll n)
#endif

// bf16 fragment rows in LDS are 32 bytes = [16 k], read with ds_read_b128 by lane (li, hi) at row li, k half hi.  With the
// plain placement `row * 32 + half * 16` the 16 lanes the hardware services together ({0-3, 12-15, 20-27}, ... of one
// lane half: MI355X_MICROARCH.md, LDS) land on only 8 of the 16 16-byte slots of a bank row -- every fragment read is a
// 2-way bank conflict.  Swapping the two halves of the rows whose index has bit 3 set makes the 16 slots distinct:
// byte offset of (row, half) inside a [rows][16 k] piece.  Writers and readers use the same function.
// -DWN_NO_FRAG_SWIZZLE restores the plain placement (A/B builds).
static __host__ __device__ __forceinline__ int wn_frag_off(int row, int half) {
#ifdef WN_NO_FRAG_SWIZZLE
    return row * 32 + half * 16;
#else
    return row * 32 + ((half ^ ((row >> 3) & 1)) << 4);
#endif
}

// Row (M index) of accumulator register r for a lane whose upper-half flag is hi (= lane>>5).
static __device__ __forceinline__ int mfma32_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

static __device__ __forceinline__ f32x16 f32x16_zero() {
    f32x16 z;
    WN_UNROLL
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    return z;
}

// Logistic / tanh on the hardware transcendental units: v_exp_f32 (2^x) and v_rcp_f32 are 1-ulp
// instructions, so sigmoid = rcp(1 + 2^(-x log2 e)) and tanh = sign(x) (1 - e)/(1 + e) with
// e = 2^(-2|x| log2 e) carry an ABSOLUTE error of a few 1e-7 -- far inside the 1e-4 parity gate --
// at ~1/4 of the instruction count of ocml expf/tanhf + IEEE division (the gate math is the
// VALU-heavy part of the fused residual-block kernels).
#ifdef WN_EMU
static inline float wn_exp2(float x) { return exp2f(x); }
static inline float wn_rcp(float x) { return 1.0f / x; }
static inline float wn_log2(float x) { return log2f(x); }
#else
static __device__ __forceinline__ float wn_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
static __device__ __forceinline__ float wn_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
static __device__ __forceinline__ float wn_log2(float x) { return __builtin_amdgcn_logf(x); }
#endif
static __device__ __forceinline__ float wn_sigmoid(float x) {
    const float xc = fminf(fmaxf(x, -80.0f), 80.0f);  // keeps 2^(...) finite; sigmoid(+-80) is 1/0 in fp32
    return wn_rcp(1.0f + wn_exp2(-1.4426950408889634f * xc));
}
static __device__ __forceinline__ float wn_tanh(float x) {
    const float ax = fminf(fabsf(x), 40.0f);
    const float e = wn_exp2(-2.8853900817779268f * ax);
    const float t = (1.0f - e) * wn_rcp(1.0f + e);
    return copysignf(t, x);
}

static __device__ __forceinline__ float wave_reduce_sum(float v) {
    WN_UNROLL
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
// Sum over each aligned group of 16 lanes (a DPP "row"); every lane of the group ends with the same bits.
#ifdef WN_EMU
static inline float wn_row16_sum(float v) {
    for (int m = 1; m <= 8; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}
#else
template <int CTRL>
static __device__ __forceinline__ float wn_dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
static __device__ __forceinline__ float wn_row16_sum(float v) {
    v += wn_dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
    v += wn_dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
    v += wn_dpp_f32<0x141>(v);  // row_half_mirror: the other quad of the 8
    v += wn_dpp_f32<0x140>(v);  // row_mirror: the other 8 of the 16
    return v;
}
#endif
static __device__ __forceinline__ float wave_reduce_max(float v) {
    WN_UNROLL
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
