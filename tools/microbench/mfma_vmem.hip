// Micro-benchmark (gfx950): what does a global-memory instruction cost a wave that is streaming v_mfma_f32_32x32x16_bf16?
//   per MFMA slot: NL loads (dword or dwordx4 per lane, L2-resident 64 KB window) and/or NS dword stores, one wave per SIMD
//   hipcc --offload-arch=gfx950 -O3 -o mfma_vmem mfma_vmem.hip && ./mfma_vmem
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f4;

typedef __amdgpu_buffer_rsrc_t rsrc_t;
static __device__ rsrc_t make_buf(void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(p, (short)0, (int)bytes, 0x00020000); }

template <int MFMA, int NL, int WIDE, int NS>
__global__ __launch_bounds__(256) void k(long long* out, float* buf, int iters) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
    float* base = buf + (size_t)blockIdx.x * 65536 + wave * 16384;   // 64 KB per wave, 256 KB per block
    const rsrc_t R = make_buf(base, 65536);
    float s = 0.f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        float v[NL > 0 ? 16 * NL * (WIDE ? 4 : 1) : 1];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (MFMA) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                const int row = (m * NL + q) & 7;
                if (WIDE) {
                    const f4 x = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(R, lane * 16, row * 4096, 0));
                    v[(m * NL + q) * 4] = x.x; v[(m * NL + q) * 4 + 1] = x.y; v[(m * NL + q) * 4 + 2] = x.z; v[(m * NL + q) * 4 + 3] = x.w;
                } else {
                    v[m * NL + q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(R, lane * 4, row * 4096, 0));
                }
            }
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const int row = (m * NS + q) & 7;
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)it, R, lane * 4, 32768 + row * 256, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (NL > 0) {
#pragma unroll
            for (int i = 0; i < 16 * NL * (WIDE ? 4 : 1); ++i) s += v[i];
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) buf[0] = s;
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MFMA, int NL, int WIDE, int NS>
static void run(const char* name) {
    long long* out; float* buf;
    hipMalloc(&out, 256 * 4 * 8); hipMalloc(&buf, (size_t)256 * 65536 * 4);
    hipMemset(buf, 0, (size_t)256 * 65536 * 4);
    const int iters = 500;
    hipLaunchKernelGGL((k<MFMA, NL, WIDE, NS>), dim3(256), dim3(256), 0, 0, out, buf, iters);
    hipLaunchKernelGGL((k<MFMA, NL, WIDE, NS>), dim3(256), dim3(256), 0, 0, out, buf, iters);
    hipDeviceSynchronize();
    long long h[4];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-64s %7.1f cycles per slot\n", name, (double)h[0] / iters / 16);
    hipFree(out); hipFree(buf);
}

int main() {
    run<1, 0, 0, 0>("MFMA only");
    run<1, 1, 0, 0>("MFMA + 1 dword load");
    run<1, 2, 0, 0>("MFMA + 2 dword loads");
    run<1, 4, 0, 0>("MFMA + 4 dword loads");
    run<1, 1, 1, 0>("MFMA + 1 dwordx4 load");

    run<1, 0, 0, 1>("MFMA + 1 dword store");
    run<1, 0, 0, 2>("MFMA + 2 dword stores");
    run<1, 1, 0, 1>("MFMA + 1 dword load + 1 dword store");
    run<0, 1, 0, 0>("no MFMA, 1 dword load per slot");
    run<0, 4, 0, 0>("no MFMA, 4 dword loads per slot");
    run<0, 1, 1, 0>("no MFMA, 1 dwordx4 load per slot");

    run<0, 0, 0, 1>("no MFMA, 1 dword store per slot");
    run<0, 0, 0, 4>("no MFMA, 4 dword stores per slot");
    return 0;
}
