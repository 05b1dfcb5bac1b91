// Micro-benchmark (gfx950): a wave moves 64-channel x 32-sample tiles of an activation tensor [B][64][T] through registers
// in the MFMA operand / accumulator layout (lane = sample, registers = channels) with NM MFMAs per tile in between.
//   mode 0: 32 dword loads + 32 dword stores per tile (one channel row per instruction, 128 B per row)
//   mode 1: 8 dwordx4 loads + 8 dwordx4 stores per tile (lane = 4 samples of one channel), 4x4 quad transposes (DPP)
//           between the memory layout and the register layout
// Same bytes, 4x fewer memory instructions.  hipcc --offload-arch=gfx950 -O3 -o tile_io tile_io.hip && ./tile_io
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
static __device__ rsrc_t make_buf(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000); }
// lanes whose position in their quad is in `BANKS` take the value of the quad_perm partner, the others keep `keep`
template <int CTRL, int BANKS> static __device__ __forceinline__ float dpp_take(float keep, float from) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, from), CTRL, 0xf, BANKS, false));
}
// on entry lane j of a quad holds M[j][0..3]; on exit M[0..3][j]   (two butterfly stages, 6 instructions each)
static __device__ __forceinline__ void quad_transpose(float (&v)[4], bool, bool) {
    float a[4];
    a[0] = dpp_take<0xB1, 0xA>(v[0], v[1]);   // lanes 1, 3 of the quad: element 0 comes from the neighbour's element 1
    a[1] = dpp_take<0xB1, 0x5>(v[1], v[0]);
    a[2] = dpp_take<0xB1, 0xA>(v[2], v[3]);
    a[3] = dpp_take<0xB1, 0x5>(v[3], v[2]);
    v[0] = dpp_take<0x4E, 0xC>(a[0], a[2]);   // lanes 2, 3: element 0 comes from the lane two over, its element 2
    v[2] = dpp_take<0x4E, 0x3>(a[2], a[0]);
    v[1] = dpp_take<0x4E, 0xC>(a[1], a[3]);
    v[3] = dpp_take<0x4E, 0x3>(a[3], a[1]);
}

template <int MODE, int NM>
__global__ __launch_bounds__(512) void k(const float* X, float* Y, int B, int T, long long* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, hi = lane >> 5;
    const int tiles_per_b = T / 32, ntiles = B * tiles_per_b;
    const int nw = gridDim.x * 8, w = blockIdx.x * 8 + wave;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)0.0f; b[i] = (__bf16)1.0f; }
    const bool j0 = lane & 1, j1 = lane & 2;
    const int T4 = T * 4;
    const long long t0 = __builtin_readcyclecounter();
    int count = 0;
    for (int tile = w; tile < ntiles; tile += nw, ++count) {
        const int bb = tile / tiles_per_b, tt = (tile - bb * tiles_per_b) * 32;
        const rsrc_t Xr = make_buf(X + (long)bb * 64 * T, 64u * T4), Yr = make_buf(Y + (long)bb * 64 * T, 64u * T4);
        float x[32];
        if (MODE == 0) {
            const int voff = (4 * hi * T + tt + li) * 4;
#pragma unroll
            for (int s = 0; s < 32; ++s)   // register s = 4 g + e: channel 8 g + 4 hi + e
                x[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(Xr, voff, (8 * (s >> 2) + (s & 3)) * T4, 0));
        } else {
            const int voff = ((4 * hi + (li & 3)) * T + tt + (li & ~3)) * 4;   // lane = channel (li & 3) of its group, samples 4 (li >> 2) ..
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(Xr, voff, 8 * g * T4, 0));
                float q[4] = {v.x, v.y, v.z, v.w};
                quad_transpose(q, j0, j1);
                x[4 * g] = q[0]; x[4 * g + 1] = q[1]; x[4 * g + 2] = q[2]; x[4 * g + 3] = q[3];
            }
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 32; ++s) x[s] = x[s] * 2.0f + acc[s & 3][s >> 2];
        if (MODE == 0) {
            const int voff = (4 * hi * T + tt + li) * 4;
#pragma unroll
            for (int s = 0; s < 32; ++s)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x[s]), Yr, voff, (8 * (s >> 2) + (s & 3)) * T4, 0);
        } else {
            const int voff = ((4 * hi + (li & 3)) * T + tt + (li & ~3)) * 4;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                float q[4] = {x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]};
                quad_transpose(q, j0, j1);
                f4 v; v.x = q[0]; v.y = q[1]; v.z = q[2]; v.w = q[3];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), Yr, voff, 8 * g * T4, 0);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) { out[wave * 2] = t1 - t0; out[wave * 2 + 1] = count; }
}

template <int MODE, int NM>
static float run(const char* name, const float* X, float* Y, int B, int T, long long* out, bool check) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NM>), dim3(240), dim3(512), 0, 0, X, Y, B, T, out);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<MODE, NM>), dim3(240), dim3(512), 0, 0, X, Y, B, T, out);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[16]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0;
    if (check) {
        const size_t n = (size_t)B * 64 * T;
        float* hy = (float*)malloc(n * 4); (void)hipMemcpy(hy, Y, n * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < n; i += 997) sum += hy[i];
        free(hy);
    }
    printf("%-40s %7.1f us per pass, %5.2f TB/s; wave 0: %6.0f cycles per tile (%lld tiles)  checksum %.3f\n", name, ms * 100.0,
           2.0 * B * 64 * (double)T * 4 / (ms / 10 * 1e-3) / 1e12, (double)h[0] / (double)h[1], h[1], sum);
    return ms;
}

int main() {
    const int B = 8, T = 20000;
    const size_t n = (size_t)B * 64 * T;
    float *X, *Y; long long* out;
    (void)hipMalloc(&X, n * 4); (void)hipMalloc(&Y, n * 4); (void)hipMalloc(&out, 256);
    float* hx = (float*)malloc(n * 4);
    for (size_t i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) % 1000) * 0.001f;
    (void)hipMemcpy(X, hx, n * 4, hipMemcpyHostToDevice);
    run<0, 0>("dword,   no MFMA", X, Y, B, T, out, true);
    run<1, 0>("dwordx4 + quad transposes, no MFMA", X, Y, B, T, out, true);
    run<0, 96>("dword,   96 MFMAs per tile", X, Y, B, T, out, false);
    run<1, 96>("dwordx4 + quad transposes, 96 MFMAs", X, Y, B, T, out, false);
    run<0, 240>("dword,   240 MFMAs per tile", X, Y, B, T, out, false);
    run<1, 240>("dwordx4 + quad transposes, 240 MFMAs", X, Y, B, T, out, false);
    return 0;
}
