// stream_probe.hip -- how fast can ONE workgroup (one CU) stream a read-only buffer?
// Sizes the decode kernel's weight stream (tools/decode_bench.py): per-CU read rate vs footprint
// (L2-resident or not), threads per workgroup and loads in flight per wave.
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe tools/stream_probe.hip && ./stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef int rsrc_t __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ void k_stream(const float* buf, long n_f4, int iters, float* out) {
    // every thread keeps DEPTH 16-byte loads in flight; the workgroup walks the buffer cyclically
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(buf), (short)0, (int)(n_f4 * 16), 0x00020000);
    const int nt = blockDim.x;
    const long stride = (long)nt * 16;
    const long wrap = n_f4 * 16;
    float4 ring[DEPTH];
    long pos = ((long)blockIdx.x * 7919 * stride) % wrap;
    for (int j = 0; j < DEPTH; ++j) {
        ring[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, (int)pos, 0));
        pos += stride; if (pos >= wrap) pos = 0;
    }
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            acc += ring[j].x + ring[j].y + ring[j].z + ring[j].w;
            ring[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, (int)pos, 0));
            pos += stride; if (pos >= wrap) pos = 0;
        }
    }
    for (int j = 0; j < DEPTH; ++j) acc += ring[j].x;
    out[blockIdx.x * nt + threadIdx.x] = acc;
}

template <int DEPTH>
static void run(const float* buf, long bytes, int threads, int blocks, float* out) {
    const long n_f4 = bytes / 16;
    const long per_iter = (long)threads * 16 * DEPTH;
    const int iters = (int)((64L << 20) / per_iter);  // 64 MB per workgroup
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_stream<DEPTH><<<blocks, threads>>>(buf, n_f4, iters / 8, out);
    hipEventRecord(e0);
    k_stream<DEPTH><<<blocks, threads>>>(buf, n_f4, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double gb = (double)per_iter * iters / 1e9;
    printf("footprint %7.2f MB  threads %4d  depth %2d  blocks %3d : %7.1f GB/s per workgroup (%8.1f GB/s total)\n",
           bytes / 1048576.0, threads, DEPTH, blocks, gb / (ms * 1e-3), gb * blocks / (ms * 1e-3));
}

int main() {
    float *buf, *out;
    const long maxb = 256L << 20;
    hipMalloc(&buf, maxb);
    hipMemset(buf, 0, maxb);
    hipMalloc(&out, 256 * 1024 * 4);
    const long foot[] = {1L << 20, 3L << 20, 5L << 20, 64L << 20};
    for (long f : foot) {
        run<18>(buf, f, 512, 1, out);
        run<18>(buf, f, 1024, 1, out);
        run<8>(buf, f, 512, 1, out);
        run<32>(buf, f, 512, 1, out);
        run<18>(buf, f, 256, 1, out);
    }
    run<18>(buf, 5L << 20, 512, 8, out);
    run<18>(buf, 5L << 20, 512, 64, out);
    run<18>(buf, 5L << 20, 512, 256, out);
    return 0;
}
