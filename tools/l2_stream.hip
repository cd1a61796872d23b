// Tuning aid: how fast can ONE workgroup (one CU) stream an L2-resident array?  (k_chain1's weight stream)
// build: hipcc --offload-arch=gfx950 -O3 tools/l2_stream.hip -o tools/l2_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NLOAD>
__global__ __launch_bounds__(1024) void k_stream(const float *w, size_t floats_per_chunk, int nchunk, int nthreads, float *out)
{
    const int t = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    if (t < nthreads) {
        for (int c = 0; c < nchunk; ++c) {
            const float *base = w + (size_t)c * floats_per_chunk + (size_t)t * 4;
            f32x4 v[NLOAD];
#pragma unroll
            for (int k = 0; k < NLOAD; ++k) v[k] = *(const f32x4 *)(base + (size_t)k * nthreads * 4);
#pragma unroll
            for (int k = 0; k < NLOAD; ++k) acc += v[k];
        }
    }
    if (acc.x == 12345.0f) out[t] = acc.y;
}

int main()
{
    const size_t maxtotal = 16u << 20;
    float *w, *out;
    hipMalloc(&w, maxtotal);
    const size_t total0 = maxtotal;
    hipMalloc(&out, 4096);
    hipMemset(w, 0, total0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (size_t total : {(size_t)2 << 20, (size_t)5 << 19, (size_t)3 << 20, (size_t)13 << 18, (size_t)7 << 19, (size_t)4 << 20, (size_t)6 << 20, (size_t)12 << 20}) {
        const int nthreads = 832;
        for (int nblocks : {1, 16}) {
            const int NL = 8;
            const size_t chunk_floats = (size_t)NL * nthreads * 4;
            const int nchunk = (int)(total / 4 / chunk_floats);
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_stream<8>, dim3(nblocks), dim3(1024), 0, 0, w, chunk_floats, nchunk, nthreads, out);
            hipEventRecord(e0);
            const int R = 20;
            for (int rep = 0; rep < R; ++rep) hipLaunchKernelGGL(k_stream<8>, dim3(nblocks), dim3(1024), 0, 0, w, chunk_floats, nchunk, nthreads, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / R, bytes = (double)nchunk * chunk_floats * 4;
            printf("MB %.2f threads %4d blocks %2d: %.1f us per launch, %.1f GB/s per CU (%.1f B/clk @2.4GHz)\n", total / 1048576.0, nthreads, nblocks, us, bytes / us / 1e3, bytes / us / 1e3 / 2.4);
        }
    }
    return 0;
}
