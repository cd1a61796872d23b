// Are the activation caches' rows of different frames at the same location a channel hot spot?  42 workgroups x 4 waves store
// 640-byte rows (16 bytes per lane) to  base + f * frame_stride + q * 640  for 16 frames per workgroup at nearly the same q -- the
// pattern of a chain tile's stores -- with the frame stride the engine has (1024 rows) and padded ones.
//   hipcc --offload-arch=gfx950 -O3 -o store_stride_probe.bin store_stride_probe.hip && ./store_stride_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(float *base, size_t frame_rows, int iters, unsigned long long *out, int mode)
{
    const int L4 = threadIdx.x;                       // 256 lanes: 16 columns x 40 chunks = 640 chunks -> 3 instructions
    const f32x4 v = {1.0f, 2.0f, 3.0f, (float)L4};
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const int q0 = 300 + it;                      // the frontier moves on
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int idx = r * 256 + L4;
            if (idx < 640) {
                const int j = idx / 40, c = idx - 40 * j;
                const size_t f = (size_t)((blockIdx.x * 3 + j * 8) % 128);   // frames spread
                const size_t row = f * frame_rows + q0 + (j & 3);
                float *p = base + row * 160 + 4 * c;
                if (mode == 0) *(f32x4 *)p = v;
                else asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

int main()
{
    const size_t rows_max = 140 * 1100;
    float *base; (void)hipMalloc(&base, rows_max * 160 * 4);
    unsigned long long *out; (void)hipMalloc(&out, 4096);
    for (int mode = 0; mode < 2; ++mode)
        for (size_t fr : {(size_t)1024, (size_t)1025, (size_t)1027, (size_t)1040}) {
            const int iters = 200;
            hipLaunchKernelGGL(k, dim3(42), dim3(256), 0, 0, base, fr, 10, out, mode);
            hipLaunchKernelGGL(k, dim3(42), dim3(256), 0, 0, base, fr, iters, out, mode);
            (void)hipDeviceSynchronize();
            unsigned long long h[42];
            (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 42; ++i) s += h[i];
            printf("%s stores, frame stride %zu rows: %.0f cycles per 10 KB of rows per workgroup (%.1f B/clk per CU)\n", mode ? "write-through" : "plain", fr,
                   s / 42 / iters, 10240.0 / (s / 42 / iters));
        }
    return 0;
}
