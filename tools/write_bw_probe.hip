// Tuning aid: what does a streaming WRITE reach on this chip, by store flavour and grid?  (The decoder's up-sampling pass and the
// 4 -> 64 projection write 2-4 GB per 128 views at ~2.3-2.5 TB/s; reads of the same size reach 5 TB/s.)
//   hipcc --offload-arch=gfx950 -O3 tools/write_bw_probe.hip -o /tmp/write_bw_probe && /tmp/write_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(256) void k_write(f32x4 *out, size_t n4, const f32x4 *in)
{
    const f32x4 v = {1.0f, 2.0f, 3.0f, (float)blockIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        if (MODE == 0) out[i] = v;
        else if (MODE == 1) __builtin_nontemporal_store(v, &out[i]);
        else if (MODE == 2) out[i] = in[i];                                       // copy: read + write
        else if (MODE == 3) __builtin_nontemporal_store(__builtin_nontemporal_load(&in[i]), &out[i]);
        else if (MODE == 4) { f32x4 t = in[i]; asm volatile("" : "+v"(t)); if (t[0] == 12345.0f) out[i] = t; }   // read only
    }
}
// contiguous chunk per workgroup instead of a grid-stride interleave
template <int MODE> __global__ __launch_bounds__(256) void k_write_chunk(f32x4 *out, size_t n4)
{
    const f32x4 v = {1.0f, 2.0f, 3.0f, (float)blockIdx.x};
    const size_t per = n4 / gridDim.x, base = per * blockIdx.x;
    for (size_t i = threadIdx.x; i < per; i += 256) {
        if (MODE == 0) out[base + i] = v; else __builtin_nontemporal_store(v, &out[base + i]);
    }
}
int main()
{
    const size_t bytes = (size_t)4 << 30, n4 = bytes / 16;
    f32x4 *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch, double moved) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %7.3f ms  %6.2f TB/s\n", name, ms / 5, moved / (ms / 5 * 1e-3) / 1e12);
    };
    for (int grid : {2048, 8192, 65536}) {
        printf("grid %d\n", grid);
        run("  plain store", [&] { hipLaunchKernelGGL(k_write<0>, dim3(grid), dim3(256), 0, 0, a, n4, b); }, bytes);
        run("  nontemporal store", [&] { hipLaunchKernelGGL(k_write<1>, dim3(grid), dim3(256), 0, 0, a, n4, b); }, bytes);
        run("  copy (plain)", [&] { hipLaunchKernelGGL(k_write<2>, dim3(grid), dim3(256), 0, 0, a, n4, b); }, 2.0 * bytes);
        run("  copy (nontemporal both)", [&] { hipLaunchKernelGGL(k_write<3>, dim3(grid), dim3(256), 0, 0, a, n4, b); }, 2.0 * bytes);
        run("  read only", [&] { hipLaunchKernelGGL(k_write<4>, dim3(grid), dim3(256), 0, 0, a, n4, b); }, bytes);
        run("  plain store, chunk per workgroup", [&] { hipLaunchKernelGGL(k_write_chunk<0>, dim3(grid), dim3(256), 0, 0, a, n4); }, bytes);
        run("  nontemporal store, chunk per workgroup", [&] { hipLaunchKernelGGL(k_write_chunk<1>, dim3(grid), dim3(256), 0, 0, a, n4); }, bytes);
    }
    return 0;
}
