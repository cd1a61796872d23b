// Exhaustive check that the compositor's short square root (csrc/splat.hip: sqrt_rn_unit) equals sqrtf bit for bit on every float
// in [2^-20, 2] (it is applied to values clamped to [1e-3, 1]).   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/sqrt_probe.hip -o tools/sqrt_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ float sqrt_rn_unit(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float em = __builtin_fmaf(-sm, s, x), ep = __builtin_fmaf(-sp, s, x);
    float r = em <= 0.0f ? sm : s;
    r = ep > 0.0f ? sp : r;
    return r;
}
__global__ void k(uint32_t lo, uint32_t n, unsigned long long *bad, uint32_t *first)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = __uint_as_float(lo + i);
    if (__float_as_uint(sqrt_rn_unit(x)) != __float_as_uint(sqrtf(x))) {
        if (atomicAdd(bad, 1ull) == 0) *first = lo + i;
    }
}
int main()
{
    const float a = 9.5367431640625e-07f, b = 2.0f;   // 2^-20 .. 2
    uint32_t lo, hi;
    memcpy(&lo, &a, 4); memcpy(&hi, &b, 4);
    unsigned long long *bad; uint32_t *first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4); hipMemset(bad, 0, 8); hipMemset(first, 0, 4);
    const uint32_t n = hi - lo + 1;
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, lo, n, bad, first);
    unsigned long long hb; uint32_t hf;
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    printf("checked %u floats in [2^-20, 2]: %llu mismatches (first bits 0x%08x)\n", n, hb, hf);
    return hb != 0;
}
