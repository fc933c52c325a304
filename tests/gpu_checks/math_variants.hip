// math_variants.hip -- EXPERIMENT (not part of the product or the test suite): counts, over all 2^32
// inputs, how many results of candidate short rcp/sqrt sequences differ from the IEEE expansions.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ bool safe(float x) { float a = fabsf(x); return a >= 0x1.0p-60f && a <= 0x1.0p+60f; }

template <int V> __device__ __forceinline__ float rcp_v(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    if (V >= 1) { float e = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e, r, r); }
    if (V >= 2) { float e = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e, r, r); }
    if (V >= 3) { float e = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e, r, r); }
    return r;
}
template <int V> __device__ __forceinline__ float sqrt_v(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    if (V == 1 || V == 3) { const float r = __builtin_fmaf(-h, g, 0.5f); g = __builtin_fmaf(g, r, g); h = __builtin_fmaf(h, r, h); }
    if (V == 3) { const float r = __builtin_fmaf(-h, g, 0.5f); g = __builtin_fmaf(g, r, g); h = __builtin_fmaf(h, r, h); }
    const float d = __builtin_fmaf(-g, g, x);
    g = __builtin_fmaf(d, h, g);
    if (V == 2) { const float d2 = __builtin_fmaf(-g, g, x); g = __builtin_fmaf(d2, h, g); }
    return g;
}
// sqrt from v_sqrt_f32 (1 ulp) + one residual correction with h = 0.5 * rcp(g)
__device__ __forceinline__ float sqrt_w(float x)
{
    float g = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rcpf(g);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}

template <int KIND, int V>
__global__ void count_bad(unsigned long long* out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        if (!safe(x)) continue;
        float got, want;
        if (KIND == 0) { got = rcp_v<V>(x); want = 1.0f / x; }
        else if (KIND == 1) { if (x < 0) continue; got = sqrt_v<V>(x); want = sqrtf(x); }
        else { if (x < 0) continue; got = sqrt_w(x); want = sqrtf(x); }
        if (__float_as_uint(got) != __float_as_uint(want)) bad++;
    }
    if (bad) atomicAdd(out, bad);
}

template <int KIND, int V> void run(const char* name)
{
    unsigned long long* d; unsigned long long h = 0;
    hipMalloc(&d, 8); hipMemcpy(d, &h, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((count_bad<KIND, V>), dim3(4096), dim3(256), 0, 0, d);
    hipDeviceSynchronize();
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-40s mismatches: %llu\n", name, h);
    hipFree(d);
}

int main()
{
    run<0, 0>("rcp: bare v_rcp");
    run<0, 1>("rcp: + 1 Newton (3 instr)");
    run<0, 2>("rcp: + 2 steps (5 instr)");
    run<0, 3>("rcp: + 3 steps (7 instr)");
    run<1, 0>("sqrt: rsq, residual (5 instr)");
    run<1, 1>("sqrt: rsq, Newton, residual (8 instr)");
    run<1, 2>("sqrt: rsq, residual x2 (7 instr)");
    run<1, 3>("sqrt: rsq, Newton x2, residual (11)");
    run<2, 0>("sqrt: v_sqrt + rcp + residual (5 instr)");
    return 0;
}
