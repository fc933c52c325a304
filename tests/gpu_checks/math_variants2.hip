// EXPERIMENT: can 1.0f/sqrtf(x) (two roundings) be had from ONE transcendental (v_rsq_f32)?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int V> __device__ __forceinline__ float cand(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    const float h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    g = __builtin_fmaf(d, h, g);              // = sqrtf(x) exactly (verified separately)
    float r = y;                               // ~ 1/g
    { const float e = __builtin_fmaf(-g, r, 1.0f); r = __builtin_fmaf(e, r, r); }
    if (V >= 2) { const float e = __builtin_fmaf(-g, r, 1.0f); r = __builtin_fmaf(e, r, r); }
    if (V >= 3) { const float e = __builtin_fmaf(-g, r, 1.0f); r = __builtin_fmaf(e, r, r); }
    return r;
}
template <int V> __global__ void count_bad(unsigned long long* out, unsigned* first)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 31); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        if (!(x >= 0x1.0p-60f && x <= 0x1.0p+60f)) continue;
        const float got = cand<V>(x), want = 1.0f / sqrtf(x);
        if (__float_as_uint(got) != __float_as_uint(want)) { bad++; atomicMin(first, (unsigned)i); }
    }
    if (bad) atomicAdd(out, bad);
}
template <int V> void run(const char* name)
{
    unsigned long long* d; unsigned long long h = 0; unsigned* f; unsigned hf = 0xffffffffu;
    hipMalloc(&d, 8); hipMemcpy(d, &h, 8, hipMemcpyHostToDevice); hipMalloc(&f, 4); hipMemcpy(f, &hf, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((count_bad<V>), dim3(4096), dim3(256), 0, 0, d, f);
    hipDeviceSynchronize();
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, f, 4, hipMemcpyDeviceToHost);
    printf("%-44s mismatches: %llu (first 0x%08x)\n", name, h, hf);
}
int main()
{
    run<1>("rsq -> sqrt, 1/g from rsq + 1 Newton");
    run<2>("rsq -> sqrt, 1/g from rsq + 2 Newton");
    run<3>("rsq -> sqrt, 1/g from rsq + 3 Newton");
    return 0;
}
