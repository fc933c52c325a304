// store_flush.hip -- experiment (not product): what does the end-of-kernel write-back of dirty L2 lines cost a
// store-heavy kernel, and which store flavour avoids it?
//
// The trace kernel writes 41.5 MB of records per 1080p frame into a 32 MiB write-back L2 (8 x 4 MiB).  At the end
// of a dispatch the dirty lines have to reach HBM before the next dependent dispatch starts, and during that
// write-back nothing computes.  This program times kernels that only store N bytes, as plain stores, as `nt`
// stores and as `sc1` (write-through) stores, in two shapes: 16 bytes per lane contiguous (full 128-byte lines
// per wave instruction) and the trace kernel's shape (five dword stores per lane at a 20-byte lane stride).
// Back-to-back launches on one stream, HIP events around 200 of them.
//   hipcc --offload-arch=gfx950 -O3 -o store_flush store_flush.hip && ./store_flush
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

enum { PLAIN = 0, NT = 1, SC1 = 2, SC0SC1 = 3 };

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int F>
__device__ __forceinline__ void store16(uint4* p, uint4 v4)
{
    u32x4 v = {v4.x, v4.y, v4.z, v4.w};
    if (F == PLAIN) *p = v4;
    else if (F == NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
    else if (F == SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

template <int F>
__device__ __forceinline__ void store4(uint32_t* p, uint32_t v)
{
    if (F == PLAIN) *p = v;
    else if (F == NT) __builtin_nontemporal_store(v, p);
    else if (F == SC1) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// work: dummy VALU iterations per thread before the stores (0 = pure store kernel)
template <int F>
__global__ __launch_bounds__(256) void wide(uint4* out, size_t n16, int work)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float a = (float)i;
    for (int k = 0; k < work; k++) a = a * 1.0001f + 0.5f;
    if (i < n16) store16<F>(out + i, make_uint4((uint32_t)i, __float_as_uint(a), 2u, 3u));
}

// the trace kernel's shape: each lane owns a 20-byte record, 5 dword stores at a 20-byte lane stride
template <int F>
__global__ __launch_bounds__(256) void rec20(uint32_t* out, size_t nrec, int work)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float a = (float)i;
    for (int k = 0; k < work; k++) a = a * 1.0001f + 0.5f;
    if (i < nrec) {
        uint32_t* p = out + i * 5;
#pragma unroll
        for (int k = 0; k < 5; k++) store4<F>(p + k, (uint32_t)i + k + __float_as_uint(a));
    }
}

// records transposed through LDS into 16-byte stores (what rtx_expand does): wave-local, 80 x 16 B per wave
template <int F>
__global__ __launch_bounds__(256) void rec20_lds(uint32_t* out, size_t nrec, int work)
{
    __shared__ __attribute__((aligned(16))) uint32_t s[256 * 5];
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float a = (float)i;
    for (int k = 0; k < work; k++) a = a * 1.0001f + 0.5f;
    uint32_t* sw = s + wave * 320;
#pragma unroll
    for (int k = 0; k < 5; k++) sw[lane * 5 + k] = (uint32_t)i + k + __float_as_uint(a);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const size_t w0 = (size_t)blockIdx.x * 256 + wave * 64;
    if (w0 + 64 <= nrec) {
        uint4* d4 = reinterpret_cast<uint4*>(out + w0 * 5);
        const uint4* s4 = reinterpret_cast<const uint4*>(sw);
        store16<F>(d4 + lane, s4[lane]);
        if (lane < 16) store16<F>(d4 + 64 + lane, s4[64 + lane]);
    }
}

template <typename L>
float time_launches(L launch, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; i++) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms * 1e3f / reps;
}

int main()
{
    const size_t cap = 256u << 20;
    void* buf = nullptr;
    CK(hipMalloc(&buf, cap));
    CK(hipMemset(buf, 0, cap));
    // clock run-in
    for (int i = 0; i < 3000; i++) hipLaunchKernelGGL(wide<PLAIN>, dim3(8192), dim3(256), 0, 0, (uint4*)buf, (size_t)8192 * 256, 64);
    CK(hipDeviceSynchronize());
    const double sizes_mb[] = {1.0, 4.0, 8.3, 16.0, 32.0, 41.5, 64.0, 128.0};
    const int works[] = {0, 400};
    const char* fl[] = {"plain", "nt", "sc1", "sc0sc1"};
    for (int work : works) {
        printf("# dummy VALU iterations per thread before the stores: %d  (us per launch, back-to-back on one stream)\n", work);
        printf("%8s | %-28s | %-28s | %-28s\n", "MB", "16B/lane  plain nt sc1 sc0sc1", "5 dwords @20B  (same order)", "20B via LDS -> 16B stores");
        for (double mb : sizes_mb) {
            const size_t bytes = (size_t)(mb * 1e6);
            const size_t n16 = bytes / 16, nrec = bytes / 20;
            const unsigned b16 = (unsigned)((n16 + 255) / 256), brec = (unsigned)((nrec + 255) / 256);
            float t[3][4];
#define RUN(F) \
    t[0][F] = time_launches([&] { hipLaunchKernelGGL(wide<F>, dim3(b16), dim3(256), 0, 0, (uint4*)buf, n16, work); }, 200); \
    t[1][F] = time_launches([&] { hipLaunchKernelGGL(rec20<F>, dim3(brec), dim3(256), 0, 0, (uint32_t*)buf, nrec, work); }, 200); \
    t[2][F] = time_launches([&] { hipLaunchKernelGGL(rec20_lds<F>, dim3(brec), dim3(256), 0, 0, (uint32_t*)buf, nrec, work); }, 200);
            RUN(PLAIN) RUN(NT) RUN(SC1) RUN(SC0SC1)
#undef RUN
            printf("%8.1f | %6.2f %6.2f %6.2f %6.2f  | %6.2f %6.2f %6.2f %6.2f  | %6.2f %6.2f %6.2f %6.2f\n", mb, t[0][0], t[0][1], t[0][2], t[0][3],
                   t[1][0], t[1][1], t[1][2], t[1][3], t[2][0], t[2][1], t[2][2], t[2][3]);
            (void)fl;
        }
    }
    // an empty kernel, for the launch floor
    float t0 = time_launches([&] { hipLaunchKernelGGL(wide<PLAIN>, dim3(1), dim3(256), 0, 0, (uint4*)buf, (size_t)0, 0); }, 200);
    printf("empty launch: %.2f us\n", t0);
    hipFree(buf);
    return 0;
}
