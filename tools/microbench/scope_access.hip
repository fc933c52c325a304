// What a device-scope (agent-scope, sc1) access costs on gfx950 when thousands of workgroups make one each -- the question behind
// the look-back of a one-launch Minimize (EXPERIMENTS.md R4.5).  2025 workgroups of 256 threads (one 1080p frame of 1024-slot
// blocks); wave 0 of each does ONE access of the named kind, nothing else; the kernel's duration by HIP events, median of 200.
//   hipcc --offload-arch=gfx950 -O3 -o scope_access scope_access.hip && ./scope_access
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

enum Kind { kNone, kStorePlain, kStoreAgent, kLoadPlain, kLoadAgentOwn, kLoadAgentSameLine, kLoadAgentWave, kAtomicOwn, kAtomicSame, kPublishThenRead1, kPublishThenRead63 };

__global__ __launch_bounds__(256) void k(int kind, uint64_t* a, uint64_t* sink, uint32_t epoch)
{
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    if (lane >= 64) return;
    uint64_t v = 0;
    switch (kind) {
    case kNone: break;
    case kStorePlain: if (lane == 0) a[b * 16] = epoch; break;                                  // one 128-byte line per block
    case kStoreAgent: if (lane == 0) __hip_atomic_store(&a[b * 16], (uint64_t)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
    case kLoadPlain: if (lane == 0) v = a[b * 16]; break;
    case kLoadAgentOwn: if (lane == 0) v = __hip_atomic_load(&a[b * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
    case kLoadAgentSameLine: if (lane == 0) v = __hip_atomic_load(&a[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
    case kLoadAgentWave: v = __hip_atomic_load(&a[(b & ~63u) + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; // 64 entries of the block's group
    case kAtomicOwn: if (lane == 0) v = atomicAdd((unsigned long long*)&a[b * 16], 1ull); break;
    case kAtomicSame: if (lane == 0) v = atomicAdd((unsigned long long*)&a[0], 1ull); break;
    case kPublishThenRead1:
    case kPublishThenRead63: {
        // publish my entry, then wait for the entry of block b-1 (or of every earlier block of my group of 64)
        if (lane == 0) __hip_atomic_store(&a[b], ((uint64_t)epoch << 32) | b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t first = b & ~63u;
        bool mine = kind == kPublishThenRead63 ? first + lane < b : (lane == 0 && b > 0);
        const uint32_t j = kind == kPublishThenRead63 ? first + lane : b - 1;
        if (mine) {
            for (int i = 0; i < (1 << 20); i++) {
                const uint64_t e = __hip_atomic_load(&a[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)(e >> 32) == epoch) { v = e; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        break;
    }
    }
    if (v == 0x123456789abcull) sink[0] = v;
}

int main()
{
    const int nb = 2025;
    uint64_t *a, *sink;
    hipMalloc(&a, (size_t)nb * 16 * 8 + 4096);
    hipMalloc(&sink, 64);
    hipMemset(a, 0, (size_t)nb * 16 * 8 + 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[] = {"nothing", "plain store, own line", "agent-scope store, own line", "plain load, own line", "agent-scope load, own line",
                           "agent-scope load, ONE line for all", "agent-scope load, 64 lanes x 8 B of the group", "atomic add, own line", "atomic add, ONE address",
                           "publish, wait for block b-1", "publish, wait for the group's earlier blocks"};
    uint32_t epoch = 0;
    for (int kind = 0; kind <= kPublishThenRead63; kind++) {
        std::vector<float> t;
        for (int r = 0; r < 220; r++) {
            epoch++;
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, kind, a, sink, epoch);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (r >= 20) t.push_back(ms * 1000.f);
        }
        std::sort(t.begin(), t.end());
        printf("%-50s median %7.2f us   min %7.2f   p90 %7.2f\n", names[kind], t[t.size() / 2], t[0], t[t.size() * 9 / 10]);
    }
    return 0;
}
