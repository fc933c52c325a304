// rtx_post.hip -- what RayTracingManager::Update runs around the trace kernel, on the GPU:
// UpdateObjects (RayTracingManager.cu:10-44, 89-107) and Minimize (RayTracingManager.cu:167-319),
// plus rtx_update, the whole of Update in one call.
#include "rtx_ctx.h"
#include "rtx_group.h"
#include "rtx_device.hpp"
#include "rtx_records.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "rtx_post_kernels.inc" // namespace rtx: the kernels launched below

namespace {

int ensure_min_buffers(rtx_ctx* ctx, size_t n_blocks, bool need_out)
{
    const size_t need = n_blocks * (sizeof(uint32_t) + sizeof(uint64_t)) + 64;
    if (ctx->scan_bytes < need) {
        if (ctx->d_scan) {
            hipFree(ctx->d_scan);
            ctx->d_scan = nullptr;
            ctx->scan_bytes = 0;
        }
        if (hipMalloc(&ctx->d_scan, need) != hipSuccess) {
            return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, "hipMalloc failed for the minimise scratch");
        }
        ctx->scan_bytes = need;

    }
    if (need_out && !ctx->d_min) {
        // m_minimizedResultArray is as large as the frame (RayTracingManager.cu:66)
        if (hipMalloc((void**)&ctx->d_min, ctx->capacity) != hipSuccess) {
            return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, "hipMalloc failed for the minimise buffer");
        }
    }
    return RTX_OK;
}

int launch_minimize_chain(rtx_ctx* ctx, void* d_scan, int mode, size_t w, size_t h, const uint8_t* d_in, uint8_t* d_out, uint64_t** d_total)
{
    const uint64_t n_slots = (uint64_t)w * h;
    const size_t n_blocks = (size_t)((n_slots + rtx::kSlotsPerBlock - 1) / rtx::kSlotsPerBlock);
    // scratch layout: [total u64][offsets u64 x n_blocks][sums u32 x n_blocks]
    uint64_t* total = (uint64_t*)d_scan;
    uint64_t* offsets = total + 8;
    uint32_t* sums = (uint32_t*)(offsets + n_blocks);
    const bool rgb = !(mode == RTX_BIT_ASCII || mode == RTX_BIT_PIXEL); // MinimizeResults, RayTracingManager.cu:167-179
    hipStream_t st = ctx->stream;
    if (rgb) {
        hipLaunchKernelGGL((rtx::rtx_min_count<20>), dim3((unsigned)n_blocks), dim3(rtx::kThreads), 0, st, d_in, n_slots, (uint32_t)w, sums);
    } else {
        hipLaunchKernelGGL((rtx::rtx_min_count<12>), dim3((unsigned)n_blocks), dim3(rtx::kThreads), 0, st, d_in, n_slots, (uint32_t)w, sums);
    }
    if (rgb) {
        hipLaunchKernelGGL((rtx::rtx_min_scatter<20>), dim3((unsigned)n_blocks), dim3(rtx::kThreads), 0, st, d_in, n_slots, (uint32_t)w, sums, d_out, total);
    } else {
        hipLaunchKernelGGL((rtx::rtx_min_scatter<12>), dim3((unsigned)n_blocks), dim3(rtx::kThreads), 0, st, d_in, n_slots, (uint32_t)w, sums, d_out, total);
    }
    RTX_HIP(ctx, hipGetLastError());
    *d_total = total;
    return RTX_OK;
}

size_t words_scan_bytes(uint64_t n_slots)
{
    const size_t n_blocks = (size_t)((n_slots + rtx::kWSlotsPerBlock - 1) / rtx::kWSlotsPerBlock);
    return n_blocks * (sizeof(uint32_t) + sizeof(uint64_t)) + 64;
}

// Minimize from W*H pixel words (every mode but SDL) on the context's stream as three launches; scratch laid out as launch_minimize's.
int launch_minimize_words_chain(rtx_ctx* ctx, void* d_scan, int mode, size_t w, size_t h, const uint32_t* d_words, uint8_t* d_out, uint64_t** d_total, uint32_t lead)
{
    const uint64_t n_slots = (uint64_t)w * h;
    const unsigned n_blocks = (unsigned)((n_slots + rtx::kWSlotsPerBlock - 1) / rtx::kWSlotsPerBlock);
    uint64_t* total = (uint64_t*)d_scan;
    uint64_t* offsets = total + 8;
    uint32_t* sums = (uint32_t*)(offsets + n_blocks);
    hipStream_t st = ctx->stream;
#define RTX_MINW(M)                                                                                                                        \
    do {                                                                                                                                   \
        hipLaunchKernelGGL((rtx::rtx_minw_count<M>), dim3(n_blocks), dim3(rtx::kThreads), 0, st, d_words, n_slots, (uint32_t)w, sums, lead);      \
        hipLaunchKernelGGL(rtx::rtx_min_offsets, dim3(1), dim3(rtx::kThreads), 0, st, sums, n_blocks, offsets, total);                      \
        hipLaunchKernelGGL((rtx::rtx_minw_scatter<M>), dim3(n_blocks), dim3(rtx::kThreads), 0, st, d_words, n_slots, (uint32_t)w, offsets, d_out, lead); \
    } while (0)
    switch (mode) {
    case RTX_BIT_ASCII: RTX_MINW(RTX_K_BIT_ASCII); break;
    case RTX_BIT_PIXEL: RTX_MINW(RTX_K_BIT_PIXEL); break;
    case RTX_RGB_ASCII: RTX_MINW(RTX_K_RGB_ASCII); break;
    case RTX_RGB_PIXEL: RTX_MINW(RTX_K_RGB_PIXEL); break;
    case RTX_RGB_NORMALS: RTX_MINW(RTX_K_RGB_NORMALS); break;
    default: return rtx_fail(ctx, RTX_ERR_INVALID_MODE, "no pixel words in this mode");
    }
#undef RTX_MINW
    RTX_HIP(ctx, hipGetLastError());
    *d_total = total;
    return RTX_OK;
}

// The look-back tables of rtx_minw_fused: agg (one entry per block) then 64 replicas of grp (one entry per 64 blocks); zeroed when allocated, tagged by
// epoch afterwards.  One set per context: every minimise launch runs on the context's stream, one after the other.
int ensure_look_tables(rtx_ctx* ctx, size_t n_blocks)
{
    if (ctx->look_blocks >= n_blocks && ctx->d_look) return RTX_OK;
    if (ctx->d_look) {
        RTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        hipFree(ctx->d_look);
        ctx->d_look = nullptr;
        ctx->look_blocks = 0;
    }
    size_t cap = 4096;
    while (cap < n_blocks) cap *= 2;
    const size_t bytes = 2 * cap * sizeof(uint64_t); // agg[cap], then 64 replicas of grp[cap / 64]
    if (hipMalloc((void**)&ctx->d_look, bytes) != hipSuccess) {
        ctx->d_look = nullptr;
        return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, "hipMalloc failed for the minimise look-back tables");
    }
    RTX_HIP(ctx, hipMemsetAsync(ctx->d_look, 0, bytes, ctx->stream));
    ctx->look_blocks = cap;
    return RTX_OK;
}

// Minimize from the records of a W*H frame on the context's stream: one launch (rtx_min_fused, RTX_OPT_MINIMIZE_FUSED) or the two of
// launch_minimize_chain.  As for the word form: (*d_total)[0] will hold the stream's length, and after a fused launch
// (ctx->min_fused_epoch != 0) (*d_total)[1] == that epoch says that blocks gave up -- settle_minimize redoes the frame.
int launch_minimize(rtx_ctx* ctx, void* d_scan, int mode, size_t w, size_t h, const uint8_t* d_in, uint8_t* d_out, uint64_t** d_total)
{
    const uint64_t n_slots = (uint64_t)w * h;
    const uint64_t n_blocks = (n_slots + rtx::kRSlotsPerBlock - 1) / rtx::kRSlotsPerBlock;
    ctx->min_fused_epoch = 0;
    if (ctx->opt_min_fused == 0 || n_blocks > (1u << 24)) return launch_minimize_chain(ctx, d_scan, mode, w, h, d_in, d_out, d_total);
    int rc = ensure_look_tables(ctx, (size_t)n_blocks);
    if (rc != RTX_OK) return rc;
    if (++ctx->look_epoch == 0u) {
        RTX_HIP(ctx, hipMemsetAsync(ctx->d_look, 0, 2 * ctx->look_blocks * sizeof(uint64_t), ctx->stream));
        ctx->look_epoch = 1u;
    }
    const uint32_t epoch = ctx->look_epoch;
    uint64_t* total = (uint64_t*)d_scan;
    uint64_t* agg = ctx->d_look;
    uint64_t* grp = agg + ctx->look_blocks;
    const uint32_t ng = (uint32_t)(ctx->look_blocks / rtx::kLookGroup);
    const uint32_t polls = ctx->opt_min_fused == 2 ? 0u : rtx::kLookPolls;
    const bool rgb = !(mode == RTX_BIT_ASCII || mode == RTX_BIT_PIXEL); // MinimizeResults, RayTracingManager.cu:167-179
    if (rgb) {
        hipLaunchKernelGGL((rtx::rtx_min_fused<20>), dim3((unsigned)n_blocks), dim3(rtx::kThreads), 0, ctx->stream, d_in, n_slots, (uint32_t)w, agg, grp, ng, epoch, polls, d_out,
                           total);
    } else {
        hipLaunchKernelGGL((rtx::rtx_min_fused<12>), dim3((unsigned)n_blocks), dim3(rtx::kThreads), 0, ctx->stream, d_in, n_slots, (uint32_t)w, agg, grp, ng, epoch, polls, d_out,
                           total);
    }
    RTX_HIP(ctx, hipGetLastError());
    ctx->min_fused_epoch = epoch;
    *d_total = total;
    return RTX_OK;
}

int settle_minimize(rtx_ctx* ctx, void* d_scan, int mode, size_t w, size_t h, const uint8_t* d_in, uint8_t* d_out, const uint64_t got[2], uint64_t* total)
{
    *total = got[0];
    const uint32_t epoch = ctx->min_fused_epoch;
    ctx->min_fused_epoch = 0;
    if (epoch == 0u || got[1] != (uint64_t)epoch) return RTX_OK;
    ctx->stat_min_fallbacks++;
    uint64_t* d_total = nullptr;
    int rc = launch_minimize_chain(ctx, d_scan, mode, w, h, d_in, d_out, &d_total);
    if (rc != RTX_OK) return rc;
    RTX_HIP(ctx, hipMemcpyAsync(total, d_total, sizeof *total, hipMemcpyDeviceToHost, ctx->stream));
    RTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RTX_OK;
}

// Minimize from W*H pixel words on the context's stream: one launch (rtx_minw_fused, RTX_OPT_MINIMIZE_FUSED) or the three above.
// (*d_total)[0] will hold the stream's length; after a fused launch (ctx->min_fused_epoch != 0) (*d_total)[1] == that epoch says
// that blocks gave up: settle_minimize_words then redoes the frame with the three launches.
// (`pair`: where the one-launch form leaves the two result words instead of at d_scan -- pinned host memory as the device addresses it,
// for a caller that reads them after a synchronisation without a copy; the three-launch form does not take it)
int launch_minimize_words(rtx_ctx* ctx, void* d_scan, int mode, size_t w, size_t h, const uint32_t* d_words, uint8_t* d_out, uint64_t** d_total, uint32_t lead = 0,
                          uint64_t* pair = nullptr)
{
    const uint64_t n_slots = (uint64_t)w * h;
    const uint64_t n_blocks = (n_slots + rtx::kWSlotsPerBlock - 1) / rtx::kWSlotsPerBlock;
    ctx->min_fused_epoch = 0;
    if (ctx->opt_min_fused == 0 || n_blocks > (1u << 24)) return launch_minimize_words_chain(ctx, d_scan, mode, w, h, d_words, d_out, d_total, lead);
    int rc = ensure_look_tables(ctx, (size_t)n_blocks);
    if (rc != RTX_OK) return rc;
    if (++ctx->look_epoch == 0u) {
        // the tags have wrapped: forget every entry
        RTX_HIP(ctx, hipMemsetAsync(ctx->d_look, 0, 2 * ctx->look_blocks * sizeof(uint64_t), ctx->stream));
        ctx->look_epoch = 1u;
    }
    const uint32_t epoch = ctx->look_epoch;
    uint64_t* total = pair ? pair : (uint64_t*)d_scan;
    uint64_t* agg = ctx->d_look;
    uint64_t* grp = agg + ctx->look_blocks;
    const uint32_t ng = (uint32_t)(ctx->look_blocks / rtx::kLookGroup);
    const uint32_t polls = ctx->opt_min_fused == 2 ? 0u : rtx::kLookPolls;
    hipStream_t st = ctx->stream;
#define RTX_MINF(M) \
    hipLaunchKernelGGL((rtx::rtx_minw_fused<M>), dim3((unsigned)n_blocks), dim3(rtx::kThreads), 0, st, d_words, n_slots, (uint32_t)w, agg, grp, ng, epoch, polls, d_out, total, lead)
    switch (mode) {
    case RTX_BIT_ASCII: RTX_MINF(RTX_K_BIT_ASCII); break;
    case RTX_BIT_PIXEL: RTX_MINF(RTX_K_BIT_PIXEL); break;
    case RTX_RGB_ASCII: RTX_MINF(RTX_K_RGB_ASCII); break;
    case RTX_RGB_PIXEL: RTX_MINF(RTX_K_RGB_PIXEL); break;
    case RTX_RGB_NORMALS: RTX_MINF(RTX_K_RGB_NORMALS); break;
    default: return rtx_fail(ctx, RTX_ERR_INVALID_MODE, "no pixel words in this mode");
    }
#undef RTX_MINF
    RTX_HIP(ctx, hipGetLastError());
    ctx->min_fused_epoch = epoch;
    *d_total = total;
    return RTX_OK;
}

// got[0], got[1]: the two words at *d_total as the host read them after the launches of launch_minimize_words.  A fused launch
// whose blocks gave up is redone here as three launches (the stream is synchronised again); *total = the stream's length.
int settle_minimize_words(rtx_ctx* ctx, void* d_scan, int mode, size_t w, size_t h, const uint32_t* d_words, uint8_t* d_out, const uint64_t got[2], uint64_t* total,
                          uint32_t lead = 0)
{
    *total = got[0];
    const uint32_t epoch = ctx->min_fused_epoch;
    ctx->min_fused_epoch = 0;
    if (epoch == 0u || got[1] != (uint64_t)epoch) return RTX_OK;
    ctx->stat_min_fallbacks++;
    uint64_t* d_total = nullptr;
    int rc = launch_minimize_words_chain(ctx, d_scan, mode, w, h, d_words, d_out, &d_total, lead);
    if (rc != RTX_OK) return rc;
    RTX_HIP(ctx, hipMemcpyAsync(total, d_total, sizeof *total, hipMemcpyDeviceToHost, ctx->stream));
    RTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RTX_OK;
}

// Update traces pixel words and minimises from them (the records of the frame are never written) unless the caller asked for
// the record form (RTX_OPT_UPDATE_WORDS = 0) or the mode has no words (RTX_SDL).
bool update_from_words(const rtx_ctx* ctx, int mode) { return mode != RTX_SDL && ctx->opt_update_words != 0; }

int ensure_words_buffer(rtx_ctx* ctx, uint32_t** buf, size_t* cap, size_t need)
{
    if (*cap >= need) return RTX_OK;
    if (*buf) {
        RTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        hipFree(*buf);
    }
    *buf = nullptr;
    *cap = 0;
    if (hipMalloc((void**)buf, need * sizeof(uint32_t)) != hipSuccess) return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, "hipMalloc failed for the pixel-word buffer");
    *cap = need;
    return RTX_OK;
}

// The frame of `p` as W*H pixel words in *d_words, complete in stream order on the context's stream: sharded over the
// group's devices and gathered, or one launch into `own` (grown as needed).
int trace_words(rtx_ctx* ctx, const rtx_params* p, int mode, uint32_t** own, size_t* own_cap, const uint32_t** d_words)
{
    if (ctx->group) return rtxgroup::render_words(ctx, p, mode, d_words);
    int rc = ensure_words_buffer(ctx, own, own_cap, (size_t)p->x * (size_t)p->y);
    if (rc != RTX_OK) return rc;
    if ((rc = rtx_render_rows(ctx, p, mode, 0, (size_t)p->y, *own, 0, ctx->stream, RTX_RENDER_COMPACT)) != RTX_OK) return rc;
    *d_words = *own;
    return RTX_OK;
}

// The blocking Update with ONE host wait: the Minimize launch stores the stream straight into the caller's buffer (pinned
// host memory, addressed by the device) and its length into a pinned word, so that no copy is queued and nothing is waited for twice.
// A console-sized frame's Update is three launches, two small copies and two waits -- 45 us of which the kernels are 11 -- and this
// takes a copy, the stream's copy and a wait out of it.  *done = false: not this way (the buffer is not device-addressable, or the
// one-launch Minimize is off): the caller goes the usual way.
int update_host_write(rtx_ctx* ctx, const rtx_params* p, int mode, void* host_out, size_t* out_bytes, bool* done)
{
    *done = false;
    if (ctx->opt_min_fused == 0 || ((uintptr_t)host_out & 15u) != 0) return RTX_OK;
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    void* d_host = nullptr;
    if (hipHostGetDevicePointer(&d_host, host_out, 0) != hipSuccess || d_host == nullptr) {
        (void)hipGetLastError(); // pageable memory: the usual way
        return RTX_OK;
    }
    if (!ctx->h_pair) {
        if (hipHostMalloc((void**)&ctx->h_pair, 2 * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) {
            ctx->h_pair = nullptr;
            return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, "hipHostMalloc failed for the stream's length");
        }
    }
    void* d_pair = nullptr;
    RTX_HIP(ctx, hipHostGetDevicePointer(&d_pair, ctx->h_pair, 0));
    const size_t W = (size_t)p->x, H = (size_t)p->y;
    const uint32_t* d_words = nullptr;
    int rc = trace_words(ctx, p, mode, &ctx->d_words, &ctx->words_cap, &d_words);
    if (rc != RTX_OK) return rc;
    const uint64_t n_slots = (uint64_t)W * H;
    if ((rc = ensure_min_buffers(ctx, (size_t)((n_slots + rtx::kWSlotsPerBlock - 1) / rtx::kWSlotsPerBlock), false)) != RTX_OK) return rc;
    ctx->h_pair[0] = 0;
    ctx->h_pair[1] = 0;
    uint64_t* d_total = nullptr;
    if ((rc = launch_minimize_words(ctx, ctx->d_scan, mode, W, H, d_words, (uint8_t*)d_host, &d_total, 0u, (uint64_t*)d_pair)) != RTX_OK) return rc;
    RTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t total = 0;
    // (blocks that gave up: the three launches, into the same buffer, their length read back the usual way)
    if ((rc = settle_minimize_words(ctx, ctx->d_scan, mode, W, H, d_words, (uint8_t*)d_host, ctx->h_pair, &total)) != RTX_OK) return rc;
    ctx->stat_host_writes++;
    *out_bytes = (size_t)total;
    *done = true;
    return RTX_OK;
}

// rtx_update on a device group WITHOUT a gather (RTX_OPT_GROUP_UPDATE): the whole Update is bound by the copy of the minimised
// stream over one PCIe link, and a group has one link per device.  Every rank traces its rows as pixel words -- and the row above
// them, whose last pixel is the colour its first pixel is compared with (RayTracingManager.cu:181-319 carries the last emitted colour
// across rows) -- minimises its own rows (launch_minimize_words with `lead` = W) and, once the lengths of the ranks before it are
// known on the host, copies its part of the stream to its place in host_out from its own device.  The ranks' parts, in rank
// order, are the bytes the root would have made of the gathered frame.  false in *done: nothing was delivered, gather instead.
int update_group_direct(rtx_ctx* root, const rtx_params* p, int mode, void* host_out, size_t* out_bytes)
{
    const int N = rtx_group_size(root);
    const size_t W = (size_t)p->x, H = (size_t)p->y;
    struct Part {
        size_t row0 = 0, rows = 0, bytes = 0, offset = 0;
    };
    std::vector<Part> part((size_t)N);
    for (int r = 0; r < N; r++) {
        if (rtx_group_rows(root, H, r, &part[(size_t)r].row0, &part[(size_t)r].rows) != RTX_OK) return rtx_fail(root, RTX_ERR_INVALID_ARGUMENT, "rtx_group_rows failed");
    }
    struct Queued {
        const uint32_t* d_words = nullptr;
        uint32_t lead = 0;
    };
    std::vector<Queued> queued((size_t)N);
    // a rank's device work: trace (its rows and the one above), minimise, the two words of the result on their way to the host
    auto queue_rows = [&](int r, rtx_ctx* m) -> int {
        Part& q = part[(size_t)r];
        if (q.rows == 0) return RTX_OK;
        RTX_HIP(m, hipSetDevice(m->device));
        const size_t above = q.row0 > 0 ? 1u : 0u;
        int rc2 = ensure_words_buffer(m, &m->d_words, &m->words_cap, (q.rows + above) * W);
        if (rc2 != RTX_OK) return rc2;
        if (!m->h_pair && hipHostMalloc((void**)&m->h_pair, 2 * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) {
            m->h_pair = nullptr;
            return rtx_fail(m, RTX_ERR_OUT_OF_MEMORY, "hipHostMalloc failed for a rank's stream length");
        }
        if ((rc2 = rtx_render_rows(m, p, mode, q.row0 - above, q.rows + above, m->d_words, q.row0 - above, m->stream, RTX_RENDER_COMPACT)) != RTX_OK) return rc2;
        const uint64_t n_slots = (uint64_t)W * q.rows;
        if ((rc2 = ensure_min_buffers(m, (size_t)((n_slots + rtx::kWSlotsPerBlock - 1) / rtx::kWSlotsPerBlock), true)) != RTX_OK) return rc2;
        queued[(size_t)r].d_words = m->d_words + above * W;
        queued[(size_t)r].lead = (uint32_t)(above * W);
        uint64_t* d_total = nullptr;
        if ((rc2 = launch_minimize_words(m, m->d_scan, mode, W, q.rows, queued[(size_t)r].d_words, m->d_min, &d_total, queued[(size_t)r].lead)) != RTX_OK) return rc2;
        RTX_HIP(m, hipMemcpyAsync(m->h_pair, d_total, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, m->stream));
        return RTX_OK;
    };
    auto await_rows = [&](int r, rtx_ctx* m) -> int {
        Part& q = part[(size_t)r];
        if (q.rows == 0) return RTX_OK;
        RTX_HIP(m, hipSetDevice(m->device));
        RTX_HIP(m, hipStreamSynchronize(m->stream));
        uint64_t total = 0;
        const int rc2 = settle_minimize_words(m, m->d_scan, mode, W, q.rows, queued[(size_t)r].d_words, m->d_min, m->h_pair, &total, queued[(size_t)r].lead);
        if (rc2 != RTX_OK) return rc2;
        q.bytes = (size_t)total;
        return RTX_OK;
    };
    auto queue_copy = [&](int r, rtx_ctx* m) -> int {
        const Part& q = part[(size_t)r];
        if (q.bytes == 0) return RTX_OK;
        RTX_HIP(m, hipSetDevice(m->device));
        RTX_HIP(m, hipMemcpyAsync((uint8_t*)host_out + q.offset, m->d_min, q.bytes, hipMemcpyDeviceToHost, m->stream));
        return RTX_OK;
    };
    auto await_copy = [&](int r, rtx_ctx* m) -> int {
        if (part[(size_t)r].bytes == 0) return RTX_OK;
        RTX_HIP(m, hipSetDevice(m->device));
        RTX_HIP(m, hipStreamSynchronize(m->stream));
        return RTX_OK;
    };
    // With a submission thread per rank a rank queues and waits in one go (the ranks wait side by side); on the caller's thread alone
    // everything is queued first, so that the ranks' device work still overlaps.
    const bool threads = rtxgroup::threads_active(root);
    int rc;
    if (threads) {
        rc = rtxgroup::run_on_ranks(root, [&](int r, rtx_ctx* m) -> int {
            const int rc2 = queue_rows(r, m);
            return rc2 != RTX_OK ? rc2 : await_rows(r, m);
        });
    } else {
        rc = rtxgroup::run_on_ranks(root, queue_rows);
        const int rcw = rtxgroup::run_on_ranks(root, await_rows); // (also after a failure: nothing may still be running on a rank's buffers)
        if (rc == RTX_OK) rc = rcw;
    }
    if (rc != RTX_OK) {
        (void)hipSetDevice(root->device);
        return rc;
    }
    size_t at = 0;
    for (int r = 0; r < N; r++) {
        part[(size_t)r].offset = at;
        at += part[(size_t)r].bytes;
    }
    if (threads) {
        rc = rtxgroup::run_on_ranks(root, [&](int r, rtx_ctx* m) -> int {
            const int rc2 = queue_copy(r, m);
            return rc2 != RTX_OK ? rc2 : await_copy(r, m);
        });
    } else {
        rc = rtxgroup::run_on_ranks(root, queue_copy);
        const int rcw = rtxgroup::run_on_ranks(root, await_copy);
        if (rc == RTX_OK) rc = rcw;
    }
    RTX_HIP(root, hipSetDevice(root->device));
    if (rc != RTX_OK) return rc;
    *out_bytes = at;
    return RTX_OK;
}

} // namespace

extern "C" {

int rtx_update_objects(rtx_ctx* ctx, double dt)
{
    if (!ctx) return RTX_ERR_INVALID_ARGUMENT;
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = rtx_sync_scene(ctx);
    if (rc != RTX_OK) return rc;
    if (ctx->ns == 0) return RTX_OK;
    // cell lists being built ahead of time on the side stream read the sphere positions: the step waits for them
    for (auto& sl : ctx->cell_cache) {
        if (sl.ever_built && sl.built_on_aux) RTX_HIP(ctx, hipStreamWaitEvent(ctx->stream, sl.ev_built, 0));
    }
    const unsigned blocks = (ctx->ns + rtx::kThreads - 1) / rtx::kThreads;
    const bool sorted = ctx->sorted_gen == ctx->scene_gen && ctx->d_sorted_geom.p != nullptr;
    hipLaunchKernelGGL(rtx::rtx_update_spheres, dim3(blocks), dim3(rtx::kThreads), 0, ctx->stream,
                       (float4*)ctx->d_sph_geom.p, (float4*)ctx->d_sph_motion.p, ctx->ns, dt, sorted ? (float4*)ctx->d_sorted_geom.p : nullptr,
                       sorted ? (const uint32_t*)ctx->d_pos_of.p : nullptr);
    RTX_HIP(ctx, hipGetLastError());
    ctx->ns_moved_since_build = true;
    // How far a sphere can have moved (dispatch orders age with it; cell lists are valid within it: rtx_plan.hpp).  A step
    // moves a sphere by at most |speed dt| -- the clamp to [-10, 10] only shortens the move -- once it has been through one
    // step; the FIRST step after an edit may pull a sphere from anywhere onto +-10 (Sphere.cu:18-22), so it counts as
    // an edit.  A dt that is not a number moves spheres to NaN: an edit as well.
    const double step = std::fabs(dt) * (double)ctx->max_speed;
    if (!ctx->physics_settled || !(step == step) || !(step < 1.0e30)) {
        ctx->lists_gen++; // (object counts and array addresses are what they were: recorded graphs stay valid)
        ctx->cell_policy.invalidate();
        ctx->physics_settled = (step == step) && (step < 1.0e30);
        ctx->scene_drift += 1.0e3;
    } else {
        ctx->scene_drift += step + 2.0e-6; // (+ the rounding of y to float: half an ulp of 10)
    }
    return ctx->group ? rtxgroup::update_objects(ctx, dt) : RTX_OK; // every rank steps its replica: the same arithmetic on the same values
}

void* rtx_minimized_device_ptr(rtx_ctx* ctx) { return ctx ? ctx->d_min : nullptr; }

int rtx_ansi256_map(rtx_ctx* ctx, uint32_t first_rgb, size_t count, void* d_out, void* stream_v)
{
    if (!ctx || (count && !d_out)) return RTX_ERR_INVALID_ARGUMENT;
    if ((uint64_t)first_rgb + (uint64_t)count > (1ull << 24)) return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "rtx_ansi256_map: range exceeds 2^24 colours");
    if (count == 0) return RTX_OK;
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream_v ? (hipStream_t)stream_v : ctx->stream;
    const uint64_t threads = ((uint64_t)count + 3u) / 4u;
    const unsigned blocks = (unsigned)((threads + rtx::kThreads - 1) / rtx::kThreads);
    hipLaunchKernelGGL(rtx::rtx_ansi_map, dim3(blocks), dim3(rtx::kThreads), 0, st, first_rgb, (uint64_t)count, ctx->d_grey, (uint8_t*)d_out);
    RTX_HIP(ctx, hipGetLastError());
    return RTX_OK;
}

int rtx_minimize(rtx_ctx* ctx, int mode, size_t w, size_t h, const void* d_in, void* d_out, size_t* out_bytes)
{
    if (!ctx || !out_bytes) return RTX_ERR_INVALID_ARGUMENT;
    if (mode < RTX_BIT_ASCII || mode > RTX_SDL) return rtx_fail(ctx, RTX_ERR_INVALID_MODE, "invalid rendering mode");
    if (w == 0 || h == 0 || w >= (1ull << 31) || h >= (1ull << 31)) return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "w/h must be in [1, 2^31)");
    if (!d_in) {
        if (20 * w * h > ctx->capacity) return rtx_fail(ctx, RTX_ERR_TOO_LARGE, "frame larger than the context was created for");
        d_in = ctx->d_frame;
    }
    if (((uintptr_t)d_in & 15u) != 0 || (d_out && ((uintptr_t)d_out & 15u) != 0)) {
        return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "minimise buffers must be 16-byte aligned");
    }
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n_slots = (uint64_t)w * h;
    const size_t n_blocks = (size_t)((n_slots + rtx::kSlotsPerBlock - 1) / rtx::kSlotsPerBlock);
    if (!d_out && 20 * w * h > ctx->capacity) return rtx_fail(ctx, RTX_ERR_TOO_LARGE, "minimise output larger than the context's buffer");
    int rc = ensure_min_buffers(ctx, n_blocks, d_out == nullptr);
    if (rc != RTX_OK) return rc;
    if (!d_out) d_out = ctx->d_min;
    uint64_t* d_total = nullptr;
    rc = launch_minimize(ctx, ctx->d_scan, mode, w, h, (const uint8_t*)d_in, (uint8_t*)d_out, &d_total);
    if (rc != RTX_OK) return rc;
    uint64_t got[2] = {0, 0}, total = 0;
    RTX_HIP(ctx, hipMemcpyAsync(got, d_total, sizeof got, hipMemcpyDeviceToHost, ctx->stream));
    RTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = settle_minimize(ctx, ctx->d_scan, mode, w, h, (const uint8_t*)d_in, (uint8_t*)d_out, got, &total)) != RTX_OK) return rc;
    *out_bytes = (size_t)total;
    return RTX_OK;
}

int rtx_minimize_words(rtx_ctx* ctx, int mode, size_t w, size_t h, const void* d_words, void* d_out, size_t* out_bytes)
{
    if (!ctx || !out_bytes || !d_words) return RTX_ERR_INVALID_ARGUMENT;
    if (mode < RTX_BIT_ASCII || mode >= RTX_SDL) return rtx_fail(ctx, RTX_ERR_INVALID_MODE, "rtx_minimize_words: not a character mode");
    if (w == 0 || h == 0 || w >= (1ull << 31) || h >= (1ull << 31)) return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "w/h must be in [1, 2^31)");
    if (((uintptr_t)d_words & 3u) != 0 || (d_out && ((uintptr_t)d_out & 15u) != 0)) {
        return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "rtx_minimize_words: words must be 4-byte, the output 16-byte aligned");
    }
    if (!d_out && 20 * w * h > ctx->capacity) return rtx_fail(ctx, RTX_ERR_TOO_LARGE, "minimise output larger than the context's buffer");
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n_slots = (uint64_t)w * h;
    const size_t n_blocks = (size_t)((n_slots + rtx::kWSlotsPerBlock - 1) / rtx::kWSlotsPerBlock);
    int rc = ensure_min_buffers(ctx, n_blocks, d_out == nullptr);
    if (rc != RTX_OK) return rc;
    if (!d_out) d_out = ctx->d_min;
    uint64_t* d_total = nullptr;
    if ((rc = launch_minimize_words(ctx, ctx->d_scan, mode, w, h, (const uint32_t*)d_words, (uint8_t*)d_out, &d_total)) != RTX_OK) return rc;
    uint64_t got[2] = {0, 0}, total = 0;
    RTX_HIP(ctx, hipMemcpyAsync(got, d_total, sizeof got, hipMemcpyDeviceToHost, ctx->stream));
    RTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = settle_minimize_words(ctx, ctx->d_scan, mode, w, h, (const uint32_t*)d_words, (uint8_t*)d_out, got, &total)) != RTX_OK) return rc;
    *out_bytes = (size_t)total;
    return RTX_OK;
}

int rtx_update(rtx_ctx* ctx, const rtx_params* params, int mode, double dt, int run_physics, void* host_out, size_t* out_bytes)
{
    if (!ctx || !params || !host_out || !out_bytes) return RTX_ERR_INVALID_ARGUMENT;
    int rc;
    // RayTracingManager.cu:89-107: physics first
    if (run_physics && (rc = rtx_update_objects(ctx, dt)) != RTX_OK) return rc;
    size_t n = 0;
    if (update_from_words(ctx, mode)) {
        // :120-134 + :146 on 4-byte pixel words: the trace stores words, the minimise pass reads them and writes the very
        // stream it would have made of the records (which are never written: the context's frame buffer keeps what it held)
        if (mode < RTX_BIT_ASCII || mode > RTX_SDL) return rtx_fail(ctx, RTX_ERR_INVALID_MODE, "invalid rendering mode");
        if (params->x == 0 || params->y == 0 || 20 * (uint64_t)params->x * (uint64_t)params->y > ctx->capacity) {
            return rtx_fail(ctx, params->x && params->y ? RTX_ERR_TOO_LARGE : RTX_ERR_INVALID_ARGUMENT, "frame larger than the context was created for, or empty");
        }
        if (rtxgroup::update_direct_wanted(ctx)) {
            // no gather: every rank minimises its own rows and copies them over its own PCIe link
            rc = update_group_direct(ctx, params, mode, host_out, &n);
            rtxgroup::update_direct_done(ctx, rc == RTX_OK);
            if (rc == RTX_OK) {
                *out_bytes = n;
                return RTX_OK;
            }
            (void)hipGetLastError(); // (the group gathers on its root from now on; this frame too)
        }
        if (ctx->opt_update_host_write != 0) { // (a group that gathers on its root: the root's Minimize launch writes the host buffer)
            // the Minimize launch writes the stream into the caller's pinned buffer itself (RTX_OPT_UPDATE_HOST_WRITE): one host wait.  At
            // any size in this blocking form -- 1080p: 0.370 ms against 0.395 with the copy queued after a wait for the length
            bool done = false;
            if ((rc = update_host_write(ctx, params, mode, host_out, &n, &done)) != RTX_OK) return rc;
            if (done) {
                *out_bytes = n;
                return RTX_OK;
            }
        }
        const uint32_t* d_words = nullptr;
        if ((rc = trace_words(ctx, params, mode, &ctx->d_words, &ctx->words_cap, &d_words)) != RTX_OK) return rc;
        if ((rc = rtx_minimize_words(ctx, mode, (size_t)params->x, (size_t)params->y, d_words, nullptr, &n)) != RTX_OK) return rc;
        if (n) {
            RTX_HIP(ctx, hipMemcpyAsync(host_out, ctx->d_min, n, hipMemcpyDeviceToHost, ctx->stream));
            RTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        *out_bytes = n;
        return RTX_OK;
    }
    // :86 + :120-134: zero semantics and trace
    if ((rc = rtx_render(ctx, params, mode)) != RTX_OK) return rc;
    // :146: minimise on the device; :143 then only moves the minimised stream across PCIe
    if ((rc = rtx_minimize(ctx, mode, (size_t)params->x, (size_t)params->y, nullptr, nullptr, &n)) != RTX_OK) return rc;
    if (n) {
        RTX_HIP(ctx, hipMemcpyAsync(host_out, ctx->d_min, n, hipMemcpyDeviceToHost, ctx->stream));
        RTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    *out_bytes = n;
    return RTX_OK;
}

// ---- pipelined Update: the frame sequence of RayTracingManager::Update split in two calls so
// that the copy of frame k's minimised stream to the host overlaps the trace of frame k+1.
int rtx_update_begin(rtx_ctx* ctx, const rtx_params* params, int mode, double dt, int run_physics, void* host_out, int* ticket)
{
    if (!ctx || !params || !host_out || !ticket) return RTX_ERR_INVALID_ARGUMENT;
    if (mode < RTX_BIT_ASCII || mode > RTX_SDL) return rtx_fail(ctx, RTX_ERR_INVALID_MODE, "invalid rendering mode");
    const size_t w = (size_t)params->x, h = (size_t)params->y;
    if (w == 0 || h == 0 || 20 * w * h > ctx->capacity) return rtx_fail(ctx, RTX_ERR_TOO_LARGE, "frame larger than the context was created for");
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    const unsigned si = ctx->upd_next;
    rtx_ctx::UpdateSlot& sl = ctx->upd[si];
    if (sl.busy) return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "rtx_update_begin: both slots are in flight; call rtx_update_end first");
    if (!ctx->copy_stream) {
        // A stream of another priority than the render streams': the runtime spreads a process's streams of one priority over four
        // hardware queues, and a copy that lands in the queue of the context's own stream holds back the next frame's kernels until
        // it is done -- the pipelined Update then costs copy + kernels (0.376 ms) instead of the copy alone (0.326), which is what
        // bench.py's default line measured whenever four render streams had been created first.
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest ||
            hipStreamCreateWithPriority(&ctx->copy_stream, hipStreamNonBlocking, greatest) != hipSuccess) {
            (void)hipGetLastError();
            ctx->copy_stream = nullptr;
            RTX_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        }
    }
    // every member under its own null check: a call that fails half-way (out of memory) leaves a slot the next
    // call completes, instead of one that looks initialised with null buffers behind it
    const bool from_words = update_from_words(ctx, mode);
    if (!from_words && !sl.d_frame) {
        uint8_t* f = nullptr;
        if (hipMalloc((void**)&f, ctx->capacity) != hipSuccess) return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, "hipMalloc failed for an update slot's frame");
        const hipError_t me = hipMemsetAsync(f, 0, ctx->capacity, ctx->stream);
        if (me != hipSuccess) {
            hipFree(f);
            return rtx_hip_fail(ctx, me, "hipMemsetAsync(update slot frame)");
        }
        sl.d_frame = f;
    }
    if (!sl.d_min && hipMalloc((void**)&sl.d_min, ctx->capacity) != hipSuccess) {
        sl.d_min = nullptr;
        return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, "hipMalloc failed for an update slot's minimise buffer");
    }
    if (!sl.h_total && hipHostMalloc((void**)&sl.h_total, 2 * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) {
        sl.h_total = nullptr;
        return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, "hipHostMalloc failed for an update slot");
    }
    if (!sl.ev_ready) RTX_HIP(ctx, hipEventCreateWithFlags(&sl.ev_ready, hipEventDisableTiming));
    if (!sl.ev_copied) RTX_HIP(ctx, hipEventCreateWithFlags(&sl.ev_copied, hipEventDisableTiming));
    const uint64_t n_slots = (uint64_t)w * h;
    const size_t n_blocks = (size_t)((n_slots + rtx::kSlotsPerBlock - 1) / rtx::kSlotsPerBlock);
    const size_t need = n_blocks * (sizeof(uint32_t) + sizeof(uint64_t)) + 64;
    if (sl.scan_bytes < need) {
        if (sl.d_scan) hipFree(sl.d_scan);
        sl.d_scan = nullptr;
        sl.scan_bytes = 0;
        RTX_HIP(ctx, hipMalloc(&sl.d_scan, need));
        sl.scan_bytes = need;
    }
    int rc;
    if (run_physics && (rc = rtx_update_objects(ctx, dt)) != RTX_OK) return rc;
    if (from_words && rtxgroup::update_direct_wanted(ctx)) {
        // a group whose ranks minimise and copy their own rows (RTX_OPT_GROUP_UPDATE): N links carry the stream at once, which is
        // worth more than the overlap of one link's copy with the next trace -- the frame is complete when this call returns
        size_t n = 0;
        rc = update_group_direct(ctx, params, mode, host_out, &n);
        rtxgroup::update_direct_done(ctx, rc == RTX_OK);
        if (rc == RTX_OK) {
            sl.bytes = n;
            RTX_HIP(ctx, hipEventRecord(sl.ev_copied, ctx->copy_stream));
            sl.busy = true;
            *ticket = (int)si;
            ctx->upd_next = si ^ 1u;
            return RTX_OK;
        }
        (void)hipGetLastError(); // (the group gathers on its root from now on; this frame too)
    }
    uint64_t* d_total = nullptr;
    const uint32_t* d_words = nullptr;
    sl.host_write = false;
    if (from_words && !ctx->group && ctx->opt_min_fused != 0 && ((uintptr_t)host_out & 15u) == 0 &&
        (ctx->opt_update_host_write > 0 || (ctx->opt_update_host_write < 0 && (uint64_t)w * h <= (1u << 17)))) {
        // a small frame: the Minimize launch stores the stream and its length in host memory itself, and NOTHING is waited for here --
        // rtx_update_end waits for the frame.  (At console sizes the copy form's two waits per frame were what the pipelined Update cost.)
        void *d_host = nullptr, *d_pair = nullptr;
        if (hipHostGetDevicePointer(&d_host, host_out, 0) == hipSuccess && d_host != nullptr && hipHostGetDevicePointer(&d_pair, sl.h_total, 0) == hipSuccess) {
            if ((rc = trace_words(ctx, params, mode, &sl.d_words, &sl.words_cap, &d_words)) != RTX_OK) return rc;
            sl.h_total[0] = 0;
            sl.h_total[1] = 0;
            if ((rc = launch_minimize_words(ctx, sl.d_scan, mode, w, h, d_words, (uint8_t*)d_host, &d_total, 0u, (uint64_t*)d_pair)) != RTX_OK) return rc;
            RTX_HIP(ctx, hipEventRecord(sl.ev_ready, ctx->stream));
            sl.host_write = true;
            sl.hw_epoch = ctx->min_fused_epoch;
            sl.hw_mode = mode;
            sl.hw_w = w;
            sl.hw_h = h;
            sl.hw_words = d_words;
            sl.hw_out = (uint8_t*)d_host;
            ctx->stat_host_writes++;
            sl.busy = true;
            *ticket = (int)si;
            ctx->upd_next = si ^ 1u;
            return RTX_OK;
        }
        (void)hipGetLastError(); // pageable memory: the copy form
    }
    if (from_words) {
        // pixel words into the slot's own buffer (a group: into the group's, gathered), minimised from there
        if ((rc = trace_words(ctx, params, mode, &sl.d_words, &sl.words_cap, &d_words)) != RTX_OK) return rc;
        if ((rc = launch_minimize_words(ctx, sl.d_scan, mode, w, h, d_words, sl.d_min, &d_total)) != RTX_OK) return rc;
    } else {
    // the slot's frame buffer is caller-style memory for rtx_render_rows: whole frame, with the zero
    // semantics of the per-frame memset (the buffer starts zeroed; SDL frames write nothing, so clear)
    const bool rgb = mode >= RTX_RGB_ASCII;
    if (mode == RTX_SDL) {
        RTX_HIP(ctx, hipMemsetAsync(sl.d_frame, 0, 20 * w * h, ctx->stream));
    }
    if (ctx->group) {
        if ((rc = rtxgroup::render_frame(ctx, params, mode, sl.d_frame, rgb ? RTX_RENDER_DEFAULT : RTX_RENDER_ZERO_TAIL)) != RTX_OK) return rc;
    } else if ((rc = rtx_render_rows(ctx, params, mode, 0, h, sl.d_frame, 0, ctx->stream, rgb ? RTX_RENDER_DEFAULT : RTX_RENDER_ZERO_TAIL)) != RTX_OK) {
        return rc;
    }
    if ((rc = launch_minimize(ctx, sl.d_scan, mode, w, h, sl.d_frame, sl.d_min, &d_total)) != RTX_OK) return rc;
    }
    RTX_HIP(ctx, hipMemcpyAsync(sl.h_total, d_total, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    RTX_HIP(ctx, hipEventRecord(sl.ev_ready, ctx->stream));
    // the length is needed on the host to size the copy: wait for this frame's kernels (the previous frame's
    // copy keeps running on the copy stream meanwhile)
    RTX_HIP(ctx, hipEventSynchronize(sl.ev_ready));
    {
        uint64_t total = 0;
        if (from_words) {
            rc = settle_minimize_words(ctx, sl.d_scan, mode, w, h, d_words, sl.d_min, sl.h_total, &total);
        } else {
            rc = settle_minimize(ctx, sl.d_scan, mode, w, h, sl.d_frame, sl.d_min, sl.h_total, &total);
        }
        if (rc != RTX_OK) return rc;
        sl.h_total[0] = total;
    }
    sl.bytes = (size_t)*sl.h_total;
    if (sl.bytes) {
        RTX_HIP(ctx, hipMemcpyAsync(host_out, sl.d_min, sl.bytes, hipMemcpyDeviceToHost, ctx->copy_stream));
    }
    RTX_HIP(ctx, hipEventRecord(sl.ev_copied, ctx->copy_stream));
    sl.busy = true;
    *ticket = (int)si;
    ctx->upd_next = si ^ 1u;
    return RTX_OK;
}

int rtx_update_end(rtx_ctx* ctx, int ticket, size_t* out_bytes)
{
    if (!ctx || !out_bytes || ticket < 0 || ticket > 1) return RTX_ERR_INVALID_ARGUMENT;
    rtx_ctx::UpdateSlot& sl = ctx->upd[ticket];
    if (!sl.busy) return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "rtx_update_end: no frame in flight under this ticket");
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    if (sl.host_write) {
        RTX_HIP(ctx, hipEventSynchronize(sl.ev_ready));
        sl.host_write = false;
        sl.busy = false;
        uint64_t total = 0;
        ctx->min_fused_epoch = sl.hw_epoch;
        const int rc = settle_minimize_words(ctx, sl.d_scan, sl.hw_mode, sl.hw_w, sl.hw_h, sl.hw_words, sl.hw_out, sl.h_total, &total);
        if (rc != RTX_OK) return rc;
        *out_bytes = (size_t)total;
        return RTX_OK;
    }
    RTX_HIP(ctx, hipEventSynchronize(sl.ev_copied));
    *out_bytes = sl.bytes;
    sl.busy = false;
    return RTX_OK;
}

} // extern "C"


// ---- the direction-sorted copy of the sphere array (what KArgs::sph_geom and sph_od point at for the trace kernels): spheres ordered by a Morton code of the direction
// (azimuth, elevation) in which they lie from `origin` -- the camera's position at the first launch after a scene edit -- so that
// the spheres of a coarse cell, or of a macro tile's pyramid, are neighbours in memory.  Host-side sort (rtxplan::direction_order)
// over the positions the spheres were created with (physics moves them by a few units at most: the order stays good enough),
// then one gather on the device from the live arrays.
int rtx_sort_scene(rtx_ctx* ctx, const float origin[3])
{
    const uint32_t ns = ctx->ns;
    if (ns == 0 || ctx->h_centres.size() != ns) return RTX_OK;
    std::vector<uint32_t> order, pos_of;
    rtxplan::direction_order(&ctx->h_centres[0].x, sizeof(float4) / sizeof(float), ns, origin, order, pos_of);
    // (a scene edit: nothing may still read the old copy)
    RTX_HIP(ctx, hipDeviceSynchronize());
    for (DeviceArray* a : {&ctx->d_sorted_geom, &ctx->d_sorted_od, &ctx->d_sorted_idx, &ctx->d_pos_of}) {
        const size_t elem = (a == &ctx->d_sorted_geom || a == &ctx->d_sorted_od) ? sizeof(float4) : sizeof(uint32_t);
        if (a->cap < ns) {
            if (a->p) hipFree(a->p);
            a->p = nullptr;
            a->cap = 0;
            size_t cap = 1024;
            while (cap < ns) cap *= 2;
            if (hipMalloc(&a->p, cap * elem) != hipSuccess) {
                (void)hipGetLastError();
                return RTX_OK; // no sorted copy: staging reads the scene array (sorted_gen stays behind)
            }
            a->cap = cap;
        }
    }
    RTX_HIP(ctx, hipMemcpyAsync(ctx->d_sorted_idx.p, order.data(), ns * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    RTX_HIP(ctx, hipMemcpyAsync(ctx->d_pos_of.p, pos_of.data(), ns * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(rtx::rtx_gather_spheres, dim3((ns + rtx::kThreads - 1) / rtx::kThreads), dim3(rtx::kThreads), 0, ctx->stream,
                       (const float4*)ctx->d_sph_geom.p, (const float4*)ctx->d_sph_od.p, (const uint32_t*)ctx->d_sorted_idx.p,
                       (float4*)ctx->d_sorted_geom.p, (float4*)ctx->d_sorted_od.p, ns);
    RTX_HIP(ctx, hipGetLastError());
    RTX_HIP(ctx, hipStreamSynchronize(ctx->stream)); // (the staging vectors go out of scope; other streams may render next)
    ctx->sorted_gen = ctx->scene_gen;
    return RTX_OK;
}
