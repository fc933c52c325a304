// rtx_render.cpp -- the launches behind the C ABI (include/rtx.h): rtx_render_rows and what is built on it (rtx_render,
// rtx_submit_frames, rtx_submit_slabs with its batched launch), rtx_expand, recorded launches (rtx_graph_*).
//
// Replaces RayTracing::RayTrace's dispatch (RayTracing.cu:170-199, 757-768) and the trace step of RayTracingManager::Update
// (RayTracingManager.cu:120-134).  Context, options and the scene store: rtx_api.cpp.
#include "rtx_ctx.h"
#include "rtx_group.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#define RTX_X_SECTION_HOST
#include "rtx_experiment.inc"
#undef RTX_X_SECTION_HOST

namespace {

// horizontal / vertical extent of one pixel on the view plane
double pixel_aspect(const rtx_params* p)
{
    double sx, sy;
    rtxplan::pixel_steps(p->inv_v, p->element1, p->element2, p->x, p->y, &sx, &sy);
    return sx / sy;
}

bool uses_culling_kernel(const rtx_ctx* ctx)
{
    switch (ctx->opt_kernel) {
    case RTX_KERNEL_BRUTE: return false;
    case RTX_KERNEL_BINNED: return true;
    default: return ctx->ns > 64; // below that the frustum set-up costs more than it saves
    }
}

// (a locally dense view -- rtxplan::ViewDensity -- is planned like a dense scene: two-level culling from 256 spheres on)
bool uses_two_level(const rtx_ctx* ctx, bool view_dense = false)
{
    const uint32_t from = view_dense ? 256u : 2048u;
    return uses_culling_kernel(ctx) && ctx->ns > 0 && (ctx->opt_two_level >= 1 || (ctx->opt_two_level < 0 && ctx->ns >= from));
}

// A stream in capture (rtx_graph_begin) records launches: nothing that synchronises or keeps per-launch state may
// run.  Checked before anything is queued, so that a refused call leaves the capture as it found it.
int check_recordable(rtx_ctx* ctx, hipStream_t stream, bool* capturing)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    RTX_HIP(ctx, hipStreamIsCapturing(stream, &st));
    *capturing = st == hipStreamCaptureStatusActive;
    if (!*capturing) return RTX_OK;
    if (ctx->ns > ctx->ns_uploaded || ctx->np > ctx->np_uploaded) {
        return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "graph capture: the scene has objects that are not uploaded yet (render once before capturing)");
    }
    if (uses_two_level(ctx)) {
        return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "graph capture: launches with the two-level pre-pass cannot be recorded (its counters alternate per launch)");
    }
    return RTX_OK;
}

} // namespace

extern "C" {

// ------------------------------------------------------------------ render
//
// rtx_render_rows = validate -> zero-fill semantics -> plan (rtx_plan.hpp: tile shape, cell grid, dispatch-order and
// cell-reuse decisions, all pure and unit-tested on the CPU) -> the buffers, events and launches those decisions call for.

namespace {

// The per-frame memset of RayTracingManager::Update (RayTracingManager.cu:86,161-165), reduced to the bytes an earlier
// frame can have left non-zero: the trace kernel writes column W-1's NULs itself, so only the 8-bit modes' unused tail
// (and an SDL frame, which writes nothing) ever needs zeroing.
int zero_fill_semantics(rtx_ctx* ctx, int mode, uint64_t W, uint64_t H, void* d_out, bool own, bool compact, unsigned flags, hipStream_t stream)
{
    const bool rgb = mode >= RTX_RGB_ASCII;
    if (mode == RTX_SDL && own && ctx->dirty_hi > 0) {
        // RayTrace_SDL writes nothing (RayTracing.cu:787-794), so the frame is what the per-frame memset left: all NUL
        const int ze = rtx_k_launch_zero(d_out, (ctx->dirty_hi + 3) & ~(size_t)3, stream);
        if (ze != 0) return rtx_hip_fail(ctx, (hipError_t)ze, "zero-fill launch");
        ctx->dirty_hi = 0;
    }
    if (!rgb && mode != RTX_SDL && !compact) {
        size_t zero_lo = 12 * W * H, zero_hi = zero_lo;
        if (own) {
            zero_hi = ctx->dirty_hi > zero_lo ? (ctx->dirty_hi < 20 * W * H ? ctx->dirty_hi : 20 * W * H) : zero_lo;
        } else if (flags & RTX_RENDER_ZERO_TAIL) {
            zero_hi = 20 * W * H;
        }
        if (zero_hi > zero_lo) {
            const int ze = rtx_k_launch_zero((uint8_t*)d_out + zero_lo, zero_hi - zero_lo, stream);
            if (ze != 0) return rtx_hip_fail(ctx, (hipError_t)ze, "zero-fill launch");
        }
    }
    if (own && mode != RTX_SDL) {
        // bytes [0, dirty_hi) of the context's buffer may be non-zero afterwards
        const size_t frame_hi = 20 * W * H;
        if (rgb) {
            ctx->dirty_hi = ctx->dirty_hi > frame_hi ? ctx->dirty_hi : frame_hi;
        } else if (ctx->dirty_hi <= frame_hi) {
            ctx->dirty_hi = 12 * W * H; // the tail up to frame_hi was zero or has just been zeroed
        }
    }
    return RTX_OK;
}

// The library's own side stream (balancing passes, cell lists built ahead of time).  One per context, whatever the
// number of render streams: every stream a process creates shifts how the others share its four hardware queues.
hipStream_t side_stream(rtx_ctx* ctx)
{
    if (!ctx->aux_stream && hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        ctx->aux_stream = nullptr;
    }
    return ctx->aux_stream;
}

// An event of the slot's pool recorded on `stream` now: "everything this stream has queued so far that reads the slot".
// A stream handle the caller has destroyed since cannot be recorded on: then the whole device is waited for instead.
void slot_mark_done(rtx_ctx::CellCacheSlot& sl, hipStream_t stream)
{
    if (sl.n_done == sl.done.size()) {
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            hipDeviceSynchronize();
            return;
        }
        sl.done.push_back(ev);
    }
    if (hipEventRecord(sl.done[sl.n_done], stream) != hipSuccess) {
        (void)hipGetLastError();
        hipDeviceSynchronize();
        return;
    }
    sl.n_done++;
}

void remove_stream(std::vector<hipStream_t>& v, hipStream_t s)
{
    for (size_t i = 0; i < v.size(); i++) {
        if (v[i] == s) {
            v[i] = v.back();
            v.pop_back();
            return;
        }
    }
}

bool contains(const std::vector<hipStream_t>& v, hipStream_t s)
{
    for (hipStream_t e : v) {
        if (e == s) return true;
    }
    return false;
}

// ---- two-level culling: cell lists

void fill_cell_args(KArgs& a, const rtxplan::CellGrid& g)
{
    a.cell_log2gx = g.gx;
    a.cell_log2gy = g.gy;
    a.cells_x = g.cells_x;
    a.cells_y = g.cells_y;
    a.cell_cap = g.cap;
}

// Per-frame lists in the render stream's own scratch (the form before round 3; still what a fast-moving camera gets):
// one binning launch in front of every trace launch, counters alternating between two buffers so that no memset sits
// between frames.
int cells_per_frame(rtx_ctx* ctx, hipStream_t stream, KArgs& a, const rtxplan::CellGrid& g)
{
    const size_t need_list = (size_t)g.n_cells * g.cap, need_count = 2 * (size_t)g.n_cells;
    rtx_ctx::CellScratch* cs = nullptr;
    for (auto& e : ctx->cell_scratch) {
        if (e.stream == stream) cs = &e;
    }
    if (!cs) {
        if (ctx->cell_scratch.size() >= 16) {
            // a caller that keeps creating streams: recycle the oldest set (its stream may be gone, so wait for the device)
            hipDeviceSynchronize();
            rtx_ctx::CellScratch& old = ctx->cell_scratch.front();
            if (old.list) hipFree(old.list);
            if (old.count) hipFree(old.count);
            ctx->cell_scratch.erase(ctx->cell_scratch.begin());
        }
        ctx->cell_scratch.emplace_back();
        cs = &ctx->cell_scratch.back();
        cs->stream = stream;
    }
    if (cs->list_words < need_list) {
        // launches queued on this stream may still read the old lists
        if (cs->list) {
            hipStreamSynchronize(stream);
            hipFree(cs->list);
        }
        cs->list = nullptr;
        cs->list_words = 0;
        if (hipMalloc((void**)&cs->list, need_list * sizeof(uint32_t)) != hipSuccess) {
            return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, "hipMalloc failed for the coarse cell lists");
        }
        cs->list_words = need_list;
    }
    if (cs->count_words < need_count || cs->n_cells != g.n_cells) {
        // (re)start the two alternating counter buffers from zero for this cell grid
        if (cs->count_words < need_count) {
            if (cs->count) {
                hipStreamSynchronize(stream);
                hipFree(cs->count);
            }
            cs->count = nullptr;
            cs->count_words = 0;
            if (hipMalloc((void**)&cs->count, need_count * sizeof(uint32_t)) != hipSuccess) {
                return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, "hipMalloc failed for the coarse cell counts");
            }
            cs->count_words = need_count;
        }
        RTX_HIP(ctx, hipMemsetAsync(cs->count, 0, cs->count_words * sizeof(uint32_t), stream));
        cs->n_cells = g.n_cells;
        cs->flip = 0;
    }
    fill_cell_args(a, g);
    a.bin_theta = a.bin_delta = 0.0f;
    a.cell_list_out = cs->list;
    a.cell_count_out = cs->count + (size_t)cs->flip * g.n_cells;
    a.cell_count_zero = cs->count + (size_t)(cs->flip ^ 1u) * g.n_cells;
    cs->flip ^= 1u;
    a.cell_max_out = ctx->d_cell_max;
    const int be = rtx_k_launch_bin_cells(&a, g.splits, stream);
    if (be != 0) return rtx_hip_fail(ctx, (hipError_t)be, "cell binning launch");
    if (ctx->d_cell_max && (ctx->per_frame_bins++ & 15u) == 0u) { // (a copy packet between kernels costs a few us: now and then only)
        RTX_HIP(ctx, hipMemcpyAsync((void*)ctx->h_cell_max, ctx->d_cell_max, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    }
    a.cell_list = cs->list;
    a.cell_count = a.cell_count_out;
    ctx->stat_cell_per_frame++;
    return RTX_OK;
}

// Bins into cache slot `si` on `build` (the render stream that missed, or the side stream), with the motion budget the
// lists are to last for.  Ordered after every launch that still reads the slot's old lists and after its last build.
int cell_slot_build(rtx_ctx* ctx, int si, hipStream_t build, hipStream_t render, bool on_side, KArgs a, const rtxplan::CellGrid& g, const rtxplan::CellBudget& budget)
{
    rtx_ctx::CellCacheSlot& sl = ctx->cell_cache[si];
    const size_t need_list = (size_t)g.n_cells * g.cap, need_count = (size_t)g.n_cells;
    if (sl.list_words < need_list || sl.count_words < need_count) {
        hipDeviceSynchronize(); // rare (a larger grid or scene): nothing may still read the old buffers
        if (sl.list_words < need_list) {
            if (sl.list) hipFree(sl.list);
            sl.list = nullptr;
            sl.list_words = 0;
            if (hipMalloc((void**)&sl.list, need_list * sizeof(uint32_t)) != hipSuccess) {
                return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, "hipMalloc failed for the coarse cell lists");
            }
            sl.list_words = need_list;
        }
        if (sl.count_words < need_count) {
            if (sl.count) hipFree(sl.count);
            sl.count = nullptr;
            sl.count_words = 0;
            if (hipMalloc((void**)&sl.count, need_count * sizeof(uint32_t)) != hipSuccess) {
                return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, "hipMalloc failed for the coarse cell counts");
            }
            sl.count_words = need_count;
        }
        sl.readers.clear();
        sl.n_done = 0;
    }
    if (!sl.ev_built) RTX_HIP(ctx, hipEventCreateWithFlags(&sl.ev_built, hipEventDisableTiming));
    // after the slot's previous build, and after the launches that read its old lists: the streams that have not launched
    // anything since their last read of it are marked now, the others were when they moved on (cell_slot_use)
    if (sl.ever_built) RTX_HIP(ctx, hipStreamWaitEvent(build, sl.ev_built, 0));
    for (hipStream_t r : sl.readers) {
        if (r != build) slot_mark_done(sl, r);
    }
    sl.readers.clear();
    for (size_t i = 0; i < sl.n_done; i++) {
        if (hipStreamWaitEvent(build, sl.done[i], 0) != hipSuccess) {
            (void)hipGetLastError();
            hipDeviceSynchronize();
        }
    }
    sl.n_done = 0;
    if (on_side && ctx->ns_moved_since_build) {
        // the sphere positions the lists are built from are those of this moment: physics steps queued so far (on the
        // context's stream) are done before the pass reads them; rtx_update_objects in turn waits for a pass in flight
        if (!ctx->ev_physics) RTX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_physics, hipEventDisableTiming));
        RTX_HIP(ctx, hipEventRecord(ctx->ev_physics, ctx->stream));
        RTX_HIP(ctx, hipStreamWaitEvent(build, ctx->ev_physics, 0));
        ctx->ns_moved_since_build = false;
    }
    (void)render;
    sl.waited.clear();
    RTX_HIP(ctx, hipMemsetAsync(sl.count, 0, need_count * sizeof(uint32_t), build));
    fill_cell_args(a, g);
    a.bin_theta = budget.theta;
    a.bin_delta = budget.delta;
    a.cell_list_out = sl.list;
    a.cell_count_out = sl.count;
    a.cell_count_zero = nullptr;
    a.cell_max_out = ctx->d_cell_max;
    const int be = rtx_k_launch_bin_cells(&a, g.splits, build);
    if (be != 0) return rtx_hip_fail(ctx, (hipError_t)be, "cell binning launch");
    RTX_HIP(ctx, hipEventRecord(sl.ev_built, build));
    if (ctx->d_cell_max) RTX_HIP(ctx, hipMemcpyAsync((void*)ctx->h_cell_max, ctx->d_cell_max, sizeof(uint32_t), hipMemcpyDeviceToHost, build));
    sl.ever_built = true;
    sl.known_ready = false;
    sl.built_on_aux = on_side;
    if (!on_side) sl.waited.push_back(build); // in order on the stream that built
    return RTX_OK;
}

// The trace launch on `stream` reads cache slot `si`.
int cell_slot_use(rtx_ctx* ctx, int si, hipStream_t stream, KArgs& a, const rtxplan::CellGrid& g)
{
    rtx_ctx::CellCacheSlot& sl = ctx->cell_cache[si];
    if (!contains(sl.waited, stream)) {
        RTX_HIP(ctx, hipStreamWaitEvent(stream, sl.ev_built, 0));
        sl.waited.push_back(stream);
    }
    // this stream's reads of the OTHER set end here: mark them, so that a rebuild of that set need not wait for anything
    // this stream queues from now on
    rtx_ctx::CellCacheSlot& other = ctx->cell_cache[si ^ 1];
    if (contains(other.readers, stream)) {
        slot_mark_done(other, stream);
        remove_stream(other.readers, stream);
    }
    if (!contains(sl.readers, stream)) {
        if (sl.readers.size() >= 32 || sl.n_done >= 64) { // a caller that keeps creating streams: settle everything, start over
            hipDeviceSynchronize();
            sl.readers.clear();
            sl.n_done = 0;
        }
        sl.readers.push_back(stream);
    }
    fill_cell_args(a, g);
    a.cell_list = sl.list;
    a.cell_count = sl.count;
    return RTX_OK;
}

// Static dispatch order of a two-level grid: position b (linear block index) -> macro tile, such that the blocks that
// share an XCD (b, b + 8, b + 16, ... under the observed round-robin placement; speed only, any permutation renders the
// same frame) walk a contiguous run of cells, cell by cell.  Then a cell's list and the spheres on it are fetched into
// ONE XCD's L2 instead of into as many as the cell has tiles (config 5: FETCH_SIZE of the trace kernel 9.8 -> about 3 MB
// per launch), and an XCD's L2 holds an eighth of the scene instead of all of it.
const uint32_t* xcd_tile_order(rtx_ctx* ctx, const rtxplan::TileShape& t, const rtxplan::CellGrid& g)
{
    const uint64_t n = (uint64_t)t.grid_x * t.grid_y;
    // (grids of several dispatch rounds only, unless asked for: where every workgroup is resident at once the tiles that share
    // a CU decide the launch's length, and those are dealt by measured work, not by neighbourhood -- C2 with two-level
    // culling: 30.4 us alone under this order, 24.7 under the balanced one)
    if (ctx->opt_xcd_order == 0 || (ctx->opt_xcd_order < 0 && n <= rtxplan::resident_slots(ctx->n_cu)) || n < 64 || n > (1u << 22) ||
        t.grid_x > 0xffffu || t.grid_y > 0xffffu) {
        return nullptr;
    }
    const uint64_t key[2] = {((uint64_t)t.grid_x << 32) | t.grid_y, ((uint64_t)g.gx << 32) | g.gy};
    // a small cache, so that a caller alternating a few grids on one context (slabs of different heights, two resolutions)
    // does not pay a device synchronisation and a blocking upload at every switch
    rtx_ctx::XcdOrder* slot = nullptr;
    for (auto& e : ctx->xcd_orders) {
        if (e.p && e.key[0] == key[0] && e.key[1] == key[1]) {
            e.last_use = ++ctx->order_clock;
            return e.p;
        }
        if (!slot || (slot->p && (!e.p || e.last_use < slot->last_use))) slot = &e; // an empty entry, else the least recently used
    }
    // (rare: a new grid) launches in flight may still read the order being replaced
    if (slot->p) hipDeviceSynchronize();
    if (slot->cap < n) {
        if (slot->p) hipFree(slot->p);
        slot->p = nullptr;
        slot->cap = 0;
        slot->key[0] = slot->key[1] = 0;
        if (hipMalloc((void**)&slot->p, n * sizeof(uint32_t)) != hipSuccess) {
            (void)hipGetLastError();
            slot->p = nullptr;
            return nullptr; // frame order
        }
        slot->cap = n;
    }
    std::vector<uint32_t> order(n);
    rtxplan::xcd_cell_order(t.grid_x, t.grid_y, g.gx, g.gy, order.data());
    if (hipMemcpy(slot->p, order.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        slot->key[0] = slot->key[1] = 0;
        return nullptr;
    }
    slot->key[0] = key[0];
    slot->key[1] = key[1];
    slot->last_use = ++ctx->order_clock;
    return slot->p;
}

// Two-level culling for this launch: cell lists from the cache when they cover the camera (building or prefetching as
// rtxplan::CellCachePolicy says), else binned for this frame alone.
// still_only: a scene too small for a pre-pass per frame gets lists only while camera and scene rest (CellCachePolicy).
int prepare_cells(rtx_ctx* ctx, const rtx_params* p, hipStream_t stream, const rtxplan::TileShape& shape, uint64_t row0, uint64_t rows, double aspect,
                  KArgs& a, const uint32_t** static_order, bool still_only)
{
    // how many render streams the caller keeps frames in flight on (of the last 16 two-level launches)
    ctx->recent_streams[ctx->recent_pos++ & 15u] = stream;
    {
        unsigned distinct = 0;
        for (unsigned i = 0; i < 16u; i++) {
            if (!ctx->recent_streams[i]) continue;
            bool dup = false;
            for (unsigned j = 0; j < i; j++) dup = dup || ctx->recent_streams[j] == ctx->recent_streams[i];
            distinct += dup ? 0u : 1u;
        }
        ctx->render_streams_seen = distinct;
    }
    // capacity feedback: one word per context, reset when the grid or the scene changes
    if (!ctx->d_cell_max) {
        void* h = nullptr;
        if (hipMalloc((void**)&ctx->d_cell_max, sizeof(uint32_t)) != hipSuccess || hipHostMalloc(&h, sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            if (ctx->d_cell_max) hipFree(ctx->d_cell_max);
            ctx->d_cell_max = nullptr; // no feedback: the default capacity and the whole-scene fallback
        } else {
            ctx->h_cell_max = (volatile uint32_t*)h;
            *ctx->h_cell_max = 0u;
            RTX_HIP(ctx, hipMemsetAsync(ctx->d_cell_max, 0, sizeof(uint32_t), stream));
        }
    }
    rtxplan::CellGrid g = rtxplan::plan_cells(shape, ctx->ns, aspect, ctx->opt_cell_capacity, 0);
    {
        const uint64_t id[3] = {(p->x << 32) | p->y, ((uint64_t)row0 << 32) | rows,
                                ((uint64_t)g.gx << 56) ^ ((uint64_t)g.gy << 48) ^ ((uint64_t)shape.grid_x << 24) ^ (uint64_t)shape.grid_y ^ (ctx->lists_gen << 8)};
        if (std::memcmp(id, ctx->cell_grid_id, sizeof id) != 0) {
            std::memcpy(ctx->cell_grid_id, id, sizeof id);
            ctx->cell_cap_floor = 0;
            if (ctx->d_cell_max) {
                // (in stream order after earlier passes on this stream; a pass still running on another stream may add its
                // maximum once more: harmless, it only keeps the lists a little longer than needed)
                RTX_HIP(ctx, hipMemsetAsync(ctx->d_cell_max, 0, sizeof(uint32_t), stream));
                *ctx->h_cell_max = 0u;
            }
        }
        if (ctx->h_cell_max) ctx->cell_cap_floor = rtxplan::cell_capacity_wanted(*ctx->h_cell_max, g.cap > ctx->cell_cap_floor ? g.cap : ctx->cell_cap_floor, ctx->cell_cap_floor);
        if (ctx->cell_cap_floor) g = rtxplan::plan_cells(shape, ctx->ns, aspect, ctx->opt_cell_capacity, ctx->cell_cap_floor);
    }
    *static_order = still_only ? nullptr : xcd_tile_order(ctx, shape, g);
    if (ctx->opt_cell_reuse == 0) return still_only ? RTX_OK : cells_per_frame(ctx, stream, a, g);
    rtxplan::CellKey key;
    key.W = p->x;
    key.H = p->y;
    key.row0 = row0;
    key.rows = rows;
    key.lw = shape.lw;
    key.lnx = shape.lnx;
    key.nsub = shape.nsub;
    key.gx = g.gx;
    key.gy = g.gy;
    key.cap = g.cap;
    key.ns = ctx->ns;
    rtxplan::CellCamera cam;
    cam.view = rtxplan::view_of(p->inv_v, p->cam_pos);
    cam.e1 = p->element1;
    cam.e2 = p->element2;
    cam.W = p->x;
    cam.H = p->y;
    cam.drift = ctx->scene_drift;
    cam.scene_gen = ctx->lists_gen;
    double sx, sy;
    rtxplan::pixel_steps(p->inv_v, p->element1, p->element2, p->x, p->y, &sx, &sy);
    bool ready[2];
    for (int s = 0; s < 2; s++) {
        rtx_ctx::CellCacheSlot& sl = ctx->cell_cache[s];
        if (sl.ever_built && !sl.known_ready) {
            if (hipEventQuery(sl.ev_built) == hipSuccess) sl.known_ready = true;
            else (void)hipGetLastError(); // hipErrorNotReady
        }
        ready[s] = sl.ever_built && sl.known_ready;
    }
    const rtxplan::CellCachePolicy::Decision d = ctx->cell_policy.decide(key, cam, sx * g.cell_w, sy * g.cell_h, ready, still_only);
    int rc = RTX_OK;
    switch (d.action) {
    case rtxplan::CellCachePolicy::kSkip:
        return RTX_OK; // no lists: the workgroups stage the whole scene
    case rtxplan::CellCachePolicy::kPerFrame:
        return cells_per_frame(ctx, stream, a, g);
    case rtxplan::CellCachePolicy::kBuild:
        if ((rc = cell_slot_build(ctx, d.slot, stream, stream, false, a, g, d.budget)) != RTX_OK) {
            ctx->cell_policy.invalidate();
            return rc;
        }
        ctx->stat_cell_builds++;
        break;
    case rtxplan::CellCachePolicy::kUse:
        ctx->stat_cell_hits++;
        break;
    }
    if (d.prefetch) {
        // Where the lists for the next stretch are built.  A caller with ONE render stream: on the library's side stream,
        // beside its frames (config 5, camera turning, one launch at a time: 66 us per frame against 78 with the pre-pass in
        // line).  A caller with frames in flight on several streams: on the render stream of this launch, in front of it --
        // the other streams' frames cover it, and no stream of the library's own has to find room among the process's four
        // hardware queues (the side stream there: 60 us per frame against 43 without reuse).
        const bool several = ctx->render_streams_seen >= 2;
        hipStream_t side = several ? stream : side_stream(ctx);
        if (!side || cell_slot_build(ctx, d.prefetch_slot, side, stream, !several, a, g, d.prefetch_budget) != RTX_OK) {
            ctx->cell_policy.invalidate(); // (the slot was promised to the policy; take everything back and bin this frame alone)
            return cells_per_frame(ctx, stream, a, g);
        }
        ctx->stat_cell_prefetches++;
    }
    return cell_slot_use(ctx, d.slot, stream, a, g);
}

// ---- dispatch order

// The set of order buffers for (stream, tile grid): found, or made (recycling the least recently used of 16).
rtx_ctx::TileOrder* tile_order_set(rtx_ctx* ctx, hipStream_t stream, const uint64_t key[3], uint32_t n_tiles)
{
    rtx_ctx::TileOrder* to = nullptr;
    for (auto& e : ctx->tile_orders) {
        if (e.stream == stream && std::memcmp(e.key, key, sizeof e.key) == 0) to = &e;
    }
    if (!to) {
        if (ctx->tile_orders.size() >= 16) {
            // (a set whose order a recorded launch reads is never recycled: the graph keeps its address)
            size_t lru = ctx->tile_orders.size();
            for (size_t i = 0; i < ctx->tile_orders.size(); i++) {
                if (!ctx->tile_orders[i].frozen && (lru == ctx->tile_orders.size() || ctx->tile_orders[i].last_use < ctx->tile_orders[lru].last_use)) lru = i;
            }
            if (lru == ctx->tile_orders.size()) return nullptr; // every set is frozen: frame order
            hipDeviceSynchronize(); // launches (and a pass) may still use its buffers; its stream may be gone
            rtx_ctx::TileOrder& old = ctx->tile_orders[lru];
            if (old.cost) hipFree(old.cost);
            if (old.order) hipFree(old.order);
            if (old.factor) hipFree(old.factor);
            if (old.ev_rec) hipEventDestroy(old.ev_rec);
            if (old.ev_done) hipEventDestroy(old.ev_done);
            ctx->tile_orders.erase(ctx->tile_orders.begin() + (long)lru);
        }
        rtx_ctx::TileOrder fresh;
        fresh.stream = stream;
        std::memcpy(fresh.key, key, sizeof fresh.key);
        // cost: the estimates by tile, then the workgroups' start and end times by dispatch position; two orders: the one in
        // use and the one a balancing pass writes
        if (hipMalloc((void**)&fresh.cost, 3 * (size_t)n_tiles * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc((void**)&fresh.order, 2 * (size_t)n_tiles * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc((void**)&fresh.factor, (size_t)n_tiles * sizeof(float)) != hipSuccess ||
            hipEventCreateWithFlags(&fresh.ev_rec, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&fresh.ev_done, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (fresh.cost) hipFree(fresh.cost);
            if (fresh.order) hipFree(fresh.order);
            if (fresh.factor) hipFree(fresh.factor);
            if (fresh.ev_rec) hipEventDestroy(fresh.ev_rec);
            if (fresh.ev_done) hipEventDestroy(fresh.ev_done);
            return nullptr; // no memory for it: render in frame order
        }
        fresh.cap = n_tiles;
        fresh.id = ctx->next_order_id++;
        ctx->tile_orders.push_back(fresh);
        to = &ctx->tile_orders.back();
    }
    to->last_use = ++ctx->order_clock;
    return to;
}

// ---- the pieces of rtx_render_rows

struct RenderCall { // what the arguments of a call come to
    uint64_t W = 0, H = 0, row0 = 0, rows = 0, out_row_base = 0;
    void* d_out = nullptr;
    bool own = false, compact = false, values = false;
};

int validate_render_call(rtx_ctx* ctx, const rtx_params* p, int mode, size_t row0, size_t rows, void* d_out, size_t out_row_base, unsigned flags, RenderCall* c)
{
    if (mode < RTX_BIT_ASCII || mode > RTX_SDL) {
        return rtx_fail(ctx, RTX_ERR_INVALID_MODE, "invalid rendering mode"); // RayTracing.cu:863-865
    }
    const uint64_t W = p->x, H = p->y;
    if (W == 0 || H == 0 || W >= (1ull << 31) || H >= (1ull << 31)) {
        return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "params.x / params.y must be in [1, 2^31)");
    }
    if (row0 > H) row0 = H;
    if (rows > H - row0) rows = H - row0;
    const bool own = (d_out == nullptr);
    if (own) {
        if (out_row_base != 0) return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "out_row_base must be 0 with the context's buffer");
        if (20 * W * H > ctx->capacity) return rtx_fail(ctx, RTX_ERR_TOO_LARGE, "frame larger than the context was created for");
        d_out = ctx->d_frame;
    } else if (row0 < out_row_base) {
        return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "row0 is before out_row_base");
    }
    if (((uintptr_t)d_out & 3u) != 0) return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "output buffer must be 4-byte aligned");
    const bool values = (flags & RTX_RENDER_VALUES) != 0;
    const bool compact = values || (flags & RTX_RENDER_COMPACT) != 0;
    if (compact && (own || mode == RTX_SDL || (flags & RTX_RENDER_ZERO_TAIL) || (values && (flags & RTX_RENDER_COMPACT)))) {
        return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "RTX_RENDER_COMPACT / RTX_RENDER_VALUES need a caller buffer, a character mode, no RTX_RENDER_ZERO_TAIL, and exclude each other");
    }
    if (values && ((uintptr_t)d_out & 15u) != 0) return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "RTX_RENDER_VALUES: output buffer must be 16-byte aligned");
    if ((flags & RTX_RENDER_ZERO_TAIL) && out_row_base != 0) {
        return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "RTX_RENDER_ZERO_TAIL needs a full-frame buffer (out_row_base 0)");
    }
    c->W = W;
    c->H = H;
    c->row0 = row0;
    c->rows = rows;
    c->out_row_base = out_row_base;
    c->d_out = d_out;
    c->own = own;
    c->compact = compact;
    c->values = values;
    return RTX_OK;
}

// Camera, frame geometry, scene arrays and output of a launch (everything but the plan).
void fill_frame_args(const rtx_ctx* ctx, const rtx_params* p, const RenderCall& c, KArgs& a)
{
    std::memset(&a, 0, sizeof a);
    std::memcpy(a.m, p->inv_v, 12 * sizeof(float));
    a.ox = p->cam_pos[0];
    a.oy = p->cam_pos[1];
    a.oz = p->cam_pos[2];
    a.e1 = p->element1;
    a.e2 = p->element2;
    a.far = p->cam_far;
    a.fW = (float)c.W; // (float)(params->x), RayTracing.cu:17
    a.fH = (float)c.H;
    const rtxplan::EdgeBasis eb = rtxplan::edge_basis(p->inv_v, p->element1, p->element2, (uint64_t)c.W, (uint64_t)c.H);
    for (int k = 0; k < 3; k++) {
        a.edge_up_p[k] = eb.up_p[k];
        a.edge_up_q[k] = eb.up_q[k];
        a.edge_right_p[k] = eb.right_p[k];
        a.edge_right_q[k] = eb.right_q[k];
        a.edge_fwd[k] = eb.fwd[k];
    }
    a.edge_pp = eb.pp;
    a.edge_qrqr = eb.qrqr;
    a.edge_qcqc = eb.qcqc;
    a.W = (uint32_t)c.W;
    a.H = (uint32_t)c.H;
    a.row0 = (uint32_t)c.row0;
    a.row_end = (uint32_t)(c.row0 + c.rows);
    a.out_row_base = (uint32_t)c.out_row_base;
    a.ns = ctx->ns;
    a.np = ctx->np;
    a.sph_geom = a.sph_scene_geom = (const float4*)ctx->d_sph_geom.p;
    a.sph_od = (const float4*)ctx->d_sph_od.p;
    if (ctx->opt_sorted_store != 0 && ctx->sorted_gen == ctx->scene_gen && ctx->d_sorted_geom.p != nullptr) {
        // the trace kernels read the direction-sorted copies, by position
        a.sph_geom = (const float4*)ctx->d_sorted_geom.p;
        a.sph_od = (const float4*)ctx->d_sorted_od.p;
        a.sph_sorted_idx = (const uint32_t*)ctx->d_sorted_idx.p;
        a.sph_pos_of = (const uint32_t*)ctx->d_pos_of.p;
    }
    a.pl_a = (const float4*)ctx->d_pl_a.p;
    a.pl_b = (const float4*)ctx->d_pl_b.p;
    a.pl_od = (const float4*)ctx->d_pl_od.p;
    a.grey = ctx->d_grey;
    a.out = (uint8_t*)c.d_out;
    a.compact = c.values ? 2u : (c.compact ? 1u : 0u);
    a.cells_x = a.cells_y = 1;
    RTX_X_FILL_KARGS(a); // (experiment build only)
}

// View-density feedback (rtxplan::ViewDensity): does this launch take part?  A scene that is sparse by its numbers can be
// locally dense from where the camera stands; the launches report their longest candidate list and the plan follows.  Not while
// recording (a graph keeps one plan); scenes that are dense by their numbers always run the dense plan; explicit sub-tile
// counts are taken as they are.  And only where one workgroup can hold a launch up: sparse plans whose grid is one dispatch
// round (config 2 at 1080p, the slabs of a sharded frame).  A grid of several rounds refills its CUs as workgroups end; there
// the dense plan bought nothing alone and cost a turning camera 20 % with frames in flight (config 3: 82.6 -> 101.7 us).
bool view_adaptation_applies(const rtx_ctx* ctx, const rtxplan::TileRequest& q, bool capturing)
{
    if (!q.cull || capturing || ctx->opt_view_adapt == 0 || ctx->opt_subtiles || ctx->ns < 256u ||
        !((double)ctx->ns / ((double)q.W * (double)q.H) < rtxplan::kDenseScene)) {
        return false;
    }
    rtxplan::TileRequest sparse_q = q;
    sparse_q.view_dense = false;
    const rtxplan::TileShape sparse = rtxplan::plan_tiles(sparse_q);
    return (uint64_t)sparse.grid_x * sparse.grid_y <= rtxplan::resident_slots(ctx->n_cu) || ctx->opt_view_adapt > 0;
}

// ... the launch's side of it: where its workgroups report to.
int density_feedback_args(rtx_ctx* ctx, hipStream_t stream, bool view_dense, KArgs& a)
{
    if (!ctx->d_longest) {
        void* h = nullptr;
        if (hipMalloc((void**)&ctx->d_longest, 3 * sizeof(uint32_t)) != hipSuccess || hipHostMalloc(&h, sizeof(uint32_t), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_longest, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (ctx->d_longest) hipFree(ctx->d_longest);
            if (h) hipHostFree(h);
            ctx->d_longest = nullptr;
            ctx->opt_view_adapt = 0; // no feedback: the plan follows the scene's numbers alone
            return RTX_OK;
        }
        ctx->h_longest = (volatile uint32_t*)h;
        *ctx->h_longest = 0u;
        RTX_HIP(ctx, hipMemsetAsync(ctx->d_longest, 0, 3 * sizeof(uint32_t), stream));
    }
    a.longest_list = ctx->d_longest;
    a.longest_from = view_dense ? rtxplan::ViewDensity::kLightDense : rtxplan::ViewDensity::kReportSparse;
    a.longest_slot = ctx->longest_epoch % 3u;
    return RTX_OK;
}

// ... and after the launch: an epoch is 8 launches; when it ends its word is copied to the host and a copy that has landed is one
// observation.  Three words in rotation: epoch e fills word e mod 3 and zeroes word (e + 1) mod 3.  Epoch e + 1 begins when the copy
// of word e mod 3 is QUEUED (on the stream of e's last launch), epoch e + 2 only once that copy has LANDED -- so the word a launch
// zeroes (last filled two epochs ago) is never one whose copy is still to run, whatever streams the launches are on.
int density_feedback_collect(rtx_ctx* ctx, hipStream_t stream)
{
    if (!ctx->d_longest || (++ctx->longest_launches & 7u) != 0u) return RTX_OK;
    bool landed = !ctx->longest_copy_pending;
    if (ctx->longest_copy_pending) {
        if (hipEventQuery(ctx->ev_longest) == hipSuccess) landed = true;
        else (void)hipGetLastError();
        if (landed) {
            if (ctx->view_density.observe(*ctx->h_longest)) ctx->stat_density_switches++;
            ctx->longest_copy_pending = false;
        }
    }
    if (landed) {
        RTX_HIP(ctx, hipMemcpyAsync((void*)ctx->h_longest, ctx->d_longest + (ctx->longest_epoch % 3u), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        RTX_HIP(ctx, hipEventRecord(ctx->ev_longest, stream));
        ctx->longest_copy_pending = true;
        ctx->longest_epoch++;
    }
    return RTX_OK;
}

// Dispatch order of this launch (culling kernels; grids of at least two workgroups per CU): which order it reads, whether it
// leaves estimates.  No launch ever reads an order that is being written: rtx_order_tiles is queued on `stream` itself, and
// rtx_balance_tiles, which runs on the context's side stream, writes the half of the order buffer that no launch uses until
// `stream` has waited for the pass.
// batch_n > 0: a batched launch (rtx_trace_batch) of that many frames.  Its workgroups run tile position b / batch_n of frame
// b % batch_n, so an order that is simply heaviest first gives every CU (blocks c, c + n_cu, ...) a stratified sample of the
// costs: rtx_order_tiles without the dealing, from the estimates one frame of the batch leaves; a set of its own per batch size.
int dispatch_order_args(rtx_ctx* ctx, hipStream_t stream, const rtx_params* p, const RenderCall& c, const rtxplan::TileShape& shape, bool capturing,
                        const uint32_t* static_order, KArgs& a, rtx_ctx::TileOrder** to_out, rtxplan::DispatchOrder::Decision* od, uint32_t batch_n = 0)
{
    *to_out = nullptr;
    const uint64_t nt = (uint64_t)shape.grid_x * shape.grid_y;
    if (ctx->opt_tile_order != 0 && ctx->n_cu > 0) {
        // auto (-1): only grids that fit one dispatch round -- there the order balances the CUs; with several rounds the
        // hardware's refill does that, and ordering by cost would only separate tiles that share 128-byte lines
        const uint64_t slots = rtxplan::resident_slots(ctx->n_cu);
        // ... except when the caller renders on ONE stream (of the last 16 two-level launches): nothing then overlaps the
        // end of a launch, and what ends it is the lifetime of the workgroups dispatched last -- heaviest first, they are the
        // light ones (config 5 alone 44.6 -> 41.4 us, config 3 103.1 -> 97.2; with frames in flight on several streams the same
        // order costs 1-2 %, and from seven rounds on -- 8K -- it costs more than the tail is worth)
        const bool lone_stream = batch_n == 0 && ctx->opt_tile_order == -1 && ctx->render_streams_seen == 1 && nt > slots && nt <= 6 * slots;
        const int64_t opt_eff = (lone_stream || (batch_n != 0 && ctx->opt_tile_order < 0)) ? 16 : ctx->opt_tile_order;
        const bool wanted = ctx->opt_tile_order > 0 || nt <= slots || lone_stream || batch_n != 0;
        const bool one_round = batch_n == 0 && nt <= slots && nt <= 2048u && ctx->n_cu <= 1024 && shape.tiles256 >= 2000u; // (from half a megapixel: a pass costs 10 us)
        if (wanted && shape.grid_x <= 0xffffu && shape.grid_y <= 0xffffu && nt * (batch_n ? batch_n : 1u) >= 2u * (uint64_t)ctx->n_cu && nt <= (1u << 22)) {
            const uint64_t key[3] = {(c.W << 32) | c.H, (c.row0 << 32) | c.rows,
                                     ((uint64_t)shape.lw << 40) | ((uint64_t)shape.lnx << 32) | ((uint64_t)shape.nsub << 16) | ((uint64_t)batch_n << 4) | 1u};
            rtx_ctx::TileOrder* to = nullptr;
            if (capturing) {
                // A recorded launch keeps the pointers it was recorded with: it may run under the order this grid has
                // converged to on this stream, and from then on that order is frozen (no pass writes either half again);
                // without one it runs in frame order.  Nothing is derived during a capture.
                for (auto& e : ctx->tile_orders) {
                    if (e.stream == stream && std::memcmp(e.key, key, sizeof e.key) == 0 && e.plan.have_order() && !e.plan.pass_pending()) {
                        e.frozen = true;
                        bool counted = false;
                        for (uint64_t id : ctx->capture_frozen) counted = counted || id == e.id;
                        if (!counted) {
                            ctx->capture_frozen.push_back(e.id);
                            e.frozen_refs++;
                        }
                        a.tile_order = e.order + (size_t)e.plan.current_half() * e.cap;
                    }
                }
            } else if (one_round && !side_stream(ctx)) {
                // no side stream for the passes: frame order
            } else if ((to = tile_order_set(ctx, stream, key, (uint32_t)nt)) != nullptr) {
                if (to->frozen) {
                    a.tile_order = to->order + (size_t)to->plan.current_half() * to->cap; // as recorded; nothing new is derived
                } else {
                    *od = to->plan.next(rtxplan::view_of(p->inv_v, p->cam_pos), ctx->scene_drift, one_round, opt_eff);
                    if (od->switch_order) RTX_HIP(ctx, hipStreamWaitEvent(stream, to->ev_done, 0)); // the pass: by now long done
                    a.tile_cost = od->leave_estimates ? to->cost : nullptr;
                    a.tile_order = od->use_order ? to->order + (size_t)od->half * to->cap : nullptr;
                    to->base = static_order;
                    to->batch = batch_n != 0;
                    *to_out = to;
                }
            }
        }
    }
    if (a.tile_order == nullptr && !od->leave_estimates) a.tile_order = static_order; // (two-level grids of several rounds: by XCD)
    return RTX_OK;
}

// ... and after the launch: the pass that derives the next order from what the launch left.
int dispatch_order_derive(rtx_ctx* ctx, hipStream_t stream, const rtxplan::TileShape& shape, const KArgs& a, rtx_ctx::TileOrder* to,
                          const rtxplan::DispatchOrder::Decision& od)
{
    const uint32_t nt = shape.grid_x * shape.grid_y;
    if (od.sort_now && od.balance) {
        // every workgroup resident at once: deal the tiles so that the groups of workgroups that share a CU carry equal
        // work, correcting the estimates by how long each group took in the launch just queued (rtx_balance_tiles).
        // The pass is told the order that launch REALLY ran under (none, if the one in hand was stale).
        uint32_t* next_order = to->order + (size_t)(od.half ^ 1) * to->cap;
        hipError_t e = hipEventRecord(to->ev_rec, stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->aux_stream, to->ev_rec, 0);
        if (e != hipSuccess) return rtx_hip_fail(ctx, e, "tile order events");
        const int oe = rtx_k_launch_balance_tiles(to->cost, nt, shape.grid_x, (uint32_t)ctx->n_cu, od.prev_is_order ? a.tile_order : nullptr, to->factor,
                                                  od.have_factor ? 1 : 0, 1, next_order, ctx->aux_stream);
        if (oe != 0) return rtx_hip_fail(ctx, (hipError_t)oe, "tile order launch");
        e = hipEventRecord(to->ev_done, ctx->aux_stream);
        if (e != hipSuccess) return rtx_hip_fail(ctx, e, "tile order events");
        ctx->stat_order_passes++;
    } else if (od.sort_now) {
        // the estimates do not depend on the order they were produced under, so one pass settles a static view;
        // the second pass and the periodic ones follow a moving camera / scene
        uint32_t* cur_order = to->order + (size_t)od.half * to->cap;
        // (a batched launch interleaves its frames tile by tile: plain heaviest first, nothing dealt)
        const int oe = rtx_k_launch_order_tiles(to->cost, nt, shape.grid_x, (uint32_t)ctx->n_cu, to->batch ? 0u : rtxplan::kResidentPerCU * (uint32_t)ctx->n_cu, to->base,
                                                cur_order, stream);
        if (oe != 0) return rtx_hip_fail(ctx, (hipError_t)oe, "tile order launch");
        ctx->stat_order_passes++;
    }
    to->plan.launched(od);
    return RTX_OK;
}

bool same_projection(const rtx_params& x, const rtx_params& y)
{
    return x.x == y.x && x.y == y.y && float_to_bits(x.element1) == float_to_bits(y.element1) && float_to_bits(x.element2) == float_to_bits(y.element2) &&
           float_to_bits(x.cam_far) == float_to_bits(y.cam_far);
}

bool same_camera(const rtx_params& x, const rtx_params& y)
{
    return std::memcmp(x.inv_v, y.inv_v, sizeof x.inv_v) == 0 && std::memcmp(x.cam_pos, y.cam_pos, sizeof x.cam_pos) == 0;
}

// Rows [row0, row0 + rows) of n frames with ONE launch on `stream` (rtx_trace_batch: KBatch, rtx_kernels.h) -- a rank's slab
// of every frame of a round, which frame by frame is too small a grid to fill the GPU (a 135-row slab of 1080p: 255
// workgroups on 256 CUs).  *done = false and RTX_OK: this batch is not one the batched kernel takes (a plan with per-wave
// refinement or two-level culling, frames that differ in more than their camera, an output form other than records / compact
// words); the caller then queues the slabs one by one.
int render_batch(rtx_ctx* ctx, size_t n, const rtx_params* params, int mode, size_t row0, size_t rows, void* const* d_outs, size_t out_row_base,
                 hipStream_t stream, unsigned flags, bool* done)
{
    *done = false;
    if (n < 2 || n > (size_t)kMaxBatch || mode < RTX_BIT_ASCII || mode >= RTX_SDL || (flags & ~(unsigned)RTX_RENDER_COMPACT) != 0u) return RTX_OK;
    if (!uses_culling_kernel(ctx) || uses_two_level(ctx, false) || ctx->opt_refine == 1) return RTX_OK;
    for (size_t i = 0; i < n; i++) {
        if (!d_outs[i] || !same_projection(params[0], params[i])) return RTX_OK;
    }
    RenderCall c;
    int rc;
    for (size_t i = 0; i < n; i++) {
        if ((rc = validate_render_call(ctx, &params[i], mode, row0, rows, d_outs[i], out_row_base, flags, &c)) != RTX_OK) return rc;
    }
    if (c.rows == 0) {
        *done = true;
        return RTX_OK;
    }
    bool capturing = false;
    if ((rc = check_recordable(ctx, stream, &capturing)) != RTX_OK) return rc;
    if (!capturing) {
        if ((rc = rtx_sync_scene(ctx)) != RTX_OK) return rc;
        if (ctx->opt_sorted_store != 0 && ctx->sorted_gen != ctx->scene_gen && ctx->ns >= 256u) {
            if ((rc = rtx_sort_scene(ctx, params[0].cam_pos)) != RTX_OK) return rc;
            ctx->lists_gen++;
            ctx->cell_policy.invalidate();
        }
    }
    KArgs a;
    fill_frame_args(ctx, &params[0], c, a);
    const double aspect = pixel_aspect(&params[0]);
    rtxplan::TileRequest q;
    q.W = c.W;
    q.H = c.H;
    q.rows = c.rows;
    q.ns = ctx->ns;
    q.aspect = aspect;
    q.n_cu = ctx->n_cu;
    q.cull = true;
    q.opt_subtiles = (int)ctx->opt_subtiles;
    q.opt_tile_log2w = (int)ctx->opt_tile_log2w;
    q.opt_refine = (int)ctx->opt_refine;
    // the plan of the whole batch: as many 256-pixel tiles as all its frames have, so that the sub-tile count is chosen for
    // the grid the GPU really sees (one dispatch round where that is possible)
    if (ctx->opt_subtiles == 0) q.rows = c.rows * n;
    rtxplan::TileShape shape = rtxplan::plan_tiles(q);
    if (ctx->opt_subtiles == 0) {
        q.opt_subtiles = (int)shape.nsub;
        q.opt_tile_log2w = (int)shape.lw;
        q.rows = c.rows;
        shape = rtxplan::plan_tiles(q); // the same sub-tiles over one frame's rows: that frame's grid
    }
    if (shape.refine) return RTX_OK;
    a.tile_log2w = shape.lw;
    a.sub_log2nx = shape.lnx;
    a.nsub = shape.nsub;
    a.refine = 0u;
    a.batch_n = (uint32_t)n;
    a.batch_gx = shape.grid_x;
    a.batch_gy = shape.grid_y;
    bool identical = true;
    for (size_t i = 1; i < n; i++) identical = identical && same_camera(params[0], params[i]);
    const uint32_t* static_order = nullptr;
    if (identical && !capturing && ctx->opt_two_level < 0 && ctx->opt_cell_reuse != 0 && ctx->ns >= 256u &&
        (uint64_t)shape.grid_x * shape.grid_y <= rtxplan::resident_slots(ctx->n_cu)) {
        // exact cell lists while camera and scene rest, as for a plain launch; every frame of the batch is that one view
        if ((rc = prepare_cells(ctx, &params[0], stream, shape, c.row0, c.rows, aspect, a, &static_order, true)) != RTX_OK) return rc;
    }
    rtx_ctx::TileOrder* to = nullptr;
    rtxplan::DispatchOrder::Decision od;
    if ((rc = dispatch_order_args(ctx, stream, &params[0], c, shape, capturing, static_order, a, &to, &od, (uint32_t)n)) != RTX_OK) return rc;
    KBatch kb;
    std::memset(&kb, 0, sizeof kb);
    for (size_t i = 0; i < n; i++) {
        KFrame& f = kb.f[i];
        const rtx_params& p = params[i];
        std::memcpy(f.m, p.inv_v, 12 * sizeof(float));
        f.ox = p.cam_pos[0];
        f.oy = p.cam_pos[1];
        f.oz = p.cam_pos[2];
        const rtxplan::EdgeBasis eb = rtxplan::edge_basis(p.inv_v, p.element1, p.element2, (uint64_t)c.W, (uint64_t)c.H);
        for (int k = 0; k < 3; k++) {
            f.edge_up_p[k] = eb.up_p[k];
            f.edge_up_q[k] = eb.up_q[k];
            f.edge_right_p[k] = eb.right_p[k];
            f.edge_right_q[k] = eb.right_q[k];
            f.edge_fwd[k] = eb.fwd[k];
        }
        f.edge_pp = eb.pp;
        f.edge_qrqr = eb.qrqr;
        f.edge_qcqc = eb.qcqc;
        f.out = (uint8_t*)d_outs[i];
    }
    int herr = 0;
    const char* name = rtx_k_launch_trace_batch(&a, &kb, mode, stream, &herr);
    if (!name) return rtx_fail(ctx, RTX_ERR_INVALID_MODE, "invalid rendering mode or tile configuration");
    if (herr != 0) return rtx_hip_fail(ctx, (hipError_t)herr, "batched trace kernel launch");
    ctx->last_kernel = name;
    ctx->stat_batched_launches++;
    if (to && (rc = dispatch_order_derive(ctx, stream, shape, a, to, od)) != RTX_OK) return rc;
    *done = true;
    return RTX_OK;
}

} // namespace

int rtx_frame_zero_semantics(rtx_ctx* ctx, int mode, uint64_t W, uint64_t H, unsigned flags)
{
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    return zero_fill_semantics(ctx, mode, W, H, ctx->d_frame, true, false, flags, ctx->stream);
}

int rtx_render_rows(rtx_ctx* ctx, const rtx_params* p, int mode, size_t row0, size_t rows, void* d_out,
                    size_t out_row_base, void* stream_v, unsigned flags)
{
    if (!ctx || !p) return RTX_ERR_INVALID_ARGUMENT;
    RenderCall c;
    int rc = validate_render_call(ctx, p, mode, row0, rows, d_out, out_row_base, flags, &c);
    if (rc != RTX_OK) return rc;
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = stream_v ? (hipStream_t)stream_v : ctx->stream;
    bool capturing = false;
    if ((rc = check_recordable(ctx, stream, &capturing)) != RTX_OK) return rc;
    if (capturing && c.own) {
        // the zero-fill of the context's buffer is decided from what earlier launches left in it (dirty_hi): a recorded
        // launch would replay that decision whatever the replays in between have written
        return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "graph capture: recorded launches need a caller buffer (d_out), not the context's own");
    }
    if (!capturing) {
        if ((rc = rtx_sync_scene(ctx)) != RTX_OK) return rc;
        // the direction-sorted copies the trace kernels read: rebuilt at the first launch after a scene edit, around this camera
        if (ctx->opt_sorted_store != 0 && ctx->sorted_gen != ctx->scene_gen && ctx->ns >= 256u) {
            if ((rc = rtx_sort_scene(ctx, p->cam_pos)) != RTX_OK) return rc;
            ctx->lists_gen++; // (cell lists hold positions in those arrays)
            ctx->cell_policy.invalidate();
        }
    }
    if ((rc = zero_fill_semantics(ctx, mode, c.W, c.H, c.d_out, c.own, c.compact, flags, stream)) != RTX_OK) return rc;
    if (c.rows == 0) return RTX_OK;

    KArgs a;
    fill_frame_args(ctx, p, c, a);

    // ---- plan (rtx_plan.hpp): tile shape -- for a dense scene, or for a view that the launches before this one found locally dense
    const int cull = uses_culling_kernel(ctx) ? 1 : 0;
    const double aspect = pixel_aspect(p);
    rtxplan::TileRequest q;
    q.W = c.W;
    q.H = c.H;
    q.rows = c.rows;
    q.ns = ctx->ns;
    q.aspect = aspect;
    q.n_cu = ctx->n_cu;
    q.cull = cull != 0;
    q.opt_subtiles = (int)ctx->opt_subtiles;
    q.opt_tile_log2w = (int)ctx->opt_tile_log2w;
    q.opt_refine = (int)ctx->opt_refine;
    const bool adapt = view_adaptation_applies(ctx, q, capturing);
    q.view_dense = adapt && ctx->view_density.dense();
    q.in_flight = ctx->render_streams_seen >= 2; // (of the last 16 two-level launches: prepare_cells keeps count)
    const rtxplan::TileShape shape = rtxplan::plan_tiles(q);
    a.tile_log2w = shape.lw;
    a.sub_log2nx = shape.lnx;
    a.nsub = shape.nsub;
    a.refine = shape.refine ? 1u : 0u;

    // ---- cell lists (never while capturing: check_recordable)
    const uint32_t* static_order = nullptr;
    if (uses_two_level(ctx, q.view_dense)) {
        if ((rc = prepare_cells(ctx, p, stream, shape, c.row0, c.rows, aspect, a, &static_order, false)) != RTX_OK) return rc;
    } else if (cull && !capturing && ctx->opt_two_level < 0 && ctx->opt_cell_reuse != 0 && ctx->ns >= 256u &&
               (uint64_t)shape.grid_x * shape.grid_y <= rtxplan::resident_slots(ctx->n_cu)) {
        // under 2048 spheres a pre-pass per frame does not pay, lists that cost nothing do: exact ones, while everything rests.
        // (Grids of one dispatch round only: there every workgroup's staging sits on the launch's critical path -- config 2
        // 25.9 -> 24.7 us alone.  A grid of several rounds with 8 sub-tiles per workgroup has the staging amortised and hidden,
        // and the list's two dependent loads cost more than they save: config 4 237 -> 260 us.)
        if ((rc = prepare_cells(ctx, p, stream, shape, c.row0, c.rows, aspect, a, &static_order, true)) != RTX_OK) return rc;
    }

    // ---- dispatch order, view-density feedback, launch
    rtx_ctx::TileOrder* to = nullptr;
    rtxplan::DispatchOrder::Decision od;
    if (cull && (rc = dispatch_order_args(ctx, stream, p, c, shape, capturing, static_order, a, &to, &od)) != RTX_OK) return rc;
    if (adapt && (rc = density_feedback_args(ctx, stream, q.view_dense, a)) != RTX_OK) return rc;
    int herr = 0;
    const char* name = rtx_k_launch_trace(&a, mode, cull, stream, &herr);
    if (!name) return rtx_fail(ctx, RTX_ERR_INVALID_MODE, "invalid rendering mode or tile configuration");
    if (herr != 0) return rtx_hip_fail(ctx, (hipError_t)herr, "trace kernel launch");
    ctx->last_kernel = name;
    if (adapt && (rc = density_feedback_collect(ctx, stream)) != RTX_OK) return rc;
    if (to && (rc = dispatch_order_derive(ctx, stream, shape, a, to, od)) != RTX_OK) return rc;
    return RTX_OK;
}

int rtx_render(rtx_ctx* ctx, const rtx_params* params, int mode)
{
    if (!ctx || !params) return RTX_ERR_INVALID_ARGUMENT;
    if (ctx->group) return rtxgroup::render_frame(ctx, params, mode, nullptr, RTX_RENDER_DEFAULT); // sharded over the group's devices
    return rtx_render_rows(ctx, params, mode, 0, params->y, nullptr, 0, nullptr, RTX_RENDER_DEFAULT);
}

int rtx_submit_frames(rtx_ctx* ctx, size_t n, const rtx_params* params, int mode, void* const* d_outs, void* const* streams)
{
    if (!ctx || (n && (!params || !d_outs || !streams))) return RTX_ERR_INVALID_ARGUMENT;
    if (ctx->group && n) return rtxgroup::render_frames(ctx, n, params, mode, d_outs, streams); // sharded, a chunk of frames per rank and call
    for (size_t i = 0; i < n; i++) {
        if (!d_outs[i]) return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "rtx_submit_frames: null frame buffer");
        const int rc = rtx_render_rows(ctx, &params[i], mode, 0, (size_t)params[i].y, d_outs[i], 0, streams[i], RTX_RENDER_DEFAULT);
        if (rc != RTX_OK) return rc;
    }
    return RTX_OK;
}

int rtx_submit_slabs(rtx_ctx* ctx, size_t n, const rtx_params* params, int mode, size_t row0, size_t rows,
                     void* const* d_outs, size_t out_row_base, void* const* streams, void* after, unsigned flags)
{
    if (!ctx || (n && (!params || !d_outs || !streams))) return RTX_ERR_INVALID_ARGUMENT;
    if (n == 0) return RTX_OK;
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t join = static_cast<hipStream_t>(after);
    if (join) {
        // refuse what cannot be recorded before the fork below pulls the render streams into a capture
        bool capturing = false;
        const int rc = check_recordable(ctx, join, &capturing);
        if (rc != RTX_OK) return rc;
    }
    // distinct render streams of this call, each with its event
    std::vector<rtx_ctx::JoinEvent*> used;
    ctx->join_events.reserve(64); // pointers into the vector stay valid
    if (join) {
        if (!ctx->ev_fork) RTX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        RTX_HIP(ctx, hipEventRecord(ctx->ev_fork, join));
        for (size_t i = 0; i < n; i++) {
            hipStream_t s = streams[i] ? static_cast<hipStream_t>(streams[i]) : ctx->stream;
            if (s == join) continue; // already ordered
            rtx_ctx::JoinEvent* je = nullptr;
            for (auto& e : ctx->join_events) {
                if (e.stream == s) je = &e;
            }
            if (!je) {
                if (ctx->join_events.size() >= 64) return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "rtx_submit_slabs: more than 64 distinct streams");
                rtx_ctx::JoinEvent fresh;
                fresh.stream = s;
                RTX_HIP(ctx, hipEventCreateWithFlags(&fresh.ev, hipEventDisableTiming));
                ctx->join_events.push_back(fresh);
                je = &ctx->join_events.back();
            }
            bool seen = false;
            for (auto* u : used) seen = seen || (u == je);
            if (!seen) {
                RTX_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_fork, 0));
                used.push_back(je);
            }
        }
    }
    for (size_t i = 0; i < n; i++) {
        if (!d_outs[i]) return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "rtx_submit_slabs: null slab buffer");
    }
    for (size_t i = 0; i < n;) {
        // consecutive slabs on one stream: ONE launch renders them all (up to kMaxBatch at a time), when the batched kernel
        // takes them (render_batch); else, and for a slab alone on its stream, a launch each
        size_t run = 1;
        while (i + run < n && run < (size_t)kMaxBatch && streams[i + run] == streams[i]) run++;
        bool batched = false;
        if (run >= 2 && ctx->opt_batch != 0) {
            hipStream_t s = streams[i] ? static_cast<hipStream_t>(streams[i]) : ctx->stream;
            const int rc = render_batch(ctx, run, &params[i], mode, row0, rows, &d_outs[i], out_row_base, s, flags, &batched);
            if (rc != RTX_OK) return rc;
        }
        if (!batched) {
            for (size_t k = i; k < i + run; k++) {
                const int rc = rtx_render_rows(ctx, &params[k], mode, row0, rows, d_outs[k], out_row_base, streams[k], flags);
                if (rc != RTX_OK) return rc;
            }
        }
        i += run;
    }
    for (auto* u : used) {
        RTX_HIP(ctx, hipEventRecord(u->ev, u->stream));
        RTX_HIP(ctx, hipStreamWaitEvent(join, u->ev, 0));
    }
    return RTX_OK;
}

int rtx_expand(rtx_ctx* ctx, int mode, const void* d_compact, void* d_out, const rtx_segment* segments, size_t n_segments, void* stream_v)
{
    if (!ctx || (n_segments && (!d_compact || !d_out || !segments))) return RTX_ERR_INVALID_ARGUMENT;
    if (mode < RTX_BIT_ASCII || mode >= RTX_SDL) return rtx_fail(ctx, RTX_ERR_INVALID_MODE, "rtx_expand: not a character mode");
    if ((((uintptr_t)d_compact | (uintptr_t)d_out) & 3u) != 0) return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "rtx_expand: buffers must be 4-byte aligned");
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = stream_v ? (hipStream_t)stream_v : ctx->stream;
    const uint64_t S = mode >= RTX_RGB_ASCII ? 20u : 12u;
    size_t done = 0;
    while (done < n_segments) {
        ExpandArgs e;
        std::memset(&e, 0, sizeof(e));
        e.src = (const uint32_t*)d_compact;
        e.dst = (uint8_t*)d_out;
        e.aligned16 = ((uintptr_t)d_out & 15u) == 0 ? 1u : 0u;
        uint64_t blocks = 0;
        while (done < n_segments && e.nseg < (uint32_t)kMaxExpandSeg) {
            const rtx_segment& g = segments[done];
            if (g.n_pixels >= (1ull << 32)) return rtx_fail(ctx, RTX_ERR_TOO_LARGE, "rtx_expand: segment of 2^32 pixels or more");
            const uint64_t nb = (g.n_pixels + (uint64_t)kExpandPixels - 1u) / (uint64_t)kExpandPixels;
            if (blocks + nb >= (1ull << 31)) {
                if (e.nseg == 0) return rtx_fail(ctx, RTX_ERR_TOO_LARGE, "rtx_expand: segment too large for one launch");
                break;
            }
            done++;
            if (g.n_pixels == 0) continue;
            e.first_block[e.nseg] = (uint32_t)blocks;
            e.npix[e.nseg] = (uint32_t)g.n_pixels;
            e.src_px[e.nseg] = g.src_pixel;
            e.dst_px[e.nseg] = g.dst_pixel;
            if ((g.dst_pixel * S) & 15u) e.aligned16 = 0u;
            blocks += nb;
            e.nseg++;
            e.first_block[e.nseg] = (uint32_t)blocks;
        }
        if (e.nseg == 0) continue;
        const int he = rtx_k_launch_expand(&e, mode, (unsigned)blocks, stream);
        if (he != 0) return rtx_hip_fail(ctx, (hipError_t)he, "expand launch");
    }
    return RTX_OK;
}

int rtx_graph_begin(rtx_ctx* ctx, void* stream_v)
{
    if (!ctx) return RTX_ERR_INVALID_ARGUMENT;
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = stream_v ? (hipStream_t)stream_v : ctx->stream;
    // relaxed: other threads of the process (a collective library's proxies) may go on calling the runtime
    RTX_HIP(ctx, hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed));
    ctx->capture_frozen.clear();
    return RTX_OK;
}

namespace {

// The graph that held these dispatch-order sets frozen is gone (or was never made): a set no live graph reads is balanced again.
void release_frozen_orders(rtx_ctx* ctx, const std::vector<uint64_t>& ids)
{
    for (uint64_t id : ids) {
        for (auto& t : ctx->tile_orders) {
            if (t.id == id && t.frozen_refs > 0 && --t.frozen_refs == 0) t.frozen = false;
        }
    }
}

} // namespace

// What rtx_graph_end hands out: the executable graph and the scene generation it was recorded on.  A recorded launch
// keeps raw pointers into the scene arrays and the object counts of that moment; rtx_scene_add_* / rtx_scene_clear
// reallocate the arrays or change the counts, after which a replay would read freed memory or a stale count.
struct rtx_graph_handle {
    hipGraphExec_t exec = nullptr;
    uint64_t scene_gen = 0;
    std::vector<uint64_t> frozen; // dispatch-order sets its launches read (released by rtx_graph_destroy)
};

int rtx_graph_end(rtx_ctx* ctx, void* stream_v, void** graph_out)
{
    if (!ctx || !graph_out) return RTX_ERR_INVALID_ARGUMENT;
    *graph_out = nullptr;
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = stream_v ? (hipStream_t)stream_v : ctx->stream;
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(stream, &graph);
    std::vector<uint64_t> frozen;
    frozen.swap(ctx->capture_frozen);
    if (e != hipSuccess || !graph) {
        if (graph) hipGraphDestroy(graph);
        release_frozen_orders(ctx, frozen);
        return rtx_hip_fail(ctx, e != hipSuccess ? e : hipErrorUnknown, "hipStreamEndCapture");
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) {
        release_frozen_orders(ctx, frozen);
        return rtx_hip_fail(ctx, e, "hipGraphInstantiate");
    }
    rtx_graph_handle* h = new (std::nothrow) rtx_graph_handle();
    if (!h) {
        hipGraphExecDestroy(exec);
        release_frozen_orders(ctx, frozen);
        return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, "rtx_graph_end: out of host memory");
    }
    h->exec = exec;
    h->scene_gen = ctx->scene_gen;
    h->frozen.swap(frozen);
    *graph_out = h;
    return RTX_OK;
}

int rtx_graph_launch(rtx_ctx* ctx, void* graph, void* stream_v)
{
    if (!ctx || !graph) return RTX_ERR_INVALID_ARGUMENT;
    rtx_graph_handle* h = static_cast<rtx_graph_handle*>(graph);
    if (h->scene_gen != ctx->scene_gen) {
        return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "rtx_graph_launch: the scene was edited after this graph was recorded (its launches keep the old object "
                                                       "counts and array addresses): re-capture after a scene edit");
    }
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    RTX_HIP(ctx, hipGraphLaunch(h->exec, stream_v ? (hipStream_t)stream_v : ctx->stream));
    return RTX_OK;
}

void rtx_graph_destroy(rtx_ctx* ctx, void* graph)
{
    if (ctx && graph) {
        rtx_graph_handle* h = static_cast<rtx_graph_handle*>(graph);
        hipSetDevice(ctx->device);
        if (h->exec) hipGraphExecDestroy(h->exec);
        release_frozen_orders(ctx, h->frozen);
        delete h;
    }
}

} // extern "C"
