// rtx_ctx.h -- the context behind the C ABI, shared by rtx_api.cpp and rtx_post.hip.
#pragma once

#include "../../include/rtx.h"
#include "rtx_kernels.h"
#include "rtx_plan.hpp"

#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

struct DeviceArray {
    void* p = nullptr;
    size_t cap = 0; // elements
};

struct HostPlane {
    float4 a, b, c, od;
};

struct rtx_group; // rtx_group.cpp: the device group a context is the root of

// bit casts (the creation index of an object travels in the .w of its float4s)
inline float bits_to_float(uint32_t u)
{
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

inline uint32_t float_to_bits(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

struct rtx_ctx {
    int device = 0;
    rtx_group* group = nullptr;     // not null: this context is the root of a device group (rtx_group_create)
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    size_t max_w = 0, max_h = 0, capacity = 0;
    uint8_t* d_frame = nullptr;
    uint8_t* d_min = nullptr;       // minimise output (allocated on first use)
    void* d_scan = nullptr;         // minimise scratch
    size_t scan_bytes = 0;
    uint32_t* d_words = nullptr;    // rtx_update's pixel words (W * H; allocated on first use)
    size_t words_cap = 0;
    int64_t opt_update_words = -1;  // -1 auto (on), 0 off: rtx_update traces pixel words and minimises from them
    int64_t opt_update_host_write = -1; // RTX_OPT_UPDATE_HOST_WRITE: -1 auto (frames up to 2^17 slots), 0 off, 1 on
    uint64_t stat_host_writes = 0;
    uint64_t* h_pair = nullptr;     // two pinned words: a rank's stream length and failure word (rtx_update on a group, RTX_OPT_GROUP_UPDATE)
    uint64_t* d_look = nullptr;     // rtx_minw_fused's look-back tables (agg, grp: rtx_post.hip), zeroed when allocated
    size_t look_blocks = 0;         // ... sized for this many blocks
    uint32_t look_epoch = 0;        // of the last fused launch (0: none yet; never used as a tag)
    uint32_t min_fused_epoch = 0;   // not 0: the minimise launches just queued were the fused kind, under this epoch
    int64_t opt_min_fused = -1;     // RTX_OPT_MINIMIZE_FUSED: -1 auto (on), 0 three launches, 1 on, 2 on with blocks that give up (tests)
    uint64_t stat_min_fallbacks = 0; // fused minimise launches that gave up and were redone as three launches
    uint8_t* d_grey = nullptr;
    size_t dirty_hi = 0;            // bytes of d_frame that may be non-zero

    // scene: host staging for objects not yet uploaded + device SoA (the device copy is the truth
    // once uploaded, because UpdateObjects moves spheres there)
    std::vector<float4> h_sph_geom, h_sph_color, h_sph_od, h_sph_motion;
    std::vector<HostPlane> h_planes;
    uint32_t ns = 0, np = 0, next_gidx = 0;
    uint32_t ns_uploaded = 0, np_uploaded = 0;
    DeviceArray d_sph_geom, d_sph_color, d_sph_od, d_sph_motion, d_pl_a, d_pl_b, d_pl_c, d_pl_od;
    // the direction-sorted copy of the sphere array that staging reads (rtx_sort_scene): geometry by sorted position, sorted
    // position -> sphere index, sphere index -> sorted position; valid while sorted_gen == scene_gen
    DeviceArray d_sorted_geom, d_sorted_od, d_sorted_idx, d_pos_of;
    std::vector<float4> h_centres;  // cx cy cz r as created, by sphere index (the sort's input; the device copy moves under physics)
    uint64_t sorted_gen = 0;
    int64_t opt_sorted_store = -1;  // -1 auto (on), 0 off
    std::vector<uint8_t> kind_of;   // per creation index: 1 plane, 2 sphere (Object3D.h:14)
    std::vector<uint32_t> local_of; // per creation index: index within its kind

    int64_t opt_kernel = RTX_KERNEL_AUTO;
    int64_t opt_tile_log2w = 0;
    int64_t opt_subtiles = 0;
    int64_t opt_two_level = -1;     // -1 auto, 0 off, 1 on
    int64_t opt_refine = -1;        // -1 auto, 0 off, 1 on
    int64_t opt_cell_capacity = 0;  // entries per coarse cell list; 0 = 4 ns / cells + 1024
    // two-level culling scratch, one set per stream that renders (launches on one stream are ordered, so a
    // set is never shared by frames in flight on different streams)
    struct CellScratch {
        hipStream_t stream = nullptr;
        uint32_t* list = nullptr;                 // cells x capacity sphere indices
        uint32_t* count = nullptr;                // two alternating buffers of one counter per cell
        size_t list_words = 0, count_words = 0;
        uint32_t n_cells = 0;                     // the cell grid the counters were last zeroed for
        uint32_t flip = 0;                        // which counter buffer the next launch accumulates into
    };
    std::vector<CellScratch> cell_scratch;

    // dispatch order of the macro tiles, one set per (render stream, tile grid): everything that touches a set is queued
    // on that one stream, in order (trace writes the estimates, rtx_order_tiles turns them into the order the following
    // traces read) or on the context's side stream behind events (rtx_balance_tiles).  The decisions are
    // rtxplan::DispatchOrder's; this holds the buffers and events.
    struct TileOrder {
        hipStream_t stream = nullptr;
        uint64_t key[3] = {0, 0, 0}; // identifies the tile grid the buffers describe
        uint32_t* cost = nullptr;
        uint32_t* order = nullptr;   // two orders of `cap` tiles: the one in use and the one a balancing pass writes
        float* factor = nullptr;     // per-tile correction of the estimate (rtx_balance_tiles)
        hipEvent_t ev_rec = nullptr, ev_done = nullptr;
        size_t cap = 0;              // tiles the buffers hold
        uint64_t last_use = 0;       // ctx->order_clock at the last launch (least recently used set is recycled)
        bool frozen = false;         // a recorded (HIP graph) launch reads the order in use: nothing is derived for this set any more
        uint32_t frozen_refs = 0;    // ... by this many live graphs (rtx_graph_destroy of the last one releases the set)
        uint64_t id = 0;             // stable name of the set (the vector's entries move)
        bool batch = false;          // the set of a batched launch (rtx_trace_batch): plain heaviest first, nothing dealt
        const uint32_t* base = nullptr; // the static XCD-aware order of the launch in hand, or nullptr (rtx_order_tiles sorts per XCD label within it)
        rtxplan::DispatchOrder plan;
    };
    std::vector<TileOrder> tile_orders;
    uint64_t order_clock = 0, next_order_id = 1;
    std::vector<uint64_t> capture_frozen; // ids of the sets the capture in progress has frozen (handed to the graph by rtx_graph_end)

    // Coarse-cell lists that outlive a frame (two-level culling): two sets, shared by all render streams, each valid for
    // cameras within the motion budget it was binned with (rtxplan::CellCachePolicy).  A set is (re)built on the render
    // stream that missed, or ahead of time on the side stream; readers on other streams wait for `ev_built` once.
    struct CellCacheSlot {
        uint32_t* list = nullptr;
        uint32_t* count = nullptr;
        size_t list_words = 0, count_words = 0;
        hipEvent_t ev_built = nullptr;
        bool ever_built = false;
        bool built_on_aux = false;              // the last build ran on the side stream
        bool known_ready = false;               // the last build is known to have finished (hipEventQuery said so once)
        std::vector<hipStream_t> waited;        // streams that are already ordered after the last build
        // Launches that read this set: `readers` are the streams whose latest launch read it (no event yet: recorded when the
        // stream first reads the other set, or when this set is rebuilt, whichever comes first); `done` are events recorded
        // right after a stream's last launch reading it.  A rebuild waits for exactly these -- not for everything the
        // render streams have queued since, which would drain the frames in flight at every rebuild.
        std::vector<hipStream_t> readers;
        std::vector<hipEvent_t> done;           // in use: done[0 .. n_done)
        size_t n_done = 0;
    };
    CellCacheSlot cell_cache[2];
    rtxplan::CellCachePolicy cell_policy;
    // capacity feedback: the binning passes keep the longest list needed in d_cell_max; it is copied to the pinned word
    // h_cell_max after a build (and now and then on the per-frame path) and read, unsynchronised, when the next launch is planned
    uint32_t* d_cell_max = nullptr;
    volatile uint32_t* h_cell_max = nullptr;
    uint32_t cell_cap_floor = 0;                // capacity the lists of the current grid are planned with at least
    uint64_t cell_grid_id[3] = {0, 0, 0};       // the grid (and scene generation) the two words above belong to
    uint64_t per_frame_bins = 0;
    struct XcdOrder {                           // static dispatch order of a two-level grid: a cell's tiles share an XCD
        uint32_t* p = nullptr;
        size_t cap = 0;
        uint64_t key[2] = {0, 0};               // (tile grid, cell shape)
        uint64_t last_use = 0;
    };
    XcdOrder xcd_orders[4];                     // the grids seen last (least recently used is replaced)
    int64_t opt_cell_reuse = -1;                // -1 auto (on), 0 off: bin per frame as before round 3
    int64_t opt_xcd_order = -1;                 // -1 auto (on for two-level grids), 0 off
    // view-density feedback (rtxplan::ViewDensity): the longest candidate list the trace workgroups of an epoch (8 culling
    // launches) report, copied to the pinned word when the epoch ends and taken as an observation once that copy has landed
    uint32_t* d_longest = nullptr;              // three words in rotation: the epoch being filled, the next one (zeroed), the one before (being copied)
    volatile uint32_t* h_longest = nullptr;
    hipEvent_t ev_longest = nullptr;
    bool longest_copy_pending = false;
    uint32_t longest_epoch = 0, longest_launches = 0;
    rtxplan::ViewDensity view_density;
    int64_t opt_view_adapt = -1;                // -1 auto (on), 0 off
    uint64_t stat_density_switches = 0;
    hipStream_t recent_streams[16] = {nullptr}; // the render streams of the last two-level launches
    unsigned recent_pos = 0, render_streams_seen = 0;
    hipEvent_t ev_physics = nullptr;            // orders a build on the side stream after the physics steps queued so far
    bool ns_moved_since_build = false;          // rtx_update_objects ran since the last such ordering
    uint64_t scene_gen = 1;                     // bumped by every scene edit: object counts / array addresses (recorded graphs belong to one)
    uint64_t lists_gen = 1;                     // ... and by the first physics step after one: what cell lists belong to
    bool physics_settled = false;               // every sphere has been through Sphere::Update since the last edit (|y| <= 10)
    uint64_t stat_order_passes = 0;
    uint64_t stat_cell_builds = 0, stat_cell_prefetches = 0, stat_cell_hits = 0, stat_cell_per_frame = 0;

    hipStream_t aux_stream = nullptr; // the balancing passes' stream (created with the first pass)
    double scene_drift = 0.0;        // how far any sphere can have moved since the context was created (rtx_update_objects:
                                     // |dt| x the largest |speed x mover|; scene edits add 1e3): dispatch orders go stale with it
    float max_speed = 0.0f;          // largest |speed * mover| any sphere was given: what a physics step moves it by per unit of dt
    int64_t opt_batch = -1;         // -1 auto (on), 0 off: rtx_submit_slabs renders consecutive slabs of one stream with one launch
    uint64_t stat_batched_launches = 0;
    int64_t opt_tile_order = -1;    // -1 = auto (grids of one dispatch round, period 16), 0 = off, k = on: re-derive the order after
                                    // the 1st and 2nd frame of a grid, then every k-th
    int n_cu = 0;                   // compute units of the device

    // events of rtx_submit_slabs' fork/join: one for `after`, one per distinct render stream seen
    hipEvent_t ev_fork = nullptr;
    struct JoinEvent {
        hipStream_t stream = nullptr;
        hipEvent_t ev = nullptr;
    };
    std::vector<JoinEvent> join_events;

    // pipelined Update (rtx_update_begin / rtx_update_end): two slots, each with its own frame, minimise
    // buffer, scan scratch and events; the copy of slot k's stream to the host runs on copy_stream while slot
    // k^1 is being traced
    struct UpdateSlot {
        uint8_t* d_frame = nullptr;  // (the record form only)
        uint32_t* d_words = nullptr; // (the word form only)
        size_t words_cap = 0;
        uint8_t* d_min = nullptr;
        void* d_scan = nullptr;
        size_t scan_bytes = 0;
        uint64_t* h_total = nullptr; // pinned
        hipEvent_t ev_ready = nullptr, ev_copied = nullptr;
        size_t bytes = 0;
        bool busy = false;
        // a small frame whose Minimize launch writes the caller's buffer itself (RTX_OPT_UPDATE_HOST_WRITE): nothing was waited for in
        // rtx_update_begin; rtx_update_end waits for ev_ready and reads the length from h_total
        bool host_write = false;
        uint32_t hw_epoch = 0;
        int hw_mode = 0;
        size_t hw_w = 0, hw_h = 0;
        const uint32_t* hw_words = nullptr;
        uint8_t* hw_out = nullptr;
    };
    UpdateSlot upd[2];
    hipStream_t copy_stream = nullptr;
    unsigned upd_next = 0;

    std::string error;
    const char* last_kernel = "";
};


// rtx_api.cpp
int rtx_fail(rtx_ctx* ctx, int status, const std::string& msg);
int rtx_hip_fail(rtx_ctx* ctx, hipError_t e, const char* what);
int rtx_sync_scene(rtx_ctx* ctx);
void rtx_scene_edited(rtx_ctx* ctx);
int rtx_sort_scene(rtx_ctx* ctx, const float origin[3]); // rtx_post.hip
// the zero-fill bookkeeping of the context's own frame buffer for a frame of `mode` whose records something other than
// rtx_render_rows is about to write there (a group's gathered slabs, its expanded words), on the context's stream
extern "C" int rtx_frame_zero_semantics(rtx_ctx* ctx, int mode, uint64_t W, uint64_t H, unsigned flags); // (hidden: not part of the ABI)
bool rtx_group_stat(const rtx_ctx* ctx, int option, int64_t* value); // rtx_group.cpp

#define RTX_HIP(ctx, call)                          \
    do {                                            \
        hipError_t e__ = (call);                    \
        if (e__ != hipSuccess) {                    \
            return rtx_hip_fail((ctx), e__, #call); \
        }                                           \
    } while (0)
