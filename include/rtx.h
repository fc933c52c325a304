/*
 * rtx.h -- C ABI of the MI355X (gfx950) ray-trace hot path.
 *
 * This is the drop-in boundary for ONE path of EmilHogstedt/Raytracing-in-Windows-Console:
 * RayTracingManager::Update -> RayTracing::RayTrace -> RayTrace_<MODE> kernels
 * (primary-ray generation, sphere/plane closest hit, Blinn-Phong shading, ANSI record
 * write) plus the Minimize pass and the UpdateObjects step that Update runs around it.
 * Plain pointers and sizes only; every entry point returns an int status (0 = RTX_OK) and
 * never exits the process.  Reference citations are file:line under ConsoleProject/.
 *
 * The reference has no FFI layer: its seams are in-process C++.  Each entry
 * point below names the reference interface it replaces; include/rtx_compat.hpp rebuilds
 * the reference's classes (RayTracingManager, RayTracing, Scene3D, ...) on top of this ABI,
 * and INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Threading: one caller per context, blocking unless a call says otherwise (the reference
 * drives the path from its main thread only, Engine3D.cpp:81-107).
 */
#ifndef RTX_H
#define RTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden; only this header's entry points are exported */
#pragma GCC visibility push(default)

typedef struct rtx_ctx rtx_ctx;

/* Status codes. */
enum rtx_status {
    RTX_OK = 0,
    RTX_ERR_INVALID_ARGUMENT = 1,
    RTX_ERR_INVALID_MODE = 2,    /* RayTracing.cu:863-865 asserts; this ABI reports */
    RTX_ERR_HIP = 3,             /* a HIP runtime call failed: rtx_last_error() has the text (pch.h:45-53 exits instead) */
    RTX_ERR_OUT_OF_MEMORY = 4,
    RTX_ERR_NO_DEVICE = 5,       /* no gfx950 device visible: the product never falls back to a CPU path */
    RTX_ERR_TOO_LARGE = 6        /* frame larger than the context was created for */
};

/* enum RenderingMode, RayTracingManager.h:21 (same values). */
enum rtx_mode {
    RTX_BIT_ASCII = 0,
    RTX_BIT_PIXEL = 1,
    RTX_RGB_ASCII = 2,
    RTX_RGB_PIXEL = 3,
    RTX_RGB_NORMALS = 4,
    RTX_SDL = 5
};

/* Bytes per pixel record: SIZE_8BIT / SIZE_RGB, RayTracing.h:120-124. */
#define RTX_SIZE_8BIT 12
#define RTX_SIZE_RGB 20

/* struct RayTracingCPUToGPUData, RayTracingManager.h:9-19: the logical payload without the
 * six vptrs the reference's Matrix/Vector classes carry.  inv_v is row-major (row1..row4). */
typedef struct rtx_params {
    float inv_v[16];
    float cam_pos[3];
    float element1;  /* projection matrix [0][0], Engine3D.cpp:94 */
    float element2;  /* projection matrix [1][1], Engine3D.cpp:95 */
    float cam_far;   /* Engine3D.cpp:96 */
    uint64_t x;      /* console width  W, Engine3D.cpp:92 */
    uint64_t y;      /* console height H, Engine3D.cpp:93 */
} rtx_params;

/* Which kernel family renders (RTX_OPT_KERNEL). */
enum rtx_kernel {
    RTX_KERNEL_AUTO = 0,   /* binned when it pays, brute otherwise */
    RTX_KERNEL_BRUTE = 1,  /* every pixel tests every object, scene tiles staged in LDS */
    RTX_KERNEL_BINNED = 2  /* per-workgroup conservative frustum culling, then the same exact tests */
};

enum rtx_option {
    RTX_OPT_KERNEL = 1,       /* enum rtx_kernel */
    RTX_OPT_TILE_LOG2_W = 2,  /* log2 of the sub-tile width in pixels (2..6; a sub-tile is 256 pixels); 0 = choose from the camera */
    RTX_OPT_SUBTILES = 3,     /* sub-tiles per workgroup in the binned kernel (1, 2, 4, 8, 16); 0 = default */
    RTX_OPT_TWO_LEVEL = 4,    /* coarse-cell pre-pass before the binned kernel (one launch: blocks of 4x4 cells, then the cells):
                               * -1 auto (from 2048 spheres; from 256 while the view is locally dense, RTX_OPT_VIEW_ADAPT; smaller
                               * scenes get exact lists while camera and scene rest, see RTX_OPT_CELL_REUSE), 0 off, 1 on (2 is
                               * accepted and means 1) */
    RTX_OPT_CELL_CAPACITY = 7, /* entries per coarse-cell list of that pre-pass; 0 = auto (4 * spheres / cells + 1024, so that the
                               * scratch is O(spheres)).  A cell whose list does not fit falls back to the whole scene: slower,
                               * the same frame */
    RTX_OPT_TILE_ORDER = 6,   /* binned kernel: dispatch the macro tiles heaviest first and dealt evenly over the CUs, from the work
                               * estimates (and, for grids of one dispatch round, the measured durations) that earlier frames of
                               * the same tile grid on the same stream left behind.  Speed only: the frame is the same in any
                               * order.  -1 = auto (default): on for tile grids whose workgroups are all resident at once (5 or 6
                               * sub-tiles per workgroup are chosen to make that so, e.g. at 1080p); those are balanced by
                               * rtx_balance_tiles on a stream of the library's own, every 4th frame for the first 64 frames of a
                               * grid, then every 64th (one 1080p launch alone: 29.3 -> 25.9 us).  Larger grids (several dispatch
                               * rounds, up to six): heaviest first only while the caller renders on a single stream -- then nothing
                               * overlaps the end of a launch, and the workgroups dispatched last should be the light ones (a dense
                               * 1080p scene alone 44.6 -> 41.4 us); with frames in flight on several streams the order gains
                               * nothing, costs 1-2 % and at 8K 10 % by separating tiles that share 128-byte lines: off.
                               * 0 = frame order; k > 0 = on for every grid, the order re-derived every k-th frame */
    RTX_OPT_CELL_REUSE = 8,   /* coarse-cell lists of that pre-pass outlive the frame: binned with every sphere's culling margin grown by a
                               * motion budget (about sixteen frames of the camera's and the spheres' current motion), the lists serve
                               * every later frame whose camera stays within it -- for a static view: all of them -- and are rebuilt
                               * ahead of time, beside the frames, on the library's side stream.  A camera too fast for that (a
                               * quarter of a cell in under four frames) gets the per-frame pre-pass.  Scenes under 2048 spheres, for
                               * which a pre-pass per frame does not pay, get exact lists while camera and scene rest (two launches
                               * in a row without motion) and none otherwise.  Same frames either way: a list is always a superset
                               * of what a pixel ray of its cell can hit.  -1 auto (on), 0 off, 1 on */
    RTX_OPT_XCD_ORDER = 9,    /* two-level grids: dispatch the macro tiles so that a cell's tiles (and neighbouring cells) run on one
                               * XCD, whose L2 then holds that part of the scene alone.  Speed only.  -1 auto (on), 0 off, 1 on */
    RTX_OPT_VIEW_ADAPT = 10,  /* a scene that is sparse by its numbers can be locally dense from where the camera stands (config 2 seen along
                               * its long axis: 95 candidates on one macro tile instead of 9, one launch alone 63 us instead of 27).  The
                               * trace workgroups report their longest candidate list; when it passes 28 the following launches are
                               * planned as for a dense scene (2 sub-tiles per workgroup, two-level culling, per-wave refinement: 23-27 us
                               * on those views), and as before again once it has stayed short.  Same frames either way.
                               * -1 auto (on), 0 off, 1 on */
    RTX_OPT_SORTED_STORE = 11,/* staging reads a copy of the sphere array sorted by direction (Morton code of azimuth and elevation as seen
                               * from the camera of the first launch after a scene edit), kept in step by rtx_update_objects: the spheres
                               * of a coarse cell are then neighbours in memory (config 5: a quarter of the lines per launch).  Speed
                               * only; ties are still broken by creation order.  Scenes from 256 spheres.  -1 auto (on), 0 off, 1 on */
    RTX_OPT_BATCH = 14,       /* rtx_submit_slabs: consecutive slabs on ONE stream are traced by one launch (up to 16 frames' rows per launch; the
                               * frames' cameras and output buffers travel in the kernel arguments), instead of a launch each -- a rank's slab of
                               * a sharded 1080p frame is 255 workgroups, far too few to fill 256 CUs.  Taken where the plan has neither
                               * two-level culling nor per-wave refinement and the frames differ in camera only.  -1 auto (on), 0 off, 1 on */
    RTX_OPT_UPDATE_WORDS = 15, /* rtx_update / rtx_update_begin trace 4-byte pixel words (RTX_RENDER_COMPACT) and minimise from them instead of
                               * writing the 12 / 20-byte records and reading them back twice: the same minimised stream with a fifth of the
                               * memory traffic (1080p RGB: 8.3 MB written + 16.6 MB read instead of 41.5 + 83).  The context's frame buffer
                               * is then not written by an Update (it keeps what the last rtx_render left).  -1 auto (on), 0 off (records:
                               * the frame buffer holds the frame after every Update, as the reference's m_deviceResultArray does), 1 on */
    RTX_OPT_MINIMIZE_FUSED = 17, /* Minimize -- from pixel words (rtx_minimize_words, rtx_update with RTX_OPT_UPDATE_WORDS) and from records (rtx_minimize) -- as ONE launch: every block counts
                               * its slots, publishes its length and finds its place in the stream by a two-level look-back over the lengths of
                               * the blocks before it, instead of three launches (count, offsets, scatter; two for records) that read their input twice.  A launch
                               * whose blocks gave up waiting (bounded polling; never seen) is redone as three launches: RTX_STAT_MINIMIZE_FALLBACKS.
                               * -1 auto (on), 0 off, 1 on, 2 on with blocks that give up on purpose (tests of that path) */
    RTX_OPT_GROUP_EXCHANGE = 12, /* device groups (rtx_group_create): enum rtx_group_exchange -- how the slabs reach the root */
    RTX_OPT_GROUP_THREADS = 16, /* device groups: a submission thread per rank other than the root queues that rank's launch and copy while the
                               * caller's thread queues the root's (a rank's share is ~15 us of host work; on one thread 8 ranks cost 132 us per
                               * 1080p frame).  The call still returns only when everything is queued.  -1 auto (on where the list names two or
                               * more distinct devices; with all ranks on one GPU the threads were measured to change nothing), 0 off, 1 on */
    RTX_OPT_UPDATE_HOST_WRITE = 19, /* rtx_update / rtx_update_begin (word form, one device) when the caller's buffer is pinned, device-addressable memory
                               * (rtx_host_alloc's is): the Minimize launch stores the stream and its length straight into host memory instead of
                               * leaving them to two copies.  The blocking form then waits for the device ONCE (console-sized frames 45 -> 27 us per
                               * Update, 1080p 0.395 -> 0.370 ms); the pipelined form does not wait at all before rtx_update_end (console sizes 35 ->
                               * 19 us).  The same bytes; a pageable buffer quietly takes the copy form.  -1 auto (blocking: always; pipelined: frames
                               * up to 2^17 slots -- beyond that a launch that runs at PCIe speed delays the next frame's kernels), 0 off, 1 on */
    RTX_OPT_GROUP_UPDATE = 18, /* device groups: how rtx_update / rtx_update_begin hand the minimised stream to the host.  0: the ranks' pixel words
                               * are gathered on the root, which minimises the frame and copies the stream over ITS PCIe link (the whole Update is
                               * bound by that copy: 0.33 ms per 1080p RGB frame).  1: no gather -- every rank traces its rows and the row above them,
                               * minimises its own rows (the colour carried over from the last pixel above) and copies its part of the stream to its
                               * place in the host buffer over its OWN link, N links at once (SURVEY.md 8(e)'s alternative).  The same bytes either
                               * way.  -1 auto: 1 where the list names two or more distinct devices.  A HIP error on the direct path makes the group
                               * fall back to 0 for good (RTX_STAT_GROUP_DIRECT_UPDATES counts the direct ones) */
    RTX_OPT_GROUP_WIRE = 13,  /* device groups: enum rtx_group_wire -- what travels: compact pixel words (default) or records */
    RTX_OPT_REFINE = 5        /* per-wave refinement of the candidate list in the binned kernel: -1 auto (dense scenes), 0 off, 1 on
                               * (needs at most 4 sub-tiles per workgroup and a macro tile of at most 64 x 64 pixels; otherwise it
                               * stays off) */
};

/* Read-only counters (rtx_get_option): how the coarse-cell lists of two-level culling were obtained so far. */
enum rtx_stat {
    RTX_STAT_CELL_BUILDS = 101,     /* binned in line because no cached list covered the camera */
    RTX_STAT_CELL_PREFETCHES = 102, /* binned ahead of time on the side stream */
    RTX_STAT_CELL_HITS = 103,       /* launches served by cached lists */
    RTX_STAT_CELL_PER_FRAME = 104,  /* launches that binned for themselves alone (reuse off, or a fast camera) */
    RTX_STAT_ORDER_PASSES = 105,    /* dispatch-order passes queued (rtx_balance_tiles / rtx_order_tiles) */
    RTX_STAT_ORDERS_FROZEN = 106,   /* dispatch orders a live recorded graph reads (kept as they are until it is destroyed) */
    RTX_STAT_VIEW_DENSE = 108,      /* 1 while launches are planned as for a dense scene because of what earlier launches saw */
    RTX_STAT_DENSITY_SWITCHES = 109,/* how often that changed */
    RTX_STAT_BATCHED_LAUNCHES = 114, /* launches that rendered several frames' slabs at once (RTX_OPT_BATCH) */
    RTX_STAT_GROUP_DIRECT_UPDATES = 116, /* Updates of a device group whose ranks minimised and copied their own rows (RTX_OPT_GROUP_UPDATE) */
    RTX_STAT_UPDATE_HOST_WRITES = 117, /* blocking Updates whose Minimize launch wrote the caller's host buffer itself (RTX_OPT_UPDATE_HOST_WRITE) */
    RTX_STAT_MINIMIZE_FALLBACKS = 115, /* fused Minimize launches redone as three launches (RTX_OPT_MINIMIZE_FUSED) */
    RTX_STAT_GROUP_SIZE = 110,      /* logical ranks of the device group this context is the root of (1: a plain context) */
    RTX_STAT_GROUP_EXCHANGE = 111,  /* the exchange the last sharded frame used: RTX_EXCHANGE_PEER_COPY or RTX_EXCHANGE_RCCL (0: none yet) */
    RTX_STAT_GROUP_GATHERS = 112,   /* sharded frames gathered so far */
    RTX_STAT_GROUP_BYTES = 113,     /* bytes the last gather moved between devices */
    RTX_STAT_CELL_CAPACITY_FLOOR = 107 /* entries per cell list the current grid is planned with at least (0: the default capacity has
                                     * sufficed); grown from the longest list the binning passes report */
};

/* Flags of rtx_render_rows. */
enum rtx_render_flags {
    RTX_RENDER_DEFAULT = 0,
    /* 8-bit modes use only the first 12*W*H bytes of the 20*W*H frame (RayTracing.cu:238);
     * with this flag the call also zero-fills bytes [12*W*H, 20*W*H) of a full-frame buffer,
     * as the reference's per-frame cudaMemset leaves them (RayTracingManager.cu:86,161-165). */
    RTX_RENDER_ZERO_TAIL = 1,
    /* The output is one 4-byte pixel word per pixel (W*rows words, same row addressing as the records, caller's
     * buffer only) instead of the S-byte record: byte 0..2 = the record's colour values (r, g, b; or the
     * xterm-256 index in byte 0 for the 8-bit modes), byte 3 = the glyph; 0 = pixel without a visible hit;
     * 0xffffffff = column W-1.  rtx_expand turns the words into the very records rtx_render_rows would have
     * written.  For the row-sharded multi-GPU loop: a rank ships 4 instead of 20 bytes per pixel over xGMI and
     * the GPU that assembles the frame writes the records.  No reference counterpart.  Not with RTX_SDL. */
    RTX_RENDER_COMPACT = 2,
    /* The output is the 8 floats behind each record instead of the record (32 bytes per pixel, 16-byte aligned
     * caller's buffer, same row addressing): distance (99999999.f without a hit), shadingValue, normal.xyz,
     * colour.xyz -- struct RayTraceReturnData (RayTracing.h:17-23) after RayTrace (RayTracing.cu:81-168), i.e.
     * the values the 1e-5 parity tolerance is stated on.  Normal and colour are defined for pixels with a hit
     * only; column W-1 is all zero.  For checks, not for the frame loop.  Not with RTX_SDL. */
    RTX_RENDER_VALUES = 4
};

/* One run of pixels for rtx_expand: n_pixels words starting at word src_pixel of the compact buffer become
 * the records starting at record dst_pixel of the output buffer. */
typedef struct rtx_segment {
    uint64_t src_pixel;
    uint64_t dst_pixel;
    uint64_t n_pixels;
} rtx_segment;

/* ---- context: RayTracingManager::RayTracingManager / ~RayTracingManager (RayTracingManager.cu:53-74).
 * Owns the device params block, the 20*max_w*max_h device result buffer (PrintMachine::GetMaxSize(),
 * PrintMachine.cpp:140) and the scene store.  `device` is the HIP device ordinal. */
int rtx_create(int device, size_t max_w, size_t max_h, rtx_ctx** out);
void rtx_destroy(rtx_ctx* ctx);

/* ---- device groups: the `rtx_create(ndev, ...)` of SURVEY.md 8(b) / 8(e).  One context that renders on ndev devices of this
 * process: the frame shards by pixel rows -- logical rank g (device devices[g]; NULL = devices 0 .. ndev-1) traces rows
 * [g*H/ndev, (g+1)*H/ndev) with the global row index in ray generation (RayTracing.cu:12,16) -- and the slabs are gathered
 * into the device memory of rank 0 (the root, devices[0]) by RCCL (ncclCommInitAll; grouped ncclSend / ncclRecv over xGMI,
 * ragged last slab) or hipMemcpyPeerAsync; the root then holds the byte-identical 20*W*H buffer a one-device rtx_render
 * produces (a block of rows is one contiguous byte range, RayTracing.cu:238,457).  The returned context IS the group: every
 * entry point takes it like any other context, and on it
 *   rtx_scene_* / rtx_update_objects / rtx_set_option   apply to every rank's replica of the scene (<= 6.8 MB, replicated),
 *   rtx_render, rtx_update, rtx_update_begin / _end      trace sharded and deliver on the root exactly what they deliver on one
 *                                                        device (frame buffer, minimised stream),
 *   rtx_submit_frames                                    n whole frames, sharded: every rank traces its rows of up to 16 frames with one
 *                                                        launch (RTX_OPT_BATCH) and the slabs of the chunk travel together -- the form
 *                                                        for throughput; the frames are complete in stream order on the context's own
 *                                                        stream, which the streams[i] given are made to wait for,
 *   rtx_render_rows, rtx_submit_slabs, rtx_expand, rtx_minimize, graphs   act on the root's device alone (caller's buffers there),
 *   rtx_destroy                                          releases the whole group.
 * The reference's single consumer, RayTracingManager::Update (RayTracingManager.cu:76-154, hand-off at :150), therefore
 * runs unchanged over N GPUs: include/rtx_compat.hpp takes the device list from RTX_DEVICES or Device::set_devices().
 * A device may appear more than once (several logical ranks on one GPU: how a one-GPU box walks N = 4 or 8; the gather is
 * then peer / same-device copies -- an RCCL communicator needs one rank per GPU).  ndev in [1, 64]. */
int rtx_group_create(int ndev, const int* devices, size_t max_w, size_t max_h, rtx_ctx** out);
int rtx_group_size(const rtx_ctx* ctx);                      /* logical ranks (1 for a plain context) */
rtx_ctx* rtx_group_member(rtx_ctx* ctx, int rank);           /* rank's member context, for reading (statistics, kernel names); rank 0 = ctx */
int rtx_group_rows(const rtx_ctx* ctx, size_t h, int rank, size_t* row0, size_t* rows); /* the rows rank traces of an h-row frame */
const char* rtx_group_exchange_note(const rtx_ctx* ctx);     /* one line: how the last gather moved its bytes, or why RCCL is not in use */

enum rtx_group_exchange {
    RTX_EXCHANGE_AUTO = 0,      /* RCCL where the list names ndev > 1 distinct devices and librccl loads and initialises; else peer copies */
    RTX_EXCHANGE_PEER_COPY = 1, /* hipMemcpyPeerAsync on the sender's stream, ordered by events */
    RTX_EXCHANGE_RCCL = 2,      /* grouped ncclSend / ncclRecv (distinct devices only) */
    RTX_EXCHANGE_RCCL_ALL = 3   /* ... with the root's own slab sent to itself as well: walks the RCCL path at ndev = 1 (tests) */
};
enum rtx_group_wire {
    RTX_WIRE_AUTO = 0,    /* compact */
    RTX_WIRE_RECORDS = 1, /* the 12 / 20-byte records travel and land at their place in the frame */
    RTX_WIRE_COMPACT = 2  /* 4-byte pixel words travel (RTX_RENDER_COMPACT); the root writes the records (rtx_expand) or minimises from the words */
};

/* Text of the last error on this context (or of the last failed rtx_create when ctx is NULL). */
const char* rtx_last_error(const rtx_ctx* ctx);
const char* rtx_version(void);
int rtx_set_option(rtx_ctx* ctx, int option, int64_t value);
int rtx_get_option(const rtx_ctx* ctx, int option, int64_t* value);

/* ---- scene: Scene3D::CreateSphere / CreatePlane / GetObjects (Scene3D.h:15-25, Scene3D.cpp:36-105).
 * Append-only; creation order is the closest-hit tie-break order (RayTracing.cu:100-136).
 * No 5 MB arena cap (Scene3D.h:6).  The add calls return the new object's index (>= 0) or -status. */
int rtx_scene_clear(rtx_ctx* ctx);
int rtx_scene_add_sphere(rtx_ctx* ctx, const float pos[3], float radius, const float rgb[3]);
int rtx_scene_add_plane(rtx_ctx* ctx, const float pos[3], const float normal[3], const float rgb[3],
                        float width, float height);
/* Bulk append: n records of 7 floats (cx cy cz r R G B). */
int rtx_scene_add_spheres(rtx_ctx* ctx, size_t n, const float* xyzr_rgb);
unsigned rtx_scene_count(const rtx_ctx* ctx);
/* Sphere::mover / Sphere::speed (Sphere.cu:9-12): the reference draws speed from rand(); here the caller sets it.
 * The reference only ever holds mover = -1 or +1 (Sphere.cu:9,21); any int is accepted, a step then moves the sphere by
 * speed * mover * dt (Sphere.cu:17) and the library's motion bounds (cell-list reuse, dispatch orders) count |speed * mover|. */
int rtx_scene_set_sphere_motion(rtx_ctx* ctx, unsigned index, int mover, float speed);
/* Reads object `index` back from the device store: type (1 plane, 2 sphere, Object3D.h:14) and
 * 11 floats (sphere: cx cy cz r R G B mover speed 0 0; plane: px py pz nx ny nz R G B w h). */
int rtx_scene_get_object(rtx_ctx* ctx, unsigned index, int* type, float out[11]);

/* ---- render: RayTracing::RayTrace (RayTracing.h:31-38, RayTracing.cu:797-867) together with the
 * zero-fill that precedes it in RayTracingManager::Update (RayTracingManager.cu:86).
 *
 * rtx_render: whole frame into the context's own device buffer, which afterwards holds exactly
 * the bytes the reference's buffer holds after memset + kernel (20*W*H of them).  Asynchronous
 * on the context's stream, like the reference's launch; rtx_synchronize waits
 * (RayTracingManager.cu:137). */
int rtx_render(rtx_ctx* ctx, const rtx_params* params, int mode);

/* Row-slab form for multi-GPU frames: traces rows [row0, row0+rows) with the GLOBAL row index in
 * ray generation (RayTracing.cu:12,16) and writes each row r at d_out + (r - out_row_base)*W*S.
 * d_out is device memory of the caller (NULL = the context's buffer, out_row_base then 0);
 * `stream` is a hipStream_t (NULL = the context's stream). */
int rtx_render_rows(rtx_ctx* ctx, const rtx_params* params, int mode, size_t row0, size_t rows,
                    void* d_out, size_t out_row_base, void* stream, unsigned flags);

/* Queues n whole frames with one call (a renderer keeping several frames in flight): frame i is traced with
 * params[i] into d_outs[i] (each a 20*W*H device buffer of the caller) on streams[i] (hipStream_t).  Frames
 * queued on different streams with different buffers may execute concurrently; on one stream they run in
 * order.  Asynchronous; the caller synchronises its streams.  Same effect as n rtx_render_rows calls.
 * Scene changes (rtx_scene_add_*, rtx_update_objects, rtx_scene_clear) must not race with frames in flight on
 * other streams: synchronise those streams first. */
int rtx_submit_frames(rtx_ctx* ctx, size_t n, const rtx_params* params, int mode, void* const* d_outs, void* const* streams);

/* The row-sharded form of rtx_submit_frames (one rank's share of n consecutive frames in the multi-GPU loop,
 * SURVEY.md 8(e)): rows [row0, row0+rows) of frame i are traced with params[i] into d_outs[i], whose first byte
 * is row out_row_base, on streams[i] (consecutive slabs given the SAME stream are traced by one launch, RTX_OPT_BATCH).  If `after` (a hipStream_t) is not NULL the n slabs are ordered after
 * everything queued on `after` so far, and `after` is made to wait for all of them (event fork/join inside the
 * call), so that the caller can queue the exchange of the slabs on `after` right away.  `flags` as for
 * rtx_render_rows (RTX_RENDER_COMPACT: d_outs[i] receives pixel words).  No reference
 * counterpart (the reference renders whole frames on one device, RayTracingManager.cu:122-135). */
int rtx_submit_slabs(rtx_ctx* ctx, size_t n, const rtx_params* params, int mode, size_t row0, size_t rows,
                     void* const* d_outs, size_t out_row_base, void* const* streams, void* after, unsigned flags);

/* Compact pixel words (RTX_RENDER_COMPACT) -> records of `mode` (12 bytes per pixel for the 8-bit modes, 20 for
 * the RGB ones; the record encoders of RayTracing.cu:206-252, 287-332, 370-472, 507-609, 645-751), for n_segments
 * runs of pixels, on `stream` (NULL = the context's).  d_compact and d_out are device memory of the caller,
 * 4-byte aligned (16-byte aligned destinations take the fast path).  Asynchronous. */
int rtx_expand(rtx_ctx* ctx, int mode, const void* d_compact, void* d_out, const rtx_segment* segments,
               size_t n_segments, void* stream);

/* ---- replayable launch sequences (HIP graphs): a frame loop that queues the same launches round after round (the
 * row-sharded loop: a round's slab launches, a round's expansions) pays one host call per replay instead of one per
 * launch.  rtx_graph_begin puts `stream` (a hipStream_t) into capture; the rtx_submit_slabs / rtx_submit_frames /
 * rtx_render_rows / rtx_expand calls that follow on it (rtx_submit_slabs may fork to its other streams and joins them
 * back) are recorded instead of executed; rtx_graph_end ends the capture and returns an executable graph;
 * rtx_graph_launch replays it on a stream.  Recorded launches keep their arguments (camera, buffers, row range) and
 * the scene arrays' addresses and object counts: a graph belongs to the scene it was recorded on, and after
 * rtx_scene_add_* / rtx_scene_clear rtx_graph_launch refuses it with RTX_ERR_INVALID_ARGUMENT ("re-capture after a scene
 * edit"; rtx_update_objects moves spheres in place and is fine).  Recorded launches need caller buffers (not the
 * context's own frame, whose zero-fill depends on what earlier launches left in it).  Dispatch order: nothing is derived
 * while recording; a recorded launch runs under the order its tile grid has converged to on that stream, if any -- that
 * order is then frozen while a graph that reads it lives (rtx_graph_destroy of the last one releases it) -- and in frame order otherwise.
 * Not recordable, and reported as RTX_ERR_INVALID_ARGUMENT: a launch that needs a scene upload (render once before
 * capturing) or the two-level pre-pass (its lists change from launch to launch).
 * No reference counterpart (one launch per frame on the default stream, RayTracingManager.cu:127-134). */
int rtx_graph_begin(rtx_ctx* ctx, void* stream);
int rtx_graph_end(rtx_ctx* ctx, void* stream, void** graph_out);
int rtx_graph_launch(rtx_ctx* ctx, void* graph, void* stream);
void rtx_graph_destroy(rtx_ctx* ctx, void* graph);

int rtx_synchronize(rtx_ctx* ctx);

/* The context's device result buffer (m_deviceResultArray, RayTracingManager.h:45) and its size. */
void* rtx_frame_device_ptr(rtx_ctx* ctx);
size_t rtx_frame_capacity(const rtx_ctx* ctx);
/* Blocking device-to-host copy of the first `bytes` of that buffer (RayTracingManager.cu:143). */
int rtx_read_frame(rtx_ctx* ctx, void* host_out, size_t bytes);

/* ---- Minimize: RayTracingManager::MinimizeResults / Minimize8bit / MinimizeRGB
 * (RayTracingManager.cu:167-319), on the GPU.  d_in is a 20*W*H frame in device memory (NULL =
 * the context's buffer), d_out device memory with room for S*W*H bytes (NULL = the context's
 * own minimise buffer).  *out_bytes receives the minimised length.  Blocking. */
int rtx_minimize(rtx_ctx* ctx, int mode, size_t w, size_t h, const void* d_in, void* d_out, size_t* out_bytes);
void* rtx_minimized_device_ptr(rtx_ctx* ctx);
/* The same pass over W*H compact pixel words (RTX_RENDER_COMPACT: what a sharded frame's slabs travel as) instead of records:
 * the stream is byte for byte what rtx_minimize makes of the records rtx_expand would write from these words.  A word
 * 0xffffffff outside column W-1 is an empty slot (a slot whose record would be all NUL).  d_words: device memory, 4-byte
 * aligned; d_out as for rtx_minimize.  Blocking.  Not with RTX_SDL. */
int rtx_minimize_words(rtx_ctx* ctx, int mode, size_t w, size_t h, const void* d_words, void* d_out, size_t* out_bytes);

/* ---- UpdateObjects: the physics kernel Update launches before tracing (RayTracingManager.cu:10-44,
 * 89-107; Sphere.cu:15-23), with a launch shape that stays valid past 1024 objects. */
int rtx_update_objects(rtx_ctx* ctx, double dt);

/* ---- RayTracingManager::Update (RayTracingManager.cu:76-154) in one call: params upload, zero
 * semantics, UpdateObjects(dt) when run_physics != 0, trace, GPU minimise, and the copy of the
 * minimised stream to host_out (room for 20*W*H bytes).  On return *out_bytes is what the
 * reference hands to PrintMachine::SetDataInBackBuffer (RayTracingManager.cu:150).  (By default the frame is traced as pixel
 * words and minimised from those, RTX_OPT_UPDATE_WORDS: the stream is the same, the context's frame buffer is left alone.) */
int rtx_update(rtx_ctx* ctx, const rtx_params* params, int mode, double dt, int run_physics,
               void* host_out, size_t* out_bytes);

/* Pipelined form of rtx_update (the copy of frame k overlapped with the trace of frame k+1).
 * rtx_update_begin queues physics, trace and minimise of a frame, waits for them, then starts the copy of the
 * minimised stream into host_out (pinned memory from rtx_host_alloc for full PCIe rate) on a separate stream and
 * returns a ticket; rtx_update_end(ticket) waits for that copy and reports its length.  At most two frames may
 * be in flight; host_out must stay valid and unread until rtx_update_end. */
int rtx_update_begin(rtx_ctx* ctx, const rtx_params* params, int mode, double dt, int run_physics, void* host_out, int* ticket);
int rtx_update_end(rtx_ctx* ctx, int ticket, size_t* out_bytes);

/* ---- ansi256_from_rgb (ANSIRGB.h:141-189) on its own: the xterm-256 index (16..255) of each packed 0xRRGGBB
 * value first_rgb, first_rgb+1, ..., first_rgb+count-1 (first_rgb + count <= 2^24), one byte per value into
 * d_out (device memory of the caller), computed by the very device function and grey lookup the two 8-bit trace
 * kernels use (RayTracing.cu:210,291).  Lets a checker cover all 2^24 inputs, which no frame does.  Asynchronous
 * on `stream` (NULL = the context's). */
int rtx_ansi256_map(rtx_ctx* ctx, uint32_t first_rgb, size_t count, void* d_out, void* stream);

/* ---- pinned host memory for the buffers Update copies into (m_minimizedResultArray / m_hostResultArray,
 * RayTracingManager.cu:62-66, which the reference allocates pageable): device-to-host copies into it run at
 * PCIe rate instead of through a staging bounce.  Optional: any host pointer is accepted by rtx_update. */
void* rtx_host_alloc(rtx_ctx* ctx, size_t bytes);
void rtx_host_free(rtx_ctx* ctx, void* p);

/* ---- measurement helpers (bench.py): HIP events on the context's stream. */
int rtx_timer_start(rtx_ctx* ctx);
int rtx_timer_stop(rtx_ctx* ctx, float* elapsed_ms); /* records, synchronises, returns start->stop */
/* Name of the kernel the last rtx_render/rtx_render_rows launched (for matching rocprofv3 rows). */
const char* rtx_last_kernel_name(const rtx_ctx* ctx);

/* ---- host-side input builders (no GPU work; pure fp32 host math).
 * rtx_camera_params: what Engine3D::Render fills (Engine3D.cpp:90-97) from Camera3D::Init/Update/
 * GetInverseVMatrix (Camera3D.cpp:8-48, 51-98, 207-376) for a camera at pos with rotation
 * (pitch, yaw, roll); NULL pos/rot = the reference's start pose (Camera3D.h:59-62). */
int rtx_camera_params(size_t w, size_t h, const float pos[3], const float rot[3], rtx_params* out);
/* SURVEY.md Appendix D synthetic scenes (the BASELINE.json configs): fills n_spheres*7 floats
 * (cx cy cz r R G B) and n_planes*11 floats (px py pz nx ny nz R G B w h), n_planes <= 6. */
int rtx_synth_scene(uint32_t seed, size_t n_spheres, size_t n_planes, float element1, float element2,
                    float* spheres_out, float* planes_out);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
