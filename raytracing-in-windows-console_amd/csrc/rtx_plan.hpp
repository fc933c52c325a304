// rtx_plan.hpp -- the host-side planning of a trace launch, as pure functions and small state machines.
//
// No HIP in here: everything rtx_render_rows decides before it queues anything -- the tile shape of a launch, the
// coarse-cell grid of two-level culling, when a dispatch order is derived / used / stale, whether cached cell lists
// still cover the current camera -- is computed from plain numbers, so that it compiles with any C++17 compiler and is
// unit-tested on the CPU (tests/host/test_plan.cpp under -fsanitize=address,undefined).  rtx_api.cpp turns the
// decisions into launches, events and buffers.
//
// Replaces the launch-shape arithmetic of RayTracingManager::Update (RayTracingManager.cu:120-134: a fixed 16x16 block
// grid) and fills in the culling the reference only sketches in comments (RayTracingManager.cu:21-24, 109-117).
#pragma once

#include <algorithm>
#include <cmath>
#include <limits>
#include <cstdint>
#include <cstring>
#include <vector>

#ifndef RTX_WAVES_PER_EU
#define RTX_WAVES_PER_EU 7 // waves per SIMD the trace kernels are compiled for (72 VGPRs)
#endif

namespace rtxplan {

// A 256-thread workgroup puts one wave on each of a CU's four SIMDs, so a CU holds as many workgroups of the trace
// kernels as a SIMD holds waves (their LDS, under 160 KB / 7, is sized to allow it).  The one place this number lives.
constexpr uint32_t kResidentPerCU = RTX_WAVES_PER_EU;

// ------------------------------------------------------------------------------------------------ views

// The part of the frame parameters the per-tile work depends on: the rotation (upper 3x3 of inverseVMatrix, row-major)
// and the camera position.
struct View {
    float rot[9] = {0};
    float pos[3] = {0};
};

inline View view_of(const float inv_v[16], const float cam_pos[3])
{
    View v;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) v.rot[3 * r + c] = inv_v[4 * r + c];
    }
    for (int k = 0; k < 3; k++) v.pos[k] = cam_pos[k];
    return v;
}

// How far two views are apart for the dispatch order's purposes: the largest change of a rotation entry (radians, for
// small turns) or of a tenth of a position unit.  NaN counts as "as far as can be".
inline float view_distance(const View& a, const View& b)
{
    float d = 0.0f;
    for (int k = 0; k < 12; k++) {
        const float x = k < 9 ? a.rot[k] : a.pos[k - 9], y = k < 9 ? b.rot[k] : b.pos[k - 9];
        const float e = std::fabs(x - y) * (k < 9 ? 1.0f : 0.1f);
        d = (e > d || e != e) ? e : d;
    }
    return d == d ? d : 1.0e30f;
}

// Extent of one pixel step on the view plane (tangent units): horizontally 2 e1 / W |M col 0|, vertically
// 2 e2 / H |M col 1|.  Returns false (and 1, 1) for degenerate parameters.
inline bool pixel_steps(const float inv_v[16], float e1, float e2, uint64_t W, uint64_t H, double* sx, double* sy)
{
    const float* m = inv_v;
    const double cx = std::sqrt((double)m[0] * m[0] + (double)m[4] * m[4] + (double)m[8] * m[8]);
    const double cy = std::sqrt((double)m[1] * m[1] + (double)m[5] * m[5] + (double)m[9] * m[9]);
    *sx = 2.0 * std::fabs((double)e1) / (double)W * cx;
    *sy = 2.0 * std::fabs((double)e2) / (double)H * cy;
    if (!(*sx > 0.0) || !(*sy > 0.0) || !std::isfinite(*sx) || !std::isfinite(*sy)) {
        *sx = *sy = 1.0;
        return false;
    }
    return true;
}

// The culling pyramids' planes, well conditioned.  A pixel direction is linear in the view-plane point (cx, cy):
//   w(cx, cy) = A cx + B cy + C,   A = e1 (m0, m4, m8),  B = e2 (m1, m5, m9),  C = (m2, m6, m10).
// The plane through the apex and a ROW edge cy = y (all cx) has normal w(x0, y) x w(x1, y) = (x0 - x1) A x (B y + C), the plane
// through a COLUMN edge cx = x has normal (y0 - y1) B x (A x + C).  With P = A x B, Qr = A x C, Qc = B x C and the two signs
// sr = sign(Qr . B), sc = sign(Qc . A) -- constants of the frame --
//   n_up(y)    = sr ( y P + Qr)   is the normal of the row edge y that points towards larger cy (up the frame),
//   n_right(x) = sc (-x P + Qc)   the normal of the column edge x that points towards larger cx,
// so a rectangle's inward normals are -n_up(y_top), +n_up(y_bottom), +n_right(x_left), -n_right(x_right): no orientation test on
// the device.  The kernels used to take the cross product of the two corner DIRECTIONS in fp32; at 8K the reference's
// horizontal tangent extent is 25 (element1 = 0.577 H / 100), a 16-column tile at the frame's edge spans 1.7e-4 rad, and that
// cross product of two nearly parallel vectors of length 25 loses the plane: by 4e-6 rad with the reference's own camera
// matrices (most of the half pixel the pyramid is grown by, 5.3e-6 rad there), by up to 8e-5 rad with a rolled camera -- then a
// culling kernel drops rows of a large, far sphere's cap (tools/wide_view_directed_gpu.py).  P, Qr, Qc are cross
// products of (for a camera matrix) perpendicular vectors, computed here in double from the fp32 parameters and rounded once;
// on the device one multiply-add per component gives the normal to ~4e-7 whatever the tile's size.  pp, qrqr, qcqc are the
// squared lengths the device uses to refuse an ill-conditioned sum (a sheared matrix: |y P + Qr|^2 < (y^2 |P|^2 + |Qr|^2) / 4
// -> the plane never culls).
//
// The fifth plane is ONE for the whole frame: fwd, the unit vector along C, when every pixel ray of the frame (half a pixel out,
// like the pyramids) points into its front half-space by a margin -- true for every frame the reference can produce (tangent
// extents are finite: less than 180 degrees).  It culls what lies behind the camera; the per-tile axis planes of rounds 1-2 did
// the same job tile by tile (each needing its four corner directions checked against its own axis at the head of every
// workgroup) and culled nothing more: what the four side planes of a tile let through behind the apex is a thin region along the
// tile's backward axis, which lies behind the camera plane as well.  Zero when the frame is too wide or the matrix degenerate.
struct EdgeBasis {
    float up_p[3] = {0, 0, 0}, up_q[3] = {0, 0, 0};         // n_up(y) = y up_p + up_q
    float right_p[3] = {0, 0, 0}, right_q[3] = {0, 0, 0};   // n_right(x) = x right_p + right_q
    float pp = 0, qrqr = 0, qcqc = 0;                        // |P|^2, |Qr|^2, |Qc|^2, rounded up
    float fwd[3] = {0, 0, 0};                                // unit normal of the camera plane, or zero
};

inline EdgeBasis edge_basis(const float inv_v[16], float e1, float e2, uint64_t W, uint64_t H)
{
    const float* m = inv_v;
    const double A[3] = {(double)e1 * m[0], (double)e1 * m[4], (double)e1 * m[8]};
    const double B[3] = {(double)e2 * m[1], (double)e2 * m[5], (double)e2 * m[9]};
    const double C[3] = {(double)m[2], (double)m[6], (double)m[10]};
    auto cross = [](const double* a, const double* b, double* o) {
        o[0] = a[1] * b[2] - a[2] * b[1];
        o[1] = a[2] * b[0] - a[0] * b[2];
        o[2] = a[0] * b[1] - a[1] * b[0];
    };
    auto dot = [](const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    double P[3], Qr[3], Qc[3];
    cross(A, B, P);
    cross(A, C, Qr);
    cross(B, C, Qc);
    EdgeBasis e;
    const double det_r = dot(Qr, B), det_c = dot(Qc, A); // -det[A B C] and +det[A B C]
    const double scale = std::sqrt(dot(A, A) * dot(B, B) * dot(C, C));
    bool ok = std::isfinite(scale) && std::isfinite(det_r) && std::isfinite(det_c) && std::fabs(det_r) > 1e-9 * scale && std::fabs(det_c) > 1e-9 * scale;
    const double sr = det_r > 0.0 ? 1.0 : -1.0, sc = det_c > 0.0 ? 1.0 : -1.0;
    for (int k = 0; k < 3 && ok; k++) {
        e.up_p[k] = (float)(sr * P[k]);
        e.up_q[k] = (float)(sr * Qr[k]);
        e.right_p[k] = (float)(-sc * P[k]);
        e.right_q[k] = (float)(sc * Qc[k]);
        ok = ok && std::isfinite(e.up_p[k]) && std::isfinite(e.up_q[k]) && std::isfinite(e.right_q[k]);
    }
    if (ok) {
        // squared lengths, rounded UP a little: they only feed the refusal test, which must not be passed by rounding
        e.pp = (float)(dot(P, P) * (1.0 + 1e-6));
        e.qrqr = (float)(dot(Qr, Qr) * (1.0 + 1e-6));
        e.qcqc = (float)(dot(Qc, Qc) * (1.0 + 1e-6));
        ok = std::isfinite(e.pp) && std::isfinite(e.qrqr) && std::isfinite(e.qcqc);
    }
    if (!ok) {
        // degenerate, NaN or overflowing parameters: zero normals, and lengths of +inf make every side plane fail the refusal
        // test as well -- nothing is ever culled
        e = EdgeBasis();
        e.pp = e.qrqr = e.qcqc = std::numeric_limits<float>::infinity();
        return e;
    }
    // the camera plane: along C, if the frame's four corner rays (half a pixel out) all make less than ~89.99 degrees with it
    const double cl = std::sqrt(dot(C, C));
    if (cl > 0.0 && W > 0 && H > 0) {
        const double f[3] = {C[0] / cl, C[1] / cl, C[2] / cl};
        const double xs = 1.0 + 1.0 / (double)W, ys = 1.0 + 1.0 / (double)H;
        double lo = 1.0;
        for (int i = 0; i < 4; i++) {
            const double cx = (i & 1) ? xs : -xs, cy = (i & 2) ? ys : -ys;
            const double w[3] = {A[0] * cx + B[0] * cy + C[0], A[1] * cx + B[1] * cy + C[1], A[2] * cx + B[2] * cy + C[2]};
            const double wl = std::sqrt(dot(w, w));
            lo = std::fmin(lo, wl > 0.0 ? dot(w, f) / wl : -1.0);
        }
        if (lo > 1.0e-4) {
            for (int k = 0; k < 3; k++) e.fwd[k] = (float)f[k];
        }
    }
    return e;
}

// log2 of the width of the squarest (on the view plane) rectangle of `pixels` pixels, clamped to [lo, hi]:
// minimise |log2((2^l * aspect) / (pixels / 2^l))|  ->  l = (log2(pixels) - log2(aspect)) / 2.
inline uint32_t squarest_log2w(double aspect, int pixels, int lo, int hi)
{
    long l = std::lround(0.5 * (std::log2((double)pixels) - std::log2(aspect)));
    if (l < lo) l = lo;
    if (l > hi) l = hi;
    return (uint32_t)l;
}

// ------------------------------------------------------------------------------------------------ tile shape

struct TileRequest {
    uint64_t W = 0, H = 0, rows = 0; // frame, and the rows this launch traces
    uint32_t ns = 0;                 // spheres
    double aspect = 1.0;             // horizontal / vertical extent of a pixel on the view plane
    int n_cu = 0;
    bool cull = false;               // the culling (binned) kernel family
    int opt_subtiles = 0;            // RTX_OPT_SUBTILES (0 = choose)
    int opt_tile_log2w = 0;          // RTX_OPT_TILE_LOG2_W (0 = choose)
    int opt_refine = -1;             // RTX_OPT_REFINE (-1 auto)
    bool view_dense = false;         // the launches before this one saw long candidate lists (ViewDensity): plan as for a dense scene
    bool in_flight = false;          // the caller keeps frames in flight on several streams: throughput counts, not one launch's latency
};

struct TileShape {
    uint32_t lw = 6, lnx = 0, nsub = 1; // sub-tile 2^lw x 2^(8-lw); macro tile 2^lnx sub-tiles wide, nsub >> lnx high
    uint32_t mw = 64, mh = 4;           // macro tile, pixels
    uint32_t grid_x = 1, grid_y = 1;    // workgroups
    bool refine = false;                // per-wave refinement kernels (dense scenes)
    bool dense = false;                 // planned as a dense scene (by its density, or because the view is locally dense)
    uint64_t tiles256 = 0;              // 256-pixel tiles in the launch
};

constexpr double kDenseScene = 0.004; // spheres per pixel from which a scene counts as dense

// Sub-tiles (256 pixels each) per workgroup, sub-tile shape and macro-tile layout: as square as possible on the view
// plane, so that the culling pyramid is tight.  More sub-tiles amortise the per-workgroup set-up (tables, frustum,
// staging) but leave fewer workgroups to fill the CUs and, with culling, lengthen the candidate lists.  Measured
// (profiles/r01_d_subtiles.txt, r03_n): brute kernel 1 / 4 / 8 at 400x150 / 1080p / 4K; binned kernel 4 (5: one dispatch round) at
// 1080p, 8 from 4K on, 2 for dense scenes.
inline TileShape plan_tiles(const TileRequest& q)
{
    TileShape t;
    t.tiles256 = (q.W * q.rows + 255u) / 256u;
    const double density = q.view_dense ? kDenseScene : ((q.W && q.H) ? (double)q.ns / ((double)q.W * (double)q.H) : 0.0);
    int nsub;
    if (q.opt_subtiles) {
        nsub = q.opt_subtiles;
    } else if (q.cull) {
        nsub = density >= kDenseScene ? 2 : (t.tiles256 >= 24000u ? 8 : 4); // (config 3, 32 400 tiles: 90.3 -> 85.2 us per frame in flight with 8, alone the same)
        // A scene that is dense by its numbers, frames in flight: 4 sub-tiles.  The kernel is then bound by VALU issue (config 5:
        // 0.91 of what the SIMDs can issue), and a workgroup's set-up -- tile, cell list, tables, pyramids, staging: 226 of 826 VALU
        // instructions per pixel-wave with 2 sub-tiles -- is shared by twice the pixels: 826 -> 762 instructions, 32.6 -> 28.8 us
        // per frame with 6 in flight.  One launch alone is bound by the workgroups' latency instead and loses (41.6 -> 43.9 us: half
        // as many workgroups, each twice as long), so a caller on a single stream keeps 2.  (r04, profiles/r04_d_c5_plan.md)
        const bool dense_in_flight = density >= kDenseScene && !q.view_dense && q.in_flight;
        // One dispatch round: up to 1080p (and for the row slabs of a sharded frame) take the smallest count that lets
        // every workgroup be resident at once -- fewer, larger workgroups would leave CUs short of waves (a 135-row
        // slab: 14.0 us with 4 sub-tiles, 9.5 with 1; 270 rows: 17.1 -> 11.9 with 2), more would need a second round;
        // and with the tiles dealt so that every CU carries the same work one 1080p launch alone takes 25.4 instead of
        // 29.7 us (profiles/r02_f_cu_balance.md, tools/slab_shapes_gpu.py).  1080p: 8100 tiles / 1792 slots -> 5.
        if (nsub == 4 && q.n_cu > 0) {
            const uint64_t slots = (uint64_t)kResidentPerCU * (uint64_t)q.n_cu;
            const uint64_t one_round = (t.tiles256 + slots - 1) / slots;
            if (one_round <= 6) nsub = (int)(one_round ? one_round : 1);
        }
        if (dense_in_flight) nsub = 4;
    } else {
        nsub = 1;
        while (nsub < 8 && t.tiles256 >= (uint64_t)nsub * 4000u) nsub *= 2; // keep about 2000 workgroups or more
    }
    uint32_t lw;
    if (q.opt_tile_log2w) {
        lw = (uint32_t)q.opt_tile_log2w;
    } else if (!q.cull) {
        lw = 6u; // brute: 64x4, a wave writes one contiguous row span
    } else {
        // (at least 8 pixels wide: a wave's row segment is then 160 bytes of records.  At 8K the reference camera's pixels are
        // 24 : 1 and the squarest sub-tile would be 4 x 64: tightest culling, but 80-byte segments -- config 4 243 -> 207 us per
        // frame, config 3 91.5 -> 90.0, with 8 wide)
        lw = squarest_log2w(q.aspect, 256, 3, 6);
    }
    // macro tile of 256*nsub pixels, nx = 2^lnx sub-tiles wide and nsub/nx high, at most 128 x 128.  A sub-tile count
    // that is not a power of two is stacked vertically (nx = 1).
    uint32_t lnx = 0;
    if ((nsub & (nsub - 1)) == 0) {
        int lsub = 0;
        while ((1 << lsub) < nsub) lsub++;
        const uint32_t lmw = squarest_log2w(q.aspect, 256 * nsub, (int)lw, (int)lw + lsub);
        lnx = lmw - lw;
        while (lw + lnx > 7u) lnx--;                           // width  <= 128
        while ((8u - lw) + ((uint32_t)lsub - lnx) > 7u) lnx++; // height <= 128
    } else {
        while ((256u >> lw) * (uint32_t)nsub > 128u && lw < 6u) lw++; // keep the stack within 128 rows
    }
    t.lw = lw;
    t.lnx = lnx;
    t.nsub = (uint32_t)nsub;
    t.mw = (1u << lw) << lnx;
    t.mh = ((256u >> lw) * (uint32_t)nsub) >> lnx;
    t.grid_x = (uint32_t)((q.W + t.mw - 1) / t.mw);
    t.grid_y = (uint32_t)((q.rows + t.mh - 1) / t.mh);
    // dense scenes: every wave narrows the workgroup's candidate list to its own 64 pixels before scanning it
    const bool want = q.opt_refine == 1 || (q.opt_refine < 0 && density >= kDenseScene);
    t.refine = q.cull && want && nsub <= 4 && t.mw <= 64u && t.mh <= 64u; // (the REFINE kernels' tables: 64 x 64)
    t.dense = density >= kDenseScene;
    return t;
}

// Workgroups of the trace kernels the device holds at once.
inline uint64_t resident_slots(int n_cu) { return (uint64_t)kResidentPerCU * (uint64_t)(n_cu > 0 ? n_cu : 0); }

// ------------------------------------------------------------------------------------------------ coarse cells

#ifndef RTX_CELLS_LARGE
#define RTX_CELLS_LARGE 1024
#endif
#ifndef RTX_CELLS_SMALL
#define RTX_CELLS_SMALL 256
#endif
#ifndef RTX_BIN_WGS
#define RTX_BIN_WGS 1024
#endif
#ifndef RTX_BIN_MIN_ITEMS
#define RTX_BIN_MIN_ITEMS 1024 // spheres a binning workgroup walks at least (two staging steps)
#endif

struct CellGrid {
    uint32_t gx = 0, gy = 0;           // a cell is 2^gx x 2^gy macro tiles
    uint32_t cells_x = 1, cells_y = 1, n_cells = 1, n_blocks = 1;
    uint32_t cap = 1;                  // entries per cell list
    uint32_t splits = 1;               // binning workgroups per block of 4 x 4 cells
    uint32_t cell_w = 0, cell_h = 0;   // pixels
};

// Two-level culling for large scenes: coarse cells of 2^gx x 2^gy macro tiles -- about 256 of them, about 1024 from
// 16384 spheres on, where shorter cell lists save the trace workgroups a staging step -- binned by ONE pre-pass launch
// (rtx_bin_cells: blocks of 4 x 4 cells, then the cells of each block).
inline CellGrid plan_cells(const TileShape& t, uint32_t ns, double aspect, int64_t opt_cell_capacity, uint32_t cap_floor = 0)
{
    CellGrid c;
    const uint32_t tiles_x = t.grid_x, tiles_y = t.grid_y;
    const uint64_t want_cells = ns >= 16384u ? (uint64_t)RTX_CELLS_LARGE : (uint64_t)RTX_CELLS_SMALL;
    uint32_t gx = 0, gy = 0;
    // grow the cell, keeping it square-ish on the view plane, until about want_cells cells remain
    while ((uint64_t)((tiles_x + (1u << gx) - 1) >> gx) * ((tiles_y + (1u << gy) - 1) >> gy) > want_cells) {
        const double wcell = (double)(t.mw << gx) * aspect, hcell = (double)(t.mh << gy);
        if (wcell <= hcell && ((tiles_x + (1u << gx) - 1) >> gx) > 1u) gx++;
        else if (((tiles_y + (1u << gy) - 1) >> gy) > 1u) gy++;
        else gx++;
    }
    c.gx = gx;
    c.gy = gy;
    c.cells_x = (tiles_x + (1u << gx) - 1) >> gx;
    c.cells_y = (tiles_y + (1u << gy) - 1) >> gy;
    c.n_cells = c.cells_x * c.cells_y;
    c.n_blocks = ((c.cells_x + 3u) >> 2) * ((c.cells_y + 3u) >> 2);
    c.cell_w = t.mw << gx;
    c.cell_h = t.mh << gy;
    // Entries per cell list: four times the share of a uniform scene plus a floor, so that the scratch is O(spheres) --
    // 4 ns + 1024 cells words -- instead of cells x ns.  A cell that needs more (a clustered scene) falls back to
    // staging the whole scene: slower, never wrong.
    // `cap_floor`: what the longest list of this grid was seen to need (cell_capacity_wanted below): a view that packs the
    // scene into a few cells grows the lists instead of falling off the whole-scene cliff frame after frame.
    uint32_t cap = opt_cell_capacity > 0 ? (uint32_t)opt_cell_capacity : (uint32_t)(4ull * ns / c.n_cells) + 1024u;
    if (opt_cell_capacity <= 0 && cap < cap_floor) cap = cap_floor;
    if (cap > ns) cap = ns; // a list never holds more than the scene
    if (cap == 0) cap = 1;
    c.cap = cap;
    // the sphere array is split over several workgroups per block so that the pre-pass fills the chip: about 1024
    // workgroups in all, each with at least RTX_BIN_MIN_ITEMS spheres
    uint32_t splits = (uint32_t)RTX_BIN_WGS / c.n_blocks;
    const uint32_t max_splits = (ns + (uint32_t)RTX_BIN_MIN_ITEMS - 1u) / (uint32_t)RTX_BIN_MIN_ITEMS;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1u) splits = 1u;
    c.splits = splits;
    return c;
}

// Static dispatch order of a two-level grid: order[b] = bx | by << 16 of the macro tile that dispatch position b (linear
// block index) renders.  The blocks b, b + 8, b + 16, ... share an XCD under the round-robin placement observed on MI355X
// (b % 8 labels the blocks that share one, not the XCD's id); they are given WHOLE CELLS: cell (cx, cy) belongs to label
// (cx + 3 cy) % 8, so that a cell's list and the spheres on it are fetched into one XCD's L2 instead of into as many as the
// cell has tiles, while every XCD still samples cells from all over the frame (a contiguous band per XCD was measured
// first: a view that packs the scene into one side of the frame then leaves seven XCDs waiting for the eighth).  A label's
// tiles beyond its share of positions (cells differ in size at the grid's edge) take the positions other labels leave
// empty.  A bijection by construction; any permutation renders the same frame: a wrong placement guess costs speed only.
inline void xcd_cell_order(uint32_t grid_x, uint32_t grid_y, uint32_t gx, uint32_t gy, uint32_t* order)
{
    const uint32_t n = grid_x * grid_y, nxcd = 8u;
    const uint32_t cw = 1u << gx, ch = 1u << gy;
    const uint32_t cells_x = (grid_x + cw - 1u) >> gx, cells_y = (grid_y + ch - 1u) >> gy;
    std::vector<uint32_t> mine[8], spill;
    for (uint32_t cy = 0; cy < cells_y; cy++) {
        for (uint32_t cx = 0; cx < cells_x; cx++) {
            std::vector<uint32_t>& v = mine[(cx + 3u * cy) & 7u];
            for (uint32_t ty = cy * ch; ty < (cy + 1u) * ch && ty < grid_y; ty++) {
                for (uint32_t tx = cx * cw; tx < (cx + 1u) * cw && tx < grid_x; tx++) v.push_back(tx | (ty << 16));
            }
        }
    }
    std::vector<uint32_t> holes;
    for (uint32_t x = 0; x < nxcd; x++) {
        const uint32_t slots = n / nxcd + (x < n % nxcd ? 1u : 0u); // positions x, x + 8, x + 16, ...
        for (uint32_t k = 0; k < slots; k++) {
            if (k < mine[x].size()) order[k * nxcd + x] = mine[x][k];
            else holes.push_back(k * nxcd + x);
        }
        for (size_t k = slots; k < mine[x].size(); k++) spill.push_back(mine[x][k]);
    }
    for (size_t i = 0; i < holes.size() && i < spill.size(); i++) order[holes[i]] = spill[i];
}

// Capacity feedback: `seen` is the longest list any cell of the grid has needed so far (the binning pass keeps counting
// past the capacity and reports the maximum), `cap` the capacity in use.  Returns the floor to plan with from now on:
// unchanged while the lists have a fifth to spare, else 1.5 x seen rounded up to 256 entries (a growth step reallocates the
// lists behind a device synchronisation: few, large steps).
inline uint32_t cell_capacity_wanted(uint32_t seen, uint32_t cap, uint32_t floor_now)
{
    if ((uint64_t)seen * 5u <= (uint64_t)cap * 4u) return floor_now;
    const uint64_t want = ((uint64_t)seen * 3u / 2u + 64u + 255u) / 256u * 256u;
    const uint32_t w = want > 0xffffffffull ? 0xffffffffu : (uint32_t)want;
    return w > floor_now ? w : floor_now;
}

// ------------------------------------------------------------------------------------------------ direction order

// The order of the direction-sorted copies of the sphere arrays (KArgs::sph_geom for the trace kernels): positions by a Morton
// code of the direction -- azimuth atan2(dx, dz), elevation asin(dy / |d|), 16 bits each over their range -- in which a sphere's
// centre lies from `origin`; equal codes keep creation order.  centres: x y z (stride `stride` floats).  order[p] = sphere index
// at position p, pos_of[k] = position of sphere k.  NaN / zero-length directions sort as direction (0, 0).
inline void direction_order(const float* centres, size_t stride, uint32_t n, const float origin[3], std::vector<uint32_t>& order, std::vector<uint32_t>& pos_of)
{
    std::vector<float> az(n), el(n);
    float az_lo = 1e30f, az_hi = -1e30f, el_lo = 1e30f, el_hi = -1e30f;
    for (uint32_t k = 0; k < n; k++) {
        const float dx = centres[k * stride + 0] - origin[0], dy = centres[k * stride + 1] - origin[1], dz = centres[k * stride + 2] - origin[2];
        const float len = std::sqrt(dx * dx + dy * dy + dz * dz);
        float a = std::atan2(dx, dz), e = len > 0.0f ? std::asin(std::fmax(-1.0f, std::fmin(1.0f, dy / len))) : 0.0f;
        if (!(a == a)) a = 0.0f;
        if (!(e == e)) e = 0.0f;
        az[k] = a;
        el[k] = e;
        az_lo = std::fmin(az_lo, a);
        az_hi = std::fmax(az_hi, a);
        el_lo = std::fmin(el_lo, e);
        el_hi = std::fmax(el_hi, e);
    }
    const float sa = az_hi > az_lo ? 65535.0f / (az_hi - az_lo) : 0.0f, se = el_hi > el_lo ? 65535.0f / (el_hi - el_lo) : 0.0f;
    auto spread = [](uint32_t x) {
        x &= 0xffffu;
        x = (x | (x << 8)) & 0x00ff00ffu;
        x = (x | (x << 4)) & 0x0f0f0f0fu;
        x = (x | (x << 2)) & 0x33333333u;
        x = (x | (x << 1)) & 0x55555555u;
        return x;
    };
    std::vector<uint64_t> keyed(n);
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t qa = (uint32_t)((az[k] - az_lo) * sa), qe = (uint32_t)((el[k] - el_lo) * se);
        keyed[k] = ((uint64_t)(spread(qa) | (spread(qe) << 1)) << 32) | k;
    }
    std::sort(keyed.begin(), keyed.end());
    order.resize(n);
    pos_of.resize(n);
    for (uint32_t p = 0; p < n; p++) {
        order[p] = (uint32_t)(keyed[p] & 0xffffffffu);
        pos_of[order[p]] = p;
    }
}

// ------------------------------------------------------------------------------------------------ locally dense views

// A sparse scene can look dense from where the camera stands: config 2's 1024 spheres seen along the scene's long axis put
// 95 candidates on one 16 x 80-pixel macro tile (default view: 9 at most), and a launch alone takes as long as its slowest
// workgroup: 63 us against 27.  The dense-scene configuration (2 sub-tiles per workgroup, two-level culling, per-wave
// refinement) renders those views in 23-27 us but the default view in 31 (profiles/r03_i_worst_view_configs.txt).  So the
// choice follows what the launches see: every workgroup with a long candidate list reports its length (atomicMax into a
// word that is copied to the host once per epoch of 8 launches); past kHeavy candidates on a macro tile of the sparse plan the
// following launches use the dense one, and return when kCalm epochs in a row saw no list of kLightDense entries on the dense
// plan's smaller tiles.  The observation is a dozen launches old: it follows a camera that moves, it cannot follow a cut.
// Longest list by yaw off the default view, sparse plan / dense plan (profiles/r03_k_candidates_by_plan.txt): 0: 9 / 9,
// 0.2: 20 / 14, 0.4: 36 / 19, 0.6: 52 / 34, 1.0: 81 / 54, 1.4: 95 / 58 -- the plan turns dense near 0.3 rad and sparse again
// under 0.15.
class ViewDensity {
public:
    static constexpr uint32_t kHeavy = 28u;      // sparse plan: longest list from which the view counts as locally dense ...
    static constexpr uint32_t kReportSparse = 12u; // ... workgroups report lists from this length on
    static constexpr uint32_t kLightDense = 12u; // dense plan (macro tiles 2.5 times smaller): workgroups report lists from this length
                                                 // on, and the view counts as sparse again when kCalm epochs in a row reported none
    static constexpr int kCalm = 3;

    bool dense() const { return dense_; }
    uint32_t report_from() const { return dense_ ? kLightDense : kReportSparse; }
    void reset()
    {
        dense_ = false;
        calm_ = 0;
    }
    // `longest`: the longest candidate list the launches of one epoch reported (0: none reached report_from()).  An observation
    // made under the other plan (the copy was in flight when the plan changed) reads safely: a sparse-plan value keeps a dense
    // plan dense, a dense-plan value is below kHeavy.  Returns true when the plan changes.
    bool observe(uint32_t longest)
    {
        if (!dense_) {
            if (longest >= kHeavy) {
                dense_ = true;
                calm_ = 0;
                return true;
            }
            return false;
        }
        if (longest < kLightDense) {
            if (++calm_ >= kCalm) {
                dense_ = false;
                calm_ = 0;
                return true;
            }
        } else {
            calm_ = 0;
        }
        return false;
    }

private:
    bool dense_ = false;
    int calm_ = 0;
};

// ------------------------------------------------------------------------------------------------ dispatch order

// When a launch of a tile grid leaves work estimates, when a new dispatch order is derived from them, and which order
// (if any) a launch runs under.  One per (stream, tile grid).  Any permutation renders the same frame: only speed
// depends on this.
//
// Grids of one dispatch round (`one_round`): rtx_balance_tiles runs beside the frames on the library's side stream.
// The launch a pass reads is followed by kLag - 1 launches that keep the old order and leave no estimates (the pass may
// still be reading them); the launch after those waits for the pass -- by then long done -- and switches to the order it
// wrote, in the other half of the order buffer.  An order is used only while the view is within kNear of the view it
// was measured on (a stale heaviest-first order is a random order, which costs 3-5 % with frames in flight), refreshed
// as soon as it is stale, and not derived at all while the camera moves faster than kNear per 16 frames.
// Other grids: rtx_order_tiles is queued on the render stream itself, after the 1st and 2nd launch and then every
// period-th.
class DispatchOrder {
public:
    static constexpr int kLag = 3; // (at 2 the wait still stalls a 25 us frame: the pass starts 10 us after the launch it reads ends)
    static constexpr float kNear = 0.004f; // radians (rotation entries) and tenths of a unit of camera position

    struct Decision {
        bool switch_order = false;   // before the launch: wait for the pending pass, then use the half it wrote
        bool leave_estimates = false; // the launch stores its estimates and times (tile_cost != nullptr)
        bool use_order = false;      // the launch reads the order in half `half`
        int half = 0;
        bool sort_now = false;       // after the launch: derive a new order from what it left
        bool balance = false;        // ... with rtx_balance_tiles on the side stream (one-round grids), else rtx_order_tiles in line
        bool prev_is_order = false;  // the launch ran under an order (the pass must map positions to tiles through it)
        bool have_factor = false;    // the per-tile correction factors carry over
    };

    void reset()
    {
        have_order_ = have_factor_ = false;
        frames_ = 0;
        cur_ = 0;
        pending_ = 0;
    }
    // A pass may still be queued when the grid changes: the caller waits for it, then calls reset().
    bool pass_pending() const { return pending_ != 0; }
    int current_half() const { return cur_; }
    bool have_order() const { return have_order_; }
    uint64_t frames() const { return frames_; }

    // opt_tile_order: RTX_OPT_TILE_ORDER (-1 auto, k > 0 refresh period); drift: how far any sphere can have moved
    // since the context was created (scene edits add 1e3).
    Decision next(const View& v, double drift, bool one_round, int64_t opt_tile_order)
    {
        Decision d;
        bool skip_estimates = false;
        if (one_round) {
            if (pending_ >= kLag) {
                d.switch_order = true;
                cur_ ^= 1;
                have_order_ = true;
                have_factor_ = true;
                pending_ = 0;
                order_view_ = pending_view_;
                order_drift_ = pending_drift_;
            } else if (pending_ >= 1) {
                pending_++;
                skip_estimates = true;
            }
        }
        const uint64_t period = opt_tile_order > 0 ? (uint64_t)opt_tile_order : (one_round ? 64u : 16u);
        d.use_order = have_order_;
        if (one_round) {
            // (spheres that rtx_update_objects moves count like a camera that moves: by the farthest any can have gone)
            const float step = std::fmax(view_distance(v, last_view_), 0.1f * (float)(drift - last_drift_));
            const float moved = std::fmax(view_distance(v, order_view_), 0.1f * (float)(drift - order_drift_));
            last_view_ = v;
            last_drift_ = drift;
            const bool too_fast = step * 16.0f > kNear;
            const bool stale = have_order_ && moved > kNear;
            d.use_order = have_order_ && !stale;
            // the corrections settle over a dozen passes: one every fourth launch at first, then every period-th, or as
            // soon as the order has gone stale
            d.sort_now = !skip_estimates && !too_fast && pending_ == 0 && (frames_ < 64 || (frames_ + 1) % period == 0 || stale);
            if (d.sort_now) {
                pending_view_ = v;
                pending_drift_ = drift;
            }
            d.balance = true;
        } else {
            d.sort_now = frames_ < 2 || (frames_ + 1) % period == 0;
        }
        d.leave_estimates = d.sort_now; // estimates and times: only the launches a pass follows leave them
        d.half = cur_;
        d.prev_is_order = d.use_order;   // the pass maps dispatch positions to tiles through the order the launch really ran under
        d.have_factor = have_factor_;    // (per tile, whatever order they were learnt under)
        return d;
    }

    // After the launch (and the pass, if any) has been queued.
    void launched(const Decision& d)
    {
        frames_++;
        if (d.sort_now && d.balance) {
            pending_ = 1;
        } else if (d.sort_now) {
            have_order_ = true; // rtx_order_tiles wrote the half in use, in stream order
        }
    }

private:
    View last_view_, order_view_, pending_view_; // of the last launch, of the launch the order in use was measured on, of the launch the pass in flight reads
    double last_drift_ = 0.0, order_drift_ = 0.0, pending_drift_ = 0.0;
    int cur_ = 0;
    int pending_ = 0; // 0 = none; 1 = a pass was queued after the last launch; 2.. = launches since
    bool have_order_ = false, have_factor_ = false;
    uint64_t frames_ = 0;
};

// ------------------------------------------------------------------------------------------------ cell-list reuse

// Coarse-cell lists (rtx_bin_cells) stay valid while the camera and the scene have moved less than the margins the lists
// were built with.  Soundness (one more term in DESIGN.md's "Culling soundness"): let the lists be built for camera
// (o, M) with every sphere's culling margin grown to
//     margin' = margin(r, |O| + delta) + delta + theta (|O| + delta + margin(r, |O| + delta)).
// A ray of a later camera (o', M') for the same pixel has |o' - o| <= delta (sphere motion counted in) and a unit
// direction d' with |d' - d| <= theta, where d (the old direction) satisfies n.d >= 0 for the five inward plane normals
// of the cell's pyramid.  If the fp32 test reports a hit, the ray passes within R' <= margin(r, |O'|) of the centre c at a
// point p = o' + t d' with -3u|O'| <= t <= |O'| + R'; then
//     n.(c - o) = n.(c - p) + t n.d' + n.(o' - o) >= -R' - theta (|O'| + R') - 3u|O'| - delta >= -margin',
// so the sphere is in the cell's list.  For rotations |d' - d| <= ||M' - M||_F + 2 eps when both matrices are orthonormal
// to within eps (then |M p| = |p| (1 +- eps)); other matrices get no rotation budget (the lists are then reused only
// for an identical matrix).
struct CellCamera {
    View view;
    float e1 = 0, e2 = 0;
    uint64_t W = 0, H = 0;
    double drift = 0.0;       // ctx->scene_drift
    uint64_t scene_gen = 0;   // bumped by every scene edit
};

struct CellKey { // the cell grid and everything its lists depend on besides the camera pose
    uint64_t W = 0, H = 0, row0 = 0, rows = 0;
    uint32_t lw = 0, lnx = 0, nsub = 0, gx = 0, gy = 0, cap = 0, ns = 0;
    bool operator==(const CellKey& o) const
    {
        return W == o.W && H == o.H && row0 == o.row0 && rows == o.rows && lw == o.lw && lnx == o.lnx && nsub == o.nsub && gx == o.gx &&
               gy == o.gy && cap == o.cap && ns == o.ns;
    }
};

struct CellBudget {
    float theta = 0.0f; // bound on |d' - d|
    float delta = 0.0f; // bound on |o' - o| + sphere motion
};

// Largest singular value of a 3x3 matrix (row-major), from the largest eigenvalue of A^T A in closed form (trigonometric
// solution of the symmetric 3x3 characteristic polynomial, in double), grown by 1e-6 relative + 1e-12: an upper bound on
// |A v| for unit v.  (For the difference of two rotations this is 2 sin(phi/2); the Frobenius norm would say sqrt(2)
// times that.)
inline double spectral_norm3(const double a[9])
{
    double b[3][3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            b[i][j] = 0.0;
            for (int k = 0; k < 3; k++) b[i][j] += a[3 * k + i] * a[3 * k + j];
        }
    }
    const double p1 = b[0][1] * b[0][1] + b[0][2] * b[0][2] + b[1][2] * b[1][2];
    const double tr = b[0][0] + b[1][1] + b[2][2];
    double lmax;
    if (!(tr == tr) || !(p1 == p1)) return 1.0e30;
    if (p1 <= 1.0e-300) {
        lmax = std::fmax(b[0][0], std::fmax(b[1][1], b[2][2]));
    } else {
        const double q = tr / 3.0;
        const double p2 = (b[0][0] - q) * (b[0][0] - q) + (b[1][1] - q) * (b[1][1] - q) + (b[2][2] - q) * (b[2][2] - q) + 2.0 * p1;
        const double pp = std::sqrt(p2 / 6.0);
        double c[3][3];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) c[i][j] = (b[i][j] - (i == j ? q : 0.0)) / pp;
        }
        double r = 0.5 * (c[0][0] * (c[1][1] * c[2][2] - c[1][2] * c[2][1]) - c[0][1] * (c[1][0] * c[2][2] - c[1][2] * c[2][0]) +
                          c[0][2] * (c[1][0] * c[2][1] - c[1][1] * c[2][0]));
        r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
        lmax = q + 2.0 * pp * std::cos(std::acos(r) / 3.0);
    }
    // never below the largest column norm (a lower bound of the spectral norm that is exact to rounding): guards the
    // closed form's cancellation for nearly singular A^T A
    lmax = std::fmax(lmax, std::fmax(b[0][0], std::fmax(b[1][1], b[2][2])));
    return std::sqrt(lmax > 0.0 ? lmax : 0.0) * (1.0 + 1.0e-6) + 1.0e-12;
}

// ||M^T M - I||_F of the upper 3x3: how far from orthonormal.
inline float orthonormal_defect(const View& v)
{
    double s = 0.0;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            double g = 0.0;
            for (int k = 0; k < 3; k++) g += (double)v.rot[3 * k + i] * (double)v.rot[3 * k + j];
            g -= (i == j) ? 1.0 : 0.0;
            s += g * g;
        }
    }
    const double r = std::sqrt(s);
    return r == r ? (float)r : 1.0e30f;
}

// What a camera `now` has used of the budgets of lists built for `built`: {bound on |d' - d|, bound on |o' - o| + drift}.
// Infinite when anything else the lists depend on differs.
inline CellBudget cell_motion(const CellCamera& built, const CellCamera& now)
{
    CellBudget m;
    const float inf = 1.0e30f;
    if (built.W != now.W || built.H != now.H || built.e1 != now.e1 || built.e2 != now.e2 || built.scene_gen != now.scene_gen ||
        !(now.drift >= built.drift)) {
        m.theta = m.delta = inf;
        return m;
    }
    double p = 0.0, dm[9];
    bool same_rot = true;
    for (int k = 0; k < 9; k++) {
        dm[k] = (double)now.view.rot[k] - (double)built.view.rot[k];
        same_rot = same_rot && (now.view.rot[k] == built.view.rot[k]);
    }
    for (int k = 0; k < 3; k++) {
        const double e = (double)now.view.pos[k] - (double)built.view.pos[k];
        p += e * e;
    }
    if (same_rot) {
        m.theta = 0.0f; // the very same matrix: the very same directions, orthonormal or not
    } else {
        const float eps = std::fmax(orthonormal_defect(built.view), orthonormal_defect(now.view));
        m.theta = eps <= 1.0e-3f ? (float)(spectral_norm3(dm) * 1.0001 + 2.0 * eps + 1.0e-6) : inf;
    }
    const double dp = std::sqrt(p) * 1.0001 + (now.drift - built.drift);
    m.delta = (dp == dp && dp < 1.0e30) ? (float)dp : inf;
    if (m.delta > 0.0f) m.delta = m.delta * 1.0001f + 1.0e-30f;
    if (!(m.theta == m.theta)) m.theta = inf;
    return m;
}

// Two sets of lists (double-buffered) and the rule for using, prefetching and rebuilding them.
class CellCachePolicy {
public:
    float frames_per_build = 16.0f; // a budget lasts about this many frames of the current motion (8: lists a fifth shorter, but twice the
                                    // rebuilds -- config 5 with a turning camera and 6 frames in flight: 52.9 us per frame against 42.5)
    float prefetch_at = 0.5f;      // the fraction of a budget used from which the next lists are built ahead of time
    static constexpr float kCapFraction = 0.25f;   // ... but no more than this fraction of a cell's smaller angular extent
    static constexpr float kMinFrames = 4.0f;      // ... and reuse is given up when that would last fewer frames than this
    static constexpr int kStillFrames = 2;         // still_only: launches in a row without motion before lists are built

    enum Action {
        kUse = 0,       // the lists of slot `slot` cover this camera (slot may differ from the one used last: a finished prefetch)
        kBuild = 1,     // none does: bin for this camera into slot `slot` with `budget`, in line, then use it
        kPerFrame = 2,  // the camera moves too fast for reuse to pay: bin per frame into the stream's own scratch, as without a cache
        kSkip = 3       // (still_only) no lists for this launch: its workgroups stage the whole scene
    };
    struct Decision {
        Action action = kPerFrame;
        int slot = 0;
        CellBudget budget;        // kBuild: what to build with
        bool prefetch = false;    // additionally: start building slot `prefetch_slot` for this camera on the side stream
        int prefetch_slot = 0;
        CellBudget prefetch_budget;
    };

    void invalidate()
    {
        slot_[0].valid = slot_[1].valid = false;
        have_last_ = false;
    }
    bool slot_valid(int s) const { return slot_[s].valid; }
    const CellBudget& slot_budget(int s) const { return slot_[s].budget; }

    // cell_tan_w / cell_tan_h: a cell's extent on the view plane (tangent units); the cap on the rotation budget.
    // ready[s]: the build of slot s has finished on the device (a slot still being built ahead of time is used only if
    // nothing finished covers the camera: switching to it at once would make every render stream wait for the build and
    // take the overlap away).
    //
    // still_only: scenes too small for the pre-pass to pay frame by frame (under 2048 spheres: whole-scene staging costs a
    // workgroup about what a pre-pass per frame costs the launch) still gain from lists that cost nothing: while the camera
    // and the scene REST, exact lists are built once and used until something moves (config 2: 18.7 -> 17.4 us per frame
    // in flight, 25.8 -> 24.6 alone); a launch whose camera or scene has moved since the launch before gets kSkip.
    Decision decide(const CellKey& key, const CellCamera& cam, double cell_tan_w, double cell_tan_h, const bool ready[2], bool still_only = false)
    {
        Decision d;
        // the step from the previous frame (whatever path that one took) sets the budgets of anything built now
        // (after a scene edit or a change of the projection there is no step to speak of: budgets of zero, as for a first frame)
        CellBudget step;
        if (have_last_ && last_.W == cam.W && last_.H == cam.H && last_.e1 == cam.e1 && last_.e2 == cam.e2 && last_.scene_gen == cam.scene_gen &&
            cam.drift >= last_.drift) {
            step = cell_motion(last_, cam);
        }
        const bool comparable = have_last_ && last_.W == cam.W && last_.H == cam.H && last_.e1 == cam.e1 && last_.e2 == cam.e2 &&
                                last_.scene_gen == cam.scene_gen && cam.drift >= last_.drift;
        still_frames_ = (comparable && step.theta == 0.0f && step.delta == 0.0f) ? still_frames_ + 1 : 0;
        last_ = cam;
        have_last_ = true;
        if (still_only) {
            for (int s = 0; s < 2; s++) {
                if (!slot_[s].valid || !(slot_[s].key == key)) continue;
                const CellBudget m = cell_motion(slot_[s].built_for, cam);
                if (m.theta == 0.0f && m.delta == 0.0f) {
                    d.action = kUse;
                    d.slot = s;
                    active_ = s;
                    return d;
                }
            }
            if (still_frames_ < kStillFrames) {
                d.action = kSkip;
                return d;
            }
            d.action = kBuild;
            d.slot = slot_[active_].valid ? (active_ ^ 1) : active_;
            d.budget = CellBudget();
            slot_[d.slot].valid = true;
            slot_[d.slot].key = key;
            slot_[d.slot].built_for = cam;
            slot_[d.slot].budget = d.budget;
            active_ = d.slot;
            return d;
        }
        const float theta_cap = (float)(kCapFraction * std::fmin(cell_tan_w, cell_tan_h));
        // ... capped; a camera that uses the cap up in fewer than kMinFrames frames is too fast for reuse to pay
        CellBudget want;
        want.theta = std::fmin(frames_per_build * step.theta, theta_cap);
        want.delta = frames_per_build * step.delta;
        const bool too_fast = !(kMinFrames * step.theta <= theta_cap) || !(want.delta < 1.0e29f);

        int best = -1;
        float best_used = 2.0f;
        bool best_ready = false;
        for (int s = 0; s < 2; s++) {
            if (!slot_[s].valid || !(slot_[s].key == key)) continue;
            const CellBudget m = cell_motion(slot_[s].built_for, cam);
            if (!(m.theta <= slot_[s].budget.theta) || !(m.delta <= slot_[s].budget.delta)) continue;
            // fraction of the budget used: prefer finished lists, then the fresher ones
            const float ut = slot_[s].budget.theta > 0.0f ? m.theta / slot_[s].budget.theta : 0.0f;
            const float ud = slot_[s].budget.delta > 0.0f ? m.delta / slot_[s].budget.delta : 0.0f;
            const float used = std::fmax(ut, ud);
            if (best < 0 || (ready[s] && !best_ready) || (ready[s] == best_ready && used < best_used)) {
                best_used = used;
                best = s;
                best_ready = ready[s];
            }
        }
        if (best >= 0) {
            d.action = kUse;
            d.slot = best;
            active_ = best;
            // more than half used and the other slot does not already hold something fresher: rebuild there, beside the frames
            const int other = best ^ 1;
            if (best_used > prefetch_at && !too_fast) {
                bool other_fresher = false;
                if (slot_[other].valid && slot_[other].key == key) {
                    const CellBudget m = cell_motion(slot_[other].built_for, cam);
                    other_fresher = m.theta <= prefetch_at * slot_[other].budget.theta && m.delta <= prefetch_at * slot_[other].budget.delta;
                }
                if (!other_fresher) {
                    d.prefetch = true;
                    d.prefetch_slot = other;
                    d.prefetch_budget = want;
                    slot_[other].valid = true;
                    slot_[other].key = key;
                    slot_[other].built_for = cam;
                    slot_[other].budget = want;
                }
            }
            return d;
        }
        if (too_fast) {
            d.action = kPerFrame;
            return d;
        }
        d.action = kBuild;
        d.slot = active_ ^ 1; // the slot not used last: launches in flight may still read the other
        if (!slot_[active_].valid) d.slot = active_;
        d.budget = want;
        slot_[d.slot].valid = true;
        slot_[d.slot].key = key;
        slot_[d.slot].built_for = cam;
        slot_[d.slot].budget = want;
        active_ = d.slot;
        return d;
    }

private:
    struct Slot {
        bool valid = false;
        CellKey key;
        CellCamera built_for;
        CellBudget budget;
    };
    Slot slot_[2];
    int active_ = 0;
    CellCamera last_;
    bool have_last_ = false;
    int still_frames_ = 0;
};

} // namespace rtxplan
