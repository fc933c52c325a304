// rtx_kernels.h -- launch interface between the host API (rtx_api.cpp) and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// Kernel-side mode numbers = enum RenderingMode (RayTracingManager.h:21).
enum {
    RTX_K_BIT_ASCII = 0,
    RTX_K_BIT_PIXEL = 1,
    RTX_K_RGB_ASCII = 2,
    RTX_K_RGB_PIXEL = 3,
    RTX_K_RGB_NORMALS = 4,
    RTX_K_SDL = 5
};

// Kernel arguments, passed by value (they land in SGPRs through scalar loads).
struct KArgs {
    float m[12];          // first three rows of inverseVMatrix
    float ox, oy, oz;     // camPos
    float e1, e2, far;    // element1, element2, camFarDist
    float fW, fH;         // (float)W, (float)H
    uint32_t W, H;
    uint32_t row0, row_end;   // rows traced by this launch (global row indices)
    uint32_t out_row_base;    // the row stored at out[0]
    uint32_t ns, np;          // spheres, planes
    uint32_t tile_log2w;      // a sub-tile is 2^lw x 2^(8-lw) pixels (one pixel per thread)
    uint32_t nsub;            // sub-tiles per workgroup (power of two)
    uint32_t sub_log2nx;      // a workgroup's macro tile is 2^lnx sub-tiles wide, nsub >> lnx high
    // Scene, SoA in HBM (creation order within each kind; .w of the colour arrays carries the
    // creation index across kinds as uint bits; the raw colours stay in host-visible arrays of the
    // context for rtx_scene_get_object):
    // What the TRACE kernels read, by position: the direction-sorted copies when there are some (scenes from 256 spheres,
    // RTX_OPT_SORTED_STORE) -- spheres ordered by a Morton code of the direction in which they lie from where the camera stood when
    // the scene was last edited, so that a cell's ~250 spheres sit in a few dozen 128-byte lines instead of 250 (config 5: 9.3 MB
    // of lines per launch in creation order, 2.2 MB sorted) -- else the scene arrays themselves, where position = sphere index.
    // Cell lists hold positions, candidates carry positions, the winner's records are fetched by position.
    const float4* sph_geom;   // cx cy cz r
    const float4* sph_od;     // R/255 G/255 B/255 gidx  (RayTracing.cu:144: colour / 255.0f, hoisted to upload time)
    const uint32_t* sph_sorted_idx; // position -> sphere index (creation order), looked up only to break an exact tie in t; nullptr: identity
    // What rtx_bin_cells reads: the scene array in creation order (in the sorted copy a block's spheres fall into one workgroup's
    // share), and the translation it writes the lists through (nullptr: identity).
    const float4* sph_scene_geom;
    const uint32_t* sph_pos_of;
    const float4* pl_a;       // px py pz width
    const float4* pl_b;       // nx ny nz height
    const float4* pl_od;      // R/255 G/255 B/255 gidx
    const uint8_t* grey;      // 256-byte grey lookup of the xterm-256 mapper
    // Two-level culling (large scenes): per coarse cell a list of sphere indices (cell_cap entries apart, any order) and
    // its count; a count above cell_cap means the list did not fit and the cell's workgroups stage the whole scene.
    // cell_list == nullptr: every workgroup stages the whole scene.
    const uint32_t* cell_list;
    const uint32_t* cell_count;
    uint32_t* cell_list_out;  // rtx_bin_cells writes these two ...
    uint32_t* cell_count_out;
    uint32_t* cell_count_zero; // ... and zeroes this one (the other counter buffer, for the next launch on the stream)
    uint32_t cell_log2gx, cell_log2gy; // a cell is 2^gx x 2^gy macro tiles
    uint32_t cells_x, cells_y;
    uint32_t cell_cap;        // entries per cell list
    // rtx_bin_cells only: the motion budget the lists are built with (rtx_plan.hpp, "cell-list reuse"): they stay valid for
    // cameras whose ray directions differ by at most bin_theta (chord of unit vectors) and whose position -- sphere
    // motion counted in -- by at most bin_delta from this launch's.  0, 0: this camera only.
    float bin_theta, bin_delta;
    uint32_t* cell_max_out;   // rtx_bin_cells: the longest list any cell has needed so far (atomicMax; past half the capacity only), or nullptr
    // Heaviest-first dispatch (speed only; any permutation of the macro tiles renders the same frame): tile_order[b] =
    // bx | by << 16 of the macro tile that workgroup b (linear block index, x fastest) renders, built by
    // rtx_order_tiles / rtx_balance_tiles from tile_cost, the work estimate every workgroup of an earlier launch left
    // for its tile: tile_cost[tile], followed by the workgroups' start and end times (100 MHz clock) by dispatch
    // position: tile_cost[n_tiles + b], tile_cost[2 n_tiles + b].  Either may be nullptr (identity order / no estimate
    // wanted).
    const uint32_t* tile_order;
    uint32_t* tile_cost;
    // View-density feedback for the host (rtx_plan.hpp, ViewDensity), culling kernels: workgroups whose candidate list holds at
    // least longest_from entries atomicMax its length into longest_list[longest_slot] (three words in rotation, slot = epoch
    // mod 3); the first workgroup zeroes the NEXT slot, which the launches of the next epoch fill -- never the previous one,
    // whose copy to the host may still be queued on another stream (an epoch only begins once the copy of the epoch before
    // the last has landed, so the slot being zeroed is always one whose copy is done).  nullptr: no feedback.
    uint32_t* longest_list;
    uint32_t longest_from, longest_slot;
    uint8_t* out;             // records of row out_row_base start here
    uint32_t refine;          // culling kernels: per-wave refinement of the candidate list (dense scenes; nsub <= 2)
    uint32_t compact;         // 1 = RTX_RENDER_COMPACT: out holds one 4-byte pixel word per pixel instead of a record;
                              // 2 = RTX_RENDER_VALUES: out holds 8 floats per pixel (distance, shadingValue, normal, colour)
    // (experiment build only: its extra arguments; nothing in the product build)
    // the culling pyramids' planes (rtx_plan.hpp, EdgeBasis): n_up(cy) = cy up_p + up_q and n_right(cx) = cx right_p + right_q,
    // the normals of a row / column edge that point up / right in the frame; the squared lengths |P|^2, |Qr|^2, |Qc|^2 of the
    // refusal test; the camera plane's unit normal (zero: none)
    float edge_up_p[3], edge_up_q[3], edge_right_p[3], edge_right_q[3];
    float edge_pp, edge_qrqr, edge_qcqc;
    float edge_fwd[3];
    // batched launches (rtx_trace_batch, KBatch): frames in the launch (0: a plain launch), the tile grid of ONE frame
    uint32_t batch_n, batch_gx, batch_gy;
#define RTX_X_SECTION_KARGS
#include "rtx_experiment.inc"
#undef RTX_X_SECTION_KARGS
};

// A batched trace launch (rtx_trace_batch): ONE grid renders the same rows of up to kMaxBatch frames -- a rank's slab of every
// frame of a round in the row-sharded loop, where a single slab (255 workgroups for 135 rows of 1080p) is far too small to fill
// 256 CUs.  What differs from frame to frame travels beside KArgs in the kernel arguments themselves (no upload, recordable in a
// HIP graph as is): camera matrix and position, the culling pyramids' edge basis of that camera, the output buffer.  Frame size,
// projection scalars, far distance, rows, scene and plan are the launch's.  Workgroup b renders tile position b / n of frame
// b % n: the n copies of a tile are neighbours in dispatch order, so that with the tiles sorted heaviest first every CU's share
// (blocks c, c + n_cu, ...) is a stratified sample of the cost distribution.
constexpr int kMaxBatch = 16;
struct KFrame {
    float m[12];
    float ox, oy, oz;
    float edge_up_p[3], edge_up_q[3], edge_right_p[3], edge_right_q[3];
    float edge_pp, edge_qrqr, edge_qcqc;
    float edge_fwd[3];
    uint8_t* out;
};
struct KBatch {
    KFrame f[kMaxBatch];
};

// Arguments of rtx_expand_words (compact pixel words -> records), by value.
constexpr int kMaxExpandSeg = 16;
constexpr int kExpandPixels = 1024;             // pixels per workgroup
struct ExpandArgs {
    const uint32_t* src;                    // compact words
    uint8_t* dst;                           // records
    uint32_t nseg;
    uint32_t aligned16;                     // every destination segment starts on a 16-byte boundary
    uint32_t first_block[kMaxExpandSeg + 1]; // workgroups [first_block[k], first_block[k+1]) expand segment k
    uint32_t npix[kMaxExpandSeg];
    uint64_t src_px[kMaxExpandSeg];         // first pixel of the segment in src / dst
    uint64_t dst_px[kMaxExpandSeg];
};

extern "C" {
int rtx_k_launch_expand(const ExpandArgs* e, int mode, unsigned blocks, void* stream);
// Launches the trace kernel for `mode`; returns the kernel's name (NULL for an invalid mode) and
// the hipGetLastError() value in *hip_error.
const char* rtx_k_launch_trace(const KArgs* a, int mode, int cull, void* stream, int* hip_error);
// The batched form (culling kernels without per-wave refinement, records or compact words): a->batch_n frames of kb in one launch.
const char* rtx_k_launch_trace_batch(const KArgs* a, const KBatch* kb, int mode, void* stream, int* hip_error);
int rtx_k_launch_bin_cells(const KArgs* a, unsigned splits, void* stream);
int rtx_k_launch_zero(void* p, size_t bytes, void* stream);
// tile_cost[n_tiles] (grid gx wide) -> tile_order[n_tiles], heaviest first, dealt over n_cu compute units so that the
// workgroups each unit receives in the first dispatch round (blocks c, c + n_cu, c + 2 n_cu, ...) carry equal work.
// The same for a grid whose workgroups are all resident at once (n_tiles <= 2048, n_cu <= 1024): tile_cost holds 3 n_tiles
// words (estimates by tile; start and end times by dispatch position, as the trace kernel leaves them), factor[n_tiles]
// the per-tile corrections carried from launch to launch, prev_order the order the times were taken under (NULL: frame
// order; may be tile_order itself).
int rtx_k_launch_balance_tiles(const uint32_t* tile_cost, uint32_t n_tiles, uint32_t gx, uint32_t n_cu, const uint32_t* prev_order, float* factor,
                               int have_factor, int have_times, uint32_t* tile_order, void* stream);
int rtx_k_launch_order_tiles(const uint32_t* tile_cost, uint32_t n_tiles, uint32_t gx, uint32_t n_cu, uint32_t first_round, const uint32_t* base_order,
                             uint32_t* tile_order, void* stream);
}
