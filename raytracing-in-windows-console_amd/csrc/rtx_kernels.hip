// rtx_kernels.hip -- gfx950 kernels of the ray-trace hot path.
//
// One thread per pixel per pass (as the reference, RayTracingManager.cu:120-125).  A 256-thread workgroup owns a macro tile
// of NSUB sub-tiles, each 2^lw x 2^(8-lw) pixels, and handles one sub-tile per pass.  Once per workgroup: the spheres it has
// to consider -- the whole scene, or the list of its coarse cell (rtx_bin_cells; lists may outlive the frame, rtx_plan.hpp) --
// are walked 512 at a time (two coalesced float4 loads per thread, the next step prefetched), the ray-independent terms
// otc = o - c and cc = Dot(otc,otc) - r*r are hoisted (Sphere.cu:34-37), spheres whose inflated bound cannot touch the macro
// tile's pyramid are culled (CULL), and survivors are appended to a candidate list in LDS; the per-column and per-row terms
// of ray generation, the decimal-digit table, the glyph ramp and the planes the tile can see are staged in LDS as well.
// Then, per sub-tile, every thread runs the reference's exact ray/sphere test over the list (wave-uniform index, LDS broadcast
// reads; REFINE: over the part of it that can touch the wave's own 64 pixels) and the plane tests from the LDS table.  The
// winner alone is shaded and encoded (the reference re-derives normal/colour on every improving hit, RayTracing.cu:123-135,
// but only the last survives).
//
// Closest hit = lexicographic minimum of (t, creation order): the same object the reference's in-order scan with strict '<'
// keeps (RayTracing.cu:123).  Spheres are known by their position in the arrays the kernel reads (KArgs::sph_geom: the
// direction-sorted copies when there are some); the creation order is looked up only in an exact tie (comes_first).
#include "rtx_device.hpp"
#include "rtx_kernels.h"
#include "rtx_records.hpp"

// The experiment build's hooks (ABL, STAMP, RTX_X_*): empty / constant false in the product build.
#define RTX_X_SECTION_DEVICE
#include "rtx_experiment.inc"
#undef RTX_X_SECTION_DEVICE
#define RTX_X_SECTION_HOST
#include "rtx_experiment.inc"
#undef RTX_X_SECTION_HOST

namespace rtx {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's outstanding
// global loads (s_waitcnt vmcnt(0)), which would expose the latency of every prefetch in flight; in
// these kernels threads exchange data through LDS alone (global memory is read-only input or
// write-only output), so waiting for LDS operations is sufficient.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kThreads = 256;
constexpr int kListCapBrute = 1024;  // candidate records per flush (brute: every sphere is a candidate)
constexpr int kListCapCull = 704;    // culling kernels: room for one more 512-sphere step while <= 192 are listed; 8 workgroups per CU
constexpr int kListCapRefine = 640;  // REFINE: ... while <= 128 are listed
constexpr float kNoHit = 99999999.f; // RayTracing.h:21

// RayTracing.h:97-115 (68 glyphs).
__constant__ __attribute__((aligned(4))) const char kRamp[68] = {
    ' ', '.', '`', '^', '"', ',', ':', ';', 'I', 'l', '!', 'i', '>', '<', '~', '+', '_',
    '-', '?', '*', ']', '[', '}', '{', '1', ')', '(', '|', '/', 't', 'f', 'j', 'r', 'x',
    'n', 'u', 'v', 'c', 'z', 'm', 'w', 'X', 'Y', 'U', 'J', 'C', 'L', 'q', 'p', 'd', 'b',
    'k', 'h', 'a', 'o', '#', '%', 'Z', 'O', '8', 'B', '$', '0', 'Q', 'M', '&', 'W', '@'
};

// Conservative inflation of a sphere for culling.  A ray whose fp32 test reports a hit passes, in exact
// arithmetic, within R of the centre with R^2 = r^2 (1+2u) + 15.2u |otc|^2 (u = 2^-24; derivation in
// DESIGN.md "Culling soundness"), and no further than 3u|otc| behind the apex.  kappa = 2e-6 is 2.2x
// that constant; kDelta |otc| covers the apex term and the rounding of the plane evaluation, of the
// plane normals and of the hardware sqrt used below (together < 1.5e-6 |otc|).
constexpr float kKappa = 2.0e-6f;
constexpr float kDelta = 5.0e-6f;

struct TileFrustum {
    V3 n[5]; // inward unit normals of the four side planes, then the frame's camera plane
};

__device__ __forceinline__ V3 cross(V3 a, V3 b)
{
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// Direction (unnormalised) through view-plane point (cx, cy): the linear part of make_ray.
__device__ __forceinline__ V3 view_dir(const Camera& c, float cx, float cy)
{
    const float vx = cx * c.e1, vy = cy * c.e2;
    return v3(c.m[0] * vx + c.m[1] * vy + c.m[2], c.m[4] * vx + c.m[5] * vy + c.m[6], c.m[8] * vx + c.m[9] * vy + c.m[10]);
}

// true when the sphere (hoisted form) provably cannot be hit by any pixel ray of the tile.
__device__ __forceinline__ bool tile_culls(const TileFrustum& f, float ox, float oy, float oz, float oo, float r, float& margin)
{
    // margin >= R + 3u|otc| + evaluation slack (see kKappa); hardware sqrt is within 1 ulp
    margin = __builtin_amdgcn_sqrtf(r * r * (1.0f + kKappa) + kKappa * oo) + kDelta * __builtin_amdgcn_sqrtf(oo);
    bool out = false;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        // n . (c - o) = -(n . otc); outside when it is below -margin.  NaN compares false: keep.
        const float d = f.n[k].x * ox + f.n[k].y * oy + f.n[k].z * oz;
        out = out || (d > margin);
    }
    return out;
}

// The same test with the margin grown for cell lists that are to outlive this frame (rtx_plan.hpp, "cell-list reuse"):
// valid for every camera whose pixel directions are within theta (chord) of this one's and whose position, sphere motion
// counted in, is within delta:   margin' = m + delta + theta (|O| + delta + m),   m = margin(r, |O| + delta),
// times 1 + 4e-6 for the rounding of these few operations (hardware sqrt included).  `inside` reports whether some
// such camera can lie inside or on the sphere (then it is never culled).  theta = delta = 0 gives tile_culls' margin
// within that factor.
__device__ __forceinline__ bool tile_culls_moving(const TileFrustum& f, float ox, float oy, float oz, float oo, float r, float theta, float delta, float& margin,
                                                  bool& inside)
{
    const float dist = __builtin_amdgcn_sqrtf(oo) * (1.0f + 1.0e-6f) + delta; // >= |O| + delta
    const float m = __builtin_amdgcn_sqrtf(r * r * (1.0f + kKappa) + kKappa * (dist * dist)) + kDelta * dist;
    margin = (m + delta + theta * (dist + m)) * (1.0f + 4.0e-6f);
    inside = !(__builtin_amdgcn_sqrtf(oo) * (1.0f - 1.0e-6f) - delta > fabsf(r) * (1.0f + 1.0e-6f));
    bool out = false;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const float d = f.n[k].x * ox + f.n[k].y * oy + f.n[k].z * oz;
        out = out || (d > margin);
    }
    return out;
}

// One plane of the pyramid spanned by the pixel centres of columns [col0, col0+w) and rows [row0, row0+h), grown by
// half a pixel on every side, selected by k: 0 the top edge (row0), 1 the right edge (col0 + w), 2 the bottom edge
// (row0 + h), 3 the left edge (col0), 4 the camera plane -- arranged so that five lanes compute the five planes side by
// side.  Pixel directions are linear in (cx, cy), so every pixel ray of the rectangle lies in the convex cone of the four
// corner directions, and the plane through the apex and an edge cy = y (cx = x) has the normal y up_p + up_q (x right_p +
// right_q), oriented up (right) in the frame, from per-frame vectors (rtx_plan.hpp, EdgeBasis: computed on the host in
// double).  NOT the cross product of the two corner directions: for a thin tile far off the view axis those are nearly
// parallel and long, and the fp32 product loses the plane (8K, first 16 columns: 4e-6 rad with the reference's own
// camera matrices -- most of the half pixel, 5.3e-6 rad there -- and up to 8e-5 rad with a rolled camera, where a culling
// kernel then drops rows of a large far sphere's cap: tools/wide_view_directed_gpu.py).
// The fifth plane is the same for every rectangle of the frame: the camera plane (it culls what lies behind the apex; all
// pixel rays of the frame point into its front half-space, which the host has checked).
// An ill-conditioned normal (degenerate or sheared matrix, NaN) becomes the zero vector, which never culls.  This is
// culling geometry, not reference arithmetic: hardware rcp/rsq (1 ulp) are used, and the slack in tile_culls covers their
// error and the normal's (4e-7 from the two roundings per component and the rounded basis; 4e-7 from the edge
// coordinate's own rounding).
__device__ __forceinline__ V3 tile_plane(const KArgs& a, const Camera& c, uint32_t col0, uint32_t row0, uint32_t w, uint32_t h, uint32_t k)
{
    if (k == 4u) {
        return v3(a.edge_fwd[0], a.edge_fwd[1], a.edge_fwd[2]);
    }
    const bool row_edge = (k & 1u) == 0u;
    // the edge's coordinate: cy of row boundary e (half a pixel above row e), cx of column boundary e
    const uint32_t e = row_edge ? row0 + (k == 2u ? h : 0u) : col0 + (k == 1u ? w : 0u);
    const float t = row_edge ? (c.fH - 2.0f * (float)e + 1.0f) * __builtin_amdgcn_rcpf(c.fH) : (2.0f * (float)e - 1.0f - c.fW) * __builtin_amdgcn_rcpf(c.fW);
    const V3 p = row_edge ? v3(a.edge_up_p[0], a.edge_up_p[1], a.edge_up_p[2]) : v3(a.edge_right_p[0], a.edge_right_p[1], a.edge_right_p[2]);
    const V3 q = row_edge ? v3(a.edge_up_q[0], a.edge_up_q[1], a.edge_up_q[2]) : v3(a.edge_right_q[0], a.edge_right_q[1], a.edge_right_q[2]);
    V3 n = v3(t * p.x + q.x, t * p.y + q.y, t * p.z + q.z);
    // refuse a sum that cancelled (the two terms are perpendicular for a camera matrix: no cancellation at all)
    const float hyp2 = (t * t) * a.edge_pp + (row_edge ? a.edge_qrqr : a.edge_qcqc);
    const float len2 = dot(n, n);
    if (!(len2 >= 0.25f * hyp2) || !(len2 > 1.0e-30f && len2 < 1.0e30f)) {
        return v3(0.0f, 0.0f, 0.0f);
    }
    // inward: down from the top edge, left from the right edge, up from the bottom edge, right from the left edge
    const float s = (k < 2u ? -1.0f : 1.0f) * __builtin_amdgcn_rsqf(len2);
    return mulf(n, s);
}

// true when no pixel ray of the rectangle (columns [col0, col0+w), rows [row0, row0+h), half a pixel out, as for the
// pyramids) can hit the bounded plane {n, num = Dot(planePos - origin, n), bounds xlo xhi zlo zhi}, so that the
// workgroup may leave the plane out of its table.  Plane::Trace (Plane.cu:38-72) reports a hit iff dn < 0 (beyond
// FLT_EPSILON), t = num / dn > 0 and the hit point lies strictly inside the bounds in x and z.
//  * num >= 0: a ray with dn < 0 gets t <= 0 -- exactly, the fp32 quotient has the sign of its operands -- and every
//    other ray is rejected on dn: no hit, whatever the rectangle.
//  * Pixel directions are convex combinations of the four corner directions w_i, and w . n is linear: if it is
//    positive at all corners (by more than 1e-5 |w||n|) every dn is positive: no hit.
//  * If it is negative at all corners, every ray hits the unbounded plane and the hit points lie in the convex
//    hull of the four corner hit points (central projection of a convex cone onto a plane that cuts all its rays).
//    The plane is invisible when their bounding interval, widened by the worst rounding error of the per-pixel
//    arithmetic, misses the bounds in x or in z.  That error: |dn| >= q = min |w_i . n| / max |w_i| over the cone,
//    dn carries an absolute error below 4e-7, so t and the hit offset d t carry a relative error below
//    4e-7 / q + 1e-6; the bound used is 2e-6 / q + 1e-5 of (|origin| + largest corner offset), and grazing
//    rectangles (q < 1e-3) are kept.
// Anything undecided (mixed signs, NaNs, degenerate matrices) keeps the plane.  Culling geometry, not reference
// arithmetic: hardware rcp/sqrt.
__device__ __forceinline__ bool plane_invisible(const Camera& c, uint32_t col0, uint32_t row0, uint32_t w, uint32_t h, V3 n, float num, float4 bd)
{
    if (num >= 0.0f) {
        return true;
    }
    const float rW = __builtin_amdgcn_rcpf(c.fW), rH = __builtin_amdgcn_rcpf(c.fH);
    const float x0 = (2.0f * (float)col0 - 1.0f - c.fW) * rW;
    const float x1 = (2.0f * (float)(col0 + w) - 1.0f - c.fW) * rW;
    const float y0 = (c.fH - 2.0f * (float)row0 + 1.0f) * rH;
    const float y1 = (c.fH - 2.0f * (float)(row0 + h) + 1.0f) * rH;
    const float nn = dot(n, n);
    bool away = true, facing = true;
    float min_abs = __builtin_inff(), max_len2 = 0.0f;
    float hx_lo = __builtin_inff(), hx_hi = -__builtin_inff(), hz_lo = __builtin_inff(), hz_hi = -__builtin_inff(), span_x = 0.0f, span_z = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const V3 wi = view_dir(c, (i == 1 || i == 2) ? x1 : x0, i >= 2 ? y1 : y0);
        const float wn = dot(wi, n), len2 = dot(wi, wi);
        const float eps = 1.0e-5f * __builtin_amdgcn_sqrtf(len2 * nn);
        away = away && (wn > eps);
        facing = facing && (wn < -eps);
        min_abs = fminf(min_abs, fabsf(wn));
        max_len2 = fmaxf(max_len2, len2);
        const float si = num * __builtin_amdgcn_rcpf(wn); // > 0 where the corner faces the plane
        const float dx = wi.x * si, dz = wi.z * si;
        hx_lo = fminf(hx_lo, c.ox + dx);
        hx_hi = fmaxf(hx_hi, c.ox + dx);
        hz_lo = fminf(hz_lo, c.oz + dz);
        hz_hi = fmaxf(hz_hi, c.oz + dz);
        span_x = fmaxf(span_x, fabsf(dx));
        span_z = fmaxf(span_z, fabsf(dz));
    }
    if (away) {
        return true;
    }
    if (!facing) {
        return false;
    }
    const float q = min_abs * __builtin_amdgcn_rsqf(max_len2 * nn);
    if (!(q > 1.0e-3f)) {
        return false;
    }
    const float rel = 2.0e-6f * __builtin_amdgcn_rcpf(q) + 1.0e-5f;
    const float ex = rel * (fabsf(c.ox) + span_x) + 1.0e-30f, ez = rel * (fabsf(c.oz) + span_z) + 1.0e-30f;
    return (hx_hi + ex <= bd.x) || (hx_lo - ex >= bd.y) || (hz_hi + ez <= bd.z) || (hz_lo - ez >= bd.w);
}

struct Best {
    float t;
    uint32_t k; // the sphere's position in the arrays staging reads (= its index without sorted copies), or 0xffffffff
};

// Does the sphere at position p come before the one at position q in creation order?  (The reference keeps the first of two
// objects at the same distance, RayTracing.cu:123; q = 0xffffffff: no sphere yet.)  Rare path: the translation table's address
// is parked in LDS (rare[kRareSortedIdx]; 0 = positions are sphere indices), two loads when the arrays are sorted.
constexpr int kRareSortedIdx = 4;
__device__ __forceinline__ bool comes_first(const unsigned long long* rare, uint32_t p, uint32_t q)
{
    asm volatile("" ::: "memory"); // (keeps the read below in this branch instead of hoisting it out of the candidate loop)
    const uint32_t* sorted_idx = reinterpret_cast<const uint32_t*>((uintptr_t)rare[kRareSortedIdx]);
    if (q == 0xffffffffu || sorted_idx == nullptr) {
        return p < q;
    }
    return sorted_idx[p] < sorted_idx[q];
}

template <int MODE>
__device__ __forceinline__ Fields pixel_fields(const KArgs& a, const uint8_t* s_ramp, V3 normal, V3 colour, float shadingValue)
{
    constexpr bool kRgb = (MODE == RTX_K_RGB_ASCII || MODE == RTX_K_RGB_PIXEL || MODE == RTX_K_RGB_NORMALS);
    Fields f;
    if (kRgb) {
        if (MODE == RTX_K_RGB_NORMALS) {
            f.c0 = u8_sat(normal.x * 255.0f);
            f.c1 = u8_sat(normal.y * 255.0f);
            f.c2 = u8_sat(normal.z * 255.0f);
        } else {
            f.c0 = u8_sat(colour.x);
            f.c1 = u8_sat(colour.y);
            f.c2 = u8_sat(colour.z);
        }
        f.glyph = (MODE == RTX_K_RGB_ASCII) ? (uint32_t)s_ramp[ramp_index(shadingValue)] : (uint32_t)' ';
    } else {
        f.c0 = ansi256_from_rgb(u8_sat(colour.x), u8_sat(colour.y), u8_sat(colour.z), a.grey);
        f.c1 = f.c2 = 0u;
        f.glyph = (MODE == RTX_K_BIT_ASCII) ? (uint32_t)s_ramp[ramp_index(shadingValue)] : (uint32_t)' ';
    }
    return f;
}

// OUT: what a pixel's result is stored as -- a kernel per form, so that the frame loop's kernel carries no
// trace of the other two.
enum { kOutRecords = 0, kOutCompact = 1, kOutValues = 2 };

template <int MODE, int OUT>
__device__ __forceinline__ void encode_and_store(const KArgs& a, const Camera& cam, const uint32_t* s_digits, const uint8_t* s_ramp, bool in_frame, bool is_newline_col,
                                                 uint32_t row, uint32_t col, float distance, V3 normal, V3 colour, float shadingValue)
{
    constexpr bool kRgb = (MODE == RTX_K_RGB_ASCII || MODE == RTX_K_RGB_PIXEL || MODE == RTX_K_RGB_NORMALS);
    constexpr uint32_t S = kRgb ? 20u : 12u;
    if (!in_frame) {
        return;
    }
    const bool visible = distance <= cam.far; // RayTracing.cu:207,288,371,508,646
    if (OUT == kOutValues) {
        // RTX_RENDER_VALUES: the floats behind the record (RayTraceReturnData, RayTracing.h:17-23), for parity checks
        float4* o = reinterpret_cast<float4*>(a.out) + ((size_t)(row - a.out_row_base) * a.W + col) * 2u;
        if (is_newline_col) {
            o[0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            o[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        } else {
            o[0] = make_float4(distance, shadingValue, normal.x, normal.y);
            o[1] = make_float4(normal.z, colour.x, colour.y, colour.z);
        }
        return;
    }
    if (OUT == kOutCompact) {
        uint32_t word = kCompactNewline;
        if (!is_newline_col) {
            word = kCompactMiss;
            if (visible) {
                const Fields f = pixel_fields<MODE>(a, s_ramp, normal, colour, shadingValue);
                word = f.c0 | (f.c1 << 8) | (f.c2 << 16) | (f.glyph << 24);
            }
        }
        reinterpret_cast<uint32_t*>(a.out)[(size_t)(row - a.out_row_base) * a.W + col] = word;
        return;
    }
    uint32_t* dst = reinterpret_cast<uint32_t*>(a.out + ((size_t)(row - a.out_row_base) * a.W + col) * S);
    if (is_newline_col) {
        // column W-1 stays NUL (RayTracing.cu:187 never writes it; the reference memsets it)
#pragma unroll
        for (uint32_t i = 0; i < S / 4u; i++) {
            dst[i] = 0u;
        }
        return;
    }
    uint32_t w[5];
    Fields f = {0u, 0u, 0u, 0u};
    if (visible) {
        f = pixel_fields<MODE>(a, s_ramp, normal, colour, shadingValue);
    }
    record_words<MODE>(visible, f, s_digits, w);
#pragma unroll
    for (uint32_t i = 0; i < S / 4u; i++) {
        dst[i] = w[i];
    }
}

// Exact tests of one ray against the candidate list (index is wave-uniform: LDS broadcast reads).
// HOISTED: ray.divTwoA was computed before the loop (dense scenes: several candidates per pass reach the exact test, so
// once per ray is cheaper than once per exact test; sparse scenes: 1 in 3 passes reaches none)
template <bool HOISTED>
__device__ __forceinline__ void test_candidate(Ray& ray, const float4 sr, const uint32_t* s_idx, const unsigned long long* rare, uint32_t i, Best& best, uint32_t& slow)
{
    float s;
    const bool rejected = sphere_reject(ray, sr.x, sr.y, sr.z, sr.w, s);
    RTX_X_CANDIDATE_BEGIN();
    if (!rejected) {
        if (!HOISTED) ray.divTwoA = rcp_cr(2.0f * ray.a); // RayTracing.cu:93; only the hit path reads it
        float t;
        if (sphere_hit(ray, s, sr.w, t)) {
            const uint32_t ki = s_idx[i];
            if (t < best.t || (t == best.t && comes_first(rare, ki, best.k))) {
                best.t = t;
                best.k = ki;
                RTX_X_CANDIDATE_UPDATED();
            }
        }
    }
    RTX_X_CANDIDATE_END(rejected, slow);
}
template <bool HOISTED>
__device__ __forceinline__ void scan_candidates(Ray& ray, const float4* s_rec, const uint32_t* s_idx, const unsigned long long* rare, uint32_t total, Best& best,
                                                uint32_t& slow)
{
    for (uint32_t i = 0; i < total; i++) {
        test_candidate<HOISTED>(ray, s_rec[i], s_idx, rare, i, best, slow);
    }
}

// Ray of pixel (row, col) from the staged per-column / per-row terms:
//   A = (m0, m4, m8) * (convertedX * e1),  B = (m1, m5, m9) * (convertedY * e2)
// so that d.k = ((A.k + B.k) + m[4k+2] * 1.0f) + m[4k+3] * 0.0f, the order of Matrix::Mult
// (MyMath.h:310-319) applied to (vx, vy, 1, 0) as in RayTracing.cu:19-23.
__device__ __forceinline__ Ray ray_from_tables(const Camera& c, float4 A, float4 B)
{
    V3 w;
    w.x = ((A.x + B.x) + c.m[2]) + c.m[3] * 0.0f;
    w.y = ((A.y + B.y) + c.m[6]) + c.m[7] * 0.0f;
    w.z = ((A.z + B.z) + c.m[10]) + c.m[11] * 0.0f;
    Ray r;
    r.o = v3(c.ox, c.oy, c.oz);
    r.d = normalize_gpu(w);
    r.a = dot(r.d, r.d);
    r.fourA = 4.0f * r.a;
    r.divTwoA = 0.0f; // filled on the hit path only
    return r;
}

// Work estimate of one wave's pass over one sub-tile, in VALU instructions (tools/ablate_pmc_gpu.sh: about 100 for
// ray generation, planes and a miss record; 3 per candidate rejected; 330 more when anything is hit: the hit tests'
// slow path, the winner's normal, shading and the record).  Only the order of the sums matters.
constexpr uint32_t kCostPass = 100u, kCostCandidate = 3u, kCostShaded = 330u;

constexpr int kMaxMacro = 128;    // macro tile is at most 128 x 128 pixels
constexpr int kMaxMacroRefine = 64; // ... 64 x 64 for the REFINE kernels
constexpr int kChunk = 2 * kThreads; // spheres staged per barrier: two per thread
constexpr int kPlaneTable = 16;   // planes hoisted into LDS; further planes take the direct path

constexpr uint32_t kStageFull = 0xffffffffu; // stage_chunk: the step's survivors would not fit the list (nothing was written)

// One staging step: items [base, base + 512) of `ns` items, two per thread.  g0/g1 (sphere indices k0/k1)
// are this thread's geometry records, already loaded: the caller prefetches the next step's first.  Hoists the
// ray-independent terms, culls, and appends survivors to the LDS list in index order (wave ballots,
// per-wave counts through LDS, one barrier).  Returns the new list length (uniform), or kStageFull when the step's
// survivors would take the list past `cap` entries (then nothing is written).
// (Keeping the geometry and colour records of the first list entries in LDS, so that a pass takes its winner's records
// from there instead of a round trip to memory, was measured and dropped: no gain, profiles/r02_c_single_launch_experiments.md.)
template <bool CULL>
__device__ __forceinline__ uint32_t stage_chunk(const Camera& cam, const TileFrustum& fr, uint32_t ns, uint32_t base, float4 g0, float4 g1,
                                                uint32_t k0, uint32_t k1, float4* s_rec, uint32_t* s_idx, uint32_t (*s_wcnt)[8],
                                                uint32_t parity, uint32_t total, bool drop_all, uint32_t cap, float* s_margin = nullptr)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    bool keep[2];
    float4 rec[2];
    float mg[2] = {0.0f, 0.0f};
    const float4 g[2] = {g0, g1};

    // the second half of the step is empty when at most 256 items are left (short cell lists, the tail of a scene):
    // skip its arithmetic (uniform branch)
    const bool second_half_empty = base + (uint32_t)kThreads >= ns;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (h == 1 && second_half_empty) {
            keep[1] = false;
            rec[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            break;
        }
        const uint32_t k = base + (uint32_t)h * kThreads + tid; // item number; valid while below ns (the item count)
        // objectToCam = origin - spherePos; c = Dot(otc,otc) - r*r   (Sphere.cu:34-37)
        const float ox = cam.ox - g[h].x, oy = cam.oy - g[h].y, oz = cam.oz - g[h].z;
        const float oo = ox * ox + oy * oy + oz * oz;
        const float cc = oo - (g[h].w * g[h].w);
        rec[h] = make_float4(ox, oy, oz, cc);
        keep[h] = (k < ns) && !drop_all;
        if (CULL) {
            // cc <= 0: the camera is inside or on the sphere; keep (the exact test decides)
            const bool culled = tile_culls(fr, ox, oy, oz, oo, g[h].w, mg[h]);
            keep[h] = keep[h] && !(cc > 0.0f && culled);
            if (!(cc > 0.0f)) {
                mg[h] = __builtin_inff(); // camera inside or on the sphere: never culled, by any pyramid
            }
        }
    }
    const unsigned long long m0 = __ballot(keep[0]), m1 = __ballot(keep[1]);
    if (lane == 0) {
        s_wcnt[parity][wave] = (uint32_t)__popcll(m0);
        s_wcnt[parity][4 + wave] = (uint32_t)__popcll(m1);
    }
    // LDS-only barrier: the caller's prefetched global loads stay in flight across it.  It also orders
    // the previous scan's list reads before the writes below.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    uint32_t before = 0, sum = 0;
#pragma unroll
    for (uint32_t w = 0; w < 8u; w++) {
        const uint32_t c = s_wcnt[parity][w];
        sum += c;
        if (w < wave) before += c; // first halves of earlier waves (wave <= 3)
    }
    // a step whose survivors do not fit the list is not written at all (uniform: every thread sees the same counts)
    const uint32_t grown = __builtin_amdgcn_readfirstlane(total + sum);
    if (grown > cap) {
        return kStageFull;
    }
    // index order: [first-half survivors of waves 0..3][second-half survivors of waves 0..3]
    uint32_t first_total = 0, before1 = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4u; w++) {
        first_total += s_wcnt[parity][w];
        if (w < wave) before1 += s_wcnt[parity][4 + w];
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    if (keep[0]) {
        const uint32_t pos = total + before + (uint32_t)__popcll(m0 & below);
        s_rec[pos] = rec[0];
        s_idx[pos] = k0;
        if (s_margin) s_margin[pos] = mg[0];
    }
    if (keep[1]) {
        const uint32_t pos = total + first_total + before1 + (uint32_t)__popcll(m1 & below);
        s_rec[pos] = rec[1];
        s_idx[pos] = k1;
        if (s_margin) s_margin[pos] = mg[1];
    }
    return grown;
}

// The spheres a workgroup stages: all of them (list == nullptr), or the index list its coarse cell
// received from rtx_bin_cells (two-level culling for large scenes).
// A sphere is known to the trace kernels by its POSITION in the arrays they read: the direction-sorted copies when there are
// some (KArgs::sph_sorted_*), else the scene arrays themselves, where position = sphere index.  Only an exact tie in t needs the
// creation order (comes_first).
struct Items {
    const float4* geom;   // geometry by position
    const uint32_t* list; // positions (any order), or nullptr: all of [0, count)
    uint32_t count;       // number of items
};

__device__ __forceinline__ Items scene_items(const KArgs& a)
{
    Items it;
    it.geom = a.sph_geom;
    it.list = nullptr;
    it.count = a.ns;
    return it;
}

// Item i: its position and its geometry record.  Past the end any valid record is returned (ignored).
__device__ __forceinline__ float4 load_item(const Items& it, uint32_t i, uint32_t& k)
{
    if (it.count == 0u) {
        k = 0u;
        return make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const uint32_t ii = i < it.count ? i : it.count - 1u;
    k = it.list ? it.list[ii] : ii;
    return it.geom[k];
}

constexpr int kRefineSub = 4;      // REFINE: at most this many sub-tiles (lanes 5 .. 5 + 16 per sub-tile build the pyramids)
constexpr int kWaveListCap = 128;  // REFINE: candidates a wave keeps for its own 64 pixels; more -> it scans the whole list

// REFINE (dense scenes, long candidate lists): before a wave scans the workgroup's list for its 64 pixels it
// tests the list, one entry per lane, against the pyramid of just those pixels -- same conservative test and
// margin as for the macro tile -- and scans the survivors only.
#ifndef RTX_WAVES_PER_EU
#define RTX_WAVES_PER_EU 7 // 72 VGPRs: 7 workgroups per CU instead of 6 (20.4 -> 19.6 us per frame; 8 needs spills and gains nothing)
#endif
// Where a workgroup stands: in a plain launch the grid is the frame's tile grid; in a batched one (rtx_trace_batch) it is 1-D,
// workgroup b on tile position b / n of frame b % n.  Computed where they are used (at the head and at the very end of the
// kernel), from the block index, rather than carried in scalar registers across the pass loop.
template <bool BATCH>
struct Where {
    static __device__ __forceinline__ uint32_t pos(const KArgs& a) { return BATCH ? blockIdx.x / a.batch_n : blockIdx.y * gridDim.x + blockIdx.x; }
    static __device__ __forceinline__ uint32_t grid_x(const KArgs& a) { return BATCH ? a.batch_gx : gridDim.x; }
    static __device__ __forceinline__ uint32_t n_tiles(const KArgs& a) { return BATCH ? a.batch_gx * a.batch_gy : gridDim.x * gridDim.y; }
    static __device__ __forceinline__ bool first_block() { return BATCH ? blockIdx.x == 0u : (blockIdx.x == 0u && blockIdx.y == 0u); }
    // one frame of a batch leaves the work estimates (the tiles' costs differ little from frame to frame of a round)
    static __device__ __forceinline__ bool leaves_cost(const KArgs& a) { return BATCH ? blockIdx.x % a.batch_n == 0u : true; }
};

// The body of the trace kernels lives in rtx_trace_body.inc, included into the two entry points below.
template <int MODE, bool CULL, int OUT, bool REFINE>
__global__ __launch_bounds__(kThreads, RTX_WAVES_PER_EU) void rtx_trace(const KArgs a)
{
    constexpr bool BATCH = false;
#include "rtx_trace_body.inc"
}

// The same rows of a.batch_n frames in one launch (KBatch, rtx_kernels.h): a 1-D grid of batch_n x tiles workgroups, workgroup b
// on tile position b / batch_n of frame b % batch_n.  The frame's camera, edge basis and output buffer replace the launch's.
template <int MODE, int OUT>
__global__ __launch_bounds__(kThreads, RTX_WAVES_PER_EU) void rtx_trace_batch(const KArgs a0, const KBatch kb)
{
    const uint32_t frame = blockIdx.x % a0.batch_n;
    KArgs a = a0;
    const KFrame& f = kb.f[frame];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        a.m[i] = f.m[i];
    }
    a.ox = f.ox;
    a.oy = f.oy;
    a.oz = f.oz;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        a.edge_up_p[i] = f.edge_up_p[i];
        a.edge_up_q[i] = f.edge_up_q[i];
        a.edge_right_p[i] = f.edge_right_p[i];
        a.edge_right_q[i] = f.edge_right_q[i];
        a.edge_fwd[i] = f.edge_fwd[i];
    }
    a.edge_pp = f.edge_pp;
    a.edge_qrqr = f.edge_qrqr;
    a.edge_qcqc = f.edge_qcqc;
    a.out = f.out;
    constexpr bool BATCH = true, CULL = true, REFINE = false;
#include "rtx_trace_body.inc"
}

// Level 1 of the two-level culling used for large scenes: the scene is binned into coarse cells (blocks of
// 2^gx x 2^gy macro tiles, a few hundred to a thousand of them) and every trace workgroup then stages its cell's
// list instead of the whole scene.  ONE launch, two levels inside it: workgroup (block, split) walks its share of the
// sphere array against the pyramid of a BLOCK of 4 x 4 cells -- the same conservative test as the per-tile one, on a
// larger rectangle -- gathers the survivors in LDS (hoisted centre and culling margin, as the REFINE pass of the trace
// kernel keeps them) and tests each of them against the 16 cell pyramids of the block, one (survivor, cell) pair
// per thread.  Sound level by level: a sphere that a pixel ray of a cell can hit can be hit by a ray of the
// cell's block.  Tests: blocks x spheres + 16 x survivors instead of cells x spheres.
//
// Lists are compact: cell c owns cell_cap entries (a few times the average; host: rtx_render_rows) and the
// workgroups append with one reservation per cell and flush -- atomicAdd on the cell's counter, which keeps
// counting past the capacity.  A trace workgroup that finds count > cell_cap stages the whole scene instead
// (always a superset), so a list that does not fit costs time, never a pixel.  The order of a list does not
// matter: the closest hit is the minimum of (t, creation index).
//
// The counters alternate between two buffers from launch to launch: this launch accumulates into
// cell_count_out and zeroes the 16 counters of its block in cell_count_zero, which the next launch on this
// stream accumulates into -- no memset between frames.
constexpr int kBlockCells = 16; // cells per block: 4 x 4
constexpr int kBinCap = 1024;   // block survivors gathered in LDS between cell phases

// Appends the block survivors [0, n) in s_rec / s_idx to the lists of the cells they can touch.
__device__ __forceinline__ void bin_cells_of_block(const KArgs& a, const float4* s_rec, const uint32_t* s_idx, uint32_t n, const float4 (*s_cellfr)[5],
                                                   uint32_t* s_cnt, uint32_t* s_base, uint32_t* s_pos, const uint32_t* s_cellid)
{
    const uint32_t tid = threadIdx.x;
    __syncthreads(); // survivors complete
    if (tid < (uint32_t)kBlockCells) {
        s_cnt[tid] = 0u;
        s_pos[tid] = 0u;
    }
    __syncthreads();
    // pass 1: the pairs' verdicts (kept in a bit mask: pair p = tid + 256 i is bit i) and the cells' counts
    const uint32_t pairs = n * (uint32_t)kBlockCells;
    unsigned long long keep = 0ull;
    for (uint32_t p = tid, i = 0; p < pairs; p += (uint32_t)kThreads, i++) {
        const uint32_t sv = p >> 4, c = p & 15u;
        if (s_cellid[c] == 0xffffffffu) {
            continue; // cell beyond the edge of the grid
        }
        const float4 r = s_rec[sv];
        bool out = false;
#pragma unroll
        for (int k = 0; k < 4; k++) { // (the four side planes: the block's survivors have passed the camera plane)
            const float4 nk = s_cellfr[c][k];
            out = out || (nk.x * r.x + nk.y * r.y + nk.z * r.z > r.w); // as tile_culls, margin in r.w (+inf: never)
        }
        if (!out) {
            keep |= 1ull << i;
            atomicAdd(&s_cnt[c], 1u);
        }
    }
    __syncthreads();
    if (tid < (uint32_t)kBlockCells && s_cnt[tid] != 0u) {
        s_base[tid] = atomicAdd(a.cell_count_out + s_cellid[tid], s_cnt[tid]); // one reservation per cell and flush
        // capacity feedback for the host (rtx_plan.hpp, cell_capacity_wanted): only lists past half their capacity report
        const uint32_t need = s_base[tid] + s_cnt[tid];
        if (a.cell_max_out != nullptr && need > (a.cell_cap >> 1)) atomicMax(a.cell_max_out, need);
    }
    __syncthreads();
    for (uint32_t p = tid, i = 0; p < pairs; p += (uint32_t)kThreads, i++) {
        if ((keep >> i) & 1ull) {
            const uint32_t sv = p >> 4, c = p & 15u;
            const uint32_t at = s_base[c] + atomicAdd(&s_pos[c], 1u);
            if (at < a.cell_cap) {
                a.cell_list_out[(size_t)s_cellid[c] * a.cell_cap + at] = a.sph_pos_of != nullptr ? a.sph_pos_of[s_idx[sv]] : s_idx[sv];
            }
        }
    }
    __syncthreads(); // the survivor buffer may be refilled
}

__global__ __launch_bounds__(kThreads) void rtx_bin_cells(const KArgs a)
{
    static_assert(kBinCap * kBlockCells <= 64 * kThreads, "a thread's pairs must fit the 64-bit verdict mask");
    __shared__ float4 s_rec[kBinCap];              // ox oy oz margin
    __shared__ uint32_t s_idx[kBinCap];
    __shared__ float4 s_cellfr[kBlockCells][5];
    __shared__ uint32_t s_cellid[kBlockCells];     // global cell number, or 0xffffffff beyond the grid
    __shared__ uint32_t s_cnt[kBlockCells], s_base[kBlockCells], s_pos[kBlockCells];
    __shared__ uint32_t s_wcnt[2][8];
    __shared__ float s_frustum[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t tw = 1u << a.tile_log2w, th = (uint32_t)kThreads >> a.tile_log2w;
    const uint32_t nx = 1u << a.sub_log2nx, ny = a.nsub >> a.sub_log2nx;
    const uint32_t cw = (tw * nx) << a.cell_log2gx, ch = (th * ny) << a.cell_log2gy; // cell, pixels
    const uint32_t blocks_x = (a.cells_x + 3u) >> 2;
    const uint32_t bby = blockIdx.x / blocks_x, bbx = blockIdx.x - bby * blocks_x;   // this workgroup's block of cells

    Camera cam;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        cam.m[i] = a.m[i];
    }
    cam.ox = a.ox; cam.oy = a.oy; cam.oz = a.oz;
    cam.e1 = a.e1; cam.e2 = a.e2; cam.far = a.far;
    cam.fW = a.fW; cam.fH = a.fH;

    // pyramids: lanes 0..4 the block's five planes, lanes 5..84 plane k of cell c = (lane - 5) / 5
    if (tid < 5u + 5u * (uint32_t)kBlockCells) {
        if (tid < 5u) {
            const V3 n = tile_plane(a, cam, bbx * 4u * cw, a.row0 + bby * 4u * ch, 4u * cw, 4u * ch, tid);
            s_frustum[3 * tid + 0] = n.x;
            s_frustum[3 * tid + 1] = n.y;
            s_frustum[3 * tid + 2] = n.z;
        } else {
            const uint32_t q = tid - 5u, c = q / 5u, k = q - c * 5u;
            const uint32_t cx = bbx * 4u + (c & 3u), cy = bby * 4u + (c >> 2);
            const V3 n = tile_plane(a, cam, cx * cw, a.row0 + cy * ch, cw, ch, k);
            s_cellfr[c][k] = make_float4(n.x, n.y, n.z, 0.0f);
        }
    }
    if (tid < (uint32_t)kBlockCells) {
        const uint32_t cx = bbx * 4u + (tid & 3u), cy = bby * 4u + (tid >> 2);
        const bool valid = cx < a.cells_x && cy < a.cells_y;
        s_cellid[tid] = valid ? cy * a.cells_x + cx : 0xffffffffu;
        // the other counter buffer, for the next launch on this stream (one workgroup per block does it)
        if (valid && blockIdx.y == 0u && a.cell_count_zero != nullptr) {
            a.cell_count_zero[cy * a.cells_x + cx] = 0u;
        }
    }
    __syncthreads();
    TileFrustum fr;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        fr.n[k].x = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s_frustum[3 * k + 0])));
        fr.n[k].y = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s_frustum[3 * k + 1])));
        fr.n[k].z = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s_frustum[3 * k + 2])));
    }

    // The pass walks the scene array in creation order -- in the direction-sorted copy a block's spheres sit in one stretch, i.e.
    // in one workgroup's share: 36 us instead of 18 -- and translates the survivors to positions when it writes the lists.
    Items items;
    items.geom = a.sph_scene_geom;
    items.list = nullptr;
    items.count = a.ns;
    // this workgroup's share of the spheres: [lo, hi), a multiple of the step size except at the end
    const uint32_t ns = a.ns, splits = gridDim.y;
    const uint32_t steps = (ns + kChunk - 1) / kChunk;
    const uint32_t lo = (uint32_t)(((uint64_t)steps * blockIdx.y) / splits) * kChunk;
    const uint32_t hi_raw = (uint32_t)(((uint64_t)steps * (blockIdx.y + 1)) / splits) * kChunk;
    const uint32_t hi = hi_raw < ns ? hi_raw : ns;

    uint32_t total = 0, parity = 0;
    uint32_t k0, k1;
    float4 g0 = load_item(items, lo + tid, k0), g1 = load_item(items, lo + kThreads + tid, k1);
    for (uint32_t base = lo; base < hi; base += kChunk, parity ^= 1u) {
        const float4 g[2] = {g0, g1};
        const uint32_t kk[2] = {k0, k1};
        g0 = load_item(items, base + kChunk + tid, k0);
        g1 = load_item(items, base + kChunk + kThreads + tid, k1);
        if (total > (uint32_t)(kBinCap - kChunk)) {
            bin_cells_of_block(a, s_rec, s_idx, total, s_cellfr, s_cnt, s_base, s_pos, s_cellid);
            total = 0;
        }
        bool keep[2];
        float4 rec[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t k = base + (uint32_t)h * kThreads + tid;
            const float ox = cam.ox - g[h].x, oy = cam.oy - g[h].y, oz = cam.oz - g[h].z;
            const float oo = ox * ox + oy * oy + oz * oz;
            const float cc = oo - (g[h].w * g[h].w);
            float mg;
            bool inside;
            const bool culled = tile_culls_moving(fr, ox, oy, oz, oo, g[h].w, a.bin_theta, a.bin_delta, mg, inside);
            const bool outside = cc > 0.0f && !inside; // every camera the lists are for is strictly outside the sphere
            keep[h] = (k < hi) && !(outside && culled);
            // a camera inside or on the sphere: never culled, by any pyramid
            rec[h] = make_float4(ox, oy, oz, outside ? mg : __builtin_inff());
        }
        const unsigned long long m0 = __ballot(keep[0]), m1 = __ballot(keep[1]);
        if (lane == 0) {
            s_wcnt[parity][wave] = (uint32_t)__popcll(m0);
            s_wcnt[parity][4 + wave] = (uint32_t)__popcll(m1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        uint32_t before = 0, before1 = 0, first_total = 0, sum = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4u; w++) {
            const uint32_t c0 = s_wcnt[parity][w], c1 = s_wcnt[parity][4 + w];
            first_total += c0;
            sum += c0 + c1;
            if (w < wave) {
                before += c0;
                before1 += c1;
            }
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        if (keep[0]) {
            const uint32_t pos = total + before + (uint32_t)__popcll(m0 & below);
            s_rec[pos] = rec[0];
            s_idx[pos] = kk[0];
        }
        if (keep[1]) {
            const uint32_t pos = total + first_total + before1 + (uint32_t)__popcll(m1 & below);
            s_rec[pos] = rec[1];
            s_idx[pos] = kk[1];
        }
        total = __builtin_amdgcn_readfirstlane(total + sum);
    }
    if (total) {
        bin_cells_of_block(a, s_rec, s_idx, total, s_cellfr, s_cnt, s_base, s_pos, s_cellid);
    }
}

// Zero-fills [begin, end) of the frame (bytes, 4-aligned): the 8-bit modes' unused tail.
__global__ __launch_bounds__(kThreads) void rtx_zero_fill(uint32_t* p, size_t n_words)
{
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n_words; i += (size_t)gridDim.x * kThreads) {
        p[i] = 0u;
    }
}

// rtx_order_tiles: the dispatch order of the next launches' macro tiles from the work estimates the workgroups of
// an earlier launch left in tile_cost.  One workgroup; a counting sort over 1024 cost classes (heaviest class first;
// the order within a class is whatever the atomics give -- any permutation renders the same frame, so only speed
// depends on it, and every tile receives exactly one rank by construction whatever tile_cost holds).
//
// (For grids of several dispatch rounds, and small ones; grids of one round go through rtx_balance_tiles below.)
// Why order at all: the first 7 workgroups per CU of a launch all become resident at once (1792 of the 2040 that
// config 2 had with 4 sub-tiles per workgroup), blocks b, b + n_cu, ... on one CU (observed: round-robin; nothing
// depends on it but speed), and finish when their CU has worked through whatever it was dealt.  In frame order the tiles of a CU differ by 3x in work and the expensive rows come last:
// the slowest CU takes a third longer than the average and the second-round workgroups start late and run long
// (profiles/r02_b_stamps_sub4.txt).  So: the tiles go out heaviest first, the first `first_round` of them dealt
// boustrophedon over the CUs (round k ascending for even k, descending for odd k) so that every CU's share
// carries the same work, and the light remainder fills the slots as they free up.
constexpr int kOrderThreads = 1024, kOrderBins = 1024;

// With `base` (the static XCD-aware order of a two-level grid: position -> packed tile, blocks b, b + 8, ... share an XCD and are
// handed whole cells, rtx_plan.hpp xcd_cell_order) the sort is PER LABEL: the tiles at the positions p = L mod 8 of `base` are
// put back, heaviest first, at the same positions -- every XCD keeps its cells (and its L2 their lists and spheres: config 5's
// trace kernel fetches 6 MB per launch this way and 19 MB under a plain heaviest-first order), within an XCD the heavy tiles
// come first.  1024 classes = 8 labels x 128 cost classes; a tile's position is 8 x (its rank within the label) + label.
__global__ __launch_bounds__(kOrderThreads) void rtx_order_tiles(const uint32_t* __restrict__ cost, uint32_t n, uint32_t gx, uint32_t n_cu,
                                                                  uint32_t first_round, const uint32_t* __restrict__ base, uint32_t* __restrict__ order)
{
    __shared__ uint32_t s_hist[kOrderBins];
    __shared__ uint32_t s_lo[kOrderThreads / 64], s_hi[kOrderThreads / 64];
    __shared__ uint32_t s_start[8];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (uint32_t i = tid; i < n; i += kOrderThreads) {
        const uint32_t c = cost[i];
        lo = c < lo ? c : lo;
        hi = c > hi ? c : hi;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t l2 = __shfl_xor(lo, d), h2 = __shfl_xor(hi, d);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if (lane == 0u) {
        s_lo[wave] = lo;
        s_hi[wave] = hi;
    }
    s_hist[tid] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kOrderThreads / 64; k++) {
        lo = s_lo[k] < lo ? s_lo[k] : lo;
        hi = s_hi[k] > hi ? s_hi[k] : hi;
    }
    const uint64_t range = (uint64_t)(hi - lo) + 1u;
    auto bin_of = [&](uint32_t c) { return (uint32_t)(((uint64_t)(hi - c) * (uint64_t)kOrderBins) / range); }; // 0 = heaviest
    // per label: the class of the tile at position p of `base`
    auto tile_at = [&](uint32_t p) { const uint32_t packed = base[p]; return (packed >> 16) * gx + (packed & 0xffffu); };
    auto class_at = [&](uint32_t p) { return ((p & 7u) << 7) | (bin_of(cost[tile_at(p)]) >> 3); };
    for (uint32_t i = tid; i < n; i += kOrderThreads) {
        atomicAdd(&s_hist[base != nullptr ? class_at(i) : bin_of(cost[i])], 1u);
    }
    __syncthreads();
    // exclusive prefix sum over the classes (one per thread): wave scan, then the waves' totals
    const uint32_t mine = s_hist[tid];
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= (uint32_t)d) incl += o;
    }
    __syncthreads();
    if (lane == 63u) s_lo[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (int k = 0; k < kOrderThreads / 64; k++) {
        before += (uint32_t)k < wave ? s_lo[k] : 0u;
    }
    s_hist[tid] = before + incl - mine; // now: next free rank of the class
    if ((tid & 127u) == 0u) s_start[tid >> 7] = before + incl - mine; // (per label: where its ranks begin)
    __syncthreads();
    if (base != nullptr) {
        for (uint32_t p = tid; p < n; p += kOrderThreads) {
            const uint32_t label = p & 7u;
            const uint32_t rank = atomicAdd(&s_hist[class_at(p)], 1u) - s_start[label];
            order[8u * rank + label] = base[p]; // (the label's set has exactly as many tiles as there are positions = label mod 8)
        }
        return;
    }
    const uint32_t rounds = n_cu ? first_round / n_cu : 0u; // whole rounds dealt boustrophedon
    for (uint32_t i = tid; i < n; i += kOrderThreads) {
        const uint32_t rank = atomicAdd(&s_hist[bin_of(cost[i])], 1u);
        uint32_t blk = rank;
        if (n_cu && rank < rounds * n_cu) {
            const uint32_t k = rank / n_cu, c = rank - k * n_cu;
            blk = (k & 1u) ? k * n_cu + (n_cu - 1u - c) : rank;
        }
        const uint32_t by = i / gx, bx = i - by * gx;
        order[blk] = bx | (by << 16);
    }
}

// rtx_balance_tiles: the dispatch order for a tile grid whose workgroups are all resident at once (n <= slots per CU x
// n_cu; one workgroup, like rtx_order_tiles).  Then the blocks b, b + n_cu, b + 2 n_cu, ... share a compute unit
// (observed on every launch: profiles/r02_f_cu_balance.md; which physical CU varies, the groups do not), nothing is
// refilled, and the launch ends when the slowest group does -- in 1080p frame order a third later than the average
// one.  Two steps:
//   feedback  The instruction-count estimate ranks tiles well but is off by +-14 % on a group's sum (dependent
//             chains, the planes, divergence), while a group's duration repeats from launch to launch to 0.3 us.  So the
//             trace workgroups also leave their start and end times by dispatch position, and every tile carries a
//             correction factor: the tiles of a group that took longer than the mean group are scaled up, the others
//             down (damped, clamped), launch after launch.  cost' = (estimate + set-up) x factor.
//   dealing   The n - (rounds-1) n_cu lightest tiles are the last round (the hardware puts those blocks on groups
//             0, 1, ...).  The other rounds go heaviest first, and within a round the k-th heaviest tile goes to the
//             group with the k-th smallest sum so far.
// Any permutation renders the same frame; every position receives exactly one tile whatever the inputs hold (ranks come
// from counting, positions from ranks).
RTX_X_BALANCE_STAMPS();
// The pass runs beside the next frame's trace launch (a stream of its own), so it is shaped to fit into a slot that
// launch leaves free: 256 threads, the trace kernel's register budget, under 38 KB of LDS -- a CU holding six trace
// workgroups has room for exactly that.  (As one 1024-thread workgroup it needed a CU to itself and the frame beside it
// paid 20 us for the seven workgroups that CU could not take.)
constexpr int kBalanceMaxTiles = 2048;
constexpr int kBalanceThreads = 256;
constexpr int kBalancePerThread = kBalanceMaxTiles / kBalanceThreads;
constexpr int kBalanceBins = 1024;
constexpr uint32_t kCostSetup = 1800u; // per workgroup: tables, planes, staging (4 waves x ~4.5 passes' worth of time)

__global__ __launch_bounds__(kBalanceThreads, RTX_WAVES_PER_EU) void rtx_balance_tiles(const uint32_t* __restrict__ cost, uint32_t n, uint32_t gx,
                                                                                       uint32_t G, const uint32_t* prev_order, float* factor,
                                                                                       uint32_t have_factor, uint32_t have_times, uint32_t* order,
                                                                                       float damping, float max_step)
{
    // s_a: the estimates by tile, later the corrected costs by rank; s_b: the tile that ran at each position, later the
    // tile by rank
    __shared__ uint32_t s_a[kBalanceMaxTiles];
    __shared__ uint16_t s_b[kBalanceMaxTiles];
    __shared__ float s_f[kBalanceMaxTiles];        // factor by tile
    __shared__ float s_c[kBalanceMaxTiles];        // corrected cost by tile; before that, per group: earliest start, latest end
    int32_t* s_first = reinterpret_cast<int32_t*>(s_c);
    int32_t* s_last = s_first + kOrderThreads;
    static_assert(2 * kOrderThreads <= kBalanceMaxTiles, "the groups' times share s_c");
    __shared__ uint32_t s_hist[kBalanceBins];
    __shared__ float s_sum[kOrderThreads];         // per group: duration, then the sum dealt so far
    __shared__ float s_red[kBalanceThreads / 64];
    __shared__ uint32_t s_lo[kBalanceThreads / 64], s_hi[kBalanceThreads / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    constexpr int kWaves = kBalanceThreads / 64;

    BSTAMP(0);
    for (uint32_t g = tid; g < (uint32_t)kOrderThreads; g += kBalanceThreads) {
        s_first[g] = 0x7fffffff;
        s_last[g] = (int32_t)0x80000000;
        s_hist[g] = 0u;
    }
    const uint32_t ref = have_times ? cost[n] : 0u;
    __syncthreads();
    // ---- everything this needs from memory, requested before anything is used (clamped indices instead of branches:
    // one round trip instead of eight); then the groups' first start and last end (a group's clocks are one XCD's:
    // only differences within a group are used)
    uint32_t c_in[kBalancePerThread], t0[kBalancePerThread], t1[kBalancePerThread], pk[kBalancePerThread];
    float f_in[kBalancePerThread];
#pragma unroll
    for (int q = 0; q < kBalancePerThread; q++) {
        const uint32_t i = tid + (uint32_t)q * kBalanceThreads;
        const uint32_t ii = i < n ? i : n - 1u;
        c_in[q] = cost[ii];
        t0[q] = have_times ? cost[n + ii] : 0u;
        t1[q] = have_times ? cost[2u * n + ii] : 0u;
        pk[q] = prev_order != nullptr ? prev_order[ii] : 0u;
        f_in[q] = have_factor ? factor[ii] : 1.0f;
    }
#pragma unroll
    for (int q = 0; q < kBalancePerThread; q++) {
        const uint32_t i = tid + (uint32_t)q * kBalanceThreads;
        if (i < n) {
            const uint32_t po = prev_order != nullptr ? (pk[q] >> 16) * gx + (pk[q] & 0xffffu) : i;
            s_a[i] = c_in[q];
            s_b[i] = (uint16_t)(po < n ? po : i);
            s_f[i] = (f_in[q] >= 0.5f && f_in[q] <= 2.0f) ? f_in[q] : 1.0f;
            if (have_times) {
                atomicMin(&s_first[i % G], (int32_t)(t0[q] - ref));
                atomicMax(&s_last[i % G], (int32_t)(t1[q] - ref));
            }
        }
    }
    __syncthreads();
    BSTAMP(1);
    // ---- feedback: how long each group took, against the mean
    float tot = 0.0f;
    for (uint32_t g = tid; g < G; g += kBalanceThreads) {
        float dur = 0.0f;
        if (have_times) {
            dur = (float)(s_last[g] - s_first[g]);
            dur = (dur > 0.0f && dur < 1.0e8f) ? dur : 0.0f; // (a second: nonsense)
        }
        s_sum[g] = dur;
        tot += dur;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d);
    if (lane == 0u) s_red[wave] = tot;
    __syncthreads();
    tot = 0.0f;
#pragma unroll
    for (int k = 0; k < kWaves; k++) tot += s_red[k];
    const float mean = tot / (float)G;
    if (have_times && mean > 0.0f) {
        for (uint32_t p = tid; p < n; p += kBalanceThreads) { // a dispatch position; s_b: the tile that ran there
            const float d = s_sum[p % G];
            if (d > 0.0f) {
                float r = 1.0f + damping * (d / mean - 1.0f);
                r = r < 1.0f - max_step ? 1.0f - max_step : (r > 1.0f + max_step ? 1.0f + max_step : r);
                const uint32_t t = s_b[p];                      // (a permutation: every tile once)
                const float f = s_f[t] * r;
                s_f[t] = f < 0.5f ? 0.5f : (f > 2.0f ? 2.0f : f);
            }
        }
    }
    __syncthreads();
    BSTAMP(2);
    // ---- corrected costs; ranks by a counting sort over 1024 classes, heaviest first
    uint32_t lo = 0xffffffffu, hi = 0u;
    float total = 0.0f;
    for (uint32_t i = tid; i < n; i += kBalanceThreads) {
        const float f = s_f[i];
        factor[i] = f;
        const uint32_t c_in = s_a[i];
        const float c = (float)(c_in < (1u << 24) ? c_in + kCostSetup : (1u << 24)) * f;
        s_c[i] = c;
        total += c;
        const uint32_t u = (uint32_t)c;
        lo = u < lo ? u : lo;
        hi = u > hi ? u : hi;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t l2 = __shfl_xor(lo, d), h2 = __shfl_xor(hi, d);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
        total += __shfl_xor(total, d);
    }
    if (lane == 0u) {
        s_lo[wave] = lo;
        s_hi[wave] = hi;
        s_red[wave] = total;
    }
    __syncthreads();
    total = 0.0f;
#pragma unroll
    for (int k = 0; k < kWaves; k++) {
        lo = s_lo[k] < lo ? s_lo[k] : lo;
        hi = s_hi[k] > hi ? s_hi[k] : hi;
        total += s_red[k];
    }
    const float to_bin = (float)kBalanceBins / ((float)(hi - lo) + 1.0f);
    auto bin_of = [&](float c) { // 0 = heaviest
        const uint32_t b = (uint32_t)((float)(hi - (uint32_t)c) * to_bin);
        return b < (uint32_t)kBalanceBins ? b : (uint32_t)kBalanceBins - 1u;
    };
    // exclusive prefix sums over the 1024 classes in s_hist: four consecutive classes per thread, wave scan, wave totals
    auto scan_hist = [&]() {
        const uint32_t h0 = s_hist[4u * tid], h1 = s_hist[4u * tid + 1u], h2 = s_hist[4u * tid + 2u], h3 = s_hist[4u * tid + 3u];
        const uint32_t mine = h0 + h1 + h2 + h3;
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= (uint32_t)d) incl += o;
        }
        if (lane == 63u) s_lo[wave] = incl;
        __syncthreads();
        uint32_t before = incl - mine;
#pragma unroll
        for (int k = 0; k < kWaves; k++) {
            before += (uint32_t)k < wave ? s_lo[k] : 0u;
        }
        s_hist[4u * tid] = before;
        s_hist[4u * tid + 1u] = before + h0;
        s_hist[4u * tid + 2u] = before + h0 + h1;
        s_hist[4u * tid + 3u] = before + h0 + h1 + h2;
        __syncthreads();
    };
    for (uint32_t i = tid; i < n; i += kBalanceThreads) {
        atomicAdd(&s_hist[bin_of(s_c[i])], 1u);
    }
    __syncthreads();
    scan_hist();
    BSTAMP(3);
    for (uint32_t i = tid; i < n; i += kBalanceThreads) {
        const float c = s_c[i];
        const uint32_t rank = atomicAdd(&s_hist[bin_of(c)], 1u);
        s_a[rank] = __float_as_uint(c); // (the estimates are not needed any more, nor the positions)
        s_b[rank] = (uint16_t)i;
    }
    __syncthreads();
    BSTAMP(4);
    // ---- dealing
    auto put = [&](uint32_t pos, uint32_t rank) {
        const uint32_t t = s_b[rank];
        const uint32_t ty = t / gx, tx = t - ty * gx;
        order[pos] = tx | (ty << 16);
    };
    const uint32_t rounds = (n + G - 1u) / G, tail0 = (rounds - 1u) * G;
    for (uint32_t g = tid; g < G; g += kBalanceThreads) {
        float sum0 = 0.0f;
        if (tail0 + g < n) {
            sum0 = __uint_as_float(s_a[tail0 + g]);
            put(tail0 + g, tail0 + g);
        }
        s_sum[g] = sum0;
    }
    // Per round: rank the groups by their sums with the same counting sort (1024 classes over [0, 1.25 x the mean final
    // sum]: a class is 0.12 % of a group's work wide, far below what the estimates are good for; ties take the order the
    // atomics give), then the group with the k-th smallest sum takes the k-th heaviest tile of the round.
    const float sum_to_bin = total > 0.0f ? (float)kBalanceBins * (float)G / (1.25f * total) : 0.0f;
    auto sum_bin = [&](float v) {
        const uint32_t b = (uint32_t)(v * sum_to_bin);
        return b < (uint32_t)kBalanceBins ? b : (uint32_t)kBalanceBins - 1u;
    };
    BSTAMP(5);
    for (uint32_t r = 0; r + 1u < rounds; r++) {
        for (uint32_t k = tid; k < (uint32_t)kBalanceBins; k += kBalanceThreads) s_hist[k] = 0u;
        __syncthreads();
        for (uint32_t g = tid; g < G; g += kBalanceThreads) {
            atomicAdd(&s_hist[sum_bin(s_sum[g])], 1u);
        }
        __syncthreads();
        scan_hist();
        for (uint32_t g = tid; g < G; g += kBalanceThreads) {
            const float v = s_sum[g];
            const uint32_t k = atomicAdd(&s_hist[sum_bin(v)], 1u); // < G: the classes hold G groups in all
            put(r * G + g, r * G + k);
            s_sum[g] = v + __uint_as_float(s_a[r * G + k]);
        }
        __syncthreads();
    }
    BSTAMP(6);
}

// rtx_expand: compact pixel words -> records, for up to kMaxExpandSeg segments (a segment = a run of pixels that
// is contiguous in both buffers: one rank's rows of one frame).  Pure streaming: 4 bytes read, S written per pixel.
// A workgroup takes kExpandPixels consecutive pixels of one segment, 256 at a time; each wave transposes the
// records of its 64 pixels through its own piece of LDS (pixel p at dword p*S/4, odd stride: conflict-free), so
// that its 64*S bytes leave as whole 16-byte stores in address order -- wave-local, no workgroup barrier.
template <int MODE>
__global__ __launch_bounds__(kThreads) void rtx_expand_words(const ExpandArgs e)
{
    constexpr bool kRgb = (MODE == RTX_K_RGB_ASCII || MODE == RTX_K_RGB_PIXEL || MODE == RTX_K_RGB_NORMALS);
    constexpr uint32_t SW = kRgb ? 5u : 3u; // dwords per record
    __shared__ uint32_t s_digits[256];
    __shared__ __attribute__((aligned(16))) uint32_t s_rec[kThreads * SW];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    s_digits[tid] = digits_word(tid);
    uint32_t k = 0;
    while (k + 1u < e.nseg && blockIdx.x >= e.first_block[k + 1u]) {
        k++;
    }
    const uint32_t npix = e.npix[k];
    const uint32_t* src = e.src + e.src_px[k];
    uint32_t* dst_seg = reinterpret_cast<uint32_t*>(e.dst) + e.dst_px[k] * SW;
    const uint32_t base = (blockIdx.x - e.first_block[k]) * (uint32_t)kExpandPixels;
    uint32_t* s_wave = s_rec + wave * 64u * SW;
    __syncthreads();
#pragma unroll 1
    for (uint32_t c = 0; c < (uint32_t)kExpandPixels; c += (uint32_t)kThreads) {
        const uint32_t w0 = base + c + wave * 64u; // first pixel of this wave's 64
        if (w0 >= npix) {
            break;
        }
        const uint32_t p = w0 + lane;
        const bool valid = p < npix;
        const uint32_t word = valid ? src[p] : kCompactNewline;
        uint32_t w[SW];
        if (word == kCompactNewline) {
#pragma unroll
            for (uint32_t i = 0; i < SW; i++) {
                w[i] = 0u;
            }
        } else {
            Fields f;
            f.c0 = word & 255u;
            f.c1 = (word >> 8) & 255u;
            f.c2 = (word >> 16) & 255u;
            f.glyph = word >> 24;
            record_words<MODE>(word != kCompactMiss, f, s_digits, w);
        }
        uint32_t* dst = dst_seg + (size_t)w0 * SW;
        if (e.aligned16 && w0 + 64u <= npix) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the previous round's reads of s_wave are done
#pragma unroll
            for (uint32_t i = 0; i < SW; i++) {
                s_wave[lane * SW + i] = w[i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // wave-local: LDS is in order within a wave
            const uint4* s4 = reinterpret_cast<const uint4*>(s_wave);
            uint4* d4 = reinterpret_cast<uint4*>(dst);
            constexpr uint32_t n4 = 16u * SW; // 16-byte pieces of the wave's 64 records: 80 or 48
            if (lane < n4) {
                d4[lane] = s4[lane];
            }
            if (64u + lane < n4) {
                d4[64u + lane] = s4[64u + lane];
            }
        } else if (valid) {
#pragma unroll
            for (uint32_t i = 0; i < SW; i++) {
                dst[(size_t)lane * SW + i] = w[i];
            }
        }
    }
}

} // namespace rtx

extern "C" int rtx_k_launch_expand(const ExpandArgs* e, int mode, unsigned blocks, void* stream_v)
{
    using namespace rtx;
    hipStream_t stream = (hipStream_t)stream_v;
    switch (mode) {
    case RTX_K_BIT_ASCII: hipLaunchKernelGGL((rtx_expand_words<RTX_K_BIT_ASCII>), dim3(blocks), dim3(kThreads), 0, stream, *e); break;
    case RTX_K_BIT_PIXEL: hipLaunchKernelGGL((rtx_expand_words<RTX_K_BIT_PIXEL>), dim3(blocks), dim3(kThreads), 0, stream, *e); break;
    case RTX_K_RGB_ASCII: hipLaunchKernelGGL((rtx_expand_words<RTX_K_RGB_ASCII>), dim3(blocks), dim3(kThreads), 0, stream, *e); break;
    case RTX_K_RGB_PIXEL: hipLaunchKernelGGL((rtx_expand_words<RTX_K_RGB_PIXEL>), dim3(blocks), dim3(kThreads), 0, stream, *e); break;
    case RTX_K_RGB_NORMALS: hipLaunchKernelGGL((rtx_expand_words<RTX_K_RGB_NORMALS>), dim3(blocks), dim3(kThreads), 0, stream, *e); break;
    default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

extern "C" const char* rtx_k_launch_trace(const KArgs* a, int mode, int cull, void* stream_v, int* hip_error)
{
    using namespace rtx;
    hipStream_t stream = (hipStream_t)stream_v;
    const uint32_t lw = a->tile_log2w;
    const uint32_t tw = 1u << lw, th = (uint32_t)kThreads >> lw;
    const uint32_t nx = 1u << a->sub_log2nx, ny = a->nsub >> a->sub_log2nx;
    const uint32_t mw = tw * nx, mh = th * ny;
    *hip_error = 0;
    if (a->nsub == 0 || nx * ny != a->nsub || mw > (uint32_t)kMaxMacro || mh > (uint32_t)kMaxMacro || mw + mh > (uint32_t)kThreads) {
        return nullptr;
    }
    if (cull && a->refine && (mw > (uint32_t)kMaxMacroRefine || mh > (uint32_t)kMaxMacroRefine)) {
        return nullptr; // (rtx_render_rows does not ask for this)
    }
    const uint32_t rows = a->row_end - a->row0;
    dim3 grid((a->W + mw - 1u) / mw, (rows + mh - 1u) / mh, 1), block(kThreads, 1, 1);
    const char* name = nullptr;
    const unsigned lds_pad = rtx_x_lds_pad(); // 0 in the product build
#define RTX_LAUNCH(M, C, O, SUFFIX)                                                          \
    do {                                                                                     \
        if (C && a->refine) {                                                                \
            hipLaunchKernelGGL((rtx_trace<M, C, O, C>), grid, block, lds_pad, stream, *a);   \
            name = "rtx_trace<" #M "," #C SUFFIX ",refine>";                                 \
        } else {                                                                             \
            hipLaunchKernelGGL((rtx_trace<M, C, O, false>), grid, block, lds_pad, stream, *a); \
            name = "rtx_trace<" #M "," #C SUFFIX ">";                                        \
        }                                                                                    \
    } while (0)
#define RTX_LAUNCH_OUT(M, C)                                   \
    do {                                                       \
        if (a->compact == 0u) {                                \
            RTX_LAUNCH(M, C, kOutRecords, "");                 \
        } else if (a->compact == 1u) {                         \
            RTX_LAUNCH(M, C, kOutCompact, ",compact");         \
        } else {                                               \
            RTX_LAUNCH(M, C, kOutValues, ",values");           \
        }                                                      \
    } while (0)
#define RTX_LAUNCH_MODE(M)            \
    do {                              \
        if (cull) {                   \
            RTX_LAUNCH_OUT(M, true);  \
        } else {                      \
            RTX_LAUNCH_OUT(M, false); \
        }                             \
    } while (0)
    switch (mode) {
    case RTX_K_BIT_ASCII: RTX_LAUNCH_MODE(RTX_K_BIT_ASCII); break;
    case RTX_K_BIT_PIXEL: RTX_LAUNCH_MODE(RTX_K_BIT_PIXEL); break;
    case RTX_K_RGB_ASCII: RTX_LAUNCH_MODE(RTX_K_RGB_ASCII); break;
    case RTX_K_RGB_PIXEL: RTX_LAUNCH_MODE(RTX_K_RGB_PIXEL); break;
    case RTX_K_RGB_NORMALS: RTX_LAUNCH_MODE(RTX_K_RGB_NORMALS); break;
    case RTX_K_SDL:
        // RayTrace_SDL stores nothing, whatever the output form
        if (cull) {
            RTX_LAUNCH(RTX_K_SDL, true, kOutRecords, "");
        } else {
            RTX_LAUNCH(RTX_K_SDL, false, kOutRecords, "");
        }
        break;
    default: return nullptr;
    }
#undef RTX_LAUNCH_MODE
#undef RTX_LAUNCH_OUT
#undef RTX_LAUNCH
    *hip_error = (int)hipGetLastError();
    return name;
}

extern "C" const char* rtx_k_launch_trace_batch(const KArgs* a, const KBatch* kb, int mode, void* stream_v, int* hip_error)
{
    using namespace rtx;
    hipStream_t stream = (hipStream_t)stream_v;
    const uint32_t lw = a->tile_log2w;
    const uint32_t tw = 1u << lw, th = (uint32_t)kThreads >> lw;
    const uint32_t nx = 1u << a->sub_log2nx, ny = a->nsub >> a->sub_log2nx;
    const uint32_t mw = tw * nx, mh = th * ny;
    *hip_error = 0;
    if (a->nsub == 0 || nx * ny != a->nsub || mw > (uint32_t)kMaxMacro || mh > (uint32_t)kMaxMacro || mw + mh > (uint32_t)kThreads || a->refine ||
        a->batch_n == 0 || a->batch_n > (uint32_t)kMaxBatch || a->compact > 1u) {
        return nullptr;
    }
    const uint32_t rows = a->row_end - a->row0;
    const uint64_t gx = (a->W + mw - 1u) / mw, gy = (rows + mh - 1u) / mh, blocks = gx * gy * a->batch_n;
    if (gx != a->batch_gx || gy != a->batch_gy || blocks == 0 || blocks >= (1ull << 31)) {
        return nullptr;
    }
    const dim3 grid((uint32_t)blocks, 1, 1), block(kThreads, 1, 1);
    const char* name = nullptr;
    const unsigned lds_pad = rtx_x_lds_pad(); // 0 in the product build
#define RTX_LAUNCH_BATCH(M)                                                                               \
    do {                                                                                                  \
        if (a->compact == 0u) {                                                                           \
            hipLaunchKernelGGL((rtx_trace_batch<M, kOutRecords>), grid, block, lds_pad, stream, *a, *kb); \
            name = "rtx_trace_batch<" #M ">";                                                             \
        } else {                                                                                          \
            hipLaunchKernelGGL((rtx_trace_batch<M, kOutCompact>), grid, block, lds_pad, stream, *a, *kb); \
            name = "rtx_trace_batch<" #M ",compact>";                                                     \
        }                                                                                                 \
    } while (0)
    switch (mode) {
    case RTX_K_BIT_ASCII: RTX_LAUNCH_BATCH(RTX_K_BIT_ASCII); break;
    case RTX_K_BIT_PIXEL: RTX_LAUNCH_BATCH(RTX_K_BIT_PIXEL); break;
    case RTX_K_RGB_ASCII: RTX_LAUNCH_BATCH(RTX_K_RGB_ASCII); break;
    case RTX_K_RGB_PIXEL: RTX_LAUNCH_BATCH(RTX_K_RGB_PIXEL); break;
    case RTX_K_RGB_NORMALS: RTX_LAUNCH_BATCH(RTX_K_RGB_NORMALS); break;
    default: return nullptr;
    }
#undef RTX_LAUNCH_BATCH
    *hip_error = (int)hipGetLastError();
    return name;
}

extern "C" int rtx_k_launch_bin_cells(const KArgs* a, unsigned splits, void* stream_v)
{
    const unsigned blocks = ((a->cells_x + 3u) >> 2) * ((a->cells_y + 3u) >> 2);
    hipLaunchKernelGGL(rtx::rtx_bin_cells, dim3(blocks, splits, 1), dim3(rtx::kThreads), 0, (hipStream_t)stream_v, *a);
    return (int)hipGetLastError();
}

extern "C" int rtx_k_launch_order_tiles(const uint32_t* tile_cost, uint32_t n_tiles, uint32_t gx, uint32_t n_cu, uint32_t first_round,
                                        const uint32_t* base_order, uint32_t* tile_order, void* stream_v)
{
    if (n_tiles == 0 || gx == 0 || gx > 0xffffu || (n_tiles + gx - 1u) / gx > 0xffffu) {
        return (int)hipErrorInvalidValue;
    }
    if (first_round > n_tiles) first_round = n_tiles;
    hipLaunchKernelGGL(rtx::rtx_order_tiles, dim3(1), dim3(rtx::kOrderThreads), 0, (hipStream_t)stream_v, tile_cost, n_tiles, gx, n_cu, first_round, base_order, tile_order);
    return (int)hipGetLastError();
}

extern "C" int rtx_k_launch_balance_tiles(const uint32_t* tile_cost, uint32_t n_tiles, uint32_t gx, uint32_t n_cu, const uint32_t* prev_order,
                                          float* factor, int have_factor, int have_times, uint32_t* tile_order, void* stream_v)
{
    if (n_tiles == 0 || n_tiles > (uint32_t)rtx::kBalanceMaxTiles || gx == 0 || n_cu == 0 || n_cu > (uint32_t)rtx::kOrderThreads) {
        return (int)hipErrorInvalidValue;
    }
    RTX_X_BALANCE_DEBUG();
    // how much of a group's deviation from the mean duration goes into its tiles' factors per pass, and the largest step
    float damping = 0.7f, max_step = 0.18f;
    RTX_X_BALANCE_PARAMS(damping, max_step);
    hipLaunchKernelGGL(rtx::rtx_balance_tiles, dim3(1), dim3(rtx::kBalanceThreads), 0, (hipStream_t)stream_v, tile_cost, n_tiles, gx, n_cu, prev_order,
                       factor, (uint32_t)have_factor, (uint32_t)have_times, tile_order, damping, max_step);
    return (int)hipGetLastError();
}

extern "C" int rtx_k_launch_zero(void* p, size_t bytes, void* stream_v)
{
    if (bytes == 0) {
        return 0;
    }
    const size_t words = bytes / 4;
    size_t blocks = (words + rtx::kThreads - 1) / rtx::kThreads;
    if (blocks > 2048) {
        blocks = 2048;
    }
    hipLaunchKernelGGL(rtx::rtx_zero_fill, dim3((unsigned)blocks), dim3(rtx::kThreads), 0, (hipStream_t)stream_v, (uint32_t*)p, words);
    return (int)hipGetLastError();
}
