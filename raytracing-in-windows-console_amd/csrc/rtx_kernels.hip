// rtx_kernels.hip -- gfx950 kernels of the ray-trace hot path.
//
// One thread per pixel (as the reference, RayTracingManager.cu:120-125), 256-thread workgroups
// over a 2^lw x 2^(8-lw) pixel tile.  Each workgroup walks the sphere array in chunks of 256
// (coalesced float4 loads), hoists the ray-independent terms otc = o - c and
// cc = Dot(otc,otc) - r*r (Sphere.cu:34-37), optionally culls spheres whose inflated bound
// cannot touch the tile's frustum, and appends survivors to a candidate list in LDS.  When the
// list fills (or the scene ends) every thread runs the reference's exact ray/sphere test over
// the list (wave-uniform index, LDS broadcast reads).  Planes are tested from scalar loads.
// The winner alone is shaded and encoded (the reference re-derives normal/colour on every
// improving hit, RayTracing.cu:123-135, but only the last survives).
//
// Closest hit = lexicographic minimum of (t, creation index): the same object the reference's
// in-order scan with strict '<' keeps (RayTracing.cu:123).
#include "rtx_device.hpp"
#include "rtx_kernels.h"

namespace rtx {

constexpr int kThreads = 256;
constexpr int kListCap = 1024;     // candidate records per flush: 16 KiB + 4 KiB of LDS
constexpr float kNoHit = 99999999.f; // RayTracing.h:21

// RayTracing.h:97-115 (68 glyphs).
__constant__ const char kRamp[68] = {
    ' ', '.', '`', '^', '"', ',', ':', ';', 'I', 'l', '!', 'i', '>', '<', '~', '+', '_',
    '-', '?', '*', ']', '[', '}', '{', '1', ')', '(', '|', '/', 't', 'f', 'j', 'r', 'x',
    'n', 'u', 'v', 'c', 'z', 'm', 'w', 'X', 'Y', 'U', 'J', 'C', 'L', 'q', 'p', 'd', 'b',
    'k', 'h', 'a', 'o', '#', '%', 'Z', 'O', '8', 'B', '$', '0', 'Q', 'M', '&', 'W', '@'
};

// Conservative inflation of a sphere for culling.  A ray whose fp32 test reports a hit passes,
// in exact arithmetic, within R = sqrt(r^2 (1+2u) + 15u |otc|^2) of the centre (u = 2^-24;
// derivation in DESIGN.md "Culling soundness"), and no further than 3u|otc| behind the apex.
// kappa = 4e-6 > 4 * 15u; sqrt(a+b) <= sqrt(a) + sqrt(b) gives the linear form used here, and
// kSlack covers the rounding of the plane evaluation itself.
constexpr float kKappa = 4.0e-6f;
constexpr float kSqrtKappaPlusSlack = 2.0e-3f + 1.0e-4f;

struct TileFrustum {
    V3 n[5]; // inward unit normals of the four side planes, then the tile axis
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

// Side planes of the pyramid spanned by the tile's pixel centres, grown by half a pixel on every
// side.  Pixel directions are linear in (cx, cy), so every pixel ray of the tile lies in the
// convex cone of the four corner directions.  A normal that cannot be oriented (degenerate
// matrix, NaN) becomes the zero vector, which never culls.
__device__ __forceinline__ TileFrustum tile_frustum(const Camera& c, uint32_t col0, uint32_t row0, uint32_t tw, uint32_t th)
{
    const float x0 = (2.0f * (float)col0 - 1.0f - c.fW) / c.fW;
    const float x1 = (2.0f * (float)(col0 + tw) - 1.0f - c.fW) / c.fW;
    const float y0 = (c.fH - 2.0f * (float)row0 + 1.0f) / c.fH;        // top edge (larger cy)
    const float y1 = (c.fH - 2.0f * (float)(row0 + th) + 1.0f) / c.fH;  // bottom edge
    const V3 c00 = view_dir(c, x0, y0), c10 = view_dir(c, x1, y0), c11 = view_dir(c, x1, y1), c01 = view_dir(c, x0, y1);
    const V3 axis = view_dir(c, 0.5f * (x0 + x1), 0.5f * (y0 + y1));
    V3 raw[4] = {cross(c00, c10), cross(c10, c11), cross(c11, c01), cross(c01, c00)};
    TileFrustum f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        V3 n = raw[k];
        float side = dot(n, axis);
        if (!(side > 0.0f)) {
            n = mulf(n, -1.0f);
            side = -side;
        }
        const float len2 = dot(n, n);
        if (side > 0.0f && len2 > 0.0f && len2 < 3.0e38f) {
            f.n[k] = mulf(n, 1.0f / sqrtf(len2));
        } else {
            f.n[k] = v3(0.0f, 0.0f, 0.0f);
        }
    }
    const float alen2 = dot(axis, axis);
    f.n[4] = (alen2 > 0.0f && alen2 < 3.0e38f) ? mulf(axis, 1.0f / sqrtf(alen2)) : v3(0.0f, 0.0f, 0.0f);
    return f;
}

// true when the sphere (hoisted form) provably cannot be hit by any pixel ray of the tile.
__device__ __forceinline__ bool tile_culls(const TileFrustum& f, float ox, float oy, float oz, float oo, float r)
{
    // margin >= R + 3u|otc| + evaluation slack (see kKappa)
    const float margin = r * (1.0f + kKappa) + kSqrtKappaPlusSlack * sqrtf(oo);
    bool out = false;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        // n . (c - o) = -(n . otc); outside when it is below -margin.  NaN compares false: keep.
        const float d = f.n[k].x * ox + f.n[k].y * oy + f.n[k].z * oz;
        out = out || (d > margin);
    }
    return out;
}

struct Best {
    float t;
    uint32_t k; // local sphere index, or 0xffffffff
};

template <int MODE>
__device__ __forceinline__ void encode_and_store(const KArgs& a, const Camera& cam, const Ray& ray, bool in_frame, bool is_newline_col,
                                                 uint32_t row, uint32_t col, float distance, V3 normal, V3 colour, float shadingValue)
{
    constexpr bool kRgb = (MODE == RTX_K_RGB_ASCII || MODE == RTX_K_RGB_PIXEL || MODE == RTX_K_RGB_NORMALS);
    constexpr uint32_t S = kRgb ? 20u : 12u;
    if (!in_frame) {
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
    const bool visible = distance <= cam.far; // RayTracing.cu:207,288,371,508,646
    const uint32_t ESC_BR = 0x1bu | (0x5bu << 8); // ESC [
    if (kRgb) {
        uint32_t w0, w1, w2, w3, w4;
        if (visible) {
            uint32_t r, g, b;
            if (MODE == RTX_K_RGB_NORMALS) {
                r = u8_sat(normal.x * 255.0f);
                g = u8_sat(normal.y * 255.0f);
                b = u8_sat(normal.z * 255.0f);
            } else {
                r = u8_sat(colour.x);
                g = u8_sat(colour.y);
                b = u8_sat(colour.z);
            }
            const uint32_t dr = digits3(r), dg = digits3(g), db = digits3(b);
            const uint32_t kind = (MODE == RTX_K_RGB_ASCII) ? '3' : '4';
            const uint32_t glyph = (MODE == RTX_K_RGB_ASCII) ? (uint32_t)(uint8_t)kRamp[ramp_index(shadingValue)] : (uint32_t)' ';
            w0 = ESC_BR | (kind << 16) | ((uint32_t)'8' << 24);
            w1 = (uint32_t)';' | ((uint32_t)'2' << 8) | ((uint32_t)';' << 16) | ((dr & 255u) << 24);
            w2 = (dr >> 8) | ((uint32_t)';' << 16) | ((dg & 255u) << 24);
            w3 = (dg >> 8) | ((uint32_t)';' << 16) | ((db & 255u) << 24);
            w4 = (db >> 8) | ((uint32_t)'m' << 16) | (glyph << 24);
        } else {
            // ESC [ 4 8 ; 2 ; \0 \0 0 ; \0 \0 0 ; \0 \0 0 m ' '
            w0 = ESC_BR | ((uint32_t)'4' << 16) | ((uint32_t)'8' << 24);
            w1 = (uint32_t)';' | ((uint32_t)'2' << 8) | ((uint32_t)';' << 16);
            w2 = ((uint32_t)'0' << 8) | ((uint32_t)';' << 16);
            w3 = ((uint32_t)'0' << 8) | ((uint32_t)';' << 16);
            w4 = ((uint32_t)'0' << 8) | ((uint32_t)'m' << 16) | ((uint32_t)' ' << 24);
        }
        dst[0] = w0;
        dst[1] = w1;
        dst[2] = w2;
        dst[3] = w3;
        dst[4] = w4;
    } else {
        uint32_t w0, w1, w2;
        if (visible) {
            const uint32_t index = ansi256_from_rgb(u8_sat(colour.x), u8_sat(colour.y), u8_sat(colour.z), a.grey);
            const uint32_t d = digits3(index);
            const uint32_t kind = (MODE == RTX_K_BIT_ASCII) ? '3' : '4';
            const uint32_t glyph = (MODE == RTX_K_BIT_ASCII) ? (uint32_t)(uint8_t)kRamp[ramp_index(shadingValue)] : (uint32_t)' ';
            w0 = ESC_BR | (kind << 16) | ((uint32_t)'8' << 24);
            w1 = (uint32_t)';' | ((uint32_t)'5' << 8) | ((uint32_t)';' << 16) | ((d & 255u) << 24);
            w2 = (d >> 8) | ((uint32_t)'m' << 16) | (glyph << 24);
        } else {
            // ESC [ 4 8 ; 5 ; \0 1 6 m ' '
            w0 = ESC_BR | ((uint32_t)'4' << 16) | ((uint32_t)'8' << 24);
            w1 = (uint32_t)';' | ((uint32_t)'5' << 8) | ((uint32_t)';' << 16);
            w2 = (uint32_t)'1' | ((uint32_t)'6' << 8) | ((uint32_t)'m' << 16) | ((uint32_t)' ' << 24);
        }
        dst[0] = w0;
        dst[1] = w1;
        dst[2] = w2;
    }
}

template <int MODE, bool CULL>
__global__ __launch_bounds__(kThreads) void rtx_trace(const KArgs a)
{
    __shared__ float4 s_rec[kListCap];
    __shared__ uint32_t s_idx[kListCap];
    __shared__ uint32_t s_wcnt[2][4]; // survivors per wave of the current chunk, double-buffered

    const uint32_t tid = threadIdx.x;
    const uint32_t lw = a.tile_log2w;
    const uint32_t tw = 1u << lw, th = (uint32_t)kThreads >> lw;
    const uint32_t col0 = blockIdx.x * tw;
    const uint32_t row0 = a.row0 + blockIdx.y * th;
    const uint32_t col = col0 + (tid & (tw - 1u));
    const uint32_t row = row0 + (tid >> lw);

    Camera cam;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        cam.m[i] = a.m[i];
    }
    cam.ox = a.ox; cam.oy = a.oy; cam.oz = a.oz;
    cam.e1 = a.e1; cam.e2 = a.e2; cam.far = a.far;
    cam.fW = a.fW; cam.fH = a.fH;

    const bool in_frame = col < a.W && row < a.row_end;
    const bool newline_col = col + 1u == a.W;
    // lanes outside the frame trace a clamped pixel so that every lane runs the same loops
    const Ray ray = make_ray(cam, row < a.row_end ? row : a.row_end - 1u, col < a.W ? col : a.W - 1u);

    TileFrustum fr;
    if (CULL) {
        fr = tile_frustum(cam, col0, row0, tw, th);
    }

    Best best;
    best.t = kNoHit;
    best.k = 0xffffffffu;

    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    uint32_t total = 0; // candidates in the list; identical in every thread
    uint32_t parity = 0;
    for (uint32_t base = 0; base < a.ns; base += kThreads, parity ^= 1u) {
        // ---- stage one chunk: hoist, cull, append in index order
        const uint32_t k = base + tid;
        bool keep = false;
        float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < a.ns) {
            const float4 g = a.sph_geom[k]; // cx cy cz r
            // objectToCam = origin - spherePos; c = Dot(otc,otc) - r*r   (Sphere.cu:34-37)
            const float ox = cam.ox - g.x, oy = cam.oy - g.y, oz = cam.oz - g.z;
            const float oo = ox * ox + oy * oy + oz * oz;
            const float cc = oo - (g.w * g.w);
            rec = make_float4(ox, oy, oz, cc);
            keep = true;
            if (CULL) {
                // cc <= 0: the camera is inside or on the sphere; keep (the exact test decides)
                if (cc > 0.0f && tile_culls(fr, ox, oy, oz, oo, g.w)) {
                    keep = false;
                }
            }
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) {
            s_wcnt[parity][wave] = (uint32_t)__popcll(m);
        }
        __syncthreads(); // also orders the previous flush's list reads before the writes below
        const uint32_t c0 = s_wcnt[parity][0], c1 = s_wcnt[parity][1], c2 = s_wcnt[parity][2], c3 = s_wcnt[parity][3];
        if (keep) {
            const uint32_t before = (wave > 0 ? c0 : 0u) + (wave > 1 ? c1 : 0u) + (wave > 2 ? c2 : 0u);
            const uint32_t pos = total + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            s_rec[pos] = rec;
            s_idx[pos] = k;
        }
        total = __builtin_amdgcn_readfirstlane(total + c0 + c1 + c2 + c3);
        const bool last = base + kThreads >= a.ns;
        if (total > (uint32_t)(kListCap - kThreads) || last) {
            __syncthreads();
            // ---- exact tests over the candidate list (index is wave-uniform: LDS broadcast)
            for (uint32_t i = 0; i < total; i++) {
                const float4 sr = s_rec[i];
                float s;
                if (!sphere_reject(ray, sr.x, sr.y, sr.z, sr.w, s)) {
                    float t;
                    if (sphere_hit(ray, s, sr.w, t)) {
                        const uint32_t ki = s_idx[i];
                        if (t < best.t || (t == best.t && ki < best.k)) {
                            best.t = t;
                            best.k = ki;
                        }
                    }
                }
            }
            total = 0;
        }
    }

    // ---- winner among spheres: creation index for the tie-break against planes
    uint32_t best_gidx = 0xffffffffu;
    float4 wgeom = make_float4(0.f, 0.f, 0.f, 0.f), wcol = make_float4(0.f, 0.f, 0.f, 0.f);
    if (best.k != 0xffffffffu) {
        wgeom = a.sph_geom[best.k];
        wcol = a.sph_color[best.k];
        best_gidx = __float_as_uint(wcol.w);
    }

    // ---- planes (few; wave-uniform index -> scalar loads)
    bool plane_won = false;
    V3 plane_n = v3(0.f, 0.f, 0.f), plane_col = v3(0.f, 0.f, 0.f);
    for (uint32_t j = 0; j < a.np; j++) {
        const float4 pa = a.pl_a[j]; // px py pz width
        const float4 pb = a.pl_b[j]; // nx ny nz height
        float t;
        if (plane_hit(ray, v3(pa.x, pa.y, pa.z), v3(pb.x, pb.y, pb.z), pa.w, pb.w, t)) {
            const float4 pc = a.pl_c[j]; // R G B gidx
            const uint32_t gi = __float_as_uint(pc.w);
            if (t < best.t || (t == best.t && gi < best_gidx)) {
                best.t = t;
                best_gidx = gi;
                plane_won = true;
                plane_n = v3(pb.x, pb.y, pb.z);
                plane_col = v3(pc.x, pc.y, pc.z);
            }
        }
    }

    // ---- shade the winner (RayTracing.cu:123-157)
    float distance = kNoHit, shadingValue = 0.0f;
    V3 normal = v3(0.f, 0.f, 0.f), colour = v3(0.f, 0.f, 0.f);
    if (plane_won || best.k != 0xffffffffu) {
        V3 n0, objc;
        if (plane_won) {
            n0 = plane_n;
            objc = plane_col;
        } else {
            // Sphere.cu:67: (origin + direction * t1 - spherePos).Normalize_GPU()
            n0 = normalize_gpu(sub(add(ray.o, mulf(ray.d, best.t)), v3(wgeom.x, wgeom.y, wgeom.z)));
            objc = v3(wcol.x, wcol.y, wcol.z);
        }
        distance = best.t;
        normal = normalize_gpu(n0);                                     // RayTracing.cu:129
        shadingValue = normal.x * 1.0f + normal.y * 0.0f + normal.z * 0.0f; // Dot(normal, (1,0,0)), :133
        if (MODE != RTX_K_RGB_NORMALS) {
            colour = shade(ray, distance, normal, objc);
        }
    }

    if (MODE != RTX_K_SDL) {
        encode_and_store<MODE>(a, cam, ray, in_frame, newline_col, row, col, distance, normal, colour, shadingValue);
    }
}

// Zero-fills [begin, end) of the frame (bytes, 4-aligned): the 8-bit modes' unused tail.
__global__ __launch_bounds__(kThreads) void rtx_zero_fill(uint32_t* p, size_t n_words)
{
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n_words; i += (size_t)gridDim.x * kThreads) {
        p[i] = 0u;
    }
}

} // namespace rtx

extern "C" const char* rtx_k_launch_trace(const KArgs* a, int mode, int cull, void* stream_v, int* hip_error)
{
    using namespace rtx;
    hipStream_t stream = (hipStream_t)stream_v;
    const uint32_t lw = a->tile_log2w;
    const uint32_t tw = 1u << lw, th = (uint32_t)kThreads >> lw;
    const uint32_t rows = a->row_end - a->row0;
    dim3 grid((a->W + tw - 1u) / tw, (rows + th - 1u) / th, 1), block(kThreads, 1, 1);
    const char* name = nullptr;
#define RTX_LAUNCH(M, C)                                                           \
    do {                                                                           \
        hipLaunchKernelGGL((rtx_trace<M, C>), grid, block, 0, stream, *a);         \
        name = "rtx_trace<" #M "," #C ">";                                         \
    } while (0)
#define RTX_LAUNCH_MODE(M)      \
    do {                        \
        if (cull) {             \
            RTX_LAUNCH(M, true);  \
        } else {                \
            RTX_LAUNCH(M, false); \
        }                       \
    } while (0)
    switch (mode) {
    case RTX_K_BIT_ASCII: RTX_LAUNCH_MODE(RTX_K_BIT_ASCII); break;
    case RTX_K_BIT_PIXEL: RTX_LAUNCH_MODE(RTX_K_BIT_PIXEL); break;
    case RTX_K_RGB_ASCII: RTX_LAUNCH_MODE(RTX_K_RGB_ASCII); break;
    case RTX_K_RGB_PIXEL: RTX_LAUNCH_MODE(RTX_K_RGB_PIXEL); break;
    case RTX_K_RGB_NORMALS: RTX_LAUNCH_MODE(RTX_K_RGB_NORMALS); break;
    case RTX_K_SDL: RTX_LAUNCH_MODE(RTX_K_SDL); break;
    default: *hip_error = 0; return nullptr;
    }
#undef RTX_LAUNCH_MODE
#undef RTX_LAUNCH
    *hip_error = (int)hipGetLastError();
    return name;
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
