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

namespace rtx {

constexpr int kThreads = 256;
constexpr int kItems = 2;                      // slots per thread
constexpr int kSlotsPerBlock = kThreads * kItems;

// ---------------------------------------------------------------- UpdateObjects
// Sphere::Update, Sphere.cu:15-23 (long double is double in device code); Plane::Update is a no-op
// (Plane.cu:14-18).  One thread per sphere with a launch shape that is valid for any count: the
// reference's block of `count` threads stops launching past 1024 objects.
__global__ __launch_bounds__(kThreads) void rtx_update_spheres(float4* geom, float4* motion, uint32_t ns, double dt, float4* sorted_geom,
                                                               const uint32_t* pos_of)
{
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= ns) {
        return;
    }
    float4 g = geom[i];
    float4 mv = motion[i];
    int mover = (int)__float_as_uint(mv.x);
    const float speed = mv.y;
    // m_center.y += speed * mover * dt;
    g.y = (float)((double)g.y + (double)(speed * (float)mover) * dt);
    if (g.y < -10.0f || g.y > 10.0f) {
        const float r = g.y < -10.0f ? -10.0f : g.y; // MyMath::Clamp, MyMath.cu:29-34
        g.y = r > 10.0f ? 10.0f : r;
        mover *= -1;
    }
    mv.x = __uint_as_float((uint32_t)mover);
    geom[i] = g;
    motion[i] = mv;
    if (sorted_geom != nullptr) {
        sorted_geom[pos_of[i]] = g; // the direction-sorted copy staging reads (rtx_sort_scene) moves with it
    }
}

// sorted[p] = geom[order[p]]: the direction-sorted copy of the sphere array, from the live one.
__global__ __launch_bounds__(kThreads) void rtx_gather_spheres(const float4* geom, const float4* od, const uint32_t* order, float4* sorted_geom,
                                                               float4* sorted_od, uint32_t ns)
{
    const uint32_t p = blockIdx.x * kThreads + threadIdx.x;
    if (p < ns) {
        sorted_geom[p] = geom[order[p]];
        sorted_od[p] = od[order[p]];
    }
}

// ---------------------------------------------------------------- ansi256_from_rgb over a range of inputs
// The mapper of the 8-bit trace kernels (rtx_device.hpp, ANSIRGB.h:141-189) applied to packed 0xRRGGBB values
// first .. first+count-1, four per thread, one dword store each.
__global__ __launch_bounds__(kThreads) void rtx_ansi_map(uint32_t first, uint64_t count, const uint8_t* grey, uint8_t* out)
{
    const uint64_t i0 = ((uint64_t)blockIdx.x * kThreads + threadIdx.x) * 4u;
    if (i0 >= count) {
        return;
    }
    uint32_t v[4];
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t rgb = first + (uint32_t)i0 + k;
        v[k] = ansi256_from_rgb((rgb >> 16) & 255u, (rgb >> 8) & 255u, rgb & 255u, grey);
    }
    if (i0 + 4u <= count && (((uintptr_t)(out + i0)) & 3u) == 0u) {
        *reinterpret_cast<uint32_t*>(out + i0) = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
    } else {
        for (uint32_t k = 0; k < 4u && i0 + k < count; k++) {
            out[i0 + k] = (uint8_t)v[k];
        }
    }
}

// ---------------------------------------------------------------- Minimize
//
// The reference scans the frame byte by byte on one CPU thread.  Restated per slot (a slot is one
// S-byte record position; W slots per row, the last one being the row's NUL column):
//   * NUL-column slot           -> emits '\n'                        (RayTracingManager.cu:223-239)
//   * slot starting with ESC    -> emits the whole record if its colour digits differ from the
//                                  colour of the previous ESC slot in scan order (rows included),
//                                  else only its last byte (the glyph)   (:193-220)
//   * any other slot (all NUL)  -> emits nothing                     (:241-245)
// "latestColor" only moves when the colour differs, so comparing with the previous ESC slot is the
// same test.  Output offsets are an exclusive prefix sum of the emitted lengths.
template <int S>
struct Slot {
    uint32_t w[S / 4];
};

template <int S>
__device__ __forceinline__ bool same_colour(const Slot<S>& a, const Slot<S>& b)
{
    if (S == 12) {
        // bytes 7, 8, 9
        return ((a.w[1] ^ b.w[1]) & 0xff000000u) == 0u && ((a.w[2] ^ b.w[2]) & 0x0000ffffu) == 0u;
    }
    // bytes 7-9, 11-13, 15-17
    return ((a.w[1] ^ b.w[1]) & 0xff000000u) == 0u && ((a.w[2] ^ b.w[2]) & 0xff00ffffu) == 0u &&
           ((a.w[3] ^ b.w[3]) & 0xff00ffffu) == 0u && ((a.w[4] ^ b.w[4]) & 0x0000ffffu) == 0u;
}

template <int S>
__device__ __forceinline__ Slot<S> load_slot(const uint8_t* in, uint64_t i)
{
    const uint32_t* p = reinterpret_cast<const uint32_t*>(in + i * S);
    Slot<S> s;
#pragma unroll
    for (int k = 0; k < S / 4; k++) {
        s.w[k] = p[k];
    }
    return s;
}

// Emitted length of slot i; `rec` receives the record when the slot is a pixel.
template <int S>
__device__ __forceinline__ uint32_t slot_length(const uint8_t* in, uint64_t i, uint32_t W, Slot<S>& rec)
{
    const uint32_t col = (uint32_t)(i % W);
    if (col == W - 1u) {
        return 1u; // newline
    }
    rec = load_slot<S>(in, i);
    if ((rec.w[0] & 0xffu) != 0x1bu) {
        return 0u;
    }
    // previous ESC slot in scan order: for a rendered frame this is slot i-1, or i-2 across a row end
    uint64_t j = i;
    while (j > 0) {
        --j;
        if ((uint32_t)(j % W) == W - 1u) {
            continue;
        }
        if (in[j * S] == 0x1bu) {
            const Slot<S> prev = load_slot<S>(in, j);
            return same_colour<S>(rec, prev) ? 1u : (uint32_t)S;
        }
    }
    return (uint32_t)S; // first pixel of the frame
}

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_wave, uint32_t& block_total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= (uint32_t)d) {
            incl += o;
        }
    }
    if (lane == 63u) {
        s_wave[wave] = incl;
    }
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kThreads / 64; k++) {
        const uint32_t t = s_wave[k];
        base += (uint32_t)k < wave ? t : 0u;
        total += t;
    }
    __syncthreads();
    block_total = total;
    return base + incl - v;
}

// ---- staged form.  A block owns kSlotsPerBlock consecutive slots.  Its input bytes (plus a halo of the two
// slots before it, which is where the previous ESC slot of a rendered frame always is) are brought into
// LDS with coalesced 16-byte loads; slots are then read from LDS at their 12/20-byte stride (5 or 3 dwords:
// odd strides are bank-conflict free).  The scatter kernel also builds its output bytes in LDS and writes
// them out with aligned 16-byte stores, so that HBM sees only full-width traffic.
constexpr int kHaloOff = 48; // LDS byte offset of the block's first slot: 16-byte aligned, room for 2 x 20 halo bytes

template <int S>
__device__ __forceinline__ void stage_input(const uint8_t* __restrict__ in, uint64_t first_slot, uint64_t n_slots, uint8_t* s_in)
{
    const uint64_t b0 = first_slot * S;
    const uint64_t total_bytes = n_slots * S;
    const uint64_t b1 = b0 + (uint64_t)kSlotsPerBlock * S < total_bytes ? b0 + (uint64_t)kSlotsPerBlock * S : total_bytes;
    const uint32_t nbytes = (uint32_t)(b1 - b0);           // multiple of 4
    const uint32_t n16 = nbytes / 16u;
    const uint4* src = reinterpret_cast<const uint4*>(in + b0);
    uint4* dst = reinterpret_cast<uint4*>(s_in + kHaloOff);
    for (uint32_t i = threadIdx.x; i < n16; i += kThreads) {
        dst[i] = src[i];
    }
    // tail dwords (when the frame ends inside this block) and the halo
    const uint32_t* src32 = reinterpret_cast<const uint32_t*>(in + b0);
    uint32_t* dst32 = reinterpret_cast<uint32_t*>(s_in + kHaloOff);
    for (uint32_t i = n16 * 4u + threadIdx.x; i < nbytes / 4u; i += kThreads) {
        dst32[i] = src32[i];
    }
    if (threadIdx.x < 2u * S / 4u) {
        const uint32_t hd = threadIdx.x; // dword of the halo, counted from its start
        uint32_t v = 0u;
        if (b0 >= 2u * S) {
            v = reinterpret_cast<const uint32_t*>(in + b0 - 2u * S)[hd];
        } else if (b0 >= S && hd >= S / 4u) {
            v = reinterpret_cast<const uint32_t*>(in + b0 - S)[hd - S / 4u];
        }
        reinterpret_cast<uint32_t*>(s_in + kHaloOff - 2 * S)[hd] = v;
    }
}

template <int S>
__device__ __forceinline__ Slot<S> lds_slot(const uint8_t* s_in, int li)
{
    const uint32_t* p = reinterpret_cast<const uint32_t*>(s_in + kHaloOff + li * S);
    Slot<S> r;
#pragma unroll
    for (int k = 0; k < S / 4; k++) {
        r.w[k] = p[k];
    }
    return r;
}

// Emitted length of global slot g = first_slot + li, whose bytes (and halo) are staged in LDS.
template <int S>
__device__ __forceinline__ uint32_t staged_length(const uint8_t* in, const uint8_t* s_in, uint64_t g, int li, uint32_t col, uint32_t W, Slot<S>& rec)
{
    if (col == W - 1u) {
        return 1u; // newline
    }
    rec = lds_slot<S>(s_in, li);
    if ((rec.w[0] & 0xffu) != 0x1bu) {
        return 0u;
    }
    if (g == 0) {
        return (uint32_t)S; // first pixel of the frame
    }
    // previous ESC slot: slot g-1, or g-2 when g-1 is the previous row's NUL column
    const int back = col == 0u ? 2 : 1;
    if (g >= (uint64_t)back) {
        const Slot<S> prev = lds_slot<S>(s_in, li - back);
        if ((prev.w[0] & 0xffu) == 0x1bu) {
            return same_colour<S>(rec, prev) ? 1u : (uint32_t)S;
        }
    }
    // not a fully rendered frame (empty slots in between): walk back through global memory
    uint64_t j = g;
    while (j > 0) {
        --j;
        if ((uint32_t)(j % W) == W - 1u) {
            continue;
        }
        if (in[j * S] == 0x1bu) {
            const Slot<S> prev = load_slot<S>(in, j);
            return same_colour<S>(rec, prev) ? 1u : (uint32_t)S;
        }
    }
    return (uint32_t)S;
}

template <int S>
__global__ __launch_bounds__(kThreads) void rtx_min_count(const uint8_t* in, uint64_t n_slots, uint32_t W, uint32_t* block_sums)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_in[kHaloOff + kSlotsPerBlock * S];
    __shared__ uint32_t s_wave[kThreads / 64];
    const uint64_t base = (uint64_t)blockIdx.x * kSlotsPerBlock;
    stage_input<S>(in, base, n_slots, s_in);
    const uint32_t col0 = (uint32_t)(base % W); // one 64-bit division per thread; columns below are 32-bit
    __syncthreads();
    uint32_t sum = 0;
#pragma unroll
    for (int it = 0; it < kItems; it++) {
        const int li = it * kThreads + (int)threadIdx.x;
        const uint64_t g = base + (uint64_t)li;
        if (g < n_slots) {
            Slot<S> rec;
            sum += staged_length<S>(in, s_in, g, li, (col0 + (uint32_t)li) % W, W, rec);
        }
    }
    uint32_t total;
    block_exclusive_scan(sum, s_wave, total);
    if (threadIdx.x == 0) {
        block_sums[blockIdx.x] = total;
    }
}

template <int S>
__global__ __launch_bounds__(kThreads) void rtx_min_scatter(const uint8_t* in, uint64_t n_slots, uint32_t W, const uint32_t* block_sums, uint8_t* out, uint64_t* total_out)
{
    __shared__ uint64_t s_part[kThreads / 64];
    __shared__ __attribute__((aligned(16))) uint8_t s_in[kHaloOff + kSlotsPerBlock * S];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[16 + kSlotsPerBlock * S];
    __shared__ uint32_t s_wave[kThreads / 64];
    const uint64_t base = (uint64_t)blockIdx.x * kSlotsPerBlock;
    stage_input<S>(in, base, n_slots, s_in);
    // where this block's output starts: the sum of the lengths of the blocks before it (a few KB of L2 reads
    // per block; cheaper than a separate scan launch)
    uint64_t part = 0;
    for (uint32_t i = threadIdx.x; i < blockIdx.x; i += kThreads) {
        part += block_sums[i];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        part += __shfl_xor(part, d);
    }
    if ((threadIdx.x & 63u) == 0u) {
        s_part[threadIdx.x >> 6] = part;
    }
    __syncthreads();
    uint64_t G = 0;
#pragma unroll
    for (int k = 0; k < kThreads / 64; k++) {
        G += s_part[k];
    }
    const uint32_t pad = (uint32_t)(G & 15u);     // the LDS image is laid out with the same 16-byte phase
    const uint32_t col0 = (uint32_t)(base % W);
    __syncthreads();

    uint32_t carry = 0;
#pragma unroll 1
    for (int it = 0; it < kItems; it++) {
        const int li = it * kThreads + (int)threadIdx.x;
        const uint64_t g = base + (uint64_t)li;
        Slot<S> rec;
        uint32_t len = 0;
        const uint32_t col = (col0 + (uint32_t)li) % W;
        if (g < n_slots) {
            len = staged_length<S>(in, s_in, g, li, col, W, rec);
        }
        uint32_t total;
        const uint32_t excl = block_exclusive_scan(len, s_wave, total);
        uint8_t* dst = s_out + pad + carry + excl;
        if (len == (uint32_t)S) {
#pragma unroll
            for (int k = 0; k < S / 4; k++) {
                const uint32_t w = rec.w[k];
                dst[4 * k + 0] = (uint8_t)(w);
                dst[4 * k + 1] = (uint8_t)(w >> 8);
                dst[4 * k + 2] = (uint8_t)(w >> 16);
                dst[4 * k + 3] = (uint8_t)(w >> 24);
            }
        } else if (len == 1u) {
            const bool newline = col == W - 1u;
            dst[0] = newline ? (uint8_t)'\n' : (uint8_t)(rec.w[S / 4 - 1] >> 24);
        }
        carry += total;
    }
    __syncthreads();

    // copy out: bytes [G, G + carry).  Head up to the first 16-byte boundary and tail after the last one go
    // out byte by byte (neighbouring blocks own the other bytes of those 16-byte lines); the body as uint4.
    const uint32_t n = carry;
    const uint32_t head = n < ((16u - pad) & 15u) ? n : ((16u - pad) & 15u);
    const uint32_t body16 = (n - head) / 16u;
    const uint32_t tail = n - head - body16 * 16u;
    if (threadIdx.x < head) {
        out[G + threadIdx.x] = s_out[pad + threadIdx.x];
    }
    const uint4* src = reinterpret_cast<const uint4*>(s_out + pad + head); // pad + head is 0 mod 16
    uint4* dst16 = reinterpret_cast<uint4*>(out + G + head);
    for (uint32_t i = threadIdx.x; i < body16; i += kThreads) {
        dst16[i] = src[i];
    }
    if (threadIdx.x < tail) {
        out[G + head + body16 * 16u + threadIdx.x] = s_out[pad + head + body16 * 16u + threadIdx.x];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        *total_out = G + n; // length of the minimised stream
    }
}

// ---------------------------------------------------------------- Minimize from compact pixel words
//
// The same pass over the 4-byte pixel words the trace kernels can store instead of records (RTX_RENDER_COMPACT, rtx.h): a word
// carries everything its record is made of, so the pass reads 4 bytes per pixel where the record form reads 12 or 20 -- twice.
// rtx_update traces words and minimises from them: 8.3 MB written and 2 x 8.3 MB read per 1080p frame instead of 41.5 MB and
// 2 x 41.5 MB (the full-size records never exist), and a device group gathers the words it minimises from (no expansion).
// Slot rules as above, stated on words:
//   * column W-1                          -> '\n'
//   * word 0xffffffff elsewhere           -> an empty slot (all NUL as a record): emits nothing, is not an ESC slot
//   * any other word (0 = miss)           -> an ESC slot: its record is record_words<MODE>(word); the colour Minimize compares
//     (bytes 7-9 [, 11-13, 15-17] of the record: the decimal digits) is a function of the word's colour bytes alone -- the
//     digits of (r, g, b) or of the xterm index; a miss carries the digits of (0, 0, 0) resp. of index 16 (App. B) -- so
//     "same colour digits" is "same colour key".
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
constexpr int kWPerThread = 4;
constexpr int kWSlotsPerBlock = kThreads * kWPerThread; // 1024 slots = 4 KB of words per block
constexpr uint32_t kNoWord = 0xffffffffu;               // = kCompactNewline

template <int MODE>
__device__ __forceinline__ uint32_t colour_key(uint32_t w)
{
    constexpr bool kRgb = (MODE == RTX_K_RGB_ASCII || MODE == RTX_K_RGB_PIXEL || MODE == RTX_K_RGB_NORMALS);
    return kRgb ? (w & 0x00ffffffu) : (w != 0u ? (w & 0xffu) : 16u);
}

// s_w[0..1] = the two words before the block's first slot (kNoWord where the frame begins), s_w[2 + li] = word of slot base + li
// `lead`: how many words BEFORE words[0] exist and belong to the same frame (0, or the W words of the row above a slab: a rank of
// a device group minimises its own rows and needs the last pixel of the row before them; rtx_update on a group, RTX_OPT_GROUP_UPDATE)
__device__ __forceinline__ void stage_words(const uint32_t* __restrict__ words, uint64_t base, uint64_t n_slots, uint32_t* s_w, uint32_t lead)
{
    const uint32_t tid = threadIdx.x;
    const uint64_t g0 = base + (uint64_t)tid * kWPerThread;
    if (g0 + kWPerThread <= n_slots && ((uintptr_t)(words + g0) & 15u) == 0u) {
        const uint4 v = *reinterpret_cast<const uint4*>(words + g0);
        s_w[2 + tid * kWPerThread + 0] = v.x;
        s_w[2 + tid * kWPerThread + 1] = v.y;
        s_w[2 + tid * kWPerThread + 2] = v.z;
        s_w[2 + tid * kWPerThread + 3] = v.w;
    } else {
#pragma unroll
        for (int k = 0; k < kWPerThread; k++) {
            s_w[2 + tid * kWPerThread + k] = g0 + k < n_slots ? words[g0 + k] : kNoWord;
        }
    }
    if (tid < 2u) {
        s_w[tid] = base + tid + lead >= 2u ? words[(int64_t)(base + tid) - 2] : kNoWord;
    }
}

// Column of this thread's first slot: the block's first column by one scalar 64-bit division (the block index is uniform), the
// thread's by a 32-bit one (a 64-bit division per thread is ~100 instructions, a third of what this pass executes).
__device__ __forceinline__ uint32_t first_column(uint64_t base, uint32_t W)
{
    // (a 64-bit remainder is ~130 instructions, a tenth of what a wave of the one-launch form executes: 32-bit where the slot fits)
    const uint32_t col0 = (uint32_t)__builtin_amdgcn_readfirstlane((base >> 32) == 0u ? (uint32_t)base % W : (uint32_t)(base % W));
    return (col0 + threadIdx.x * (uint32_t)kWPerThread) % W; // (col0 < W < 2^31 and the offset < 1024: no overflow)
}

// The previous ESC slot of slot g is not among the two staged before it (a frame that was only partly rendered: empty slots in
// between): walk back through global memory.  Rare, and kept out of line so that the passes' straight-line code stays short.
template <int MODE>
__device__ __noinline__ uint32_t word_length_walk(const uint32_t* __restrict__ words, uint64_t g, uint32_t W, uint32_t w, uint32_t lead)
{
    constexpr uint32_t S = (MODE == RTX_K_RGB_ASCII || MODE == RTX_K_RGB_PIXEL || MODE == RTX_K_RGB_NORMALS) ? 20u : 12u;
    int64_t j = (int64_t)g;
    while (j > -(int64_t)lead) {
        --j;
        if ((uint32_t)((uint64_t)(j + (int64_t)lead) % W) == W - 1u) { // (lead is a multiple of W)
            continue;
        }
        const uint32_t pw = words[j];
        if (pw != kNoWord) {
            return colour_key<MODE>(pw) == colour_key<MODE>(w) ? 1u : S;
        }
    }
    return S;
}

// Emitted length of slot g (column col, staged at s_w[2 + li]); `w` receives its word.  Selects, and one branch for the rare walk.
template <int MODE>
__device__ __forceinline__ uint32_t word_length(const uint32_t* __restrict__ words, const uint32_t* s_w, uint64_t g, int li, uint32_t col, uint32_t W, uint32_t& w,
                                                uint32_t lead)
{
    constexpr uint32_t S = (MODE == RTX_K_RGB_ASCII || MODE == RTX_K_RGB_PIXEL || MODE == RTX_K_RGB_NORMALS) ? 20u : 12u;
    w = s_w[2 + li];
    // previous ESC slot: slot g-1, or g-2 when g-1 is the previous row's last column (both staged: s_w[0..1] precede the block)
    const int back = col == 0u ? 2 : 1;
    const uint32_t pw = s_w[2 + li - back];
    const bool newline = col == W - 1u, empty = w == kNoWord, first = g == 0 && lead == 0u;
    const bool staged = g + lead >= (uint64_t)back && pw != kNoWord;
    uint32_t len = colour_key<MODE>(pw) == colour_key<MODE>(w) ? 1u : S;
    len = first ? S : len;     // first pixel of the frame
    len = empty ? 0u : len;    // empty slot
    len = newline ? 1u : len;
    if (!newline && !empty && !first && !staged) {
        len = word_length_walk<MODE>(words, g, W, w, lead);
    }
    return len;
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void rtx_minw_count(const uint32_t* __restrict__ words, uint64_t n_slots, uint32_t W, uint32_t* block_sums, uint32_t lead)
{
    __shared__ uint32_t s_w[2 + kWSlotsPerBlock];
    __shared__ uint32_t s_wave[kThreads / 64];
    const uint64_t base = (uint64_t)blockIdx.x * kWSlotsPerBlock;
    stage_words(words, base, n_slots, s_w, lead);
    uint32_t col = first_column(base, W);
    __syncthreads();
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < kWPerThread; k++) {
        const int li = (int)threadIdx.x * kWPerThread + k;
        const uint64_t g = base + (uint64_t)li;
        uint32_t w;
        if (g < n_slots) {
            sum += word_length<MODE>(words, s_w, g, li, col, W, w, lead);
        }
        col = col + 1u == W ? 0u : col + 1u;
    }
    uint32_t total;
    block_exclusive_scan(sum, s_wave, total);
    if (threadIdx.x == 0) {
        block_sums[blockIdx.x] = total;
    }
}

// The blocks' sums -> their offsets in the stream (exclusive scan; one workgroup, 2048 sums per step) and the stream's length.
// A launch of its own between the two passes: the scatter pass then finds its place with one load, where summing the preceding
// blocks' sums in every block cost a chain of dependent L2 round trips per block and O(blocks^2) loads per frame (32 400 blocks
// at 8K).  (Letting the count pass's last block do this -- a ticket drawn with atomicAdd -- was measured: 2025 device-scope
// atomics on one address take 46 us, 22 ns each.)
__global__ __launch_bounds__(kThreads) void rtx_min_offsets(const uint32_t* __restrict__ block_sums, uint32_t nb, uint64_t* __restrict__ offsets, uint64_t* total_out)
{
    __shared__ uint32_t s_wave[kThreads / 64];
    uint64_t carry = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += 8u * kThreads) {
        uint32_t v[8], mine = 0;
#pragma unroll
        for (uint32_t q = 0; q < 8u; q++) {
            const uint32_t i = b0 + threadIdx.x * 8u + q;
            v[q] = i < nb ? block_sums[i] : 0u;
            mine += v[q];
        }
        uint32_t step_total;
        uint64_t at = carry + block_exclusive_scan(mine, s_wave, step_total);
#pragma unroll
        for (uint32_t q = 0; q < 8u; q++) {
            const uint32_t i = b0 + threadIdx.x * 8u + q;
            if (i < nb) offsets[i] = at;
            at += v[q];
        }
        carry += step_total;
    }
    if (threadIdx.x == 0) {
        *total_out = carry; // length of the minimised stream
    }
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void rtx_minw_scatter(const uint32_t* __restrict__ words, uint64_t n_slots, uint32_t W, const uint64_t* __restrict__ offsets,
                                                             uint8_t* out, uint32_t lead)
{
    constexpr bool kRgb = (MODE == RTX_K_RGB_ASCII || MODE == RTX_K_RGB_PIXEL || MODE == RTX_K_RGB_NORMALS);
    constexpr uint32_t S = kRgb ? 20u : 12u;
    __shared__ uint32_t s_w[2 + kWSlotsPerBlock];
    __shared__ uint32_t s_digits[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[16 + kWSlotsPerBlock * S];
    __shared__ uint32_t s_wave[kThreads / 64];
    const uint64_t base = (uint64_t)blockIdx.x * kWSlotsPerBlock;
    const uint64_t G = offsets[blockIdx.x]; // where this block's output starts (the count pass's last block left it)
    stage_words(words, base, n_slots, s_w, lead);
    s_digits[threadIdx.x] = digits_word(threadIdx.x);
    uint32_t col = first_column(base, W);
    __syncthreads();
    const uint32_t pad = (uint32_t)(G & 15u); // the LDS image is laid out with the same 16-byte phase as its destination

    uint32_t len[kWPerThread], wd[kWPerThread], cols[kWPerThread], mine = 0;
#pragma unroll
    for (int k = 0; k < kWPerThread; k++) {
        const int li = (int)threadIdx.x * kWPerThread + k;
        const uint64_t g = base + (uint64_t)li;
        len[k] = 0u;
        wd[k] = kNoWord;
        cols[k] = col;
        if (g < n_slots) {
            len[k] = word_length<MODE>(words, s_w, g, li, col, W, wd[k], lead);
        }
        mine += len[k];
        col = col + 1u == W ? 0u : col + 1u;
    }
    uint32_t n;
    uint32_t at = pad + block_exclusive_scan(mine, s_wave, n);
#pragma unroll
    for (int k = 0; k < kWPerThread; k++) {
        uint8_t* dst = s_out + at;
        if (len[k] == S) {
            Fields f;
            f.c0 = wd[k] & 255u;
            f.c1 = (wd[k] >> 8) & 255u;
            f.c2 = (wd[k] >> 16) & 255u;
            f.glyph = wd[k] >> 24;
            uint32_t r[S / 4];
            record_words<MODE>(wd[k] != kCompactMiss, f, s_digits, r);
            // the record lands at an arbitrary byte offset of the LDS image: dword stores without an alignment promise (gfx950
            // takes unaligned LDS accesses: one ds_write_b32 each, where byte stores were twenty)
#pragma unroll
            for (uint32_t q = 0; q < S / 4u; q++) {
                *reinterpret_cast<u32_unaligned*>(dst + 4u * q) = r[q];
            }
        } else if (len[k] == 1u) {
            // the row's newline, or the glyph alone (the last byte of the record: ' ' for a miss)
            dst[0] = cols[k] == W - 1u ? (uint8_t)'\n' : (wd[k] == kCompactMiss ? (uint8_t)' ' : (uint8_t)(wd[k] >> 24));
        }
        at += len[k];
    }
    __syncthreads();

    // copy out: bytes [G, G + n), head and tail of partial 16-byte lines byte by byte, the body as uint4
    const uint32_t head = n < ((16u - pad) & 15u) ? n : ((16u - pad) & 15u);
    const uint32_t body16 = (n - head) / 16u;
    const uint32_t tail = n - head - body16 * 16u;
    if (threadIdx.x < head) {
        out[G + threadIdx.x] = s_out[pad + threadIdx.x];
    }
    const uint4* src = reinterpret_cast<const uint4*>(s_out + pad + head); // pad + head is 0 mod 16
    uint4* dst16 = reinterpret_cast<uint4*>(out + G + head);
    for (uint32_t i = threadIdx.x; i < body16; i += kThreads) {
        dst16[i] = src[i];
    }
    if (threadIdx.x < tail) {
        out[G + head + body16 * 16u + threadIdx.x] = s_out[pad + head + body16 * 16u + threadIdx.x];
    }
}

// ---- the same pass as ONE launch (rtx_minw_fused): count, offsets and scatter of the three launches above in one kernel, the
// block offsets by a two-level look-back.  Three dependent launches of a few microseconds each pay two launch gaps and read the
// words twice; here a block counts its slots, publishes its length, builds its output bytes in LDS while the other blocks do the
// same, and then finds where its bytes go:
//   * agg[b]  = (epoch << 32) | length of block b            published by every block as soon as it has counted
//   * grp[r][g] = (epoch << 32) | length of blocks 64g..64g+63  published by the LAST block of the group, which reads the other 63
//     lengths for its own offset anyway; 64 replicas r (rows of `ng` entries), block b reads replica b % 64
//   * offset of block b = sum of grp[b % 64][0 .. b/64) + sum of agg[64 (b/64) .. b): one wave, one or a few loads per lane, two
//     dependent steps for every block however many blocks there are (no chain of prefixes from block to block).
// A block waits only for blocks with smaller indices; workgroups are dispatched in index order on every XCD, so the unfinished
// block with the smallest index never waits and the launch drains.  All the same nothing here spins without a bound: a lane that
// has polled max_polls times gives up, the block writes the launch's epoch into the failure word and stores nothing, a group's
// last block that gave up publishes a poisoned total (the later blocks then give up at once), and the host runs the three
// launches above instead (launch_minimize_words, settle_minimize_words).  Entries carry the launch's epoch, so the tables are
// never cleared between launches; they are zeroed when allocated and epoch 0 is never used.
constexpr uint32_t kLookGroup = 64u;
constexpr uint32_t kLookPoison = 0xffffffffu;
constexpr uint32_t kLookPolls = 1u << 18; // x >= 0.5 us per poll: at least a tenth of a second

__device__ __forceinline__ bool look_wait(const uint64_t* p, uint32_t epoch, uint32_t max_polls, uint32_t& value)
{
    for (uint32_t i = 0;; i++) {
        const uint64_t e = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(e >> 32) == epoch) {
            value = (uint32_t)e;
            return true;
        }
        if (i >= max_polls) {
            value = 0u;
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// The look-back of block b, whose length n is already published in agg[b]: done by the first wave, the block's offset in *s_G and
// the verdict in *s_ok (both in LDS), a barrier, and the verdict returned to every thread.  The barrier also completes whatever the
// block wrote to LDS before the call (its output image).
__device__ __forceinline__ bool look_back(uint32_t b, uint32_t n, const uint64_t* agg, uint64_t* grp, uint32_t ng, uint32_t epoch, uint32_t max_polls, uint64_t* s_G,
                                          uint32_t* s_ok)
{
    if (threadIdx.x < 64u) {
        const uint32_t lane = threadIdx.x;
        const uint32_t g = b / kLookGroup, first = g * kLookGroup;
        uint64_t before = 0, in_group = 0;
        bool ok_group = true, ok_before = true;
        if (first + lane < b) {
            uint32_t v;
            ok_group = look_wait(&agg[first + lane], epoch, max_polls, v);
            in_group = v;
        }
        if (max_polls == 0u && b % 3u == 1u) {
            ok_group = false; // (tests: a launch whose blocks give up, RTX_OPT_MINIMIZE_FUSED = 2)
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            in_group += __shfl_xor(in_group, d);
        }
        ok_group = __all(ok_group);
        if (b % kLookGroup == kLookGroup - 1u) {
            // the group's total, once per replica (lane r writes replica r: 64 lines) and BEFORE this block looks at the totals
            // of the groups before it: a total depends on its own group only, so all of them appear at about the same time
            // (published after that look, they formed a chain, 0.85 us per group: 30 us at 1080p)
            const uint32_t t = ok_group ? (uint32_t)in_group + n : kLookPoison;
            __hip_atomic_store(&grp[(size_t)lane * ng + g], ((uint64_t)epoch << 32) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // (replica b % 64 of the totals, so that a total's readers are spread over 64 lines)
        const uint64_t* my_grp = grp + (size_t)(b % kLookGroup) * ng;
        for (uint32_t q = lane; q < g && ok_before; q += 64u) {
            uint32_t v;
            ok_before = look_wait(&my_grp[q], epoch, max_polls, v) && v != kLookPoison;
            before += v;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            before += __shfl_xor(before, d);
        }
        ok_before = __all(ok_before);
        if (lane == 0u) {
            *s_G = before + in_group;
            *s_ok = (ok_group && ok_before) ? 1u : 0u;
        }
    }
    __syncthreads();
    return *s_ok != 0u;
}

// Bytes [0, n) of the LDS image `img` (16-byte aligned) to out[G, G + n): the image starts at offset 0 whatever G's phase, so a 16-byte
// line of the destination is five aligned dwords of the image and a byte shift (four unaligned dword reads become an unaligned
// ds_read_b128, which is slow: the one-launch form ran 45 us with them).  Head and tail of partial lines byte by byte.
__device__ __forceinline__ void copy_out_image(const uint8_t* img, uint32_t n, uint64_t G, uint8_t* out)
{
    const uint32_t pad = (uint32_t)(G & 15u);
    const uint32_t head = n < ((16u - pad) & 15u) ? n : ((16u - pad) & 15u);
    const uint32_t body16 = (n - head) / 16u;
    const uint32_t tail = n - head - body16 * 16u;
    if (threadIdx.x < head) {
        out[G + threadIdx.x] = img[threadIdx.x];
    }
    const uint32_t* src32 = reinterpret_cast<const uint32_t*>(img) + (head >> 2);
    const uint32_t shift = head & 3u;
    uint4* dst16 = reinterpret_cast<uint4*>(out + G + head);
    for (uint32_t i = threadIdx.x; i < body16; i += kThreads) {
        // (the fifth dword of the image's last line may lie past the image: read only what exists)
        const uint32_t a0 = src32[4u * i], a1 = src32[4u * i + 1u], a2 = src32[4u * i + 2u], a3 = src32[4u * i + 3u];
        const uint32_t a4 = shift != 0u ? src32[4u * i + 4u] : 0u;
        uint4 v;
        v.x = __builtin_amdgcn_alignbyte(a1, a0, shift);
        v.y = __builtin_amdgcn_alignbyte(a2, a1, shift);
        v.z = __builtin_amdgcn_alignbyte(a3, a2, shift);
        v.w = __builtin_amdgcn_alignbyte(a4, a3, shift);
        dst16[i] = v;
    }
    if (threadIdx.x < tail) {
        out[G + head + body16 * 16u + threadIdx.x] = img[head + body16 * 16u + threadIdx.x];
    }
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void rtx_minw_fused(const uint32_t* __restrict__ words, uint64_t n_slots, uint32_t W, uint64_t* agg, uint64_t* grp, uint32_t ng,
                                                           uint32_t epoch, uint32_t max_polls, uint8_t* out, uint64_t* total_out, uint32_t lead)
{
    constexpr bool kRgb = (MODE == RTX_K_RGB_ASCII || MODE == RTX_K_RGB_PIXEL || MODE == RTX_K_RGB_NORMALS);
    constexpr uint32_t S = kRgb ? 20u : 12u;
    // the words first (2 + 1024 dwords) and the scan's partial sums behind them, then -- once every thread has its lengths -- the
    // output bytes from offset 0, at most 1024 x S of them.  (What bounds this launch is the order its dependency imposes on the whole
    // GPU -- every block reads and counts, then every block waits two memory round trips, then every block writes: per-block
    // time stamps of a 1080p frame show lengths published at 2-4 us, offsets known at 6-8, the last byte written at 12.7 -- not the
    // number of resident blocks or of instructions: a build with 20 480 bytes of LDS, eight blocks per CU and all 2025 blocks in
    // one dispatch round was no faster, nor was halving the instructions of the length pass.  EXPERIMENTS.md R4.5.)
    __shared__ __attribute__((aligned(16))) uint8_t s_buf[kWSlotsPerBlock * S];
    __shared__ uint32_t s_digits[256];
    __shared__ uint64_t s_G;
    __shared__ uint32_t s_ok;
    static_assert(sizeof(s_buf) >= (2 + kWSlotsPerBlock) * 4 + 8 + (kThreads / 64) * 4, "words + partial sums fit the image's space");
    uint32_t* s_w = reinterpret_cast<uint32_t*>(s_buf);
    uint32_t* s_wave = s_w + 2 + kWSlotsPerBlock + 2;
    const uint32_t b = blockIdx.x;
    const uint64_t base = (uint64_t)b * kWSlotsPerBlock;
    stage_words(words, base, n_slots, s_w, lead);
    s_digits[threadIdx.x] = digits_word(threadIdx.x);
    uint32_t col = first_column(base, W);
    __syncthreads();

    uint32_t len[kWPerThread], wd[kWPerThread], cols[kWPerThread], mine = 0;
#pragma unroll
    for (int k = 0; k < kWPerThread; k++) {
        const int li = (int)threadIdx.x * kWPerThread + k;
        const uint64_t g = base + (uint64_t)li;
        len[k] = 0u;
        wd[k] = kNoWord;
        cols[k] = col;
        if (g < n_slots) {
            len[k] = word_length<MODE>(words, s_w, g, li, col, W, wd[k], lead);
        }
        mine += len[k];
        col = col + 1u == W ? 0u : col + 1u;
    }
    uint32_t n;
    uint32_t at = block_exclusive_scan(mine, s_wave, n); // (its barriers: every thread has read its words and the sums; s_buf is free)
    if (threadIdx.x == 0) {
        __hip_atomic_store(&agg[b], ((uint64_t)epoch << 32) | n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // the output bytes, built while the other blocks publish their lengths
#pragma unroll
    for (int k = 0; k < kWPerThread; k++) {
        uint8_t* dst = s_buf + at;
        if (len[k] == S) {
            Fields f;
            f.c0 = wd[k] & 255u;
            f.c1 = (wd[k] >> 8) & 255u;
            f.c2 = (wd[k] >> 16) & 255u;
            f.glyph = wd[k] >> 24;
            uint32_t r[S / 4];
            record_words<MODE>(wd[k] != kCompactMiss, f, s_digits, r);
#pragma unroll
            for (uint32_t q = 0; q < S / 4u; q++) {
                *reinterpret_cast<u32_unaligned*>(dst + 4u * q) = r[q];
            }
        } else if (len[k] == 1u) {
            dst[0] = cols[k] == W - 1u ? (uint8_t)'\n' : (wd[k] == kCompactMiss ? (uint8_t)' ' : (uint8_t)(wd[k] >> 24));
        }
        at += len[k];
    }

    if (!look_back(b, n, agg, grp, ng, epoch, max_polls, &s_G, &s_ok)) {
        if (threadIdx.x == 0) {
            total_out[1] = epoch; // the host runs the three-launch form over the same words
        }
        return;
    }
    copy_out_image(s_buf, n, s_G, out);
    if (b == gridDim.x - 1u && threadIdx.x == 0) {
        total_out[0] = s_G + n; // length of the minimised stream
    }
}

// The record form (rtx_min_count -> rtx_min_scatter above) as one launch, on the same look-back: a block owns 1024 consecutive slots,
// a thread four of them; the records are staged in LDS, a thread takes its four into registers, the block counts and publishes, the
// bytes it keeps go back into the same LDS (from offset 0: every record has been read by then), and the offset comes from the blocks
// before it.  The records are read once (41.5 MB at 1080p RGB) where the two launches read them twice.
constexpr int kRSlotsPerBlock = kThreads * 4;

template <int S>
__device__ __forceinline__ void stage_records(const uint8_t* __restrict__ in, uint64_t first_slot, uint64_t n_slots, uint8_t* s_in)
{
    const uint64_t b0 = first_slot * S;
    const uint64_t total_bytes = n_slots * S;
    const uint64_t b1 = b0 + (uint64_t)kRSlotsPerBlock * S < total_bytes ? b0 + (uint64_t)kRSlotsPerBlock * S : total_bytes;
    const uint32_t nbytes = (uint32_t)(b1 - b0); // multiple of 4; in + b0 is 16-byte aligned (1024 S is a multiple of 16)
    const uint32_t n16 = nbytes / 16u;
    const uint4* src = reinterpret_cast<const uint4*>(in + b0);
    uint4* dst = reinterpret_cast<uint4*>(s_in + kHaloOff);
    for (uint32_t i = threadIdx.x; i < n16; i += kThreads) {
        dst[i] = src[i];
    }
    const uint32_t* src32 = reinterpret_cast<const uint32_t*>(in + b0);
    uint32_t* dst32 = reinterpret_cast<uint32_t*>(s_in + kHaloOff);
    for (uint32_t i = n16 * 4u + threadIdx.x; i < nbytes / 4u; i += kThreads) {
        dst32[i] = src32[i];
    }
    if (threadIdx.x < 2u * S / 4u) {
        const uint32_t hd = threadIdx.x; // dword of the halo (the two slots before the block), counted from its start
        uint32_t v = 0u;
        if (b0 >= 2u * S) {
            v = reinterpret_cast<const uint32_t*>(in + b0 - 2u * S)[hd];
        } else if (b0 >= S && hd >= S / 4u) {
            v = reinterpret_cast<const uint32_t*>(in + b0 - S)[hd - S / 4u];
        }
        reinterpret_cast<uint32_t*>(s_in + kHaloOff - 2 * S)[hd] = v;
    }
}

template <int S>
__global__ __launch_bounds__(kThreads) void rtx_min_fused(const uint8_t* in, uint64_t n_slots, uint32_t W, uint64_t* agg, uint64_t* grp, uint32_t ng, uint32_t epoch,
                                                          uint32_t max_polls, uint8_t* out, uint64_t* total_out)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_in[kHaloOff + kRSlotsPerBlock * S];
    __shared__ uint32_t s_wave[kThreads / 64];
    __shared__ uint64_t s_G;
    __shared__ uint32_t s_ok;
    const uint32_t b = blockIdx.x;
    const uint64_t base = (uint64_t)b * kRSlotsPerBlock;
    stage_records<S>(in, base, n_slots, s_in);
    const uint32_t col0 = (uint32_t)__builtin_amdgcn_readfirstlane((base >> 32) == 0u ? (uint32_t)base % W : (uint32_t)(base % W));
    uint32_t col = (col0 + threadIdx.x * 4u) % W;
    __syncthreads();

    Slot<S> rec[4];
    uint32_t len[4], cols[4], mine = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int li = (int)threadIdx.x * 4 + k;
        const uint64_t g = base + (uint64_t)li;
        len[k] = 0u;
        cols[k] = col;
        if (g < n_slots) {
            len[k] = staged_length<S>(in, s_in, g, li, col, W, rec[k]);
        }
        mine += len[k];
        col = col + 1u == W ? 0u : col + 1u;
    }
    uint32_t n;
    uint32_t at = block_exclusive_scan(mine, s_wave, n); // (its barriers: every thread holds its records; s_in is free)
    if (threadIdx.x == 0) {
        __hip_atomic_store(&agg[b], ((uint64_t)epoch << 32) | n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint8_t* dst = s_in + at;
        if (len[k] == (uint32_t)S) {
#pragma unroll
            for (int q = 0; q < S / 4; q++) {
                *reinterpret_cast<u32_unaligned*>(dst + 4 * q) = rec[k].w[q];
            }
        } else if (len[k] == 1u) {
            dst[0] = cols[k] == W - 1u ? (uint8_t)'\n' : (uint8_t)(rec[k].w[S / 4 - 1] >> 24);
        }
        at += len[k];
    }
    if (!look_back(b, n, agg, grp, ng, epoch, max_polls, &s_G, &s_ok)) {
        if (threadIdx.x == 0) {
            total_out[1] = epoch; // the host runs the two-launch form over the same records
        }
        return;
    }
    copy_out_image(s_in, n, s_G, out);
    if (b == gridDim.x - 1u && threadIdx.x == 0) {
        total_out[0] = s_G + n;
    }
}

} // namespace rtx

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
int launch_minimize_words(rtx_ctx* ctx, void* d_scan, int mode, size_t w, size_t h, const uint32_t* d_words, uint8_t* d_out, uint64_t** d_total, uint32_t lead = 0)
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
    uint64_t* total = (uint64_t*)d_scan;
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
