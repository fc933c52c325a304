// rtx_post.hip -- what RayTracingManager::Update runs around the trace kernel, on the GPU:
// UpdateObjects (RayTracingManager.cu:10-44, 89-107) and Minimize (RayTracingManager.cu:167-319),
// plus rtx_update, the whole of Update in one call.
#include "rtx_ctx.h"

#include <cstring>

namespace rtx {

constexpr int kThreads = 256;
constexpr int kItems = 8;                      // slots per thread
constexpr int kSlotsPerBlock = kThreads * kItems;

// ---------------------------------------------------------------- UpdateObjects
// Sphere::Update, Sphere.cu:15-23 (long double is double in device code); Plane::Update is a no-op
// (Plane.cu:14-18).  One thread per sphere with a launch shape that is valid for any count: the
// reference's block of `count` threads stops launching past 1024 objects.
__global__ __launch_bounds__(kThreads) void rtx_update_spheres(float4* geom, float4* motion, uint32_t ns, double dt)
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

template <int S>
__global__ __launch_bounds__(kThreads) void rtx_min_count(const uint8_t* in, uint64_t n_slots, uint32_t W, uint32_t* block_sums)
{
    __shared__ uint32_t s_wave[kThreads / 64];
    const uint64_t base = (uint64_t)blockIdx.x * kSlotsPerBlock;
    uint32_t sum = 0;
#pragma unroll
    for (int it = 0; it < kItems; it++) {
        const uint64_t i = base + (uint64_t)it * kThreads + threadIdx.x;
        if (i < n_slots) {
            Slot<S> rec;
            sum += slot_length<S>(in, i, W, rec);
        }
    }
    uint32_t total;
    block_exclusive_scan(sum, s_wave, total);
    if (threadIdx.x == 0) {
        block_sums[blockIdx.x] = total;
    }
}

// Exclusive scan of the per-block sums (one workgroup), total length to *total_out.
__global__ __launch_bounds__(1024) void rtx_min_scan_blocks(const uint32_t* block_sums, uint64_t* block_offsets, uint32_t n_blocks, uint64_t* total_out)
{
    __shared__ uint64_t s_wave[16];
    __shared__ uint64_t s_carry;
    if (threadIdx.x == 0) {
        s_carry = 0;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < n_blocks; base += 1024u) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < n_blocks ? (uint64_t)block_sums[i] : 0ull;
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t o = __shfl_up(incl, d);
            if (lane >= (uint32_t)d) {
                incl += o;
            }
        }
        if (lane == 63u) {
            s_wave[wave] = incl;
        }
        __syncthreads();
        uint64_t before = 0, total = 0;
        for (uint32_t k = 0; k < 16u; k++) {
            const uint64_t t = s_wave[k];
            before += k < wave ? t : 0ull;
            total += t;
        }
        const uint64_t carry = s_carry;
        if (i < n_blocks) {
            block_offsets[i] = carry + before + incl - v;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            s_carry = carry + total;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *total_out = s_carry;
    }
}

template <int S>
__global__ __launch_bounds__(kThreads) void rtx_min_scatter(const uint8_t* in, uint64_t n_slots, uint32_t W, const uint64_t* block_offsets, uint8_t* out)
{
    __shared__ uint32_t s_wave[kThreads / 64];
    const uint64_t base = (uint64_t)blockIdx.x * kSlotsPerBlock;
    uint64_t carry = block_offsets[blockIdx.x];
#pragma unroll 1
    for (int it = 0; it < kItems; it++) {
        const uint64_t i = base + (uint64_t)it * kThreads + threadIdx.x;
        Slot<S> rec;
        uint32_t len = 0;
        if (i < n_slots) {
            len = slot_length<S>(in, i, W, rec);
        }
        uint32_t total;
        const uint32_t excl = block_exclusive_scan(len, s_wave, total);
        uint8_t* dst = out + carry + excl;
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
            const bool newline = (uint32_t)(i % W) == W - 1u;
            dst[0] = newline ? (uint8_t)'\n' : (uint8_t)(rec.w[S / 4 - 1] >> 24);
        }
        carry += total;
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

int launch_minimize(rtx_ctx* ctx, void* d_scan, int mode, size_t w, size_t h, const uint8_t* d_in, uint8_t* d_out, uint64_t** d_total)
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
    hipLaunchKernelGGL(rtx::rtx_min_scan_blocks, dim3(1), dim3(1024), 0, st, sums, offsets, (uint32_t)n_blocks, total);
    if (rgb) {
        hipLaunchKernelGGL((rtx::rtx_min_scatter<20>), dim3((unsigned)n_blocks), dim3(rtx::kThreads), 0, st, d_in, n_slots, (uint32_t)w, offsets, d_out);
    } else {
        hipLaunchKernelGGL((rtx::rtx_min_scatter<12>), dim3((unsigned)n_blocks), dim3(rtx::kThreads), 0, st, d_in, n_slots, (uint32_t)w, offsets, d_out);
    }
    RTX_HIP(ctx, hipGetLastError());
    *d_total = total;
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
    const unsigned blocks = (ctx->ns + rtx::kThreads - 1) / rtx::kThreads;
    hipLaunchKernelGGL(rtx::rtx_update_spheres, dim3(blocks), dim3(rtx::kThreads), 0, ctx->stream,
                       (float4*)ctx->d_sph_geom.p, (float4*)ctx->d_sph_motion.p, ctx->ns, dt);
    RTX_HIP(ctx, hipGetLastError());
    return RTX_OK;
}

void* rtx_minimized_device_ptr(rtx_ctx* ctx) { return ctx ? ctx->d_min : nullptr; }

int rtx_minimize(rtx_ctx* ctx, int mode, size_t w, size_t h, const void* d_in, void* d_out, size_t* out_bytes)
{
    if (!ctx || !out_bytes) return RTX_ERR_INVALID_ARGUMENT;
    if (mode < RTX_BIT_ASCII || mode > RTX_SDL) return rtx_fail(ctx, RTX_ERR_INVALID_MODE, "invalid rendering mode");
    if (w == 0 || h == 0 || w >= (1ull << 31) || h >= (1ull << 31)) return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "w/h must be in [1, 2^31)");
    if (!d_in) {
        if (20 * w * h > ctx->capacity) return rtx_fail(ctx, RTX_ERR_TOO_LARGE, "frame larger than the context was created for");
        d_in = ctx->d_frame;
    }
    if (((uintptr_t)d_in & 3u) != 0) return rtx_fail(ctx, RTX_ERR_INVALID_ARGUMENT, "input frame must be 4-byte aligned");
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
    uint64_t total = 0;
    RTX_HIP(ctx, hipMemcpyAsync(&total, d_total, sizeof total, hipMemcpyDeviceToHost, ctx->stream));
    RTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out_bytes = (size_t)total;
    return RTX_OK;
}

int rtx_update(rtx_ctx* ctx, const rtx_params* params, int mode, double dt, int run_physics, void* host_out, size_t* out_bytes)
{
    if (!ctx || !params || !host_out || !out_bytes) return RTX_ERR_INVALID_ARGUMENT;
    int rc;
    // RayTracingManager.cu:89-107: physics first
    if (run_physics && (rc = rtx_update_objects(ctx, dt)) != RTX_OK) return rc;
    // :86 + :120-134: zero semantics and trace
    if ((rc = rtx_render(ctx, params, mode)) != RTX_OK) return rc;
    // :146: minimise on the device; :143 then only moves the minimised stream across PCIe
    size_t n = 0;
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
    if (!ctx->copy_stream) RTX_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (!sl.d_frame) {
        RTX_HIP(ctx, hipMalloc((void**)&sl.d_frame, ctx->capacity));
        RTX_HIP(ctx, hipMemsetAsync(sl.d_frame, 0, ctx->capacity, ctx->stream));
        RTX_HIP(ctx, hipMalloc((void**)&sl.d_min, ctx->capacity));
        RTX_HIP(ctx, hipHostMalloc((void**)&sl.h_total, sizeof(uint64_t), hipHostMallocDefault));
        RTX_HIP(ctx, hipEventCreateWithFlags(&sl.ev_ready, hipEventDisableTiming));
        RTX_HIP(ctx, hipEventCreateWithFlags(&sl.ev_copied, hipEventDisableTiming));
    }
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
    // the slot's frame buffer is caller-style memory for rtx_render_rows: whole frame, with the zero
    // semantics of the per-frame memset (the buffer starts zeroed; SDL frames write nothing, so clear)
    const bool rgb = mode >= RTX_RGB_ASCII;
    if (mode == RTX_SDL) {
        RTX_HIP(ctx, hipMemsetAsync(sl.d_frame, 0, 20 * w * h, ctx->stream));
    }
    if ((rc = rtx_render_rows(ctx, params, mode, 0, h, sl.d_frame, 0, ctx->stream, rgb ? RTX_RENDER_DEFAULT : RTX_RENDER_ZERO_TAIL)) != RTX_OK) return rc;
    uint64_t* d_total = nullptr;
    if ((rc = launch_minimize(ctx, sl.d_scan, mode, w, h, sl.d_frame, sl.d_min, &d_total)) != RTX_OK) return rc;
    RTX_HIP(ctx, hipMemcpyAsync(sl.h_total, d_total, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    RTX_HIP(ctx, hipEventRecord(sl.ev_ready, ctx->stream));
    // the length is needed on the host to size the copy: wait for this frame's kernels (the previous frame's
    // copy keeps running on the copy stream meanwhile)
    RTX_HIP(ctx, hipEventSynchronize(sl.ev_ready));
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
