// rtx_group.cpp -- one context over several devices (rtx_group_create, include/rtx.h): the row-sharded frame behind the
// C ABI.  The reference's single consumer is RayTracingManager::Update (RayTracingManager.cu:76-154, hand-off at :150) on one
// device; here the same calls -- rtx_scene_*, rtx_render, rtx_update, rtx_update_begin/_end -- on a group context trace the
// frame on N devices and assemble it on the root's.
//
//   rank g (member context g: its own scene replica, stream and slab buffer on device devices[g]) traces rows
//   [g H / N, (g + 1) H / N) with the global row index (RayTracing.cu:12,16) -- the root straight into the destination, the
//   others into their slab -- and the slabs are gathered into the root's device memory at their byte offsets (a block of rows
//   is one contiguous byte range, RayTracing.cu:238,457):
//     * RCCL: ncclCommInitAll over the device list, one grouped ncclSend (rank g, on its stream) / ncclRecv (root, on the
//       root's stream) pair per non-empty slab -- slabs are ragged when N does not divide H, hence send/recv and not
//       ncclGather; librccl is loaded on first use (dlopen), so a single-device process never maps it;
//     * peer copy: hipMemcpyPeerAsync on the sender's stream, ordered by events -- the form for device lists that repeat a
//       device (several logical ranks on one GPU: how the one-GPU test box walks N = 4 and N = 8) and the fallback when RCCL
//       cannot be loaded or initialised.
//   Wire format: compact pixel words by default (4 instead of 12 / 20 bytes per pixel cross xGMI; the root expands them
//   through record_words<MODE>, rtx_expand, or -- rtx_update -- minimises straight from the words), or records.
//
// Single caller thread (the reference drives the path from its main thread, Engine3D.cpp:81-107): the launches of all ranks
// are queued by that thread, device by device; nothing here blocks except where the ABI says so.
#include "rtx_group.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// librccl, once per process (the copy the process already holds, if any: PyTorch ships one under the same SONAME)
RcclApi* rccl_api(std::string* why)
{
    static RcclApi api;
    static bool tried = false;
    static std::string error;
    if (!tried) {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) {
            error = std::string("librccl could not be loaded: ") + (dlerror() ? dlerror() : "?");
        } else {
            api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(dlsym(api.lib, "ncclCommInitAll"));
            api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
            api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(api.lib, "ncclGroupStart"));
            api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(api.lib, "ncclGroupEnd"));
            api.Send = reinterpret_cast<decltype(api.Send)>(dlsym(api.lib, "ncclSend"));
            api.Recv = reinterpret_cast<decltype(api.Recv)>(dlsym(api.lib, "ncclRecv"));
            api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
            if (!api.CommInitAll || !api.CommDestroy || !api.GroupStart || !api.GroupEnd || !api.Send || !api.Recv) {
                error = "librccl lacks ncclCommInitAll / ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd";
                api.CommInitAll = nullptr;
            }
        }
    }
    if (!api.CommInitAll) {
        if (why) *why = error;
        return nullptr;
    }
    return &api;
}

} // namespace

// One submission thread per rank other than the root: a rank's share of a frame is a launch, a copy and two events -- about 15 us
// of host work -- and on the caller's one thread N ranks cost N times that per frame (8 ranks: 132 us per C2 frame on the one-GPU
// box, where the GPU work is 42).  The caller posts every rank its job, queues the root's own launch meanwhile, and waits for the
// SUBMISSIONS (not the GPU work) before it orders the root's stream after the ranks' events.
struct RankWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, busy = false, quit = false;
    int rc = RTX_OK;
};

struct rtx_group {
    int n = 0;
    std::vector<rtx_ctx*> member; // [0] = the root (owned by the caller)
    std::vector<int> device;
    bool distinct = true;         // no device appears twice in the list
    std::vector<char> direct;     // per rank: its device and the root's address each other's memory (same device, or peer access enabled) and strided copies between them work
    std::vector<hipEvent_t> ev_done; // per rank, on its device: its slab has arrived on the root (peer copies)
    hipEvent_t ev_free = nullptr;    // root's device: the destination may be overwritten
    uint32_t* d_words = nullptr;     // root's device: the frame as compact words (W * H)
    size_t words_cap = 0;
    uint8_t* d_root_slab = nullptr;  // root's device: its own slab, only when the root's rows travel through RCCL too (RTX_EXCHANGE_RCCL_ALL)
    size_t root_slab_cap = 0;
    int64_t opt_exchange = RTX_EXCHANGE_AUTO, opt_wire = RTX_WIRE_AUTO;
    std::vector<ncclComm_t> comms;   // RCCL communicators, one per rank (empty: not initialised)
    bool rccl_failed = false;        // initialisation was tried and failed: peer copies from then on
    std::string rccl_error;
    int exchange_in_use = RTX_EXCHANGE_PEER_COPY;
    uint64_t stat_gathers = 0, stat_last_bytes = 0;
    std::vector<std::unique_ptr<RankWorker>> workers; // [r] for r >= 1 (empty: jobs run on the caller's thread)
    int64_t opt_update = -1;                           // RTX_OPT_GROUP_UPDATE: -1 auto, 0 gather on the root, 1 every rank its own rows
    bool update_direct_failed = false;                 // a HIP error on the direct path: gather from now on
    uint64_t stat_direct_updates = 0;
    int64_t opt_threads = -1;                          // -1 auto (on with two or more distinct devices), 0 off, 1 on
};

namespace {

int member_fail(rtx_ctx* root, int rank, rtx_ctx* m, int rc)
{
    char head[48];
    std::snprintf(head, sizeof head, "group rank %d: ", rank);
    return rtx_fail(root, rc, std::string(head) + (m ? m->error : std::string("?")));
}

void worker_main(RankWorker* w, int device)
{
    hipSetDevice(device);
    std::unique_lock<std::mutex> lock(w->m);
    for (;;) {
        w->cv.wait(lock, [&] { return w->has_job || w->quit; });
        if (w->quit) return;
        w->has_job = false;
        std::function<int()> job;
        job.swap(w->job);
        lock.unlock();
        const int rc = job();
        lock.lock();
        w->rc = rc;
        w->busy = false;
        w->cv.notify_all();
    }
}

bool threads_in_use(rtx_group* g)
{
    // auto: where the list names at least two DISTINCT devices -- there the ranks' GPU work overlaps and the caller's thread is what
    // is left in the way.  With every rank on one GPU (the one-GPU box's walk) the GPU serialises the ranks anyway and the threads were
    // measured to change nothing (8 ranks, a frame per call: 114 us with, 120 without; 8 frames per call: 39.1 / 39.4): off.
    if (g->opt_threads == 0 || g->n < 2) return false;
    if (g->opt_threads < 0) {
        bool two = false;
        for (int r = 1; r < g->n; r++) two = two || g->device[(size_t)r] != g->device[0];
        if (!two) return false;
    }
    if (g->workers.empty()) {
        g->workers.resize((size_t)g->n);
        for (int r = 1; r < g->n; r++) {
            g->workers[(size_t)r].reset(new RankWorker());
            g->workers[(size_t)r]->th = std::thread(worker_main, g->workers[(size_t)r].get(), g->device[(size_t)r]);
        }
    }
    return true;
}

// Rank r's job: on its submission thread, or here and now.
void post(rtx_group* g, int r, std::function<int()> fn, bool threaded, std::vector<int>& rcs)
{
    if (!threaded) {
        rcs[(size_t)r] = fn();
        return;
    }
    RankWorker* w = g->workers[(size_t)r].get();
    std::lock_guard<std::mutex> lock(w->m);
    w->job = std::move(fn);
    w->has_job = true;
    w->busy = true;
    w->rc = RTX_OK;
    w->cv.notify_all();
}

// Every posted job has been SUBMITTED (its launches and copies are queued); the first failure, if any, by rank.
int wait_posted(rtx_ctx* root, rtx_group* g, bool threaded, std::vector<int>& rcs)
{
    int first_rc = RTX_OK, first_r = 0;
    for (int r = 1; r < g->n; r++) {
        if (threaded) {
            RankWorker* w = g->workers[(size_t)r].get();
            std::unique_lock<std::mutex> lock(w->m);
            w->cv.wait(lock, [&] { return !w->busy; });
            rcs[(size_t)r] = w->rc;
        }
        if (rcs[(size_t)r] != RTX_OK && first_rc == RTX_OK) {
            first_rc = rcs[(size_t)r];
            first_r = r;
        }
    }
    return first_rc == RTX_OK ? RTX_OK : member_fail(root, first_r, g->member[(size_t)first_r], first_rc);
}

void stop_workers(rtx_group* g)
{
    for (auto& w : g->workers) {
        if (!w) continue;
        {
            std::lock_guard<std::mutex> lock(w->m);
            w->quit = true;
            w->cv.notify_all();
        }
        if (w->th.joinable()) w->th.join();
    }
    g->workers.clear();
}

// rows of rank r for a frame of H rows: the partition SURVEY.md 8(e) names (and sharding.row_bounds uses)
inline uint64_t bound(uint64_t H, int r, int n) { return H * (uint64_t)r / (uint64_t)n; }

bool rccl_wanted(const rtx_group* g)
{
    if (g->opt_exchange == RTX_EXCHANGE_PEER_COPY || g->rccl_failed || !g->distinct) return false;
    if (g->opt_exchange == RTX_EXCHANGE_RCCL_ALL) return true;
    return g->n > 1; // AUTO and RCCL: where there is a peer to exchange with
}

// Communicators on first use.  Failure is not an error of the frame: the group falls back to peer copies and says why
// (RTX_STAT_GROUP_EXCHANGE, rtx_group_exchange_note).
bool rccl_ready(rtx_ctx* root, rtx_group* g)
{
    if (!g->comms.empty()) return true;
    std::string why;
    RcclApi* api = rccl_api(&why);
    if (!api) {
        g->rccl_failed = true;
        g->rccl_error = why;
        return false;
    }
    std::vector<ncclComm_t> comms((size_t)g->n, nullptr);
    const ncclResult_t rc = api->CommInitAll(comms.data(), g->n, g->device.data());
    if (rc != ncclSuccess) {
        g->rccl_failed = true;
        g->rccl_error = std::string("ncclCommInitAll failed: ") + (api->GetErrorString ? api->GetErrorString(rc) : "?");
        (void)hipGetLastError();
        return false;
    }
    g->comms.swap(comms);
    (void)root;
    return true;
}

// The frame of `p` sharded over the ranks into d_dst on the root's device, S bytes per pixel (4: compact words; 12 / 20:
// records), rank r's rows at byte offset bound(r) * W * S; complete in stream order on the root's stream.
// own_frame: d_dst is the root's own frame buffer (records): the root's launch then goes through the context's zero-fill
// bookkeeping exactly as a single-device rtx_render would.
int gather_frame(rtx_ctx* root, rtx_group* g, const rtx_params* p, int mode, bool compact, uint8_t* d_dst, bool own_frame, unsigned root_flags)
{
    const uint64_t W = p->x, H = p->y;
    const uint64_t S = compact ? 4u : (mode >= RTX_RGB_ASCII ? 20u : 12u);
    const unsigned slab_flags = compact ? (unsigned)RTX_RENDER_COMPACT : (unsigned)RTX_RENDER_DEFAULT;
    const int n = g->n;
    const bool use_rccl = rccl_wanted(g) && rccl_ready(root, g);
    const bool root_through_rccl = use_rccl && g->opt_exchange == RTX_EXCHANGE_RCCL_ALL;
    g->exchange_in_use = use_rccl ? RTX_EXCHANGE_RCCL : RTX_EXCHANGE_PEER_COPY;
    int rc;

    if (root_through_rccl) {
        // (before any job is posted: nothing below this point returns while a rank's submission thread is at work)
        const size_t need = (size_t)(bound(H, 1, n) * W * S);
        if (g->root_slab_cap < need) {
            RTX_HIP(root, hipSetDevice(root->device));
            if (g->d_root_slab) {
                RTX_HIP(root, hipStreamSynchronize(root->stream));
                hipFree(g->d_root_slab);
            }
            g->d_root_slab = nullptr;
            g->root_slab_cap = 0;
            if (hipMalloc((void**)&g->d_root_slab, need) != hipSuccess) return rtx_fail(root, RTX_ERR_OUT_OF_MEMORY, "hipMalloc failed for the root's slab");
            g->root_slab_cap = need;
        }
    }
    // ---- every rank traces its rows (and, with peer copies, sends them): a job per rank, on its submission thread
    const bool threaded = threads_in_use(g);
    std::vector<int> rcs((size_t)n, RTX_OK);
    for (int r = 1; r < n; r++) {
        const uint64_t rows = bound(H, r + 1, n) - bound(H, r, n);
        if (rows * W * S > g->member[(size_t)r]->capacity) return rtx_fail(root, RTX_ERR_TOO_LARGE, "frame larger than the group was created for");
    }
    if (!use_rccl) {
        // the destination may still be read by what the root queued before this frame (the previous frame's expansion or
        // minimise pass): the ranks' copies wait for that, their traces do not have to
        RTX_HIP(root, hipSetDevice(root->device));
        RTX_HIP(root, hipEventRecord(g->ev_free, root->stream));
    }
    uint64_t moved = 0;
    for (int r = 1; r < n; r++) {
        const uint64_t r0 = bound(H, r, n), rows = bound(H, r + 1, n) - r0;
        if (rows == 0) continue;
        rtx_ctx* m = g->member[(size_t)r];
        const size_t bytes = (size_t)(rows * W * S);
        moved += bytes;
        const rtx_params pp = *p;
        uint8_t* dst = d_dst + r0 * W * S;
        hipEvent_t ev_free = g->ev_free, ev_done = g->ev_done[(size_t)r];
        const int root_device = root->device;
        post(g, r, [=]() -> int {
            // the member's own frame buffer is its slab: rows [r0, r0 + rows) at its start
            const int rc2 = rtx_render_rows(m, &pp, mode, (size_t)r0, (size_t)rows, m->d_frame, (size_t)r0, m->stream, slab_flags);
            if (rc2 != RTX_OK) return rc2;
            m->dirty_hi = m->capacity; // (the buffer is used as scratch: whatever a later plain render on this member assumes about it is void)
            if (!use_rccl) {
                RTX_HIP(m, hipSetDevice(m->device));
                RTX_HIP(m, hipStreamWaitEvent(m->stream, ev_free, 0));
                RTX_HIP(m, hipMemcpyPeerAsync(dst, root_device, m->d_frame, m->device, bytes, m->stream));
                RTX_HIP(m, hipEventRecord(ev_done, m->stream));
            }
            return RTX_OK;
        }, threaded, rcs);
    }
    {
        const uint64_t rows0 = bound(H, 1, n);
        if (root_through_rccl) {
            // test form (one-GPU box: RCCL at N = 1): the root's rows too are traced into a slab and travel through the exchange
            rc = rows0 ? rtx_render_rows(root, p, mode, 0, (size_t)rows0, g->d_root_slab, 0, root->stream, slab_flags) : RTX_OK;
            if (rc == RTX_OK && own_frame) rc = rtx_frame_zero_semantics(root, mode, W, H, 0u);
        } else if (own_frame) {
            rc = rtx_render_rows(root, p, mode, 0, (size_t)rows0, nullptr, 0, root->stream, RTX_RENDER_DEFAULT);
        } else {
            rc = rtx_render_rows(root, p, mode, 0, (size_t)rows0, d_dst, 0, root->stream, root_flags | slab_flags);
        }
    }
    {
        // (the ranks' submissions are waited for whatever the root's own launch returned: their jobs hold pointers into this frame)
        const int wrc = wait_posted(root, g, threaded, rcs);
        if (rc != RTX_OK) return rc;
        if (wrc != RTX_OK) return wrc;
    }

    // ---- the slabs travel to the root
    if (use_rccl) {
        moved = 0;
        RcclApi* api = rccl_api(nullptr);
        ncclResult_t nrc = api->GroupStart();
        for (int r = root_through_rccl ? 0 : 1; r < n && nrc == ncclSuccess; r++) {
            const uint64_t r0 = bound(H, r, n), rows = bound(H, r + 1, n) - r0;
            if (rows == 0) continue;
            rtx_ctx* m = g->member[(size_t)r];
            const size_t bytes = (size_t)(rows * W * S);
            const void* src = r == 0 ? (const void*)g->d_root_slab : (const void*)m->d_frame;
            nrc = api->Send(src, bytes, ncclChar, 0, g->comms[(size_t)r], m->stream);
            if (nrc == ncclSuccess) nrc = api->Recv(d_dst + r0 * W * S, bytes, ncclChar, r, g->comms[0], root->stream);
            moved += bytes;
        }
        const ncclResult_t erc = api->GroupEnd();
        if (nrc == ncclSuccess) nrc = erc;
        if (nrc != ncclSuccess) {
            return rtx_fail(root, RTX_ERR_HIP, std::string("RCCL exchange failed: ") + (api->GetErrorString ? api->GetErrorString(nrc) : "?"));
        }
        RTX_HIP(root, hipSetDevice(root->device));
    } else {
        // (the copies were queued by the ranks' jobs, behind ev_free; the root's stream goes on when they have landed)
        RTX_HIP(root, hipSetDevice(root->device));
        for (int r = 1; r < n; r++) {
            if (bound(H, r + 1, n) - bound(H, r, n) == 0) continue;
            RTX_HIP(root, hipStreamWaitEvent(root->stream, g->ev_done[(size_t)r], 0));
        }
    }
    g->stat_gathers++;
    g->stat_last_bytes = moved;
    return RTX_OK;
}

int ensure_words(rtx_ctx* root, rtx_group* g, uint64_t W, uint64_t H)
{
    const size_t need = (size_t)(W * H);
    if (g->words_cap >= need) return RTX_OK;
    RTX_HIP(root, hipSetDevice(root->device));
    if (g->d_words) {
        RTX_HIP(root, hipDeviceSynchronize());
        hipFree(g->d_words);
    }
    g->d_words = nullptr;
    g->words_cap = 0;
    if (hipMalloc((void**)&g->d_words, need * sizeof(uint32_t)) != hipSuccess) return rtx_fail(root, RTX_ERR_OUT_OF_MEMORY, "hipMalloc failed for the group's word buffer");
    g->words_cap = need;
    return RTX_OK;
}

bool wire_is_compact(const rtx_group* g) { return g->opt_wire != RTX_WIRE_RECORDS; }

int validate_frame(rtx_ctx* root, const rtx_params* p, int mode)
{
    if (mode < RTX_BIT_ASCII || mode > RTX_SDL) return rtx_fail(root, RTX_ERR_INVALID_MODE, "invalid rendering mode");
    const uint64_t W = p->x, H = p->y;
    if (W == 0 || H == 0 || W >= (1ull << 31) || H >= (1ull << 31)) return rtx_fail(root, RTX_ERR_INVALID_ARGUMENT, "params.x / params.y must be in [1, 2^31)");
    if (20 * W * H > root->capacity) return rtx_fail(root, RTX_ERR_TOO_LARGE, "frame larger than the context was created for");
    return RTX_OK;
}

} // namespace

namespace rtxgroup {

void destroy(rtx_group* g)
{
    if (!g) return;
    stop_workers(g);
    if (!g->comms.empty()) {
        RcclApi* api = rccl_api(nullptr);
        for (size_t r = 0; r < g->comms.size(); r++) {
            if (api && g->comms[r]) {
                hipSetDevice(g->device[r]);
                api->CommDestroy(g->comms[r]);
            }
        }
    }
    for (int r = 1; r < g->n; r++) {
        if ((size_t)r < g->member.size() && g->member[(size_t)r]) rtx_destroy(g->member[(size_t)r]);
    }
    for (size_t r = 0; r < g->ev_done.size(); r++) {
        if (g->ev_done[r]) {
            hipSetDevice(g->device[r]);
            hipEventDestroy(g->ev_done[r]);
        }
    }
    if (!g->device.empty()) hipSetDevice(g->device[0]);
    if (g->ev_free) hipEventDestroy(g->ev_free);
    if (g->d_words) hipFree(g->d_words);
    if (g->d_root_slab) hipFree(g->d_root_slab);
    delete g;
}

int scene_clear(rtx_ctx* root)
{
    rtx_group* g = root->group;
    for (int r = 1; r < g->n; r++) {
        const int rc = rtx_scene_clear(g->member[(size_t)r]);
        if (rc != RTX_OK) return member_fail(root, r, g->member[(size_t)r], rc);
    }
    return RTX_OK;
}

int scene_add_sphere(rtx_ctx* root, const float pos[3], float radius, const float rgb[3])
{
    rtx_group* g = root->group;
    for (int r = 1; r < g->n; r++) {
        const int idx = rtx_scene_add_sphere(g->member[(size_t)r], pos, radius, rgb);
        if (idx < 0) return member_fail(root, r, g->member[(size_t)r], -idx);
    }
    return RTX_OK;
}

int scene_add_plane(rtx_ctx* root, const float pos[3], const float normal[3], const float rgb[3], float width, float height)
{
    rtx_group* g = root->group;
    for (int r = 1; r < g->n; r++) {
        const int idx = rtx_scene_add_plane(g->member[(size_t)r], pos, normal, rgb, width, height);
        if (idx < 0) return member_fail(root, r, g->member[(size_t)r], -idx);
    }
    return RTX_OK;
}

int scene_set_sphere_motion(rtx_ctx* root, unsigned index, int mover, float speed)
{
    rtx_group* g = root->group;
    for (int r = 1; r < g->n; r++) {
        const int rc = rtx_scene_set_sphere_motion(g->member[(size_t)r], index, mover, speed);
        if (rc != RTX_OK) return member_fail(root, r, g->member[(size_t)r], rc);
    }
    return RTX_OK;
}

int set_option(rtx_ctx* root, int option, int64_t value)
{
    rtx_group* g = root->group;
    if (option == RTX_OPT_GROUP_EXCHANGE) {
        if (value < RTX_EXCHANGE_AUTO || value > RTX_EXCHANGE_RCCL_ALL) return rtx_fail(root, RTX_ERR_INVALID_ARGUMENT, "RTX_OPT_GROUP_EXCHANGE: enum rtx_group_exchange");
        if ((value == RTX_EXCHANGE_RCCL || value == RTX_EXCHANGE_RCCL_ALL) && !g->distinct) {
            return rtx_fail(root, RTX_ERR_INVALID_ARGUMENT, "RTX_OPT_GROUP_EXCHANGE: RCCL needs a device list without repeats (one communicator rank per GPU)");
        }
        g->opt_exchange = value;
        if (value != RTX_EXCHANGE_PEER_COPY) g->rccl_failed = false; // asked for again: try again
        return RTX_OK;
    }
    if (option == RTX_OPT_GROUP_THREADS) {
        if (value < -1 || value > 1) return rtx_fail(root, RTX_ERR_INVALID_ARGUMENT, "RTX_OPT_GROUP_THREADS: -1 (auto), 0 or 1");
        g->opt_threads = value;
        if (value == 0) stop_workers(g);
        return RTX_OK;
    }
    if (option == RTX_OPT_GROUP_UPDATE) {
        if (value < -1 || value > 1) return rtx_fail(root, RTX_ERR_INVALID_ARGUMENT, "RTX_OPT_GROUP_UPDATE: -1 (auto), 0 or 1");
        g->opt_update = value;
        if (value == 1) g->update_direct_failed = false; // asked for again: try again
        return RTX_OK;
    }
    if (option == RTX_OPT_GROUP_WIRE) {
        if (value < RTX_WIRE_AUTO || value > RTX_WIRE_COMPACT) return rtx_fail(root, RTX_ERR_INVALID_ARGUMENT, "RTX_OPT_GROUP_WIRE: enum rtx_group_wire");
        g->opt_wire = value;
        return RTX_OK;
    }
    for (int r = 1; r < g->n; r++) {
        const int rc = rtx_set_option(g->member[(size_t)r], option, value);
        if (rc != RTX_OK) return member_fail(root, r, g->member[(size_t)r], rc);
    }
    return RTX_OK;
}

int update_objects(rtx_ctx* root, double dt)
{
    rtx_group* g = root->group;
    for (int r = 1; r < g->n; r++) {
        const int rc = rtx_update_objects(g->member[(size_t)r], dt);
        if (rc != RTX_OK) return member_fail(root, r, g->member[(size_t)r], rc);
    }
    return RTX_OK;
}

int render_frame(rtx_ctx* root, const rtx_params* p, int mode, void* d_out, unsigned flags)
{
    rtx_group* g = root->group;
    int rc = validate_frame(root, p, mode);
    if (rc != RTX_OK) return rc;
    const uint64_t W = p->x, H = p->y;
    const bool own = d_out == nullptr;
    if (mode == RTX_SDL || (g->n == 1 && g->opt_exchange != RTX_EXCHANGE_RCCL_ALL)) {
        // RayTrace_SDL writes nothing (RayTracing.cu:787-794); a group of one has nothing to gather
        return rtx_render_rows(root, p, mode, 0, (size_t)H, d_out, 0, root->stream, flags);
    }
    if (!own && (((uintptr_t)d_out & 3u) != 0)) return rtx_fail(root, RTX_ERR_INVALID_ARGUMENT, "output buffer must be 4-byte aligned");
    if (!wire_is_compact(g)) {
        if (!own && mode < RTX_RGB_ASCII && (flags & RTX_RENDER_ZERO_TAIL)) {
            RTX_HIP(root, hipSetDevice(root->device));
            RTX_HIP(root, hipMemsetAsync((uint8_t*)d_out + 12 * W * H, 0, 8 * W * H, root->stream));
        }
        return gather_frame(root, g, p, mode, false, own ? root->d_frame : (uint8_t*)d_out, own, 0u);
    }
    // compact words into the group's buffer, then the records (rtx_expand: the same record_words<MODE> the trace kernel uses)
    if ((rc = ensure_words(root, g, W, H)) != RTX_OK) return rc;
    if ((rc = gather_frame(root, g, p, mode, true, (uint8_t*)g->d_words, false, 0u)) != RTX_OK) return rc;
    if (own) {
        if ((rc = rtx_frame_zero_semantics(root, mode, W, H, 0u)) != RTX_OK) return rc;
    } else if (mode < RTX_RGB_ASCII && (flags & RTX_RENDER_ZERO_TAIL)) {
        RTX_HIP(root, hipSetDevice(root->device));
        RTX_HIP(root, hipMemsetAsync((uint8_t*)d_out + 12 * W * H, 0, 8 * W * H, root->stream));
    }
    const rtx_segment seg = {0u, 0u, W * H};
    return rtx_expand(root, mode, g->d_words, own ? (void*)root->d_frame : d_out, &seg, 1, root->stream);
}

// n whole frames, sharded, into d_outs[i] (records, on the root's device): every rank traces its rows of a chunk of frames
// with ONE call (rtx_submit_slabs on its member context: one batched launch where the plan allows, RTX_OPT_BATCH), the slabs
// of the chunk are gathered, the root expands.  The host cost per frame of a sharded launch-by-launch loop (a launch, a copy
// and two events per rank and frame: ~15 us per rank on the caller's one thread) is paid once per chunk instead.
int render_frames(rtx_ctx* root, size_t n, const rtx_params* params, int mode, void* const* d_outs, void* const* streams)
{
    rtx_group* g = root->group;
    const int N = g->n;
    int rc;
    for (size_t i = 0; i < n; i++) {
        if ((rc = validate_frame(root, &params[i], mode)) != RTX_OK) return rc;
        if (!d_outs[i] || ((uintptr_t)d_outs[i] & 3u) != 0) return rtx_fail(root, RTX_ERR_INVALID_ARGUMENT, "rtx_submit_frames: null or misaligned frame buffer");
    }
    // the frames are traced and assembled on the root's own stream: first order it after whatever the caller's streams hold for
    // these buffers (on one device rtx_submit_frames queues frame i ON streams[i]; here the streams are made to wait at the end)
    if (streams) {
        RTX_HIP(root, hipSetDevice(root->device));
        for (size_t i = 0; i < n; i++) {
            hipStream_t sc = (hipStream_t)streams[i];
            if (!sc || sc == root->stream) continue;
            bool seen = false;
            for (size_t k = 0; k < i; k++) seen = seen || streams[k] == streams[i];
            if (seen) continue;
            RTX_HIP(root, hipEventRecord(g->ev_done[0], sc));
            RTX_HIP(root, hipStreamWaitEvent(root->stream, g->ev_done[0], 0));
        }
    }
    bool uniform = mode != RTX_SDL && N > 1;
    for (size_t i = 1; i < n && uniform; i++) {
        uniform = params[i].x == params[0].x && params[i].y == params[0].y;
    }
    if (!uniform) {
        // frames of different sizes (or nothing to shard): one by one
        for (size_t i = 0; i < n; i++) {
            if ((rc = render_frame(root, &params[i], mode, d_outs[i], RTX_RENDER_DEFAULT)) != RTX_OK) return rc;
        }
    } else {
        const uint64_t W = params[0].x, H = params[0].y;
        const bool compact = wire_is_compact(g);
        const uint64_t S = compact ? 4u : (mode >= RTX_RGB_ASCII ? 20u : 12u);
        const unsigned slab_flags = compact ? (unsigned)RTX_RENDER_COMPACT : (unsigned)RTX_RENDER_DEFAULT;
        // frames per chunk: what the batched kernel takes, and what the ranks' slab buffers (their frame buffers) hold
        size_t chunk = 16;
        for (int r = 1; r < N; r++) {
            const uint64_t rows = bound(H, r + 1, N) - bound(H, r, N);
            if (rows) chunk = std::min<size_t>(chunk, (size_t)(g->member[(size_t)r]->capacity / (rows * W * S)));
        }
        if (chunk == 0) return rtx_fail(root, RTX_ERR_TOO_LARGE, "frame larger than the group was created for");
        const bool use_rccl = rccl_wanted(g) && g->opt_exchange != RTX_EXCHANGE_RCCL_ALL && rccl_ready(root, g);
        g->exchange_in_use = use_rccl ? RTX_EXCHANGE_RCCL : RTX_EXCHANGE_PEER_COPY;
        std::vector<void*> ptrs(chunk), strs(chunk);
        for (size_t first = 0; first < n; first += chunk) {
            const size_t m = std::min(chunk, n - first);
            if (compact && (rc = ensure_words(root, g, W, H * m)) != RTX_OK) return rc;
            // where frame i of the chunk is assembled: its words in the group's buffer, or its records in the caller's
            auto dest = [&](size_t i) { return compact ? (uint8_t*)(g->d_words + i * W * H) : (uint8_t*)d_outs[first + i]; };
            const bool threaded = threads_in_use(g);
            std::vector<int> rcs((size_t)N, RTX_OK);
            if (!use_rccl) {
                RTX_HIP(root, hipSetDevice(root->device));
                RTX_HIP(root, hipEventRecord(g->ev_free, root->stream));
            }
            uint64_t moved = 0;
            const rtx_params* pfirst = &params[first];
            uint8_t* dest0 = dest(0);
            for (int r = 1; r < N; r++) {
                const uint64_t r0 = bound(H, r, N), rows = bound(H, r + 1, N) - r0;
                if (rows == 0) continue;
                rtx_ctx* mem = g->member[(size_t)r];
                const size_t bytes = (size_t)(rows * W * S);
                moved += bytes * m;
                hipEvent_t ev_free = g->ev_free, ev_done = g->ev_done[(size_t)r];
                const int root_device = root->device;
                const bool strided = compact && m > 1 && g->direct[(size_t)r];
                auto* direct_flag = &g->direct[(size_t)r];
                std::vector<uint8_t*> dests(m);
                for (size_t i = 0; i < m; i++) dests[i] = dest(i) + r0 * W * S;
                post(g, r, [=]() -> int {
                    // the rank's rows of the chunk's frames back to back in its buffer: ONE call, one batched launch where the plan allows
                    std::vector<void*> p2(m), s2(m);
                    for (size_t i = 0; i < m; i++) {
                        p2[i] = mem->d_frame + i * bytes;
                        s2[i] = mem->stream;
                    }
                    const int rc2 = rtx_submit_slabs(mem, m, pfirst, mode, (size_t)r0, (size_t)rows, p2.data(), (size_t)r0, s2.data(), nullptr, slab_flags);
                    if (rc2 != RTX_OK) return rc2;
                    mem->dirty_hi = mem->capacity;
                    if (!use_rccl) {
                        RTX_HIP(mem, hipSetDevice(mem->device));
                        RTX_HIP(mem, hipStreamWaitEvent(mem->stream, ev_free, 0));
                        bool copied = false;
                        if (strided) {
                            // the chunk's slabs lie back to back here and H*W words apart on the root: one strided copy.  A runtime
                            // that refuses the strided form between two devices gets the plain copies below, from then on.
                            copied = hipMemcpy2DAsync(dest0 + r0 * W * S, (size_t)(H * W * S), mem->d_frame, bytes, bytes, m, hipMemcpyDeviceToDevice, mem->stream) == hipSuccess;
                            if (!copied) {
                                (void)hipGetLastError();
                                *direct_flag = 0;
                            }
                        }
                        if (!copied) {
                            for (size_t i = 0; i < m; i++) {
                                RTX_HIP(mem, hipMemcpyPeerAsync(dests[i], root_device, mem->d_frame + i * bytes, mem->device, bytes, mem->stream));
                            }
                        }
                        RTX_HIP(mem, hipEventRecord(ev_done, mem->stream));
                    }
                    return RTX_OK;
                }, threaded, rcs);
            }
            {
                const uint64_t rows0 = bound(H, 1, N);
                for (size_t i = 0; i < m; i++) {
                    ptrs[i] = dest(i);
                    strs[i] = root->stream;
                }
                rc = rows0 ? rtx_submit_slabs(root, m, &params[first], mode, 0, (size_t)rows0, ptrs.data(), 0, strs.data(), nullptr, slab_flags) : RTX_OK;
            }
            {
                const int wrc = wait_posted(root, g, threaded, rcs);
                if (rc != RTX_OK) return rc;
                if (wrc != RTX_OK) return wrc;
            }
            if (use_rccl) {
                moved = 0;
                RcclApi* api = rccl_api(nullptr);
                ncclResult_t nrc = api->GroupStart();
                for (int r = 1; r < N && nrc == ncclSuccess; r++) {
                    const uint64_t r0 = bound(H, r, N), rows = bound(H, r + 1, N) - r0;
                    if (rows == 0) continue;
                    rtx_ctx* mem = g->member[(size_t)r];
                    const size_t bytes = (size_t)(rows * W * S);
                    for (size_t i = 0; i < m && nrc == ncclSuccess; i++) {
                        nrc = api->Send(mem->d_frame + i * bytes, bytes, ncclChar, 0, g->comms[(size_t)r], mem->stream);
                        if (nrc == ncclSuccess) nrc = api->Recv(dest(i) + r0 * W * S, bytes, ncclChar, r, g->comms[0], root->stream);
                        moved += bytes;
                    }
                }
                const ncclResult_t erc = api->GroupEnd();
                if (nrc == ncclSuccess) nrc = erc;
                if (nrc != ncclSuccess) return rtx_fail(root, RTX_ERR_HIP, std::string("RCCL exchange failed: ") + (api->GetErrorString ? api->GetErrorString(nrc) : "?"));
                RTX_HIP(root, hipSetDevice(root->device));
            } else {
                RTX_HIP(root, hipSetDevice(root->device));
                for (int r = 1; r < N; r++) {
                    if (bound(H, r + 1, N) - bound(H, r, N) == 0) continue;
                    RTX_HIP(root, hipStreamWaitEvent(root->stream, g->ev_done[(size_t)r], 0));
                }
            }
            g->stat_gathers += m;
            g->stat_last_bytes = moved / m;
            if (compact) {
                for (size_t i = 0; i < m; i++) {
                    const rtx_segment seg = {0u, 0u, W * H};
                    if ((rc = rtx_expand(root, mode, g->d_words + i * W * H, d_outs[first + i], &seg, 1, root->stream)) != RTX_OK) return rc;
                }
            }
        }
    }
    // the caller's streams see the frames once the root's stream has them
    bool any = false;
    for (size_t i = 0; i < n; i++) any = any || (streams && streams[i] && (hipStream_t)streams[i] != root->stream);
    if (any) {
        RTX_HIP(root, hipSetDevice(root->device));
        RTX_HIP(root, hipEventRecord(g->ev_done[0], root->stream));
        for (size_t i = 0; i < n; i++) {
            hipStream_t s = (hipStream_t)streams[i];
            if (!s || s == root->stream) continue;
            bool seen = false;
            for (size_t k = 0; k < i; k++) seen = seen || streams[k] == streams[i];
            if (!seen) RTX_HIP(root, hipStreamWaitEvent(s, g->ev_done[0], 0));
        }
    }
    return RTX_OK;
}

int run_on_ranks(rtx_ctx* root, const std::function<int(int, rtx_ctx*)>& fn)
{
    rtx_group* g = root->group;
    const bool threaded = threads_in_use(g);
    std::vector<int> rcs((size_t)g->n, RTX_OK);
    for (int r = 1; r < g->n; r++) {
        rtx_ctx* m = g->member[(size_t)r];
        post(g, r, [&fn, r, m]() -> int { return fn(r, m); }, threaded, rcs);
    }
    const int rc0 = fn(0, root);
    const int wrc = wait_posted(root, g, threaded, rcs);
    return rc0 != RTX_OK ? rc0 : wrc;
}

bool threads_active(rtx_ctx* root) { return root->group != nullptr && threads_in_use(root->group); }

bool update_direct_wanted(const rtx_ctx* root)
{
    const rtx_group* g = root->group;
    if (!g || g->n < 2 || g->opt_update == 0 || g->update_direct_failed) return false;
    if (g->opt_update == 1) return true;
    for (int r = 1; r < g->n; r++) {
        if (g->device[(size_t)r] != g->device[0]) return true; // (auto: where there is more than one PCIe link to use)
    }
    return false;
}

void update_direct_done(rtx_ctx* root, bool ok)
{
    rtx_group* g = root->group;
    if (ok) {
        g->stat_direct_updates++;
    } else {
        g->update_direct_failed = true;
    }
}

int render_words(rtx_ctx* root, const rtx_params* p, int mode, const uint32_t** d_words)
{
    rtx_group* g = root->group;
    int rc = validate_frame(root, p, mode);
    if (rc != RTX_OK) return rc;
    if (mode == RTX_SDL) return rtx_fail(root, RTX_ERR_INVALID_MODE, "no pixel words in RTX_SDL");
    if ((rc = ensure_words(root, g, p->x, p->y)) != RTX_OK) return rc;
    if ((rc = gather_frame(root, g, p, mode, true, (uint8_t*)g->d_words, false, 0u)) != RTX_OK) return rc;
    *d_words = g->d_words;
    return RTX_OK;
}

} // namespace rtxgroup

extern "C" {

int rtx_group_create(int ndev, const int* devices, size_t max_w, size_t max_h, rtx_ctx** out)
{
    if (!out) return rtx_fail(nullptr, RTX_ERR_INVALID_ARGUMENT, "rtx_group_create: out is NULL");
    *out = nullptr;
    if (ndev < 1 || ndev > 64) return rtx_fail(nullptr, RTX_ERR_INVALID_ARGUMENT, "rtx_group_create: ndev must be in [1, 64]");
    rtx_group* g = new (std::nothrow) rtx_group();
    if (!g) return rtx_fail(nullptr, RTX_ERR_OUT_OF_MEMORY, "rtx_group_create: out of host memory");
    g->n = ndev;
    for (int r = 0; r < ndev; r++) {
        g->device.push_back(devices ? devices[r] : r);
        for (int q = 0; q < r; q++) g->distinct = g->distinct && g->device[(size_t)q] != g->device[(size_t)r];
    }
    g->member.assign((size_t)ndev, nullptr);
    g->ev_done.assign((size_t)ndev, nullptr);
    rtx_ctx* root = nullptr;
    int rc = RTX_OK;
    for (int r = 0; r < ndev && rc == RTX_OK; r++) {
        rc = rtx_create(g->device[(size_t)r], max_w, max_h, &g->member[(size_t)r]);
        if (rc == RTX_OK && hipEventCreateWithFlags(&g->ev_done[(size_t)r], hipEventDisableTiming) != hipSuccess) {
            rc = rtx_fail(nullptr, RTX_ERR_HIP, "rtx_group_create: hipEventCreate failed");
        }
    }
    root = g->member[0];
    if (rc == RTX_OK) {
        if (hipSetDevice(g->device[0]) != hipSuccess || hipEventCreateWithFlags(&g->ev_free, hipEventDisableTiming) != hipSuccess) {
            rc = rtx_fail(nullptr, RTX_ERR_HIP, "rtx_group_create: hipEventCreate failed");
        }
    }
    if (rc != RTX_OK) {
        const std::string msg = rtx_last_error(nullptr);
        rtxgroup::destroy(g); // destroys the members other than the root
        if (root) rtx_destroy(root);
        return rtx_fail(nullptr, rc, msg);
    }
    // direct access between the root's device and every other one, both ways, where the hardware offers it (xGMI):
    // hipMemcpyPeerAsync then moves the bytes GPU to GPU instead of staging them through the host
    g->direct.assign((size_t)ndev, 1);
    for (int r = 1; r < ndev; r++) {
        const int a = g->device[0], b = g->device[(size_t)r];
        if (a == b) continue;
        int can_ab = 0, can_ba = 0;
        bool ok = true;
        if (hipDeviceCanAccessPeer(&can_ab, a, b) == hipSuccess && can_ab) {
            hipSetDevice(a);
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            ok = ok && (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled);
        } else {
            ok = false;
        }
        if (hipDeviceCanAccessPeer(&can_ba, b, a) == hipSuccess && can_ba) {
            hipSetDevice(b);
            const hipError_t e = hipDeviceEnablePeerAccess(a, 0);
            ok = ok && (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled);
        } else {
            ok = false;
        }
        (void)hipGetLastError();
        g->direct[(size_t)r] = ok ? 1 : 0; // (without it hipMemcpyPeerAsync still works, staged by the runtime; strided copies are not used)
    }
    (void)hipGetLastError();
    hipSetDevice(g->device[0]);
    root->group = g;
    *out = root;
    return RTX_OK;
}

int rtx_group_size(const rtx_ctx* ctx) { return ctx ? (ctx->group ? ctx->group->n : 1) : 0; }

rtx_ctx* rtx_group_member(rtx_ctx* ctx, int rank)
{
    if (!ctx) return nullptr;
    if (!ctx->group) return rank == 0 ? ctx : nullptr;
    return (rank >= 0 && rank < ctx->group->n) ? ctx->group->member[(size_t)rank] : nullptr;
}

int rtx_group_rows(const rtx_ctx* ctx, size_t h, int rank, size_t* row0, size_t* rows)
{
    const int n = rtx_group_size(ctx);
    if (!ctx || !row0 || !rows || rank < 0 || rank >= n) return RTX_ERR_INVALID_ARGUMENT;
    *row0 = (size_t)bound(h, rank, n);
    *rows = (size_t)(bound(h, rank + 1, n) - bound(h, rank, n));
    return RTX_OK;
}

const char* rtx_group_exchange_note(const rtx_ctx* ctx)
{
    if (!ctx || !ctx->group) return "single device: nothing is exchanged";
    const rtx_group* g = ctx->group;
    if (g->exchange_in_use == RTX_EXCHANGE_RCCL) return "RCCL: grouped ncclSend / ncclRecv, one communicator rank per device (ncclCommInitAll)";
    if (!g->distinct) return "hipMemcpyPeerAsync (the device list repeats a device: several logical ranks share a GPU)";
    if (g->rccl_failed) return g->rccl_error.c_str();
    return "hipMemcpyPeerAsync";
}

} // extern "C"

// statistics of the group for rtx_get_option (rtx_api.cpp)
bool rtx_group_stat(const rtx_ctx* ctx, int option, int64_t* value)
{
    const rtx_group* g = ctx->group;
    switch (option) {
    case RTX_STAT_GROUP_SIZE: *value = g ? g->n : 1; return true;
    case RTX_STAT_GROUP_EXCHANGE: *value = g ? g->exchange_in_use : 0; return true;
    case RTX_STAT_GROUP_GATHERS: *value = g ? (int64_t)g->stat_gathers : 0; return true;
    case RTX_STAT_GROUP_BYTES: *value = g ? (int64_t)g->stat_last_bytes : 0; return true;
    case RTX_OPT_GROUP_EXCHANGE: *value = g ? g->opt_exchange : 0; return true;
    case RTX_OPT_GROUP_WIRE: *value = g ? g->opt_wire : 0; return true;
    case RTX_OPT_GROUP_THREADS: *value = g ? g->opt_threads : 0; return true;
    case RTX_OPT_GROUP_UPDATE: *value = g ? g->opt_update : 0; return true;
    case RTX_STAT_GROUP_DIRECT_UPDATES: *value = g ? (int64_t)g->stat_direct_updates : 0; return true;
    default: return false;
    }
}
