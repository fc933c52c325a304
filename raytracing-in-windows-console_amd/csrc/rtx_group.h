// rtx_group.h -- a context that renders on several devices (rtx_group_create): what the entry points of rtx_api.cpp /
// rtx_post.hip call when the context they are handed is the root of a device group.
//
// One process, one member context (scene replica, streams, slab buffer) per logical rank, rank 0 = the root = the context
// the caller holds.  Frames shard by pixel rows -- rank g traces rows [g H / N, (g + 1) H / N) with the GLOBAL row index in ray
// generation (RayTracing.cu:12,16); a block of rows is one contiguous byte range of the frame (RayTracing.cu:238,457) --
// and the slabs are gathered into the root's device memory: grouped ncclSend / ncclRecv over xGMI (RCCL, ncclCommInitAll)
// or hipMemcpyPeerAsync; SURVEY.md 8(e).
#pragma once

#include "rtx_ctx.h"

#include <functional>

struct rtx_group; // defined in rtx_group.cpp

namespace rtxgroup {

// the group a context is the root of, or nullptr
inline rtx_group* of(const rtx_ctx* ctx) { return ctx ? ctx->group : nullptr; }

void destroy(rtx_group* g); // members other than the root, buffers, communicators (the root context is the caller's)

// scene edits and options, replicated: called by the root's entry point AFTER it has applied the call to itself
int scene_clear(rtx_ctx* root);
int scene_add_sphere(rtx_ctx* root, const float pos[3], float radius, const float rgb[3]);
int scene_add_plane(rtx_ctx* root, const float pos[3], const float normal[3], const float rgb[3], float width, float height);
int scene_set_sphere_motion(rtx_ctx* root, unsigned index, int mover, float speed);
int set_option(rtx_ctx* root, int option, int64_t value);
int update_objects(rtx_ctx* root, double dt); // every member steps its own replica (the same arithmetic on the same values)

// The whole frame of `p`, sharded over the members, as RECORDS in d_out on the root's device (nullptr: the root's own frame
// buffer, with its zero-fill bookkeeping), complete in stream order on the root's stream.
int render_frame(rtx_ctx* root, const rtx_params* p, int mode, void* d_out, unsigned flags);
// n whole frames (rtx_submit_frames on a group): every rank traces its rows of all of them with one call, one gather per chunk.
int render_frames(rtx_ctx* root, size_t n, const rtx_params* params, int mode, void* const* d_outs, void* const* streams);
// ... as compact pixel words (W * H of them, include/rtx.h RTX_RENDER_COMPACT) in the group's word buffer on the root's device,
// complete in stream order on the root's stream; *d_words receives the buffer.  What rtx_update minimises from.
int render_words(rtx_ctx* root, const rtx_params* p, int mode, const uint32_t** d_words);


// fn(rank, member) for every rank: rank 0 on the caller's thread, the others on their submission threads where those are in use
// (RTX_OPT_GROUP_THREADS) -- and then a job may wait for its own device without holding the others up -- or one after the other.
// Returns when all have returned; the first failure by rank, reported on the root.
int run_on_ranks(rtx_ctx* root, const std::function<int(int, rtx_ctx*)>& fn);
bool threads_active(rtx_ctx* root); // the ranks other than the root have submission threads (started here if they are wanted and not yet running)
// rtx_update without a gather (RTX_OPT_GROUP_UPDATE): wanted for this group?  ... and what rtx_post.hip reports back
bool update_direct_wanted(const rtx_ctx* root);
void update_direct_done(rtx_ctx* root, bool ok); // ok: counted; not ok: the group gathers on its root from now on

} // namespace rtxgroup
