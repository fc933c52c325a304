// temporary: minimise / physics entry points, implemented in rtx_post.hip
#include "../../include/rtx.h"
extern "C" {
int rtx_minimize(rtx_ctx*, int, size_t, size_t, const void*, void*, size_t*) { return RTX_ERR_INVALID_ARGUMENT; }
void* rtx_minimized_device_ptr(rtx_ctx*) { return nullptr; }
int rtx_update_objects(rtx_ctx*, double) { return RTX_ERR_INVALID_ARGUMENT; }
int rtx_update(rtx_ctx*, const rtx_params*, int, double, int, void*, size_t*) { return RTX_ERR_INVALID_ARGUMENT; }
}
