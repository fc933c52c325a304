/* c_abi_demo.c -- the drop-in boundary from plain C: include/rtx.h only, no C++ and no torch.
 * Renders the reference's start scene (Scene3D.cpp:28-33) through rtx_update and writes the minimised
 * ANSI stream to stdout (pipe it to a terminal: `./c_abi_demo 120 40 2`).
 *   gcc -std=c99 -Iinclude examples/c_abi_demo.c -L<pkg> -lrtx_hip -Wl,-rpath,<pkg> -Wl,-rpath-link,/opt/rocm/lib */
#include "rtx.h"

#include <stdio.h>
#include <stdlib.h>

int main(int argc, char** argv)
{
    const size_t W = argc > 1 ? (size_t)atoi(argv[1]) : 120, H = argc > 2 ? (size_t)atoi(argv[2]) : 40;
    const int mode = argc > 3 ? atoi(argv[3]) : RTX_RGB_ASCII;
    rtx_ctx* ctx = NULL;
    if (rtx_create(0, W, H, &ctx) != RTX_OK) {
        fprintf(stderr, "rtx_create: %s\n", rtx_last_error(NULL));
        return 1;
    }
    static const float spheres[5][7] = {
        {0.0f, 10.0f, 20.0f, 7.0f, 255.0f, 1.0f, 1.0f},   {5.0f, 10.0f, 20.0f, 6.0f, 1.0f, 255.0f, 1.0f},
        {10.0f, 10.0f, 40.0f, 10.0f, 1.0f, 1.0f, 255.0f}, {5.0f, 10.0f, 20.0f, 3.0f, 225.0f, 210.0f, 20.0f},
        {-5.0f, 10.0f, 40.0f, 4.0f, 225.0f, 10.0f, 220.0f}};
    rtx_scene_add_spheres(ctx, 5, &spheres[0][0]);
    const float pos[3] = {0.0f, -3.0f, 30.0f}, nrm[3] = {0.0f, 1.0f, 0.0f}, grey[3] = {100.0f, 100.0f, 100.0f};
    rtx_scene_add_plane(ctx, pos, nrm, grey, 10.0f, 20.0f);

    rtx_params params;
    if (rtx_camera_params(W, H, NULL, NULL, &params) != RTX_OK) {
        fprintf(stderr, "rtx_camera_params failed\n");
        return 1;
    }
    char* out = (char*)rtx_host_alloc(ctx, 20 * W * H);
    size_t n = 0;
    if (!out || rtx_update(ctx, &params, mode, 0.016, 1, out, &n) != RTX_OK) {
        fprintf(stderr, "rtx_update: %s\n", rtx_last_error(ctx));
        return 1;
    }
    fwrite(out, 1, n, stdout);
    fputs("\x1b[m\n", stdout);
    rtx_host_free(ctx, out);
    rtx_destroy(ctx);
    return 0;
}
