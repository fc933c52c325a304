import importlib, sys
sys.path.insert(0, "/root/repo")
import torch
R = importlib.import_module("raytracing-in-windows-console_amd")
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sub = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for W, H in ((400, 150), (1920, 1080), (3840, 2160)):
    ctx = R.Context(W, H)
    ctx.set_reference_default_scene()
    ctx.set_option(R.OPT_TILE_LOG2_W, tile)
    ctx.set_option(R.OPT_SUBTILES, sub)
    p = R.camera_params(W, H)
    for mode in (R.RGB_ASCII,):
        for _ in range(5):
            ctx.render(p, mode)
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(100):
            ctx.render(p, mode)
        ms = ctx.timer_stop() / 100
        print("default scene %dx%d %s: %.2f us/frame, %.1f Grays/s  [%s]" % (W, H, R.MODE_NAMES[mode], ms * 1e3, (W - 1) * H / ms / 1e6, ctx.last_kernel))
    ctx.close()
