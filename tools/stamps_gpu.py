#!/usr/bin/env python3
"""Diagnostic (librtx_hip_ablate.so): per-workgroup shader-clock stamps of one trace launch.
Prints when workgroups start, how long each phase takes, and how many workgroups overlap."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RTX_LIB"] = "librtx_hip_ablate.so"
import torch  # noqa: E402

sub = int(sys.argv[1]) if len(sys.argv) > 1 else 4
light = len(sys.argv) > 2 and sys.argv[2] == "light"     # only the start and end stamps of every workgroup (far less intrusive)
order = int(sys.argv[3]) if len(sys.argv) > 3 else -1    # RTX_OPT_TILE_ORDER
if light:
    os.environ["RTX_ABLATE"] = str(0x8000)
R = importlib.import_module("raytracing-in-windows-console_amd")
p, sph, pl = R.config_inputs("C2")
ctx = R.Context(1920, 1080)
ctx.set_scene(sph, pl)
ctx.set_option(R.OPT_SUBTILES, sub)
if order >= 0:
    ctx.set_option(R.OPT_TILE_ORDER, order)
for _ in range(5):
    ctx.render(p, R.RGB_ASCII)
ctx.synchronize()
nwg = 16384
buf = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
os.environ["RTX_STAMPS_PTR"] = str(buf.data_ptr())
ctx.render(p, R.RGB_ASCII)
ctx.synchronize()
del os.environ["RTX_STAMPS_PTR"]
s = buf.cpu().numpy().reshape(nwg, 16)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "stamps_sub%d%s_order%d.npy" % (sub, "_light" if light else "", order)), s[:4096])   # raw, for offline analysis (row = linear block id)
s = s[s[:, 15] != 0]
n = s.shape[0]
if light:
    s[:, 0:3] = 0   # (the raw dump keeps the per-wave SIMD / end time / work estimate of slots 0-11)
t0 = s[:, 0].min()
rt0 = s[:, 15].min()
npass = sub
last = 3 + npass - 1
print("workgroups", n, "sub-tiles", sub, "light" if light else "full", "tile order option", order)
if light:
    xcc = ((s[:, 14] >> 32) & 0xff).astype(int)
    st = (s[:, 15] - s[:, 15].min()) / 100.0
    en = (s[:, 13] - s[:, 15].min()) / 100.0
    q = [0, 10, 50, 90, 100]
    print("start us percentiles", np.round(np.percentile(st, q), 2))
    print("end   us percentiles", np.round(np.percentile(en, q), 2))
    print("lifetime us percentiles", np.round(np.percentile(en - st, q), 2), "sum of lifetimes", round(float((en - st).sum()), 1))
    ts = np.linspace(0, en.max(), 26)
    print("t (us)              ", [round(float(t), 1) for t in ts])
    print("resident workgroups ", [int(((st <= t) & (en > t)).sum()) for t in ts])
    sys.exit(0)
start = (s[:, 0] - t0)
end = (s[:, last] - t0)
print("kernel span (shader clocks): %d; by 100 MHz realtime: %.2f us  => shader clock %.2f GHz" % (
    end.max(), (s[:, 15].max() - rt0) / 100.0, end.max() / max(1, (s[:, 15].max() - rt0)) * 0.1))
q = [0, 10, 50, 90, 100]
print("start time percentiles      ", np.percentile(start, q).astype(int))
print("end time percentiles        ", np.percentile(end, q).astype(int))
print("lifetime percentiles        ", np.percentile(end - start, q).astype(int))
print("tables+frustum (0->1)       ", np.percentile(s[:, 1] - s[:, 0], q).astype(int))
print("staging (1->2)              ", np.percentile(s[:, 2] - s[:, 1], q).astype(int))
prev = s[:, 2]
for j in range(npass):
    cur = s[:, 3 + j]
    ok = cur != 0
    print("pass %d                      " % j, np.percentile((cur - prev)[ok], q).astype(int), "skipped" if not ok.all() else "")
    prev = np.where(ok, cur, prev)
# timeline on the global 100 MHz clock (s_memrealtime): starts, ends, residency
xcc = ((s[:, 14] >> 32) & 0xff).astype(int)
print("workgroups per XCC", np.bincount(xcc))
st = (s[:, 15] - s[:, 15].min()) / 100.0   # us
en = (s[:, 13] - s[:, 15].min()) / 100.0
print("start us percentiles", np.round(np.percentile(st, q), 2))
print("end   us percentiles", np.round(np.percentile(en, q), 2))
print("lifetime us percentiles", np.round(np.percentile(en - st, q), 2))
ts = np.linspace(0, en.max(), 26)
print("t (us)              ", [round(float(t), 1) for t in ts])
print("resident workgroups ", [int(((st <= t) & (en > t)).sum()) for t in ts])
