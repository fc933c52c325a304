#!/usr/bin/env python3
"""Generates tests/golden/golden.json and c1_scene.npz from the CPU oracle.

The reference cannot run here (no CUDA, no Windows headers) and ships no golden vectors, so the
vectors are produced by the oracle, which tests/test_oracle_pins.py pins to the known answers
SURVEY.md 8(c) recorded from the reference's own sources.  Oracle configuration for every entry:
pinned pow32, saturating float->uint8 (flags = 0), i.e. exactly what the HIP path implements.

Hashes are standard FNV-1a-64 (offset basis 14695981039346656037) of the full zero-initialised
20*W*H buffer after the trace, and of the Minimize output.

Usage: python tests/golden/make_golden.py [--big]     (--big adds C3, C4, C5: minutes of CPU)
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle as O  # noqa: E402
import util as U  # noqa: E402


def entry(p, scene, mode, threads):
    W, H = int(p.x), int(p.y)
    t = time.time()
    buf = O.render(p, scene, mode, threads=threads)
    mini = O.minimize(mode, buf, W, H)
    S = 20 if mode >= O.RGB_ASCII else 12
    rec = buf[:S * W * H].reshape(H, W, S)
    return {
        "frame_fnv1a64": O.fnv1a64(buf),
        "frame_sha256": hashlib.sha256(buf.tobytes()).hexdigest(),   # for checkers that may not load the oracle (bench.py --verify)
        "minimized_fnv1a64": O.fnv1a64(mini),
        "minimized_bytes": int(mini.size),
        "foreground_pixels": int((rec[:, :W - 1, 2] == ord("3")).sum()) if mode in (O.BIT_ASCII, O.RGB_ASCII) else None,
        "oracle_seconds": round(time.time() - t, 2),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    R = U.pkg()
    out_path = os.path.join(HERE, "golden.json")
    gold = {}
    if os.path.exists(out_path):
        with open(out_path) as f:
            gold = json.load(f)
    gold["_about"] = "oracle-generated (pinned pow32, saturating u8 conversion); see make_golden.py"

    # reference default scene, default camera
    for (w, h) in ((400, 150), (1920, 1080)):
        p = O.camera_params(w, h)
        sc = O.Scene.reference_default()
        for mode in range(5):
            gold["default_%dx%d_%s" % (w, h, O.MODE_NAMES[mode])] = entry(p, sc, mode, args.threads)

    # BASELINE configs on the synthetic scenes
    todo = [("C1", range(5)), ("C2", (O.BIT_ASCII, O.RGB_ASCII))]
    if args.big:
        todo += [("C3", (O.RGB_ASCII,)), ("C4", (O.RGB_ASCII,)), ("C5", (O.RGB_ASCII,))]
    for name, modes in todo:
        p, sph, pl = R.config_inputs(name)
        sc = O.Scene.from_arrays(sph, pl)
        op = U.oracle_params(p)
        for mode in modes:
            key = "%s_%s" % (name, O.MODE_NAMES[mode])
            gold[key] = entry(op, sc, mode, args.threads)
            print(key, gold[key], flush=True)
        if name == "C1":
            np.savez(os.path.join(HERE, "c1_scene.npz"), spheres=sph, planes=pl,
                     inv_v=np.array(p.inv_v[:], dtype=np.float32), cam_pos=np.array(p.cam_pos[:], dtype=np.float32),
                     scalars=np.array([p.element1, p.element2, p.cam_far], dtype=np.float32),
                     wh=np.array([p.x, p.y], dtype=np.int64))
            # one frame verbatim, so that a diff can be localised without the oracle
            buf = O.render(op, sc, O.RGB_ASCII)
            np.savez_compressed(os.path.join(HERE, "c1_rgb_ascii_frame.npz"), frame=buf)
        with open(out_path, "w") as f:
            json.dump(gold, f, indent=1, sort_keys=True)
    with open(out_path, "w") as f:
        json.dump(gold, f, indent=1, sort_keys=True)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
