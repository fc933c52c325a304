"""the reference's Engine3D loop on a POSIX terminal (examples/console_engine.cpp over
include/rtx_compat.hpp), driven through a pseudo-terminal.

The test types keys into the pty (WASD, space / z, arrow keys, 1..5 and F-keys for SetRenderingMode as
Engine3D.cpp:178-197, x to quit), the engine runs in lock-step (one key per frame, fixed dt) and its printer
thread (PrintMachine.cpp:257-306 restated) writes every minimised frame to the terminal.  What arrives on the
master side must be, byte for byte, cursor-home + the stream rtx_update returns for the same camera, mode, dt and
scene (re-rendered here through the C ABI from the engine's trace) + colour reset, frame after frame."""
import os
import pty
import select
import subprocess
import time

import numpy as np
import pytest

import util as U

pytestmark = pytest.mark.gpu

W, H, DT = 96, 32, 0.26   # 4 frames per second of engine time: a sphere is spawned every 4th frame (Engine3D.cpp:60-69)

KEYS = [b"3", b"w", b"w", b"d", b"\x1b[A", b"\x1b[D", b"1", b" ", b"z", b"2", b"\x1bOS", b"\x1b[15~", b"s", b"a", b"\x1b[C", b"\x1b[B", b"w", b"x"]


def _drain(master, proc, deadline):
    out = bytearray()
    while time.time() < deadline:
        r, _, _ = select.select([master], [], [], 0.2)
        if r:
            try:
                chunk = os.read(master, 1 << 16)
            except OSError:     # EIO: the slave side has been closed
                break
            if not chunk:
                break
            out += chunk
        elif proc.poll() is not None:
            break
    return bytes(out)


def test_console_engine_under_a_pty(tmp_path):
    R = U.pkg()
    exe = os.path.join(R.PKG_DIR, "console_engine")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    trace_path = str(tmp_path / "trace.txt")
    cmd = [exe, str(W), str(H), "--lockstep", "--dt", repr(DT), "--no-status", "--trace", trace_path]
    try:
        master, slave = pty.openpty()
    except OSError:
        # a box without pty devices (the GPU boxes of this pipeline): the same bytes through pipes.  The terminal
        # handling itself (raw mode, key decoding on a real pty) is covered by tests/test_console_keys.py on CPU.
        master = slave = None
    if master is not None:
        proc = subprocess.Popen(cmd, stdin=slave, stdout=slave, stderr=subprocess.PIPE, close_fds=True)
        os.close(slave)
        try:
            os.write(master, b"".join(KEYS))
            got = _drain(master, proc, time.time() + 120)
            rc = proc.wait(timeout=30)
        finally:
            if proc.poll() is None:
                proc.kill()
            os.close(master)
        err = proc.stderr.read()
        got = got.replace(b"\r\n", b"\n")   # the tty's output processing (ONLCR); the stream itself holds no CR
    else:
        proc = subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        got, err = proc.communicate(b"".join(KEYS), timeout=120)
        rc = proc.returncode
    assert rc == 0, err.decode(errors="replace")

    # ---- what the engine says it rendered
    frames, spawns_after = [], {}
    with open(trace_path) as f:
        for line in f:
            t = line.split()
            if t[0] == "frame":
                frames.append({"mode": int(t[3]), "dt": float.fromhex(t[5]), "pos": [float.fromhex(v) for v in t[7:10]],
                               "rot": [float.fromhex(v) for v in t[11:14]], "bytes": int(t[15])})
            elif t[0] == "spawn":
                spawns_after.setdefault(len(frames) - 1, []).append([float.fromhex(v) for v in t[1:8]])
    assert len(frames) == len(KEYS) - 1     # one frame per key; 'x' quits before rendering
    assert sum(len(v) for v in spawns_after.values()) == len(frames) // 4

    # ---- the keys did what Engine3D::CheckKeyboard / Camera3D::Move / AddRot make them do
    modes = [fr["mode"] for fr in frames]
    assert modes[0] == 2 and modes[6] == 0 and modes[9] == 1 and modes[10] == 3 and modes[11] == 4 and modes[-1] == 4
    f32 = np.float32
    yaw0 = f32(np.pi)
    step = f32(DT) * f32(10.0)
    # 'w' twice from the start pose, 10 units per second along the NORMALISED static forward vector
    # (-sin yaw, -cos yaw, -cos yaw) (Camera3D.cpp:57-59, 142-163: its y component takes part in the length, not in the
    # motion), i.e. z grows by step / sqrt(2) per frame at yaw = pi
    sf = np.array([-np.sin(yaw0), -np.cos(yaw0), -np.cos(yaw0)], dtype=f32)
    sf = sf / f32(np.sqrt(f32(sf @ sf)))
    want_z = f32(0.0)
    for _ in range(2):
        want_z = want_z + sf[2] * step
    assert abs(frames[2]["pos"][2] - float(want_z)) < 1e-4 and abs(frames[2]["pos"][0]) < 1e-4
    assert frames[3]["pos"][0] < -1.0                      # 'd': along staticRight = (cos yaw, ...) = -x at yaw pi
    assert abs(frames[4]["rot"][0] - (-0.05)) < 1e-6       # arrow up: 25 counts * 0.002, pitch -= p * speed
    assert abs(frames[5]["rot"][1] - (float(yaw0) + 0.05)) < 1e-5
    assert abs(frames[7]["pos"][1] - float(step)) < 1e-5 and abs(frames[8]["pos"][1]) < 1e-5   # space up, z back down
    assert all(abs(fr["dt"] - DT) < 1e-12 for fr in frames)

    # ---- and the terminal received exactly rtx_update's streams
    want = bytearray(b"\x1b[?25l\x1b[2J")
    with R.Context(W, H) as ctx:
        ctx.set_reference_default_scene()
        for i, fr in enumerate(frames):
            p = R.camera_params(W, H, fr["pos"], fr["rot"])
            stream = ctx.update(p, fr["mode"], fr["dt"], run_physics=True)
            assert stream.size == fr["bytes"], "frame %d" % i
            want += b"\x1b[H" + stream.tobytes() + b"\x1b[m"
            for sp in spawns_after.get(i, []):
                ctx.add_sphere(sp[0], sp[1:4], sp[4:7])
    want += b"\x1b[m\x1b[?25h\n"
    assert len(got) == len(want)
    assert got == bytes(want)
