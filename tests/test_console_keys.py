"""The terminal side of examples/console_engine.cpp without a GPU: under a real pseudo-terminal the engine puts the
tty into raw input mode (no echo, no line buffering, as the reference reads key state directly, Engine3D.cpp:110-197),
decodes WASD / space / z / arrow keys / 1..5 / F1..F5 / x, and restores the terminal on exit."""
import os
import pty
import select
import subprocess
import termios
import time

import pytest

import util as U


def test_raw_mode_and_key_decoding_on_a_pty():
    R = U.pkg()
    exe = os.path.join(R.PKG_DIR, "console_engine")
    if not os.path.exists(exe):
        R.build()
    try:
        master, slave = pty.openpty()
    except OSError:
        pytest.skip("no pty devices here")
    before = termios.tcgetattr(slave)
    assert before[3] & termios.ECHO and before[3] & termios.ICANON
    proc = subprocess.Popen([exe, "--keys-only"], stdin=slave, stdout=slave, stderr=subprocess.PIPE, close_fds=True)
    out = bytearray()

    def read_until(marker, timeout=20.0):
        end = time.time() + timeout
        while marker not in out and time.time() < end:
            r, _, _ = select.select([master], [], [], 0.2)
            if r:
                try:
                    out.extend(os.read(master, 4096))
                except OSError:
                    break
        assert marker in out, bytes(out)

    try:
        read_until(b"raw 1")
        during = termios.tcgetattr(slave)
        assert not (during[3] & termios.ECHO) and not (during[3] & termios.ICANON)
        keys = [(b"w", "w"), (b"A", "a"), (b"s", "s"), (b"d", "d"), (b" ", "space"), (b"z", "shift"), (b"\x1b[A", "up"), (b"\x1b[B", "down"),
                (b"\x1b[C", "right"), (b"\x1b[D", "left"), (b"1", "mode0"), (b"5", "mode4"), (b"\x1bOP", "mode0"), (b"\x1bOQ", "mode1"),
                (b"\x1bOR", "mode2"), (b"\x1bOS", "mode3"), (b"\x1b[15~", "mode4"), (b"\x1b[11~", "mode0"), (b"k", "other")]
        for raw, name in keys:
            n = out.count(b"key ")
            os.write(master, raw)
            end = time.time() + 10
            while out.count(b"key ") == n and time.time() < end:
                r, _, _ = select.select([master], [], [], 0.2)
                if r:
                    out.extend(os.read(master, 4096))
            assert bytes(out).replace(b"\r\n", b"\n").rstrip().split(b"\n")[-1] == b"key " + name.encode(), (raw, bytes(out)[-80:])
        os.write(master, b"\x1b")             # a lone Escape: quit (Engine3D.cpp:173-176)
        read_until(b"key quit")
        assert proc.wait(timeout=10) == 0
        after = termios.tcgetattr(slave)
        assert after[3] & termios.ECHO and after[3] & termios.ICANON      # restored
    finally:
        if proc.poll() is None:
            proc.kill()
        os.close(master)
        os.close(slave)
