#!/usr/bin/env python3
"""Static ISA budget of a trace kernel by stage: compiles csrc/rtx_kernels.hip for gfx950 with line tables, attributes every
instruction of one kernel to the source line it came from (.loc), maps lines to stages (ray generation, candidate loop, exact test,
winner fetch, planes, normals, shade, pow32, encode, stores ...) and counts VALU / SALU / LDS / VMEM / SMEM / wait / branch
instructions per stage.  Static counts: a loop body counts once (the weights column of the markdown output multiplies by the
trip counts measured for config 2).  CPU only (hipcc cross-compiles).

  python tools/isa_budget.py [kernel-substring] [--md]      default kernel: rtx_trace<2, true, 0, false>  (C2 RGB_ASCII, culling)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "raytracing-in-windows-console_amd")
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -fno-slp-vectorize "
         "-gline-tables-only -S --cuda-device-only").split()


def function_ranges(path, names):
    """{name: (first_line, last_line)} of the named functions in a source file (brace matching from the line that names them)."""
    lines = open(path).read().split("\n")
    out = {}
    for name in names:
        for i, ln in enumerate(lines):
            if re.search(r"\b%s\s*\(" % re.escape(name), ln) and ("__device__" in ln or "__device__" in lines[max(0, i - 1)] or "inline" in ln) and ";" not in ln.split("//")[0]:
                depth, j, seen = 0, i, False
                while j < len(lines):
                    depth += lines[j].count("{") - lines[j].count("}")
                    seen = seen or "{" in lines[j]
                    if seen and depth == 0:
                        break
                    j += 1
                out[name] = (i + 1, j + 1)
                break
    return out


def classify(mn):
    if mn.startswith("s_waitcnt") or mn == "s_nop" or mn.startswith("s_barrier") or mn.startswith("s_sleep"):
        return "wait"
    if mn.startswith("s_cbranch") or mn == "s_branch" or mn == "s_endpgm" or mn.startswith("s_setpc") or mn.startswith("s_swappc"):
        return "branch"
    if mn.startswith("s_load") or mn.startswith("s_buffer_load") or mn.startswith("s_memrealtime") or mn.startswith("s_memtime"):
        return "smem"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if mn.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if mn.startswith("v_"):
        return "valu"
    return "other"


def main():
    want = next((a for a in sys.argv[1:] if not a.startswith("--")), "rtx_trace<2, true, 0, false>")
    md = "--md" in sys.argv
    asm = os.path.join(PKG, "build", "rtx_kernels_lines.s")
    os.makedirs(os.path.dirname(asm), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", asm, "csrc/rtx_kernels.hip"], cwd=PKG, stderr=subprocess.DEVNULL)
    text = open(asm).read().split("\n")
    files = {}
    for ln in text:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', ln)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3))
    # the kernel's label
    labels = [ln.split(":")[0] for ln in text if re.match(r"^_Z\w+:", ln)]
    dem = subprocess.check_output(["c++filt"] + labels).decode().strip().split("\n")
    sym = None
    for lab, d in zip(labels, dem):
        if want in d:
            sym = lab
            name = d
            break
    if sym is None:
        raise SystemExit("no kernel matching %r" % want)
    start = text.index(next(ln for ln in text if ln.startswith(sym + ":")))
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))

    dev = function_ranges(os.path.join(PKG, "csrc", "rtx_device.hpp"),
                          ["rcp_generic", "sqrt_generic", "rcp_fast", "sqrt_fast", "in_safe_range", "rcp_cr", "sqrt_cr", "rcp_sqrt_cr", "normalize_gpu", "clampf",
                           "minf", "pow32", "u8_sat", "sphere_reject", "sphere_hit", "plane_hit", "shade", "ramp_index", "ansi_distance", "cube_level",
                           "cube_value", "palette_grey_value", "ansi256_from_rgb", "dot", "sub", "add", "mulf", "v3"])
    ker = function_ranges(os.path.join(PKG, "csrc", "rtx_kernels.hip"),
                          ["tile_culls", "tile_plane", "plane_invisible", "comes_first", "pixel_fields", "encode_and_store", "test_candidate", "scan_candidates",
                           "ray_from_tables", "stage_chunk", "load_item", "scene_items", "view_dir", "cross", "lds_barrier"])
    rec = function_ranges(os.path.join(PKG, "csrc", "rtx_records.hpp"), ["record_words", "digits_word"])

    body_lines = open(os.path.join(PKG, "csrc", "rtx_trace_body.inc")).read().split("\n")

    def anchor(pattern):
        return next(i + 1 for i, ln in enumerate(body_lines) if pattern in ln)

    a_tables = anchor("---- per-workgroup tables")
    a_stage = anchor("---- stage the whole scene once")
    a_pass = anchor("for (uint32_t j = 0; j < nsub; j++)")
    a_scan0 = anchor("uint32_t scanned = total;")
    a_over = anchor("// rare: more candidates than the list holds")
    a_winner = anchor("---- winner among spheres")
    a_planes = anchor("---- planes: hoisted form")
    a_shade = anchor("---- shade the winner")
    a_encode = anchor("encode_and_store<MODE, OUT>")
    a_cost = anchor("// per wave and pass: ray generation, planes and encoding")

    def body_stage(line):
        if line < a_tables: return "head (tile, cell list, first loads)"
        if line < a_stage: return "tables + pyramid"
        if line < a_pass: return "staging (cull + compact), plane table"
        if line < a_scan0: return "ray generation"
        if line < a_over: return "candidate loop"
        if line < a_winner: return "overflow fallback (rare)"
        if line < a_planes: return "winner fetch"
        if line < a_shade: return "planes"
        if line < a_encode: return "normal + shade"
        if line < a_cost: return "encode + store"
        return "cost estimate / tail"

    def in_range(rngs, names, line):
        return any(n in rngs and rngs[n][0] <= line <= rngs[n][1] for n in names)

    # pass 1: every instruction with the stage its own source line names (None: a helper -- dot, normalize_gpu, rcp / sqrt
    # sequences -- used by several stages); basic blocks from labels and branches
    insts = []   # (block, mnemonic, stage or None)
    block = 0
    fno, line = 0, 0
    for ln in text[start:end]:
        if re.match(r"^\.LBB\w+:", ln):
            block += 1
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
        if m:
            fno, line = int(m.group(1)), int(m.group(2))
            continue
        m = re.match(r"\s+([a-z_0-9]+)(\s|$)", ln)
        if not m or ln.strip().startswith((".", ";")):
            continue
        mn = m.group(1)
        f = files.get(fno, "")
        stage = None
        if f == "rtx_trace_body.inc":
            stage = body_stage(line)
        elif f == "rtx_device.hpp":
            if in_range(dev, ["pow32"], line): stage = "pow32"
            elif in_range(dev, ["shade", "clampf", "minf"], line): stage = "normal + shade"
            elif in_range(dev, ["sphere_reject"], line): stage = "candidate loop"
            elif in_range(dev, ["sphere_hit"], line): stage = "exact test (sphere_hit)"
            elif in_range(dev, ["ansi_distance", "cube_level", "cube_value", "palette_grey_value", "ansi256_from_rgb", "u8_sat", "ramp_index"], line): stage = "encode + store"
            elif in_range(dev, ["plane_hit"], line): stage = "planes"
            # (rcp / sqrt / normalize helpers: whichever stage called them last)
        elif f == "rtx_records.hpp":
            stage = "encode + store"
        elif f == "rtx_kernels.hip":
            if in_range(ker, ["stage_chunk", "tile_culls", "load_item", "scene_items"], line): stage = "staging (cull + compact), plane table"
            elif in_range(ker, ["tile_plane"], line): stage = "tables + pyramid"
            elif in_range(ker, ["plane_invisible", "view_dir", "cross"], line): stage = "staging (cull + compact), plane table"
            elif in_range(ker, ["test_candidate", "scan_candidates", "comes_first"], line): stage = "candidate loop"
            elif in_range(ker, ["ray_from_tables"], line): stage = "ray generation"
            elif in_range(ker, ["pixel_fields", "encode_and_store"], line): stage = "encode + store"
        insts.append((block, mn, stage))
        if classify(mn) == "branch":
            block += 1
    # pass 2: a helper instruction belongs to the stage most of its basic block's named instructions belong to (a block without
    # any: to the previous block's stage)
    by_block = {}
    for b, mn, st in insts:
        if st is not None:
            by_block.setdefault(b, {})
            by_block[b][st] = by_block[b].get(st, 0) + 1
    stages, order = {}, []
    prev = "head (tile, cell list, first loads)"
    last_block = -1
    for b, mn, st in insts:
        if b != last_block:
            if b in by_block:
                prev = max(by_block[b], key=lambda k: by_block[b][k])
            last_block = b
        stage = st if st is not None else prev
        if stage not in stages:
            stages[stage] = {}
            order.append(stage)
        c = classify(mn)
        stages[stage][c] = stages[stage].get(c, 0) + 1

    cols = ["valu", "salu", "lane", "lds", "vmem", "smem", "wait", "branch", "other"]
    tot = {c: sum(s.get(c, 0) for s in stages.values()) for c in cols}
    if md:
        print("| stage | " + " | ".join(cols) + " |")
        print("|---|" + "---|" * len(cols))
        for st in order:
            print("| %s | " % st + " | ".join(str(stages[st].get(c, 0)) for c in cols) + " |")
        print("| **all** | " + " | ".join(str(tot[c]) for c in cols) + " |")
    else:
        print(name)
        print("%-44s" % "stage" + "".join("%8s" % c for c in cols))
        for st in order:
            print("%-44s" % st + "".join("%8d" % stages[st].get(c, 0) for c in cols))
        print("%-44s" % "all" + "".join("%8d" % tot[c] for c in cols))


if __name__ == "__main__":
    main()
