#!/usr/bin/env python3
"""What the VALU instructions the SQ class counters do NOT name cost to issue, per kernel, from the kernel's own ISA (runs in the build container).

    python tools/isa_mix.py [--out profiles/r05_final/isa_other_mix.json] [--valu-per-px k1=2561 k2=1684 ...]

tools/issue_model.py prices a kernel's measured dynamic instruction mix class by class.  The SQ counters name five classes (fp32 add / mul / fma,
transcendental, conversion, int32); everything else — compares and moves (2.7 cycles per wave64 instruction on this part), min / max / med3,
v_cndmask behind its compare, v_fma_mix, fract / floor (4.2), packed fp32 (4.55 for two results) — lands in "other", which rounds 3-4 priced at a
flat 4.2 and so over-priced the kernels whose "other" is mostly compares and moves (K1: issue share 1.07 of its own run time).  Here the
composition of "other" is read from the compiled kernel: every basic block's VALU opcodes, the blocks inside loops weighted by ONE trip factor per
kernel chosen so that the weighted static instruction count equals the measured dynamic VALU count per pixel (the hot loops — march steps,
denoise taps — are what the dynamic count is made of).  Output: per kernel the weighted opcode histogram of "other", its mean issue cost, and the
same for the named classes (a cross-check of the class mapping against the counters' own per-class counts).
"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "realism-effects_amd", "csrc")
# cycles per wave64 instruction per SIMD at 8 waves / SIMD (profiles/r03_microbench/valu_rates2.txt)
RATE_FAST, RATE_GEN, RATE_PK, RATE_TRANS, RATE_CND_COLD = 2.7, 4.2, 4.55, 8.3, 22.0
KERNELS = {  # key -> (source, mangled-name fragment, contraction flag)
    "k1_ssgi_march": ("k1_ssgi.hip", "k1_ssgi_marchILi6ELb0ELb0ELi0E", "off"),  # PROJ_CENTRED | PROJ_TABLE_POW2: the 16:9 frames' specialisation
    "k2_temporal_reproject": ("k2_temporal.hip", "k2_temporal_reprojectILi0ELi2ELb1ELb0ELb1E", "off"),
    "k3_poisson_denoise_pass0": ("k3_denoise.hip", "k3_tiledILb1ELi2ELi74ELb1EEE", "fast-honor-pragmas"),
    "k3_poisson_denoise_pass1": ("k3_denoise.hip", "k3_tiledILb0ELi2ELi76ELb1EEE", "fast-honor-pragmas"),
    "k4_compose": ("k4_compose.hip", "k4_composeILb1E", "fast-honor-pragmas"),
}


def op_class(op):
    """(counter class, issue cycles) of a VALU opcode"""
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op):
        return "trans", RATE_TRANS
    if re.match(r"v_pk_(fma|mul|add)_f32", op):
        return "other", RATE_PK
    if re.match(r"v_(add|sub|subrev)_f32", op):
        return "add", RATE_FAST
    if re.match(r"v_mul_(f32|legacy_f32)", op):
        return "mul", RATE_FAST
    if re.match(r"v_(fma|fmac|fmamk|fmaak|mad)_f32", op):
        return "fma", RATE_FAST
    if op.startswith("v_cvt_"):
        return "cvt", RATE_GEN
    if re.match(r"v_(add|sub|subrev)_(u32|i32|co_u32)|v_addc|v_subb", op) or re.match(r"v_(and|or|xor|not)_b32|v_(lshl|lshr|ashr)(rev)?_(b32|i32)", op):
        return "int", RATE_FAST
    if re.match(r"v_(mul_(i32|u32)_[iu]24|mul_(lo|hi)_[iu]32|mad_[iu]32_[iu]24|mad_[iu]64|lshl_add_u32|add_lshl_u32|add3_u32|lshl_or_b32|and_or_b32|or3_b32|bfe_[iu]32|bfi_b32|"
                r"min_[iu]32|max_[iu]32|med3_[iu]32|min3_[iu]32|max3_[iu]32|xad_u32|alignbit|mbcnt)", op):
        return "int", RATE_GEN
    if re.match(r"v_(cmp|cmpx)_", op) or re.match(r"v_mov_b32", op):
        return "other", RATE_FAST
    return "other", RATE_GEN  # min / max / med3 f32, cndmask, fma_mix, fract, floor, perm, pk_mov, readlane-free rest


def kernel_blocks(asm, frag):
    """ordered [(label, [opcodes], in_loop)] of the first kernel whose mangled name contains frag"""
    lines = open(asm).read().split("\n")
    start = next(i for i, ln in enumerate(lines) if re.match(r"^[_A-Za-z0-9$]+:", ln) and frag in ln.split(":")[0])
    blocks, cur, order = collections.OrderedDict(), "entry", {}
    blocks[cur] = []
    for ln in lines[start + 1:]:
        s = ln.strip()
        if s.startswith(".Lfunc_end") or s.startswith(".section"):
            break
        m = re.match(r"^(\.LBB[0-9_]+):", s)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        blocks[cur].append(s)
    labels = list(blocks)
    idx = {lab: i for i, lab in enumerate(labels)}
    in_loop = [False] * len(labels)
    for i, lab in enumerate(labels):  # a branch to an earlier (or the same) block closes a loop over the blocks in between
        for s in blocks[lab]:
            m = re.match(r"s_c?branch\S*\s+(\.LBB[0-9_]+)", s)
            if m and m.group(1) in idx and idx[m.group(1)] <= i:
                for j in range(idx[m.group(1)], i + 1):
                    in_loop[j] = True
    return [(lab, [s.split()[0] for s in blocks[lab] if s.startswith("v_") and not s.startswith("v_readlane") and not s.startswith("v_readfirstlane")], in_loop[i])
            for i, lab in enumerate(labels)]


def compile_asm(src, contract, out):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-ffp-contract=" + contract, "-Wno-unused-function",
           "-Wno-unused-value", "-Wno-unused-result", "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def mix(key, valu_per_px, tmp="/tmp"):
    src, frag, contract = KERNELS[key]
    asm = os.path.join(tmp, "isa_mix_%s.s" % src.replace(".hip", "") )
    if not os.path.exists(asm) or os.path.getmtime(asm) < os.path.getmtime(os.path.join(CSRC, src)):
        compile_asm(src, contract, asm)
    blocks = kernel_blocks(asm, frag)
    flat = sum(len(ops) for _, ops, lp in blocks if not lp)
    loop = sum(len(ops) for _, ops, lp in blocks if lp)
    # one trip factor for every block inside a loop: weighted static count == measured dynamic VALU instructions per wavefront-of-64-pixels
    trip = max((valu_per_px - flat) / loop, 1.0) if (valu_per_px and loop) else 1.0
    hist = collections.Counter()
    for _, ops, lp in blocks:
        for op in ops:
            hist[op] += trip if lp else 1.0
    per_class, cost_class = collections.Counter(), collections.Counter()
    other = collections.Counter()
    for op, n in hist.items():
        c, rate = op_class(op)
        per_class[c] += n
        cost_class[c] += n * rate
        if c == "other":
            other[op] += n
    tot = sum(per_class.values())
    return {"static_valu_outside_loops": flat, "static_valu_in_loops": loop, "loop_trip_factor": round(trip, 2), "weighted_valu": round(tot, 1),
            "class_per_px": {c: round(n, 1) for c, n in per_class.items()},
            "other_rate": round(cost_class["other"] / max(per_class["other"], 1e-9), 3),
            "int_rate": round(cost_class["int"] / max(per_class["int"], 1e-9), 3),
            "other_top": {op: round(n, 1) for op, n in other.most_common(12)}}


def main():
    out_path, valu = None, {}
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--out":
            out_path = args.pop(0)
        elif a == "--valu-per-px":
            while args and "=" in args[0]:
                k, v = args.pop(0).split("=")
                valu[k] = float(v)
    res = {}
    for key in KERNELS:
        want = next((v for k, v in valu.items() if key.startswith(k) or k == key), None)
        res[key] = mix(key, want)
        r = res[key]
        print("%-26s trip x%-6.2f weighted VALU %7.1f  classes %s\n%26s other: %.2f cycles/instr (int: %.2f)  %s" % (
            key, r["loop_trip_factor"], r["weighted_valu"], r["class_per_px"], "", r["other_rate"], r["int_rate"], r["other_top"]))
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
