#!/usr/bin/env python3
"""Static issue-cost model of a gfx950 kernel's ISA (development helper, runs in the build container).

    hipcc ... -S --cuda-device-only k3_denoise.hip -o k3.s
    python tools/isa_cost.py k3.s k3_tiledILb0ELi2 [--loops]

Every VALU instruction is priced with the issue rate MEASURED on MI355X (tools/microbench/valu_rates.hip,
profiles/r0N_*/valu_rates.txt, cycles per wave64 instruction per SIMD at 8 waves/SIMD): these kernels run with their VALU pipes
saturated (SQ_ACTIVE_INST_VALU*4 / (SIMDs * cycles) = 0.9-1.1), so a kernel's time is, to first order, the sum of its
executed VALU instructions' issue cycles.  The tool prints that sum per basic block / loop so that an edit's effect on a loop body can
be read before spending GPU minutes; trip counts are the reader's business (--trip LABEL=N).
"""
import re
import sys
from collections import Counter, OrderedDict

# cycles per wave64 instruction per SIMD at 8 waves / SIMD (profiles/r03_microbench/valu_rates2.txt; r02's table had v_fma_f32 at 4.0 —
# its operands were (a, s, s): the same register twice costs a second read cycle; with distinct sources fma issues like add / mul)
RATES = OrderedDict([
    ("trans", 8.3),    # v_exp/log/rcp/rsq/sqrt/sin/cos_f32
    ("pk", 4.55),      # v_pk_{fma,mul,add}_f32, v_pk_mov_b32
    ("fast", 2.7),     # v_add/sub/mul/fma/fmac/fmamk/fmaak_f32, v_mov/and/or/xor_b32, v_add/sub_u32, v_cmp_*
    ("cndmask", 11.0), # v_cndmask_b32 (measured 10.8 behind a fresh v_cmp, 22 with a loop-invariant VCC: see profiles/r03_microbench)
    ("generic", 4.2),  # everything else: v_cvt_*, v_min/max/med3, v_floor/fract, v_mad_i32_i24, v_lshl_add, v_fma_mix_f32, v_bfe, v_perm ...
])
TRANS = re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_(f32|f16|legacy_f32)")
PK = re.compile(r"^v_pk_")
FAST = re.compile(r"^v_((add|sub|subrev|mul|fma|fmac|fmamk|fmaak)_f32|(mov|and|or|xor)_b32|(add|sub|subrev)_u32|cmp_)")
CND = re.compile(r"^v_cndmask_b32")


def classify(op):
    if TRANS.match(op):
        return "trans"
    if PK.match(op):
        return "pk"
    if CND.match(op):
        return "cndmask"
    if FAST.match(op):
        return "fast"
    return "generic"


def parse(path, kernel_substr):
    """-> ordered {label: [ops]} of the first kernel whose mangled name contains kernel_substr"""
    blocks = OrderedDict()
    cur = None
    inside = False
    for line in open(path):
        s = line.strip()
        if not inside:
            if s.endswith(":") or re.match(r"^[_A-Za-z0-9.$]+:", s):
                name = s.split(":")[0]
                if kernel_substr in name and not name.startswith("."):
                    inside = True
                    cur = "entry"
                    blocks[cur] = []
            continue
        if s.startswith(".Lfunc_end") or s.startswith("s_endpgm") and False:
            break
        m = re.match(r"^(\.LBB[0-9_]+):", s)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        if not s or s.startswith(";") or s.startswith("."):
            if s.startswith(".section") or s.startswith(".rodata"):
                break
            continue
        op = s.split()[0]
        blocks[cur].append((op, s))
    return blocks


def cost(ops):
    c = Counter()
    for op, _ in ops:
        if op.startswith("v_") and not op.startswith("v_readlane") and not op.startswith("v_readfirstlane"):
            c[classify(op)] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
            c["vmem"] += 1
        elif op.startswith("s_waitcnt") or op.startswith("s_nop"):
            c["wait/nop"] += 1
        elif op.startswith("s_cbranch") or op.startswith("s_branch"):
            c["branch"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    cyc = sum(RATES[k] * c[k] for k in RATES)
    return c, cyc


def main():
    if len(sys.argv) < 3:
        print(__doc__)
        sys.exit(2)
    blocks = parse(sys.argv[1], sys.argv[2])
    if not blocks:
        sys.exit("kernel not found")
    trips = {}
    for a in sys.argv[3:]:
        if a.startswith("--trip"):
            continue
        if "=" in a:
            k, v = a.split("=")
            trips[k] = float(v)
    tot = Counter()
    totc = 0.0
    wsum = 0.0
    print("%-12s %5s %5s %5s %5s %5s %5s | %4s %4s %4s %4s | %8s" % ("block", "valu", "trans", "pk", "fast", "cnd", "gen", "lds", "vmem", "salu", "br", "cycles"))
    for lab, ops in blocks.items():
        c, cyc = cost(ops)
        nv = sum(c[k] for k in RATES)
        if nv == 0 and c["lds"] + c["vmem"] == 0:
            continue
        t = trips.get(lab, 1.0)
        print("%-12s %5d %5d %5d %5d %5d %5d | %4d %4d %4d %4d | %8.0f%s" % (lab, nv, c["trans"], c["pk"], c["fast"], c["cndmask"], c["generic"], c["lds"], c["vmem"], c["salu"],
                                                                       c["branch"], cyc, ("  x%g" % t) if t != 1 else ""))
        tot.update(c)
        totc += cyc
        wsum += cyc * t
    nv = sum(tot[k] for k in RATES)
    print("%-12s %5d %5d %5d %5d %5d %5d | %4d %4d %4d %4d | %8.0f   weighted %.0f" % ("static sum", nv, tot["trans"], tot["pk"], tot["fast"], tot["cndmask"], tot["generic"], tot["lds"],
                                                                                  tot["vmem"], tot["salu"], tot["branch"], totc, wsum))


if __name__ == "__main__":
    main()
