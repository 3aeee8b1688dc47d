#!/usr/bin/env python3
"""Development helper (build container, round 6): how evenly a kernel's VALU instructions spread their VGPR source reads over the register file's
two parity banks, per basic block.

tools/microbench/valu_banks.hip measured on MI355X: a wave64 VALU instruction costs max(its issue cost, ~1.55 cycles x the operand reads the busier of
the two banks — even / odd register index — serves per instruction, averaged over a few consecutive instructions).  A three-VGPR-source fma whose
sources all sit in one parity runs at 4.6 cycles instead of 2.7; balanced streams pay nothing.  This tool prints, per block, the VALU count, the source
reads per bank and a small in-order queue model's cycles with and without the bank limit.

    python tools/vgpr_parity.py x.s <kernel-name-fragment> [--min 20]
"""
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from isa_cost import classify, parse, RATES  # noqa: E402

READ_DST = re.compile(r"^v_(fmac|mac|fmaak|dot2c|pk_fmac)_")
BANK_CYCLES = 1.55


def vgprs(tok):
    tok = tok.strip().lstrip("-|").rstrip("|")
    m = re.match(r"^v(\d+)$", tok)
    if m:
        return [int(m.group(1))]
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def sources(line):
    """VGPR source registers of one VALU instruction line"""
    parts = line.split(None, 1)
    op = parts[0]
    if len(parts) < 2:
        return op, []
    ops = [t for t in re.split(r",\s*", parts[1].split(" op_sel")[0].split(" neg_")[0].split(" dst_sel")[0].split(" quad_perm")[0].split(" row_")[0].split(" bitop3")[0])]
    srcs = []
    start = 1
    if op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        start = 1 if not op.startswith("v_cmp") else (1 if ops and (ops[0].startswith("vcc") or ops[0].startswith("s[")) else 0)
    if READ_DST.match(op):
        srcs += vgprs(ops[0])
    for t in ops[start:]:
        srcs += vgprs(t)
    return op, srcs


def blocks_with_lines(path, frag):
    out, cur, inside = {}, None, False
    for line in open(path):
        s = line.strip()
        if not inside:
            if re.match(r"^[_A-Za-z0-9.$]+:", s) and frag in s.split(":")[0] and not s.startswith("."):
                inside, cur = True, "entry"
                out[cur] = []
            continue
        if s.startswith(".Lfunc_end"):
            break
        if re.match(r"^\.LBB\d+_\d+:", s):
            cur = s.split(":")[0]
            out[cur] = []
            continue
        if s.startswith("v_"):
            out[cur].append(s.split(";")[0].strip())
    return out


def model(instrs, banks):
    """in-order issue: instruction i issues when the issue port is free; its reads occupy each bank for BANK_CYCLES per operand; the issue port may run
    at most `slack` cycles ahead of a bank (operand collection is decoupled by a couple of instructions)"""
    t_issue, bank_free, slack = 0.0, [0.0, 0.0], 6.0
    for op, srcs in instrs:
        cost = RATES[classify(op)]
        if banks:
            n = [sum(1 for r in srcs if r % 2 == p) for p in (0, 1)]
            for p in (0, 1):
                bank_free[p] = max(bank_free[p], t_issue) + BANK_CYCLES * n[p]
            t_issue = max(t_issue + cost, max(bank_free) - slack)
        else:
            t_issue += cost
    return t_issue


if __name__ == "__main__":
    path, frag = sys.argv[1], sys.argv[2]
    mn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 20
    print("%-12s %5s %6s %6s %6s | %8s %8s %6s" % ("block", "valu", "even", "odd", "skew", "no-bank", "bank", "ratio"))
    for name, lines in blocks_with_lines(path, frag).items():
        ins = [sources(l) for l in lines]
        if len(ins) < mn:
            continue
        ev = sum(1 for _, s in ins for r in s if r % 2 == 0)
        od = sum(1 for _, s in ins for r in s if r % 2 == 1)
        a, b = model(ins, False), model(ins, True)
        print("%-12s %5d %6d %6d %6.2f | %8.0f %8.0f %6.3f" % (name, len(ins), ev, od, max(ev, od) / max(1, ev + od), a, b, b / a))
