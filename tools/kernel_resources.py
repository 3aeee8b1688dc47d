#!/usr/bin/env python3
"""Development helper (build container): registers, spills, scratch and LDS of every kernel in a gfx950 assembly file, from the code-object
metadata hipcc writes (`hipcc -S --cuda-device-only ... -o x.s`).

    python tools/kernel_resources.py x.s [name-fragment]
"""
import re
import sys

text = open(sys.argv[1]).read()
frag = sys.argv[2] if len(sys.argv) > 2 else ""
meta = text[text.rfind("amdhsa.kernels:"):]
for block in re.split(r"\n  - ", meta)[1:]:
    f = dict(re.findall(r"\.(\w+):\s+(\S+)", block))
    name = f.get("name", "?")
    if frag not in name:
        continue
    print("%-90s vgpr %3s agpr %3s sgpr %3s spill %3s scratch %4s lds %6s" % (name[:90], f.get("vgpr_count"), f.get("agpr_count", "-"), f.get("sgpr_count"),
          f.get("vgpr_spill_count"), f.get("private_segment_fixed_size"), f.get("group_segment_fixed_size")))
