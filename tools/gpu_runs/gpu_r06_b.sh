#!/bin/bash
# round 6, call B: issue rates of packed-half min / max, the fp32 three-operand forms, SGPR-pair operands of packed fp32 (tools/microbench/valu_rates3.hip)
O=gpurun_out/r06_b; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 200 tools/microbench/bin/valu_rates3 > $O/valu_rates3.txt 2>&1; cat $O/valu_rates3.txt
