#!/bin/bash
# round 6, call G: bounds on what an exact compose fold could save (tools/fold_edge_cost.py), 4K and 8K
O=gpurun_out/r06_g; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/fold_edge_cost.py 3840 2160 > $O/fold_edge_cost.txt 2>&1
timeout 300 python tools/fold_edge_cost.py 7680 4320 >> $O/fold_edge_cost.txt 2>&1
cat $O/fold_edge_cost.txt
