#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r02_g; mkdir -p $OUT
export TMPDIR=/tmp
( time python tools/parity_configs.py --impl hip --size 7680x4320 --steps 40 --refine 5 --it 3 --frames 3 --out $OUT/parity_configs4.txt ) 2>&1 | grep -E "^#|real" | grep -v "^# OpenGL"
( time python tools/parity_configs.py --impl hip --size 1920x1080 --steps 8 --refine 2 --it 0 --frames 2 --out $OUT/parity_configs0.txt ) 2>&1 | grep -E "^#|real" | grep -v "^# OpenGL"
( time python tools/parity_configs.py --impl hip --size 1920x1080 --steps 20 --refine 5 --it 1 --frames 2 --out $OUT/parity_configs1.txt ) 2>&1 | grep -E "^#|real" | grep -v "^# OpenGL"
( time python tools/parity_configs.py --impl hip --size 3840x2160 --steps 20 --refine 5 --it 1 --frames 2 --out $OUT/parity_configs2.txt ) 2>&1 | grep -E "^#|real" | grep -v "^# OpenGL"
