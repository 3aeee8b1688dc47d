#!/bin/bash
# round 2, GPU call C: K3 AoS / K4 timings, packed-math microbenchmark, PCIe streaming, one-GPU multi-rank flow, parity reports of every config
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r02_c; mkdir -p $OUT
export TMPDIR=/tmp
bash tools/time_variants.sh > $OUT/variants.log 2>&1
grep -E "^==|K1 ssgi|K2 temporal|K3 pass|K4 comp|K1\+K2|sha1" $OUT/variants.log
tools/microbench/bin/valu_rates > $OUT/valu_rates.txt 2>&1; grep -E "pk_|med3|cvt|cndmask|min3|i24|sin" $OUT/valu_rates.txt | grep "8 waves"
timeout 300 python tools/pcie_rate.py 20 > $OUT/pcie_rate.txt 2>&1; cat $OUT/pcie_rate.txt
rm -f /tmp/multirank.err; timeout 600 bash tools/multirank_one_gpu.sh > $OUT/multirank.txt 2>&1; cat $OUT/multirank.txt; tail -15 /tmp/multirank.err > $OUT/multirank.err; grep -E "Error|error" $OUT/multirank.err | head -5
( time python tools/parity_configs.py --impl hip --size 1920x1080 --steps 8 --refine 2 --it 0 --frames 2 --out $OUT/parity_configs0.txt ) 2>&1 | grep -E "^#|real" | grep -v "^# OpenGL"
( time python tools/parity_configs.py --impl hip --size 1920x1080 --steps 20 --refine 5 --it 1 --frames 2 --out $OUT/parity_configs1.txt ) 2>&1 | grep -E "^#|real" | grep -v "^# OpenGL"
( time python tools/parity_configs.py --impl hip --size 3840x2160 --steps 20 --refine 5 --it 1 --frames 2 --out $OUT/parity_configs2.txt ) 2>&1 | grep -E "^#|real" | grep -v "^# OpenGL"
( time python tools/parity_configs.py --impl hip --size 7680x4320 --steps 40 --refine 5 --it 3 --frames 2 --out $OUT/parity_configs4.txt ) 2>&1 | grep -E "^#|real" | grep -v "^# OpenGL"
