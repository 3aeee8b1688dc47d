#!/bin/bash
# round 4, call H: K3 LDS row pitch padded (bank mapping) — time + LDS conflict counters; what the bounded history gather moves at 4K
# (interval / rows / column blocks); the whole -m gpu suite with the strict-metric twins, the 16-frame test and the 4K streamed test
mkdir -p gpurun_out/r04_h
cd "$GRAFT_REPO_ROOT"
( timeout 300 bash tools/time_variants.sh ) > gpurun_out/r04_h/variants.txt 2>&1
grep "==\|K3 pass\|^frame" gpurun_out/r04_h/variants.txt
( cd /tmp && export TMPDIR=/tmp
  for lib in realism-effects_amd/csrc/librfx_hip.so realism-effects_amd/csrc/variants/librfx_b_k3pad1.so; do
    timeout 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $GRAFT_REPO_ROOT/gpurun_out/r04_h/pmc_$(basename $lib .so) -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/quick_time.py --lib $GRAFT_REPO_ROOT/$lib 3840 2160 2 K3 > /dev/null 2>&1
    echo "== $lib"; python $GRAFT_REPO_ROOT/tools/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/r04_h/pmc_$(basename $lib .so) | grep k3_tiled
  done ) > gpurun_out/r04_h/k3_lds_counters.txt 2>&1
cat gpurun_out/r04_h/k3_lds_counters.txt
rm -rf gpurun_out/r04_h/pmc_*
timeout 300 python tools/history_rows_report.py 3840x2160 1 > gpurun_out/r04_h/history_rows_4k.txt 2>&1
cut -c1-260 gpurun_out/r04_h/history_rows_4k.txt
timeout 1800 python -m pytest tests -m gpu -x -q -s > gpurun_out/r04_h/pytest_gpu.log 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r04_h/pytest_gpu.log | tail -4
grep "frame .* composed outside" gpurun_out/r04_h/pytest_gpu.log
