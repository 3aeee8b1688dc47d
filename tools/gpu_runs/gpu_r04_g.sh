#!/bin/bash
# round 4, call G: K2 tile heights under the no-SLP build; what the round did to the bits at 4K (round-3 library vs HEAD, stage by stage); the whole -m gpu suite
mkdir -p gpurun_out/r04_g
cd "$GRAFT_REPO_ROOT"
( timeout 300 bash tools/time_variants.sh ) > gpurun_out/r04_g/variants.txt 2>&1
grep "==\|K2 temporal\|^frame" gpurun_out/r04_g/variants.txt
timeout 600 python tools/diff_libs.py $PWD/realism-effects_amd/csrc/variants/librfx_a_old.so $PWD/realism-effects_amd/csrc/librfx_hip.so 3840x2160 2 > gpurun_out/r04_g/diff_r03_vs_head_4k.txt 2>&1
tail -16 gpurun_out/r04_g/diff_r03_vs_head_4k.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04_g/pytest_gpu.log 2>&1
tail -5 gpurun_out/r04_g/pytest_gpu.log
