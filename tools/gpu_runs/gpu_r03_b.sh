#!/bin/bash
# Round 3, GPU call B: select-idiom microbenchmark, per-kernel times + target sha1s of the instruction-diet build (K3 clamp-free LDS
# footprints + v_fma_mix lerps, K2 fused history taps + whole-frame specialisation, K1 centred projection / wave-uniform index guard),
# its parity against the reference GLSL at 480x270 and 1080p under both vUv models.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_runs/gpu_r03_b.sh'
set -x
O=gpurun_out/r03_b; mkdir -p $O
timeout 120 tools/microbench/bin/valu_rates2 > $O/valu_rates2.txt 2>&1; grep -E "cndmask|sel_bits|max_f32|med3_f32|fract|ashr|fma_same|bfi|class|legacy" $O/valu_rates2.txt | head -40
timeout 200 python tools/quick_time.py 3840 2160 20 > $O/quick_time_4k.txt 2>&1; cat $O/quick_time_4k.txt
for so in realism-effects_amd/csrc/variants/librfx_*.so; do
  [ -f "$so" ] || continue
  echo "== $so"
  timeout 200 python tools/quick_time.py --lib $PWD/$so 3840 2160 20 2>&1 | grep -v "^scene" | tee -a $O/variants.txt
done
timeout 200 python tools/gpu_runs/uv_model_check.py > $O/uv_model_check.txt 2>&1; tail -24 $O/uv_model_check.txt
export RFX_TEST_UV_REFERENCE=1
timeout 400 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -s -v -k "configs[1]" --durations=5 > $O/configs1_both_models.txt 2>&1
grep -E "PASSED|FAILED|passed|failed|UNEXPLAINED [1-9]" $O/configs1_both_models.txt | tail -20
