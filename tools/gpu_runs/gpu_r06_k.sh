#!/bin/bash
# round 6, call K: the quick GPU subset with its durations, and the default bench.py run (now with the configs[4] 8K extra) with its wall time
O=gpurun_out/r06_k; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m "gpu and quick" -q -x --durations=0 ) > $O/pytest_quick.log 2>&1; grep -E "passed|failed|error|^real" $O/pytest_quick.log | tail -4
grep -E "^[0-9.]+s (call|setup)" $O/pytest_quick.log | head -12
( time python bench.py ) > $O/bench.json 2> $O/bench.err; tail -4 $O/bench.err; python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r06_k/bench.json") if l.startswith("{")][-1])
print("value", j["value"], "ms_per_step", j["ms_per_step"], "cold", j.get("ms_per_step_cold"), "kernel_ms", j["kernel_ms"], "roofline", j["roofline"]["kernel"], j["roofline"]["frac"])
print("north_star", j["north_star_chain"])
print("configs4", json.dumps(j.get("configs4_8k")))
print("cpu", j.get("cpu_baseline"))
PY
