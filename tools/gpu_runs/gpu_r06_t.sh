#!/bin/bash
# round 6, call T (re-run as V after the fixes: sticky HIP error cleared in fail(), the one-device tests ask for 16 hardware queues):
# the one-process peer test in its threaded form, the N-process flows, and the quick subset after rfx_peer.hip's host-side change
# (the two barrier epochs are named before the launches: same launches, same arguments).
set -x
mkdir -p gpurun_out/r06_t
timeout 900 python -m pytest tests -q -m gpu -k "peer" 2>&1 | tail -15 > gpurun_out/r06_t/pytest_peer.log
timeout 900 python -m pytest tests -q -m "gpu and quick" 2>&1 | tail -8 > gpurun_out/r06_t/pytest_quick.log
cat gpurun_out/r06_t/pytest_peer.log gpurun_out/r06_t/pytest_quick.log
