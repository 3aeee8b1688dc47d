#!/bin/bash
# round 6, call U: why the one-process peer tests fail when one host thread per context issues the calls (call T) while the single-threaded
# form passed (call R): hypothesis — the process's streams share GPU_MAX_HW_QUEUES (4) hardware queues, and two exchange streams that land
# on one queue serialise the two barrier kernels (bounded poll -> RFX_EDEVICE).  Same tests with 4 (default), 8 and 16 queues.
set -x
mkdir -p gpurun_out/r06_u
for q in default 8 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 600 python -m pytest tests -q -m gpu -k "peer_history_gather_between or node_peer" 2>&1 | tail -6 > gpurun_out/r06_u/pytest_queues_$q.log
  echo "== GPU_MAX_HW_QUEUES=$q"; cat gpurun_out/r06_u/pytest_queues_$q.log
done
