#!/bin/bash
# same box: graph vs eager ms/step of the fp32 cfg3 step under HIP runtime knobs that touch queues / cross-stream signals
for e in "X=1" "ROC_SYSTEM_SCOPE_SIGNAL=0" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=2" "DEBUG_HIP_DYNAMIC_QUEUES=0" "DEBUG_HIP_FORCE_GRAPH_QUEUES=2" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "GPU_STREAMOPS_CP_WAIT=1" "ROC_ACTIVE_WAIT_TIMEOUT=0"; do
  echo "== $e"
  env $e timeout 300 python tools/launch_modes_ab.py --blocks 1 --steps 12 2>&1 | grep "^block"
done
