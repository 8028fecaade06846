#!/bin/bash
# the training step under a few HIP runtime settings (process environment only), same box, back to back
cd "$(dirname "$0")/.."
run() { echo -n "$1: "; env $1 timeout -k 10 200 python tools/step_only.py 2>&1 | tail -1; }
run X=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run GPU_MAX_HW_QUEUES=2
run AMD_SERIALIZE_KERNEL=0
run HIP_GRAPH_MEMPOOL=0
run X=0
