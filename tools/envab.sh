#!/bin/bash
# A/B of HIP runtime knobs on the replayed step (measurement only)
run() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --prof-steps 0 > gpurun_out/envab.log 2>gpurun_out/envab.err; python tools/benchsum.py gpurun_out/envab.log 2>&1 | tail -1; }
run X=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run GPU_FLUSH_ON_EXECUTION=0
run ROC_SYSTEM_SCOPE_SIGNAL=0
run AMD_DIRECT_DISPATCH=1
run X=2
