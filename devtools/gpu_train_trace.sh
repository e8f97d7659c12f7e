#!/bin/bash
# kernel trace (timeline) of a few training steps: gpurun_out/trt/tr_kernel_trace.csv, analysed by devtools/trace_gaps.py
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trt -o tr -- python3 $R/bench.py --steps 3 --warmup 2 --train-steps ${TRAIN_STEPS:-12} --no-cpu-baseline --streams 1 > $R/gpurun_out/trt.log 2>&1 || { tail -20 $R/gpurun_out/trt.log; exit 1; }
ls $R/gpurun_out/trt
cd $R && python devtools/trace_gaps.py gpurun_out/trt/tr_kernel_trace.csv
