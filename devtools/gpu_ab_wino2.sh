#!/bin/bash
# A/B of the second-generation Winograd kernel per stage on ONE box: SIR_WINO2 = 0 (old kernels), 7 (all three stages), 6 (conv2 on the
# first generation), two alternating rounds; prints throughput and the conv kernels' HIP-event averages
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for round in 1 2; do
for v in ${VALUES:-0 7 6}; do
  env SIR_WINO2=$v timeout -k 10 300 python bench.py --steps 60 --warmup 10 --repeats 3 --train-steps 30 --no-cpu-baseline --no-augment --no-host-feed --no-dist-leg --sustain-seconds 0 > gpurun_out/abw_$v.json 2> gpurun_out/abw_$v.err || { tail -5 gpurun_out/abw_$v.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/abw_$v.json").read().strip().splitlines()[-1])
k=d["kernels_avg_ms"]; t=d["train"]["kernels_avg_ms"]
print("SIR_WINO2=$v infer", d["value"], "serial", d["single_stream"]["value"], "train", d["train"]["value"], d["train"]["ms_per_step"],
      "| conv2", k["conv2_mfma_bn_relu_pool"], "conv3", k["conv3_mfma_bn_relu_pool"], "| train conv2", t["train_conv2_fwd"], "conv3", t["train_conv3_fwd"], "dgrad3", t["bwd_conv3_dgrad"])
PY
done
done
