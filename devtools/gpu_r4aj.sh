#!/bin/bash
# direct f16x3 convolution kernel (conv_direct_f16x3_kernel.h) against the direct bf16x6 kernel: results, timings, knock-outs
set -e
mkdir -p gpurun_out/r4aj
timeout -k 10 240 speech-intent-recognizer_amd/lib/bench_conv direct16 > gpurun_out/r4aj/direct16.txt 2>&1
cat gpurun_out/r4aj/direct16.txt
