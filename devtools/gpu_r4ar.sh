#!/bin/bash
# f16x3 projection GEMM as four fat waves (devtools/kernel_ab/gemm_f16x3_w4_kernel.h) against the product kernel: time and error table, K = 1024 and 512, ragged M
set -e
mkdir -p gpurun_out/r4ar
L=speech-intent-recognizer_amd/lib
{ echo "==== M=6400 K=1024 ===="; timeout -k 10 200 $L/bench_gemm 6400 1024; echo; echo "==== M=6400 K=512 ===="; timeout -k 10 200 $L/bench_gemm 6400 512; echo; echo "==== M=1000 (ragged last tile) K=512 ===="; timeout -k 10 200 $L/bench_gemm 1000 512; } > gpurun_out/r4ar/gemm_w4.txt 2>&1
grep -E "^====|f16x3|four|knock|bf16x6 \(product" gpurun_out/r4ar/gemm_w4.txt
