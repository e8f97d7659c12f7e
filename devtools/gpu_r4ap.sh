#!/bin/bash
# issue sharing between one VALU wave and two MFMA waves per SIMD (the Winograd kernels' wave mix)
set -e
mkdir -p gpurun_out/r4ap
timeout -k 10 120 devtools/ubench/mfma_valu_mix > gpurun_out/r4ap/mix.txt 2>&1
cat gpurun_out/r4ap/mix.txt
