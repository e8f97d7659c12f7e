#!/bin/bash
# MFMA accumulator-chain probe
set -e
mkdir -p gpurun_out/r4aq
timeout -k 10 120 devtools/ubench/mfma_chain > gpurun_out/r4aq/chain.txt 2>&1
cat gpurun_out/r4aq/chain.txt
