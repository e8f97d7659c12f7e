#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes of devtools/gpu_pmc.sh (FETCH_SIZE and WRITE_SIZE, collected in
separate passes) into HBM bytes per launch for the kernels bench.py reports.

    python profiles/pmc_to_traffic.py gpurun_out/pmc profiles/r02

Corrections per MI355X_MICROARCH.md (HBM / rocprofv3): both counters are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE
is exact for 16-byte streaming stores.  Other access widths are uncalibrated -- treat the numbers as
+-2x evidence of re-read waste, not as exact byte counts.
"""
import collections
import csv
import glob
import json
import os
import sys

NAME_MAP = [
    # training-only instantiations first (substring match, first hit wins)
    # second-generation Winograd kernel <CIN, COUT, OUT_MODE, ...>: mode 2 = training forward, 3 = data gradient, 0 / 1 = inference
    ("conv3x3_wino2_bf16x6_kernel<32, 64, 2", "train_conv2_fwd"), ("conv3x3_wino2_bf16x6_kernel<64, 128, 2", "train_conv3_fwd"),
    ("conv3x3_wino2_bf16x6_kernel<128, 64, 3", "bwd_conv3_dgrad"), ("conv3x3_wino2_bf16x6_kernel<64, 32, 3", "bwd_conv2_dgrad"),
    ("conv3x3_wino2_bf16x6_kernel<32, 64, 0", "conv2_mfma_bn_relu_pool"), ("conv3x3_wino2_bf16x6_kernel<64, 128, 1", "conv3_mfma_bn_relu_pool"),
    ("conv3x3_wino_bf16x6_kernel<128, 64", "bwd_conv3_dgrad"),
    ("conv3x3_wino_bf16x6_kernel<32, 64, 2", "train_conv2_fwd"), ("conv3x3_bf16x6_ns_kernel<64, 128, 2, 2, 2", "train_conv3_fwd"),
    ("conv3x3_bf16x6_ns_kernel<64, 32", "bwd_conv2_dgrad"), ("conv3x3_bf16x6_ns_kernel<128, 64", "bwd_conv3_dgrad"),
    ("conv_wgrad_wino_bf16x6_kernel<32, 64", "bwd_conv2_wgrad"), ("conv_wgrad_wino_bf16x6_kernel<64, 128", "bwd_conv3_wgrad"),
    ("conv_wgrad_bf16x6_kernel<32, 64>", "bwd_conv2_wgrad"), ("conv_wgrad_bf16x6_kernel<64, 128>", "bwd_conv3_wgrad"),
    ("gru_quad_kernel<true", "train_gru"), ("gru_bwd_pair_kernel", "bwd_gru"), ("gru_bwd_pair_k4_kernel", "bwd_gru"), ("gru_bwd_quad_kernel", "bwd_gru"),
    ("gemm_tn2_bf16x6_kernel<true", "bwd_gru_dw"), ("gemm_tn2_bf16x6_kernel<false", "bwd_gru_dx"),
    ("gemm_tn_bf16x6_kernel<true", "bwd_gru_dw"), ("gemm_tn_bf16x6_kernel<false", "bwd_gru_dx"),
    ("bn_bwd_dz_kernel<false>", "bwd_bn2_dz"), ("bn_bwd_dz_kernel<true>", "bwd_bn3_dz"), ("conv1_bwd_kernel", "bwd_conv1"),
    ("feat_utt_kernel", "feat_frames"), ("feat_frames_kernel", "feat_frames"), ("feat_normalise_kernel", "feat_normalise"),
    ("conv1_mfma_bn_relu_pool_kernel", "conv1_bn_relu_pool"), ("conv1_bn_relu_pool_kernel", "conv1_bn_relu_pool"),
    ("conv3x3_wino_bf16x6_kernel<32, 64", "conv2_mfma_bn_relu_pool"), ("conv3x3_bf16x6_ns_kernel<32, 64", "conv2_mfma_bn_relu_pool"), ("conv3x3_bf16x6_ns_kernel<64, 128", "conv3_mfma_bn_relu_pool"),
    ("gemm_nt_bf16x6_v3_kernel", "gemm_ih"), ("gemm_nt_f16x3_kernel", "gemm_ih"),
    ("conv3x3_bf16x6_kernel<32, 64", "conv2_mfma_bn_relu_pool"), ("conv3x3_bf16x6_kernel<64, 128", "conv3_mfma_bn_relu_pool"),
    ("conv3x3_mfma_kernel<32, 64", "conv2_mfma_bn_relu_pool"), ("conv3x3_mfma_kernel<64, 128", "conv3_mfma_bn_relu_pool"),
    ("gemm_nt_bf16x6_kernel", "gemm_ih"), ("gemm_nt_bias_kernel", "gemm_ih"), ("gru_recurrence_kernel", "gru_recurrence"), ("gru_pair_kernel", "gru_recurrence"), ("gru_quad_kernel", "gru_recurrence"),
    ("attention_pool_kernel", "attention_pool_fc_argmax"),
]


def main(src, dst):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "*", "pmc_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            for pat, name in NAME_MAP:
                if pat in r["Kernel_Name"]:
                    vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    break
    out = {}
    if "gemm_ih" in vals:                             # (both layers' launches averaged)
        vals["gemm_ih_l0"] = vals["gemm_ih_l1"] = vals["train_gemm_ih_l0"] = vals["train_gemm_ih_l1"] = vals["gemm_ih"]
    if "gru_recurrence" in vals:                      # one kernel, two launches per step: bench.py reports them per layer
        vals["gru_recurrence_l0"] = vals["gru_recurrence_l1"] = vals["gru_recurrence"]
    for both, names in (("train_gru", ("train_gru_l0", "train_gru_l1")), ("bwd_gru", ("bwd_gru_l0", "bwd_gru_l1"))):
        if both in vals:                              # averages over the two layers' launches
            for n in names:
                vals[n] = vals[both]
    for name, cs in vals.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            fetch = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024 * 2
            write = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024
            out[name] = {"hbm_bytes_per_launch": int(fetch + write), "read_bytes": int(fetch), "write_bytes": int(write),
                         "launches_sampled": len(cs["FETCH_SIZE"])}
    os.makedirs(dst, exist_ok=True)
    with open(os.path.join(dst, "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc", sys.argv[2] if len(sys.argv) > 2 else "profiles/r02")
