#!/usr/bin/env python3
"""Per-kernel roofline table from the committed profiles of one round.

    python profiles/roofline_table.py profiles/r03 > profiles/r03/roofline.md

Reads   <dir>/infer_kernel_stats.csv, <dir>/train_kernel_stats.csv   (rocprofv3 --kernel-trace --stats, devtools/gpu_round.sh)
        <dir>/pmc/pmc_summary.csv                                    (rocprofv3 --pmc passes, devtools/gpu_pmc.sh + pmc_summary.py)
and the algorithmic FLOP / byte table of bench.py (SURVEY.md section 8(d) per-utterance figures x 256), and prints for
every kernel of the path: average duration, algorithmic TFLOP/s or TB/s, fraction of the peak that bounds it
(round 4 on, f16x3: 2500 / 3 = 833.3 TFLOP/s; rounds 1-3, bf16x6: 2500 / 6 = 416.7; f32 MFMA 157.3; HBM 8 TB/s --
MI355X_MICROARCH.md; the round is taken from the directory name), matrix-pipe busy
(SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)), LDS bank-conflict ratio (SQ_LDS_BANK_CONFLICT /
SQ_LDS_IDX_ACTIVE), and HBM traffic (2 x FETCH_SIZE + WRITE_SIZE, KiB) over the algorithmic bytes.
"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (FLOPS_PER_UTT, peaks; imports torch but touches no device)

B = bench.BATCH
S = 25
F = bench.FLOPS_PER_UTT
MB = 1e6

# (substring of the kernel name, label, which stats file, bound, algorithmic FLOPs per launch | None, algorithmic HBM bytes per
# launch | None).  First match wins; instantiations that only the training step uses come first.  A kernel name that serves
# two launches per step with different shapes (both GRU layers) carries the MEAN of the two.
A1 = B * 32 * 100 * 32 * 4          # conv1 output (pooled, NHWC)
Z2 = B * 32 * 100 * 64 * 4          # conv2 raw output
A2 = B * 16 * 50 * 64 * 4
Z3 = B * 16 * 50 * 128 * 4
X0 = B * S * 1024 * 4
FEAT = B * 64 * 200 * 4
GI = B * S * 1536 * 4
Y = B * S * 512 * 4
ROWS = [
    # ---- training step ----
    ("conv3x3_wino2_bf16x6_kernel<32, 64, 2", "train conv2 fwd (Winograd 2nd gen, raw out + BN partials)", "train", "mfma6", F["train_conv2_fwd"] * B, A1 + Z2),
    ("conv3x3_wino2_bf16x6_kernel<64, 128, 2", "train conv3 fwd (Winograd 2nd gen, raw out + BN partials)", "train", "mfma6", F["train_conv3_fwd"] * B, A2 + Z3),
    ("conv3x3_wino2_bf16x6_kernel<128, 64, 3", "bwd conv3 dgrad (Winograd 2nd gen)", "train", "mfma6", F["bwd_conv3_dgrad"] * B, Z3 + A2),
    ("conv3x3_wino2_bf16x6_kernel<64, 32, 3", "bwd conv2 dgrad (Winograd 2nd gen)", "train", "mfma6", F["bwd_conv2_dgrad"] * B, Z2 + A1),
    ("conv3x3_wino_bf16x6_kernel<32, 64, 2", "train conv2 fwd (Winograd, raw out + BN partials)", "train", "mfma6", F["train_conv2_fwd"] * B, A1 + Z2),
    ("conv3x3_wino_bf16x6_kernel<64, 128, 2", "train conv3 fwd (Winograd, raw out + BN partials)", "train", "mfma6", F["train_conv3_fwd"] * B, A2 + Z3),
    ("conv3x3_bf16x6_ns_kernel<64, 128, 2, 2, 2", "train conv3 fwd (direct, raw out + BN partials)", "train", "mfma6", F["train_conv3_fwd"] * B, A2 + Z3),
    ("conv3x3_wino_bf16x6_kernel<64, 32", "bwd conv2 dgrad (Winograd)", "train", "mfma6", F["bwd_conv2_dgrad"] * B, Z2 + A1),
    ("conv3x3_wino_bf16x6_kernel<128, 64", "bwd conv3 dgrad (Winograd)", "train", "mfma6", F["bwd_conv3_dgrad"] * B, Z3 + A2),
    ("conv3x3_bf16x6_ns_kernel<64, 32", "bwd conv2 dgrad (direct)", "train", "mfma6", F["bwd_conv2_dgrad"] * B, Z2 + A1),
    ("conv3x3_bf16x6_ns_kernel<128, 64", "bwd conv3 dgrad (direct)", "train", "mfma6", F["bwd_conv3_dgrad"] * B, Z3 + A2),
    ("conv_wgrad_wino_bf16x6_kernel<32, 64", "bwd conv2 wgrad (Winograd)", "train", "mfma6", F["bwd_conv2_wgrad"] * B, A1 + Z2),
    ("conv_wgrad_wino_bf16x6_kernel<64, 128", "bwd conv3 wgrad (Winograd)", "train", "mfma6", F["bwd_conv3_wgrad"] * B, A2 + Z3),
    ("conv_wgrad_bf16x6_kernel<32, 64>", "bwd conv2 wgrad", "train", "mfma6", F["bwd_conv2_wgrad"] * B, A1 + Z2),
    ("conv_wgrad_bf16x6_kernel<64, 128>", "bwd conv3 wgrad", "train", "mfma6", F["bwd_conv3_wgrad"] * B, A2 + Z3),
    ("gru_quad_kernel<true", "train GRU recurrence (l0, l1)", "train", "mfma6", F["train_gru_l0"] * B, GI + Y + B * S * 2048 * 4),
    ("gru_bwd_quad_kernel", "BPTT recurrence on the matrix cores (l1, l0)", "train", "mfma6", F["bwd_gru_l0"] * B, B * S * (2048 + 512 + 512 + 1536 + 1536) * 4),
    ("gru_bwd_pair_k4_kernel", "BPTT recurrence, four-k layout (l1, l0)", "train", "mfma6", F["bwd_gru_l0"] * B, B * S * (2048 + 512 + 512 + 1536 + 1536) * 4),
    ("gru_bwd_pair_kernel", "BPTT recurrence (l1, l0)", "train", "mfma6", F["bwd_gru_l0"] * B, B * S * (2048 + 512 + 512 + 1536 + 1536) * 4),
    ("gru_bwd_quad_kernel", "BPTT recurrence, MFMA cluster (l1, l0)", "train", "mfma6", F["bwd_gru_l0"] * B, B * S * (2048 + 512 + 512 + 1536 + 1536) * 4),
    ("gemm_tn2_bf16x6_kernel<true", "GRU dW = dG^T X (l1, l0: mean)", "train", "mfma6", (F["bwd_gru_dw_l0"] + F["bwd_gru_dw_l1"]) * B // 2,
     (2 * B * S * 1536 * 4 + B * S * (1024 + 512) * 4 // 2 + B * S * 512 * 4)),
    ("gemm_tn2_bf16x6_kernel<false, 0, 64", "GRU dX l1", "train", "mfma6", F["bwd_gru_dx_l1"] * B, GI + Y),
    ("gemm_tn2_bf16x6_kernel<false, 0, 128", "GRU dX (l0; l1 as two K halves: mean of the two launches)", "train", "mfma6",
     (F["bwd_gru_dx_l0"] + F["bwd_gru_dx_l1"]) * B // 2, GI + (X0 + Y) // 2),
    ("gemm_tn_bf16x6_kernel<true", "GRU dW = dG^T X (l1, l0: mean)", "train", "mfma6", (F["bwd_gru_dw_l0"] + F["bwd_gru_dw_l1"]) * B // 2,
     (2 * B * S * 1536 * 4 + B * S * (1024 + 512) * 4 // 2 + B * S * 512 * 4)),
    ("gemm_tn_bf16x6_kernel<false, 64", "GRU dX l1", "train", "mfma6", F["bwd_gru_dx_l1"] * B, GI + Y),
    ("gemm_tn_bf16x6_kernel<false, 128", "GRU dX l0", "train", "mfma6", F["bwd_gru_dx_l0"] * B, GI + X0),
    ("gemm_tn_bf16x6_kernel<false", "GRU dX", "train", "mfma6", (F["bwd_gru_dx_l0"] + F["bwd_gru_dx_l1"]) * B // 2, GI + (X0 + Y) // 2),
    ("bn_bwd_dz_kernel<false>", "bwd BN2 dz", "train", "hbm", None, A2 + 2 * Z2),
    ("bn_bwd_dz_kernel<true>", "bwd BN3 dz", "train", "hbm", None, X0 + 2 * Z3),
    ("bn_relu_pool_kernel", "train BN+ReLU+pool (z2 -> a2, z3 -> x0: mean)", "train", "hbm", None, (Z2 + A2 + Z3 + X0) // 2),
    ("conv1_bwd_kernel", "bwd conv1 (recompute + wgrad)", "train", "hbm", None, FEAT + A1),
    ("conv1_train", "train conv1 fwd", "train", "hbm", None, FEAT + A1),
    ("adam_multi_kernel", "Adam (multi-tensor)", "train", "hbm", None, 3261184 * 7 * 4),
    ("gemm_nt_bf16x6_v3_kernel", "input projections (l0, l1: mean)", "both", "mfma6", (F["gemm_ih_l0"] + F["gemm_ih_l1"]) * B // 2,
     (X0 + Y) * 3 // 4 + GI),
    ("gemm_nt_f16x3_kernel", "input projections, f16x3 (l0, l1: mean)", "both", "mfma6", (F["gemm_ih_l0"] + F["gemm_ih_l1"]) * B // 2,
     (X0 + Y) // 2 + GI),
    # ---- inference ----
    ("conv3x3_wino2_bf16x6_kernel<32, 64, 0", "conv2 + BN + ReLU + pool (Winograd 2nd gen)", "infer", "mfma6", F["conv2_mfma_bn_relu_pool"] * B, A1 + A2),
    ("conv3x3_wino2_bf16x6_kernel<64, 128, 1", "conv3 + BN + ReLU + pool (Winograd 2nd gen)", "infer", "mfma6", F["conv3_mfma_bn_relu_pool"] * B, A2 + X0 + X0 * 3 // 2),
    ("feat_utt_kernel", "feature kernel (waveform -> normalised log-mel)", "infer", "hbm", None, bench.FEATURE_BYTES_PER_UTT * B),
    ("conv1_conv2_fused", "conv1+conv2 fused (Winograd)", "infer", "mfma6", (F["conv1_bn_relu_pool"] + F["conv2_mfma_bn_relu_pool"]) * B, FEAT + A2),
    ("conv1_mfma_bn_relu_pool_kernel", "conv1 + BN + ReLU + pool (f32 MFMA)", "infer", "hbm", None, FEAT + A1),
    ("conv3x3_wino_bf16x6_kernel<32, 64", "conv2 + BN + ReLU + pool (Winograd)", "infer", "mfma6", F["conv2_mfma_bn_relu_pool"] * B, A1 + A2),
    ("conv3x3_wino_bf16x6_kernel<64, 128", "conv3 + BN + ReLU + pool (Winograd)", "infer", "mfma6", F["conv3_mfma_bn_relu_pool"] * B, A2 + X0 + X0 * 3 // 2),
    ("conv3x3_bf16x6_ns_kernel<32, 64", "conv2 + BN + ReLU + pool (direct)", "infer", "mfma6", F["conv2_mfma_bn_relu_pool"] * B, A1 + A2),
    ("conv3x3_bf16x6_ns_kernel<64, 128", "conv3 + BN + ReLU + pool (direct)", "infer", "mfma6", F["conv3_mfma_bn_relu_pool"] * B, A2 + X0 + X0 * 3 // 2),
    ("gru_quad_kernel<false", "GRU recurrence (l0, l1)", "infer", "mfma6", F["gru_recurrence_l0"] * B,
     (bench.GRU_ALGO_BYTES["gru_recurrence_l0"] + bench.GRU_ALGO_BYTES["gru_recurrence_l1"]) // 2),
    ("attention_pool_kernel", "attention pool + fc + argmax", "infer", "hbm", None, Y),
]
PEAK = {"mfma6": bench.PEAK_BF16X6_TFLOPS, "mfma32": bench.PEAK_F32_MFMA_TFLOPS, "hbm": bench.PEAK_HBM_GBS / 1e3}      # "mfma6": set per round in main()
SIMDS = 1024
XCDS = 8              # rocprofv3 sums GRBM_GUI_ACTIVE over the eight XCDs: kernel cycles = value / 8


def read_stats(path):
    if not os.path.exists(path):
        return {}
    out = {}
    for r in csv.DictReader(open(path)):
        out[r["Name"]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]), float(r["Percentage"]))
    return out


def read_pmc(path):
    if not os.path.exists(path):
        return {}
    out = {}
    for r in csv.DictReader(open(path)):
        out[r["kernel"]] = {k: float(v) for k, v in r.items() if k not in ("kernel", "launches") and v not in ("", None)}
    return out


def first(d, pat):
    for k, v in d.items():
        if pat in k:
            return v
    return None


def main(d):
    import re
    m = re.search(r"r(\d+)$", os.path.basename(d.rstrip("/")))
    f16x3 = bool(m) and int(m.group(1)) >= 4
    PEAK["mfma6"] = bench.PEAK_F16X3_TFLOPS if f16x3 else bench.PEAK_BF16X6_TFLOPS
    stats = {"infer": read_stats(os.path.join(d, "infer_kernel_stats.csv")), "train": read_stats(os.path.join(d, "train_kernel_stats.csv"))}
    pmc = read_pmc(os.path.join(d, "pmc", "pmc_summary.csv"))
    print(f"# Roofline table, `{os.path.relpath(d, ROOT)}` (batch {B}, T = 200; generated by `python profiles/roofline_table.py {os.path.relpath(d, ROOT)}`)\n")
    print(("Peaks: f16x3 contraction 833.3 TFLOP/s algorithmic (= 2500 / 3 products per fp32 product; rounds 1-3 ran bf16x6 at 2500 / 6 = 416.7: "
           "the same TFLOP/s was twice the fraction there)" if f16x3 else "Peaks: bf16x6 contraction 416.7 TFLOP/s algorithmic (= 2500 / 6)") +
          ", HBM 8 TB/s.  `busy` = matrix-pipe busy cycles / "
          "(kernel cycles x 1024 SIMDs); `LDS confl` = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; `traffic` = 2 x FETCH_SIZE + "
          "WRITE_SIZE (KiB) per launch, `/algo` = over the algorithmic bytes of the launch.  A blank cell = counter not collected.\n")
    print("| kernel | leg | avg us | share of leg | algorithmic | frac of peak | MFMA busy | LDS confl | traffic MB | /algo |")
    print("|---|---|---:|---:|---:|---:|---:|---:|---:|---:|")
    seen = set()
    for pat, label, leg, bound, flops, nbytes in ROWS:
        for lg in (("infer", "train") if leg == "both" else (leg,)):
            hit = [(k, v) for k, v in stats[lg].items() if pat in k and (lg, k) not in seen]
            if not hit:
                continue
            k, (us, calls, pct) = hit[0]
            seen.add((lg, k))
            if bound == "hbm":
                ach = nbytes / (us * 1e-6) / 1e12
                algo = f"{ach:.2f} TB/s"
            else:
                ach = flops / (us * 1e-6) / 1e12
                algo = f"{ach:.1f} TF"
            frac = ach / PEAK[bound]
            c = first(pmc, pat) or {}
            busy = conf = traffic = ratio = ""
            if c.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None and c.get("GRBM_GUI_ACTIVE"):
                busy = f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / XCDS * SIMDS):.2f}"
            if c.get("SQ_LDS_IDX_ACTIVE"):
                conf = f"{c.get('SQ_LDS_BANK_CONFLICT', 0.0) / c['SQ_LDS_IDX_ACTIVE']:.3f}"
            if c.get("FETCH_SIZE") is not None and c.get("WRITE_SIZE") is not None:
                t = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
                traffic = f"{t / MB:.0f}"
                if nbytes:
                    ratio = f"{t / nbytes:.2f}"
            print(f"| {label} | {lg} | {us:.1f} | {pct:.1f} % | {algo} | {frac:.3f} | {busy} | {conf} | {traffic} | {ratio} |")
    for lg in ("infer", "train"):
        rest = [(k, v) for k, v in stats[lg].items() if (lg, k) not in seen and v[2] >= 0.5]
        if rest:
            print(f"\nOther kernels of the {lg} trace above 0.5 % of its time:\n")
            for k, (us, calls, pct) in sorted(rest, key=lambda kv: -kv[1][2]):
                print(f"- `{k[:90]}`: {us:.1f} us x {calls} ({pct:.1f} %)")


if __name__ == "__main__":
    main(os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03")))
