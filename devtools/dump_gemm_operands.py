#!/usr/bin/env python3
"""Writes the REAL operands of the two GRU input projections as raw float32 files for devtools/kernel_ab/bench_gemm.hip:
one training-mode forward of the product path at batch 256 (synthetic clips, seeded weights, dropout 0.5), then
    A_l0.f32 [6400][1024] = GRU input (conv3 block output, TB_X0)     B_l0.f32 [1536][1024] = [W_ih_l0; W_ih_l0_reverse]
    A_l1.f32 [6400][512]  = dropped-out layer-0 output (TB_Y0D)       B_l1.f32 [1536][512]  = [W_ih_l1; W_ih_l1_reverse]
    bias_l0.f32 / bias_l1.f32 [1536]
Developer tool (run through gpurun): python devtools/dump_gemm_operands.py gpurun_out/ops"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ops"
    os.makedirs(out, exist_ok=True)
    import bench
    from sir_amd import _native, synth, train_ops
    from sir_amd.featurizer import get_featurizer
    from sir_amd.models.models import CNNAudioGRU
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    lib = _native.lib()
    model = CNNAudioGRU(bench.NUM_CLASSES)
    model.load_state_dict(synth.synth_state_dict(bench.NUM_CLASSES, seed=0))
    model = model.to(dev).train()
    fz = get_featurizer()
    bsz, t = 256, bench.T_PAD
    x = fz(bench.device_clips(bsz, bench.CLIP_LEN, 1234, dev), torch.full((bsz,), bench.CLIP_LEN, dtype=torch.int32, device=dev), t_pad=t)
    logits = train_ops.forward_train(model, x)
    torch.cuda.synchronize()
    offs = (C.c_size_t * 48)()
    lib.sir_model_train_workspace_offsets(fz.handle, bsz, t, offs, 48)
    ws = model._sir_train["ws"].buf
    s = t // 8

    def buf(idx, shape):
        n = 1
        for d in shape:
            n *= d
        return ws[offs[idx]: offs[idx] + 4 * n].view(torch.float32).view(shape).cpu()

    a0 = buf(4, (bsz * s, 1024))          # TB_X0
    a1 = buf(9, (bsz * s, 512))           # TB_Y0D
    g = model.gru
    b0 = torch.cat([g.weight_ih_l0, g.weight_ih_l0_reverse]).detach().cpu()
    b1 = torch.cat([g.weight_ih_l1, g.weight_ih_l1_reverse]).detach().cpu()
    c0 = torch.cat([g.bias_ih_l0, g.bias_ih_l0_reverse]).detach().cpu()
    c1 = torch.cat([g.bias_ih_l1, g.bias_ih_l1_reverse]).detach().cpu()
    for name, v in (("A_l0", a0), ("B_l0", b0), ("bias_l0", c0), ("A_l1", a1), ("B_l1", b1), ("bias_l1", c1)):
        v.contiguous().numpy().tofile(os.path.join(out, name + ".f32"))
        av = v.abs()
        print(f"{name}: shape {tuple(v.shape)} max |v| {av.max():.4g} rms {v.pow(2).mean().sqrt():.4g} "
              f"share below 2^-14: {(av[av > 0] < 2.0 ** -14).float().mean():.4f} zeros: {(v == 0).float().mean():.4f}")
    print("logits rms", float(logits.pow(2).mean().sqrt()))


if __name__ == "__main__":
    main()
