"""Drop-in for /root/reference/scripts/train.py: ``collate_fn``, ``train_epoch``, ``validate``,
``train`` with the reference's signatures, YAML keys and checkpoint format, running the step body
(forward, CE loss, backward, Adam) as hand-written HIP kernels on MI355X, plus single-node
data-parallel training (one process per GPU, utterances sharded, one RCCL all-reduce of the flat
gradient buffer per step) which the reference does not have (it pins CUDA_VISIBLE_DEVICES=0,
train.py:17).

Launch:  python -m sir_amd.scripts.train --config cfg.yaml            (1 GPU)
         python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
                -m sir_amd.scripts.train --config cfg.yaml             (8 GPUs, per-GPU batch_size)
"""
import argparse
import os

import torch
import torch.nn as nn
import yaml
from torch.utils.data import DataLoader
from tqdm import tqdm

MAX_LENGTH = 200


def load_config(config_path):
    with open(config_path, "r") as f:
        return yaml.safe_load(f)


def collate_fn(batch):
    """(mel, label) items -> ([B,64,200] float32, [B] int64); drops ``None``/empty items, trims or
    zero-pads the time axis to 200, returns (None, None) for an empty batch (train.py:49-70)."""
    mel_specs, labels = [], []
    for mel, label in batch:
        if mel is None or mel.shape[0] == 0 or mel.shape[1] == 0:
            continue
        if mel.size(1) > MAX_LENGTH:
            mel = mel[:, :MAX_LENGTH]
        elif mel.size(1) < MAX_LENGTH:
            mel = torch.nn.functional.pad(mel, (0, MAX_LENGTH - mel.size(1)))
        mel_specs.append(mel)
        labels.append(label)
    if not mel_specs:
        return None, None
    return torch.stack(mel_specs), torch.tensor(labels, dtype=torch.long)


def _loss_fn(criterion):
    """The HIP cross-entropy when the criterion is the reference's ``nn.CrossEntropyLoss()``
    (mean reduction, no weights / smoothing, train.py:242); any other criterion is called as is."""
    from sir_amd import train_ops
    if (isinstance(criterion, nn.CrossEntropyLoss) and criterion.reduction == "mean" and criterion.weight is None
            and criterion.label_smoothing == 0.0 and criterion.ignore_index == -100):
        return train_ops.fused_cross_entropy
    return criterion


class HostStager:
    """Host -> device hand-over of the batches a ``DataLoader`` yields, through a small ring of PERSISTENT pinned buffers.

    ``mel.to(device, non_blocking=True)`` (train.py:86) on what the loader hands over is the slow link of the reference-shaped
    route on this stack (devtools/dataloader_probe.py on MI355X / ROCm 7 / torch 2.10): a batch that sits in the workers'
    shared memory copies at ~50 MB/s (``pin_memory=False``: 1 k utt/s), and with ``pin_memory=True`` torch's pinning thread
    allocates a fresh pinned block per batch, which stalls for ~85 ms whenever the GPU is busy (3 k utt/s with ANY kernel in
    flight, 112 k with an idle GPU).  Here a batch is copied by the CPU into one of ``slots`` pinned buffers allocated once
    (13 MB memcpy), sent with an asynchronous copy on the caller's stream, and the buffer is reused after an event says that
    copy is done.  Tensors already on the device pass through."""

    def __init__(self, device, slots=3):
        self.device, self.slots = device, slots
        self.bufs, self.events, self.k = {}, {}, 0

    def __call__(self, t):
        if t.is_cuda:
            return t
        key = (tuple(t.shape), t.dtype)
        if key not in self.bufs:
            self.bufs[key] = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for _ in range(self.slots)]
            self.events[key] = [torch.cuda.Event() for _ in range(self.slots)]
        i = self.k % self.slots
        self.k += 1
        buf, ev = self.bufs[key][i], self.events[key][i]
        ev.synchronize()                                   # (a never-recorded event returns at once)
        buf.copy_(t)
        out = buf.to(self.device, non_blocking=True)
        ev.record()
        return out


def loader_kwargs(num_workers):
    """How this package builds its DataLoaders (the ``hbm_feature_cache: false`` route of ``train()``, ``evaluate()``):
    * ``pin_memory=False`` -- host batches go through ``HostStager``'s persistent pinned ring instead of torch's per-batch pinning;
    * worker processes from a FORK SERVER, kept alive across epochs.  The reference's ``DataLoader(num_workers=8)`` forks its
      workers from the training process (train.py:203-219); on this stack, children forked from a process that has initialised
      HIP slow that process's GPU submissions ~50x for as long as they live (a 2 ms training step takes 70-100 ms: 3 k
      utterances/s; the same loader with fork-server workers: 34 k -- profiles/r04/dataloader_probe.txt).  Fork-server workers
      start from a clean process; the dataset travels to them by pickle once (``persistent_workers``)."""
    kw = dict(num_workers=num_workers, pin_memory=False)
    if num_workers > 0:
        kw.update(multiprocessing_context="forkserver", persistent_workers=True)
    return kw


def train_epoch(model, train_loader, optimizer, criterion, device, scaler=None):
    """One epoch (train.py:72-118); returns the mean of the per-step losses.  ``scaler`` is accepted
    for signature compatibility: the HIP path always computes in fp32 (the parity target is the fp32
    CPU path), so no loss scaling is needed or applied."""
    from sir_amd import ops, train_ops
    model.train()
    loss_fn = _loss_fn(criterion)
    losses = []
    stage = HostStager(device)
    pbar = tqdm(train_loader, desc="Training", disable=_quiet())
    for batch_idx, (mel, label) in enumerate(pbar):
        if mel is None or label is None or mel.size(0) == 0:
            if train_ops.world_size() > 1:
                # the other ranks are entering this step's gradient all-reduce: join it with a zero gradient and apply
                # the same update (a bare `continue`, train.py:82-83, would leave them waiting for ever)
                optimizer.zero_grad(set_to_none=True)
                train_ops.zero_contribution_step(model)
                optimizer.step()
            continue
        mel = stage(mel)                                   # host batches: persistent pinned ring (see HostStager)
        label = stage(label)
        optimizer.zero_grad(set_to_none=True)
        output = model(mel)
        loss = loss_fn(output, label)
        loss.backward()
        optimizer.step()
        losses.append(loss.detach())
        if batch_idx % 10 == 0 and not _quiet():
            pbar.set_postfix({"loss": f"{loss.item():.4f}",
                              "GPU": f"{torch.cuda.memory_allocated() / 1024 ** 2:.1f}MB"})
    mean = torch.stack(losses).mean().item() if losses else 0.0      # one device->host sync per epoch
    ops.check_status()            # ... which is where a timed-out GRU recurrence is reported (on every rank: collective)
    return mean


def train_epoch_waveforms(model, wave_loader, optimizer, criterion, device, t_pad=200, augment=None):
    """``train_epoch`` fed with RAW waveform batches: ``wave_loader`` yields ``(wave [B, L] float32 | int16, lengths int32
    [B] | None, label int64 [B])``; the log-mel features are computed on the GPU (BASELINE configs[2]: fused HIP feature
    extraction + forward/backward + Adam) one batch ahead of the training step on a side stream
    (``sir_amd.pipeline.FeaturePrefetcher``).  Items may carry a fourth element, the lengths as a host list (what
    ``WaveformStore.epoch_batches`` yields), so that augmentation parameters can be drawn without a device sync.
    ``augment(wave_batch_index, batch_size, host_lengths | None) -> dict`` may return the featurizer's on-the-fly
    augmentation arguments (``shift``, ``noise_sigma``, ``noise_seed``, ``time_mask``, ``freq_mask``: scripts/augment.py,
    dataset.py:160-176) for that batch.  Returns the mean of the per-step losses."""
    from sir_amd import ops
    from sir_amd.pipeline import FeaturePrefetcher
    model.train()
    loss_fn = _loss_fn(criterion)
    pre = FeaturePrefetcher(t_pad=t_pad)
    losses, pending = [], []

    def submit(idx, item):
        wave, lengths, label = item[:3]
        host_lengths = item[3] if len(item) > 3 else None
        if wave is None or label is None or wave.size(0) == 0:
            return
        wave = wave.to(device, non_blocking=True)
        lengths = lengths.to(device, non_blocking=True) if lengths is not None else None
        pre.kw = augment(idx, wave.size(0), host_lengths) if augment is not None else {}
        pre.submit(wave, lengths)
        pending.append(label.to(device, non_blocking=True))

    def step():
        mel, label = pre.get(), pending.pop(0)
        optimizer.zero_grad(set_to_none=True)
        loss = loss_fn(model(mel), label)
        loss.backward()
        optimizer.step()
        pre.release()
        losses.append(loss.detach())

    for idx, item in enumerate(tqdm(wave_loader, desc="Training", disable=_quiet())):
        submit(idx, item)
        if len(pending) == 2:                           # batch idx is queued: train on batch idx - 1 beside it
            step()
    while pending:
        step()
    mean = torch.stack(losses).mean().item() if losses else 0.0
    ops.check_status()
    return mean


def make_waveform_augment(config, seed=0, epoch=0, rng=None):
    """The ``augment`` callable of ``train_epoch_waveforms`` for the YAML keys of scripts/train.py: time shift + noise
    (scripts/augment.py:98-135 gating, ``waveform_augment_prob``, default 0.7 as augment.py:98) when
    ``waveform_augment`` is on, and the dataset's SpecAugment (dataset.py:105-106, :160-176, ``augment_prob``) always --
    the cached-feature route applies that one in ``FSCIntentDataset.__getitem__``, the fused route has no dataset."""
    import random
    from sir_amd.scripts import augment as aug
    rng = rng or random.Random((int(seed) << 20) ^ int(epoch))
    wave_aug = bool(config.get("waveform_augment", False))
    wave_prob = float(config.get("waveform_augment_prob", 0.7))
    spec_prob = float(config.get("augment_prob", 0.5))

    def fn(idx, bsz, host_lengths):
        if host_lengths is None:
            raise ValueError("waveform augmentation needs the clip lengths on the host (WaveformStore yields them)")
        kw = {}
        if wave_aug:
            shift, sigma = aug.draw_batch_params(host_lengths, wave_prob, rng)
            kw.update(shift=shift, noise_sigma=sigma, noise_seed=(int(seed) << 40) ^ (int(epoch) << 24) ^ int(idx))
        if spec_prob > 0.0:
            tm, fm = aug.draw_spec_masks([1 + n // 512 for n in host_lengths], spec_prob, rng=rng)
            kw.update(time_mask=tm, freq_mask=fm)
        return kw
    return fn


def validate(model, val_loader, criterion, device, scaler=None):
    """(avg_loss, accuracy) over the loader (train.py:120-155); under data parallelism the counts are
    summed over ranks."""
    from sir_amd import ops, train_ops
    model.eval()
    loss_fn = _loss_fn(criterion)
    losses = []
    correct = torch.zeros((), dtype=torch.int64, device=device)
    total = 0
    stage = HostStager(device)
    with torch.no_grad():
        for mel, label in tqdm(val_loader, desc="Validating", disable=_quiet()):
            if mel is None or label is None or mel.size(0) == 0:
                continue
            mel = stage(mel)
            label = stage(label)
            output, predicted = model.predict(mel)
            losses.append(loss_fn(output, label))
            correct += (predicted == label).sum()
            total += label.size(0)
    counts = torch.tensor([int(correct.item()), total], dtype=torch.int64, device=device)
    ops.check_status()                                # the host has just synchronised: surface a timed-out recurrence
    train_ops.all_reduce_sum_(counts)
    accuracy = counts[0].item() / max(counts[1].item(), 1)
    avg_loss = torch.stack(losses).mean().item() if losses else 0.0
    return avg_loss, accuracy


def _quiet():
    return int(os.environ.get("RANK", "0")) != 0


def train(args, config):
    """Main training function (train.py:164-302): same YAML keys, same best-checkpoint rule
    (bare ``state_dict`` at ``save_path/best_model.pt``), same early stopping."""
    from sir_amd import _native, train_ops  # noqa: F401
    from sir_amd.models.models import CNNAudioGRU
    from sir_amd.optim import FusedAdam
    from sir_amd.scripts.dataset import FSCIntentDataset

    _native.require_hip()
    train_ops.limit_host_threads(reserve=int(config.get("num_workers", 2)))     # (the DataLoader workers get their share of the CPU quota)
    rank, world, local_rank = train_ops.init_distributed()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if rank == 0:
        print(f"Training on: {device} ({torch.cuda.get_device_name(local_rank)}), world size {world}")

    cache_dir = config.get("cache_dir", "data/cached_features")
    use_cache = config.get("use_feature_cache", True)
    # Two YAML keys beyond the reference's, both off by default (= reference behaviour: cached features through
    # DataLoader workers).  `fused_features: true` trains from RAW waveforms staged once in HBM, features computed on
    # the GPU inside the step (BASELINE configs[2]); `waveform_augment: true` (implies fused_features) adds the
    # time-shift / noise augmentation of scripts/augment.py inside the feature kernel (configs[4]).
    # `hbm_feature_cache` (default true): the cached-feature route keeps the split's cache in HBM (sir_amd/feature_store.py) and
    # assembles each batch with one gather launch instead of DataLoader workers + a host -> device copy per step -- same files,
    # same item semantics; false = the reference's DataLoader route, which at batch 256 delivers a fraction of what the training
    # step consumes (bench.py `dropin_epoch`).
    wave_aug = bool(config.get("waveform_augment", False))
    fused = bool(config.get("fused_features", False)) or wave_aug
    hbm_cache = bool(config.get("hbm_feature_cache", True)) and not fused
    train_store = None
    if fused:
        from sir_amd.waveform_store import WaveformStore
        train_store = WaveformStore(args.train_csv, args.label_map, device,
                                    sample_rate=int(config.get("sample_rate", 16000)))
    t_pad = int(config.get("mel_spec_length", MAX_LENGTH))
    if hbm_cache:
        from sir_amd.feature_store import FeatureStore
        train_dataset = FeatureStore(args.train_csv, args.label_map, device, use_cache=use_cache, cache_dir=cache_dir, mel_spec_length=t_pad)
        val_dataset = FeatureStore(args.val_csv, args.label_map, device, use_cache=use_cache, cache_dir=cache_dir, mel_spec_length=t_pad)
    else:
        train_dataset = train_store if fused else \
            FSCIntentDataset(csv_path=args.train_csv, label_map_path=args.label_map, is_training=True,
                             augment_prob=config.get("augment_prob", 0.5), use_cache=use_cache, cache_dir=cache_dir)
        val_dataset = FSCIntentDataset(csv_path=args.val_csv, label_map_path=args.label_map, is_training=False,
                                       use_cache=use_cache, cache_dir=cache_dir)
    if rank == 0:
        print(f"Datasets loaded - Train: {len(train_dataset)}, Val: {len(val_dataset)}")

    bs = config["batch_size"]                      # per-GPU batch; the global batch is bs * world
    nw = config.get("num_workers", 2)
    seed = int(config.get("seed", 0))
    train_sampler = train_ops.ShardSampler(len(train_dataset), rank, world, shuffle=True, seed=seed)
    val_sampler = train_ops.ShardSampler(len(val_dataset), rank, world, shuffle=False, pad=False)
    lkw = loader_kwargs(nw)
    train_loader = None if (fused or hbm_cache) else DataLoader(train_dataset, batch_size=bs, sampler=train_sampler, collate_fn=collate_fn, **lkw)
    val_loader = None if hbm_cache else DataLoader(val_dataset, batch_size=bs * 2, sampler=val_sampler, collate_fn=collate_fn, **lkw)

    model = CNNAudioGRU(num_classes=config.get("num_labels", 31)).to(device)
    train_ops.broadcast_module_(model)             # identical initial weights / BN buffers on every rank
    criterion = nn.CrossEntropyLoss()
    optimizer = FusedAdam(model.parameters(), lr=float(config.get("lr", 0.0003)),
                          weight_decay=float(config.get("weight_decay", 0.0001)))
    if config.get("use_amp", True) and rank == 0:
        print("use_amp requested: the HIP path computes in fp32 (parity with the fp32 CPU path); no GradScaler")

    epochs = config.get("epochs", 20)
    patience = config.get("early_stop_patience", 5)
    best_val_acc = 0
    no_improve_count = 0
    for epoch in range(epochs):
        if rank == 0:
            print(f"\nEpoch {epoch + 1}/{epochs}")
        if fused:
            batches = train_store.epoch_batches(bs, rank, world, shuffle=True, seed=seed, epoch=epoch)
            train_loss = train_epoch_waveforms(model, batches, optimizer, criterion, device,
                                               t_pad=t_pad,
                                               augment=make_waveform_augment(config, seed=seed + 977 * rank, epoch=epoch))
        elif hbm_cache:
            batches = train_dataset.epoch_batches(bs, rank, world, shuffle=True, seed=seed, epoch=epoch,
                                                  augment_prob=float(config.get("augment_prob", 0.5)))
            train_loss = train_epoch(model, batches, optimizer, criterion, device, None)
        else:
            train_sampler.set_epoch(epoch)
            train_loss = train_epoch(model, train_loader, optimizer, criterion, device, None)
        if hbm_cache:
            val_loader = val_dataset.epoch_batches(bs * 2, rank, world, shuffle=False, pad=False)
        val_loss, val_acc = validate(model, val_loader, criterion, device, None)
        if rank == 0:
            print(f"Train loss: {train_loss:.4f}, Val loss: {val_loss:.4f}, Val accuracy: {val_acc:.4f}")
        if val_acc > best_val_acc:
            best_val_acc = val_acc
            no_improve_count = 0
            if rank == 0:
                save_path = config.get("save_path", "checkpoints/")
                os.makedirs(save_path, exist_ok=True)
                torch.save(model.state_dict(), os.path.join(save_path, "best_model.pt"))
                print(f"New best model saved with accuracy: {val_acc:.4f}")
        else:
            no_improve_count += 1
            if rank == 0:
                print(f"No improvement for {no_improve_count} epochs")
        if no_improve_count >= patience:
            if rank == 0:
                print(f"Early stopping after {epoch + 1} epochs")
            break
    if rank == 0:
        print(f"Training completed. Best validation accuracy: {best_val_acc:.4f}")
    train_ops.shutdown_distributed()
    return best_val_acc


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Train intent recognition model")
    parser.add_argument("--config", type=str, default="configs/config.yaml", help="Path to config file")
    parser.add_argument("--train_csv", type=str, default=None, help="Path to training CSV")
    parser.add_argument("--val_csv", type=str, default=None, help="Path to validation CSV")
    parser.add_argument("--label_map", type=str, default="data/processed/label_map.json",
                        help="Path to label map JSON file")
    args = parser.parse_args()
    config = load_config(args.config)
    if args.train_csv is None:
        args.train_csv = config.get("train_csv")
    if args.val_csv is None:
        args.val_csv = config.get("valid_csv")
    train(args, config)
