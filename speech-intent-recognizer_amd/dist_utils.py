"""Single-node data-parallel plumbing: one process per GPU, ``torch.distributed`` with the
``nccl`` backend (= RCCL over xGMI on ROCm) on the GPU box and ``gloo`` in CPU tests.

The data path of this project shards by utterance; the only exchange step is the gradient mean of
the training step, done as ONE all-reduce over the flat fp32 gradient buffer (3 261 184 elements,
13 MB) -- see SURVEY.md section 8(e).  Inference and feature extraction use no collective.
"""
import os

import torch
import torch.distributed as dist
from torch.utils.data import Sampler


def host_cpu_share(cap=None):
    """CPUs this process may really use: min(affinity mask, cgroup CPU quota[, cap])."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, min(n, cap) if cap else n)


def limit_host_threads(reserve=0):
    """Cap torch's intra-op (OpenMP) thread pool at the CPUs the cgroup grants.  A GPU box shows every host core (256) but grants a
    share of them (16 per GPU): with the default pool every CPU-side tensor op of the main process (a 13 MB batch copy, a stack)
    wakes 256 spinning threads, the cgroup throttles the whole process tree, and the kernel-launching thread crawls (measured:
    61 ms per training step instead of 1.3 with DataLoader workers beside it, profiles/r04/dataloader_probe.txt).  The reference's
    orchestrator exports OMP_NUM_THREADS=1 for the same reason (run_pipeline.py:42); an explicit OMP_NUM_THREADS is respected."""
    if os.environ.get("OMP_NUM_THREADS"):
        return torch.get_num_threads()
    n = max(1, host_cpu_share() - reserve)
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_distributed(backend=None):
    """(rank, world, local_rank); initialises the default process group when WORLD_SIZE > 1."""
    rank, world, local_rank = env_rank()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def shutdown_distributed():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def all_reduce_sum_(t):
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_reduce_mean_(flat):
    """Gradient exchange: sum over ranks then scale by 1/world (in place on the flat buffer)."""
    w = world_size()
    if w > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / w)
    return flat


def broadcast_module_(module, src=0):
    """Parameters and buffers (BN running stats) of rank ``src`` to every rank."""
    if world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)
        from . import ops                       # writes through .data do not bump torch's version counters
        ops.bump_weights_epoch()
    return module


class ShardSampler(Sampler):
    """DistributedSampler semantics without the import-time dependency on an initialised group:
    rank r takes indices i = r (mod world) of each (optionally shuffled) epoch permutation.  With
    ``pad`` the permutation is padded by wrap-around to a multiple of ``world`` so that every rank
    runs the same number of steps (needed when a collective follows each step)."""

    def __init__(self, n, rank=0, world=1, shuffle=True, seed=0, pad=True):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.pad = n, rank, world, shuffle, seed, pad
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _indices(self):
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        if self.pad and self.world > 1 and len(idx) % self.world:
            idx += idx[: self.world - len(idx) % self.world]
        return idx[self.rank:: self.world]

    def __iter__(self):
        return iter(self._indices())

    def __len__(self):
        return len(self._indices())
