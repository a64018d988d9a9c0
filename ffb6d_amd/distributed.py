"""Multi-GPU harness: one process per GPU, batch-sharded, no data-path collective.

Every operator on the FFB6D forward path is per-sample in eval mode, so
inference scales by giving every rank its own frames; the only collectives are the control
ones used for timing (barrier + MAX all-reduce of the elapsed time).  `backend="nccl"` is RCCL
on ROCm; the same code runs on `gloo` for the CPU tests (tests/test_distributed_cpu.py).

Training (BASELINE.json config 3) wraps the model in torch DistributedDataParallel -- gradient
all-reduce over RCCL/xGMI, exactly the reference's recipe (train_lm.py:625-628) -- see
`wrap_ddp`.
"""
import os
import time

import numpy as np
import torch


class Group:
    """Thin view of the default process group (or of a single process)."""

    def __init__(self, rank=0, world=1, dist=None, device=None):
        self.rank, self.world, self.dist, self.device = rank, world, dist, device

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value):
        """MAX all-reduce of a python float (the contract's max-over-ranks timing)."""
        if self.dist is None:
            return float(value)
        t = torch.tensor([value], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        t = torch.tensor([value], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def init_from_env(backend="nccl", device=None):
    """Reads RANK / WORLD_SIZE / MASTER_* (torchrun contract); single process when WORLD_SIZE<=1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return Group(0, 1, None, device)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return Group(rank, world, dist, device if backend == "nccl" else None)


def shard_frames(config, batch_per_rank, rank, make_batch, **kw):
    """Rank r gets frames [r*batch, (r+1)*batch) of the synthetic stream of `config`
    (seeds 1000*config + sample): disjoint across ranks, independent of the world size."""
    from . import synth
    frames = [synth.make_frame(synth.frame_seed(config, rank * batch_per_rank + s), **kw)
              for s in range(batch_per_rank)] if make_batch is None else make_batch(rank)
    if make_batch is not None:
        return frames
    return {k: np.stack([f[k] for f in frames], axis=0) for k in frames[0]}


def timed_steps(step, warmup, steps, group, sync=lambda: None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + device sync on both
    sides; returns the MAX over ranks of the elapsed seconds."""
    for _ in range(warmup):
        step(False)
    sync()
    group.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    sync()
    group.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    return group.max_over_ranks(elapsed)


def wrap_ddp(model, device, sync_bn=None):
    """DistributedDataParallel exactly as the reference sets it up (train_lm.py:625-628), with
    RCCL-friendly defaults: gradients as bucket views, 25 MB buckets overlapped with backward.

    BatchNorm: the reference converts every BatchNorm to apex SyncBatchNorm before wrapping (train_lm.py:592), i.e. one
    more all-reduce of the batch statistics per BN layer and direction (~150 small collectives per step); sync_bn=True (the
    default: recipe parity, same training numerics at 8 frames per GPU) swaps in torch.nn.SyncBatchNorm.  sync_bn=False keeps
    per-rank statistics, so that the gradients are the only collective -- BASELINE.json's north_star wording ("RCCL all-reduce
    over xGMI for DDP gradients only"); `bench.py --mode train --local-bn` times that variant and says so in its metric.
    sync_bn=None (default) = True on the RCCL (nccl) backend with GPU tensors, False elsewhere (torch's SyncBatchNorm needs
    GPU tensors and a backend with GPU all_gather: the gloo test path keeps per-rank statistics)."""
    if sync_bn is None:
        import torch.distributed as dist
        sync_bn = device.type == "cuda" and dist.is_initialized() and dist.get_backend() == "nccl"
    from torch.nn.parallel import DistributedDataParallel
    if sync_bn:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    return DistributedDataParallel(model, device_ids=[device.index] if device.type == "cuda" else None,
                                   find_unused_parameters=True, gradient_as_bucket_view=True)


class count_collectives:
    """Context manager: how many collectives does the PYTHON side of a step issue?  (torch.nn.SyncBatchNorm calls torch.distributed from
    Python: one all_gather of the statistics per BatchNorm layer in forward, one all_reduce per layer in backward; DistributedDataParallel's
    bucketed gradient all-reduces are issued by its C++ reducer and are reported separately, from its logging data.)
    `.counts` = {function name: calls}."""
    NAMES = ("all_reduce", "all_gather", "all_gather_into_tensor", "reduce_scatter_tensor", "broadcast", "all_to_all_single", "barrier")

    def __enter__(self):
        import torch.distributed as dist
        self.counts, self._saved = {}, {}
        for name in self.NAMES:
            fn = getattr(dist, name, None)
            if fn is None:
                continue
            self._saved[name] = fn

            def wrapped(*a, _fn=fn, _name=name, **kw):
                self.counts[_name] = self.counts.get(_name, 0) + 1
                return _fn(*a, **kw)
            setattr(dist, name, wrapped)
        return self

    def __exit__(self, *exc):
        import torch.distributed as dist
        for name, fn in self._saved.items():
            setattr(dist, name, fn)
        return False


def ddp_bucket_count(ddp):
    """gradient buckets (= all-reduces per backward) of a DistributedDataParallel wrapper, or None"""
    try:
        data = ddp._get_ddp_logging_data()
        sizes = str(data.get("bucket_sizes", ""))
        return len([x for x in sizes.split(",") if x.strip()]) or None
    except Exception:
        return None

