"""Data parallelism for the MI355X node: one process per GPU, gradients summed with RCCL over xGMI.

Replaces the reference's data_parallel.py (ListDataParallel :52 / DDP :56, single-process
DataParallel and stock DDP with a custom ``scatter`` for the nested ``*_list`` labels, :11-50; imported
by nobody).  Here every rank builds / receives only its own shard (``shard_batch`` does the list
chunking of list_scatter :29-38), the model is replicated, BatchNorm statistics stay per rank (the
reference uses plain BatchNorm, never SyncBN), and the whole fp32 gradient (9.05 M parameters =
36 MB) is reduced as a few large flat buckets: xGMI is point-to-point, so few big collectives beat
many per-parameter ones.
"""
import torch
import torch.distributed as dist


def shard_batch(batch, rank, world_size):
    """Contiguous chunk `rank` of a collated batch dict (tensors along dim 0, ``*_list`` by item)."""
    out = {}
    for key, val in batch.items():
        n = len(val)
        chunk = (n - 1) // world_size + 1
        out[key] = val[rank * chunk:(rank + 1) * chunk]
    return out


class FlatGradAllReduce:
    """Views every parameter's ``.grad`` into a handful of contiguous fp32 buckets and all-reduces
    the buckets (sum, then divide by world size).  Buckets follow reverse registration order so the
    first bucket is complete early in backward; ``reduce()`` is called after backward."""

    def __init__(self, module, bucket_mb=16.0, process_group=None):
        self.group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        params = [p for p in module.parameters() if p.requires_grad]
        params.reverse()
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets = []
        cur, cur_n = [], 0
        for p in params:
            if cur and cur_n + p.numel() > cap:
                self.buckets.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self.buckets.append(cur)
        self.flat = []
        for bucket in self.buckets:
            total = sum(p.numel() for p in bucket)
            flat = torch.zeros(total, dtype=torch.float32, device=bucket[0].device)
            off = 0
            for p in bucket:
                p.grad = flat[off:off + p.numel()].view_as(p)  # grads accumulate straight into the bucket
                off += p.numel()
            self.flat.append(flat)

    def zero_grad(self):
        for flat in self.flat:
            flat.zero_()

    def reduce(self):
        if not dist.is_initialized():
            return
        works = [dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                 for flat in self.flat]
        for w, flat in zip(works, self.flat):
            w.wait()
            flat.div_(self.world_size)


def broadcast_module(module, src=0, process_group=None):
    """Make every rank start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)
