"""Test plumbing: the (value, index) exchange of sharded.py over ``torch.distributed`` (gloo on CPU), standing in for
RcclCollective where there is one GPU or none.  Not part of the product package (host code there is Python + ctypes only)."""
import numpy as np


class TorchCollective(object):
    """Same exchange over ``torch.distributed`` (gloo on CPU for tests; plumbing only)."""

    def __init__(self, nranks):
        self.nranks = nranks

    def allgather_best(self, val, idx):
        import torch
        import torch.distributed as dist
        v = torch.tensor([float(val)], dtype=torch.float64)
        i = torch.tensor([int(idx)], dtype=torch.int64)
        vs = [torch.zeros(1, dtype=torch.float64) for _ in range(self.nranks)]
        is_ = [torch.zeros(1, dtype=torch.int64) for _ in range(self.nranks)]
        dist.all_gather(vs, v)
        dist.all_gather(is_, i)
        return np.array([t.item() for t in vs]), np.array([t.item() for t in is_], dtype=np.int64)

    def allgather_topk(self, vals, idxs):
        import torch
        import torch.distributed as dist
        v = torch.tensor(np.asarray(vals, dtype=float))
        i = torch.tensor(np.asarray(idxs, dtype=np.int64))
        vs = [torch.zeros_like(v) for _ in range(self.nranks)]
        is_ = [torch.zeros_like(i) for _ in range(self.nranks)]
        dist.all_gather(vs, v)
        dist.all_gather(is_, i)
        return torch.cat(vs).numpy(), torch.cat(is_).numpy()
