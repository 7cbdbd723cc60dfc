"""Candidate-table sharding over the GPUs of one node (one process per GPU).

The acquisition batch of the reference (``acquisition_function(configurations)`` then
``np.argmax`` / ``argsort``, run.py:1240-1241, anchor_points_generator.py:59-61) has no
cross-candidate term (posterior.py:276-295 with full_cov=False), so the candidate rows are
split into contiguous blocks, one per rank; every rank holds a replica of the fitted model
(redundant fit, or one RCCL broadcast of L via gp_comm_bcast_fit), scores its block on its
GPU, reduces a local (best value, global row) pair, and ONE collective -- an all-gather of
16 bytes per rank over xGMI (RCCL inside libgphip, gp_comm_allgather_best) -- lets every rank
take the global best with NumPy's lowest-index tie rule.

``merge_best`` is the pure host part of that step; the collective itself is pluggable (any object with
``allgather_best`` / ``allgather_topk``) so the N > 1 logic is covered by world_size-2 gloo tests on CPU
(tests/test_sharded_gloo.py, whose torch.distributed collective lives under tests/: this package imports no torch).
"""
import numpy as np


def shard_bounds(M, rank, nranks):
    """Contiguous row block [lo, hi) of rank ``rank`` (first M % nranks ranks get one extra row)."""
    base, rem = divmod(int(M), int(nranks))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def merge_best(vals, idxs, sense):
    """Global best of per-rank (value, global index) pairs; ties -> lowest index (np.argmax/argmin).

    Ranks with an empty shard contribute idx < 0 and are ignored.
    """
    vals = np.asarray(vals, dtype=float)
    idxs = np.asarray(idxs, dtype=np.int64)
    ok = idxs >= 0
    if not ok.any():
        raise ValueError("no rank produced a candidate")
    v, i = vals[ok], idxs[ok]
    best = v.max() if sense > 0 else v.min()
    return int(i[v == best].min()), float(best)


def merge_topk(vals, idxs, k, sense):
    """The k best of the gathered per-rank (value, global index) pairs, in order; equal values -> lowest global
    index first (the order ``np.argsort(scores, kind="stable")[:k]`` gives on the unsharded table,
    anchor_points_generator.py:61).  Pairs with idx < 0 (empty slots) are ignored.  Returns (indices[k'], values[k'])
    with k' = min(k, number of valid pairs)."""
    vals = np.asarray(vals, dtype=float).reshape(-1)
    idxs = np.asarray(idxs, dtype=np.int64).reshape(-1)
    ok = idxs >= 0
    v, i = vals[ok], idxs[ok]
    order = np.lexsort((i, v if sense < 0 else -v))[:k]
    return i[order], v[order]


class RcclCollective(object):
    """All-gather of (val, idx) through libgphip's RCCL communicator (GPU ranks)."""

    def __init__(self, handle, nranks):
        self.h, self.nranks = handle, nranks

    def allgather_best(self, val, idx):
        return self.h.comm_allgather_best(val, idx, self.nranks)

    def allgather_topk(self, vals, idxs):
        return self.h.comm_allgather_topk(vals, idxs, self.nranks)


class ShardedCandidates(object):
    """Scores this rank's block of a candidate table and agrees on the global best.

    ``score_local(Xblock, sense) -> (local_idx, value)`` is the device call
    (``Acquisition*.argbest`` on the HIP path); ``collective`` is ``RcclCollective`` above (or a test double with the same two methods).
    """

    def __init__(self, rank, nranks, collective):
        self.rank, self.nranks, self.collective = int(rank), int(nranks), collective

    def argbest(self, X_all, score_local, sense=-1):
        lo, hi = shard_bounds(X_all.shape[0], self.rank, self.nranks)
        if hi > lo:
            li, val = score_local(X_all[lo:hi], sense)
            gi = lo + int(li)
        else:
            gi, val = -1, (-np.inf if sense > 0 else np.inf)
        vals, idxs = self.collective.allgather_best(val, gi)
        return merge_best(vals, idxs, sense)

    def topk(self, X_all, score_local_topk, k, sense=-1):
        """The k best rows of the whole table: every rank contributes the k best of its block
        (``score_local_topk(Xblock, k, sense) -> (local indices[k], values[k])``, -1 marking empty slots:
        ``Acquisition*.topk`` on the HIP path), ONE all-gather of ``nranks * k`` pairs, then the same merge on every
        rank (the "gather 8 x k pairs" variant of anchor_points_generator.py:61)."""
        lo, hi = shard_bounds(X_all.shape[0], self.rank, self.nranks)
        li = np.full(k, -1, dtype=np.int64)
        lv = np.full(k, np.inf if sense < 0 else -np.inf)
        if hi > lo:
            i, v = score_local_topk(X_all[lo:hi], k, sense)
            i = np.asarray(i, dtype=np.int64)
            li[:i.size] = np.where(i >= 0, lo + i, -1)
            lv[:i.size] = v
        vals, idxs = self.collective.allgather_topk(lv, li)
        return merge_topk(vals, idxs, k, sense)
