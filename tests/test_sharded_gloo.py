"""CPU: the N > 1 candidate-sharding path with world_size = 2 over gloo (oracle scores the shards)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gaussian_process_optimization_amd.sharded import ShardedCandidates
    from _collective import TorchCollective
    from oracle import cpu_ref as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, Y, Xs = O.synthetic_problem(96, 2, 301, seed=4)
        Xs[200] = Xs[17]  # an exact tie across shards
        gp = O.OracleGP(X, Y, O.Matern52(2, 1.0, 0.4), 1e-2)
        gm = O.OracleGPModel(gp)
        fmin = gm.get_fmin()

        def score_local(Xb, sense):  # stands in for Acquisition.argbest on the HIP path
            a = -O.acq_EI(gm, Xb, 0.01, fmin)[:, 0]
            i = int(np.argmin(a) if sense < 0 else np.argmax(a))
            return i, float(a[i])
        sc = ShardedCandidates(rank, world, TorchCollective(world))
        out = [sc.argbest(Xs, score_local, s) for s in (-1, +1)]
        full = -O.acq_EI(gm, Xs, 0.01, fmin)[:, 0]
        # top-k (anchor_points_generator.py:61 keeps 5 anchors): the best row is duplicated into the OTHER shard, so
        # the cross-shard tie is among the winners and must come out lowest global index first
        Xt = Xs.copy()
        b = int(np.argmin(full))
        twin = b + 151 if b < 150 else b - 150   # shards are rows [0, 151) and [151, 301)
        Xt[twin] = Xt[b]
        fullt = -O.acq_EI(gm, Xt, 0.01, fmin)[:, 0]

        def score_local_topk(Xb, k, sense):  # stands in for Acquisition.topk on the HIP path
            a = -O.acq_EI(gm, Xb, 0.01, fmin)[:, 0]
            o = np.argsort(a if sense < 0 else -a, kind="stable")[:k]
            return o, a[o]
        topk = [sc.topk(Xt, score_local_topk, 5, s) for s in (-1, +1)]
        ref_topk = [np.argsort(fullt, kind="stable")[:5], np.argsort(-fullt, kind="stable")[:5]]
        q.put((rank, out, (int(np.argmin(full)), float(full.min())), (int(np.argmax(full)), float(full.max())),
               [(t[0].tolist(), t[1].tolist()) for t in topk], [r.tolist() for r in ref_topk],
               [fullt[r].tolist() for r in ref_topk], (min(b, twin), max(b, twin))))
    finally:
        dist.destroy_process_group()


def test_sharded_argbest_world2_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, out, ref_min, ref_max, topk, ref_idx, ref_val, tie in res:
        assert out[0][0] == ref_min[0] and out[0][1] == pytest.approx(ref_min[1], rel=1e-12)
        assert out[1][0] == ref_max[0] and out[1][1] == pytest.approx(ref_max[1], rel=1e-12)
        for s in range(2):
            assert topk[s][0] == ref_idx[s]
            assert topk[s][1] == pytest.approx(ref_val[s], rel=1e-12)
        # the duplicated best row and its twin in the other shard lead the list, lower global index first
        assert topk[0][0][:2] == list(tie) and topk[0][1][0] == topk[0][1][1]
    assert res[0][1] == res[1][1] and res[0][4] == res[1][4]  # both ranks agree
