"""CPU: `python bench.py --gpus N` starts its own rank processes (launch_ranks) -- environment handed to the ranks, exit
codes, and that the launching process itself touches neither the HIP library nor torch."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_launcher(tmp_path, child_src, n=3):
    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent(child_src))
    code = textwrap.dedent("""
        import sys, json
        sys.path.insert(0, %r)
        import bench
        rc = bench.launch_ranks(%d, ["--gpus", "%d"], child_cmd=[sys.executable, %r, %r], grace_s=2.0)
        # the launcher made no GPU call: neither the ctypes binding nor torch was imported by it
        bad = [m for m in sys.modules if m == "torch" or m.endswith("._lib") or m.startswith("gaussian_process_optimization_amd")]
        print(json.dumps({"rc": rc, "gpu_modules": bad}))
    """ % (ROOT, n, n, str(child), str(tmp_path)))
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1]), out.stdout


def test_launcher_hands_every_rank_its_environment(tmp_path):
    res, stdout = _run_launcher(tmp_path, """
        import json, os, sys
        keys = ["RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY"]
        rec = {k: os.environ.get(k) for k in keys}
        open(os.path.join(sys.argv[1], "rank%s.json" % rec["RANK"]), "w").write(json.dumps(rec))
        print("line from rank " + rec["RANK"])
    """)
    assert res == {"rc": 0, "gpu_modules": []}
    recs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(3)]
    assert [r["RANK"] for r in recs] == ["0", "1", "2"] and [r["LOCAL_RANK"] for r in recs] == ["0", "1", "2"]
    assert {r["WORLD_SIZE"] for r in recs} == {"3"} and {r["MASTER_ADDR"] for r in recs} == {"127.0.0.1"}
    assert len({r["MASTER_PORT"] for r in recs}) == 1 and int(recs[0]["MASTER_PORT"]) > 0
    assert {r["HSA_ENABLE_IPC_MODE_LEGACY"] for r in recs} == {"0"}
    # only rank 0 owns stdout: the job prints ONE line
    assert "line from rank 0" in stdout and "line from rank 1" not in stdout and "line from rank 2" not in stdout


def test_launcher_returns_a_failing_ranks_code_and_stops_the_others(tmp_path):
    res, _ = _run_launcher(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(3)          # e.g. the ranks disagree on the winner
        time.sleep(60)           # the others would wait at a barrier
    """)
    assert res["rc"] == 3


def test_plain_invocation_with_gpus_gt_1_becomes_the_launcher():
    """`python bench.py --gpus 2` without WORLD_SIZE must not demand a launcher any more: main() hands over to launch_ranks
    before anything GPU-related is imported (here the ranks fail at once -- no GPU in this container -- and the launcher
    reports that as a non-zero exit instead of raising SystemExit with usage text)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--M", "256", "--N", "256"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    from gaussian_process_optimization_amd import _lib
    import ctypes
    n = ctypes.c_int(0)
    have_gpu = _lib.load_library().gp_device_count(ctypes.byref(n)) == 0 and n.value > 0
    assert "launch N>1 with" not in out.stderr
    if not have_gpu:
        assert out.returncode != 0 and "no HIP device visible" in out.stderr


def test_committed_traffic_record_belongs_to_the_current_gemm_source():
    """`roofline.traffic` is carried from committed counter passes (profiles/r04_gemm_traffic.json) and dropped when gemm.hip has
    changed since they were taken: the committed record must match the committed kernel source, or the bench line silently loses
    the field."""
    import hashlib
    sys.path.insert(0, ROOT)
    try:
        import bench
    finally:
        sys.path.pop(0)
    traffic, src = bench.static_traffic(16384, 8, 10000)
    assert traffic is not None and traffic > 1e9, src
    gem = os.path.join(ROOT, "gaussian_process_optimization_amd", "csrc", "gemm.hip")
    assert src["gemm_hip_sha256_16"] == hashlib.sha256(open(gem, "rb").read()).hexdigest()[:16]
    assert src["file"] == "profiles/r04_gemm_traffic.json" and os.path.exists(os.path.join(ROOT, src["file"]))
    # other workloads carry no static figure
    assert bench.static_traffic(8192, 8, 10000) == (None, None)


def test_rendezvous_channel_between_rank_processes(tmp_path):
    """bench.Rendezvous (the torch-free control channel of the multi-rank bench): three rank processes started by
    launch_ranks find each other through the socket file, broadcast rank 0's 128-byte id, all-gather records in rank order,
    pass barriers -- and none of them imports torch."""
    res, _ = _run_launcher(tmp_path, """
        import json, os, sys
        sys.path.insert(0, %r)
        import bench
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        rdv = bench.Rendezvous(rank, world, timeout_s=30)
        uid = rdv.bcast(bytes(range(128)) if rank == 0 else None)
        recs = rdv.allgather({"rank": rank, "v": 10.0 * rank})
        g = rdv.gather(rank * rank)
        rdv.barrier()
        out = {"uid_ok": uid == bytes(range(128)), "order": [r["rank"] for r in recs], "max": max(r["v"] for r in recs),
               "gather": g, "torch": "torch" in sys.modules}
        rdv.barrier()
        rdv.close()
        open(os.path.join(sys.argv[1], "rdv%%d.json" %% rank), "w").write(json.dumps(out))
    """ % ROOT)
    assert res["rc"] == 0
    outs = [json.load(open(tmp_path / ("rdv%d.json" % r))) for r in range(3)]
    for r, o in enumerate(outs):
        assert o["uid_ok"] and o["order"] == [0, 1, 2] and o["max"] == 20.0 and o["torch"] is False
        assert o["gather"] == ([0, 1, 4] if r == 0 else None)


def test_launcher_deadline_and_signal_forwarding(tmp_path):
    """A rank that hangs ends the job at the deadline (exit 124) instead of polling for ever; SIGTERM to the launcher reaches
    the ranks (they do not outlive it)."""
    import signal
    import time
    child = tmp_path / "hang.py"
    child.write_text("import os, sys, time\nopen(os.path.join(sys.argv[1], 'pid%s' % os.environ['RANK']), 'w').write(str(os.getpid()))\ntime.sleep(120)\n")
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.launch_ranks(2, [], child_cmd=[sys.executable, %r, %r], timeout_s=%%s))" % (ROOT, str(child), str(tmp_path)))
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, "-c", code % "2.0"], cwd=ROOT, capture_output=True, text=True, timeout=60)
    assert out.returncode == 124 and time.monotonic() - t0 < 30, (out.returncode, out.stderr)
    for f in ("pid0", "pid1"):
        os.remove(tmp_path / f)
    p = subprocess.Popen([sys.executable, "-c", code % "None"], cwd=ROOT)
    t_end = time.monotonic() + 30
    while not (os.path.exists(tmp_path / "pid0") and os.path.exists(tmp_path / "pid1")):
        assert time.monotonic() < t_end
        time.sleep(0.1)
    time.sleep(0.3)
    pids = [int(open(tmp_path / f).read()) for f in ("pid0", "pid1")]
    p.send_signal(signal.SIGTERM)
    assert p.wait(30) == 128 + signal.SIGTERM
    time.sleep(0.5)
    for pid in pids:
        try:
            os.kill(pid, 0)
            alive = True
        except ProcessLookupError:
            alive = False
        assert not alive


def test_rendezvous_under_torch_distributed_run(tmp_path):
    """The driver starts the N > 1 bench as `python -m torch.distributed.run --nproc-per-node N ... bench.py`: the agent owns
    MASTER_PORT (its own store listens there), the ranks are its children.  bench.Rendezvous must find its peers in exactly that
    set-up -- socket file keyed by MASTER_PORT and the parent's PID -- and the ranks still import no torch themselves."""
    import socket
    child = tmp_path / "rdv_child.py"
    child.write_text(textwrap.dedent("""
        import json, os, sys
        sys.path.insert(0, %r)
        import bench
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        rdv = bench.Rendezvous(rank, world, timeout_s=60)
        uid = rdv.bcast(b"u" * 128 if rank == 0 else None)
        recs = rdv.allgather((rank, os.getppid()))
        rdv.barrier()
        rdv.close()
        open(os.path.join(%r, "t%%d.json" %% rank), "w").write(json.dumps(
            {"uid": uid == b"u" * 128, "recs": recs, "torch": "torch" in sys.modules, "port": os.environ["MASTER_PORT"]}))
    """ % (ROOT, str(tmp_path))))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), str(child)], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = [json.load(open(tmp_path / ("t%d.json" % r))) for r in range(2)]
    for r in recs:
        assert r["uid"] and r["torch"] is False and r["port"] == str(port)
        assert [x[0] for x in r["recs"]] == [0, 1] and r["recs"][0][1] == r["recs"][1][1]   # one parent: the agent


def test_default_workload_is_one_job_at_every_rank_count():
    """`bench.py --gpus N`: the SAME candidate table as the N = 1 run, split into contiguous row blocks that cover it exactly
    once (strong scaling: value = job iterations/s, not multiplied by the rank count)."""
    import numpy as np
    sys.path.insert(0, ROOT)
    try:
        import bench
    finally:
        sys.path.pop(0)
    _, _, table1 = bench.synthetic(64, 8, 10000, cand_seed=1236)
    for world in (1, 2, 3, 4, 8):
        blocks, seen = [], 0
        for rank in range(world):
            lo, hi = bench.shard_bounds(10000, rank, world)
            assert lo == seen and hi > lo
            seen = hi
            blocks.append(bench.synthetic(64, 8, 10000, cand_seed=1236)[2][lo:hi])
        assert seen == 10000 and np.array_equal(np.vstack(blocks), table1)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'scaling, value = "strong", job_rate' in src and "world * job_rate" not in src


def test_rendezvous_place_is_private(tmp_path, monkeypatch):
    """The control channel's socket sits in a directory only this user can enter: the launcher's own mkdtemp directory with
    its random key, or -- under torch.distributed.run -- a per-user 0700 directory that is refused when anybody else can
    enter it."""
    sys.path.insert(0, ROOT)
    try:
        import bench
    finally:
        sys.path.pop(0)
    d = tmp_path / "priv"
    d.mkdir(mode=0o700)
    monkeypatch.setenv("GPHIP_BENCH_RDV_DIR", str(d))
    monkeypatch.setenv("GPHIP_BENCH_RDV_KEY", "k" * 32)
    monkeypatch.setenv("MASTER_PORT", "29512")
    key, path = bench.rendezvous_place()
    assert key == b"k" * 32 and os.path.dirname(path) == str(d) and "29512" in os.path.basename(path)
    os.chmod(d, 0o755)
    with pytest.raises(RuntimeError, match="private"):
        bench.rendezvous_place()
    monkeypatch.delenv("GPHIP_BENCH_RDV_DIR")
    monkeypatch.delenv("GPHIP_BENCH_RDV_KEY")
    monkeypatch.setattr(bench.tempfile, "gettempdir", lambda: str(tmp_path))
    key2, path2 = bench.rendezvous_place()
    st = os.stat(os.path.dirname(path2))
    assert st.st_uid == os.getuid() and (st.st_mode & 0o077) == 0 and len(key2) == 64
