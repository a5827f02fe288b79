"""CPU (gloo, world_size 2): the bucketed gradient exchange of parallel.GradSync over a flat
gradient buffer -- bucket ranges tile the buffer, every bucket is summed across ranks exactly
once, parameters/buffers are broadcast from rank 0, and the 1/world factor is exposed for the
fused optimizer's clip coefficient."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fastspeech2_lightning_amd import params as P
from fastspeech2_lightning_amd.parallel import GradSync


def make_store():
    S = P.ParamStore()
    S.add("a.weight", (7, 5), "id", P.init_normal)
    S.add("a.bias", (7,), "id", P.init_zeros)
    S.next_bucket()
    S.add("b.conv.weight", (6, 4, 3), "convk", P.init_normal)
    S.add_buffer("b.running_mean", torch.zeros(6))
    S.next_bucket()
    S.add("c.dw.weight", (8, 1, 9), "dw", P.init_normal)
    S.next_bucket()
    S.add("d.weight", (3, 3), "id", P.init_normal)
    return S


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S = make_store().finalize("cpu", seed=100 + rank)  # different weights per rank before the broadcast
        S.buffers["b.running_mean"].fill_(float(rank + 1))
        sync = GradSync(S)
        sync.broadcast_parameters(0)
        S.grad.copy_(torch.arange(S.total, dtype=torch.float32) * (rank + 1))
        for b in (3, 2, 1, 0):  # the backward pass completes buckets from the last to the first
            sync.bucket_ready(b)
        sync.wait()
        buf_after_start = S.buffers["b.running_mean"].clone()
        # ranks then keep their own running statistics (per-rank batches) until an evaluation / checkpoint: DDP's
        # broadcast_buffers semantics = rank 0's buffers everywhere at those points
        S.buffers["b.running_mean"].add_(10.0 * (rank + 1))
        sync.broadcast_buffers(0)
        out[rank] = dict(flat=S.flat.clone(), grad=S.grad.clone(), buf=buf_after_start,
                         buf_eval=S.buffers["b.running_mean"].clone(), scale=sync.grad_scale, ranges=sync.ranges)
    finally:
        dist.destroy_process_group()


def test_bucket_ranges_tile_the_buffer():
    S = make_store().finalize("cpu", seed=0)
    r = S.bucket_ranges()
    assert len(r) == 4 and r[0][0] == 0 and r[-1][1] == S.total
    for (s0, e0), (s1, e1) in zip(r, r[1:]):
        assert e0 == s1 and s0 < e0
    assert all(s % S.ALIGN == 0 for s, _ in r)


def test_state_dict_layout_round_trip_cpu():
    S = make_store().finalize("cpu", seed=1)
    sd = S.state_dict()
    assert sd["b.conv.weight"].shape == (6, 4, 3) and S.p("b.conv.weight").shape == (3, 6, 4)
    assert sd["c.dw.weight"].shape == (8, 1, 9) and S.p("c.dw.weight").shape == (9, 8)
    assert torch.equal(S.p("b.conv.weight")[2, 5, 1], sd["b.conv.weight"][5, 1, 2])
    S2 = make_store().finalize("cpu", seed=2)
    S2.load_state_dict(sd)
    assert torch.equal(S2.flat, S.flat)
    with pytest.raises(RuntimeError):
        S2.load_state_dict({k: v for k, v in sd.items() if k != "a.bias"})


def test_gradsync_world2_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    total = r0["grad"].numel()
    expect = torch.arange(total, dtype=torch.float32) * 3  # (1 + 2) x arange: each element summed exactly once
    assert torch.equal(r0["grad"], expect) and torch.equal(r1["grad"], expect)
    assert torch.equal(r0["flat"], r1["flat"])  # rank 0's weights everywhere
    assert float(r1["buf"][0]) == 1.0  # BatchNorm buffers too
    assert float(r0["buf_eval"][0]) == 11.0 and float(r1["buf_eval"][0]) == 11.0  # rank 0's statistics at evaluation time
    assert r0["scale"] == 0.5


def test_host_budget_divides_the_cores_between_ranks():
    """a rank's intra-op threads and DataLoader workers are its SHARE of the host, not all of it."""
    from fastspeech2_lightning_amd.parallel import apply_host_budget, host_budget
    cores = len(os.sched_getaffinity(0))
    one, eight = host_budget(1), host_budget(8)
    assert one["threads"] == cores and eight["threads"] == max(1, cores // 8)
    assert eight["workers"] == max(0, cores // 8 - 2) and host_budget(10 ** 6)["threads"] == 1
    before = torch.get_num_threads()
    try:
        assert apply_host_budget(4)["threads"] == torch.get_num_threads() == max(1, cores // 4)
    finally:
        torch.set_num_threads(before)


def test_weak_scaling_batches_share_the_structure_and_differ_in_content():
    """bench.py --gpus N: every rank's synthetic batch has rank 0's lengths and durations (same padded shapes = same
    work per GPU) and its own token ids / mel / pitch / energy."""
    from fastspeech2_lightning_amd.synthetic import synthetic_batch
    kw = dict(B=8, ts_lo=20, ts_hi=40, n_symbols=64, n_mels=80, dur_hi=9)
    a = synthetic_batch(seed=1234, **kw)
    again = synthetic_batch(seed=1234, content_seed=None, **kw)
    b = synthetic_batch(seed=1234, content_seed=1237, **kw)
    assert all(torch.equal(a[k], again[k]) for k in ("text", "mel", "duration", "pitch", "energy"))
    for k in ("src_lens", "mel_lens", "duration"):
        assert torch.equal(a[k], b[k]), k
    assert a["mel"].shape == b["mel"].shape and a["text"].shape == b["text"].shape
    assert not torch.equal(a["text"], b["text"]) and not torch.equal(a["mel"], b["mel"])
    assert torch.equal(a["text"] != 0, b["text"] != 0)  # padding positions are structure too


def test_gradsync_bucket_ids_are_ids_not_positions():
    """A bucket id that holds no parameter (an optional block that was not built) must not shift the slices of the
    later ids; an unknown id raises; a step that forgets a bucket raises at ``wait()`` instead of silently keeping a
    local gradient."""
    S = P.ParamStore()
    S.add("a.weight", (8, 4), "id", P.init_normal)
    S.next_bucket()          # bucket 1: nothing declared (e.g. no GST / speaker block)
    S.next_bucket()
    S.add("c.weight", (16,), "id", P.init_normal)
    S.finalize("cpu", seed=0)
    m = S.bucket_map()
    assert set(m) == {0, 1, 2} and m[1] is None and m[0][0] == 0 and m[2][1] == S.total and m[0][1] == m[2][0]
    sync = GradSync(S, world_size=1)
    sync.bucket_ready(2)
    sync.bucket_ready(1)
    with pytest.raises(KeyError):
        sync.bucket_ready(7)
    with pytest.raises(RuntimeError, match=r"\[0\]"):
        sync.wait()          # bucket 0 was never handed over
    sync._done.clear()
    for b in (2, 1, 0):
        sync.bucket_ready(b)
    with pytest.raises(RuntimeError, match="twice"):
        sync.bucket_ready(0)
    sync.wait()


def test_bench_launches_its_own_ranks_for_gpus_n():
    """``python bench.py --gpus 2`` with no launcher environment: the parent must start the two ranks itself (before
    any GPU call) and relay rank 0's JSON line.  ``--dry-run`` stops after rendezvous + a collective, so this runs on
    the CPU box; ``tests/test_ddp_gpu.py`` runs the same entry with the model on the GPU box."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    bench = Path(__file__).resolve().parent.parent / "bench.py"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["FS2_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, str(bench), "--gpus", "2", "--dry-run", "--steps", "5"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line == {"dry_run": True, "n_gpus": 2, "steps": 5, "warmup": 3}


def _tiles_rank(rank, world, port, q):
    import torch.distributed as dist
    from fastspeech2_lightning_amd import hip as H
    from fastspeech2_lightning_amd.parallel import share_tile_table
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    H._TILE_CACHE.clear()
    key = (20736, 1024, 256, 1, 1, 1, 0, 1, 1, 0, 0, 17, 0)
    H._TILE_CACHE[key] = 7 if rank == 0 else 12          # each rank "tuned" its own tile ...
    if rank == 1:
        H._TILE_CACHE[(4096, 256, 256, 1, 1, 1, 0, 1, 0, 0, 0, 17, 0)] = 9
    H._GROUP_TILE_CACHE.clear()
    if rank == 0:  # ... and rank 0 the tile of a grouped launch (three members of one signature)
        H._GROUP_TILE_CACHE[(key, key, key)] = 8
    n = share_tile_table(0)
    assert H._GROUP_TILE_CACHE == {(key, key, key): 8}, (rank, H._GROUP_TILE_CACHE)
    q.put((rank, n, H._TILE_CACHE[key]))
    dist.destroy_process_group()


def test_every_rank_adopts_rank_zero_tile_table():
    """Data parallel: the same GEMM tiles -- hence the same summation order -- on every rank."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tiles_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, 1, 7), (1, 1, 7)]


def test_a_failing_rank_stops_the_others():
    """``--devices N`` / ``bench.py --gpus N`` poll every rank; the first non-zero exit terminates the
    siblings (which would otherwise sit in a collective until the process-group timeout) and is what is reported."""
    import subprocess
    import sys
    import time
    from fastspeech2_lightning_amd.cli import wait_ranks
    procs = [subprocess.Popen([sys.executable, "-c", "import time; time.sleep(60)"]),
             subprocess.Popen([sys.executable, "-c", "import sys, time; time.sleep(0.5); sys.exit(3)"])]
    t0 = time.time()
    rc = wait_ranks(procs)
    assert rc == 3 and time.time() - t0 < 30
    assert all(p.poll() is not None for p in procs)
    ok = [subprocess.Popen([sys.executable, "-c", "pass"]) for _ in range(2)]
    assert wait_ranks(ok) == 0


def _dies(rank, q):
    import time
    if rank == 1:
        raise ValueError("rank 1 fails at start")
    time.sleep(60)
    q.put(rank)


def _answers(rank, q):
    q.put(rank)


def test_gather_from_ranks_fails_in_seconds_when_a_rank_dies():
    """a multiprocessing rank that raises must fail its parent's wait at once (the GPU test
    harness sat in ``q.get(timeout=300)``), and its sibling must not be left behind."""
    import time
    import torch.multiprocessing as mp
    from fastspeech2_lightning_amd.cli import gather_from_ranks
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dies, args=(r, q)) for r in range(2)]
    for p in procs:
        p.start()
    t0 = time.time()
    with pytest.raises(RuntimeError, match="exited with an error"):
        gather_from_ranks(procs, q, 2, timeout=120)
    assert time.time() - t0 < 45
    assert all(not p.is_alive() for p in procs)
    q2 = ctx.Queue()
    ok = [ctx.Process(target=_answers, args=(r, q2)) for r in range(2)]
    for p in ok:
        p.start()
    assert sorted(gather_from_ranks(ok, q2, 2, timeout=120)) == [0, 1]
    for p in ok:
        p.join(30)


def test_tile_table_of_another_signature_format_is_refused_with_a_warning():
    """a tile table saved with an older signature layout would load and never match a launch."""
    import warnings
    from fastspeech2_lightning_amd import hip as H
    saved = dict(H._TILE_CACHE)
    try:
        H._TILE_CACHE.clear()
        H._TILE_CACHE[(20736, 1024, 256, 1, 1, 1, 0, 1, 1, 0, 0, 17, 0)] = 7
        t = H.tile_table()
        assert t["__format__"] == H.TILE_KEY_FORMAT
        H._TILE_CACHE.clear()
        H.load_tile_table(t)
        assert len(H._TILE_CACHE) == 1
        old = {k: v for k, v in t.items() if k != "__format__"}
        H._TILE_CACHE.clear()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            H.load_tile_table(old)
        assert not H._TILE_CACHE and any("signature format" in str(x.message) for x in w)
    finally:
        H._TILE_CACHE.clear()
        H._TILE_CACHE.update(saved)
