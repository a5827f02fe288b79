"""GPU, two ranks sharing the one device (gloo: RCCL refuses duplicate devices): the bucketed gradient exchange of
the real model -- buckets handed over from the side stream while the backward pass continues -- must leave on every
rank the SUM of the per-rank gradients (the 1/world factor lives in the optimizer's clip coefficient)."""
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rank(rank, world, port, q, variant="plain"):
    import torch.distributed as dist
    from fastspeech2_lightning_amd.config import Stats
    from fastspeech2_lightning_amd.model import FastSpeech2
    from fastspeech2_lightning_amd.parallel import GradSync
    from oracle import cases as C

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    align, gst = variant == "learn_alignment", variant == "gst"
    # gst: BASELINE.json configs[4]'s model family -- multi-speaker (+ multilingual) + GST reference encoder (80 mel bins)
    # (the style-token layer emits 256 dims, so that case is the golden's d = 256, 1-layer configuration)
    config = C.build("e2e_gst_multispeaker_train")[0] if gst else C.small_config(learn_alignment=align)
    model = FastSpeech2(config, Stats(**C.STATS), lang2id=C.LANG2ID if gst else None,
                        speaker2id=C.SPEAKER2ID if gst else None, seed=1)  # same seed: same initial weights on both ranks
    model.postnet.dropout_p = 0.0
    model.train()
    from oracle import fs2_oracle as O
    batches = [O.synthetic_batch(B=3, ts_lo=6, ts_hi=12, n_symbols=C.N_SYMBOLS, n_mels=config.preprocessing.audio.n_mels,
                                 seed=100 + r, dur_hi=9 if gst else 4, learn_alignment=align) for r in range(world)]
    if gst:
        for r, b in enumerate(batches):
            b["speaker_id"] = torch.tensor([r, 1, 0], dtype=torch.int32) % len(C.SPEAKER2ID)
            b["language_id"] = torch.tensor([0, r, 1], dtype=torch.int32) % len(C.LANG2ID)
    # reference: both batches locally, no exchange
    local = []
    for b in batches:
        model.training_step(b)
        torch.cuda.synchronize()
        local.append(model.store.grad.clone())
    want = local[0] + local[1]
    # data parallel: this rank's batch only, buckets exchanged during the backward pass
    sync = GradSync(model.store)
    sync.broadcast_parameters(0)
    model.grad_sync = sync
    # four times: eager, then RECORDED into a launch plan (the bucket hand-offs become the plan's host callbacks, run
    # between its segments under the side stream), then replayed twice -- the exchanged gradient must come out the same
    for _ in range(4):
        model.training_step(batches[rank])
        sync.wait()
        torch.cuda.synchronize()
    assert model.plans.recorded == 1 and model.plans.replayed == 2
    segs = next(iter(model.plans.plans.values())).segments
    # one hand-off per gradient bucket (+ the in-step bad-data probe of a model that learns the alignment)
    assert sum(1 for s in segs if s[2] is not None) == len(sync.by_id) + (1 if align else 0)
    got = model.store.grad
    err = float((got - want).abs().max() / want.abs().max())
    q.put((rank, err, len(sync.ranges)))
    dist.destroy_process_group()


@pytest.mark.parametrize("variant", ["plain", "learn_alignment", "gst"])
def test_bucketed_exchange_world2_on_one_gpu(variant):
    """``learn_alignment``: the aligner's backward and the forward-sum loss run on the side stream and meet the main
    chain at the text embedding's bucket; ``gst``: the style encoder's parameter gradients are produced on the side
    stream beside the encoder's backward -- both must be complete when their bucket is handed to the all-reduce."""
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q, variant)) for r in range(2)]
    for p in procs:
        p.start()
    from fastspeech2_lightning_amd.cli import gather_from_ranks
    res = gather_from_ranks(procs, q, len(procs), timeout=300)  # a rank that raises fails the test in seconds
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, err, nbuckets in res:
        assert nbuckets >= 3
        assert err < 1e-5, (rank, err)


def _rccl_single(port, q):
    import torch.distributed as dist
    from fastspeech2_lightning_amd.config import Stats
    from fastspeech2_lightning_amd.model import FastSpeech2
    from fastspeech2_lightning_amd.parallel import GradSync
    from oracle import cases as C
    from oracle import fs2_oracle as O

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    config = C.small_config(learn_alignment=False)
    model = FastSpeech2(config, Stats(**C.STATS), seed=1)
    model.postnet.dropout_p = 0.0
    model.train()
    batch = O.synthetic_batch(B=3, ts_lo=6, ts_hi=12, n_symbols=C.N_SYMBOLS, n_mels=config.preprocessing.audio.n_mels,
                              seed=5, dur_hi=4)
    model.training_step(batch)
    torch.cuda.synchronize()
    want = model.store.grad.clone()
    sync = GradSync(model.store, force=True)  # one rank, but every bucket goes through RCCL from the side stream
    sync.broadcast_parameters(0)
    model.grad_sync = sync
    # four steps: eager, RECORDED into a launch plan (the RCCL all-reduces are issued from the recording, inside the
    # plan's memory pool and under its dispatch guard), then replayed twice with the hand-offs as host callbacks
    worst = 0.0
    for _ in range(4):
        model.training_step(batch)
        sync.wait()
        torch.cuda.synchronize()
        worst = max(worst, float((model.store.grad - want).abs().max() / want.abs().max()))
    assert model.plans.recorded == 1 and model.plans.replayed == 2, (model.plans.recorded, model.plans.replayed)
    # With RCCL's streams alive the step's side stream must still have a HARDWARE queue of its own: two spin kernels, one per
    # stream, run side by side (with ROCclr's default of 4 hardware queues they shared one and took twice as long; the
    # binding sets GPU_MAX_HW_QUEUES=8 before the runtime starts: fastspeech2_lightning_amd/hip.py)
    side = model.env._lanes[0]
    spin = int(3e-3 * 2.0e9)

    def timed(both):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.cuda._sleep(spin)
        if both:
            side.wait_event(e0)
            with torch.cuda.stream(side):
                torch.cuda._sleep(spin)
            ev = torch.cuda.Event()
            ev.record(side)
            torch.cuda.current_stream().wait_event(ev)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1)
    timed(True)
    ratio = min(timed(True) for _ in range(3)) / min(timed(False) for _ in range(3))
    q.put((worst, ratio))
    dist.destroy_process_group()


def test_rccl_call_path_with_one_rank():
    """backend "nccl" (= RCCL) with a single rank: the asynchronous bucket all-reduces issued from the side stream and
    the wait before the optimizer must leave the gradient unchanged (sum over one rank)."""
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_single, args=(port, q))
    p.start()
    from fastspeech2_lightning_amd.cli import gather_from_ranks
    err, overlap_ratio = gather_from_ranks([p], q, 1, timeout=300)[0]
    p.join(120)
    assert p.exitcode == 0
    assert err < 1e-6, err
    assert overlap_ratio < 1.5, f"main and side stream share a hardware queue beside RCCL (two spins took {overlap_ratio:.2f} x one)"


def test_bench_gpus_2_starts_its_own_ranks_and_reports_the_whole_job():
    """The driver's command, ``python bench.py --gpus 2 ...``, with no launcher environment, on the one-GPU box:
    the two ranks share the device (gloo -- RCCL refuses duplicate devices), every bucket is exchanged, and rank 0's
    line carries the whole job's frames (2 x the per-rank batch)."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    bench = Path(__file__).resolve().parent.parent / "bench.py"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(FS2_BENCH_BACKEND="gloo", FS2_GEMM_TUNE="0")
    r = subprocess.run([sys.executable, str(bench), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4",
                        "--no-roofline", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["config"]["parallelism"] == "dp2"
    assert line["config"]["global_batch"] == 8 and line["value"] > 0
    assert line["config"]["real_frames_per_step"] % 2 == 0  # both ranks' frames (same structure on every rank)


def test_tensor_on_another_device_is_refused_before_any_launch(monkeypatch):
    """launches go to the current device's stream and scratch; a tensor of another GPU must raise on the
    host.  One-GPU box: the 'other' device is simulated by moving the binding's notion of the current device."""
    from fastspeech2_lightning_amd import hip as H
    x = torch.ones(1024, device="cuda:0")
    H.axpby(x, None, 2.0, 0.0)  # fine on its own device
    monkeypatch.setattr(H, "_current_device", lambda: 1)
    with pytest.raises(RuntimeError, match="current device"):
        H.axpby(x, None, 2.0, 0.0)
    with pytest.raises(RuntimeError, match="current device"):
        H.layernorm_fwd(torch.ones(4, 256, device="cuda:0"), torch.ones(256, device="cuda:0"), torch.zeros(256, device="cuda:0"))
