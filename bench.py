#!/usr/bin/env python
"""Throughput of the FastSpeech2 feature-prediction train step on MI355X.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no launcher environment (WORLD_SIZE unset) this process starts the N ranks itself -- before it makes
any GPU call -- and only relays their output; under ``python -m torch.distributed.run --nproc-per-node N`` each
process is one rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).  One rank per GPU over RCCL.

One "step" = forward + all losses + backward + gradient exchange (N > 1) + clip + fused AdamW on one
synthetic LJSpeech-shaped batch of 32 utterances per GPU (BASELINE.json configs[1]: fp32, batch 32,
~100-128 phonemes, 80 x ~600 mel), dropout ON as in training, inputs resident in HBM before the
timed region.  ``value`` = real (unpadded) mel frames processed per second by the whole job.

The JSON line also carries
  roofline     : the dominant kernel (fp32 MFMA GEMM family) -- algorithmic FLOPs per launch / average
                 launch duration, measured live with HIP events on the launch stream, against the
                 157.3 TFLOP/s fp32 matrix peak;
  cpu_baseline : the CPU oracle (pure-PyTorch restatement of the reference, ``oracle/``) timed on this
                 box's host cores on a bounded sample of the same workload (rank 0, N = 1 only);
  split_fp32, bf16_mixed_b64, learn_alignment : the same step in the other configurations of BASELINE.json / the
                 reference (N = 1, default flags only; >= 20 timed steps each) -- reported beside ``value``, never as it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

# before the HIP runtime initialises (fastspeech2_lightning_amd/hip.py explains): with the default 4 hardware queues the
# step's side stream shares a queue with the main stream once RCCL has made its own streams.  Not for the gloo rehearsal
# (FS2_BENCH_BACKEND=gloo: several ranks on ONE GPU, a test-only set-up): two processes with eight queues each on one
# device turn gloo's host-synchronised exchange from seconds per step into minutes (round 5, tests/test_ddp_gpu.py).
if os.environ.get("FS2_BENCH_BACKEND", "nccl") == "nccl":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
else:  # (set, not defaulted: a parent process that imported the package has exported its 8)
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("FS2_BENCH_HW_QUEUES", "4")

import torch

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense
PEAK_HBM_GBS = 8000.0


def host_cores() -> int:
    """Threads for the CPU baseline: the box's CPU share (16 per GPU on the pool), not the host's total."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("FS2_BENCH_CPU_THREADS", 16))))


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


T_START = time.perf_counter()


def make_config(learn_alignment=False, gst=False):
    from fastspeech2_lightning_amd.config import FastSpeech2Config
    from fastspeech2_lightning_amd.synthetic import default_symbols
    model = dict(learn_alignment=learn_alignment)
    if gst:  # BASELINE.json configs[4] in fp32: multi-speaker + GST reference encoder, mel up to ~1200 frames
        model.update(use_global_style_token_module=True, multispeaker=True)
    return FastSpeech2Config(model=model, text=default_symbols(64))


def cpu_baseline(config, batch, sample_B=32, iters=3):
    """Oracle (CPU port of the reference path) fwd + loss + bwd + AdamW on the first ``sample_B``
    utterances of the benchmark batch."""
    from fastspeech2_lightning_amd.config import Stats
    from fastspeech2_lightning_amd.synthetic import DEFAULT_STATS
    from oracle import fs2_oracle as O  # test infrastructure: used here only as the timed CPU baseline

    torch.set_num_threads(host_cores())
    sub = {}
    src = batch["src_lens"][:sample_B]
    mel = batch["mel_lens"][:sample_B]
    Ts, Tm = int(src.max()), int(mel.max())
    for k, v in batch.items():
        if torch.is_tensor(v) and v.dim() >= 1:
            v = v[:sample_B]
            if k in ("text", "duration", "pitch", "energy"):
                v = v[:, :Ts]
            if k == "mel":
                v = v[:, :Tm]
            sub[k] = v.clone()
        else:
            sub[k] = v
    sub["max_src_len"], sub["max_mel_len"] = Ts, Tm
    n_spk = 16 if config.model.multispeaker else 0  # (the --gst configuration: 16 speakers, Rig.__init__)
    model = O.FastSpeech2Oracle(config, Stats(**DEFAULT_STATS), n_symbols=64, n_speakers=n_spk)
    model.train()
    o = config.training.optimizer
    opt = torch.optim.AdamW(model.parameters(), o.learning_rate, betas=tuple(o.betas), eps=o.eps,
                            weight_decay=o.weight_decay)

    def step():
        opt.zero_grad()
        out = model(sub)
        model.loss(out, sub, 0)["total"].backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()

    step()  # warm-up
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    dt = (time.perf_counter() - t0) / iters
    frames = int(mel.sum())
    return {"value": round(frames / dt, 1), "unit": "mel-frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"first {sample_B} utterances of the benchmark batch ({frames} frames, Ts={Ts}, Tm={Tm}), "
                      f"1 warm-up + {iters} timed fwd+loss+bwd+clip+AdamW steps, fp32, {dt:.2f} s/step"}


def kernel_source_hash() -> str:
    """Hash of the kernel sources: profiles measured with rocprofv3 (HBM traffic) are stamped with it, and a stamp that
    no longer matches the tree means the figure is stale (it is then reported as null, with the reason)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted((REPO / "fastspeech2_lightning_amd" / "csrc").glob("*")):
        if f.suffix in (".hip", ".h"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def spawn_ranks(n: int) -> int:
    """``python bench.py --gpus N`` without a launcher: start N copies of this script, one per GPU, with the rendezvous
    environment torch.distributed.run would give them.  Runs BEFORE this process touches the GPU (it never does) and
    never replaces a running process: children are ordinary subprocesses, rank 0's stdout (the JSON line) is relayed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread while all ranks are polled: a rank that dies must not leave the others
    # (and this process) waiting in a rendezvous or a collective until the process-group timeout
    import threading
    from fastspeech2_lightning_amd.cli import wait_ranks
    buf = []
    t = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    t.start()
    code = wait_ranks(procs)
    t.join(10)
    sys.stdout.write(b"".join(buf).decode())
    sys.stdout.flush()
    return code


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200,
                    help="timed steps (default 200 = ~4 s: long enough for clocks and the driver's activity sampler)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous + collectives only, no model (what the CPU-box test of the N > 1 launch path runs)")
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a hipGraph (single stream: the side stream for weight-gradient work is "
                         "switched off, graph branches run slower than two eager streams)")
    ap.add_argument("--no-graph", action="store_true", help="(default, kept for old command lines) launch eagerly")
    ap.add_argument("--refine", action="store_true", help="(default since round 5; kept for old command lines)")
    ap.add_argument("--no-refine", action="store_true",
                    help="skip the second tuner stage (the next-best GEMM tiles of the 16 heaviest shapes tried inside the "
                         "replayed step: +8-13 s of warm-up per model, measured -1.0 %% fp32 / -1.9 %% bf16-mixed per step)")
    ap.add_argument("--gst", action="store_true",
                    help="BASELINE.json configs[4] shape in fp32: multi-speaker (16) + GST style encoder, mel up to ~1200 frames")
    ap.add_argument("--precision", default="32-true", choices=["32-true", "32-split", "bf16-mixed"],
                    help="bf16-mixed = BASELINE.json configs[2] (use with --batch 64): bf16 MFMA operands, fp32 accumulation / "
                         "parameters / optimizer.  The headline metric is quoted on 32-true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-split-line", "--no-extra-legs", dest="no_extra_legs", action="store_true",
                    help="skip the secondary measurements (32-split, bf16-mixed batch 64, learned alignment)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--learn-alignment", action="store_true",
                    help="reference default config: jointly learned alignment (aligner + MAS + CTC/bin losses)")
    return ap


def config_signature(args) -> dict:
    """What a counter profile (profiles/*_gemm_traffic.json) was measured on: reported only for the same configuration."""
    return {"precision": args.precision, "batch": args.batch, "gst": bool(args.gst),
            "learn_alignment": bool(args.learn_alignment)}


class Rig:
    """One model + resident synthetic batch + optimizer, and the step the benchmark times."""

    def __init__(self, precision, batch_size, learn_alignment=False, gst=False, rank=0, world=1, local=0, force_sync=False):
        from fastspeech2_lightning_amd.config import Stats
        from fastspeech2_lightning_amd.model import FastSpeech2
        from fastspeech2_lightning_amd.parallel import GradSync
        from fastspeech2_lightning_amd.synthetic import DEFAULT_STATS, synthetic_batch

        self.config = make_config(learn_alignment, gst)
        spk = {f"spk{i}": i for i in range(16)} if gst else None
        self.model = FastSpeech2(self.config, Stats(**DEFAULT_STATS), speaker2id=spk, device=f"cuda:{local}", seed=1234,
                                 precision=precision)
        self.model.train()
        self.opt = self.model.configure_optimizers()[0][0]
        self.model.configure_gradient_clipping(self.opt, 1.0, "norm")  # Trainer(gradient_clip_val=1.0), fs2/cli/train.py:38
        # every rank gets the N = 1 batch's structure (lengths and durations: the same padded shapes, i.e. the same work
        # per GPU -- weak scaling) with its own contents; rank 0's batch is exactly the single-GPU one
        self.batch = synthetic_batch(B=batch_size, ts_lo=96, ts_hi=128, n_symbols=64, n_mels=80, seed=1234,
                                     content_seed=None if rank == 0 else 1234 + rank,
                                     dur_hi=18 if gst else 9, learn_alignment=learn_alignment)
        if gst:
            self.batch["speaker_id"] = torch.arange(batch_size, dtype=torch.int32) % 16
        self.sync = None
        if world > 1 or force_sync:
            self.sync = GradSync(self.model.store, force=force_sync)
            self.sync.broadcast_parameters(0)
            self.model.data_parallel(self.sync, rank)
            self.opt.grad_scale = self.sync.grad_scale
        self.dev_batch = self.model.prepare_batch(self.batch)  # inputs resident in HBM before the timed region
        self.frames = int(self.batch["mel_lens"].sum())
        self.padded = int(self.batch["mel"].shape[0] * self.batch["mel"].shape[1])
        self.wait_events = None  # N > 1: (before, after) HIP events around every GradSync.wait() of the timed steps
        self.main_stream = None
        if os.environ.get("FS2_BENCH_MAIN_PRIORITY") == "high":
            self.main_stream = torch.cuda.Stream(device=f"cuda:{local}", priority=torch.cuda.Stream.priority_range()[1])

    def step(self):
        if self.main_stream is not None:  # FS2_BENCH_MAIN_PRIORITY=high (measurement aid): the main chain on a high-priority stream
            with torch.cuda.stream(self.main_stream):
                return self._step()
        return self._step()

    def _step(self):
        with torch.no_grad():  # the native loop: the optimizer reads the flat gradient buffer, no autograd node
            self.model.training_step(self.dev_batch)
        if self.sync:
            if self.wait_events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.sync.wait()
                e1.record()
                self.wait_events.append((e0, e1))
            else:
                self.sync.wait()
        self.opt.step()

    def settle(self, max_steps=6):
        """Untimed steps until the step replays from its recorded launch plan (fastspeech2_lightning_amd/plan.py: the
        first step of a geometry tunes tiles, the next one with an unchanged tile table is recorded, later ones replay).
        Returns the number of extra steps run; 0 with FS2_PLAN=0 or when the step already replays."""
        plans = self.model.plans
        n = 0
        from fastspeech2_lightning_amd import plan as PL
        if not PL.ENABLED:
            return 0
        # every rank must run the SAME number of steps (each holds the bucket all-reduces): with more than one rank the
        # count is fixed -- eager with the shared tile table, recorded, replayed -- instead of "until this rank replays"
        fixed = 3 if self.sync is not None else None
        while n < (fixed or max_steps):
            before = plans.replayed
            self.step()
            n += 1
            if fixed is None and plans.replayed > before:
                break
        torch.cuda.synchronize()
        return n

    def refine(self, log=None, world=1, top=16, candidates=2):
        """Second tuner stage (``hip.refine_tiles_in_step``): the runner-up tiles of the heaviest GEMM shapes tried inside
        the replayed step.  With several ranks every decision is taken on the maximum over ranks of the timed step."""
        from fastspeech2_lightning_amd import hip as H
        from fastspeech2_lightning_amd import plan as PL
        if not (H.GEMM_TUNE and PL.ENABLED):
            return 0.0, 0
        model = self.model

        def eager_step():
            model.plan_enabled = False
            try:
                self.step()
            finally:
                model.plan_enabled = True
        agree = None
        if world > 1:
            import torch.distributed as dist

            def agree(ms):
                t = torch.tensor([ms], device=model.device_, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t)
        return H.refine_tiles_in_step(self.step, rounds=8, log=log, settle=self.settle, count_step=eager_step, agree=agree,
                                      top=int(os.environ.get("FS2_REFINE_TOP", top)),
                                      candidates=int(os.environ.get("FS2_REFINE_CANDIDATES", candidates)))

    def plan_info(self):
        plans = self.model.plans
        p = next(reversed(plans.plans.values()), None) if plans.plans else None
        return {"replayed_steps": plans.replayed, "recorded": plans.recorded, "eager_steps": plans.eager,
                "launches_per_step": p.launches if p else None, "segments": len(p.segments) if p else None}

    def host_cost(self, reps=9):
        """The host's OWN cost of a step: wall time of one step() enqueued into an EMPTY queue (after a device
        synchronisation: nothing to wait for, nothing pushing back), median of ``reps``, in ms.  The enqueue loop of the
        timed region cannot show it -- with a GPU-bound step the launch queue fills, ``hipLaunchKernel`` blocks, and the
        loop's wall time follows the step time whatever the host could do (``host_loop_ms_per_step``)."""
        xs = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            self.step()
            xs.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        xs.sort()
        return xs[len(xs) // 2]

    def timed(self, steps, run=None):
        run = run or self.step
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        self.host_enqueue_s = time.perf_counter() - t0  # the host's share: everything enqueued, nothing waited for
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def step_flops(self):
        """SURVEY.md 8d closed form on the padded shapes, backward = 2 x forward."""
        b = self.batch
        Bq, Ts_p, Tm_p = int(b["text"].shape[0]), int(b["text"].shape[1]), int(b["mel"].shape[1])
        f = 3.0 * (Bq * Ts_p * (4 * (3019264 + 1024 * Ts_p) + 1990656)
                   + Bq * Tm_p * (4 * (3019264 + 1024 * Tm_p) + 40960 + 8683520))
        if self.config.model.learn_alignment:
            f += 3.0 * Bq * (Ts_p * 868352 + Tm_p * (115200 + 240 * Ts_p))
        return f


def gemm_profile(rig, record=True):
    """Live measurement of the dominant kernel family: HIP events around every GEMM launch of one eager step, on the
    launch stream.  Every rank runs the steps (they contain the collectives); ``record`` says who keeps the events.
    A GPU-side spin first, so that the host runs ahead and the events bracket device time only.  The side stream is
    switched off for these two steps: with two streams a GEMM shares the chip with another kernel and its own launch
    duration says nothing about the kernel.  Returns (launch records, event-pair overhead in ms)."""
    from fastspeech2_lightning_amd import hip as H
    model = rig.model
    side_was, model.env.side_enabled = model.env.side_enabled, False
    H.GEMM_PROFILE = [] if record else None
    rig.step()
    torch.cuda.synchronize()
    H.GEMM_PROFILE = [] if record else None
    torch.cuda._sleep(int(0.06 * 2.0e9))
    rig.step()
    torch.cuda.synchronize()
    prof, H.GEMM_PROFILE = H.GEMM_PROFILE, None
    model.env.side_enabled = side_was
    if not record:
        return None, 0.0
    # A HIP event pair costs device time of its own (two marker packets the command processor serialises with the
    # kernel between them).  It is measured in the same run -- pairs with nothing in between, each behind a kernel so
    # that the queue is busy as it is in the step -- and taken off every interval; without it the per-launch
    # durations come out 3-4 us longer than rocprofv3's kernel durations for the same launches.
    pairs = []
    xs = torch.zeros(1 << 20, device=model.device_)
    for _ in range(40):
        H.axpby(xs, None, 1.0, 0.0, out=xs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    ov = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2]  # median, ms
    return prof, ov


def gemm_numbers(prof, ov):
    raw_ms = sum(q[0].elapsed_time(q[1]) for q in prof)
    ms = raw_ms - ov * len(prof)
    flops = sum(q[2] for q in prof)
    nbytes = sum(q[6] for q in prof)
    return {"launches_per_step": len(prof), "avg_launch_us": round(ms * 1e3 / len(prof), 2),
            "flops_per_launch": round(flops / len(prof)), "algorithmic_bytes_per_launch": round(nbytes / len(prof)),
            "gemm_ms_per_step": round(ms, 3), "event_pair_overhead_us": round(ov * 1e3, 2),
            "gemm_ms_per_step_raw_events": round(raw_ms, 3),
            "_tflops": flops / (ms * 1e-3) / 1e12, "_gbs": nbytes / (ms * 1e-3) / 1e9}


def log_gemm_breakdown(prof):
    by = {}
    for e0, e1, fl, mc, nc, r, _nb, kind in prof:
        t, f, n = by.get((mc, nc, r, kind), (0.0, 0.0, 0))
        by[(mc, nc, r, kind)] = (t + e0.elapsed_time(e1), f + fl, n + 1)
    for (mc, nc, r, kind), (t, f, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:48]:
        ak, bk, taps, sh, sk, epi, tile = kind[:7]
        log(f"gemm Mc={mc:6d} Nc={nc:5d} R={r:6d} {'NT'[ak]}{'NT'[bk]} taps={taps} sh={sh} splitk={sk:2d} epi={epi} "
            f"tile={tile} {kind[7:]} x{n:3d}: {t:7.3f} ms  {f / t / 1e9:6.1f} TFLOP/s")


def committed_traffic(sig):
    """HBM-side bytes per GEMM launch come from two separate rocprofv3 --pmc passes over this same command
    (tools/pmc_traffic.py): they cannot be collected from inside the process, so the committed figure carries the hash
    of the kernel sources and the configuration it was measured on, and is only reported while both match."""
    want, tree = (sig if isinstance(sig, dict) else config_signature(sig)), kernel_source_hash()
    reason = "no profiles/*gemm_traffic.json for this configuration"
    for tfile in sorted((REPO / "profiles").glob("r*_gemm_traffic.json"), reverse=True):
        t = json.loads(tfile.read_text())
        cfg = t.get("config")
        if cfg is None:  # files of rounds 1-2 carry no stamp: by their names, the *_bf16_* ones are bf16-mixed batch 64
            cfg = ({"precision": "bf16-mixed", "batch": 64, "gst": False, "learn_alignment": False} if "bf16" in tfile.name
                   else {"precision": "32-true", "batch": 32, "gst": False, "learn_alignment": False})
        if cfg != want:
            continue
        if t.get("kernel_source_hash") == tree:
            return round(t["hbm_bytes_per_launch"]), (f"profiles/{tfile.name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH "
                                                     f"doubled; kernel sources {tree}; {cfg})")
        reason = (f"profiles/{tfile.name} is stale: measured on kernel sources {t.get('kernel_source_hash', 'unstamped')}, "
                  f"tree is {tree}")
        break
    return None, reason


def committed_gemm_trace(sig):
    """GEMM-family kernel time per step from the committed one-stream kernel trace of this configuration
    (profiles/*_gemm_trace.json, tools/gemm_trace_sum.py), while it was measured on the tree's kernel sources."""
    want, tree = (sig if isinstance(sig, dict) else config_signature(sig)), kernel_source_hash()
    for tfile in sorted((REPO / "profiles").glob("r*_gemm_trace.json"), reverse=True):
        t = json.loads(tfile.read_text())
        if t.get("config") == want and t.get("kernel_source_hash") == tree:
            return t, f"profiles/{tfile.name}"
    return None, None


def roofline_of(sig, precision, prof, ov):
    """``sig``: the configuration (argparse namespace or config_signature dict) whose committed counter profile supplies
    ``traffic``; None: no such profile exists for this leg."""
    g = gemm_numbers(prof, ov)
    tf, gbs = g.pop("_tflops"), g.pop("_gbs")
    # which clock: the committed kernel trace of this configuration on these kernel sources when there is one (the
    # profiler's own kernel durations), else the live HIP-event intervals minus the event pair's measured device cost
    trace, trace_file = committed_gemm_trace(sig) if sig is not None else (None, None)
    g["time_source"] = "hip events around every GEMM launch, event-pair overhead subtracted"
    g["achieved_from_events_tflops"] = round(tf, 2)
    if trace is not None:
        scale = g["gemm_ms_per_step"] / trace["gemm_ms_per_step"]
        tf, gbs = tf * scale, gbs * scale
        g["time_source"] = f"{trace_file} (rocprofv3 --kernel-trace, one stream: {trace['gemm_ms_per_step']:.3f} ms of GEMM kernels per step)"
        g["gemm_ms_per_step_trace"] = round(trace["gemm_ms_per_step"], 3)
    if precision == "bf16-mixed":
        # both fractions: operand + result bytes against HBM, flops against the dense bf16 MFMA peak.
        # The ridge is ~310 flop/B; the family's intensity decides which one bounds it.
        intensity = g["flops_per_launch"] / max(g["algorithmic_bytes_per_launch"], 1)
        hbm_bound = intensity < PEAK_BF16_MFMA_TFLOPS * 1e3 / PEAK_HBM_GBS
        r = ({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)}
             if hbm_bound else
             {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
              "frac": round(tf / PEAK_BF16_MFMA_TFLOPS, 4)})
        traffic, source = committed_traffic(sig) if sig is not None else (None, "no counter profile for this leg")
        r.update(traffic=traffic, traffic_source=source, flop_per_byte=round(intensity, 1), achieved_tflops=round(tf, 2), achieved_gbs=round(gbs, 1),
                 frac_of_bf16_mfma_peak=round(tf / PEAK_BF16_MFMA_TFLOPS, 4), frac_of_hbm_peak=round(gbs / PEAK_HBM_GBS, 4),
                 kernel="gemm family on v_mfma_f32_32x32x16_bf16, fp32 accumulate")
        r.update(g)
        return r
    split_main = precision == "32-split"
    # 32-split: six bf16 MFMA products per algorithmic product -> the pipe's ceiling for fp32-accurate flops
    peak = round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1) if split_main else PEAK_FP32_MFMA_TFLOPS
    r = {"bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4)}
    if sig is not None:
        r["traffic"], r["traffic_source"] = committed_traffic(sig)
    else:
        r["traffic"] = None
    r["kernel"] = ("gemm2_kernel / gemm2p_kernel family, 32-split instances (6 x v_mfma_f32_32x32x16_bf16 per 32x32x16 block; "
                   "peak = dense bf16 MFMA peak / 6)" if split_main else
                   "gemm2_kernel / gemm2p_kernel / gemm_kernel family (fp32 v_mfma_f32_32x32x2_f32), tile autotuned per shape")
    r.update(g)
    return r


def extra_leg(name, precision, batch_size, learn_alignment, steps, local, want_roofline, gst=False, refine=True):
    """One of the secondary configurations: own model, own batch, warm-up (tile tuning), >= 20 timed steps."""
    rig = Rig(precision, batch_size, learn_alignment=learn_alignment, gst=gst, local=local)
    for _ in range(3):
        rig.step()
    torch.cuda.synchronize()
    rig.settle()
    if refine:
        t_ref, n_ref = rig.refine(log)
        log(f"{name}: in-step tile refinement: {n_ref} signatures changed, {t_ref:.2f} ms/step")
    dt = rig.timed(steps) / steps
    out = {"precision": precision, "batch_per_gpu": batch_size, "learn_alignment": learn_alignment, "gst_multispeaker": gst,
           "steps": steps,
           "ms_per_step": round(dt * 1e3, 3), "value": round(rig.frames / dt, 1), "unit": "mel-frames/s",
           "real_frames_per_step": rig.frames, "padded_frames_per_step": rig.padded}
    tf = rig.step_flops() / dt / 1e12
    out["whole_step_tflops"] = round(tf, 2)
    out["whole_step_frac_of_mfma_peak"] = round(tf / (PEAK_BF16_MFMA_TFLOPS if precision == "bf16-mixed" else PEAK_FP32_MFMA_TFLOPS), 4)
    if want_roofline:
        prof, ov = gemm_profile(rig)
        if os.environ.get("FS2_BENCH_GEMM_BREAKDOWN"):
            log_gemm_breakdown(prof)
        # (only the un-split precisions have a counter profile of their own: tools/profile_round.sh)
        sig = None if precision == "32-split" else {"precision": precision, "batch": batch_size, "gst": bool(gst),
                                                    "learn_alignment": bool(learn_alignment)}
        out["roofline"] = roofline_of(sig, precision, prof, ov)
    log(f"{name}: {dt * 1e3:.2f} ms/step, {rig.frames / dt:,.0f} mel-frames/s (enqueue loop {rig.host_enqueue_s / steps * 1e3:.2f} ms/step)")
    out["host_loop_ms_per_step"] = round(rig.host_enqueue_s / steps * 1e3, 3)
    out["host_enqueue_ms_per_step"] = round(rig.host_cost(), 3)  # (one step into an empty queue, median of 9)
    out["launch_plan"] = rig.plan_info()
    del rig
    torch.cuda.empty_cache()
    return out


def main():
    args = build_parser().parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # FS2_BENCH_BACKEND=gloo rehearses the N > 1 control flow on a box with fewer GPUs than ranks (ranks then
    # share devices; RCCL itself refuses duplicate devices).  The driver's runs use RCCL ("nccl").
    backend = os.environ.get("FS2_BENCH_BACKEND", "nccl")
    import torch.distributed as dist
    if args.dry_run:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo" if backend != "nccl" or not torch.cuda.is_available() else backend)
            t = torch.tensor([float(rank + 1)])
            dist.all_reduce(t)
            dist.barrier()
            assert float(t) == world * (world + 1) / 2
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup}), flush=True)
        return
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)

    from fastspeech2_lightning_amd import hip as H
    from fastspeech2_lightning_amd.parallel import apply_host_budget, share_tile_table

    host = apply_host_budget(int(os.environ.get("LOCAL_WORLD_SIZE", world)))  # this rank's share of the host's cores
    force_sync = bool(os.environ.get("FS2_BENCH_FORCE_SYNC"))  # one rank, collectives issued anyway (RCCL call path)
    if force_sync and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(f"cuda:{local}"))
    rig = Rig(args.precision, args.batch, args.learn_alignment, args.gst, rank, world, local, force_sync)
    model, frames, padded = rig.model, rig.frames, rig.padded
    step = rig.step
    f_step, n_params = rig.step_flops(), model.store.num_trainable

    def barrier():
        if world > 1:
            dist.barrier()

    log(f"model built ({model.store.num_trainable} parameters), batch resident: {frames} frames, padded {padded}")
    for i in range(max(args.warmup, 1)):
        t_w = time.perf_counter()
        step()
        torch.cuda.synchronize()
        log(f"warm-up step {i}: {(time.perf_counter() - t_w) * 1e3:.1f} ms")
        if i == 0 and world > 1:
            # every rank runs the tiles rank 0 tuned in its first step: same summation order on every rank
            n_sig = share_tile_table(0)
            log(f"tile table of rank 0 adopted ({n_sig} signatures)")

    use_graph = args.graph and not args.no_graph and world == 1
    model.env.side_enabled = model.env.side_enabled and not use_graph  # (FS2_SIDE_STREAM=0 keeps it off)
    # (with several ranks only on request: every rank must take the same trials and decisions -- rig.refine agrees on the
    # maximum over ranks and share_tile_table ships rank 0's candidate lists -- but that path has never met real RCCL ranks)
    if (args.refine or world == 1) and not args.no_refine and not use_graph:
        t_ref, n_ref = rig.refine(log, world, top=32, candidates=3)  # (the headline: ~45 s; the legs take 16 x 2)
        log(f"in-step tile refinement: {n_ref} signatures changed, {t_ref:.2f} ms/step")
        if os.environ.get("FS2_BENCH_SAVE_TILES"):  # the refined table, for FS2_GEMM_TILE_CACHE of a training run
            H.save_tile_cache(os.environ["FS2_BENCH_SAVE_TILES"])
    graph = None
    if use_graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        graph.replay()  # one untimed replay
        torch.cuda.synchronize()
        log("hipGraph captured and replayed once")
    run = graph.replay if graph is not None else step
    if graph is None:
        n_settle = rig.settle()
        log(f"launch plan: {rig.plan_info()} after {n_settle} more untimed step(s)")

    if rig.sync is not None:
        rig.wait_events = []
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    host_enqueue = time.perf_counter() - t0  # the host's share of the timed region: all launches enqueued, no wait yet
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    exchange_wait_ms = None
    if rig.wait_events:
        # time the main stream spent between "bucket waits enqueued" and "all buckets reduced": what is left of the
        # exchange after the backward pass it overlaps with (0 = fully hidden)
        exchange_wait_ms = sum(a.elapsed_time(b) for a, b in rig.wait_events) / len(rig.wait_events)
    rig.wait_events = None
    if world > 1:
        t = torch.tensor([elapsed, exchange_wait_ms or 0.0], device=f"cuda:{local}", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, exchange_wait_ms = float(t[0]), float(t[1])
        f = torch.tensor([frames, padded], device=f"cuda:{local}", dtype=torch.float64)
        dist.all_reduce(f, op=dist.ReduceOp.SUM)
        frames_all, padded_all = int(f[0]), int(f[1])
    else:
        frames_all, padded_all = frames, padded
    losses = {k: float(v) for k, v in model.last_losses.items()}
    plan_info = rig.plan_info()
    host_own = rig.host_cost() if (graph is None and world == 1) else None
    log(f"timed region: {elapsed / args.steps * 1e3:.2f} ms/step (host enqueue {host_enqueue / args.steps * 1e3:.2f} ms/step)")

    roofline = None
    if not args.no_roofline:
        prof, ov = gemm_profile(rig, record=rank == 0)
        if rank == 0:
            if os.environ.get("FS2_BENCH_GEMM_BREAKDOWN"):
                log_gemm_breakdown(prof)
            roofline = roofline_of(args, args.precision, prof, ov)
    log("roofline pass done")

    # The other configurations, beside the headline and never as it (`value` above is the exact fp32 MFMA path):
    #  * 32-split: fp32-accurate GEMMs on the bf16 matrix pipe (tests/test_gemm_split_gpu.py), same model and batch;
    #  * bf16_mixed_b64: BASELINE.json configs[2];
    #  * learn_alignment: the reference's default model (fs2/config/__init__.py:139-142).
    split = bf16_b64 = align = gst_leg = None
    default_cfg = config_signature(args) == {"precision": "32-true", "batch": 32, "gst": False, "learn_alignment": False}
    if default_cfg and world == 1 and not args.no_extra_legs and graph is None:
        n_leg = max(20, args.steps // 4)
        model.precision = "32-split"
        for _ in range(2):
            step()  # (tunes the tiles of the split instances)
        rig.settle()
        dt = rig.timed(n_leg) / n_leg
        model.precision = "32-true"
        split = {"precision": "32-split", "ms_per_step": round(dt * 1e3, 3), "value": round(frames / dt, 1), "unit": "mel-frames/s",
                 "steps": n_leg,
                 "note": "GEMM operands as three exact bf16 planes, six partial products per product on "
                         "v_mfma_f32_32x32x16_bf16, fp32 accumulation: the fp32 kernels' error bound (same parity tests, same "
                         "tolerances); the attention forward and dQ products the same way, dK/dV on the fp32 MFMAs; "
                         "normalisations, losses and the optimizer unchanged"}
        log(f"32-split: {dt * 1e3:.2f} ms/step")
    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cpu_cfg, cpu_batch = rig.config, rig.batch
    if default_cfg and world == 1 and not args.no_extra_legs and graph is None:
        del rig, model, step, run
        torch.cuda.empty_cache()
        rf = not args.no_refine
        bf16_b64 = extra_leg("bf16_mixed_b64", "bf16-mixed", 64, False, n_leg, local, not args.no_roofline, refine=rf)
        align = extra_leg("learn_alignment", "32-true", 32, True, n_leg, local, False, refine=rf)
        # BASELINE.json configs[4] at its per-GPU share: GST reference encoder + 16 speakers, mel up to ~1 250 frames,
        # bf16-mixed, batch 64 (fs2/gst/model.py:87-100, fs2/model.py:196-213)
        gst_leg = extra_leg("gst_bf16_b64", "bf16-mixed", 64, False, n_leg, local, not args.no_roofline, gst=True, refine=rf)
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cpu = cpu_baseline(cpu_cfg, cpu_batch)
        log("cpu baseline done")

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            "metric": "mel-frames/sec (train fwd+bwd+optimizer, whole job)",
            "value": round(frames_all * args.steps / elapsed, 1), "unit": "mel-frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"32-true": "f32", "32-split": "f32 (GEMM products from three exact bf16 planes per operand, f32 accumulate)",
                      "bf16-mixed": "bf16 MFMA operands, f32 accumulate / parameters / optimizer"}[args.precision],
            "data": "synthetic",
            "config": {"workload": (f"BASELINE.json configs[1]: fp32 train step, batch={args.batch}/GPU" if args.precision != "bf16-mixed" else
                                    f"BASELINE.json configs[2]: bf16-mixed train step, batch={args.batch}/GPU") + ", LJSpeech-shaped synthetic "
                                   "(96-128 phonemes, 80 x ~600 mel), learn_alignment=" + str(args.learn_alignment) + ", dropout on"
                                   + (" [--gst: multi-speaker + GST, mel up to ~1200 frames (configs[4] shape, fp32)]" if args.gst else ""),
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "real_frames_per_step": frames_all, "padded_frames_per_step": padded_all,
                       "precision": args.precision, "parallelism": f"dp{world}", "hipgraph": bool(graph is not None),
                       "streams": 1 if (use_graph or os.environ.get("FS2_SIDE_STREAM", "1") == "0") else 2,
                       "parameters": n_params},
            "per_gpu_value": round(frames_all * args.steps / elapsed / world, 1),
            "padded_frames_per_s": round(padded_all * args.steps / elapsed, 1),
            "loss_total": round(losses.get("total", float("nan")), 5),
            "host_enqueue_ms_per_step": None if host_own is None else round(host_own, 3),  # one step into an empty queue
            "host_loop_ms_per_step": round(host_enqueue / args.steps * 1e3, 3),  # the timed loop (queue-throttled)
            "host_threads_per_rank": host["threads"],
            "launch_plan": plan_info,
            "roofline": roofline, "cpu_baseline": cpu, "split_fp32": split, "bf16_mixed_b64": bf16_b64,
            "learn_alignment": align, "gst_bf16_b64": gst_leg,
        }
        if exchange_wait_ms is not None:
            line["exchange_wait_ms"] = round(exchange_wait_ms, 3)  # max over ranks of the per-step mean
        # whole-step view (SURVEY.md 8d closed form, padded shapes, backward = 2 x forward): algorithmic FLOPs of the
        # step / step time against the MFMA peak -- beside the dominant kernel's own roofline above
        tf = f_step / (ms_per_step * 1e-3) / 1e12
        line["whole_step"] = {"algorithmic_tflop_per_step_per_gpu": round(f_step / 1e12, 4), "achieved_tflops_per_gpu": round(tf, 2)}
        if args.precision != "bf16-mixed":
            line["whole_step"]["frac_of_fp32_mfma_peak"] = round(tf / PEAK_FP32_MFMA_TFLOPS, 4)
        else:
            line["whole_step"]["frac_of_bf16_mfma_peak"] = round(tf / PEAK_BF16_MFMA_TFLOPS, 4)
        print(json.dumps(line), flush=True)
    if world > 1 or force_sync:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
