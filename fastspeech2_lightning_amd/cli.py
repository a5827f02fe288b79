"""``fs2l train`` -- the training entry of the feature-prediction path (reference ``fs2/cli/train.py:9-41``,
``fs2/cli/cli.py:46-54``; the Trainer the reference borrows from the parent toolkit's ``train_base_command`` is a
Lightning ``Trainer(gradient_clip_val=1.0, monitor="validation/total_loss", max_epochs, max_steps, ...)``).

    fs2l train CONFIG.yaml [-c training.batch_size=8 ...] [--devices N] [--resume last.ckpt]

What it does, in the reference's order: load the YAML/JSON config (+ ``-c key=value`` overrides), read
``<preprocessing.save_dir>/stats.json``, build the speaker / language look-up tables from the two filelists, build the
model (``lang2id``, ``speaker2id``, ``stats``), then train: ``FeatureDataset -> DataLoader(collate, pinned) ->
DevicePrefetcher -> training_step -> [bucketed gradient exchange] -> fused clip + AdamW + Noam``, validate on
``validation/total_loss``, write ``last.ckpt`` / ``best.ckpt`` in Lightning's layout (weights under the reference's
keys, optimizer state in ``torch.optim.AdamW``'s format) and resume from one.

MI355X-first: one process per GPU (``--devices N`` starts the ranks itself before any GPU call, or run under
``torch.distributed.run``), RCCL gradient buckets overlapped with the backward pass, no host synchronisation inside
a step -- losses are copied out once every ``--log-every`` steps.
"""
from __future__ import annotations

import argparse
import csv
import json
import os
import sys
import time
from pathlib import Path
from typing import Optional

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the HIP runtime initialises: see fastspeech2_lightning_amd/hip.py

import torch

MONITOR = "validation/total_loss"  # fs2/cli/train.py:37
GRADIENT_CLIP_VAL = 1.0            # fs2/cli/train.py:38


# --------------------------------------------------------------------------------------------------------------
# configuration and file plumbing (host only: exercised by the CPU tests)
# --------------------------------------------------------------------------------------------------------------
def apply_overrides(raw: dict, overrides: list) -> dict:
    """``-c training.batch_size=8``: dotted path into the config dict, value parsed as JSON when it parses."""
    for item in overrides or []:
        if "=" not in item:
            raise SystemExit(f"-c expects key=value, got {item!r}")
        key, value = item.split("=", 1)
        try:
            value = json.loads(value)
        except json.JSONDecodeError:
            pass
        node = raw
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
            if not isinstance(node, dict):
                raise SystemExit(f"-c {key}: {p!r} is not a section")
        node[parts[-1]] = value
    return raw


def load_config(path, overrides=None):
    from .config import FastSpeech2Config, _load_json_or_yaml
    path = Path(path)
    raw = apply_overrides(_load_json_or_yaml(path), overrides)
    return FastSpeech2Config.model_validate(raw, context={"config_path": path})


def read_filelist(path) -> list:
    """The preprocessor's pipe-separated filelist with a header row (``basename|language|speaker|characters|
    character_tokens|phones|phone_tokens|...``); rows become the entry dicts ``FeatureDataset`` takes."""
    with open(path, encoding="utf8", newline="") as f:
        rows = list(csv.DictReader(f, delimiter="|", quoting=csv.QUOTE_NONE))
    if not rows or "basename" not in rows[0]:
        raise ValueError(f"{path}: not a '|'-separated filelist with a 'basename' column")
    return rows


def lookup_tables(*filelists) -> tuple:
    """(lang2id, speaker2id) over the given filelists: names sorted, ids dense (the parent toolkit's
    ``lookuptables_from_config`` builds them from the training and validation filelists)."""
    langs = sorted({r.get("language") or "default" for fl in filelists for r in fl})
    speakers = sorted({r.get("speaker") or "default" for fl in filelists for r in fl})
    return {n: i for i, n in enumerate(langs)}, {n: i for i, n in enumerate(speakers)}


def filter_entries(entries: list, config) -> list:
    """Rows that carry the text representation the model trains on (the reference's
    ``filter_dataset_based_on_target_text_representation_level``)."""
    from .config import TargetTrainingTextRepresentationLevel as L
    key = "character_tokens" if config.model.target_text_representation_level == L.characters else "phone_tokens"
    kept = [e for e in entries if e.get(key)]
    if not kept:
        raise ValueError(f"no filelist row has a {key!r} column value")
    return kept


def run_dir(config, args) -> Path:
    lg = getattr(config.training, "logger", None)
    get = (lambda k, d: (lg.get(k, d) if isinstance(lg, dict) else getattr(lg, k, d))) if lg is not None else (lambda k, d: d)
    base = Path(args.output_dir) if args.output_dir else Path(get("save_dir", "logs_and_checkpoints")) / str(get("name", "FeaturePredictionExperiment")) / str(get("version", "base"))
    return base


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="fs2l", description="MI355X-native FastSpeech2 feature-prediction path")
    sub = ap.add_subparsers(dest="command", required=True)
    tr = sub.add_parser("train", help="Train your Text-to-Spec model")
    tr.add_argument("config_file", type=Path)
    tr.add_argument("-c", "--config-args", action="append", default=[], metavar="KEY=VALUE")
    tr.add_argument("-d", "--devices", default="auto", help="GPUs on this node: a count or 'auto' (all visible)")
    tr.add_argument("--precision", default="32-true", choices=["32-true", "32-split", "bf16-mixed"])
    tr.add_argument("--resume", type=Path, default=None, help="checkpoint to resume from (default: <run dir>/checkpoints/last.ckpt when present)")
    tr.add_argument("--output-dir", type=Path, default=None, help="run directory (default: training.logger.save_dir/name/version)")
    tr.add_argument("--max-steps", type=int, default=None, help="overrides training.max_steps")
    tr.add_argument("--max-epochs", type=int, default=None, help="overrides training.max_epochs")
    tr.add_argument("--val-every", type=int, default=None, help="validate every N optimizer steps (default: once per epoch)")
    tr.add_argument("--ckpt-every", type=int, default=None, help="write last.ckpt every N optimizer steps (default: once per epoch)")
    tr.add_argument("--log-every", type=int, default=50)
    tr.add_argument("--seed", type=int, default=1234)
    tr.add_argument("--tune-tiles", action="store_true",
                    help="before training: the GEMM tile tuner's in-step stage on the first training batch (the runner-up "
                         "tiles of the heaviest shapes tried inside the replayed step; ~30-60 s; -1...2 %% per step).  "
                         "Weights, optimizer state and BatchNorm buffers are restored afterwards; one rank only.  Save the "
                         "table with FS2_GEMM_TILE_CACHE=<file> to reuse it")
    tr.add_argument("--dry-run", action="store_true", help="resolve config, filelists, look-up tables and the run directory, print the plan, touch no GPU")
    bm = sub.add_parser("benchmark", help="Time the forward pass on one batch of the training filelist (reference fs2/cli/benchmark.py)")
    bm.add_argument("config_file", type=Path)
    bm.add_argument("-c", "--config-args", action="append", default=[], metavar="KEY=VALUE")
    bm.add_argument("--benchmark-type", choices=["training", "inference"], default="training")
    bm.add_argument("--warmup-reps", type=int, default=10)
    bm.add_argument("--repetitions", type=int, default=300)
    bm.add_argument("--precision", default="32-true", choices=["32-true", "32-split", "bf16-mixed"])
    return ap


def plan(args) -> dict:
    """Everything ``train`` needs that does not involve the GPU."""
    config = load_config(args.config_file, args.config_args)
    t = config.training
    if not t.training_filelist or not t.validation_filelist:
        raise SystemExit("training.training_filelist and training.validation_filelist must be set")
    base = args.config_file.parent

    def resolve(p):
        p = Path(p)
        return p if p.is_absolute() else (base / p)

    config.preprocessing.save_dir = str(resolve(config.preprocessing.save_dir))
    train_rows = filter_entries(read_filelist(resolve(t.training_filelist)), config)
    val_rows = filter_entries(read_filelist(resolve(t.validation_filelist)), config)
    lang2id, speaker2id = lookup_tables(train_rows, val_rows)
    stats_path = Path(config.preprocessing.save_dir) / "stats.json"
    with open(stats_path, encoding="utf8") as f:
        stats = json.load(f)
    out = run_dir(config, args)
    if not out.is_absolute():
        out = base / out
    ckpt_dir = out / "checkpoints"
    resume = args.resume if args.resume else (ckpt_dir / "last.ckpt" if (ckpt_dir / "last.ckpt").exists() else None)
    return dict(config=config, stats=stats, lang2id=lang2id, speaker2id=speaker2id, train_rows=train_rows,
                val_rows=val_rows, run_dir=out, ckpt_dir=ckpt_dir, resume=resume,
                max_steps=args.max_steps if args.max_steps is not None else t.max_steps,
                max_epochs=args.max_epochs if args.max_epochs is not None else t.max_epochs)


# --------------------------------------------------------------------------------------------------------------
# the loop (one rank)
# --------------------------------------------------------------------------------------------------------------
class Trainer:
    def __init__(self, args, p: dict, rank: int, world: int, local_rank: int):
        from . import hip as H
        from .config import Stats
        from .data import FeatureDataset
        from .model import FastSpeech2
        from .parallel import GradSync

        self.args, self.p, self.rank, self.world = args, p, rank, world
        self.device = torch.device("cuda", local_rank)
        torch.cuda.set_device(self.device)
        config = p["config"]
        self.global_step, self.epoch = 0, 0
        ckpt = None
        if p["resume"] is not None:
            self.model, ckpt = FastSpeech2.load_from_checkpoint(p["resume"], device=str(self.device),
                                                                precision=args.precision, return_checkpoint=True)
            self.model.config.training = config.training  # schedule / filelists / batch size are this run's
            self.model.config.preprocessing.save_dir = config.preprocessing.save_dir
        else:
            self.model = FastSpeech2(config, Stats(**p["stats"]), p["lang2id"], p["speaker2id"], device=str(self.device),
                                     seed=args.seed, precision=args.precision)
            finetune = getattr(config.training, "finetune_checkpoint", None)
            if finetune:  # weights only, fresh optimizer (Lightning: load_from_checkpoint + fit without ckpt_path)
                sd = torch.load(finetune, map_location="cpu", weights_only=False)["state_dict"]
                self.model.load_state_dict(sd)
        self.model.train()
        opts, scheds = self.model.configure_optimizers()
        self.opt, self.sched = opts[0], scheds[0]["scheduler"]
        # Trainer(gradient_clip_val=1.0): Lightning passes the value through this hook before every optimizer step;
        # here it is constant, so once (the clip is part of the fused optimizer launch)
        self.model.configure_gradient_clipping(self.opt, GRADIENT_CLIP_VAL, "norm")
        if ckpt is not None:
            self.global_step, self.epoch = self.model.restore_training_state(ckpt, self.opt)
            self.sched.last_epoch = self.opt.steps_done()  # host mirror of the device-resident schedule
            self.best = ckpt.get("fs2l_best_monitor", float("inf"))
            self.batches_in_epoch = int(ckpt.get("fs2l_batches_in_epoch", 0))
        else:
            self.best, self.batches_in_epoch = float("inf"), 0
        self.sync = None
        if world > 1:
            self.sync = GradSync(self.model.store)
            self.sync.broadcast_parameters(0)
            self.model.data_parallel(self.sync, rank)
            self.opt.grad_scale = self.sync.grad_scale
        self._tiles_shared = world == 1
        cfg = self.model.config
        self.train_set = FeatureDataset(p["train_rows"], cfg, self.model.lang2id, self.model.speaker2id)
        self.val_set = FeatureDataset(p["val_rows"], cfg, self.model.lang2id, self.model.speaker2id)
        self.H = H
        self.metrics = None
        if rank == 0:
            p["ckpt_dir"].mkdir(parents=True, exist_ok=True)
            self.metrics = open(p["run_dir"] / "metrics.jsonl", "a", encoding="utf8")

    # ---- data ---------------------------------------------------------------------------------------------
    def loader(self, dataset, shuffle: bool, epoch: int, workers: int, skip_batches: int = 0):
        """Batches of one epoch in an order that depends only on (seed, epoch, world): a resumed run re-derives the
        interrupted epoch's order and drops the ``skip_batches`` it had already trained on."""
        from functools import partial

        from .data import collate
        cfg = self.model.config
        if self.world > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(dataset, self.world, self.rank, shuffle=shuffle,
                                                                      seed=self.args.seed, drop_last=False)
            sampler.set_epoch(epoch)
        elif shuffle:
            sampler = torch.utils.data.RandomSampler(dataset, generator=torch.Generator().manual_seed(self.args.seed + epoch))
        else:
            sampler = torch.utils.data.SequentialSampler(dataset)
        batches = list(torch.utils.data.BatchSampler(sampler, cfg.training.batch_size, drop_last=False))[skip_batches:]
        return torch.utils.data.DataLoader(
            dataset, batch_sampler=batches, num_workers=workers,
            collate_fn=partial(collate, learn_alignment=cfg.model.learn_alignment, pin_memory=workers == 0))

    def batches(self, dataset, shuffle, epoch, workers, skip_batches: int = 0):
        from .data import DevicePrefetcher
        return DevicePrefetcher(self.loader(dataset, shuffle, epoch, workers, skip_batches), self.model.prepare_batch,
                                self.device)

    # ---- one step -------------------------------------------------------------------------------------------
    def step(self, batch):
        self.model.current_epoch_ = self.epoch
        with torch.no_grad():  # the native loop reads the flat gradient buffer directly: no autograd node needed
            self.model.training_step(batch)
        if self.sync:
            self.sync.wait()
        self.opt.step()
        self.sched.step()  # host arithmetic only (the kernels advance the device record themselves)
        self.global_step += 1
        self.batches_in_epoch += 1
        if not self._tiles_shared:
            # every rank sums in the same order: the GEMM tiles rank 0 tuned during its first step are adopted by all
            # (shapes first seen later are tuned per rank; their tiles only change the summation order)
            from .parallel import share_tile_table
            share_tile_table(0)
            self._tiles_shared = True

    def log(self, record: dict):
        record = dict(record, step=self.global_step, epoch=self.epoch, time=round(time.time(), 3))
        if self.metrics:
            self.metrics.write(json.dumps(record) + "\n")
            self.metrics.flush()
            print(json.dumps(record), flush=True)

    def log_train(self):
        self.model.check_bad_data()  # fs2/variance_adaptor.py:289-305, read where the host synchronises anyway
        slots = self.model._loss_slots.cpu()  # the one D2H copy per log interval
        from .model import LOSS_KEYS
        rec = {f"training/{k}_loss": float(slots[i]) for i, k in enumerate(LOSS_KEYS) if k in self.model.last_losses}
        rec["training/total_loss"] = float(slots[len(LOSS_KEYS)])
        r = self.opt.record()
        rec.update(lr=r["lr"], grad_norm=r["grad_norm"])
        self.log(rec)

    def validate(self) -> float:
        from .data import validate
        cfg = self.model.config
        if self.sync is not None:
            self.sync.broadcast_buffers(0)  # DDP broadcast_buffers: every rank evaluates with rank 0's running statistics
        means = validate(self.model, self.batches(self.val_set, False, 0, cfg.training.val_data_workers))
        self.model.train()
        self.log(means)
        return means.get(MONITOR, float("inf"))

    def save(self, name: str):
        if self.rank != 0:
            return
        ckpt = self.model.checkpoint_dict(self.global_step, self.epoch, self.opt)
        ckpt["fs2l_best_monitor"] = self.best
        ckpt["fs2l_batches_in_epoch"] = self.batches_in_epoch
        path = self.p["ckpt_dir"] / name
        tmp = path.with_suffix(".tmp")
        torch.save(ckpt, tmp)
        os.replace(tmp, path)

    def after_step(self):
        a = self.args
        if a.log_every and self.global_step % a.log_every == 0:
            self.log_train()
        if a.val_every and self.global_step % a.val_every == 0:
            self.check_validation()
        if a.ckpt_every and self.global_step % a.ckpt_every == 0:
            self.save("last.ckpt")

    def check_validation(self):
        m = self.validate()
        if m < self.best:
            self.best = m
            self.save("best.ckpt")

    def tune_tiles(self):
        """``--tune-tiles``: ``hip.refine_tiles_in_step`` on the first training batch.  The trial steps are real steps, so
        everything they touch -- weights, gradient, Adam moments, the device step record (learning-rate schedule, dropout
        masks), BatchNorm running statistics and counters -- is snapshotted first and put back afterwards: training
        starts from exactly the state it would have started from, with a better tile table."""
        if self.world > 1:
            self.log({"tune_tiles": "skipped: one rank only"})
            return
        H, model, S = self.H, self.model, self.model.store
        loader = self.loader(self.train_set, True, self.epoch, 0, 0)
        batch = model.prepare_batch(next(iter(loader)))
        snap = [t.clone() for t in (S.flat, S.grad, S.adam_m, S.adam_v, model.step_state, S.bn_counters)]
        bufs = {k: v.clone() for k, v in S.buffers.items()}

        def step():
            with torch.no_grad():
                model.training_step(batch)
            self.opt.step()

        def settle():
            for _ in range(6):
                before = model.plans.replayed
                step()
                if model.plans.replayed > before:
                    break
            torch.cuda.synchronize()

        def eager_step():
            model.plan_enabled = False
            try:
                step()
            finally:
                model.plan_enabled = True
        t0 = time.perf_counter()
        for _ in range(2):
            step()  # (first-stage tuning of this geometry's shapes)
        ms, changed = H.refine_tiles_in_step(step, rounds=8, top=32, candidates=3, settle=settle, count_step=eager_step)
        with torch.no_grad():
            for dst, src in zip((S.flat, S.grad, S.adam_m, S.adam_v, model.step_state, S.bn_counters), snap):
                dst.copy_(src)
            for k, v in bufs.items():
                S.buffers[k].copy_(v)
        S.weights_changed()
        model.plans.clear()  # (their recorded loss / output tensors belong to the tuning steps)
        if os.environ.get("FS2_GEMM_TILE_CACHE"):
            H.save_tile_cache(os.environ["FS2_GEMM_TILE_CACHE"])
        self.log({"tune_tiles": {"changed": changed, "ms_per_step": round(ms, 3), "seconds": round(time.perf_counter() - t0, 1)}})

    def fit(self):
        if getattr(self.args, "tune_tiles", False):
            self.tune_tiles()
        p, cfg = self.p, self.model.config
        max_steps, max_epochs = p["max_steps"], p["max_epochs"]
        done = lambda: 0 <= max_steps <= self.global_step  # noqa: E731  (Lightning: max_steps = -1 means no limit)
        from .data import DevicePrefetcher
        while self.epoch < max_epochs and not done():
            loader = self.loader(self.train_set, True, self.epoch, cfg.training.train_data_workers, self.batches_in_epoch)
            epoch_batches = self.batches_in_epoch + len(loader)
            for batch in DevicePrefetcher(loader, self.model.prepare_batch, self.device):
                self.step(batch)
                self.after_step()
                if done():
                    break
            if self.batches_in_epoch >= epoch_batches:  # the epoch is complete (also when max_steps fell on its last batch)
                self.epoch += 1
                self.batches_in_epoch = 0
                if not self.args.val_every:
                    self.check_validation()
                if not self.args.ckpt_every:
                    self.save("last.ckpt")
        torch.cuda.synchronize()
        self.model.check_bad_data()
        self.save("last.ckpt")
        if self.metrics:
            self.metrics.close()
        return self.global_step


def spawn_ranks(n: int, argv: list) -> int:
    """``--devices N`` without a launcher: N child processes, one per GPU, started before this process makes any GPU
    call (it never does); never an exec of a process that has touched the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-m", "fastspeech2_lightning_amd"] + argv, env=env))
    return wait_ranks(procs)


def wait_ranks(procs: list, poll_s: float = 0.2) -> int:
    """Waits for every rank; when one exits non-zero the others -- which would sit in a rendezvous or a collective
    until the process-group timeout -- are terminated, and that rank's code is returned."""
    failed = None
    while any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            rc = p.poll()
            if rc not in (None, 0) and failed is None:
                failed = (r, rc)
                print(f"rank {r} exited with code {rc}: stopping the other ranks", file=sys.stderr, flush=True)
                for q in procs:
                    if q.poll() is None:
                        q.terminate()
        time.sleep(poll_s)
        if failed is not None:
            deadline = time.time() + 10.0
            while any(p.poll() is None for p in procs) and time.time() < deadline:
                time.sleep(poll_s)
            for q in procs:
                if q.poll() is None:
                    q.kill()
    if failed is not None:
        return abs(failed[1]) or 1
    return max((abs(p.returncode) for p in procs), default=0)


def gather_from_ranks(procs: list, queue, n: int, timeout: float = 300.0, poll_s: float = 0.5) -> list:
    """``n`` results from a queue that ``multiprocessing`` ranks feed, watching the ranks while waiting: a rank that
    dies (non-zero exit code) fails the wait in seconds -- its siblings, which would sit in a rendezvous or a
    collective, are terminated -- instead of after the queue's whole timeout (``wait_ranks`` for ``Process`` objects)."""
    import queue as _queue
    out, deadline = [], time.time() + timeout
    while len(out) < n:
        try:
            out.append(queue.get(timeout=poll_s))
            continue
        except _queue.Empty:
            pass
        dead = [(r, p.exitcode) for r, p in enumerate(procs) if p.exitcode not in (None, 0)]
        # (a rank that exited cleanly has already queued its result; Empty + every rank gone = results lost)
        gone = not dead and all(p.exitcode is not None for p in procs) and queue.empty()
        if dead or gone or time.time() > deadline:
            for p in procs:
                if p.is_alive():
                    p.terminate()
            for p in procs:
                p.join(10)
                if p.is_alive():
                    p.kill()
            why = (f"rank(s) {dead} exited with an error" if dead else
                   "every rank exited without a result" if gone else f"no result within {timeout:.0f} s")
            raise RuntimeError(f"gather_from_ranks: {why} ({len(out)} of {n} results)")
    return out


def train(args, argv) -> int:
    p = plan(args)
    if args.dry_run:
        print(json.dumps({
            "config": str(args.config_file), "run_dir": str(p["run_dir"]), "resume": str(p["resume"]) if p["resume"] else None,
            "train_utterances": len(p["train_rows"]), "validation_utterances": len(p["val_rows"]),
            "lang2id": p["lang2id"], "speaker2id": p["speaker2id"], "batch_size": p["config"].training.batch_size,
            "max_steps": p["max_steps"], "max_epochs": p["max_epochs"], "monitor": MONITOR,
            "gradient_clip_val": GRADIENT_CLIP_VAL, "learn_alignment": p["config"].model.learn_alignment,
            "stats": sorted(p["stats"])}))
        return 0
    world = int(os.environ.get("WORLD_SIZE", 0))
    if not world:
        n = torch.cuda.device_count() if args.devices == "auto" else int(args.devices)  # (device_count does not initialise the GPU)
        if n > 1:
            return spawn_ranks(n, argv)
        world = 1
    rank, local = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    backend = os.environ.get("FS2L_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)
    # this rank's share of the host: intra-op threads, and a cap on the DataLoader workers the config asks for
    from .parallel import apply_host_budget
    budget = apply_host_budget(int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    t = p["config"].training
    t.train_data_workers = min(int(t.train_data_workers), budget["workers"])
    t.val_data_workers = min(int(t.val_data_workers), budget["workers"])
    trainer = Trainer(args, p, rank, world, local)
    steps = trainer.fit()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"finished": True, "global_step": steps, "epoch": trainer.epoch, "best_" + MONITOR: trainer.best,
                          "checkpoint": str(p["ckpt_dir"] / "last.ckpt")}), flush=True)
    return 0


def benchmark(args) -> int:
    """reference ``fs2/cli/benchmark.py:14-80``: the first ``training.batch_size`` items of the training filelist,
    collated; ``warmup_reps`` untimed forward passes, then ``repetitions`` timed ones (HIP events, one sync each);
    prints mean and standard deviation in ms.  ``training`` = teacher-forced forward (targets in the batch),
    ``inference`` = ``forward(inference=True)``."""
    import numpy as np

    from .config import Stats
    from .data import FeatureDataset, collate
    from .model import FastSpeech2

    args.output_dir = args.resume = None
    args.max_steps = args.max_epochs = None
    p = plan(args)
    config = p["config"]
    model = FastSpeech2(config, Stats(**p["stats"]), p["lang2id"], p["speaker2id"], precision=args.precision)
    model.eval()
    ds = FeatureDataset(p["train_rows"], config, p["lang2id"], p["speaker2id"])
    n = min(config.training.batch_size, len(ds))
    batch = model.prepare_batch(collate([ds[i] for i in range(n)], learn_alignment=config.model.learn_alignment))
    inference = args.benchmark_type == "inference"
    if inference:
        batch = {k: v for k, v in batch.items() if k not in ("mel", "mel_lens", "pitch", "energy", "duration")}
        batch.update(mel=None, mel_lens=None, duration=None, max_mel_len=config.model.max_length)
    for _ in range(args.warmup_reps):
        model(batch, inference=inference)
    timings = np.zeros(args.repetitions)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(args.repetitions):
        start.record()
        model(batch, inference=inference)
        end.record()
        torch.cuda.synchronize()
        timings[rep] = start.elapsed_time(end)
    print(f"Average forward pass for {args.benchmark_type} duration after {args.repetitions} repetitions: "
          f"{timings.mean()} ms Standard Deviation: {timings.std()}")
    return 0


def main(argv: Optional[list] = None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)
    if args.command == "train":
        return train(args, argv)
    if args.command == "benchmark":
        return benchmark(args)
    return 2


if __name__ == "__main__":
    raise SystemExit(main())
