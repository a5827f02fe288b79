"""Launch plans: a training step's launch sequence recorded once per batch geometry and replayed by one C call.

The eager step is Python down to the C ABI: ~590 entry-point calls per step, each behind a wrapper that checks its
tensors, allocates its outputs, looks up a tile -- 10 ms of host time per step where the bf16-mixed step needs 11 ms of
GPU time and the learned-alignment / GST steps were enqueue-bound outright.  None of that
work depends on the batch's CONTENTS: for a given batch geometry the sequence of launches, their arguments and the
stream each runs on are the same every step (dropout masks, learning rate, BatchNorm statistics all advance through
device memory).  So:

* the first step of a geometry runs eagerly (it also warms the tile tuner and the persistent scratch buffers);
* the second one runs eagerly too, but RECORDS: ``hip.lib()`` hands out a proxy that appends every successful
  entry-point call -- op id, argument slots, main or side stream -- to a ``Recorder``; ``modules.Env`` reports its fork /
  join events; host-side hand-offs that must happen at a fixed place in the sequence (a gradient bucket's all-reduce,
  ``parallel.GradSync``) are recorded as callbacks that split the plan into segments.  The step's allocations come from
  a ``torch.cuda.MemPool`` of the plan's own, so the recorded addresses stay reserved for it;
* every later step of that geometry copies the batch into the recorded input tensors and calls
  ``fs2hip_plan_replay`` (csrc/plan.hip) once per segment: the host cost of the step is the ~5 us ``hipLaunchKernel``
  itself per launch.  The tensors the recorded step returned (losses, outputs) are the ones every replay writes.

A replay enqueues exactly what the eager step would have: results are bit-identical (``tests/test_plan_gpu.py``, five
steps with dropout on, every model variant, two data-parallel ranks).  Anything the recorder cannot see would break
that silently -- an ATen kernel inside the step -- so recording runs under a dispatch guard that refuses ATen compute on
GPU tensors (allocation, views and host-side scalars pass); the step uses ``hip.zeros`` where it used ``torch.zeros``.

``FS2_PLAN=0`` switches plans off (every step eager: the round-4 behaviour).
"""
from __future__ import annotations

import ctypes as C
import os
import struct

import torch

from . import hip as H

SLOTS = 20
SYNC = -1
MAX_STREAMS = 8

ENABLED = os.environ.get("FS2_PLAN", "1") != "0"
#: a geometry is recorded the n-th time it is seen (the steps before run eagerly: tuner, scratch buffers, tables)
RECORD_AFTER = int(os.environ.get("FS2_PLAN_RECORD_AFTER", 1))
#: geometries kept per model (real data pads every batch differently: least recently used plans are dropped)
MAX_PLANS = int(os.environ.get("FS2_PLAN_MAX", 8))


#: ATen ops the recording guard lets through (tests' poisoned allocations are ``fill_`` calls on fresh tensors: harmless to a
#: replay, which never sees fresh memory)
GUARD_ALLOW = frozenset()


class PlanCmd(C.Structure):  # mirrors Fs2PlanCmd (include/fs2hip.h)
    _fields_ = [("op", C.c_int), ("stream", C.c_int), ("a", C.c_ulonglong * SLOTS)]


def _float_bits(v) -> int:
    return struct.unpack("<I", struct.pack("<f", float(v)))[0]


_M64 = (1 << 64) - 1
_OP_IDS = {}


def op_id(name: str) -> int:
    i = _OP_IDS.get(name)
    if i is None:
        i = _OP_IDS[name] = int(H.real_lib().fs2hip_plan_op_id(name.encode()))
    return i


class PlanError(RuntimeError):
    pass


class _RecordingLib:
    """What ``hip.lib()`` returns while a step is being recorded: the library, with every successful entry-point call
    that enqueues work also appended to the recorder."""

    def __init__(self, rec: "Recorder", real):
        self._rec, self._real, self._cache = rec, real, {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            real_fn = getattr(self._real, name)
            op = op_id(name) if name.startswith("fs2hip_") and not name.startswith("fs2hip_plan_") else -1
            if op < 0:  # size queries, version, the plan functions: no stream, nothing to record
                fn = real_fn
            else:
                rec, sig = self._rec, H.SIGNATURES.get(name)

                def fn(*args, _real=real_fn, _op=op, _sig=sig, _name=name):
                    rc = _real(*args)
                    if rc == 0:
                        rec.add(_op, _sig, args, _name)
                    return rc
            self._cache[name] = fn
        return fn


class Recorder:
    """Collects one step.  Use through ``record(...)`` below."""

    def __init__(self, main_stream: int):
        self.streams = [int(main_stream or 0)]   # raw handles: [0] the main stream, then side streams as they appear
        self.cmds = []             # (op, stream index, [slot values])
        self.keep = []             # host copies of struct arguments (addresses are in the slots)
        self.n_events = 0
        self.segments = []         # [(end index in cmds, callback, stream index the callback runs under)]
        self.host_ops = []         # callables re-run after every replay (order-free host / ATen work)
        self.lib = _RecordingLib(self, H.real_lib())

    # -- streams ---------------------------------------------------------------------------------------------------
    def stream_index(self, handle) -> int:
        handle = int(handle or 0)
        try:
            return self.streams.index(handle)
        except ValueError:
            if len(self.streams) >= MAX_STREAMS:
                raise PlanError(f"launch plan: the step used more than {MAX_STREAMS} streams")
            self.streams.append(handle)
            return len(self.streams) - 1

    @property
    def side(self):  # (the first side stream's handle, or None)
        return self.streams[1] if len(self.streams) > 1 else None

    # -- entry-point calls -----------------------------------------------------------------------------------------
    def add(self, op, sig, args, name):
        *vals, stream = args
        if len(vals) > SLOTS:
            raise PlanError(f"launch plan: {name} has {len(vals)} arguments, a command holds {SLOTS}")
        slots = []
        codes = sig if sig is not None else "p" + "q" * (len(vals) - 1)  # (struct pointer[, count]): fs2hip_gemm, *_multi
        for c, v in zip(codes, vals):
            obj = getattr(v, "_obj", v)                  # C.byref(x) -> x
            if isinstance(obj, (C.Structure, C.Array)):  # a HOST struct / job array: the plan keeps its own copy
                copy = type(obj)()
                C.memmove(C.byref(copy), C.byref(obj), C.sizeof(obj))
                self.keep.append(copy)
                slots.append(C.addressof(copy))
            elif c == "p":
                slots.append(0 if v is None else int(v) & _M64)
            elif c == "f":
                slots.append(_float_bits(v))
            else:  # i, q, Q
                slots.append(int(v) & _M64)
        self.cmds.append((op, self.stream_index(stream), slots))

    # -- what modules.Env / parallel report ------------------------------------------------------------------------
    def sync(self, record_stream, wait_stream):
        """An event recorded on one stream that the other waits for (Env.side's fork, Env.join)."""
        rs, ws = self.stream_index(record_stream), self.stream_index(wait_stream)
        self.cmds.append((SYNC, 0, [self.n_events, rs, ws]))
        self.n_events += 1

    def callback(self, fn, stream_handle):
        """``fn()`` must run on the host at THIS point of the sequence in every replay, with ``stream_handle`` the current
        stream (a bucket's all-reduce).  Splits the plan into segments."""
        self.segments.append((len(self.cmds), fn, self.stream_index(stream_handle)))

    def host_op(self, fn):
        """``fn()`` is order-free host-side work of the step that launches through ATen (the BatchNorm step counters):
        run once after the recorded step and after every replay."""
        self.host_ops.append(fn)


class _AtenGuard(torch.utils._python_dispatch.TorchDispatchMode):
    """Notes every ATen op that touches GPU memory while a step is recorded, except the ones that launch nothing
    (allocation, views, metadata): such a kernel would run in the recording and be missing from every replay."""

    NO_KERNEL = frozenset((
        "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "view", "_unsafe_view", "reshape",
        "_reshape_alias", "as_strided", "slice", "select", "alias", "detach", "detach_", "expand", "unsqueeze", "squeeze",
        "transpose", "permute", "t", "unbind", "split", "split_with_sizes", "narrow", "unfold", "lift_fresh",
        "is_same_size", "sym_size", "sym_stride", "sym_numel", "sym_storage_offset", "record_stream", "resize_", "set_",
        "is_pinned", "_has_compatible_shallow_copy_type", "view_as_real", "view_as_complex", "unsafe_split",
        "unsafe_chunk", "chunk", "movedim", "flatten", "unflatten", "squeeze_", "unsqueeze_", "requires_grad_"))

    def __init__(self, allow=()):
        super().__init__()
        self.allow = frozenset(allow)
        self.seen = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        schema = getattr(func, "_schema", None)
        full = schema.name if schema is not None else str(func)
        if full.startswith("aten::"):
            base = full[6:]
            if base not in self.NO_KERNEL and base not in self.allow:
                leaves = torch.utils._pytree.tree_leaves((args, kwargs, out))
                if any(isinstance(t, torch.Tensor) and t.is_cuda for t in leaves):
                    self.seen.append(base)
        return out


class StepPlan:
    def __init__(self, rec: Recorder, pool, inputs: dict, result, device):
        self.device = device
        n = len(rec.cmds)
        self.cmds = (PlanCmd * max(n, 1))()
        for i, (op, st, slots) in enumerate(rec.cmds):
            c = self.cmds[i]
            c.op, c.stream = op, st
            for j, v in enumerate(slots):
                c.a[j] = v
        self.n = n
        self.keep = rec.keep
        self.pool = pool
        self.inputs = inputs          # key -> the device tensor the recorded step read
        self.result = result
        self.host_ops = rec.host_ops
        self.stream_handles = list(rec.streams)   # [0] is replaced by the current stream of each replay
        self.n_events = rec.n_events
        self.events = (C.c_void_p * max(rec.n_events, 1))()
        if rec.n_events:
            H._ok(H.real_lib().fs2hip_plan_events_create(self.events, rec.n_events), "plan_events_create")
        bounds, first = [], 0
        for end, fn, st in rec.segments:
            bounds.append((first, end, fn, st))
            first = end
        bounds.append((first, n, None, 0))
        self.segments = bounds
        self.launches = sum(1 for op, _, _ in rec.cmds if op >= 0)
        self._failed = C.c_int(-1)
        self.replays = 0

    def __del__(self):
        try:
            if self.n_events:
                H.real_lib().fs2hip_plan_events_destroy(self.events, self.n_events)
        except Exception:
            pass

    def feed(self, batch: dict):
        """The batch into the recorded input tensors (skipped for tensors that ARE the recorded ones)."""
        for k, dst in self.inputs.items():
            src = batch[k]
            if src is dst:
                continue
            if not torch.is_tensor(src) or src.shape != dst.shape:
                raise PlanError(f"launch plan: batch[{k!r}] does not have the recorded geometry")
            dst.copy_(src, non_blocking=True)

    def replay(self, side_streams=None):
        """``side_streams``: {raw handle: torch.cuda.Stream} of the model's side streams (host callbacks recorded under a
        side stream run under ``torch.cuda.stream`` of it)."""
        tab = (C.c_void_p * len(self.stream_handles))(*self.stream_handles)
        tab[0] = H._stream()
        L = H.real_lib()
        for first, end, fn, st in self.segments:
            if end > first:
                rc = L.fs2hip_plan_replay(self.cmds, first, end, tab, len(self.stream_handles), self.events, self.n_events,
                                          C.byref(self._failed))
                if rc != 0:
                    raise RuntimeError(f"fs2hip: launch plan command {self._failed.value} failed with code {rc}")
            if fn is not None:
                side = (side_streams or {}).get(self.stream_handles[st]) if st else None
                if side is not None:
                    with torch.cuda.stream(side):
                        fn()
                else:
                    fn()
        for fn in self.host_ops:
            fn()
        self.replays += 1
        return self.result


def record(step_fn, inputs: dict, device, guard_allow=()):
    """Runs ``step_fn()`` eagerly on the current stream while recording it.  ``inputs``: the device tensors of the
    batch the step reads (they become the plan's input buffers).  Returns (plan, what step_fn returned)."""
    if H._REC is not None:
        raise PlanError("launch plan: recording is not re-entrant")
    import gc
    pool = torch.cuda.MemPool()
    rec = Recorder(H._stream())
    guard = _AtenGuard(frozenset(guard_allow) | GUARD_ALLOW)
    # No cyclic garbage collection while allocations are routed to the pool: collecting an OLD plan (model -> plan ->
    # closures -> model is a cycle) destroys its MemPool, and the caching allocator aborts the process when a pool is
    # torn down while another one is being allocated to.
    gc_was = gc.isenabled()
    gc.disable()
    H._REC = rec
    try:
        with torch.cuda.use_mem_pool(pool, device=device), guard:
            result = step_fn()
    finally:
        H._REC = None
        if gc_was:
            gc.enable()
    for fn in rec.host_ops:
        fn()
    if guard.seen:
        raise PlanError("launch plan: the step ran ATen kernels on GPU tensors that a replay would not contain: "
                        + ", ".join(sorted(set(guard.seen))))
    return StepPlan(rec, pool, inputs, result, device), result


class PlanCache:
    """Per-model table geometry -> plan, with the 'seen n times before recording' counter and LRU eviction."""

    def __init__(self):
        self.plans, self.seen = {}, {}
        self.recorded = self.replayed = self.eager = 0

    def lookup(self, sig):
        p = self.plans.get(sig)
        if p is not None:
            self.plans[sig] = self.plans.pop(sig)  # most recently used last
        return p

    def should_record(self, sig) -> bool:
        n = self.seen.get(sig, 0)
        self.seen[sig] = n + 1
        if len(self.seen) > 4096:
            self.seen.clear()
        return n >= RECORD_AFTER

    def store(self, sig, plan):
        self.plans[sig] = plan
        while len(self.plans) > MAX_PLANS:
            self.plans.pop(next(iter(self.plans)))

    def clear(self):
        self.plans.clear()
        self.seen.clear()
