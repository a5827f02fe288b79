"""Flat parameter store.

All trainable parameters live in ONE fp32 device buffer, laid out in forward
execution order and in the layout the kernels want (k-tap conv weights as
[tap][Cout][Cin], depthwise weights as [tap][C]); gradients, Adam first and second
moments are three more buffers of the same shape.  Consequences:

* the optimizer is a single fused AdamW launch over 18 M contiguous floats;
* the data-parallel exchange is an all-reduce of contiguous slices of the
  gradient buffer, issued bucket by bucket as the backward pass (which fills
  the buffer from the end towards the beginning) completes them;
* ``state_dict()`` / ``load_state_dict()`` convert to and from the reference's
  key names and tensor layouts, so checkpoints interchange.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Callable, Optional

import torch

# layout kinds: reference tensor -> native tensor
_TO_NATIVE = {
    "id": lambda t: t,
    "convk": lambda t: t.permute(2, 0, 1),   # (Cout, Cin, K) -> [K][Cout][Cin]
    "dw": lambda t: t[:, 0, :].t(),           # (C, 1, K)     -> [K][C]
    "pw": lambda t: t[:, :, 0],               # (Cout, Cin, 1) -> [Cout][Cin]
    "conv2d": lambda t: t.permute(2, 3, 1, 0),  # (Cout, Cin, kh, kw) -> [kh][kw][Cin][Cout]
    # GRU input weights (3U, C*W) with columns c*W+w  ->  columns w*C+c (channels-last feature order); W = 2
    "gru_ih_w2": lambda t: t.reshape(t.shape[0], -1, 2).transpose(1, 2).reshape(t.shape[0], -1),
    # (N, K) with K not a multiple of 4 -> zero columns up to the next multiple (16-byte operand pieces)
    "padk4": lambda t: torch.nn.functional.pad(t, (0, (-t.shape[1]) % 4)),
}
_TO_REF = {
    "id": lambda t, shape: t,
    "convk": lambda t, shape: t.permute(1, 2, 0),
    "dw": lambda t, shape: t.t().unsqueeze(1),
    "pw": lambda t, shape: t.unsqueeze(-1),
    "conv2d": lambda t, shape: t.permute(3, 2, 0, 1),
    "gru_ih_w2": lambda t, shape: t.reshape(t.shape[0], 2, -1).transpose(1, 2).reshape(t.shape[0], -1),
    "padk4": lambda t, shape: t[:, :shape[1]],
}


def native_shape(ref_shape, kind):
    s = tuple(ref_shape)
    if kind == "convk":
        return (s[2], s[0], s[1])
    if kind == "dw":
        return (s[2], s[0])
    if kind == "pw":
        return (s[0], s[1])
    if kind == "conv2d":
        return (s[2], s[3], s[1], s[0])
    if kind == "padk4":
        return (s[0], (s[1] + 3) // 4 * 4)
    return s


# ---- initialisers on reference-shaped CPU tensors (PyTorch module defaults) ---------------
def init_linear_weight(t):
    torch.nn.init.kaiming_uniform_(t, a=math.sqrt(5))


def init_bias_for(fan_in):
    def f(t):
        bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        torch.nn.init.uniform_(t, -bound, bound)
    return f


def init_xavier(gain_name):
    def f(t):
        torch.nn.init.xavier_uniform_(t, gain=torch.nn.init.calculate_gain(gain_name))
    return f


def init_ones(t):
    t.fill_(1.0)


def init_zeros(t):
    t.zero_()


def init_normal(t):
    torch.nn.init.normal_(t)


@dataclass
class Entry:
    name: str
    ref_shape: tuple
    kind: str
    init: Callable
    offset: int = 0
    numel: int = 0
    bucket: int = 0


class ParamStore:
    ALIGN = 8  # elements: every view starts 16-byte aligned in the fp32 buffers AND in the bf16 mirror of the weights

    def __init__(self):
        self.entries: "OrderedDict[str, Entry]" = OrderedDict()
        self.buffers: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self._buffer_specs = []
        self.order_hint: list[str] = []  # reference registration order for state_dict()
        self.flat = self.grad = self.adam_m = self.adam_v = None
        self.total = 0
        self._bucket = 0
        self._pviews, self._gviews = {}, {}
        self._transposed, self._tviews, self._tviews32 = [], {}, {}
        self._transposed_bf16_only = set()

    # ---- declaration phase -----------------------------------------------------------------
    def add(self, name, ref_shape, kind="id", init=init_zeros):
        if name in self.entries:
            raise KeyError(name)
        e = Entry(name, tuple(ref_shape), kind, init)
        e.numel = int(torch.Size(native_shape(ref_shape, kind)).numel())  # storage (the padded kind is larger)
        e.offset = self.total
        e.bucket = self._bucket
        self.total += (e.numel + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.entries[name] = e
        self.order_hint.append(name)
        return name

    def add_buffer(self, name, tensor: torch.Tensor):
        self._buffer_specs.append((name, tensor))
        self.order_hint.append(name)
        return name

    def next_bucket(self):
        """Marks a gradient-bucket boundary (data-parallel exchange granularity)."""
        self._bucket += 1

    # ---- materialisation -------------------------------------------------------------------
    def finalize(self, device, seed: Optional[int] = None):
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        host = torch.zeros(self.total, dtype=torch.float32)
        state = torch.random.get_rng_state() if gen is not None else None
        if gen is not None:
            torch.manual_seed(seed)
        for e in self.entries.values():
            ref = torch.empty(e.ref_shape, dtype=torch.float32)
            e.init(ref)
            host[e.offset:e.offset + e.numel] = _TO_NATIVE[e.kind](ref).contiguous().reshape(-1)
        if state is not None:
            torch.random.set_rng_state(state)
        self.flat = host.to(device)
        self.grad = torch.zeros_like(self.flat)
        self.adam_m = torch.zeros_like(self.flat)
        self.adam_v = torch.zeros_like(self.flat)
        for name, t in self._buffer_specs:
            self.buffers[name] = t.to(device)
        # BatchNorm step counters: views of ONE int64 vector, so that a training forward advances all of them with a
        # single launch (FastSpeech2.forward) instead of one per BatchNorm layer
        names = [n for n, _ in self._buffer_specs if n.endswith("num_batches_tracked")]
        self.bn_counters = torch.zeros(len(names), dtype=torch.long, device=device)
        for i, n in enumerate(names):
            self.bn_counters[i] = self.buffers[n]
            self.buffers[n] = self.bn_counters[i]
        self.device = torch.device(device)
        self._pviews, self._gviews = {}, {}
        self.flat_bf16, self._bviews = None, {}
        return self

    # ---- bf16 mirror of the weights ("bf16-mixed" with operand storage): ONE cast pass over the flat buffer per step;
    # the GEMMs read weights from it in every orientation, so no per-layer casts and no transposed copies exist ----
    #: bumped by everything that rewrites the weights (the optimizer step, ``load_state_dict``, ``move_to``): the transposed
    #: mirrors carry the generation they were made from, and ``pt`` / ``pbt`` hand out None for a stale one -- the caller
    #: (``hip.linear_bwd_data(wt=...)``) then reads the weight itself in the other orientation, which is always right
    weights_gen = 0
    _gen_t16 = _gen_t32 = -1

    def weights_changed(self):
        self.weights_gen += 1

    def refresh_bf16(self, transposed: bool = True):
        """``transposed=False`` (evaluation / inference forwards, which never run a data gradient): the cast alone."""
        from . import hip as H
        if self.flat_bf16 is None:
            self.flat_bf16 = torch.empty(self.total, device=self.flat.device, dtype=torch.bfloat16)
        H.cast_bf16(self.flat, out=self.flat_bf16)
        if self._transposed and transposed:
            self._gen_t16 = self.weights_gen
            # transposed bf16 mirrors (one launch for all of them): the weights whose data-gradient GEMM runs in the
            # forward orientation on the weights-stationary kernel (reduction = the model width)
            if not self._tviews:
                for name in self._transposed:
                    n, k = native_shape(self.entries[name].ref_shape, self.entries[name].kind)
                    self._tviews[name] = torch.empty(k, n, device=self.flat.device, dtype=torch.bfloat16)
            H.transpose_cast_bf16_multi([(self.p(name), self._tviews[name]) for name in self._transposed])

    def refresh_transposed_fp32(self):
        """The fp32 form of the same mirrors ("32-true": the K = 256 data gradients in the forward orientation, tile 32
        of the GEMM): one launch per forward pass."""
        from . import hip as H
        if not self._transposed:
            return
        if not self._tviews32:
            for name in self._transposed:
                if name in self._transposed_bf16_only:
                    continue
                n, k = native_shape(self.entries[name].ref_shape, self.entries[name].kind)
                self._tviews32[name] = torch.empty(k, n, device=self.flat.device, dtype=torch.float32)
        if self._tviews32:
            H.transpose_cast_bf16_multi([(self.p(name), v) for name, v in self._tviews32.items()])
            self._gen_t32 = self.weights_gen

    def pt(self, name):
        """The transposed fp32 mirror of ``name`` (None when it was not declared, not refreshed yet, or made from
        weights that have changed since)."""
        return self._tviews32.get(name) if self._gen_t32 == self.weights_gen else None

    def want_transposed(self, name, bf16_only=False):
        """Declares that ``pbt(name)`` -- the weight [N, K] as bf16 [K, N] -- is wanted (kept by ``refresh_bf16``);
        ``bf16_only``: no fp32 form (``pt``) of it."""
        if len(native_shape(self.entries[name].ref_shape, self.entries[name].kind)) != 2:
            raise ValueError(f"{name}: only matrices have a transposed mirror")
        if name not in self._transposed:
            self._transposed.append(name)
            if bf16_only:
                self._transposed_bf16_only.add(name)

    def pbt(self, name):
        """The transposed bf16 mirror of ``name`` (None when it was not declared, not refreshed yet, or made from
        weights that have changed since)."""
        return self._tviews.get(name) if self._gen_t16 == self.weights_gen else None

    def pb(self, name):
        v = self._bviews.get(name)
        if v is None:
            v = self._bviews[name] = self._view(self.flat_bf16, name)
        return v

    def _view(self, base, name):
        e = self.entries[name]
        return base[e.offset:e.offset + e.numel].view(native_shape(e.ref_shape, e.kind))

    def p(self, name):
        v = self._pviews.get(name)  # the views never change (finalize() allocates once): built on first use
        if v is None:
            v = self._pviews[name] = self._view(self.flat, name)
        return v

    def g(self, name):
        v = self._gviews.get(name)
        if v is None:
            v = self._gviews[name] = self._view(self.grad, name)
        return v

    def b(self, name):
        return self.buffers[name]

    def bucket_ranges(self):
        """[(start, end)] float offsets of each gradient bucket, in declaration order."""
        out = {}
        for e in self.entries.values():
            end = e.offset + (e.numel + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            s, t = out.get(e.bucket, (e.offset, end))
            out[e.bucket] = (min(s, e.offset), max(t, end))
        return [out[k] for k in sorted(out)]

    def bucket_map(self):
        """{bucket id: (start, end) or None} for every id handed out by ``next_bucket`` (None: the bucket holds no
        parameter, e.g. a model built without one of its optional blocks)."""
        ids = sorted({e.bucket for e in self.entries.values()})
        ranges = dict(zip(ids, self.bucket_ranges()))
        return {b: ranges.get(b) for b in range(self._bucket + 1)}

    @property
    def num_trainable(self):
        return sum(e.numel for e in self.entries.values())

    # ---- any flat buffer of this layout <-> reference-shaped tensors (gradient, Adam moments) ----------------
    def export_flat(self, base: torch.Tensor, name: str) -> torch.Tensor:
        e = self.entries[name]
        return _TO_REF[e.kind](self._view(base, name), e.ref_shape).contiguous().clone()

    def import_flat(self, base: torch.Tensor, name: str, value: torch.Tensor) -> None:
        e = self.entries[name]
        if tuple(value.shape) != e.ref_shape:
            raise RuntimeError(f"{name}: shape {tuple(value.shape)} != {e.ref_shape}")
        with torch.no_grad():
            self._view(base, name).copy_(_TO_NATIVE[e.kind](value.to(torch.float32)).to(base.device))

    # ---- reference-layout state dict ---------------------------------------------------------
    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        sd = OrderedDict()
        for name in self.order_hint:
            if name in self.entries:
                e = self.entries[name]
                sd[name] = _TO_REF[e.kind](self.p(name), e.ref_shape).contiguous().clone()
            else:
                sd[name] = self.buffers[name].clone()
        return sd

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self.order_hint if k not in sd]
        unexpected = [k for k in sd if k not in self.entries and k not in self.buffers]
        if strict and (missing or unexpected):
            raise RuntimeError(f"state dict mismatch: missing {missing[:8]} unexpected {unexpected[:8]}")
        self.weights_changed()
        with torch.no_grad():
            for k, v in sd.items():
                if k in self.entries:
                    e = self.entries[k]
                    if tuple(v.shape) != e.ref_shape:
                        raise RuntimeError(f"{k}: shape {tuple(v.shape)} != {e.ref_shape}")
                    self.p(k).copy_(_TO_NATIVE[e.kind](v.to(torch.float32)).to(self.device))
                elif k in self.buffers:
                    b = self.buffers[k]
                    if tuple(v.shape) != tuple(b.shape):
                        raise RuntimeError(f"{k}: shape {tuple(v.shape)} != {tuple(b.shape)}")
                    b.copy_(v.to(b.dtype))
        return missing, unexpected

    def grad_state_dict(self):
        """Gradients in the reference's layout (for parity tests)."""
        return OrderedDict((n, _TO_REF[e.kind](self.g(n), e.ref_shape).contiguous().clone())
                           for n, e in self.entries.items())
