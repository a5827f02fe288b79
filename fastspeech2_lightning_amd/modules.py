"""Layers of the feature-prediction path as explicit forward / backward launch sequences over
the fs2hip kernels (no autograd tape: each ``fwd`` returns the tensors its ``bwd`` needs, each
``bwd`` writes its parameter gradients straight into the flat gradient buffer).

Layer structure follows the reference (file:line in each class); parameter names are the
reference's state-dict keys.  All activations are dense padded (B, T, C) fp32 -- padding rows are
processed like any other row because the reference does so too (finding 4).
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import hip as H
from . import params as P


class Ctx:
    """What a layer's forward keeps for its backward."""
    __slots__ = ("t",)

    def __init__(self, **t):
        self.t = t

    def __getattr__(self, k):
        try:
            return self.t[k]
        except KeyError:
            raise AttributeError(k)


class Env:
    """Per-model run-time switches shared by all layers."""

    def __init__(self, step_state: torch.Tensor, seed: int = 0):
        self.training = False
        self.step_state = step_state
        self.seed = seed
        self._site = 0

    def new_site(self) -> int:
        self._site += 1
        return self._site

    @property
    def stored(self) -> bool:
        """"bf16-mixed" with bf16 operand STORAGE: between GEMMs, activations and gradients exist only as the bf16
        operands those GEMMs read (written by the producing kernel: LayerNorm, a GEMM epilogue, BatchNorm + activation,
        the depthwise convolution's backward), weights are read from the bf16 mirror of the flat buffer
        (``ParamStore.refresh_bf16``, one cast pass per step) in every orientation, and a layer's bias gradient comes
        out of its weight-gradient GEMM.  The residual stream, normalisation statistics, attention, losses and the
        optimizer stay fp32."""
        return H.GEMM_BF16 == 1 and H.BF16_STORAGE

    def drop(self, p: float, site: int) -> H.Drop:
        if not self.training or p <= 0.0:
            return H.NO_DROP
        return H.Drop(p, (self.seed * 0x9E3779B1 + site * 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF, self.step_state)

    # ---- second HIP stream for parameter-gradient work ---------------------------------------------------------
    # In the backward pass a layer's weight-gradient GEMM (+ split-K finish + bias column sums) depends only on
    # tensors that already exist, and nothing reads its result before the optimizer / the bucket exchange.  It is
    # therefore enqueued on a side stream: its workgroups fill the CUs that the main chain (dX GEMM -> LayerNorm
    # backward -> ...) leaves idle in the partial last round of a GEMM and under its small kernels.  ``join()``
    # (bucket boundaries, end of backward) makes the main stream wait for it.  Tensors the side stream reads are
    # kept alive until the join, so the caching allocator cannot hand their memory to the main stream early.
    side_enabled = os.environ.get("FS2_SIDE_STREAM", "1") != "0"  # FS2_SIDE_STREAM=0: everything on one stream

    def side(self, *tensors, lane: int = 0):
        """Context manager: run the enclosed launches on a side stream, after everything enqueued so far.  ``lane``
        picks one of several side streams: independent chains (the three variance predictors) each get their own and
        run beside one another as well as beside the main chain."""
        return _SideSection(self, tensors, lane)

    def wgrad(self, dy, x, out, **kw):
        """A layer's weight-gradient GEMM: on the side stream now -- or, while the stack holds its layer's weight gradients
        back for one grouped launch (``Conformer.bwd``), only noted."""
        if H.holding_weight_gradients():
            H.linear_bwd_weight(dy, x, out, **kw)
            return
        with self.side(dy, x):
            H.linear_bwd_weight(dy, x, out, **kw)

    def side_streams(self) -> dict:
        """{raw handle: torch.cuda.Stream} of the side streams created so far."""
        return {st.cuda_stream: st for st in getattr(self, "_lanes", {}).values()}

    def join(self, only=None):
        """The current stream waits for the side streams that have run anything since their last join (``only``: just
        these lanes -- the tensors held for the side streams are then kept)."""
        lanes = getattr(self, "_lanes", None)
        if not lanes or not getattr(self, "_side_dirty", None):
            return
        for lane in sorted(self._side_dirty if only is None else self._side_dirty & set(only)):
            st = lanes[lane]
            ev = torch.cuda.Event()
            ev.record(st)
            torch.cuda.current_stream().wait_event(ev)
            if H._REC is not None:
                H._REC.sync(st.cuda_stream, H._stream())
            self._side_dirty.discard(lane)
        if not self._side_dirty:
            self._side_held = []


#: side streams, ONE per (device, lane) for the whole process: every model's Env shares them (steps of different models are
#: ordered on them exactly as on the shared main stream).  A stream per model was how the fourth model of a process -- the
#: configs[4] leg of the default bench line -- came to run 50 % slower in some runs (round 4's driver: 30.2 ms against
#: 23.1; round 5: 30.7 against 20.1): ROCclr maps HIP streams onto a few hardware queues, torch hands out pooled streams
#: round-robin, and a side stream that lands on the MAIN stream's hardware queue turns the two-stream step into one queue
#: with cross-stream waits in it.
_SIDE_STREAMS = {}


def _overlaps(main, side) -> bool:
    """True when a spin kernel on ``side`` runs beside one on ``main`` (two hardware queues), False when the two take
    twice as long as one (the streams share a hardware queue)."""
    spin = int(1.0e-3 * 2.0e9)

    def timed(both):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        with torch.cuda.stream(main):
            torch.cuda._sleep(spin)
        if both:
            side.wait_event(e0)
            with torch.cuda.stream(side):
                torch.cuda._sleep(spin)
            ev = torch.cuda.Event()
            ev.record(side)
            main.wait_event(ev)
        e1.record(main)
        e1.synchronize()
        return e0.elapsed_time(e1)
    timed(True)
    one = min(timed(False) for _ in range(2))
    two = min(timed(True) for _ in range(2))
    return two < 1.5 * one


def shared_side_stream(lane: int = 0):
    """The process-wide side stream of ``lane`` on the current device: created once, and CHECKED to run beside the
    current (main) stream -- up to eight fresh streams are tried until one's spin kernel overlaps the main stream's
    (a few milliseconds, once per process and lane)."""
    dev = torch.cuda.current_device()
    st = _SIDE_STREAMS.get((dev, lane))
    if st is None:
        main = torch.cuda.current_stream()
        tried = []
        for _ in range(8):
            st = torch.cuda.Stream()
            tried.append(st)  # (kept alive while probing: a released pooled stream would be handed out again)
            if st.cuda_stream != main.cuda_stream and all(st.cuda_stream != o.cuda_stream for o in _SIDE_STREAMS.values()) \
                    and _overlaps(main, st):
                break
        else:
            import warnings
            warnings.warn("fs2hip: no side stream with a hardware queue of its own was found (GPU_MAX_HW_QUEUES?): the "
                          "two-stream step will run as one queue")
        _SIDE_STREAMS[(dev, lane)] = st
    return st


class _SideSection:
    def __init__(self, env: Env, tensors, lane=0):
        self.env, self.tensors, self.lane = env, tensors, lane
        self.ctx = None

    def __enter__(self):
        env = self.env
        if not env.side_enabled:
            return self
        if getattr(env, "_lanes", None) is None:
            env._lanes, env._side_held, env._side_dirty = {}, [], set()
        st = env._lanes.get(self.lane)
        if st is None:
            st = env._lanes[self.lane] = shared_side_stream(self.lane)
            if self.lane == 0:
                env._side_stream = st
        ev = torch.cuda.Event()
        ev.record()
        st.wait_event(ev)
        if H._REC is not None:
            H._REC.sync(H._stream(), st.cuda_stream)
        env._side_held.extend(self.tensors)
        env._side_dirty.add(self.lane)
        self.ctx = torch.cuda.stream(st)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


# ------------------------------------------------------------------------------------------------
# declaration helpers (reference key names)
# ------------------------------------------------------------------------------------------------
def decl_linear(S: P.ParamStore, prefix, n_out, n_in, wname="weight", bname="bias", w_init=None, b_init=None):
    S.add(f"{prefix}{wname}", (n_out, n_in), "id", w_init or P.init_linear_weight)
    S.add(f"{prefix}{bname}", (n_out,), "id", b_init or P.init_bias_for(n_in))


def decl_norm(S: P.ParamStore, prefix):
    return prefix


class LayerNorm:
    def __init__(self, S: P.ParamStore, prefix, dim):
        if dim % 4 or not 0 < dim <= 1024:
            raise ValueError(f"LayerNorm over {dim} channels: the kernels take widths that are multiples of 4, up to 1024")
        self.S, self.w, self.b, self.dim = S, prefix + "weight", prefix + "bias", dim
        S.add(self.w, (dim,), "id", P.init_ones)
        S.add(self.b, (dim,), "id", P.init_zeros)

    def fwd(self, x, out_dtype=torch.float32):
        y, mean, rstd = H.layernorm_fwd(x, self.S.p(self.w), self.S.p(self.b), out_dtype=out_dtype)
        return y, (x, mean, rstd)

    def bwd(self, dy, saved, dx_add=None, nxt=None, dz_dtype=torch.float32):
        """``nxt`` = (scale, Drop, bias gradient) of the sub-module below (its ``dz_spec()``): returns (dx, dz) with
        dz = scale * dropmask * dx made, and its column sums (that bias gradient) started, by the same kernel.
        ``dy`` may be bf16 and dz is ``dz_dtype`` (bf16 operand storage)."""
        x, mean, rstd = saved
        if nxt is None:
            return H.layernorm_bwd(dy, x, self.S.p(self.w), mean, rstd, self.S.g(self.w), self.S.g(self.b), dx_add, defer=True)
        scale, drop, bias_grad = nxt
        return H.layernorm_bwd(dy, x, self.S.p(self.w), mean, rstd, self.S.g(self.w), self.S.g(self.b), dx_add, defer=True,
                               dz_scale=scale, dz_drop=drop, dz_colsum=bias_grad, dz_dtype=dz_dtype)


class BatchNorm:
    """nn.BatchNorm1d over channels of [M, C] (+ running buffers)."""

    def __init__(self, S: P.ParamStore, prefix, dim):
        self.S, self.prefix, self.dim = S, prefix, dim
        S.add(prefix + "weight", (dim,), "id", P.init_ones)
        S.add(prefix + "bias", (dim,), "id", P.init_zeros)
        S.add_buffer(prefix + "running_mean", torch.zeros(dim))
        S.add_buffer(prefix + "running_var", torch.ones(dim))
        S.add_buffer(prefix + "num_batches_tracked", torch.zeros((), dtype=torch.long))

    def stats(self, parts, training):
        S, p = self.S, self.prefix
        st = H.bn_finalize(parts, S.p(p + "weight"), S.p(p + "bias"), S.b(p + "running_mean"),
                           S.b(p + "running_var"), training=training)
        # num_batches_tracked is advanced once per training forward for all layers (ParamStore.bn_counters)
        return st

    def grads(self):
        return self.S.g(self.prefix + "weight"), self.S.g(self.prefix + "bias")


# ------------------------------------------------------------------------------------------------
# Conformer (torchaudio.models.Conformer; call sites fs2/model.py:95-119)
# ------------------------------------------------------------------------------------------------
class FeedForward:
    """LayerNorm -> Linear(D,F) -> SiLU -> Dropout -> Linear(F,D) -> Dropout; y = x + 0.5 * f(x)."""

    def __init__(self, S, env: Env, prefix, d, f, p, dims_ok=False):
        self.S, self.env, self.p, self.dims_ok = S, env, p, dims_ok
        self.ln = LayerNorm(S, prefix + "sequential.0.", d)
        self.w1, self.b1 = prefix + "sequential.1.weight", prefix + "sequential.1.bias"
        self.w2, self.b2 = prefix + "sequential.4.weight", prefix + "sequential.4.bias"
        decl_linear(S, prefix + "sequential.1.", f, d)
        decl_linear(S, prefix + "sequential.4.", d, f)
        self.s1, self.s2 = env.new_site(), env.new_site()
        self.f, self.d = f, d
        if dims_ok and d == 256:  # the data gradient through the second Linear reduces over d: forward orientation on W2^T
            S.want_transposed(self.w2)
            if f == 1024:  # ... and through the first one over f: the K = 1024 streaming kernel (bf16 operand storage)
                S.want_transposed(self.w1, bf16_only=True)

    def fwd(self, x):
        S, env = self.S, self.env
        if env.stored and self.dims_ok:
            bf = torch.bfloat16
            h, ln_saved = self.ln.fwd(x, bf)
            u = torch.empty(*x.shape[:-1], self.f, device=x.device, dtype=bf)
            a = H.linear_fwd(h, S.pb(self.w1), S.p(self.b1), epi=H.EPI_ACT, act="silu", out_pre=u,
                             drop=env.drop(self.p, self.s1), out_dtype=bf)
            y = H.linear_fwd(a, S.pb(self.w2), S.p(self.b2), epi=H.EPI_RESID, resid=x, res_scale=0.5,
                             drop=env.drop(self.p, self.s2))
            return y, Ctx(ln=ln_saved, h=h, u=u, a=a, stored=True)
        h, ln_saved = self.ln.fwd(x)
        u = torch.empty(*x.shape[:-1], self.f, device=x.device, dtype=torch.float32)
        a = H.linear_fwd(h, S.p(self.w1), S.p(self.b1), epi=H.EPI_ACT, act="silu", out_pre=u,
                         drop=env.drop(self.p, self.s1))
        y = H.linear_fwd(a, S.p(self.w2), S.p(self.b2), epi=H.EPI_RESID, resid=x, res_scale=0.5,
                         drop=env.drop(self.p, self.s2))
        return y, Ctx(ln=ln_saved, h=h, u=u, a=a, stored=False)

    def dz_spec(self):
        """(scale, Drop, bias gradient): what turns the gradient of this sub-module's output into the gradient dz of its
        last Linear's output, and where dz's column sums go (``LayerNorm.bwd(nxt=...)`` of the sub-module above)."""
        return 0.5, self.env.drop(self.p, self.s2), self.S.g(self.b2)

    def bwd(self, dy, c, dz=None, nxt=None):
        """``dz``: already made (with its bias gradient) by the LayerNorm backward above; ``nxt``: see ``LayerNorm.bwd``
        (the return value is then (dx, next dz))."""
        S, env = self.S, self.env
        fused = dz is not None
        if c.stored:
            bf = torch.bfloat16
            if not fused:  # (not reached from ConformerLayer.bwd, which always hands dz down)
                dz = H.cast_bf16(H.axpby(dy, None, 0.5, 0.0, env.drop(self.p, self.s2)))
            env.wgrad(dz, c.a, S.g(self.w2), bias_grad=None if fused else S.g(self.b2))
            du = H.linear_bwd_data(dz, S.pb(self.w2), epi=H.EPI_DACT, act="silu", aux=c.u,
                                   drop=env.drop(self.p, self.s1), out_dtype=bf, wt=S.pbt(self.w2))
            env.wgrad(du, c.h, S.g(self.w1), bias_grad=S.g(self.b1))
            dh = H.linear_bwd_data(du, S.pb(self.w1), out_dtype=bf, wt=S.pbt(self.w1))
            return self.ln.bwd(dh, c.ln, dx_add=dy, nxt=nxt, dz_dtype=bf)
        if not fused:
            dz = H.axpby(dy, None, 0.5, 0.0, env.drop(self.p, self.s2))
        env.wgrad(dz, c.a, S.g(self.w2), bias_grad=None if fused else S.g(self.b2))
        du = H.linear_bwd_data(dz, S.p(self.w2), epi=H.EPI_DACT, act="silu", aux=c.u, drop=env.drop(self.p, self.s1),
                               wt=S.pt(self.w2))
        env.wgrad(du, c.h, S.g(self.w1), bias_grad=S.g(self.b1))
        dh = H.linear_bwd_data(du, S.p(self.w1))
        return self.ln.bwd(dh, c.ln, dx_add=dy, nxt=nxt)


#: FS2_BF16_CHAIN=0 (measurement aid): under bf16 operand storage the convolution module's value|gate / depthwise result
#: and the PostNet's inner convolution results stay fp32 tensors (the state before they became bf16)
BF16_CHAIN = os.environ.get("FS2_BF16_CHAIN", "1") != "0"
#: FS2_PRED_STORED=0 (measurement aid): the variance predictors' pointwise GEMMs round fp32 operands in registers as in
#: rounds 1-4 instead of reading bf16 operands from memory
PRED_STORED = os.environ.get("FS2_PRED_STORED", "1") != "0"
#: FS2_POSTNET_IM2COL=0 (measurement aid): the PostNet's two 80-mel-bin convolutions (first layer forward, last layer data
#: gradient) on the fp32-operand tiles with per-piece tap decoding, as in rounds 1-4
POSTNET_IM2COL = os.environ.get("FS2_POSTNET_IM2COL", "1") != "0"


class SelfAttention:
    """LayerNorm -> nn.MultiheadAttention(key_padding_mask) -> Dropout; y = x + f(x)."""

    HEAD_DIMS = (16, 32, 64, 128)  # what the attention kernels are built for (fs2hip_attention_fwd refuses the rest)

    def __init__(self, S, env: Env, prefix, d, heads, p, dims_ok=False):
        self.dims_ok = dims_ok
        if heads <= 0 or d % heads or d // heads not in self.HEAD_DIMS:
            raise ValueError(f"Conformer attention: input_dim {d} / heads {heads} gives a head dimension of "
                             f"{d / max(heads, 1):g}; this build has attention kernels for head dimensions {self.HEAD_DIMS}")
        self.S, self.env, self.p, self.heads = S, env, p, heads
        self.ln = LayerNorm(S, prefix + "self_attn_layer_norm.", d)
        a = prefix + "self_attn."
        self.wi, self.bi, self.wo, self.bo = a + "in_proj_weight", a + "in_proj_bias", a + "out_proj.weight", a + "out_proj.bias"
        S.add(self.wi, (3 * d, d), "id", P.init_xavier("linear"))
        S.add(self.bi, (3 * d,), "id", P.init_zeros)
        S.add(self.wo, (d, d), "id", P.init_linear_weight)
        S.add(self.bo, (d,), "id", P.init_zeros)
        self.sa, self.so = env.new_site(), env.new_site()
        self.d = d
        if dims_ok and d == 256:
            S.want_transposed(self.wo)

    def fwd(self, x, lens):
        S, env = self.S, self.env
        B, T, _ = x.shape
        if env.stored and self.dims_ok and H.attention_b_supported(self.d // self.heads):
            # bf16 end to end: the projection writes bf16 q | k | v, the attention kernels read and write bf16
            bf = torch.bfloat16
            h, ln_saved = self.ln.fwd(x, bf)
            qkv = H.linear_fwd(h, S.pb(self.wi), S.p(self.bi), out_dtype=bf)
            ob, lse = H.attention_fwd_b(qkv, lens, B, T, self.heads, env.drop(self.p, self.sa))
            y = H.linear_fwd(ob, S.pb(self.wo), S.p(self.bo), epi=H.EPI_RESID, resid=x, drop=env.drop(self.p, self.so))
            return y, Ctx(ln=ln_saved, h=h, qkv=qkv, o=None, ob=ob, lse=lse, lens=lens, stored=True)
        if env.stored and self.dims_ok:  # other head dims: the fp32-storage attention kernels between two casts
            h, ln_saved = self.ln.fwd(x, torch.bfloat16)
            qkv = H.linear_fwd(h, S.pb(self.wi), S.p(self.bi))
            o, lse = H.attention_fwd(qkv, lens, B, T, self.heads, env.drop(self.p, self.sa))
            ob = H.cast_bf16(o)
            y = H.linear_fwd(ob, S.pb(self.wo), S.p(self.bo), epi=H.EPI_RESID, resid=x, drop=env.drop(self.p, self.so))
            return y, Ctx(ln=ln_saved, h=h, qkv=qkv, o=o, ob=ob, lse=lse, lens=lens, stored=True)
        h, ln_saved = self.ln.fwd(x)
        qkv = H.linear_fwd(h, S.p(self.wi), S.p(self.bi))
        # a training forward keeps its scores for the backward pass's dK/dV kernel (exact fp32: hip.attention_scores_kept)
        sc = None
        if env.training:
            o, lse, sc = H.attention_fwd(qkv, lens, B, T, self.heads, env.drop(self.p, self.sa), save_scores=True)
        else:
            o, lse = H.attention_fwd(qkv, lens, B, T, self.heads, env.drop(self.p, self.sa))
        y = H.linear_fwd(o, S.p(self.wo), S.p(self.bo), epi=H.EPI_RESID, resid=x, drop=env.drop(self.p, self.so))
        return y, Ctx(ln=ln_saved, h=h, qkv=qkv, o=o, lse=lse, lens=lens, stored=False, scores=sc)

    def dz_spec(self):
        return 1.0, self.env.drop(self.p, self.so), self.S.g(self.bo)

    def bwd(self, dy, c, dz=None, nxt=None):
        S, env = self.S, self.env
        B, T, _ = dy.shape
        fused = dz is not None
        if c.stored:
            bf = torch.bfloat16
            if not fused:
                dz = H.cast_bf16(H.axpby(dy, None, 1.0, 0.0, env.drop(self.p, self.so)))
            env.wgrad(dz, c.ob, S.g(self.wo), bias_grad=None if fused else S.g(self.bo))
            if c.o is None:
                do = H.linear_bwd_data(dz, S.pb(self.wo), out_dtype=bf, wt=S.pbt(self.wo))
                dqkv = H.attention_bwd_b(c.qkv, c.lens, c.ob, do, c.lse, B, T, self.heads, env.drop(self.p, self.sa))
            else:
                do = H.linear_bwd_data(dz, S.pb(self.wo))  # fp32: the attention kernels' input
                dqkv = H.cast_bf16(H.attention_bwd(c.qkv, c.lens, c.o, do, c.lse, B, T, self.heads, env.drop(self.p, self.sa)))
            env.wgrad(dqkv, c.h, S.g(self.wi), bias_grad=S.g(self.bi))
            dh = H.linear_bwd_data(dqkv, S.pb(self.wi), out_dtype=bf)
            return self.ln.bwd(dh, c.ln, dx_add=dy, nxt=nxt, dz_dtype=bf)
        if not fused:
            d_o = env.drop(self.p, self.so)
            dz = H.axpby(dy, None, 1.0, 0.0, d_o) if d_o.p > 0 else dy
        env.wgrad(dz, c.o, S.g(self.wo), bias_grad=None if fused else S.g(self.bo))
        do = H.linear_bwd_data(dz, S.p(self.wo), wt=S.pt(self.wo))
        dqkv = H.attention_bwd(c.qkv, c.lens, c.o, do, c.lse, B, T, self.heads, env.drop(self.p, self.sa), scores=c.scores)
        env.wgrad(dqkv, c.h, S.g(self.wi), bias_grad=S.g(self.bi))
        dh = H.linear_bwd_data(dqkv, S.p(self.wi))
        return self.ln.bwd(dh, c.ln, dx_add=dy, nxt=nxt)


class ConvModule:
    """LayerNorm -> Conv1d(D,2D,1) -> GLU -> depthwise Conv1d(k) -> BatchNorm1d -> SiLU -> Conv1d(D,D,1)
    -> Dropout; y = x + f(x)."""

    KERNEL_SIZES = (3, 5, 7, 9, 15, 31)  # depthwise-convolution widths the kernels are instantiated for

    def __init__(self, S, env: Env, prefix, d, k, p, dims_ok=False):
        self.dims_ok = dims_ok
        if k not in self.KERNEL_SIZES:
            raise ValueError(f"Conformer convolution module: depthwise kernel size {k}; this build carries {self.KERNEL_SIZES}")
        self.S, self.env, self.p, self.k, self.d = S, env, p, k, d
        self.ln = LayerNorm(S, prefix + "layer_norm.", d)
        q = prefix + "sequential."
        self.w1, self.b1 = q + "0.weight", q + "0.bias"
        self.wd, self.bd = q + "2.weight", q + "2.bias"
        self.w2, self.b2 = q + "5.weight", q + "5.bias"
        S.add(self.w1, (2 * d, d, 1), "pw", P.init_linear_weight)
        S.add(self.b1, (2 * d,), "id", P.init_bias_for(d))
        S.add(self.wd, (d, 1, k), "dw", P.init_linear_weight)
        S.add(self.bd, (d,), "id", P.init_bias_for(k))
        self.bn = BatchNorm(S, q + "3.", d)
        S.add(self.w2, (d, d, 1), "pw", P.init_linear_weight)
        S.add(self.b2, (d,), "id", P.init_bias_for(d))
        self.site = env.new_site()
        if dims_ok and d == 256:
            S.want_transposed(self.w2)

    def fwd(self, x):
        S, env = self.S, self.env
        B, T, _ = x.shape
        stored = env.stored and self.dims_ok
        if stored:
            # value | gate, the depthwise convolution's result and the gradients that retrace them are bf16 tensors (what
            # autocast hands these convolutions); the BatchNorm statistics are fp32 sums over the rounded values
            h, ln_saved = self.ln.fwd(x, torch.bfloat16)
            g2 = H.linear_fwd(h, S.pb(self.w1), S.p(self.b1), out_dtype=torch.bfloat16 if BF16_CHAIN else torch.float32)
        else:
            h, ln_saved = self.ln.fwd(x)
            g2 = H.linear_fwd(h, S.p(self.w1), S.p(self.b1))
        c, parts = H.dwconv_fwd(g2, S.p(self.wd), S.p(self.bd), B, T, glu=True, stats=env.training)
        stats = self.bn.stats(parts, env.training)
        if stored:
            s = H.bn_act_fwd(c, stats, "silu", bf16_only=True)
            y = H.linear_fwd(s, S.pb(self.w2), S.p(self.b2), epi=H.EPI_RESID, resid=x, drop=env.drop(self.p, self.site))
        else:
            s = H.bn_act_fwd(c, stats, "silu")
            y = H.linear_fwd(s, S.p(self.w2), S.p(self.b2), epi=H.EPI_RESID, resid=x, drop=env.drop(self.p, self.site))
        return y, Ctx(ln=ln_saved, h=h, g2=g2, c=c, stats=stats, s=s, stored=stored)

    def dz_spec(self):
        return 1.0, self.env.drop(self.p, self.site), self.S.g(self.b2)

    def bwd(self, dy, c, dz=None, nxt=None):
        S, env = self.S, self.env
        B, T, _ = dy.shape
        fused = dz is not None
        gg, gb = self.bn.grads()
        if c.stored:
            bf = torch.bfloat16
            if not fused:
                dz = H.cast_bf16(H.axpby(dy, None, 1.0, 0.0, env.drop(self.p, self.site)))
            env.wgrad(dz, c.s, S.g(self.w2), bias_grad=None if fused else S.g(self.b2))
            chain = c.c.dtype == bf
            ds = H.linear_bwd_data(dz, S.pb(self.w2), out_dtype=bf if chain else torch.float32, wt=S.pbt(self.w2))
            dc = H.bn_act_bwd(ds, c.c, c.stats, gg, gb, "silu", training=env.training, bf16_only=chain)
            dg2 = H.dwconv_bwd(dc, c.g2, S.p(self.wd), S.g(self.wd), S.g(self.bd), B, T, glu=True, out_dtype=bf)
            env.wgrad(dg2, c.h, S.g(self.w1), bias_grad=S.g(self.b1))
            dh = H.linear_bwd_data(dg2, S.pb(self.w1), out_dtype=bf)
            return self.ln.bwd(dh, c.ln, dx_add=dy, nxt=nxt, dz_dtype=bf)
        if not fused:
            d_o = env.drop(self.p, self.site)
            dz = H.axpby(dy, None, 1.0, 0.0, d_o) if d_o.p > 0 else dy
        env.wgrad(dz, c.s, S.g(self.w2), bias_grad=None if fused else S.g(self.b2))
        ds = H.linear_bwd_data(dz, S.p(self.w2), wt=S.pt(self.w2))
        dc = H.bn_act_bwd(ds, c.c, c.stats, gg, gb, "silu", training=env.training)
        dg2 = H.dwconv_bwd(dc, c.g2, S.p(self.wd), S.g(self.wd), S.g(self.bd), B, T, glu=True)
        env.wgrad(dg2, c.h, S.g(self.w1), bias_grad=S.g(self.b1))
        dh = H.linear_bwd_data(dg2, S.p(self.w1))
        return self.ln.bwd(dh, c.ln, dx_add=dy, nxt=nxt)


class ConformerLayer:
    def __init__(self, S, env, prefix, d, f, heads, k, p):
        # bf16 operand storage (Env.stored) is decided for the whole layer: the gradient a sub-module's LayerNorm
        # backward hands to the sub-module below has that sub-module's operand type
        ok = d % 8 == 0 and f % 8 == 0
        self.ffn1 = FeedForward(S, env, prefix + "ffn1.", d, f, p, ok)
        self.attn = SelfAttention(S, env, prefix, d, heads, p, ok)
        self.conv = ConvModule(S, env, prefix + "conv_module.", d, k, p, ok)
        self.ffn2 = FeedForward(S, env, prefix + "ffn2.", d, f, p, ok)
        self.final = LayerNorm(S, prefix + "final_layer_norm.", d)

    def fwd(self, x, lens):
        x, c1 = self.ffn1.fwd(x)
        x, c2 = self.attn.fwd(x, lens)
        x, c3 = self.conv.fwd(x)
        x, c4 = self.ffn2.fwd(x)
        y, c5 = self.final.fwd(x)
        return y, (c1, c2, c3, c4, c5)

    def bwd(self, dy, c):
        c1, c2, c3, c4, c5 = c
        # every LayerNorm backward also makes the dropout-masked, scaled gradient (and the bias gradient) the
        # sub-module below starts from
        d, dz = self.final.bwd(dy, c5, nxt=self.ffn2.dz_spec(), dz_dtype=torch.bfloat16 if c4.stored else torch.float32)
        d, dz = self.ffn2.bwd(d, c4, dz, nxt=self.conv.dz_spec())
        d, dz = self.conv.bwd(d, c3, dz, nxt=self.attn.dz_spec())
        d, dz = self.attn.bwd(d, c2, dz, nxt=self.ffn1.dz_spec())
        return self.ffn1.bwd(d, c1, dz)


#: FS2_WGRAD_GROUP_ROWS=16384 (measurement aid; default 0 = off): a Conformer stack whose activations have at most this
#: many rows -- the encoder, B x Ts = 4096 rows at the benchmark shape -- holds each layer's eight weight-gradient GEMMs
#: back and enqueues them as ONE grouped launch on the side stream when the layer's backward is through
#: (``fs2hip_gemm_grouped``).  Measured (same box, alternating, profiles/r05_ab_group.log): fp32 18.49-18.54 ms per step with
#: it against 18.44-18.48 without, bf16-mixed batch 64 10.73 against 10.71 -- one large launch at the end of a layer fills
#: the side stream less well than eight small ones spread under the layer's main chain: off.
WGRAD_GROUP_ROWS = int(os.environ.get("FS2_WGRAD_GROUP_ROWS", "0"))


class Conformer:
    def __init__(self, S, env, prefix, cfg):
        """Every layer but the first opens a new gradient bucket (data-parallel exchange granularity): ``buckets[i]``
        is the bucket layer i's parameters belong to; layer 0 shares the bucket that is open when the stack is declared."""
        self.env = env
        self.layers, self.buckets = [], []
        for i in range(cfg.layers):
            if i > 0:
                S.next_bucket()
            self.buckets.append(S._bucket)
            self.layers.append(ConformerLayer(S, env, f"{prefix}conformer_layers.{i}.", cfg.input_dim, cfg.feedforward_dim,
                                              cfg.heads, cfg.conv_kernel_size, cfg.dropout))

    def fwd(self, x, lens):
        saved = []
        for layer in self.layers:
            x, c = layer.fwd(x, lens)
            saved.append(c)
        return x, saved

    def bwd(self, dy, saved, layer_done=None):
        """``layer_done(bucket)`` is called after the backward of every layer whose bucket is complete with it (all
        but layer 0, whose bucket also holds what was declared before the stack)."""
        env = self.env
        group = H.GEMM_GROUP and 0 < dy.shape[0] * dy.shape[1] <= WGRAD_GROUP_ROWS and not H.holding_weight_gradients()
        for i in range(len(self.layers) - 1, -1, -1):
            if group:
                H.hold_weight_gradients(True)
            dy = self.layers[i].bwd(dy, saved[i])
            if group:  # the layer's weight gradients: one grouped launch, behind everything enqueued so far
                jobs = H.hold_weight_gradients(False)
                with env.side(*[t for j in jobs for t in j[:2]]):
                    with H.gemm_group():
                        for dy_, x_, out_, kw_ in jobs:
                            H.linear_bwd_weight(dy_, x_, out_, **kw_)
            if layer_done is not None and i > 0:
                layer_done(self.buckets[i])
        return dy


# ------------------------------------------------------------------------------------------------
# VariancePredictor  fs2/variance_adaptor.py:18-62, fs2/layers.py:20-48, fs2/blocks.py:4-19
# ------------------------------------------------------------------------------------------------
class VariancePredictor:
    def __init__(self, S, env: Env, prefix, d_in, cfg):
        self.S, self.env = S, env
        self.depthwise, self.k, self.p, self.c = cfg.depthwise, cfg.kernel_size, cfg.dropout, cfg.input_dim
        self.layers = []
        for i in range(cfg.n_layers):
            cin = d_in if i == 0 else cfg.input_dim
            q = f"{prefix}conv.{i}.layers."
            L = {"site": env.new_site(), "cin": cin}
            if cfg.depthwise:
                L["wd"], L["bd"] = q + "0.module.model.0.weight", q + "0.module.model.0.bias"
                L["wp"], L["bp"] = q + "0.module.model.1.weight", q + "0.module.model.1.bias"
                S.add(L["wd"], (cin, 1, self.k), "dw", P.init_linear_weight)
                S.add(L["bd"], (cin,), "id", P.init_bias_for(self.k))
                S.add(L["wp"], (self.c, cin, 1), "pw", P.init_linear_weight)
                S.add(L["bp"], (self.c,), "id", P.init_bias_for(cin))
            else:
                L["wc"], L["bc"] = q + "0.module.weight", q + "0.module.bias"
                S.add(L["wc"], (self.c, cin, self.k), "convk", P.init_linear_weight)
                S.add(L["bc"], (self.c,), "id", P.init_bias_for(cin * self.k))
            L["ln"] = LayerNorm(S, q + "2.", self.c)
            self.layers.append(L)
        self.wl, self.bl = prefix + "linear.weight", prefix + "linear.bias"
        S.add(self.wl, (1, self.c), "id", P.init_linear_weight)
        S.add(self.bl, (1,), "id", P.init_bias_for(self.c))

    def _stored(self, L) -> bool:
        """bf16 operand storage for this layer's pointwise GEMM (round 5): the depthwise convolution writes its result as
        the bf16 operand, the fused LayerNorm backward the bf16 gradient -- the values the register-rounding GEMMs of
        rounds 1-4 multiplied, without the fp32 tiles."""
        return self.env.stored and self.depthwise and PRED_STORED and self.c % 8 == 0 and L["cin"] % 8 == 0

    # One layer = three phases, so that several predictors can walk their layers in lockstep (``predictors_fwd`` /
    # ``predictors_bwd``) with the middle phase -- the GEMMs -- of all of them inside one ``H.gemm_group()``.
    def _fwd_conv(self, L, x, B, T):
        """Depthwise convolution in front of the pointwise GEMM (None: the layer is one k-tap convolution GEMM)."""
        S = self.S
        if self._stored(L):
            return H.dwconv_fwd(x, S.p(L["wd"]), S.p(L["bd"]), B, T, out_dtype=torch.bfloat16)[0]
        if self.depthwise:
            return H.dwconv_fwd(x, S.p(L["wd"]), S.p(L["bd"]), B, T)[0]
        return None

    def _fwd_gemm(self, L, x, c, T):
        S = self.S
        if c is not None and c.dtype == torch.bfloat16:
            return H.linear_fwd(c, S.pb(L["wp"]), S.p(L["bp"]), epi=H.EPI_ACT, act="relu")
        if c is not None:
            return H.linear_fwd(c, S.p(L["wp"]), S.p(L["bp"]), epi=H.EPI_ACT, act="relu")
        return H.linear_fwd(x, S.p(L["wc"]), S.p(L["bc"]), epi=H.EPI_ACT, act="relu", taps=self.k, T=T)

    def _fwd_norm(self, L, r):
        # LayerNorm + Dropout in one launch; the backward takes Dropout, LayerNorm and ReLU in one (norm.hip)
        ln = L["ln"]
        return H.layernorm_fwd_drop(r, self.S.p(ln.w), self.S.p(ln.b), self.env.drop(self.p, L["site"]))

    def fwd(self, x, lens):
        (pred,), (ctx,) = predictors_fwd([self], [x], [lens])
        return pred, ctx

    def _head_fwd(self, x, lens, B, T):
        return H.rowdot_fwd(x, self.S.p(self.wl), self.S.p(self.bl), lens, B, T)

    def _head_bwd(self, dpred, xl, lens, B, T):
        S = self.S
        return H.rowdot_bwd(dpred, xl, S.p(self.wl), lens, S.g(self.wl), S.g(self.bl), B, T)

    def _bwd_norm(self, L, d, c, r, ln_saved):
        """Dropout, LayerNorm and ReLU backward in one launch: the gradient of the GEMM's result (bf16 under operand
        storage, ``_stored``)."""
        S, ln = self.S, L["ln"]
        stored = c is not None and c.dtype == torch.bfloat16
        return H.layernorm_bwd_pred(d, r, S.p(ln.w), ln_saved[1], ln_saved[2], S.g(ln.w), S.g(ln.b),
                                    self.env.drop(self.p, L["site"]), out_dtype=torch.bfloat16 if stored else torch.float32)

    def _bwd_gemm(self, L, d, x, c, T):
        """Weight (+ bias) gradient and data gradient of the layer's GEMM; returns the data gradient."""
        S = self.S
        if c is not None and c.dtype == torch.bfloat16:
            H.linear_bwd_weight(d, c, S.g(L["wp"]), bias_grad=S.g(L["bp"]))
            return H.linear_bwd_data(d, S.pb(L["wp"]))
        if self.depthwise:
            H.linear_bwd_weight(d, c, S.g(L["wp"]), bias_grad=S.g(L["bp"]))
            return H.linear_bwd_data(d, S.p(L["wp"]))
        H.linear_bwd_weight(d, x, S.g(L["wc"]), taps=self.k, T=T, bias_grad=S.g(L["bc"]))
        return H.linear_bwd_data(d, S.p(L["wc"]), taps=self.k, T=T)

    def _bwd_conv(self, L, dc, x, B, T):
        if not self.depthwise:
            return dc
        S = self.S
        return H.dwconv_bwd(dc, x, S.p(L["wd"]), S.g(L["wd"]), S.g(L["bd"]), B, T)

    def bwd(self, dpred, ctx):
        return predictors_bwd([self], [dpred], [ctx])[0]

    def lockstep_key(self, x):
        """Predictors with equal keys can share ``predictors_fwd`` / ``predictors_bwd`` (same layer count and row count;
        widths may differ: a grouped launch takes members of any shape)."""
        return (len(self.layers), tuple(x.shape[:2]))


def predictors_fwd(preds, xs, lens):
    """Forward of several independent VariancePredictors (a training step feeds each its own input, built from TARGET
    embeddings: fs2/variance_adaptor.py:309-352) with equal ``lockstep_key``: layer by layer, the pointwise GEMMs of one
    layer in ONE grouped launch (``H.gemm_group``).  Alone, each of those launches is a quarter of a workgroup round
    (B x Ts rows = 256 workgroups of 64 x 64 at the benchmark shape) that takes ~13 us whatever its width.
    Returns ([prediction], [context]) in the order given; every tensor is what ``VariancePredictor.fwd`` alone computes."""
    B, T, _ = xs[0].shape
    cur, saved = list(xs), [[] for _ in preds]
    for li in range(len(preds[0].layers)):
        cs = [p._fwd_conv(p.layers[li], x, B, T) for p, x in zip(preds, cur)]
        with H.gemm_group():
            rs = [p._fwd_gemm(p.layers[li], x, c, T) for p, x, c in zip(preds, cur, cs)]
        for i, p in enumerate(preds):
            out, mean, rstd = p._fwd_norm(p.layers[li], rs[i])
            saved[i].append((cur[i], cs[i], rs[i], (rs[i], mean, rstd)))
            cur[i] = out
    out = [p._head_fwd(x, l, B, T) for p, x, l in zip(preds, cur, lens)]
    return out, [(sv, x, l) for sv, x, l in zip(saved, cur, lens)]


def predictors_bwd(preds, dpreds, ctxs):
    """Backward of ``predictors_fwd``: per layer the members' weight gradients in one grouped launch and their data
    gradients in another.  Returns the input gradients in the order given."""
    B, T, _ = ctxs[0][1].shape
    ds = [p._head_bwd(dp, xl, lens, B, T) for p, dp, (_, xl, lens) in zip(preds, dpreds, ctxs)]
    for li in range(len(preds[0].layers) - 1, -1, -1):
        lay = [(p.layers[li],) + ctx[0][li] for p, ctx in zip(preds, ctxs)]   # (L, x, c, r, ln_saved)
        ds = [p._bwd_norm(L, d, c, r, lns) for p, d, (L, x, c, r, lns) in zip(preds, ds, lay)]
        with H.gemm_group():
            dcs = [p._bwd_gemm(L, d, x, c, T) for p, d, (L, x, c, r, lns) in zip(preds, ds, lay)]
        ds = [p._bwd_conv(L, dc, x, B, T) for p, dc, (L, x, c, r, lns) in zip(preds, dcs, lay)]
    return ds


# ------------------------------------------------------------------------------------------------
# ConvAttention aligner  fs2/attn/attention.py:101-251 (built at fs2/variance_adaptor.py:151-158)
# ------------------------------------------------------------------------------------------------
class Aligner:
    """key_proj: Conv1d(n_text, 2 n_text, 3) -> ReLU -> Conv1d(2 n_text, n_att, 1);
    query_proj: Conv1d(n_mel, 2 n_mel, 3) -> ReLU -> Conv1d(2 n_mel, n_mel, 1) -> ReLU -> Conv1d(n_mel, n_att, 1);
    attn = -0.0005 * ||q - k||^2 -> log_softmax + log prior -> masked softmax -> MAS."""

    def __init__(self, S, env: Env, prefix, n_mel, n_text, n_att=80):
        self.S, self.env = S, env
        k, q = prefix + "key_proj.", prefix + "query_proj."
        self.names = dict(k0=(k + "0.conv.weight", k + "0.conv.bias"), k2=(k + "2.conv.weight", k + "2.conv.bias"),
                          q0=(q + "0.conv.weight", q + "0.conv.bias"), q2=(q + "2.conv.weight", q + "2.conv.bias"),
                          q4=(q + "4.conv.weight", q + "4.conv.bias"))

        def conv(key, cout, cin, ksz, gain):
            w, b = self.names[key]
            S.add(w, (cout, cin, ksz), "convk" if ksz > 1 else "pw", P.init_xavier(gain))
            S.add(b, (cout,), "id", P.init_bias_for(cin * ksz))

        conv("k0", 2 * n_text, n_text, 3, "relu")
        conv("k2", n_att, 2 * n_text, 1, "linear")
        conv("q0", 2 * n_mel, n_mel, 3, "relu")
        conv("q2", n_mel, 2 * n_mel, 1, "linear")
        conv("q4", n_att, n_mel, 1, "linear")

    def _w(self, key):
        w, b = self.names[key]
        return self.S.p(w), self.S.p(b)

    def fwd(self, mel, text_emb, prior, src_lens, mel_lens):
        B, Tm, _ = mel.shape
        Ts = text_emb.shape[1]
        k1 = H.linear_fwd(text_emb, *self._w("k0"), epi=H.EPI_ACT, act="relu", taps=3, T=Ts)
        kenc = H.linear_fwd(k1, *self._w("k2"))
        q1 = H.linear_fwd(mel, *self._w("q0"), epi=H.EPI_ACT, act="relu", taps=3, T=Tm)
        q2 = H.linear_fwd(q1, *self._w("q2"), epi=H.EPI_ACT, act="relu")
        qenc = H.linear_fwd(q2, *self._w("q4"))
        logits = H.attn_dist(qenc, kenc)
        logprob, soft = H.attn_softmax(logits, prior, src_lens)
        hard, hard_idx, dur = H.mas(soft, src_lens, mel_lens)
        ctx = Ctx(mel=mel, text_emb=text_emb, k1=k1, kenc=kenc, q1=q1, q2=q2, qenc=qenc, logits=logits, soft=soft,
                  hard_idx=hard_idx)
        return logprob, soft, hard, hard_idx, dur, ctx

    def bwd(self, dlogprob, bin_coef, c):
        """dlogprob: CTC gradient (or None); bin_coef: binarisation-loss coefficient (or None).
        Returns the gradient of text_emb."""
        S = self.S
        B, Tm, _ = c.mel.shape
        Ts = c.text_emb.shape[1]
        dlogits = H.attn_softmax_bwd(c.logits, c.soft, dlogprob, c.hard_idx if bin_coef is not None else None, bin_coef)
        dq, dk = H.attn_dist_bwd(dlogits, c.qenc, c.kenc)
        g = lambda key: (S.g(self.names[key][0]), S.g(self.names[key][1]))  # noqa: E731
        # query branch (the mel input needs no gradient)
        gw, gb = g("q4")
        H.linear_bwd_weight(dq, c.q2, gw, bias_grad=gb)
        d = H.linear_bwd_data(dq, self._w("q4")[0], epi=H.EPI_DACT, act="relu", aux=c.q2)
        gw, gb = g("q2")
        H.linear_bwd_weight(d, c.q1, gw, bias_grad=gb)
        d = H.linear_bwd_data(d, self._w("q2")[0], epi=H.EPI_DACT, act="relu", aux=c.q1)
        gw, gb = g("q0")
        H.linear_bwd_weight(d, c.mel, gw, taps=3, T=Tm, bias_grad=gb)
        # key branch
        gw, gb = g("k2")
        H.linear_bwd_weight(dk, c.k1, gw, bias_grad=gb)
        d = H.linear_bwd_data(dk, self._w("k2")[0], epi=H.EPI_DACT, act="relu", aux=c.k1)
        gw, gb = g("k0")
        H.linear_bwd_weight(d, c.text_emb, gw, taps=3, T=Ts, bias_grad=gb)
        return H.linear_bwd_data(d, self._w("k0")[0], taps=3, T=Ts)


# ------------------------------------------------------------------------------------------------
# GST style encoder  fs2/gst/model.py:14-280 (BASELINE config 5)
# ------------------------------------------------------------------------------------------------
class StyleEncoder:
    """ReferenceEncoder (6 x [Conv2d 3x3 s2 -> BatchNorm2d -> ReLU] -> GRU(128)) -> StyleTokenLayer (10 tanh'd
    tokens of 64 dims, 4-head attention, 256-dim output).  Channels-last throughout; the GRU input weights are
    stored with their columns permuted to the channels-last feature order."""
    CHANS, U, TOKENS, TOKEN_DIM, HEADS = (32, 32, 64, 64, 128, 128), 128, 10, 256, 4

    def __init__(self, S, env: Env, prefix, idim):
        self.S, self.env = S, env
        self.convs, cin, f = [], 1, idim
        for i, c in enumerate(self.CHANS):
            w = f"{prefix}ref_enc.convs.{3 * i}.weight"
            S.add(w, (c, cin, 3, 3), "conv2d", P.init_linear_weight)
            self.convs.append((w, BatchNorm(S, f"{prefix}ref_enc.convs.{3 * i + 1}.", c), c))
            cin, f = c, (f - 1) // 2 + 1
        if f != 2:
            raise NotImplementedError("GST: the reference encoder must leave 2 frequency bins (n_mels in 65..128)")
        g = prefix + "ref_enc.gru."
        U, ub = self.U, P.init_bias_for(self.U)
        self.wih, self.whh, self.bih, self.bhh = g + "weight_ih_l0", g + "weight_hh_l0", g + "bias_ih_l0", g + "bias_hh_l0"
        S.add(self.wih, (3 * U, f * cin), "gru_ih_w2", ub)
        S.add(self.whh, (3 * U, U), "id", ub)
        S.add(self.bih, (3 * U,), "id", ub)
        S.add(self.bhh, (3 * U,), "id", ub)
        self.embs = prefix + "stl.gst_embs"
        S.add(self.embs, (self.TOKENS, self.TOKEN_DIM // self.HEADS), "id", P.init_normal)
        m = prefix + "stl.mha."
        self.lin = {}
        for name, n_in in (("q", U), ("k", self.TOKEN_DIM // self.HEADS), ("v", self.TOKEN_DIM // self.HEADS),
                           ("out", self.TOKEN_DIM)):
            self.lin[name] = (f"{m}linear_{name}.weight", f"{m}linear_{name}.bias")
            decl_linear(S, f"{m}linear_{name}.", self.TOKEN_DIM, n_in)

    def _lw(self, name):
        w, b = self.lin[name]
        return self.S.p(w), self.S.p(b)

    def fwd(self, mel):
        S, env, U = self.S, self.env, self.U
        B, T, F = mel.shape
        x = mel.view(B, T, F, 1)
        conv_saved = []
        for w, bn, c in self.convs:
            raw = H.conv2d_s2_fwd(x, S.p(w))
            rows = raw.numel() // c
            stats = bn.stats(H.colstats(raw.view(rows, c)) if env.training else None, env.training)
            y = H.bn_act_fwd(raw.view(rows, c), stats, "relu").view(raw.shape)
            conv_saved.append((x, raw, stats))
            x = y
        _, Hh, Ww, C = x.shape
        feat = x.view(B * Hh, Ww * C)
        gi = H.linear_fwd(feat, S.p(self.wih), S.p(self.bih))            # rows (b, t)
        hs = H.zeros(Hh + 1, B, U, device=mel.device)
        gates = []
        for t in range(Hh):
            gh = H.linear_fwd(hs[t], S.p(self.whh), S.p(self.bhh))
            _, g = H.gru_gate_fwd(gi.view(-1)[t * 3 * U:], Hh * 3 * U, gh, hs[t], U, hnew=hs[t + 1])
            gates.append(g)
        ref = hs[Hh]
        q = H.linear_fwd(ref, *self._lw("q"))
        tk = H.act_apply(S.p(self.embs), "tanh")
        k = H.linear_fwd(tk, *self._lw("k"))
        v = H.linear_fwd(tk, *self._lw("v"))
        p, ctx = H.gst_attn_fwd(q, k, v, self.HEADS)
        style = H.linear_fwd(ctx, *self._lw("out"))
        return style, Ctx(conv=conv_saved, feat=feat, hs=hs, gates=gates, q=q, tk=tk, k=k, v=v, p=p, ctx=ctx,
                          shape=(B, Hh, Ww, C))

    def condition_on_gst_tokens(self, batch_size: int, index: int = 0):
        """reference ``fs2/gst/model.py:77-85`` (free inference without a reference mel): attention over the single
        token ``index`` -- its softmax weight is 1, so the style is ``linear_out(linear_v(tanh(gst_embs[index])))``
        for every utterance, whatever the (zero) query."""
        if index >= self.TOKENS:
            raise ValueError(f"We can only synthesize by conditioning on one of {self.TOKENS} GST tokens")
        S = self.S
        tk = H.act_apply(S.p(self.embs)[index:index + 1].contiguous(), "tanh")
        style = H.linear_fwd(H.linear_fwd(tk, *self._lw("v")), *self._lw("out"))
        return style.expand(batch_size, -1).contiguous()

    def bwd(self, d_style, c):
        S, env, U = self.S, self.env, self.U
        B, Hh, Ww, C = c.shape
        g = lambda name: (S.g(self.lin[name][0]), S.g(self.lin[name][1]))  # noqa: E731
        gw, gb = g("out")
        H.linear_bwd_weight(d_style, c.ctx, gw, bias_grad=gb)
        dctx = H.linear_bwd_data(d_style, self._lw("out")[0])
        dq, dkp, dvp = H.gst_attn_bwd(dctx, c.q, c.k, c.v, c.p, self.HEADS)
        NT, Fd = c.k.shape
        dk = torch.empty(NT, Fd, device=dq.device, dtype=torch.float32)
        dv = torch.empty_like(dk)
        H.colsum(dkp.view(B, NT * Fd), dk.view(-1))
        H.colsum(dvp.view(B, NT * Fd), dv.view(-1))
        gw, gb = g("k")
        H.linear_bwd_weight(dk, c.tk, gw, bias_grad=gb)
        dtk = H.linear_bwd_data(dk, self._lw("k")[0])
        gw, gb = g("v")
        H.linear_bwd_weight(dv, c.tk, gw, bias_grad=gb)
        dtk = H.axpby(dtk, H.linear_bwd_data(dv, self._lw("v")[0]))
        H.axpby(H.dact_mul(dtk, S.p(self.embs), "tanh"), None, 1.0, 0.0, out=S.g(self.embs))
        gw, gb = g("q")
        H.linear_bwd_weight(dq, c.hs[Hh], gw, bias_grad=gb)
        dh = H.linear_bwd_data(dq, self._lw("q")[0])
        # GRU backward through time
        dgi = torch.empty(B * Hh, 3 * U, device=dh.device, dtype=torch.float32)
        dgh_all = torch.empty(Hh, B, 3 * U, device=dh.device, dtype=torch.float32)
        for t in range(Hh - 1, -1, -1):
            _, dhprev = H.gru_gate_bwd(dh, c.gates[t], c.hs[t], dgi.view(-1)[t * 3 * U:], Hh * 3 * U, U, dgh=dgh_all[t])
            dh = H.axpby(dhprev, H.linear_bwd_data(dgh_all[t], S.p(self.whh)))
        H.linear_bwd_weight(dgh_all.view(Hh * B, 3 * U), c.hs[:Hh].reshape(Hh * B, U), S.g(self.whh), bias_grad=S.g(self.bhh))
        H.linear_bwd_weight(dgi, c.feat, S.g(self.wih), bias_grad=S.g(self.bih))
        d = H.linear_bwd_data(dgi, S.p(self.wih)).view(B, Hh, Ww, C)
        for i in range(len(self.convs) - 1, -1, -1):
            w, bn, ch = self.convs[i]
            x, raw, stats = c.conv[i]
            rows = raw.numel() // ch
            gg, gbb = bn.grads()
            draw = H.bn_act_bwd(d.reshape(rows, ch), raw.view(rows, ch), stats, gg, gbb, "relu", training=env.training)
            d = H.conv2d_s2_bwd(draw.view(raw.shape), x, S.p(w), S.g(w), need_dx=i > 0)


# ------------------------------------------------------------------------------------------------
# PostNet  fs2/layers.py:143-212
# ------------------------------------------------------------------------------------------------
class PostNet:
    DROPOUT = 0.5  # hard-coded in the reference (fs2/layers.py:207-209)

    def __init__(self, S, env: Env, prefix, n_mel, dim=512, k=5, n=5):
        self.S, self.env, self.k, self.n, self.n_mel = S, env, k, n, n_mel
        self.dropout_p = self.DROPOUT
        chans = [n_mel] + [dim] * (n - 1) + [n_mel]
        self.convs = []
        for i in range(n):
            q = f"{prefix}convolutions.{i}."
            w, b = q + "0.conv.weight", q + "0.conv.bias"
            S.add(w, (chans[i + 1], chans[i], k), "convk", P.init_xavier("tanh" if i < n - 1 else "linear"))
            S.add(b, (chans[i + 1],), "id", P.init_bias_for(chans[i] * k))
            self.convs.append((w, b, BatchNorm(S, q + "1.", chans[i + 1]), env.new_site()))

    def fwd(self, x):
        """bf16 operand storage (``Env.stored``; needs T >= 64 for the weight-gradient form): every 512-channel
        activation exists only as the bf16 tensor the next convolution reads -- written by the BatchNorm + tanh +
        dropout kernel -- and the convolutions read the weights from the bf16 mirror.  The first convolution's input
        (80 mel bins: not whole 64-deep K-tiles per tap) stays on the fp32-operand core."""
        S, env = self.S, self.env
        B, T, _ = x.shape
        # (the first layer's weight gradient reads the mel input cast to bf16 (K = n_mel), the last layer's reads a bf16
        # output gradient with N = n_mel rows: both need n_mel % 8 == 0 -- 100 mel bands run with fp32-stored operands
        # rounded in registers, like every GEMM outside the operand-storage mode)
        stored = env.stored and T >= 64 and self.n_mel % 8 == 0
        saved = []
        for i, (w, b, bn, site) in enumerate(self.convs):
            if x.dtype == torch.bfloat16:
                # a convolution between two bf16-only layers writes a bf16 result as well (what autocast's conv1d
                # returns): BatchNorm's statistics and both of its passes then read half the bytes.  The last
                # convolution's result meets the loss gradient in fp32 and stays fp32.
                raw_dt = torch.bfloat16 if i + 1 < self.n and BF16_CHAIN else torch.float32
                raw = H.linear_fwd(x.view(B * T, -1), S.pb(w), S.p(b), taps=self.k, T=T, out_dtype=raw_dt).view(B, T, -1)
            elif stored and POSTNET_IM2COL and x.shape[-1] % 8 == 0 and x.shape[-1] % 64 != 0:
                # round 5: the first layer's 80 mel bins are not whole K-tiles per tap, which kept this convolution on
                # the fp32-operand tiles (105 us at 168 TFLOP/s in the bf16 step).  Its input rows laid side by side per
                # tap (one 16-byte-per-thread gather, bf16) make it ONE plain K = 5 x 80 GEMM on the bf16-storage core
                # against the weight transposed per tap: the same products, rounded as before (bf16 operands), summed
                # over (tap, channel) in the same order.
                raw_dt = torch.bfloat16 if i + 1 < self.n and BF16_CHAIN else torch.float32
                xc = H.im2col_taps(x.view(B * T, -1), B, T, self.k)
                wt = H.transpose_cast_bf16(S.p(w)).view(self.k * x.shape[-1], -1)   # [taps][Cout][Cin] -> [taps * Cin][Cout]
                raw = H.matmul_kn(xc, wt, S.p(b), out_dtype=raw_dt).view(B, T, -1)
            else:
                raw = H.linear_fwd(x, S.p(w), S.p(b), taps=self.k, T=T)
            stats = bn.stats(H.colstats(raw) if env.training else None, env.training)
            act = "tanh" if i < self.n - 1 else None
            want_b = stored and i + 1 < self.n and raw.shape[-1] % 64 == 0
            out = H.bn_act_fwd(raw, stats, act, env.drop(self.dropout_p, site), bf16_only=want_b)
            saved.append((x, raw, stats))
            x = out
        return x, saved

    def bwd(self, dy, saved, need_dx=True):
        S, env = self.S, self.env
        B, T, _ = dy.shape
        for i in range(self.n - 1, -1, -1):
            w, b, bn, site = self.convs[i]
            x, raw, stats = saved[i]
            gg, gb = bn.grads()
            act = "tanh" if i < self.n - 1 else None
            need = i > 0 or need_dx
            cout = raw.shape[-1]
            drop = env.drop(self.dropout_p, site)
            # the bf16 forms follow what the forward pass kept: a layer whose input (or whose successor's input) is
            # bf16 takes its output gradient as bf16 from the BatchNorm backward kernel
            nxt_b = i + 1 < self.n and saved[i + 1][0].dtype == torch.bfloat16   # this layer's output was bf16-only
            x_b = x.dtype == torch.bfloat16
            if not (nxt_b or x_b):
                draw = H.bn_act_bwd(dy, raw, stats, gg, gb, act, drop, training=env.training)
                env.wgrad(draw, x, S.g(w), taps=self.k, T=T, bias_grad=S.g(b))
                if need:
                    dy = H.linear_bwd_data(draw, S.p(w), taps=self.k, T=T)
                continue
            dgrad_b = need and cout % 64 == 0 and x.shape[-1] % 8 == 0
            # (round 5) the last layer's data gradient reduces over taps x 80 output channels: the same side-by-side rows,
            # taken backwards in time, against the weight as it is stored
            dgrad_im2col = need and not dgrad_b and POSTNET_IM2COL and cout % 8 == 0 and x.shape[-1] % 8 == 0
            if dgrad_im2col:
                draw, draw_b = None, H.bn_act_bwd(dy, raw, stats, gg, gb, act, drop, training=env.training, bf16_only=True)
            elif need and not dgrad_b:
                draw, draw_b = H.bn_act_bwd(dy, raw, stats, gg, gb, act, drop, training=env.training, bf16_copy=True)
            else:
                draw, draw_b = None, H.bn_act_bwd(dy, raw, stats, gg, gb, act, drop, training=env.training, bf16_only=True)
            xb = x if x_b else H.cast_bf16(x)  # (first layer: the 80-bin input, cast for the weight gradient only)
            with env.side(draw_b, xb):
                H.linear_bwd_weight(draw_b.view(B * T, -1), xb.view(B * T, -1), S.g(w), taps=self.k, T=T, bias_grad=S.g(b))
            if need and dgrad_b:
                # (the gradient enters the layer below through its BatchNorm backward, which reads it beside that
                # layer's convolution result: both bf16 or both fp32)
                dy_dt = torch.bfloat16 if i > 0 and saved[i - 1][1].dtype == torch.bfloat16 else torch.float32
                dy = H.linear_bwd_data(draw_b.view(B * T, -1), S.pb(w), taps=self.k, T=T, out_dtype=dy_dt).view(B, T, -1)
            elif dgrad_im2col:
                dy_dt = torch.bfloat16 if i > 0 and saved[i - 1][1].dtype == torch.bfloat16 else torch.float32
                dyc = H.im2col_taps(draw_b.view(B * T, -1), B, T, self.k, direction=-1)
                dy = H.matmul_kn(dyc, S.pb(w).view(self.k * cout, -1), out_dtype=dy_dt).view(B, T, -1)
            elif need:
                dy = H.linear_bwd_data(draw, S.p(w), taps=self.k, T=T)
        return dy
