"""ctypes binding of the fs2hip C ABI (include/fs2hip.h).

This is the only way the product reaches the GPU kernels: there is no CPU or
PyTorch fallback.  ``lib()`` raises if ``_fs2hip.so`` is missing, and every
wrapper checks on the host that shapes, strides, dtypes and devices match what
the kernel's grid assumes before launching (a faulting kernel can take the
whole node down).  Launches go to ``torch.cuda.current_stream()``.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional

import torch

_LIB_PATH = Path(__file__).resolve().parent / "_fs2hip.so"
_lib = None

ACT_NONE, ACT_RELU, ACT_SILU, ACT_TANH = 0, 1, 2, 3
EPI_STORE, EPI_ACT, EPI_RESID, EPI_DACT = 0, 1, 2, 3
_ACT = {None: 0, "none": 0, "relu": 1, "silu": 2, "tanh": 3}

_f32p = C.c_void_p
_i32p = C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("Mc", C.c_int), ("Nc", C.c_int), ("R", C.c_int),
        ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int),
        ("a_kcontig", C.c_int), ("b_kcontig", C.c_int),
        ("taps", C.c_int), ("T", C.c_int), ("tap_mul", C.c_int), ("tap_add", C.c_int),
        ("shift_operand", C.c_int),
        ("b_tap_stride", C.c_longlong), ("c_tap_stride", C.c_longlong),
        ("bias", C.c_void_p),
        ("epi", C.c_int), ("act", C.c_int),
        ("alpha", C.c_float),
        ("resid", C.c_void_p), ("ldr", C.c_int), ("res_scale", C.c_float),
        ("aux", C.c_void_p), ("ldaux", C.c_int),
        ("out_pre", C.c_void_p), ("ldpre", C.c_int),
        ("drop_p", C.c_float), ("drop_seed", C.c_ulonglong),
        ("splitk", C.c_int), ("workspace", C.c_void_p),
    ]


def lib():
    """The loaded shared library.  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise RuntimeError(
                f"{_LIB_PATH} is missing: build it with `python -m fastspeech2_lightning_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no fallback path.")
        _lib = C.CDLL(str(_LIB_PATH))
        _declare(_lib)
    return _lib


def _declare(L):
    L.fs2hip_version.restype = C.c_int
    for name in EXPORTS:
        getattr(L, name).restype = C.c_int
    L.fs2hip_gemm.argtypes = [C.POINTER(GemmArgs), C.c_void_p]


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, dtype=torch.float32, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"fs2hip: {name} must live in GPU memory (got {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"fs2hip: {name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"fs2hip: {name} must be contiguous")
    return t


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _ok(code: int, what: str):
    if code != 0:
        raise RuntimeError(f"fs2hip: {what} failed with code {code}"
                           + (" (invalid arguments)" if code == -22 else ""))


def _rows(t: torch.Tensor) -> int:
    return t.numel() // t.shape[-1]


# ------------------------------------------------------------------------------------------
# GEMM family
# ------------------------------------------------------------------------------------------
def _gemm(**kw):
    a = GemmArgs()
    a.taps, a.alpha, a.res_scale, a.splitk = 1, 1.0, 1.0, 1
    for k, v in kw.items():
        setattr(a, k, v)
    _ok(lib().fs2hip_gemm(C.byref(a), C.c_void_p(_stream())), "gemm")


def linear_fwd(x, w, bias=None, *, epi=EPI_STORE, act=None, resid=None, res_scale=1.0, out_pre=None,
               drop_p=0.0, drop_seed=0, taps=1, T=0, out=None):
    """y[M, N] = epi(x[M, K*] @ w^T + bias).  ``w`` is [N, K] (taps == 1) or
    [taps, N, Kper] for a k-tap convolution over time (rows of x are (b, t), 'same' padding)."""
    _chk(x, name="x"); _chk(w, name="w")
    M, Kper = _rows(x), x.shape[-1]
    if taps == 1:
        N, K = w.shape
        if K != Kper:
            raise ValueError(f"linear_fwd: x has {Kper} columns, w has {K}")
    else:
        if w.dim() != 3 or w.shape[0] != taps or w.shape[2] != Kper or M % T:
            raise ValueError("linear_fwd: conv weight must be [taps, N, Kper] and rows a multiple of T")
        N = w.shape[1]
    if out is None:
        out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
    _chk(out, name="out")
    if _rows(out) != M or out.shape[-1] != N:
        raise ValueError("linear_fwd: bad output shape")
    kw = dict(A=_p(x), B=_p(w), C=_p(out), Mc=M, Nc=N, R=Kper * taps, lda=Kper, ldb=Kper, ldc=N,
              a_kcontig=1, b_kcontig=1, taps=taps, T=T, tap_mul=1, tap_add=-((taps - 1) // 2), shift_operand=0,
              b_tap_stride=N * Kper, epi=epi, act=_ACT[act], drop_p=float(drop_p), drop_seed=int(drop_seed))
    if bias is not None:
        _chk(bias, name="bias")
        if bias.numel() != N:
            raise ValueError("linear_fwd: bias size")
        kw["bias"] = _p(bias)
    if epi == EPI_RESID:
        _chk(resid, name="resid")
        if resid.shape != out.shape:
            raise ValueError("linear_fwd: residual shape")
        kw.update(resid=_p(resid), ldr=N, res_scale=float(res_scale))
    if out_pre is not None:
        _chk(out_pre, name="out_pre")
        if out_pre.shape != out.shape:
            raise ValueError("linear_fwd: out_pre shape")
        kw.update(out_pre=_p(out_pre), ldpre=N)
    _gemm(**kw)
    return out


def linear_bwd_data(dy, w, *, epi=EPI_STORE, act=None, aux=None, alpha=1.0, drop_p=0.0, drop_seed=0,
                    taps=1, T=0, out=None):
    """dx[M, K] = epi(alpha * dy[M, N] @ w) with w [N, K] (or [taps, N, Kper], transposed conv)."""
    _chk(dy, name="dy"); _chk(w, name="w")
    M, N = _rows(dy), dy.shape[-1]
    if taps == 1:
        if w.shape[0] != N:
            raise ValueError("linear_bwd_data: dy columns != w rows")
        K = w.shape[1]
    else:
        if w.dim() != 3 or w.shape[0] != taps or w.shape[1] != N or M % T:
            raise ValueError("linear_bwd_data: conv weight must be [taps, N, Kper]")
        K = w.shape[2]
    if out is None:
        out = torch.empty(*dy.shape[:-1], K, device=dy.device, dtype=torch.float32)
    _chk(out, name="out")
    if _rows(out) != M or out.shape[-1] != K:
        raise ValueError("linear_bwd_data: bad output shape")
    kw = dict(A=_p(dy), B=_p(w), C=_p(out), Mc=M, Nc=K, R=N * taps, lda=N, ldb=K, ldc=K,
              a_kcontig=1, b_kcontig=0, taps=taps, T=T, tap_mul=-1, tap_add=(taps - 1) // 2, shift_operand=0,
              b_tap_stride=N * K, epi=epi, act=_ACT[act], alpha=float(alpha),
              drop_p=float(drop_p), drop_seed=int(drop_seed))
    if epi == EPI_DACT:
        _chk(aux, name="aux")
        if aux.shape != out.shape:
            raise ValueError("linear_bwd_data: aux shape")
        kw.update(aux=_p(aux), ldaux=K)
    _gemm(**kw)
    return out


_WS = {}


def _workspace(n: int, device) -> torch.Tensor:
    """Split-K slab workspace, grown on demand and reused (stream-ordered reuse is safe: every
    user finishes with a reduce on the same stream before the next GEMM writes it)."""
    key = (device, torch.cuda.current_stream().cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.empty(max(n, 1 << 22), device=device, dtype=torch.float32)
        _WS[key] = ws
    return ws


def pick_splitk(Mc: int, Nc: int, R: int, taps: int = 1) -> int:
    tiles = ((Mc + 127) // 128) * ((Nc + 63) // 64) * taps
    s = max(1, min(64, 512 // max(tiles, 1)))
    while s > 1 and R // s < 256:
        s //= 2
    return s


def linear_bwd_weight(dy, x, out, *, taps=1, T=0):
    """dw[N, K] = dy[M, N]^T @ x[M, K]  (or dw[taps, N, Kper] for the k-tap conv).
    Written into ``out`` (a view of the flat gradient buffer)."""
    _chk(dy, name="dy"); _chk(x, name="x"); _chk(out, name="out")
    M, N, K = _rows(dy), dy.shape[-1], x.shape[-1]
    if _rows(x) != M:
        raise ValueError("linear_bwd_weight: row mismatch")
    if out.numel() != taps * N * K:
        raise ValueError("linear_bwd_weight: bad gradient shape")
    if taps > 1 and M % T:
        raise ValueError("linear_bwd_weight: rows must be a multiple of T")
    S = pick_splitk(N, K, M, taps)
    kw = dict(A=_p(dy), B=_p(x), C=_p(out), Mc=N, Nc=K, R=M, lda=N, ldb=K, ldc=K, a_kcontig=0, b_kcontig=0,
              taps=taps, T=T if taps > 1 else 0, tap_mul=1, tap_add=-((taps - 1) // 2), shift_operand=1 if taps > 1 else 0,
              c_tap_stride=N * K, splitk=S)
    if S > 1:
        ws = _workspace(S * taps * N * K, dy.device)
        kw["workspace"] = _p(ws)
        _gemm(**kw)
        n = taps * N * K
        _ok(lib().fs2hip_reduce_slabs(C.c_void_p(_p(ws)), C.c_void_p(_p(out)), C.c_longlong(n), C.c_int(S),
                                      C.c_longlong(n), C.c_void_p(_stream())), "reduce_slabs")
    else:
        _gemm(**kw)
    return out


def colsum(x, out):
    """out[N] = sum over rows of x[M, N] (bias gradients)."""
    _chk(x, name="x"); _chk(out, name="out")
    M, N = _rows(x), x.shape[-1]
    if out.numel() != N:
        raise ValueError("colsum: bad output size")
    gy = lib().fs2hip_colsum_rows(C.c_int(M))
    ws = _workspace(gy * N, x.device)
    _ok(lib().fs2hip_colsum(C.c_void_p(_p(x)), C.c_int(N), C.c_int(M), C.c_int(N), C.c_void_p(_p(ws)),
                            C.c_void_p(_p(out)), C.c_void_p(_stream())), "colsum")
    return out


# ------------------------------------------------------------------------------------------
# LayerNorm
# ------------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps=1e-5):
    _chk(x, name="x"); _chk(gamma, name="gamma"); _chk(beta, name="beta")
    M, Cc = _rows(x), x.shape[-1]
    if gamma.numel() != Cc or beta.numel() != Cc:
        raise ValueError("layernorm_fwd: parameter size")
    y = torch.empty_like(x)
    mean = torch.empty(M, device=x.device, dtype=torch.float32)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    _ok(lib().fs2hip_layernorm_fwd(C.c_void_p(_p(x)), C.c_void_p(_p(gamma)), C.c_void_p(_p(beta)), C.c_void_p(_p(y)),
                                   C.c_void_p(_p(mean)), C.c_void_p(_p(rstd)), C.c_int(M), C.c_int(Cc),
                                   C.c_float(eps), C.c_void_p(_stream())), "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dx_add=None):
    """Returns dx (+ dx_add); writes dgamma/dbeta."""
    for n, t in (("dy", dy), ("x", x), ("gamma", gamma), ("mean", mean), ("rstd", rstd), ("dgamma", dgamma), ("dbeta", dbeta)):
        _chk(t, name=n)
    M, Cc = _rows(x), x.shape[-1]
    if dy.shape != x.shape or mean.numel() != M or rstd.numel() != M or dgamma.numel() != Cc or dbeta.numel() != Cc:
        raise ValueError("layernorm_bwd: shape mismatch")
    if dx_add is not None:
        _chk(dx_add, name="dx_add")
        if dx_add.shape != x.shape:
            raise ValueError("layernorm_bwd: dx_add shape")
    dx = torch.empty_like(x)
    nblk = lib().fs2hip_layernorm_bwd_blocks(C.c_int(M))
    ws = _workspace(nblk * 2 * Cc, x.device)
    _ok(lib().fs2hip_layernorm_bwd(C.c_void_p(_p(dy)), C.c_void_p(_p(x)), C.c_void_p(_p(gamma)), C.c_void_p(_p(mean)),
                                   C.c_void_p(_p(rstd)), C.c_void_p(_p(dx_add)), C.c_void_p(_p(dx)), C.c_void_p(_p(ws)),
                                   C.c_void_p(_p(dgamma)), C.c_void_p(_p(dbeta)), C.c_int(M), C.c_int(Cc),
                                   C.c_void_p(_stream())), "layernorm_bwd")
    return dx


#: every symbol include/fs2hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "fs2hip_version", "fs2hip_gemm", "fs2hip_reduce_slabs", "fs2hip_colsum_rows", "fs2hip_colsum",
    "fs2hip_layernorm_fwd", "fs2hip_layernorm_bwd_blocks", "fs2hip_layernorm_bwd",
]
