"""Host-side cost of the pieces of one launch through hip.py (GPU box: needs cuda tensors), microseconds per call."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402
import ctypes as C  # noqa: E402

H.lib()
x = torch.randn(4096, 256, device="cuda")
w = torch.randn(256, 256, device="cuda")
N = 20000


def t(label, fn, n=N):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    print(f"{label:46s} {dt:7.2f} us", flush=True)


t("empty loop body (lambda call)", lambda: None)
t("_chk(x)", lambda: H._chk(x))
t("_p(x)", lambda: H._p(x))
t("_rows(x)", lambda: H._rows(x))
t("_stream()", H._stream)
t("_current_device()", H._current_device)
t("x.device.index", lambda: x.device.index)
t("x.is_contiguous()", x.is_contiguous)
t("torch.empty(4096, 256, cuda)", lambda: torch.empty(4096, 256, device=x.device))
t("torch.empty_like(x)", lambda: torch.empty_like(x))
t("GemmArgs()", H.GemmArgs)
kw = dict(A=H._p(x), B=H._p(w), C=H._p(x), Mc=4096, Nc=256, R=256, lda=256, ldb=256, ldc=256, a_kcontig=1, b_kcontig=1, taps=1, T=0,
          tap_mul=1, tap_add=0, shift_operand=0, epi=0, act=0, alpha=1.0, drop_p=0.0, drop_seed=0, drop_step=None)


def fill():
    a = H.GemmArgs()
    for k, v in kw.items():
        setattr(a, k, v)
    return a


t("GemmArgs() + 24 setattr", fill)
a = fill()
a.operand_bf16 = 0
t("_tile_key(a)", lambda: H._tile_key(a))
a.tile = 99  # refused at once: the cost of the ctypes call itself
L = H.lib()
s = H._stream()
t("ctypes fs2hip_gemm (refused tile: call overhead)", lambda: L.fs2hip_gemm(C.byref(a), s))
a.tile = 7
torch.cuda.synchronize()
t("ctypes fs2hip_gemm (tile 7, real launch)", lambda: L.fs2hip_gemm(C.byref(a), s), 3000)
torch.cuda.synchronize()
t("H.linear_fwd(x, w) whole wrapper", lambda: H.linear_fwd(x, w), 3000)
torch.cuda.synchronize()
g, b = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
t("H.layernorm_fwd whole wrapper", lambda: H.layernorm_fwd(x, g, b), 3000)
torch.cuda.synchronize()
ev = torch.cuda.Event()
t("torch.cuda.Event() + record + wait", lambda: (lambda e: (e.record(), torch.cuda.current_stream().wait_event(e)))(torch.cuda.Event()), 3000)
t("torch.cuda.current_stream()", torch.cuda.current_stream)
