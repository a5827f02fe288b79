"""The step's main GEMM shapes in one precision, tuned tiles (GPU only).
usage: python tools/bench_gemm_modes.py [32-true|32-split|bf16-mixed|bf16-stored]   (bf16-stored: operands already
bf16 in memory, Fs2GemmArgs.operand_bf16 == 3; the weight-gradient column then shows the register-rounding mode)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "32-true"
stored = prec == "bf16-stored"
H.set_precision("bf16-mixed" if stored else prec)
dev = "cuda"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


M = 20736
rows = []
for name, N, K in (("ffn1", 1024, 256), ("ffn2", 256, 1024), ("qkv", 768, 256), ("proj", 256, 256), ("enc ffn1 (M=4096)", 1024, 256)):
    m = 4096 if "enc" in name else M
    x, w, b = torch.randn(m, K, device=dev), torch.randn(N, K, device=dev) * K ** -0.5, torch.randn(N, device=dev)
    dy = torch.randn(m, N, device=dev)
    dw = torch.empty(N * K, device=dev)
    fl = 2.0 * m * N * K
    if stored:
        xb, wb, dyb, wt = H.cast_bf16(x), H.cast_bf16(w), H.cast_bf16(dy), H.transpose_cast_bf16(w)
        tf = timeit(lambda: H.linear_fwd(xb, wb, b, epi=H.EPI_ACT, act="silu", drop=H.Drop(0.1, 5)))
        td = timeit(lambda: H.linear_bwd_data(dyb, wt))
        tc = timeit(lambda: H.cast_bf16(dy)) if name == "ffn1" else None
        if tc:
            print(f"  (cast of a {m}x{N} fp32 tensor to bf16: {tc * 1e6:.1f} us)")
    else:
        tf = timeit(lambda: H.linear_fwd(x, w, b, epi=H.EPI_ACT, act="silu", drop=H.Drop(0.1, 5)))
        td = timeit(lambda: H.linear_bwd_data(dy, w))
    tw = timeit(lambda: H.linear_bwd_weight(dy, x, dw))
    rows.append((name, m, N, K, fl / tf / 1e12, fl / td / 1e12, fl / tw / 1e12, (tf + td + tw) * 1e6))
T, B, C = 648, 32, 512
x, w, b = torch.randn(B * T, C, device=dev), torch.randn(5, C, C, device=dev) * (5 * C) ** -0.5, torch.randn(C, device=dev)
dy, dw = torch.randn(B * T, C, device=dev), torch.empty(5 * C * C, device=dev)
fl = 2.0 * B * T * C * C * 5
if stored:
    xb, wb = H.cast_bf16(x), H.cast_bf16(w)
    tf = timeit(lambda: H.linear_fwd(xb, wb, b, taps=5, T=T))
else:
    tf = timeit(lambda: H.linear_fwd(x, w, b, taps=5, T=T))
tw = timeit(lambda: H.linear_bwd_weight(dy, x, dw, taps=5, T=T))
rows.append(("postnet conv k5", B * T, C, 5 * C, fl / tf / 1e12, float("nan"), fl / tw / 1e12, (tf + tw) * 1e6))
print(f"precision {prec}: TFLOP/s (fp32-equivalent) fwd(+silu+dropout) / bwd-data / bwd-weight, total us")
for r in rows:
    print(f"  {r[0]:20s} M={r[1]:6d} N={r[2]:5d} K={r[3]:5d}: {r[4]:6.1f} {r[5]:6.1f} {r[6]:6.1f}   {r[7]:7.1f} us")
