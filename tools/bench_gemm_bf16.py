"""The bf16-storage GEMM core (operand_bf16 == 4) on the step's shapes at BASELINE configs[2] (batch 64: 41 472 decoder
rows), per tile: microseconds, TFLOP/s and algorithmic GB/s (operands + results once).  GPU only.
usage: python tools/bench_gemm_bf16.py [rows]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402

H.set_precision("bf16-mixed")
dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 41472
bf = torch.bfloat16


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


def sweep(name, flops, nbytes, fn):
    cells = []
    for tile in H_TILES:
        H.GEMM_TILES_B = (tile,)
        H._TILE_CACHE.clear()
        try:
            t = timeit(fn)
            cells.append(f"{tile}: {t * 1e6:7.1f} us {flops / t / 1e12:6.0f} TF {nbytes / t / 1e9:5.0f} GB/s")
        except Exception as e:  # a tile that refuses the shape
            cells.append(f"{tile}: refused")
    print(f"  {name:34s} | " + " | ".join(cells))


H_TILES = H.GEMM_TILES_B
step = torch.zeros(1, dtype=torch.int64, device=dev)
drop = H.Drop(0.2, 5, step)
print(f"rows = {M}")
for name, N, K in (("ffn1", 1024, 256), ("ffn2", 256, 1024), ("qkv", 768, 256), ("proj", 256, 256), ("pw1", 512, 256)):
    x = torch.randn(M, K, device=dev).to(bf)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
    b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev).to(bf)
    res = torch.randn(M, N, device=dev)
    u = torch.empty(M, N, device=dev, dtype=bf)
    dw, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
    fl = 2.0 * M * N * K
    if name == "ffn1":
        sweep(f"{name} fwd silu+drop, bf16 u and a", fl, 2.0 * (M * K + N * K) + 4.0 * M * N,
              lambda: H.linear_fwd(x, w, b, epi=H.EPI_ACT, act="silu", out_pre=u, drop=drop, out_dtype=bf))
        aux = torch.randn(M, K, device=dev).to(bf)
    elif name == "ffn2":
        sweep(f"{name} fwd resid+drop, fp32 y", fl, 2.0 * (M * K + N * K) + 8.0 * M * N,
              lambda: H.linear_fwd(x, w, b, epi=H.EPI_RESID, resid=res, res_scale=0.5, drop=drop))
    else:
        sweep(f"{name} fwd, fp32 out", fl, 2.0 * (M * K + N * K) + 4.0 * M * N, lambda: H.linear_fwd(x, w, b))
    if name == "ffn2":
        auxu = torch.randn(M, K, device=dev).to(bf)
        sweep(f"{name} dgrad silu'+drop, bf16 du", fl, 2.0 * (M * N + N * K) + 4.0 * M * K,
              lambda: H.linear_bwd_data(dy, w, epi=H.EPI_DACT, act="silu", aux=auxu, drop=drop, out_dtype=bf))
    else:
        sweep(f"{name} dgrad, bf16 dh", fl, 2.0 * (M * N + N * K) + 2.0 * M * K, lambda: H.linear_bwd_data(dy, w, out_dtype=bf))
    sweep(f"{name} wgrad + bias grad", fl, 2.0 * (M * N + M * K) + 4.0 * N * K, lambda: H.linear_bwd_weight(dy, x, dw, bias_grad=db))
    H._PENDING_REDUCTIONS.clear()
T, B, C = 648, M // 648, 512
x = torch.randn(B * T, C, device=dev).to(bf)
w = (torch.randn(5, C, C, device=dev) * (5 * C) ** -0.5).to(bf)
b = torch.randn(C, device=dev)
dy = torch.randn(B * T, C, device=dev).to(bf)
dw, db = torch.empty(5, C, C, device=dev), torch.empty(C, device=dev)
fl = 2.0 * B * T * C * C * 5
sweep("postnet conv k5 fwd", fl, 2.0 * (B * T * C + 5 * C * C) + 4.0 * B * T * C, lambda: H.linear_fwd(x, w, b, taps=5, T=T))
sweep("postnet conv k5 dgrad", fl, 2.0 * (B * T * C + 5 * C * C) + 4.0 * B * T * C, lambda: H.linear_bwd_data(dy, w, taps=5, T=T))
sweep("postnet conv k5 wgrad", fl, 4.0 * B * T * C + 20.0 * C * C, lambda: H.linear_bwd_weight(dy, x, dw, taps=5, T=T, bias_grad=db))
H._PENDING_REDUCTIONS.clear()
# the passes the storage mode still pays for around attention
o = torch.randn(M, 256, device=dev)
t = timeit(lambda: H.cast_bf16(o))
print(f"  cast o {M}x256 fp32 -> bf16: {t * 1e6:.1f} us ({6.0 * M * 256 / t / 1e9:.0f} GB/s)")
dq = torch.randn(M, 768, device=dev)
t = timeit(lambda: H.cast_bf16(dq))
print(f"  cast dqkv {M}x768 fp32 -> bf16: {t * 1e6:.1f} us ({6.0 * M * 768 / t / 1e9:.0f} GB/s)")
