"""Isolated timing of the exact-fp32 weights-stationary streaming GEMM (tile 32, csrc/gemm_ws32.hip) against the best tiled
kernels on the K = 256 shapes of the benchmark step (forward orientation; the data gradients reach it through W^T mirrors).
usage (GPU): python tools/diag/ws32_time.py [M]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from fastspeech2_lightning_amd import hip as H
M = int(sys.argv[1]) if len(sys.argv) > 1 else 20736
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
H.GEMM_TUNE = True
for N in (1024, 768, 512, 256):
    x = torch.randn(M, 256, device="cuda"); w = torch.randn(N, 256, device="cuda") / 16; b = torch.randn(N, device="cuda")
    u = torch.empty(M, N, device="cuda"); r = torch.randn(M, N, device="cuda"); drop = H.Drop(0.2, 5)
    for tile in (7, 8, 5, 32):
        H.GEMM_TILES = (tile,); H._TILE_CACHE.clear(); H._TILE_REFUSED.clear()
        try:
            a = t(lambda: H.linear_fwd(x, w, b))
            c = t(lambda: H.linear_fwd(x, w, b, epi=H.EPI_ACT, act="silu", out_pre=u, drop=drop))
            d = t(lambda: H.linear_fwd(x, w, b, epi=H.EPI_RESID, resid=r, drop=drop))
            e = t(lambda: H.linear_fwd(x, w, None, epi=H.EPI_DACT, act="silu", aux=u, drop=drop)) if False else float("nan")
        except Exception as ex:
            print(f"N={N} tile {tile}: refused ({type(ex).__name__})", flush=True)
            continue
        print(f"M={M} N={N} tile {tile:2d}: store {a:6.1f} us ({2*M*N*256/a/1e6:5.1f} TF)   silu+drop+pre {c:6.1f} us ({2*M*N*256/c/1e6:5.1f} TF)   "
              f"resid+drop {d:6.1f} us ({2*M*N*256/d/1e6:5.1f} TF)", flush=True)
