"""Split-K sweep of the bf16-storage weight-gradient GEMMs (rows = 41 472).  GPU only."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402

H.set_precision("bf16-mixed")
dev, bf = "cuda", torch.bfloat16
M = 41472


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
    return e0.elapsed_time(e1) / n * 1e3


for name, N, K in (("ffn", 1024, 256), ("qkv", 768, 256), ("proj", 256, 256), ("pw1", 512, 256)):
    x = torch.randn(M, K, device=dev).to(bf)
    dy = torch.randn(M, N, device=dev).to(bf)
    dw, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
    row = []
    for smax in (4, 8, 16, 32, 64):
        H.SPLITK_MAX = smax
        best = None
        for tile in H.GEMM_TILES_B:
            saved = H.GEMM_TILES_B
            H.GEMM_TILES_B = (tile,)
            H._TILE_CACHE.clear()
            try:
                t = timeit(lambda: (H.linear_bwd_weight(dy, x, dw, bias_grad=db), H._PENDING_REDUCTIONS.clear()))
                if best is None or t < best[0]:
                    best = (t, tile)
            except Exception:
                pass
            H.GEMM_TILES_B = saved
        row.append(f"S<={smax}: {best[0]:6.1f} us (tile {best[1]})")
    print(f"{name:5s} {N:5d}x{K:4d}: " + " | ".join(row))
