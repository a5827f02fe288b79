"""K sweep at fixed M x N: slope = main-loop cost per K-tile, intercept = per-tile fixed cost (GPU only)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
for m, n in ((20736, 1024), (20736, 256), (20736, 512)):
    for tile in (7, 8, 9, 5, 4):
        row = []
        for k in (64, 128, 256, 512, 1024, 2048):
            x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); out = torch.empty(m, n, device=dev)
            H.GEMM_TILES = (tile,)
            H._TILE_CACHE.clear()
            t = timeit(lambda: H.linear_fwd(x, w, out=out), 20)
            row.append(f"K={k}: {t * 1e6:7.1f}us {2.0 * m * n * k / t / 1e12:6.1f}TF")
        print(f"M={m} N={n} tile {tile}: " + " | ".join(row), flush=True)
