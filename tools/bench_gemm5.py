"""Per-tile table (all nine workgroup tiles) on the main step shapes + a large square (GPU only)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
shapes = [("ffn1 fwd", "nt", 20736, 1024, 256), ("ffn2 fwd", "nt", 20736, 256, 1024), ("qkv fwd", "nt", 20736, 768, 256),
          ("proj fwd", "nt", 20736, 256, 256), ("ffn2 dx", "nn", 20736, 1024, 256), ("ffn1 dx", "nn", 20736, 256, 1024),
          ("ffn1 dw", "tn", 20736, 1024, 256), ("ffn1 fwd", "nt", 4096, 1024, 256), ("ffn2 fwd", "nt", 4096, 256, 1024),
          ("proj fwd", "nt", 4096, 256, 256), ("square", "nt", 4096, 4096, 4096), ("square", "nn", 4096, 4096, 4096)]
tiles = tuple(int(t) for t in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 2, 3, 4, 5, 6, 7, 8, 9)
for name, kind, m, n, k in shapes:
    if kind == "nt":
        x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); out = torch.empty(m, n, device=dev)
        fn = lambda: H.linear_fwd(x, w, out=out)
    elif kind == "nn":
        dy = torch.randn(m, k, device=dev); w = torch.randn(k, n, device=dev); out = torch.empty(m, n, device=dev)
        fn = lambda: H.linear_bwd_data(dy, w, out=out)
    else:
        dy = torch.randn(m, n, device=dev); x = torch.randn(m, k, device=dev); out = torch.empty(n, k, device=dev)
        fn = lambda: H.linear_bwd_weight(dy, x, out)
    res = []
    for tile in tiles:
        H.GEMM_TILES = (tile,)
        H._TILE_CACHE.clear()
        try:
            t = timeit(fn, 20)
            res.append(2.0 * m * n * k / t / 1e12)
        except Exception:
            res.append(float("nan"))
    print(f"{name:9s} {kind} M={m:6d} N={n:5d} K={k:5d} " + " ".join(f"t{t}:{r:6.1f}" for t, r in zip(tiles, res)), flush=True)
