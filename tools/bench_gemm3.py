"""Per-shape, per-tile timing of the GEMM cores on the benchmark step's shapes (GPU only)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

M, T, dev = 20736, 648, "cuda"
shapes = [("ffn1 fwd", "nt", M, 1024, 256, 1), ("ffn2 fwd", "nt", M, 256, 1024, 1), ("qkv fwd", "nt", M, 768, 256, 1),
          ("proj fwd", "nt", M, 256, 256, 1), ("post1 fwd", "nt", M, 512, 512, 5), ("post0 fwd", "nt", M, 512, 80, 5),
          ("ffn2 dx", "nn", M, 1024, 256, 1), ("ffn1 dx", "nn", M, 256, 1024, 1), ("post1 dx", "nn", M, 512, 512, 5),
          ("ffn1 dw", "tn", M, 1024, 256, 1), ("ffn2 dw", "tn", M, 256, 1024, 1), ("proj dw", "tn", M, 256, 256, 1),
          ("post1 dw", "tn", M, 512, 512, 5)]
H.GEMM_TUNE = True
for name, kind, m, n, k, taps in shapes:
    res = []
    for tile in (1, 2, 3, 4, 5, 6):
        H.GEMM_TILES = (tile,)
        H._TILE_CACHE.clear()
        if kind == "nt":
            x = torch.randn(m, k, device=dev); w = torch.randn(*((taps, n, k) if taps > 1 else (n, k)), device=dev)
            out = torch.empty(m, n, device=dev); fn = lambda: H.linear_fwd(x, w, taps=taps, T=T, out=out)
        elif kind == "nn":
            dy = torch.randn(m, k, device=dev); w = torch.randn(*((taps, k, n) if taps > 1 else (k, n)), device=dev)
            out = torch.empty(m, n, device=dev); fn = lambda: H.linear_bwd_data(dy, w, taps=taps, T=T, out=out)
        else:
            dy = torch.randn(m, n, device=dev); x = torch.randn(m, k, device=dev)
            out = torch.empty(*((taps, n, k) if taps > 1 else (n, k)), device=dev)
            fn = lambda: H.linear_bwd_weight(dy, x, out, taps=taps, T=T)
        t = timeit(fn, 10)
        res.append(2.0 * m * n * k * taps / t / 1e12)
    print(f"{name:10s} {kind} N={n:5d} K={k*taps:5d} " + " ".join(f"t{i+1}:{r:6.1f}" for i, r in enumerate(res)), flush=True)
