"""Ceiling check: vendor-library fp32 GEMM (torch.mm -> rocBLAS/hipBLASLt) vs. this build's cores on the
benchmark step's shapes (GPU only).  Not used by the product; tells how far the hand-written cores are from
what the chip's fp32 MFMA path gives in practice."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda"
shapes = []
for M in (20736, 4096):
    shapes += [("ffn1 fwd", "nt", M, 1024, 256), ("ffn2 fwd", "nt", M, 256, 1024), ("qkv fwd", "nt", M, 768, 256),
               ("proj fwd", "nt", M, 256, 256), ("ffn2 dx", "nn", M, 1024, 256), ("ffn1 dx", "nn", M, 256, 1024),
               ("ffn1 dw", "tn", M, 1024, 256), ("ffn2 dw", "tn", M, 256, 1024), ("proj dw", "tn", M, 256, 256)]
shapes += [("square", "nt", 8192, 8192, 8192), ("square", "nn", 8192, 8192, 8192), ("square", "tn", 8192, 8192, 8192)]
for name, kind, m, n, k in shapes:
    if kind == "nt":
        x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); out = torch.empty(m, n, device=dev)
        fn = lambda: H.linear_fwd(x, w, out=out)
        lib = lambda: torch.mm(x, w.t(), out=out)
    elif kind == "nn":
        dy = torch.randn(m, k, device=dev); w = torch.randn(k, n, device=dev); out = torch.empty(m, n, device=dev)
        fn = lambda: H.linear_bwd_data(dy, w, out=out)
        lib = lambda: torch.mm(dy, w, out=out)
    else:
        dy = torch.randn(m, n, device=dev); x = torch.randn(m, k, device=dev); out = torch.empty(n, k, device=dev)
        fn = lambda: H.linear_bwd_weight(dy, x, out)
        lib = lambda: torch.mm(dy.t(), x, out=out)
    fl = 2.0 * m * n * k
    H._TILE_CACHE.clear()
    fn()
    tile = list(H._TILE_CACHE.values())[-1] if H._TILE_CACHE else 0
    t_mine, t_lib = timeit(fn, 20), timeit(lib, 20)
    print(f"{name:10s} {kind} M={m:6d} N={n:5d} K={k:5d}  mine(tile {tile}): {fl / t_mine / 1e12:6.1f} TF/s {t_mine * 1e6:7.1f} us   "
          f"lib: {fl / t_lib / 1e12:6.1f} TF/s {t_lib * 1e6:7.1f} us", flush=True)
