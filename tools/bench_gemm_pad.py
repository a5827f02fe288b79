"""Hypothesis test: is the NT GEMM limited by the power-of-two row stride of A (L2/HBM channel conflicts)?
Times y = x @ w^T with x stored at leading dimension K (dense) vs K + pad."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
M = 20736
for n, k in [(1024, 256), (256, 1024), (256, 256), (768, 256)]:
    for pad in (0, 32, 64, 96):
        xs = torch.randn(M, k + pad, device=dev)
        w = torch.randn(n, k, device=dev)
        out = torch.empty(M, n, device=dev)
        res = []
        for tile in (3, 4, 5, 7):
            def fn():
                H._gemm(A=xs.data_ptr(), B=w.data_ptr(), C=out.data_ptr(), Mc=M, Nc=n, R=k, lda=k + pad, ldb=k, ldc=n,
                        a_kcontig=1, b_kcontig=1, taps=1, T=0, tap_mul=1, tap_add=0, shift_operand=0, b_tap_stride=n * k,
                        epi=0, act=0)
            H.GEMM_TILES = (tile,)
            H._TILE_CACHE.clear()
            t = timeit(fn, 10)
            res.append(2.0 * M * n * k / t / 1e12)
        print(f"N={n:5d} K={k:5d} lda=K+{pad:3d}: " + " ".join(f"t{t}:{r:6.1f}" for t, r in zip((3, 4, 5, 7), res)), flush=True)
