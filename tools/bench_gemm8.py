"""Epilogue cost on the FFN shapes: plain store vs the fused epilogues the step uses (GPU only)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
M = 20736
step = torch.zeros(4, dtype=torch.int64, device=dev)
drop = H.Drop(0.2, 12345, step)
x = torch.randn(M, 256, device=dev); w1 = torch.randn(1024, 256, device=dev); b1 = torch.randn(1024, device=dev)
a = torch.randn(M, 1024, device=dev); w2 = torch.randn(256, 1024, device=dev); b2 = torch.randn(256, device=dev)
u = torch.empty(M, 1024, device=dev); o1 = torch.empty(M, 1024, device=dev); o2 = torch.empty(M, 256, device=dev)
dz = torch.randn(M, 256, device=dev)
cases = [
    ("ffn1 fwd plain", 2.0 * M * 1024 * 256, lambda: H.linear_fwd(x, w1, b1, out=o1)),
    ("ffn1 fwd act+pre", 2.0 * M * 1024 * 256, lambda: H.linear_fwd(x, w1, b1, epi=H.EPI_ACT, act="silu", out_pre=u, out=o1)),
    ("ffn1 fwd act+pre+drop", 2.0 * M * 1024 * 256, lambda: H.linear_fwd(x, w1, b1, epi=H.EPI_ACT, act="silu", out_pre=u, drop=drop, out=o1)),
    ("ffn2 fwd plain", 2.0 * M * 1024 * 256, lambda: H.linear_fwd(a, w2, b2, out=o2)),
    ("ffn2 fwd resid+drop", 2.0 * M * 1024 * 256, lambda: H.linear_fwd(a, w2, b2, epi=H.EPI_RESID, resid=x, res_scale=0.5, drop=drop, out=o2)),
    ("ffn2 dx plain", 2.0 * M * 1024 * 256, lambda: H.linear_bwd_data(dz, w2, out=o1)),
    ("ffn2 dx dact+drop", 2.0 * M * 1024 * 256, lambda: H.linear_bwd_data(dz, w2, epi=H.EPI_DACT, act="silu", aux=u, drop=drop, out=o1)),
]
tiles = tuple(int(t) for t in sys.argv[1].split(",")) if len(sys.argv) > 1 else (7, 8, 9, 10, 11, 12)
for name, fl, fn in cases:
    res = []
    for tile in tiles:
        H.GEMM_TILES = (tile,)
        H._TILE_CACHE.clear()
        t = timeit(fn, 20)
        res.append(f"t{tile}:{fl / t / 1e12:6.1f}")
    print(f"{name:22s} " + " ".join(res), flush=True)
