"""PostNet-shaped 5-tap convolutions (M = 20736, 512 -> 512 channels): per-tile TFLOP/s for forward, backward-data
and weight-gradient GEMMs (GPU only)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
B, T, C, taps = 32, 648, 512, 5
M = B * T
tiles = tuple(int(t) for t in sys.argv[1].split(",")) if len(sys.argv) > 1 else (7, 8, 9, 10, 11, 12, 13, 14)
x = torch.randn(M, C, device=dev); w = torch.randn(taps, C, C, device=dev); dy = torch.randn(M, C, device=dev)
out = torch.empty(M, C, device=dev); dw = torch.empty(taps, C, C, device=dev)
cases = [("conv fwd", lambda: H.linear_fwd(x, w, taps=taps, T=T, out=out)),
         ("conv dx", lambda: H.linear_bwd_data(dy, w, taps=taps, T=T, out=out)),
         ("conv dw", lambda: H.linear_bwd_weight(dy, x, dw, taps=taps, T=T))]
fl = 2.0 * M * C * C * taps
for name, fn in cases:
    res = []
    for tile in tiles:
        H.GEMM_TILES = (tile,)
        H._TILE_CACHE.clear()
        t = timeit(fn, 10)
        used = list(H._TILE_CACHE.values())[-1]
        res.append(f"t{tile}{'' if used == tile else '(->' + str(used) + ')'}:{fl / t / 1e12:6.1f}")
    print(f"{name:9s} " + " ".join(res), flush=True)
