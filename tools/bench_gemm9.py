"""Where the PostNet convolution loses against the big-square GEMM rate: the same reduction depth (2560) as a plain
GEMM at the conv's size (20736 x 512), at a tile count that fills whole rounds (20480 x 512 / 16384 x 1024), and the
5-tap form (GPU only).  Per-tile TFLOP/s."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
tiles = tuple(int(t) for t in sys.argv[1].split(",")) if len(sys.argv) > 1 else (4, 9, 12, 13, 8, 11)
for name, M, N, K, taps, T in [("plain 20736x512 K2560", 20736, 512, 2560, 1, 0), ("plain 16384x512 K2560", 16384, 512, 2560, 1, 0),
                               ("plain 16384x1024 K2560", 16384, 1024, 2560, 1, 0), ("plain 8192x8192 K2560", 8192, 8192, 2560, 1, 0),
                               ("conv5 20736x512 (5x512)", 20736, 512, 512, 5, 648), ("conv5 16384x512 (5x512)", 16384, 512, 512, 5, 512)]:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) if taps == 1 else torch.randn(taps, N, K, device=dev)
    out = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K * taps
    res = []
    for tile in tiles:
        H.GEMM_TILES = (tile,)
        H._TILE_CACHE.clear()
        t = timeit(lambda: H.linear_fwd(x, w, taps=taps, T=T, out=out), 10)
        used = list(H._TILE_CACHE.values())[-1]
        res.append(f"t{tile}{'' if used == tile else '(->' + str(used) + ')'}:{fl / t / 1e12:6.1f}")
    print(f"{name:26s} " + " ".join(res), flush=True)
