import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from fastspeech2_lightning_amd import hip as H
if len(sys.argv) > 1:
    H._LIB_PATH = Path(sys.argv[1]).resolve()
H.GEMM_TUNE = False
dev = "cuda"
torch.manual_seed(0)
for (m, n, k) in ((256, 128, 64), (4100, 256, 1024)):
    dy = torch.randn(m, n, device=dev); w = torch.randn(n, k, device=dev)
    ref = dy.double() @ w.double()
    for tile in (12, 11, 10):
        H._tune_tile = lambda a, t=tile: t
        out = H.linear_bwd_data(dy, w)
        err = (out.double() - ref).abs()
        bad = (err > 1e-3 * ref.abs().max()).nonzero()
        print(f"{sys.argv[1:]} M={m} N={n} K={k} tile {tile}: max err {err.max().item():.3e} bad {len(bad)}", flush=True)
