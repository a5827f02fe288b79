import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from fastspeech2_lightning_amd import hip as H
H.GEMM_TUNE = False
dev = "cuda"
torch.manual_seed(0)
for (m, n, k) in ((256, 128, 64), (256, 128, 128), (512, 256, 64), (4100, 256, 1024)):
    dy = torch.randn(m, n, device=dev); w = torch.randn(n, k, device=dev)
    ref = dy.double() @ w.double()
    for tile in (9, 12, 8, 11, 7, 10):
        H._tune_tile = lambda a, t=tile: t
        out = H.linear_bwd_data(dy, w)
        err = (out.double() - ref).abs()
        bad = (err > 1e-3 * ref.abs().max()).nonzero()
        msg = f"M={m} N={n} K={k} tile {tile}: max err {err.max().item():.3e} bad {len(bad)}"
        if len(bad):
            rows = bad[:, 0].unique(); cols = bad[:, 1].unique()
            msg += f" rows[{rows.min().item()}..{rows.max().item()}] n={len(rows)} cols[{cols.min().item()}..{cols.max().item()}] n={len(cols)}"
        print(msg, flush=True)
