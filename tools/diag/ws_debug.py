import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent / "tests"))
from fastspeech2_lightning_amd import hip as H
import test_gemm_ws_gpu as T
M, N, tile = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
H.GEMM_TUNE = True
ref = T.reference(M, N)
with T.only_tile(H, tile):
    got = T.forms(H, M, N, H.NO_DROP)
    print("tiles used", {k[:3] + (k[8],): v for k, v in H._TILE_CACHE.items()})
for name, want in ref.items():
    g = got[name].double().cpu()
    bad = ~((g - want).abs() < 2.0 ** -6 * max(1.0, float(want.abs().max())))
    if bad.any():
        idx = bad.nonzero()
        rows = sorted(set(idx[:, 0].tolist()))
        cols = sorted(set(idx[:, 1].tolist()))
        print(name, "bad", int(bad.sum()), "rows", rows[:10], "...", rows[-3:], "cols", cols[:12], "...", cols[-4:], "nan", int(torch.isnan(g).sum()))
    else:
        print(name, "ok")
