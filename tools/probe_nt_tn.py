import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H
M, n, k, dev = 20736, 1024, 256, "cuda"
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H.GEMM_TILES = (tile,)
x = torch.randn(M, k, device=dev); w = torch.randn(n, k, device=dev); out = torch.empty(M, n, device=dev)
dy = torch.randn(M, n, device=dev); dw = torch.empty(n, k, device=dev)
for _ in range(6):
    H.linear_fwd(x, w, out=out)          # NT
for _ in range(6):
    H.linear_bwd_weight(dy, x, dw)       # TN (split-K)
torch.cuda.synchronize()
