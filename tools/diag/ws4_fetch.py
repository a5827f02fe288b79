"""A few launches of the K = 1024 data-gradient form on one tile, for a counter pass
(rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/diag/ws4_fetch.py TILE [N]): does the A operand leave the memory
side once, or once per column slice?"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402

tile = int(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
M = 43008
H.set_precision("bf16-mixed")
g = torch.Generator().manual_seed(0)
x = torch.randn(M, 1024, generator=g).bfloat16().cuda()
w = (torch.randn(N, 1024, generator=g) / 32).bfloat16().cuda()
H.GEMM_TILES_B = (tile,)
H.GEMM_TUNE = True
big = torch.empty(512 * 1024 * 1024 // 4, device="cuda")
for _ in range(3):
    big.zero_()
    H.linear_bwd_data(x, None, out_dtype=torch.bfloat16, wt=w)
torch.cuda.synchronize()
