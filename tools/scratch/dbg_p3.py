import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from fastspeech2_lightning_amd import hip as H
H.GEMM_TUNE = False
dev = "cuda"
m, n, k = 64, 64, 64
H._tune_tile = lambda a: 10
# dy[r, c] = r (row id), w = identity-ish: out[r, j] = sum_c dy[r,c] w[c,j]; w[c, j] = 1 if c == j
dy = torch.arange(m, device=dev, dtype=torch.float32)[:, None].expand(m, n).contiguous()
w = torch.eye(n, k, device=dev)
out = H.linear_bwd_data(dy, w)
print("row-id test (expect out[r, j] = r):")
print(out[24:34, :6])
dy = torch.arange(n, device=dev, dtype=torch.float32)[None, :].expand(m, n).contiguous()
out = H.linear_bwd_data(dy, w)
print("col-id test (expect out[r, j] = j):")
print(out[24:34, :8])
print(out[27, :].tolist())
