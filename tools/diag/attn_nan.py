import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from fastspeech2_lightning_amd import hip as H
torch.manual_seed(0)
for (B, T, Hh, hd, lens) in [(1, 1, 2, 128, [1]), (2, 5, 2, 64, [5, 2]), (3, 130, 2, 128, [130, 64, 1])]:
    D = Hh * hd
    g = torch.Generator().manual_seed(B * 91 + T)
    qkv = torch.randn(B, T, 3 * D, generator=g).cuda()
    dout = torch.randn(B, T, D, generator=g).cuda()
    lens_t = torch.tensor(lens, dtype=torch.int32).cuda()
    for trial in range(6):
        junk = torch.full((1 << 22,), float("nan"), device="cuda")  # poison freshly freed memory
        del junk
        o, lse, sc = H.attention_fwd(qkv, lens_t, B, T, Hh, save_scores=True)
        got = H.attention_bwd(qkv, lens_t, o, dout, lse, B, T, Hh, scores=sc)
        bad = torch.isnan(got)
        print((B, T), "trial", trial, "nan count", int(bad.sum()), "of", got.numel(),
              "| q:", int(bad[..., :D].sum()), "k:", int(bad[..., D:2 * D].sum()), "v:", int(bad[..., 2 * D:].sum()),
              "| sc nan", int(torch.isnan(sc).sum()), "sc shape", tuple(sc.shape), flush=True)
