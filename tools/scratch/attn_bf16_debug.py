import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
from fastspeech2_lightning_amd import hip as H
H.lib()
for (B, T, Hh, hd, lens) in [(1, 16, 1, 16, [16]), (1, 64, 1, 16, [64]), (2, 37, 2, 16, [37, 5]), (1, 64, 1, 128, [64]), (2, 648, 2, 128, [648, 500])]:
    g = torch.Generator().manual_seed(1)
    D = Hh * hd
    qkv = torch.randn(B, T, 3 * D, generator=g).cuda()
    lens_t = torch.tensor(lens, dtype=torch.int32).cuda()
    H.set_precision("32-true")
    o32, l32 = H.attention_fwd(qkv, lens_t, B, T, Hh)
    H.set_precision("bf16-mixed")
    o, lse = H.attention_fwd(qkv, lens_t, B, T, Hh)
    torch.cuda.synchronize()
    print((B, T, Hh, hd, lens), "nan o", int(o.isnan().sum()), "of", o.numel(), "nan lse", int(lse.isnan().sum()), "of", lse.numel(),
          "max|o-o32|", float((o - o32).nan_to_num(0).abs().max()), "max|lse-l32|", float((lse - l32).nan_to_num(0).abs().max()))
    if lse.isnan().any():
        idx = lse.isnan().nonzero()[:8].tolist()
        print("   first nan lse idx", idx)
        print("   lse row0:", lse.flatten()[:8].tolist(), " fp32:", l32.flatten()[:8].tolist())
