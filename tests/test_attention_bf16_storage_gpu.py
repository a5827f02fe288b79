"""GPU parity of the bf16-storage attention kernels (attention_bf16.hip: qkv / o / dout / dqkv ARE bf16 tensors).

* against a plain PyTorch fp32 reference evaluated on the same bf16 VALUES (so the only differences are the bf16
  rounding of the probabilities / score gradients that feed the second product, and of the results): relative L2
  error 1e-2, largest element 3e-2 of the tensor's scale, lse 2e-3;
* against the fp32 kernels of attention2.hip with dropout on: the keep mask is the same function of (seed, element
  index), so results stay within the operand precision -- a different mask would be an O(1) difference;
* padded keys get exactly zero gradients, rows of a row block that lie beyond T are never written.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from fastspeech2_lightning_amd import hip
    hip.lib()
    return hip


def ref_attention(qkv, lens, B, T, Hh):
    D = qkv.shape[-1] // 3
    hd = D // Hh
    q, k, v = qkv.view(B, T, 3, Hh, hd).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    pad = torch.arange(T, device=qkv.device)[None, :] >= lens[:, None]
    s = s.masked_fill(pad[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B, T, D), torch.logsumexp(s, dim=-1)


def rel_l2(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


CASES = [
    (1, 64, [64]), (2, 33, [33, 7]), (3, 130, [130, 64, 1]), (2, 200, [200, 129]), (2, 648, [648, 500]),
    (1, 128, [128]), (2, 129, [129, 128]), (2, 95, [31, 95]), (2, 5, [5, 2]), (1, 1, [1]), (4, 31, [31, 1, 16, 17]),
    # benchmark size (more row-block workgroups than slots), ragged lengths, one very short sequence
    (32, 648, [648, 430, 40] + [430 + 7 * i for i in range(29)]),
]


@pytest.mark.parametrize("B,T,lens", CASES)
def test_attention_bf16_storage_fwd_bwd(H, B, T, lens):
    Hh, hd = 2, 128
    D = Hh * hd
    g = torch.Generator().manual_seed(B * 1000 + T)
    qkv = torch.randn(B, T, 3 * D, generator=g).bfloat16().cuda()
    dout = torch.randn(B, T, D, generator=g).bfloat16().cuda()
    lens_t = torch.tensor(lens, dtype=torch.int32).cuda()
    qr = qkv.float().requires_grad_(True)
    ref, ref_lse = ref_attention(qr, lens_t, B, T, Hh)
    ref.backward(dout.float())
    o, lse = H.attention_fwd_b(qkv, lens_t, B, T, Hh)
    assert o.dtype == torch.bfloat16 and lse.dtype == torch.float32
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    assert rel_l2(o.float(), ref) < 1e-2
    assert (o.float() - ref).abs().max().item() < 3e-2 * ref.abs().max().item()
    assert (lse - ref_lse).abs().max().item() < 2e-3 * max(1.0, ref_lse.abs().max().item())
    dqkv = H.attention_bwd_b(qkv, lens_t, o, dout, lse, B, T, Hh)
    assert torch.isfinite(dqkv.float()).all()
    gref = qr.grad
    # per part, against the scale of the whole gradient (a single-key softmax has exactly zero dQ and dK; what the
    # kernels leave there is the bf16 rounding of O in delta)
    parts = (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D)))
    gmax, gnorm = gref.abs().max().item(), gref.norm().item()
    errs = {}
    for n, sl in parts:
        d = dqkv.float()[..., sl] - gref[..., sl]
        ref_norm = max(gref[..., sl].norm().item(), 1e-2 * gnorm)
        ref_max = max(gref[..., sl].abs().max().item(), 1e-1 * gmax)
        errs[n] = (d.norm().item() / ref_norm, d.abs().max().item() / ref_max)
    assert all(e[0] < 1.5e-2 and e[1] < 3e-2 for e in errs.values()), errs
    pad = torch.arange(T, device="cuda")[None, :] >= lens_t[:, None]
    assert (dqkv.float()[..., D:][pad] == 0).all(), "padded keys must get zero gradients"


@pytest.mark.parametrize("T", [130, 648])
def test_attention_bf16_storage_shares_the_dropout_mask(H, T):
    B, Hh, hd = 3, 2, 128
    D = Hh * hd
    g = torch.Generator().manual_seed(T)
    qkv = torch.randn(B, T, 3 * D, generator=g).bfloat16().cuda()
    dout = torch.randn(B, T, D, generator=g).bfloat16().cuda()
    lens = torch.tensor([T, T // 2, 3], dtype=torch.int32).cuda()
    drop = H.Drop(0.2, 4242)
    o0, lse0 = H.attention_fwd(qkv.float(), lens, B, T, Hh, drop)
    g0 = H.attention_bwd(qkv.float(), lens, o0, dout.float(), lse0, B, T, Hh, drop)
    o1, lse1 = H.attention_fwd_b(qkv, lens, B, T, Hh, drop)
    g1 = H.attention_bwd_b(qkv, lens, o1, dout, lse1, B, T, Hh, drop)
    valid = (torch.arange(T, device="cuda")[None, :] < lens[:, None])[..., None]
    assert (lse1 - lse0).abs().max().item() < 2e-3 * max(1.0, lse0.abs().max().item())
    eo = ((o1.float() - o0) * valid).abs().max().item() / o0.abs().max().item()
    eg = ((g1.float() - g0).view(B, T, -1) * valid).abs().max().item() / g0.abs().max().item()
    assert eo < 3e-2 and eg < 3e-2, (eo, eg)
    assert rel_l2(o1.float() * valid, o0 * valid) < 1e-2
    assert rel_l2(g1.float().view(B, T, -1) * valid, g0.view(B, T, -1) * valid) < 1.5e-2
    # dropout really is on: the undropped result differs at O(1)
    o2, _ = H.attention_fwd_b(qkv, lens, B, T, Hh)
    assert rel_l2(o2.float() * valid, o0 * valid) > 0.1


def test_attention_bf16_storage_rejects_other_head_dims(H):
    assert H.attention_b_supported(128) and not H.attention_b_supported(64)
    qkv = torch.zeros(1, 8, 3 * 64, dtype=torch.bfloat16, device="cuda")
    lens = torch.tensor([8], dtype=torch.int32).cuda()
    with pytest.raises(Exception):
        H.attention_fwd_b(qkv, lens, 1, 8, 2)


@pytest.mark.parametrize("B,T,lens", CASES + [(3, 1291, [1291, 700, 64])])
def test_bf16_storage_backward_with_spilled_ds_equals_the_recomputing_one(H, B, T, lens, monkeypatch):
    """``fs2hip_attention_bwd_b_spill`` (default): the dK/dV kernel writes dS -- masked, rounded to bf16: the operand the dQ
    product consumes -- and dQ = scale * dS . K is a product of its own.  dK and dV are the recomputing kernels' bit for bit
    (same kernel, sixteen stores more per block); dQ within bf16 rounding of theirs -- dropout on and off, ragged lengths,
    utterances that end inside a key block, T not a multiple of the tile, T = 1."""
    Hh, hd = 2, 128
    D = Hh * hd
    g = torch.Generator().manual_seed(B * 313 + T)
    qkv = torch.randn(B, T, 3 * D, generator=g).bfloat16().cuda()
    dout = torch.randn(B, T, D, generator=g).bfloat16().cuda()
    lens_t = torch.tensor(lens, dtype=torch.int32).cuda()
    for drop in (H.NO_DROP, H.Drop(0.1, 99)):
        o, lse = H.attention_fwd_b(qkv, lens_t, B, T, Hh, drop)
        monkeypatch.setattr(H, "ATTN_SPILL_B", False)
        want = H.attention_bwd_b(qkv, lens_t, o, dout, lse, B, T, Hh, drop).float()
        monkeypatch.setattr(H, "ATTN_SPILL_B", True)
        H._SCRATCH.clear()
        got = H.attention_bwd_b(qkv, lens_t, o, dout, lse, B, T, Hh, drop).float()
        assert torch.isfinite(got).all()
        assert torch.equal(got[..., D:], want[..., D:]), "dK / dV changed"
        scale = float(want.abs().max())
        err = float((got[..., :D] - want[..., :D]).abs().max())
        assert err <= 2.0 ** -7 * scale, (drop.p, err, scale)
        assert rel_l2(got[..., :D], want[..., :D]) < 4e-3 or float(want[..., :D].abs().max()) < 1e-3 * scale
