"""GPU parity of the flash-style attention kernels against a plain PyTorch fp32 reference
(softmax(QK^T/sqrt(d) + key-padding mask) V), forward and backward.  Tolerance 2e-5 of scale."""
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
    q, k, v = qkv.view(B, T, 3, Hh, hd).permute(2, 0, 3, 1, 4)  # each (B, H, T, hd)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    pad = torch.arange(T)[None, :] >= lens[:, None]
    s = s.masked_fill(pad[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B, T, D), torch.logsumexp(s, dim=-1)


@pytest.mark.parametrize("B,T,Hh,hd,lens", [
    (2, 37, 2, 16, [37, 5]), (3, 130, 2, 128, [130, 64, 1]), (2, 64, 4, 32, [64, 63]), (1, 200, 2, 64, [200]),
    (2, 648, 2, 128, [648, 500]),
    # benchmark size (more row-block workgroups than slots), ragged lengths, one very short sequence
    (32, 648, 2, 128, [648, 430, 40] + [430 + 7 * i for i in range(29)]),
])
def test_attention_fwd_bwd(H, B, T, Hh, hd, lens):
    g = torch.Generator().manual_seed(B * 1000 + T)
    D = Hh * hd
    qkv = torch.randn(B, T, 3 * D, generator=g)
    dout = torch.randn(B, T, D, generator=g)
    lens_t = torch.tensor(lens, dtype=torch.int32)
    qr = qkv.clone().requires_grad_(True)
    ref, ref_lse = ref_attention(qr, lens_t, B, T, Hh)
    ref.backward(dout)
    o, lse = H.attention_fwd(qkv.cuda(), lens_t.cuda(), B, T, Hh)
    assert (o.cpu() - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())
    assert (lse.cpu() - ref_lse).abs().max() < 2e-5 * max(1.0, ref_lse.abs().max().item())
    dqkv = H.attention_bwd(qkv.cuda(), lens_t.cuda(), o, dout.cuda(), lse, B, T, Hh)
    scale = qr.grad.abs().max().item()
    err = (dqkv.cpu() - qr.grad).abs().max().item()
    assert err < 3e-5 * scale, f"dqkv err {err} scale {scale}"


def test_attention_dropout_consistency(H):
    """With dropout the forward/backward share one regenerated mask: check d(sum(o*w))/dqkv by
    finite differences of the kernel itself (mask fixed by the seed)."""
    B, T, Hh, hd = 1, 48, 2, 16
    D = Hh * hd
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B, T, 3 * D, generator=g).cuda()
    w = torch.randn(B, T, D, generator=g).cuda()
    lens = torch.tensor([40], dtype=torch.int32).cuda()
    o, lse = H.attention_fwd(qkv, lens, B, T, Hh, H.Drop(0.3, 99))
    o2, _ = H.attention_fwd(qkv, lens, B, T, Hh)
    assert (o - o2).abs().max() > 1e-3  # dropout did something
    dqkv = H.attention_bwd(qkv, lens, o, w, lse, B, T, Hh, H.Drop(0.3, 99))
    eps = 1e-2
    for idx in [(0, 3, 5), (0, 10, D + 7), (0, 39, 2 * D + 20), (0, 45, 4)]:
        qp, qm = qkv.clone(), qkv.clone()
        qp[idx] += eps
        qm[idx] -= eps
        fp = (H.attention_fwd(qp, lens, B, T, Hh, H.Drop(0.3, 99))[0] * w).sum().item()
        fm = (H.attention_fwd(qm, lens, B, T, Hh, H.Drop(0.3, 99))[0] * w).sum().item()
        fd = (fp - fm) / (2 * eps)
        assert abs(fd - dqkv[idx].item()) < 2e-2 * max(1.0, abs(fd)), (idx, fd, dqkv[idx].item())


@pytest.mark.parametrize("B,T,Hh,hd,lens", [
    (2, 37, 2, 16, [37, 5]), (3, 130, 2, 128, [130, 64, 1]), (2, 64, 4, 32, [64, 63]), (2, 648, 2, 128, [648, 500]),
])
def test_attention_bf16_mixed(H, B, T, Hh, hd, lens):
    """precision "bf16-mixed": the operands of the five products are rounded to bf16 (relative step 2^-8), sums,
    softmax and outputs stay fp32.  Against the fp32 reference evaluated on bf16-rounded Q, K, V, dO (what remains is
    the rounding of P and dS inside the kernel): outputs within 1e-2 of the tensor scale, gradients within 2e-2; and
    the error must be above the fp32 kernels' (a silent fp32 fall-back fails the test)."""
    g = torch.Generator().manual_seed(B * 1000 + T)
    D = Hh * hd
    qkv = torch.randn(B, T, 3 * D, generator=g)
    dout = torch.randn(B, T, D, generator=g)
    lens_t = torch.tensor(lens, dtype=torch.int32)
    qr = qkv.bfloat16().float().requires_grad_(True)
    ref, ref_lse = ref_attention(qr, lens_t, B, T, Hh)
    ref.backward(dout.bfloat16().float())
    saved = H.get_precision()
    try:
        H.set_precision("bf16-mixed")
        o, lse = H.attention_fwd(qkv.cuda(), lens_t.cuda(), B, T, Hh)
        dqkv = H.attention_bwd(qkv.cuda(), lens_t.cuda(), o, dout.cuda(), lse, B, T, Hh)
    finally:
        H.set_precision(saved)
    eo = (o.cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    el = (lse.cpu() - ref_lse).abs().max().item() / max(1.0, ref_lse.abs().max().item())
    eg = (dqkv.cpu() - qr.grad).abs().max().item() / qr.grad.abs().max().item()
    assert eo < 1e-2 and el < 1e-2 and eg < 2e-2, (eo, el, eg)
    assert eo > 2e-5 or eg > 3e-5, "bf16 operands requested, fp32-exact result: the bf16 kernels did not run"


@pytest.mark.parametrize("B,T,Hh,hd,lens", [
    (2, 37, 2, 16, [37, 5]), (3, 130, 2, 128, [130, 64, 1]), (1, 200, 2, 64, [200]), (2, 648, 2, 128, [648, 500]),
    (32, 648, 2, 128, [648, 430, 40] + [430 + 7 * i for i in range(29)]),
])
def test_attention_split(H, B, T, Hh, hd, lens):
    """precision "32-split": Q.K^T, P.V and the dQ products on the bf16 pipe with operands cut into three exact bf16
    planes (head dims 64 / 128; dK/dV and the small head dims stay on the fp32 MFMAs).  Same bounds as the fp32 kernels."""
    g = torch.Generator().manual_seed(B * 1000 + T)
    D = Hh * hd
    qkv = torch.randn(B, T, 3 * D, generator=g)
    dout = torch.randn(B, T, D, generator=g)
    lens_t = torch.tensor(lens, dtype=torch.int32)
    qr = qkv.clone().requires_grad_(True)
    ref, ref_lse = ref_attention(qr, lens_t, B, T, Hh)
    ref.backward(dout)
    saved = H.get_precision()
    try:
        H.set_precision("32-split")
        o, lse = H.attention_fwd(qkv.cuda(), lens_t.cuda(), B, T, Hh)
        dqkv = H.attention_bwd(qkv.cuda(), lens_t.cuda(), o, dout.cuda(), lse, B, T, Hh)
    finally:
        H.set_precision(saved)
    assert (o.cpu() - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())
    assert (lse.cpu() - ref_lse).abs().max() < 2e-5 * max(1.0, ref_lse.abs().max().item())
    scale = qr.grad.abs().max().item()
    err = (dqkv.cpu() - qr.grad).abs().max().item()
    assert err < 3e-5 * scale, f"dqkv err {err} scale {scale}"


@pytest.mark.parametrize("precision,tol", [("32-split", 2e-5), ("bf16-mixed", 3e-2)])
@pytest.mark.parametrize("T,hd", [(130, 128), (77, 64)])
def test_attention_plane_modes_share_the_dropout_mask(H, precision, tol, T, hd):
    """The bf16-pipe variants regenerate the same keep mask as the fp32 kernels (same element index, same pair hash):
    with dropout on, their outputs and gradients stay within the mode's operand precision of the fp32 kernels'."""
    B, Hh = 3, 2
    D = Hh * hd
    g = torch.Generator().manual_seed(T)
    qkv = torch.randn(B, T, 3 * D, generator=g).cuda()
    dout = torch.randn(B, T, D, generator=g).cuda()
    lens = torch.tensor([T, T // 2, 3], dtype=torch.int32).cuda()
    drop = H.Drop(0.2, 4242)
    o0, lse0 = H.attention_fwd(qkv, lens, B, T, Hh, drop)
    g0 = H.attention_bwd(qkv, lens, o0, dout, lse0, B, T, Hh, drop)
    saved = H.get_precision()
    try:
        H.set_precision(precision)
        o1, lse1 = H.attention_fwd(qkv, lens, B, T, Hh, drop)
        g1 = H.attention_bwd(qkv, lens, o1, dout, lse1, B, T, Hh, drop)
    finally:
        H.set_precision(saved)
    valid = (torch.arange(T, device="cuda")[None, :] < lens[:, None])[..., None]
    eo = ((o1 - o0) * valid).abs().max().item() / o0.abs().max().item()
    eg = ((g1 - g0).view(B, T, -1) * valid).abs().max().item() / g0.abs().max().item()
    assert eo < tol and eg < tol, (eo, eg)
    if precision == "bf16-mixed":
        assert eo > 1e-4, "bf16 operands requested, fp32-exact result"


@pytest.mark.parametrize("B,T,Hh,hd,lens", [
    (1, 1, 2, 128, [1]), (2, 5, 2, 64, [5, 2]), (2, 31, 2, 128, [31, 17]), (3, 130, 2, 128, [130, 64, 1]),
    (2, 200, 1, 64, [200, 33]), (4, 648, 2, 128, [648, 500, 40, 333]),
    (32, 648, 2, 128, [648, 430, 40] + [430 + 7 * i for i in range(29)]),
    (3, 1291, 2, 128, [1291, 700, 64]),  # BASELINE configs[4]'s longest utterance: 41 key tiles, a 6.8 MB dS slab per head
])
def test_spilled_ds_backward_equals_the_recomputing_backward(H, B, T, Hh, hd, lens, monkeypatch):
    """``fs2hip_attention_bwd_spill`` (default in "32-true"): the dK/dV kernel writes dS out, dQ = scale * dS . K is its own
    product.  dK and dV are the recomputing kernels' bit for bit (same kernel, one store more); dQ differs only by the rounding
    of S (scale folded into K instead of Q): 2e-5 of its scale -- with attention dropout on, ragged lengths, T not a multiple of
    the tile, utterances shorter than one key tile."""
    g = torch.Generator().manual_seed(B * 77 + T)
    D = Hh * hd
    qkv = torch.randn(B, T, 3 * D, generator=g).cuda()
    dout = torch.randn(B, T, D, generator=g).cuda()
    lens_t = torch.tensor(lens, dtype=torch.int32).cuda()
    assert H.GEMM_BF16 == 0 and H.lib().fs2hip_attention_bwd_spill_supported(hd) == 1
    for drop in (H.NO_DROP, H.Drop(0.2, 1234)):
        o, lse = H.attention_fwd(qkv, lens_t, B, T, Hh, drop)
        monkeypatch.setattr(H, "ATTN_SPILL", False)
        want = H.attention_bwd(qkv, lens_t, o, dout, lse, B, T, Hh, drop)
        monkeypatch.setattr(H, "ATTN_SPILL", True)
        H._SCRATCH.clear()
        got = H.attention_bwd(qkv, lens_t, o, dout, lse, B, T, Hh, drop)
        assert torch.isfinite(got).all()
        assert torch.equal(got[..., D:], want[..., D:]), "dK / dV changed"
        scale = float(want.abs().max())  # (the whole gradient's scale: with one key dQ is zero up to the rounding of p * dP - p * delta)
        err = float((got[..., :D] - want[..., :D]).abs().max())
        assert err < 2e-5 * scale, (drop.p, err, scale)


@pytest.mark.parametrize("B,T,Hh,hd,lens", [
    (1, 1, 2, 128, [1]), (2, 5, 2, 64, [5, 2]), (2, 31, 2, 128, [31, 17]), (3, 130, 2, 128, [130, 64, 1]),
    (2, 200, 1, 64, [200, 33]), (4, 648, 2, 128, [648, 500, 40, 333]),
    (32, 648, 2, 128, [648, 430, 40] + [430 + 7 * i for i in range(29)]), (3, 1291, 2, 128, [1291, 700, 64]),
])
def test_backward_from_the_forward_passes_scores(H, B, T, Hh, hd, lens, monkeypatch):
    """``attention_fwd(save_scores=True)`` + ``attention_bwd(scores=...)``: the training forward writes its masked scores
    out, the dK/dV kernel reads them instead of recomputing K.Q^T.  The forward's o / lse are bit-identical with and without the
    store; the gradients equal the recomputing spilled-dS backward's within the rounding of S (scale folded into Q in the
    forward, into K in the recomputation): 2e-5 of the gradient's scale -- dropout on, ragged lengths, utterances that end
    inside a key tile, key blocks whose second half is all padding (scores never written: masked by the key test)."""
    g = torch.Generator().manual_seed(B * 91 + T)
    D = Hh * hd
    qkv = torch.randn(B, T, 3 * D, generator=g).cuda()
    dout = torch.randn(B, T, D, generator=g).cuda()
    lens_t = torch.tensor(lens, dtype=torch.int32).cuda()
    assert H.attention_scores_kept(hd)
    for drop in (H.NO_DROP, H.Drop(0.2, 4321)):
        o0, lse0 = H.attention_fwd(qkv, lens_t, B, T, Hh, drop)
        o, lse, sc = H.attention_fwd(qkv, lens_t, B, T, Hh, drop, save_scores=True)
        assert sc is not None and sc.shape == (B, Hh, T, (T + 31) // 32 * 32)
        assert torch.equal(o, o0) and torch.equal(lse, lse0)
        want = H.attention_bwd(qkv, lens_t, o, dout, lse, B, T, Hh, drop)
        got = H.attention_bwd(qkv, lens_t, o, dout, lse, B, T, Hh, drop, scores=sc)
        assert torch.isfinite(got).all()
        scale = float(want.abs().max())
        err = float((got - want).abs().max())
        assert err <= 2e-5 * scale, (drop.p, err, scale)  # (one key and it is dropped: everything is exactly zero)
    # and against the plain PyTorch reference (no dropout)
    qr = qkv.cpu().clone().requires_grad_(True)
    ref, _ = ref_attention(qr, lens_t.cpu(), B, T, Hh)
    ref.backward(dout.cpu())
    o, lse, sc = H.attention_fwd(qkv, lens_t, B, T, Hh, save_scores=True)
    got = H.attention_bwd(qkv, lens_t, o, dout, lse, B, T, Hh, scores=sc)
    assert float((got.cpu() - qr.grad).abs().max()) < 3e-5 * float(qr.grad.abs().max())


def test_backward_does_not_read_uninitialised_scratch(H):
    """Odd T: the dK/dV kernel's last {lse', delta'} piece of the last (utterance, head) reaches one pair into the scratch's
    slack.  With the allocator's leftovers poisoned (NaN), every gradient must still be finite and equal the clean run's --
    recomputing, spilled-dS and kept-scores forms."""
    B, T, Hh, hd, lens = 2, 5, 2, 64, [5, 2]
    D = Hh * hd
    g = torch.Generator().manual_seed(17)
    qkv = torch.randn(B, T, 3 * D, generator=g).cuda()
    dout = torch.randn(B, T, D, generator=g).cuda()
    lens_t = torch.tensor(lens, dtype=torch.int32).cuda()
    o, lse, sc = H.attention_fwd(qkv, lens_t, B, T, Hh, save_scores=True)
    clean = H.attention_bwd(qkv, lens_t, o, dout, lse, B, T, Hh, scores=sc)
    for trial in range(8):
        junk = torch.full((1 << 20,), float("nan"), device="cuda")  # freed at once: what the next torch.empty hands out
        del junk
        for kw in ({}, {"scores": sc}):
            got = H.attention_bwd(qkv, lens_t, o, dout, lse, B, T, Hh, **kw)
            assert torch.isfinite(got).all(), (trial, kw.keys())
            assert float((got - clean).abs().max()) <= 2e-5 * float(clean.abs().max())


def test_split_precision_keeps_its_scores_too(H, monkeypatch):
    """"32-split": the three-plane forward writes its scores out as well; the backward pass (fp32 dK/dV kernel + dQ from dS)
    reads them.  Gradients within 3e-5 of PyTorch fp32 and within 2e-5 of the same precision without kept scores."""
    B, T, Hh, hd, lens = 3, 130, 2, 128, [130, 64, 1]
    D = Hh * hd
    g = torch.Generator().manual_seed(23)
    qkv = torch.randn(B, T, 3 * D, generator=g).cuda()
    dout = torch.randn(B, T, D, generator=g).cuda()
    lens_t = torch.tensor(lens, dtype=torch.int32).cuda()
    qr = qkv.cpu().clone().requires_grad_(True)
    ref, _ = ref_attention(qr, lens_t.cpu(), B, T, Hh)
    ref.backward(dout.cpu())
    H.set_precision("32-split")
    try:
        assert H.attention_scores_kept(hd)
        o0, lse0 = H.attention_fwd(qkv, lens_t, B, T, Hh)
        o, lse, sc = H.attention_fwd(qkv, lens_t, B, T, Hh, save_scores=True)
        assert sc is not None and torch.equal(o, o0) and torch.equal(lse, lse0)
        want = H.attention_bwd(qkv, lens_t, o, dout, lse, B, T, Hh)
        got = H.attention_bwd(qkv, lens_t, o, dout, lse, B, T, Hh, scores=sc)
    finally:
        H.set_precision("32-true")
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    assert float((got.cpu() - qr.grad).abs().max()) < 3e-5 * float(qr.grad.abs().max())
