"""GPU parity of the learned-alignment kernels against the oracle's restatement of
fs2/attn/attention.py, fs2/attn/alignment.py and fs2/attn/attention_loss.py.
MAS and durations are bit-exact; floating point at 2e-5 of scale (CTC gradient 1e-4)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fs2_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from fastspeech2_lightning_amd import hip
    hip.lib()
    return hip


def close(a, b, tol=2e-5, msg=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    fin = torch.isfinite(b)
    assert torch.equal(torch.isfinite(a), fin), msg + ": non-finite pattern"
    scale = max(float(b[fin].abs().max()), 1e-6)
    err = float((a[fin] - b[fin]).abs().max()) / scale
    assert err < tol, f"{msg}: rel err {err:.3e}"


def make_case(B=3, T1=37, T2=11, C=80, seed=0):
    g = torch.Generator().manual_seed(seed)
    q, k = torch.randn(B, T1, C, generator=g) * 3, torch.randn(B, T2, C, generator=g) * 3
    key_lens = torch.tensor([T2, max(2, T2 - 3), max(2, T2 // 2)][:B], dtype=torch.int32)
    q_lens = torch.tensor([T1, T1 - 5, max(T2, T1 // 2)][:B], dtype=torch.int32)
    prior = O.beta_binomial_prior(q_lens, key_lens, T1, T2)
    return q, k, key_lens, q_lens, prior


def ref_attention(q, k, key_lens, prior):
    d = ((q[:, :, None, :] - k[:, None, :, :]) ** 2).sum(-1)
    logits = -0.0005 * d
    lp = F.log_softmax(logits, dim=2) + torch.log(prior + 1e-8)
    mask = torch.arange(k.shape[1])[None, None, :] >= key_lens[:, None, None]
    soft = F.softmax(lp.masked_fill(mask, -float("inf")), dim=2)
    return logits, lp, soft


def test_dist_softmax(H):
    q, k, key_lens, q_lens, prior = make_case()
    logits, lp, soft = ref_attention(q, k, key_lens, prior)
    got = H.attn_dist(q.cuda(), k.cuda())
    close(got, logits, msg="logits")
    glp, gsoft = H.attn_softmax(got, prior.cuda(), key_lens.cuda())
    close(glp, lp, msg="logprob")
    close(gsoft, soft, msg="soft")


def test_mas_bit_exact(H, golden_dir):
    g = dict(np.load(golden_dir / "units.npz"))
    cases = [g[f"mas/{t}/in"] for t in ("rand", "ties", "t2_2", "square")]
    rng = np.random.default_rng(0)
    for t1, t2 in [(200, 37), (648, 128), (64, 64), (90, 2), (300, 200)]:
        x = rng.standard_normal((t1, t2)).astype(np.float32)
        x = x - np.log(np.exp(x).sum(1, keepdims=True))
        cases.append(x.astype(np.float32))
        cases.append(np.round(x).astype(np.float32))  # tie-heavy
    refs = [O.mas_width1(c.copy()) if i >= 4 else [g[f"mas/{t}/out"] for t in ("rand", "ties", "t2_2", "square")][i]
            for i, c in enumerate(cases)]
    # two launches: every case (padded to 200 tokens -> the workgroup kernel) and the cases of at most 128 tokens
    # (-> the one-wavefront-per-utterance kernel); both must reproduce the reference's search bit for bit
    for group in (list(range(len(cases))), [i for i, c in enumerate(cases) if c.shape[1] <= 128]):
        sel = [cases[i] for i in group]
        Tm, Ts = max(c.shape[0] for c in sel), max(c.shape[1] for c in sel)
        assert (Ts <= 128) == (len(group) < len(cases))
        x = torch.zeros(len(sel), Tm, Ts)
        for n, c in enumerate(sel):
            x[n, : c.shape[0], : c.shape[1]] = torch.tensor(c)
        in_lens = torch.tensor([c.shape[1] for c in sel], dtype=torch.int32)
        out_lens = torch.tensor([c.shape[0] for c in sel], dtype=torch.int32)
        hard, idx, dur = H.mas(x.cuda(), in_lens.cuda(), out_lens.cuda(), is_log=True)
        for n, i in enumerate(group):
            c, ref = cases[i], refs[i]
            got = hard[n].cpu().numpy()
            np.testing.assert_array_equal(got[: c.shape[0], : c.shape[1]], ref, err_msg=f"case {i} (Ts={Ts})")
            assert got.sum() == c.shape[0]
            np.testing.assert_array_equal(dur[n].cpu().numpy()[: c.shape[1]], ref.sum(0).astype(np.int32))
            np.testing.assert_array_equal(idx[n].cpu().numpy()[: c.shape[0]], ref.argmax(1))
            assert (idx[n].cpu().numpy()[c.shape[0]:] == -1).all()


def test_avg_variance(H, golden_dir):
    g = dict(np.load(golden_dir / "units.npz"))
    var, durs = torch.tensor(g["avg/var"]), torch.tensor(g["avg/durs"])
    cum, _ = H.duration_cumsum(durs.cuda(), var.shape[1])
    got = H.avg_variance(var.cuda(), cum)
    np.testing.assert_allclose(got.cpu().numpy(), g["avg/out"], rtol=2e-6, atol=1e-7)


def test_ctc_bin_and_backward(H):
    q, k, key_lens, q_lens, prior = make_case(B=3, T1=41, T2=12, seed=3)
    qr, kr = q.clone().requires_grad_(True), k.clone().requires_grad_(True)
    logits, lp, soft = ref_attention(qr, kr, key_lens, prior)
    lp.retain_grad(); soft.retain_grad()
    hard = O.binarize_attention(soft[:, None], key_lens, q_lens)[:, 0]
    ctc = O.attention_ctc_loss(lp[:, None], key_lens, q_lens) * 0.1
    binl = O.attention_bin_loss(hard, soft) * 0.07
    (ctc + binl).backward()
    # forward on the GPU
    glogits = H.attn_dist(q.cuda(), k.cuda())
    glp, gsoft = H.attn_softmax(glogits, prior.cuda(), key_lens.cuda())
    ghard, gidx, gdur = H.mas(gsoft, key_lens.cuda(), q_lens.cuda())
    assert torch.equal(ghard.cpu(), hard)
    slots = torch.zeros(2, device="cuda")
    dlp = H.attn_ctc_loss(glp, key_lens.cuda(), q_lens.cuda(), 0.1, slots[0:1])
    coef = H.attn_bin_loss(gsoft, gidx, 0.07, slots[1:2])
    close(slots[0:1], ctc.detach().reshape(1), 1e-5, "ctc value")
    close(slots[1:2], binl.detach().reshape(1), 1e-5, "bin value")
    # d(ctc)/d attn_logprob alone
    lp2 = lp.detach().clone().requires_grad_(True)
    (O.attention_ctc_loss(lp2[:, None], key_lens, q_lens) * 0.1).backward()
    close(dlp, lp2.grad, 1e-4, "ctc grad")
    dlogits = H.attn_softmax_bwd(glogits, gsoft, dlp, gidx, coef)
    dq, dk = H.attn_dist_bwd(dlogits, q.cuda(), k.cuda())
    close(dq, qr.grad, 2e-4, "dq")
    close(dk, kr.grad, 2e-4, "dk")


@pytest.mark.parametrize("B,T1,T2,C", [(2, 648, 128, 80), (3, 131, 70, 80), (2, 65, 1, 20), (1, 33, 65, 128),
                                       (2, 40, 300, 80), (1, 17, 9, 200)])
def test_dist_bwd_tiled_kernels(H, B, T1, T2, C, monkeypatch):
    """The LDS-tiled distance backward (csrc/aligner.hip, dist_bwd_{q,k}_tile_kernel) against autograd of the reference's
    logits (fs2/model.py aligner: -temperature * squared distance, temperature 0.0005) and, bit for bit, against the
    one-wavefront-per-row kernels it replaces (same terms in the same order).  Shapes: the benchmark's (648 frames x 128
    symbols), ragged tails on both axes, one key, a 128-channel and a >64 KB key tile / >128-channel case (which fall back)."""
    g = torch.Generator().manual_seed(B * 1000 + T1 + T2)
    q = torch.randn(B, T1, C, generator=g)
    k = torch.randn(B, T2, C, generator=g)
    dl = torch.randn(B, T1, T2, generator=g) * 0.1
    qr, kr = q.clone().requires_grad_(), k.clone().requires_grad_()
    logits = -0.0005 * ((qr[:, :, None] - kr[:, None]) ** 2).sum(-1)
    (logits * dl).sum().backward()
    monkeypatch.delenv("FS2_DIST_BWD_TILE", raising=False)
    dq, dk = H.attn_dist_bwd(dl.cuda(), q.cuda(), k.cuda())
    close(dq, qr.grad, 2e-5, "dq")
    close(dk, kr.grad, 2e-5, "dk")
    monkeypatch.setenv("FS2_DIST_BWD_TILE", "0")
    dq0, dk0 = H.attn_dist_bwd(dl.cuda(), q.cuda(), k.cuda())
    assert torch.equal(dq, dq0) and torch.equal(dk, dk0)
    only_q, none_k = H.attn_dist_bwd(dl.cuda(), q.cuda(), k.cuda(), want_dk=False)
    assert none_k is None and torch.equal(only_q, dq0)
