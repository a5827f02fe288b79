"""bf16 tensors through the Conformer convolution module's memory-bound kernels (depthwise convolution with GLU,
BatchNorm + SiLU, their backward passes): the bf16-storage variants read bf16 inputs and must give what the fp32 kernels
give on the same VALUES -- bit for bit where the output type is the same, within one bf16 rounding where the bf16
variant rounds its result."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from fastspeech2_lightning_amd import hip
    hip.lib()
    return hip


def bf16_close(a, b, ulps=1.01):
    """|a - b| <= ulps * 2^-8 * |b| (one bf16 rounding), elementwise, plus a tiny absolute floor."""
    return bool(((a - b).abs() <= ulps * 2.0 ** -8 * b.abs() + 1e-30).all())


@pytest.mark.parametrize("B,T,C,K", [(3, 77, 256, 9), (2, 648, 256, 9), (2, 130, 64, 31), (1, 40, 192, 7)])
def test_depthwise_conv_glu_on_bf16_tensors(H, B, T, C, K):
    g = torch.Generator().manual_seed(T + K)
    x = torch.randn(B * T, 2 * C, generator=g).bfloat16().cuda()
    w = (0.3 * torch.randn(K, C, generator=g)).cuda()
    bias = (0.1 * torch.randn(C, generator=g)).cuda()
    y32, parts32 = H.dwconv_fwd(x.float(), w, bias, B, T, glu=True, stats=True)
    yb, partsb = H.dwconv_fwd(x, w, bias, B, T, glu=True, stats=True)
    assert yb.dtype == torch.bfloat16 and torch.equal(yb, y32.bfloat16())
    # the statistics are those of the rounded outputs: feed them back through the fp32 statistics kernel
    gam, bet = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    st_b = H.bn_finalize(partsb, gam, bet, None, None)
    st_r = H.bn_finalize(H.colstats(yb.float().view(B * T, C)), gam, bet, None, None)
    assert (st_b[2] - st_r[2]).abs().max().item() < 1e-5 and ((st_b[3] - st_r[3]).abs() / st_r[3]).max().item() < 1e-5
    dy = torch.randn(B, T, C, generator=g).bfloat16().cuda()
    dw32, db32 = torch.empty(K, C, device="cuda"), torch.empty(C, device="cuda")
    dwb, dbb = torch.empty(K, C, device="cuda"), torch.empty(C, device="cuda")
    dx32 = H.dwconv_bwd(dy.float(), x.float(), w, dw32, db32, B, T, glu=True, out_dtype=torch.bfloat16)
    dxb = H.dwconv_bwd(dy, x, w, dwb, dbb, B, T, glu=True, out_dtype=torch.bfloat16)
    assert torch.equal(dxb, dx32) and torch.equal(dwb, dw32) and torch.equal(dbb, db32)


@pytest.mark.parametrize("M,C,act,p", [(41472, 256, "silu", 0.0), (1000, 512, "tanh", 0.3), (77, 64, "silu", 0.0)])
def test_batchnorm_activation_on_bf16_tensors(H, M, C, act, p):
    g = torch.Generator().manual_seed(M)
    y = (1.5 * torch.randn(M, C, generator=g) + 0.3).bfloat16().cuda()
    dout = torch.randn(M, C, generator=g).bfloat16().cuda()
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
    parts32, partsb = H.colstats(y.float()), H.colstats(y)
    assert torch.equal(parts32.partial, partsb.partial)
    stats = H.bn_finalize(partsb, gamma, beta, None, None)
    drop = H.Drop(p, 11) if p > 0 else H.NO_DROP
    o32 = H.bn_act_fwd(y.float(), stats, act, drop, bf16_only=True)
    ob = H.bn_act_fwd(y, stats, act, drop, bf16_only=True)
    assert torch.equal(o32, ob)
    dg32, db32, dgb, dbb = (torch.empty(C, device="cuda") for _ in range(4))
    d32 = H.bn_act_bwd(dout.float(), y.float(), stats, dg32, db32, act, drop, bf16_only=True)
    db_ = H.bn_act_bwd(dout, y, stats, dgb, dbb, act, drop, bf16_only=True)
    assert torch.equal(d32, db_) and torch.equal(dg32, dgb) and torch.equal(db32, dbb)
    d32f = H.bn_act_bwd(dout.float(), y.float(), stats, dg32, db32, act, drop)
    dbf = H.bn_act_bwd(dout, y, stats, dgb, dbb, act, drop)
    assert dbf.dtype == torch.float32 and torch.equal(d32f, dbf)
    # one of the two in bf16 (PostNet: the fp32 loss gradient beside a bf16 convolution result, and the reverse)
    for a, b in ((dout.float(), y), (dout, y.float())):
        dmix = H.bn_act_bwd(a, b, stats, dgb, dbb, act, drop)
        assert torch.equal(d32f, dmix) and torch.equal(dg32, dgb) and torch.equal(db32, dbb)


@pytest.mark.parametrize("B,T,C,K,bf", [(2, 648, 256, 9, False), (2, 648, 256, 9, True), (3, 77, 64, 31, False),
                                       (1, 130, 192, 7, True), (2, 64, 128, 3, False), (2, 5, 64, 9, True),
                                       (1, 1, 64, 15, False)])
def test_tiled_glu_depthwise_kernels_give_the_same_bits(H, B, T, C, K, bf):
    """The LDS-tiled GLU kernels (wide loads, the GLU applied once per element) run the per-thread-window kernels'
    arithmetic in the same order: identical results, statistics and parameter-gradient partial sums."""
    import os
    g = torch.Generator().manual_seed(T + K + C)
    x = torch.randn(B * T, 2 * C, generator=g).cuda()
    dy = torch.randn(B, T, C, generator=g).cuda()
    if bf:
        x, dy = x.bfloat16(), dy.bfloat16()
    w = (0.3 * torch.randn(K, C, generator=g)).cuda()
    bias = (0.1 * torch.randn(C, generator=g)).cuda()
    out_dt = torch.bfloat16 if bf else torch.float32

    def run():
        y, parts = H.dwconv_fwd(x, w, bias, B, T, glu=True, stats=True)
        dw, db = torch.empty(K, C, device="cuda"), torch.empty(C, device="cuda")
        dx = H.dwconv_bwd(dy, x, w, dw, db, B, T, glu=True, out_dtype=out_dt)
        torch.cuda.synchronize()
        return y, parts.partial, dx, dw, db

    prev = os.environ.get("FS2_DWCONV_TILE")
    try:
        os.environ["FS2_DWCONV_TILE"] = "0"
        ref = run()
        os.environ["FS2_DWCONV_TILE"] = "1"
        got = run()
    finally:
        if prev is None:
            os.environ.pop("FS2_DWCONV_TILE", None)
        else:
            os.environ["FS2_DWCONV_TILE"] = prev
    for name, a, b in zip(("y", "stats", "dx", "dw", "db"), ref, got):
        assert torch.equal(a, b), name


@pytest.mark.parametrize("B,T,Cin,Cout", [(3, 70, 80, 512), (2, 5, 16, 24), (1, 1, 8, 8)])
def test_im2col_taps_gemm_is_the_k_tap_convolution(B, T, Cin, Cout):
    """Round 5: the PostNet's 80-mel-bin convolutions as ONE plain GEMM over side-by-side input rows
    (``hip.im2col_taps`` + ``hip.matmul_kn``): forward (dir = +1, weight transposed per tap) and data gradient (dir = -1,
    weight as stored) against ``F.conv1d`` / its transposed form on the SAME bf16-rounded operands, fp32 and bf16 inputs,
    utterance boundaries zero-padded (T = 1: only the centre tap sees data)."""
    import torch.nn.functional as F
    from fastspeech2_lightning_amd import hip as H
    g = torch.Generator().manual_seed(5)
    k = 5
    x = torch.randn(B, T, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) / (Cin * k) ** 0.5          # reference layout
    bias = torch.randn(Cout, generator=g)
    xb, wb = x.bfloat16().float(), w.bfloat16().float()
    want = F.conv1d(xb.transpose(1, 2), wb, bias, padding=2).transpose(1, 2)
    w_native = w.permute(2, 0, 1).contiguous().cuda()                        # [tap][Cout][Cin], the store's "convk" layout
    for src in (x.cuda(), x.cuda().bfloat16()):
        cols = H.im2col_taps(src.view(B * T, Cin), B, T, k)
        assert cols.dtype == torch.bfloat16 and cols.shape == (B * T, k * Cin)
        # column block `tap` of row (b, t) is x[b, t + tap - 2] (zeros outside the utterance)
        pad = F.pad(xb, (0, 0, 2, 2))
        for tap in range(k):
            assert torch.equal(cols[:, tap * Cin:(tap + 1) * Cin].float().cpu().view(B, T, Cin), pad[:, tap:tap + T])
        wt = H.transpose_cast_bf16(w_native).view(k * Cin, Cout)
        got = H.matmul_kn(cols, wt, bias.cuda()).view(B, T, Cout).cpu()
        assert float((got - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max())) * (k * Cin) ** 0.5
    # data gradient: dx[t] = sum_tap W[tap]^T dy[t - (tap - 2)]
    dy = torch.randn(B, T, Cout, generator=g)
    dyb = dy.bfloat16().float()
    want_dx = F.conv_transpose1d(dyb.transpose(1, 2), wb, padding=2).transpose(1, 2)
    dcols = H.im2col_taps(dy.cuda().bfloat16().view(B * T, Cout), B, T, k, direction=-1)
    # the stored layout of a Cin -> Cout conv weight [tap][Cout][Cin] is, flattened, the [taps * Cout][Cin] operand
    got_dx = H.matmul_kn(dcols, w_native.bfloat16().view(k * Cout, Cin)).view(B, T, Cin).cpu()
    assert float((got_dx - want_dx).abs().max()) < 2e-5 * max(1.0, float(want_dx.abs().max())) * (k * Cout) ** 0.5
