"""The weights-stationary streaming form of the bf16-storage GEMM core (csrc/gemm_ws.hip, tiles 30 / 31): W fragments in
registers for the whole launch, A streamed through LDS in 32-row tiles, two wavefronts per SIMD half a period apart,
counted vector-memory waits.  It performs the same sixteen MFMAs per output block in the same order as the tiled kernels
and shares their epilogue arithmetic and dropout masks, so every result must equal the tiled kernel's (tile 22) BIT FOR
BIT -- fp32 and bf16 results, the pre-activation output, residual, act' with bf16 / fp32 operands, dropout on -- for row
counts from one partial tile to thousands of tiles per workgroup stream, ragged column edges, every slice count; and the
data gradient through the transposed weight mirror must equal the one computed from the weight as stored.

reference call sites: the Conformer's K = 256 projections (torchaudio ConformerLayer: ffn Linear(256, 1024), in_proj,
out_proj, pointwise convolutions; fs2/model.py:193, :241) and their data gradients."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.tuned_tiles]


@pytest.fixture(scope="module")
def H():
    from fastspeech2_lightning_amd import hip
    hip.lib()
    return hip


class only_tile:
    def __init__(self, H, tile):
        self.H, self.tile = H, tile

    def __enter__(self):
        self.saved = self.H.GEMM_TILES_B
        self.H.GEMM_TILES_B = (self.tile,)
        self.H._TILE_CACHE.clear()
        return self

    def __exit__(self, *exc):
        self.H.GEMM_TILES_B = self.saved
        self.H._TILE_CACHE.clear()
        return False

    def ran(self):
        """every GEMM launched inside the block took this tile (none was refused and handed to the heuristic)"""
        return bool(self.H._TILE_CACHE) and set(self.H._TILE_CACHE.values()) == {self.tile}


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def reference(M, N):
    """float64 values of every form without dropout (the inputs are the bf16-rounded tensors ``forms`` builds)."""
    import torch.nn.functional as F
    K = 256
    x = rnd(M, K, seed=1).bfloat16().double()
    w = rnd(N, K, seed=2, scale=K ** -0.5).bfloat16().double()
    b = rnd(N, seed=3).double()
    r = rnd(M, N, seed=4).double()
    aux32 = rnd(M, N, seed=5)
    u = x @ w.t() + b
    ub = u.float().bfloat16().double()
    g = x @ w.t()

    def dact(aux):
        a = aux.double().requires_grad_(True)
        F.silu(a).backward(g)
        return a.grad
    return {"store32": u, "store16": u, "nobias16": g, "silu16": F.silu(ub), "pre16": u, "silu32": F.silu(u), "pre32": u,
            "relu16": torch.relu(u), "resid32": r + 0.5 * u, "dact16": dact(aux32.bfloat16()), "dact32aux": dact(aux32),
            "dgrad16": g, "dgrad32": g}


def forms(H, M, N, drop):
    """Every epilogue the step runs on a K = 256 GEMM; returns {name: tensor}."""
    K = 256
    x = rnd(M, K, seed=1).bfloat16().cuda()
    w = rnd(N, K, seed=2, scale=K ** -0.5).bfloat16().cuda()
    b = rnd(N, seed=3).cuda()
    r = rnd(M, N, seed=4).cuda()
    aux32 = rnd(M, N, seed=5).cuda()
    auxb = aux32.bfloat16()
    bf = torch.bfloat16
    out = {}
    out["store32"] = H.linear_fwd(x, w, b)
    out["store16"] = H.linear_fwd(x, w, b, out_dtype=bf)
    out["nobias16"] = H.linear_fwd(x, w, None, out_dtype=bf)
    u16 = torch.empty(M, N, device="cuda", dtype=bf)
    out["silu16"] = H.linear_fwd(x, w, b, epi=H.EPI_ACT, act="silu", out_pre=u16, drop=drop, out_dtype=bf)
    out["pre16"] = u16
    u32 = torch.empty(M, N, device="cuda")
    out["silu32"] = H.linear_fwd(x, w, b, epi=H.EPI_ACT, act="silu", out_pre=u32, drop=drop)
    out["pre32"] = u32
    out["relu16"] = H.linear_fwd(x, w, b, epi=H.EPI_ACT, act="relu", drop=drop, out_dtype=bf)
    out["resid32"] = H.linear_fwd(x, w, b, epi=H.EPI_RESID, resid=r, res_scale=0.5, drop=drop)
    # the data-gradient forms in the forward orientation: dy [M, 256] . Wt^T with Wt = [N_out, 256]
    out["dact16"] = H.linear_bwd_data(x, None, epi=H.EPI_DACT, act="silu", aux=auxb, drop=drop, out_dtype=bf, wt=w)
    out["dact32aux"] = H.linear_bwd_data(x, None, epi=H.EPI_DACT, act="silu", aux=aux32, drop=drop, out_dtype=bf, wt=w)
    out["dgrad16"] = H.linear_bwd_data(x, None, out_dtype=bf, wt=w)
    out["dgrad32"] = H.linear_bwd_data(x, None, wt=w)
    return out


@pytest.mark.parametrize("tile", [30, 31])
@pytest.mark.parametrize("M,N", [(1, 256), (33, 256), (1000, 512), (4100, 1024), (20736, 768), (43008, 1024), (777, 264),
                                 (2048, 1288), (70000, 256)])
def test_streaming_kernel_equals_the_tiled_kernel_bit_for_bit(H, tile, M, N):
    # without dropout: both kernels against float64 (which of the two is wrong, should they ever differ)
    ref = reference(M, N)
    for t in (22, tile):
        with only_tile(H, t):
            got = forms(H, M, N, H.NO_DROP)
        for name, want in ref.items():
            g = got[name].double().cpu()
            tol = (2.0 ** -7 if got[name].dtype == torch.bfloat16 else 1e-4) * max(1.0, float(want.abs().max()))
            assert float((g - want).abs().max()) < tol, (t, name, float((g - want).abs().max()))
    step = torch.full((1,), 3, dtype=torch.int64, device="cuda")
    drop = H.Drop(0.2, 0x5eed, step)
    with only_tile(H, 22):
        want = forms(H, M, N, drop)
    with only_tile(H, tile) as t:
        got = forms(H, M, N, drop)
        taken = {k[:3] + (k[8],) for k, v in H._TILE_CACHE.items() if v == tile}
        refused = {k for k, v in H._TILE_CACHE.items() if v != tile}
    # 64 columns per wavefront refuse fp32 results with operand quads (register budget): those launches fall back
    assert taken, "the streaming kernel never ran"
    if tile == 31:
        assert not refused, refused
    for name, w in want.items():
        g = got[name]
        assert g.dtype == w.dtype and g.shape == w.shape
        assert torch.equal(g, w), (name, float((g.float() - w.float()).abs().max()))


def test_data_gradient_through_the_transposed_mirror(H):
    """``ParamStore.pbt``: W [256, 1024] kept as bf16 W^T [1024, 256] by one multi-matrix launch; dz . W from the mirror
    (forward orientation) equals dz . W from the weight as stored (reduction-major operand) to the last bit of the fp32
    accumulation order -- both run sixteen 16-deep MFMAs in ascending reduction order."""
    M = 5000
    w = rnd(256, 1024, seed=1, scale=1 / 16)
    w2 = rnd(256, 256, seed=2, scale=1 / 16)
    wt, wt2 = torch.empty(1024, 256, device="cuda", dtype=torch.bfloat16), torch.empty(256, 256, device="cuda", dtype=torch.bfloat16)
    H.transpose_cast_bf16_multi([(w.cuda(), wt), (w2.cuda(), wt2)])
    assert torch.equal(wt.cpu(), w.bfloat16().t().contiguous()) and torch.equal(wt2.cpu(), w2.bfloat16().t().contiguous())
    dz = rnd(M, 256, seed=3).bfloat16().cuda()
    wb = w.bfloat16().cuda()
    ref = dz.double().cpu() @ wb.double().cpu()
    with only_tile(H, 22):
        stored = H.linear_bwd_data(dz, wb, out_dtype=torch.bfloat16)
    with only_tile(H, 30) as t:
        mirror = H.linear_bwd_data(dz, wb, out_dtype=torch.bfloat16, wt=wt)
        assert t.ran()
    err = float((mirror.double().cpu() - ref).abs().max())
    assert err < 2.0 ** -7 * float(ref.abs().max())
    assert float((mirror.float() - stored.float()).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())


def test_train_step_is_the_same_with_and_without_the_streaming_kernel(H):
    """A bf16-mixed train step of a one-layer d = 256 model: losses and every gradient with the streaming tiles allowed
    (forced wherever they apply) against the tiled kernels only -- equal up to the fp32 summation order of the
    data gradients that now run from the transposed mirror (1e-5 relative L2)."""
    from fastspeech2_lightning_amd.config import FastSpeech2Config, Stats
    from fastspeech2_lightning_amd.model import FastSpeech2
    from fastspeech2_lightning_amd.synthetic import DEFAULT_STATS, synthetic_batch
    conf = dict(layers=1, dropout=0.1)
    vp = dict(dropout=0.0)
    config = FastSpeech2Config(model=dict(encoder=conf, decoder=conf, learn_alignment=False,
                                          variance_predictors=dict(energy=vp, pitch=vp, duration=vp)),
                               text=dict(symbols=dict(letters=[f"s{i}" for i in range(40)])))
    batch = synthetic_batch(B=3, ts_lo=20, ts_hi=40, n_symbols=41, n_mels=80, seed=3, dur_hi=6)
    res = {}
    for tiles in ((22,), (30, 31, 22)):
        saved = H.GEMM_TILES_B
        H.GEMM_TILES_B = tiles
        H._TILE_CACHE.clear()
        try:
            model = FastSpeech2(config, Stats(**DEFAULT_STATS), seed=11, precision="bf16-mixed")
            model.train()
            if len(tiles) > 1:   # the tuner would pick by time: force the streaming kernels wherever they are legal
                orig = H._tune_tile

                def forced(a, orig=orig):
                    if a.operand_bf16 == 4:
                        for t in (30, 31):
                            a.tile = t
                            if H.lib().fs2hip_gemm(H.C.byref(a), H._stream()) == 0:
                                H._TILE_CACHE[H._tile_key(a)] = t
                                return t
                    return orig(a)
                H._tune_tile = forced
            with torch.no_grad():
                model.training_step(batch)
            res[tiles] = (dict(model.last_losses), {k: v.clone() for k, v in model.store.grad_state_dict().items()},
                          sorted(set(H._TILE_CACHE.values())))
        finally:
            H.GEMM_TILES_B = saved
            if len(tiles) > 1:
                H._tune_tile = orig
            H._TILE_CACHE.clear()
    (l0, g0, t0), (l1, g1, t1) = res[(22,)], res[(30, 31, 22)]
    assert 30 in t1 or 31 in t1, t1
    for k, v in l0.items():
        assert abs(float(l1[k]) - float(v)) <= 1e-5 * max(1.0, abs(float(v))), (k, float(l1[k]), float(v))
    num = sum(float((g1[k] - g).pow(2).sum()) for k, g in g0.items())
    den = sum(float(g.pow(2).sum()) for g in g0.values())
    assert (num / den) ** 0.5 < 1e-3, (num / den) ** 0.5


# ---- the exact-fp32 form (csrc/gemm_ws32.hip, tile 32) ---------------------------------------------------------------------
class only_tile32:
    """fp32 GEMMs on one tile id (``GEMM_TILES``)."""
    def __init__(self, H, tile):
        self.H, self.tile = H, tile

    def __enter__(self):
        self.saved = self.H.GEMM_TILES
        self.H.GEMM_TILES = (self.tile,)
        self.H._TILE_CACHE.clear()
        return self

    def __exit__(self, *exc):
        self.H.GEMM_TILES = self.saved
        self.H._TILE_CACHE.clear()
        return False


def forms32(H, M, N, drop):
    K = 256
    x = rnd(M, K, seed=1).cuda()
    w = rnd(N, K, seed=2, scale=K ** -0.5).cuda()
    b = rnd(N, seed=3).cuda()
    r = rnd(M, N, seed=4).cuda()
    aux = rnd(M, N, seed=5).cuda()
    out = {}
    out["store"] = H.linear_fwd(x, w, b)
    out["nobias"] = H.linear_fwd(x, w, None)
    u = torch.empty(M, N, device="cuda")
    out["silu"] = H.linear_fwd(x, w, b, epi=H.EPI_ACT, act="silu", out_pre=u, drop=drop)
    out["pre"] = u
    out["relu"] = H.linear_fwd(x, w, b, epi=H.EPI_ACT, act="relu", drop=drop)
    out["resid"] = H.linear_fwd(x, w, b, epi=H.EPI_RESID, resid=r, res_scale=0.5, drop=drop)
    out["dact"] = H.linear_bwd_data(x, None, epi=H.EPI_DACT, act="silu", aux=aux, drop=drop, wt=w)
    out["dgrad"] = H.linear_bwd_data(x, None, wt=w)
    return out


@pytest.mark.parametrize("M,N", [(1, 256), (33, 256), (1000, 512), (4100, 1024), (20736, 768), (20736, 1024), (777, 260),
                                 (2048, 1284), (70000, 256)])
def test_fp32_streaming_kernel_equals_the_tiled_kernel_bit_for_bit(H, M, N):
    """Tile 32 against tile 7 (64 x 64, one accumulation chain per output over ascending K-tiles with the same
    lane-to-k map): the same fp32 sums in the same order, the same epilogue function -- equal to the last bit, dropout on;
    and both against float64 at the fp32 kernels' tolerance."""
    import torch.nn.functional as F
    assert H.get_precision() == "32-true"
    K = 256
    x, w, b = rnd(M, K, seed=1).double(), rnd(N, K, seed=2, scale=K ** -0.5).double(), rnd(N, seed=3).double()
    r, aux = rnd(M, N, seed=4).double(), rnd(M, N, seed=5).double().requires_grad_(True)
    u = x @ w.t() + b
    g = x @ w.t()
    F.silu(aux).backward(g)
    ref = {"store": u, "nobias": g, "silu": F.silu(u), "pre": u, "relu": torch.relu(u), "resid": r + 0.5 * u,
           "dact": aux.grad, "dgrad": g}
    with only_tile32(H, 32):
        got = forms32(H, M, N, H.NO_DROP)
        assert set(H._TILE_CACHE.values()) == {32}, H._TILE_CACHE
    for name, want in ref.items():
        err = float((got[name].double().cpu() - want).abs().max())
        assert err < 4e-6 * 16 * max(1.0, float(want.abs().max())), (name, err)
    step = torch.full((1,), 5, dtype=torch.int64, device="cuda")
    drop = H.Drop(0.2, 0xabcde, step)
    with only_tile32(H, 7):
        want = forms32(H, M, N, drop)
    with only_tile32(H, 32):
        got = forms32(H, M, N, drop)
    for name, wv in want.items():
        assert torch.equal(got[name], wv), (name, float((got[name] - wv).abs().max()))


def test_fp32_train_step_with_the_transposed_mirrors():
    """32-true, d = 256: the K = 256 data gradients run in the forward orientation from the fp32 W^T mirrors
    (``ParamStore.pt``, refreshed once per forward pass) -- losses and gradients equal the run without the mirrors up to the
    fp32 summation order (1e-6 relative)."""
    from fastspeech2_lightning_amd import hip as H
    from fastspeech2_lightning_amd import model as MM
    from fastspeech2_lightning_amd.config import FastSpeech2Config, Stats
    from fastspeech2_lightning_amd.synthetic import DEFAULT_STATS, synthetic_batch
    conf = dict(layers=1, dropout=0.1)
    vp = dict(dropout=0.0)
    config = FastSpeech2Config(model=dict(encoder=conf, decoder=conf, learn_alignment=False,
                                          variance_predictors=dict(energy=vp, pitch=vp, duration=vp)),
                               text=dict(symbols=dict(letters=[f"s{i}" for i in range(40)])))
    batch = synthetic_batch(B=3, ts_lo=20, ts_hi=40, n_symbols=41, n_mels=80, seed=3, dur_hi=6)
    res = {}
    saved = MM.FP32_TRANSPOSED
    try:
        for on in (False, True):
            MM.FP32_TRANSPOSED = on
            H._TILE_CACHE.clear()
            model = MM.FastSpeech2(config, Stats(**DEFAULT_STATS), seed=11)
            model.train()
            with torch.no_grad():
                model.training_step(batch)
            assert (model.store.pt("decoder.conformer_layers.0.ffn1.sequential.4.weight") is not None) == on
            res[on] = (dict(model.last_losses), {k: v.clone() for k, v in model.store.grad_state_dict().items()})
    finally:
        MM.FP32_TRANSPOSED = saved
        H._TILE_CACHE.clear()
    for k, v in res[False][0].items():
        assert abs(float(res[True][0][k]) - float(v)) <= 1e-6 * max(1.0, abs(float(v))), k
    num = sum(float((res[True][1][k] - g).pow(2).sum()) for k, g in res[False][1].items())
    den = sum(float(g.pow(2).sum()) for g in res[False][1].values())
    assert (num / den) ** 0.5 < 1e-5, (num / den) ** 0.5


# ---- K = 1024 (csrc/gemm_ws4.hip, tile 33) --------------------------------------------------------------------------------
def forms_k1024(H, M, N, drop):
    """The two K = 1024 GEMMs of a feed-forward module under bf16 operand storage: the second Linear with its residual +
    dropout epilogue (fp32 result), and the first Linear's data gradient through the transposed mirror (bf16 / fp32)."""
    K = 1024
    x = rnd(M, K, seed=1).bfloat16().cuda()
    w = rnd(N, K, seed=2, scale=K ** -0.5).bfloat16().cuda()
    b = rnd(N, seed=3).cuda()
    r = rnd(M, N, seed=4).cuda()
    bf = torch.bfloat16
    return {
        "resid32": H.linear_fwd(x, w, b, epi=H.EPI_RESID, resid=r, res_scale=0.5, drop=drop),
        "store32": H.linear_fwd(x, w, b),
        "store16": H.linear_fwd(x, w, b, out_dtype=bf),
        "dgrad16": H.linear_bwd_data(x, None, out_dtype=bf, wt=w),
        "dgrad32": H.linear_bwd_data(x, None, wt=w),
    }


@pytest.mark.parametrize("tile", [33])
@pytest.mark.parametrize("M,N", [(1, 256), (33, 256), (1000, 128), (4100, 256), (8192, 256), (43008, 256), (777, 264),
                                 (2048, 1000), (70000, 256)])
def test_k1024_streaming_kernel_equals_the_tiled_kernel_bit_for_bit(H, tile, M, N):
    """Tile 33: four wavefronts, W[32 columns][1024] per wavefront in 224 accumulation + 32 vector registers (inline-asm
    MFMAs), A in 16 KB chunks through an eight-stage ring, counted waits.  The same 64 MFMAs per output block in the same
    order as the tiled kernels, their epilogue arithmetic and masks: results equal tile 22's bit for bit, and both are
    checked against float64."""
    K = 1024
    x = rnd(M, K, seed=1).bfloat16().double()
    w = rnd(N, K, seed=2, scale=K ** -0.5).bfloat16().double()
    u = x @ w.t() + rnd(N, seed=3).double()
    ref = {"resid32": rnd(M, N, seed=4).double() + 0.5 * u, "store32": u, "store16": u, "dgrad16": x @ w.t(), "dgrad32": x @ w.t()}
    for t in (22, tile):
        with only_tile(H, t) as ot:
            got = forms_k1024(H, M, N, H.NO_DROP)
            assert ot.ran(), (t, dict(H._TILE_CACHE))
        for name, want in ref.items():
            g = got[name].double().cpu()
            tol = (2.0 ** -7 if got[name].dtype == torch.bfloat16 else 2e-4) * max(1.0, float(want.abs().max()))
            assert float((g - want).abs().max()) < tol, (t, name, float((g - want).abs().max()))
    step = torch.full((1,), 3, dtype=torch.int64, device="cuda")
    drop = H.Drop(0.2, 0x5eed, step)
    with only_tile(H, 22):
        want = forms_k1024(H, M, N, drop)
    with only_tile(H, tile) as ot:
        got = forms_k1024(H, M, N, drop)
        assert ot.ran(), dict(H._TILE_CACHE)
    for name, w_ in want.items():
        g = got[name]
        assert g.dtype == w_.dtype and g.shape == w_.shape
        assert torch.equal(g, w_), (name, float((g.float() - w_.float()).abs().max()))


def test_k1024_streaming_kernel_refuses_what_it_does_not_take(H):
    """K != 1024, an activation epilogue or a pre-activation output: tile 33 answers EINVAL and the launch takes the
    library's heuristic tile -- the result is still right."""
    x = rnd(500, 512, seed=1).bfloat16().cuda()
    w = rnd(256, 512, seed=2, scale=1 / 22).bfloat16().cuda()
    with only_tile(H, 33) as ot:
        y = H.linear_fwd(x, w, None)
        assert not ot.ran()
    assert float((y.double().cpu() - x.double().cpu() @ w.double().cpu().t()).abs().max()) < 1e-3
    x = rnd(500, 1024, seed=1).bfloat16().cuda()
    w = rnd(256, 1024, seed=2, scale=1 / 32).bfloat16().cuda()
    with only_tile(H, 33) as ot:
        y = H.linear_fwd(x, w, None, epi=H.EPI_ACT, act="silu", out_dtype=torch.bfloat16)
        assert not ot.ran()
    want = torch.nn.functional.silu(x.double().cpu() @ w.double().cpu().t())
    assert float((y.double().cpu() - want).abs().max()) < 2.0 ** -7 * max(1.0, float(want.abs().max()))
