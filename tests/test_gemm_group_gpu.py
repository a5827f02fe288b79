"""Grouped GEMM launches (``fs2hip_gemm_grouped``, ``hip.gemm_group``; csrc/gemm2.hip ``gemm2g_kernel``, csrc/gemm_bf16.hip
``gemmbg_kernel``): up to eight independent GEMMs in one grid, each member's workgroups running the tile code of a launch
of its own on the member's own arguments.  So every member's result must equal its own launch's BIT FOR BIT -- forward
(bias + ReLU epilogue), data gradient, weight gradient with split-K slabs and the bias-gradient column sums, fp32 and bf16
operand storage, equal and unequal member shapes, ragged row counts, every tile the grouped kernels are built for -- and
a training step with grouping on must equal the step with it off in every loss term, gradient and weight.

reference call sites: the three variance predictors' pointwise convolutions (fs2/blocks.py:14-16 via
fs2/variance_adaptor.py:18-62; independent in training, :309-352) and the weight gradients of an encoder Conformer layer
(fs2/model.py:95-107)."""
import ctypes

import pytest
import torch

from fastspeech2_lightning_amd import hip as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(DEV).to(dtype)


class grouping:
    """``H.GEMM_GROUP`` for the block, group statistics zeroed; ``tile``: force the grouped launches' tile."""

    def __init__(self, on, tile=None):
        self.on, self.tile = on, tile

    def __enter__(self):
        self.saved = H.GEMM_GROUP, H._group_tile
        H.GEMM_GROUP = self.on
        if self.tile is not None:
            H._group_tile = lambda arr, n, members, stream, _t=self.tile: _t
        for k in H.GROUP_STATS:
            H.GROUP_STATS[k] = 0
        return self

    def __exit__(self, *exc):
        H.GEMM_GROUP, H._group_tile = self.saved
        return False


def predictor_triple(dtype, M, widths, run):
    """The three kinds of GEMM of one predictor layer, for members of the given (cin, cout) widths; ``run(section)``
    wraps every phase (grouped or not).  Returns every result tensor."""
    n = len(widths)
    xs = [rnd(M, ci, seed=10 + i, dtype=dtype) for i, (ci, co) in enumerate(widths)]
    ws = [rnd(co, ci, seed=20 + i, scale=ci ** -0.5, dtype=dtype) for i, (ci, co) in enumerate(widths)]
    bs = [rnd(co, seed=30 + i) for i, (ci, co) in enumerate(widths)]
    dys = [rnd(M, co, seed=40 + i, dtype=dtype) for i, (ci, co) in enumerate(widths)]
    dws = [torch.zeros(co, ci, device=DEV) for ci, co in widths]
    dbs = [torch.zeros(co, device=DEV) for ci, co in widths]
    prev = H.defer_slab_reductions(True)
    try:
        with run():
            fwd = [H.linear_fwd(xs[i], ws[i], bs[i], epi=H.EPI_ACT, act="relu") for i in range(n)]
        with run():
            for i in range(n):
                H.linear_bwd_weight(dys[i], xs[i], dws[i], bias_grad=dbs[i])
            dxs = [H.linear_bwd_data(dys[i], ws[i]) for i in range(n)]
        H.flush_grad_reductions()
    finally:
        H.defer_slab_reductions(prev)
        H.drop_pending_reductions()
    torch.cuda.synchronize()
    return fwd + dxs + dws + dbs


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16_storage"])
@pytest.mark.parametrize("M", [4096, 1000, 72])
def test_three_equal_members_match_their_own_launches(dtype, M):
    widths = [(256, 256)] * 3
    with grouping(False):
        want = predictor_triple(dtype, M, widths, H.gemm_group)
        assert H.GROUP_STATS["launches"] == 0
    with grouping(True):
        got = predictor_triple(dtype, M, widths, H.gemm_group)
        # forward: one launch of three; backward section: weight gradients and data gradients, one launch each
        assert H.GROUP_STATS == {"launches": 3, "members": 9, "single": 0}
    for i, (w, g) in enumerate(zip(want, got)):
        assert torch.isfinite(g.float()).all(), i
        assert torch.equal(w, g), (i, (w.float() - g.float()).abs().max().item())
    # and against plain float64 arithmetic (bf16 storage: on the rounded operands)
    x, w, b = rnd(M, 256, seed=10, dtype=dtype).double(), rnd(256, 256, seed=20, scale=1 / 16, dtype=dtype).double(), rnd(256, seed=30).double()
    ref = torch.relu(x @ w.t() + b)
    assert (got[0].double() - ref).abs().max() < 2e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16_storage"])
@pytest.mark.parametrize("tile", [7, 8, 23, 26, 22])
def test_every_grouped_tile_and_unequal_members(dtype, tile):
    if (tile in H.GROUP_TILES[0]) != (dtype == torch.float32):
        pytest.skip("tile of the other core")
    # an encoder layer's weight-gradient shapes (+ forward / data gradient of the same widths), rows not a tile multiple
    widths = [(256, 1024), (1024, 256), (256, 256), (256, 512), (256, 768), (64, 256), (256, 8), (40, 24)]
    with grouping(False):
        want = predictor_triple(dtype, 1160, widths, H.gemm_group)
    with grouping(True, tile):
        got = predictor_triple(dtype, 1160, widths, H.gemm_group)
        assert H.GROUP_STATS["launches"] == 3 and H.GROUP_STATS["members"] == 24 and H.GROUP_STATS["single"] == 0
    for i, (w, g) in enumerate(zip(want, got)):
        assert torch.isfinite(g.float()).all(), i
        assert torch.equal(w, g), (tile, i, (w.float() - g.float()).abs().max().item())


def test_more_than_eight_members_and_members_that_cannot_be_grouped():
    M = 640
    xs = [rnd(M, 64, seed=i) for i in range(10)]
    ws = [rnd(32, 64, seed=50 + i, scale=0.1) for i in range(10)]
    conv_x, conv_w = rnd(M, 64, seed=70), rnd(3, 32, 64, seed=71, scale=0.1)   # a 3-tap convolution: never grouped
    drop = H.Drop(0.5, 1234, None)

    def run_all():
        with H.gemm_group():
            a = [H.linear_fwd(x, w) for x, w in zip(xs, ws)]
            c = H.linear_fwd(conv_x, conv_w, taps=3, T=64)
            d = H.linear_fwd(xs[0], ws[0], epi=H.EPI_ACT, act="relu", drop=drop)   # epilogue dropout: alone
        torch.cuda.synchronize()
        return a + [c, d]

    with grouping(False):
        want = run_all()
    with grouping(True):
        got = run_all()
        assert H.GROUP_STATS == {"launches": 2, "members": 10, "single": 2}   # 8 + 2 grouped, two alone
    for i, (w, g) in enumerate(zip(want, got)):
        assert torch.equal(w, g), i


def test_direct_weight_gradient_inside_a_group_finishes_behind_the_launch():
    """Outside ``FastSpeech2.backward`` (no deferred slab sums) the split-K finish of a weight gradient is the wrapper's own
    second launch: inside a group it must come after the grouped launch, on a slab buffer of its own."""
    M = 8192
    dys = [rnd(M, 256, seed=i) for i in range(3)]
    xs = [rnd(M, 256, seed=10 + i) for i in range(3)]

    def run_all():
        outs = [torch.full((256, 256), float("nan"), device=DEV) for _ in range(3)]
        with H.gemm_group():
            for dy, x, o in zip(dys, xs, outs):
                H.linear_bwd_weight(dy, x, o)
        torch.cuda.synchronize()
        return outs

    assert H.pick_splitk(256, 256, M) > 1
    with grouping(False):
        want = run_all()
    with grouping(True):
        got = run_all()
        assert H.GROUP_STATS["launches"] == 1 and H.GROUP_STATS["single"] == 0
    for dy, x, w, g in zip(dys, xs, want, got):
        assert torch.equal(w, g)
        ref = dy.double().t() @ x.double()
        assert (g.double() - ref).abs().max() < 1e-3 * ref.abs().max()


def test_entry_point_refuses_members_that_cannot_share_a_launch():
    L = H.real_lib()
    s = torch.cuda.current_stream().cuda_stream
    x, w, y = rnd(128, 64), rnd(64, 64), torch.empty(128, 64, device=DEV)

    def args(**kw):
        a = H.GemmArgs()
        a.A, a.B, a.C = x.data_ptr(), w.data_ptr(), y.data_ptr()
        a.Mc, a.Nc, a.R, a.lda, a.ldb, a.ldc = 128, 64, 64, 64, 64, 64
        a.a_kcontig, a.b_kcontig, a.taps, a.alpha, a.res_scale, a.splitk = 1, 1, 1, 1.0, 1.0, 1
        for k, v in kw.items():
            setattr(a, k, v)
        return a

    def call(members, n=None):
        arr = (H.GemmArgs * len(members))(*members)
        return L.fs2hip_gemm_grouped(arr, len(members) if n is None else n, s)

    assert call([args(), args()]) == 0
    assert call([args()], n=0) == -22 and call([args()] * 9) == -22
    assert call([args(), args(b_kcontig=0, ldb=64)]) == -22            # mixed orientations
    assert call([args(), args(drop_p=0.5)]) == -22                      # epilogue dropout
    assert call([args(tile=4), args()]) == -22                          # a tile the grouped kernels do not carry
    assert call([args(), args(operand_bf16=2)]) == -22                  # 32-split operands
    assert L.fs2hip_gemm_grouped(None, 2, s) == -22
    torch.cuda.synchronize()
    assert ctypes.sizeof(H.GemmArgs) == 240  # (the kernel argument of a grouped launch holds eight of them)


def _train(variant, group):
    from tests.test_plan_gpu import assert_same_state, batches, build, run  # noqa: F401
    from fastspeech2_lightning_amd import model as FM
    cfg, prec = {}, "32-true"
    if variant == "learn_alignment":
        cfg = dict(learn_alignment=True)
    elif variant == "frame_level":
        cfg = dict(level="frame")
    elif variant == "gst_multispeaker":
        cfg = dict(gst=True, multispeaker=True, n_mels=80)
    elif variant in ("bf16-mixed", "32-split"):
        prec = variant
    saved = H.GEMM_GROUP
    H.GEMM_GROUP = group
    for k in H.GROUP_STATS:
        H.GROUP_STATS[k] = 0
    try:
        assert FM.PRED_GROUP
        model, opt, config = build(prec, plan=True, **cfg)
        bs = batches(config, 4, learn_alignment=cfg.get("learn_alignment", False), frame_level=cfg.get("level") == "frame")
        rows = run(model, opt, bs)
        return model, rows, dict(H.GROUP_STATS)
    finally:
        H.GEMM_GROUP = saved


@pytest.mark.parametrize("variant", ["plain", "frame_level", "learn_alignment", "gst_multispeaker", "bf16-mixed", "32-split"])
def test_training_steps_with_grouped_launches_equal_steps_without(variant):
    """Four optimizer steps (eager, recorded, two replays of the launch plan), dropout on."""
    from tests.test_plan_gpu import assert_same_state
    plain, want, stats0 = _train(variant, False)
    grouped, got, stats = _train(variant, True)
    assert stats0["launches"] == 0
    if variant != "32-split":   # (the grouped kernels carry exact-fp32 and bf16-storage operands; 32-split members go alone)
        assert stats["launches"] > 0 and stats["members"] >= 2 * stats["launches"], stats
    for i, (w, g) in enumerate(zip(want, got)):
        assert torch.isfinite(g).all() and torch.equal(w, g), (variant, i, w.tolist(), g.tolist())
    assert_same_state(plain, grouped)
    assert grouped.plans.replayed == 2
