"""Host side of the optimizer / scheduler pair (no GPU): they are torch's own interfaces, the Noam values are the
reference's (golden ``noam/lrs`` from the reference's ``NoamLR``), the state dicts have torch's shapes."""
import numpy as np
import torch

from fastspeech2_lightning_amd import params as P
from fastspeech2_lightning_amd.optim import FusedAdamWNoam, NoamLR, noam_scale


def _store():
    S = P.ParamStore()
    S.add("a.weight", (4, 3), "id", P.init_normal)
    S.add("a.bias", (4,), "id", P.init_zeros)
    S.add("conv.weight", (5, 4, 3), "convk", P.init_normal)
    S.finalize("cpu", seed=0)
    return S


def test_interfaces_and_schedule(golden_dir):
    S = _store()
    state = torch.zeros(4, dtype=torch.int64)
    opt = FusedAdamWNoam(S, state, 1e-3, (0.9, 0.98), 1e-9, 0.01, 4000, param_names=list(S.entries))
    assert isinstance(opt, torch.optim.Optimizer) and opt.max_grad_norm is None
    assert opt.param_groups[0]["params"][0].data_ptr() == S.flat.data_ptr()
    sched = NoamLR(opt, 4000)
    assert isinstance(sched, torch.optim.lr_scheduler.LRScheduler)
    # the reference's own NoamLR(AdamW(lr=1e-3), warmup_steps=40).get_last_lr() before each of 120 steps (make_golden.py)
    want = np.load(golden_dir / "units.npz")["noam/lrs"]
    s2 = NoamLR(FusedAdamWNoam(S, state.clone(), 1e-3, (0.9, 0.98), 1e-9, 0.0, 40), 40)
    lrs = []
    for _ in range(len(want)):
        lrs.append(s2.get_last_lr()[0])
        s2.step()
    np.testing.assert_allclose(lrs, want, rtol=1e-12)
    assert sched.get_last_lr()[0] == 1e-3 * noam_scale(0, 4000)
    for k in range(1, 6):
        sched.step()
        assert abs(sched.get_last_lr()[0] - 1e-3 * noam_scale(k, 4000)) < 1e-15
        assert opt.param_groups[0]["lr"] == sched.get_last_lr()[0]


def test_state_dicts_have_torch_shapes_and_round_trip():
    S = _store()
    state = torch.zeros(4, dtype=torch.int64)
    names = list(S.entries)
    opt = FusedAdamWNoam(S, state, 1e-3, (0.9, 0.98), 1e-9, 0.01, 10, param_names=names)
    S.adam_m.copy_(torch.arange(S.total, dtype=torch.float32))
    S.adam_v.copy_(torch.arange(S.total, dtype=torch.float32) * 2)
    opt.set_step(7)
    sd = opt.state_dict()
    assert set(sd) == {"state", "param_groups"} and len(sd["state"]) == 3
    assert sd["state"][2]["exp_avg"].shape == (5, 4, 3) and float(sd["state"][0]["step"]) == 7.0
    assert sd["param_groups"][0]["initial_lr"] == 1e-3 and sd["param_groups"][0]["decoupled_weight_decay"] is True
    # torch's own AdamW accepts it for parameters of those shapes
    ref_params = [torch.nn.Parameter(torch.zeros(S.entries[n].ref_shape)) for n in names]
    ref = torch.optim.AdamW(ref_params, 1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.01)
    ref.load_state_dict(sd)
    assert torch.equal(ref.state[ref_params[2]]["exp_avg"], sd["state"][2]["exp_avg"])
    # and back
    S2 = _store()
    opt2 = FusedAdamWNoam(S2, torch.zeros(4, dtype=torch.int64), 1e-3, (0.9, 0.98), 1e-9, 0.01, 10, param_names=names)
    opt2.load_state_dict(ref.state_dict())
    for n in names:  # (the alignment padding between entries carries no state)
        assert torch.equal(S2.export_flat(S2.adam_m, n), S.export_flat(S.adam_m, n))
        assert torch.equal(S2.export_flat(S2.adam_v, n), S.export_flat(S.adam_v, n))
    assert opt2.steps_done() == 7
    sched = NoamLR(opt2, 10)
    sched.load_state_dict(opt.torch_scheduler_state_dict())
    assert sched.last_epoch == 7 and opt2.steps_done() == 7
    # the native form round-trips too
    S3 = _store()
    opt3 = FusedAdamWNoam(S3, torch.zeros(4, dtype=torch.int64), 1e-3, (0.9, 0.98), 1e-9, 0.01, 10)
    opt3.load_state_dict(opt.native_state_dict())
    assert torch.equal(S3.adam_m, S.adam_m) and opt3.steps_done() == 7
    # zero_grad drops the delivered gradient
    opt.param.grad = torch.zeros_like(opt.param)
    opt.zero_grad()
    assert opt.param.grad is None


def test_in_step_tile_refinement_keeps_only_measured_gains():
    """``hip.refine_tiles_in_step`` (the tuner's second stage, on by default in bench.py): the runner-up tiles of the
    heaviest signatures are tried one at a time, a change is kept only when the (agreed) step time improves by more than
    ``min_gain``, everything else is restored, every change bumps the tile generation (launch plans re-record) and
    ``settle`` runs before each timing.  A fake clock stands in for the GPU."""
    from fastspeech2_lightning_amd import hip as H
    saved = dict(H._TILE_CACHE), dict(H._TILE_TIMINGS), dict(H._TILE_CALLS), H.TILE_GEN[0]
    try:
        H._TILE_CACHE.clear(); H._TILE_TIMINGS.clear(); H._TILE_CALLS.clear()
        a, b, c = ("A",), ("B",), ("C",)
        H._TILE_CACHE.update({a: 13, b: 7, c: 7})
        H._TILE_TIMINGS.update({a: [(4.0, 13), (4.2, 7), (5.0, 8)], b: [(2.0, 7), (2.1, 5)], c: [(0.1, 7), (0.2, 8)]})
        H._TILE_CALLS.update({a: [0], b: [0], c: [0]})
        # the step's true time as a function of the table: signature a is faster IN the step on its runner-up tile 7,
        # b's runner-up is slower, c is never launched in the step
        def true_ms():
            return 18.0 + {13: 0.5, 7: 0.2, 8: 0.6}[H._TILE_CACHE[a]] + {7: 0.0, 5: 0.3}[H._TILE_CACHE[b]]
        calls = {"settle": 0, "count": 0, "agree": 0}

        def count_step():
            calls["count"] += 1
            H._TILE_CALLS[a][0] += 3
            H._TILE_CALLS[b][0] += 8

        def settle():
            calls["settle"] += 1

        def agree(ms):
            calls["agree"] += 1
            return ms + 0.01  # (the maximum over ranks)
        logs = []
        gen0 = H.TILE_GEN[0]
        base, changed = H.refine_tiles_in_step(lambda: None, rounds=4, candidates=2, top=8, min_gain=0.004, log=logs.append,
                                               settle=settle, count_step=count_step, agree=agree,
                                               timer=lambda fn, n: true_ms())
        assert changed == 1 and H._TILE_CACHE == {a: 7, b: 7, c: 7}
        assert abs(base - 18.21) < 1e-9 and len(logs) == 1 and "13 -> 7" in logs[0]
        # timings: the baseline, a's two runner-ups, b's one (c is not launched in the step): each twice; settle before each
        # and once more at the end
        assert calls["count"] == 1 and calls["agree"] == 8 and calls["settle"] == 9
        assert H.TILE_GEN[0] > gen0
    finally:
        H._TILE_CACHE.clear(); H._TILE_CACHE.update(saved[0])
        H._TILE_TIMINGS.clear(); H._TILE_TIMINGS.update(saved[1])
        H._TILE_CALLS.clear(); H._TILE_CALLS.update(saved[2])
        H.TILE_GEN[0] = saved[3]
