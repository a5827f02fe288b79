"""A training step replayed from its recorded launch plan (``fastspeech2_lightning_amd/plan.py``, ``fs2hip_plan_replay``)
enqueues exactly what the eager step enqueues: two models from one seed, one with plans switched off, fed the same
batches -- same geometry, fresh contents every step -- must agree BIT FOR BIT on every loss term of every step, and
after the run on every weight, Adam moment, BatchNorm buffer and counter.  Dropout is on (masks advance through the
device step counter), so a replay that missed a launch, a fill or an event would show at once."""
import pytest
import torch

from fastspeech2_lightning_amd import hip as H
from fastspeech2_lightning_amd import plan as PL
from fastspeech2_lightning_amd.config import Stats
from fastspeech2_lightning_amd.synthetic import synthetic_batch
from oracle import cases as C

pytestmark = pytest.mark.gpu


def build(precision="32-true", plan=True, **cfg):
    from fastspeech2_lightning_amd.model import FastSpeech2
    cfg.setdefault("learn_alignment", False)
    if cfg.get("gst"):  # the style-token layer emits 256 dims: the golden's d = 256, 1-layer configuration, dropout on
        dump = C.build("e2e_gst_multispeaker_train")[0].model_checkpoint_dump()
        for blk in ("encoder", "decoder"):
            dump["model"][blk]["dropout"] = 0.2
        for vp in dump["model"]["variance_predictors"].values():
            vp["dropout"] = 0.2
        from fastspeech2_lightning_amd.config import FastSpeech2Config
        config = FastSpeech2Config(**dump)
    else:
        config = C.small_config(dropout=0.2, **cfg)
    config.training.optimizer.learning_rate = 1e-2
    config.training.optimizer.warmup_steps = 3
    spk = {f"s{i}": i for i in range(4)} if cfg.get("multispeaker") else None
    model = FastSpeech2(config, Stats(**C.STATS), speaker2id=spk, lang2id=spk, seed=5, precision=precision)
    model.plan_enabled = plan
    model.train()
    opt = model.configure_optimizers()[0][0]
    model.configure_gradient_clipping(opt, 1.0, "norm")
    return model, opt, config


def batches(config, n, learn_alignment=False, seed=21, **kw):
    """n batches of ONE geometry with different contents."""
    out = []
    for i in range(n):
        b = synthetic_batch(B=4, ts_lo=6, ts_hi=12, n_symbols=C.N_SYMBOLS, n_mels=config.preprocessing.audio.n_mels,
                            seed=seed, content_seed=100 + i, dur_hi=4, learn_alignment=learn_alignment, **kw)
        b["speaker_id"] = torch.arange(4, dtype=torch.int32) % 4
        b["language_id"] = torch.arange(4, dtype=torch.int32) % 4
        b["basename"] = [f"utt{i}_{j}" for j in range(4)]
        out.append(b)
    return out


def run(model, opt, bs):
    rows = []
    for b in bs:
        with torch.no_grad():
            model.training_step(b)
        rows.append(model._loss_slots.clone())
        opt.step()
    torch.cuda.synchronize()
    return rows


def assert_same_state(a, b):
    for name in ("flat", "adam_m", "adam_v", "grad", "bn_counters"):
        assert torch.equal(getattr(a.store, name), getattr(b.store, name)), name
    for k in a.store.buffers:
        assert torch.equal(a.store.buffers[k], b.store.buffers[k]), k
    assert torch.equal(a.step_state, b.step_state)


@pytest.mark.parametrize("variant", ["plain", "learn_alignment", "gst_multispeaker", "frame_level", "bf16-mixed", "32-split"])
def test_replayed_steps_equal_eager_steps_bit_for_bit(variant):
    cfg, prec = {}, "32-true"
    if variant == "learn_alignment":
        cfg = dict(learn_alignment=True)
    elif variant == "gst_multispeaker":
        cfg = dict(gst=True, multispeaker=True, n_mels=80)
    elif variant == "frame_level":
        cfg = dict(level="frame")
    elif variant in ("bf16-mixed", "32-split"):
        prec = variant
    eager, opt_e, config = build(prec, plan=False, **cfg)
    planned, opt_p, _ = build(prec, plan=True, **cfg)
    bs = batches(config, 6, learn_alignment=cfg.get("learn_alignment", False), frame_level=cfg.get("level") == "frame")
    want = run(eager, opt_e, bs)
    got = run(planned, opt_p, bs)
    assert planned.plans.recorded == 1 and planned.plans.replayed == 4 and planned.plans.eager == 1
    assert eager.plans.recorded == 0
    for i, (w, g) in enumerate(zip(want, got)):
        assert torch.isfinite(w).all() and torch.equal(w, g), (variant, i, w.tolist(), g.tolist())
    assert_same_state(eager, planned)
    for k in ("output", "postnet_output", "duration_prediction", "pitch_prediction", "energy_prediction", "tgt_mask"):
        assert torch.equal(eager.last_output[k], planned.last_output[k]), k
    plan = next(iter(planned.plans.plans.values()))
    assert plan.launches > 100 and plan.n_events > 4


def test_two_geometries_alternate_between_their_plans_and_an_unseen_one_runs_eagerly():
    eager, opt_e, config = build(plan=False)
    planned, opt_p, _ = build(plan=True)
    a = batches(config, 4, seed=21)
    b = batches(config, 4, seed=22)
    c = batches(config, 1, seed=23)
    assert a[0]["mel"].shape != b[0]["mel"].shape
    order = [a[0], b[0], a[1], b[1], a[2], c[0], b[2], a[3], b[3]]
    want = run(eager, opt_e, order)
    got = run(planned, opt_p, order)
    assert planned.plans.recorded == 2 and planned.plans.replayed == 4 and planned.plans.eager == 3
    for i, (w, g) in enumerate(zip(want, got)):
        assert torch.equal(w, g), i
    assert_same_state(eager, planned)


def test_recorded_outputs_are_rewritten_in_place_and_host_batches_are_fed():
    """The tensors a replay returns are the recorded step's (as with a captured graph); the batch may arrive as host
    tensors or as fresh device tensors every step."""
    planned, opt, config = build(plan=True)
    bs = batches(config, 4)
    with torch.no_grad():
        planned.training_step(bs[0]); opt.step()
        planned.training_step(bs[1]); opt.step()          # recorded
        out1 = planned.last_output["postnet_output"]
        keep = out1.clone()
        planned.training_step(planned.prepare_batch(bs[2])); opt.step()   # replay, device tensors that are not the recorded ones
        out2 = planned.last_output["postnet_output"]
        assert out2 is out1 and not torch.equal(out2, keep)
        planned.training_step(bs[3]); opt.step()          # replay, host tensors
    assert planned.plans.replayed == 2
    torch.cuda.synchronize()


def test_an_aten_kernel_inside_the_step_fails_the_recording_loudly(monkeypatch):
    """A kernel the recorder cannot see would silently be missing from every replay: recording refuses it."""
    planned, opt, config = build(plan=True)
    bs = batches(config, 2)
    real = H.mask_from_lens

    def leaky(lens, T):
        m = real(lens, T)
        return m | (lens.view(-1, 1) > 10 ** 6)   # an ATen kernel on GPU tensors
    with torch.no_grad():
        planned.training_step(bs[0])
        monkeypatch.setattr(H, "mask_from_lens", leaky)
        with pytest.raises(PL.PlanError, match="ATen"):
            planned.training_step(bs[1])
    assert H._REC is None


def test_plans_follow_the_switches_in_their_signature(monkeypatch):
    planned, opt, config = build(plan=True)
    bs = batches(config, 6)
    with torch.no_grad():
        for b in bs[:3]:
            planned.training_step(b); opt.step()
        assert planned.plans.replayed == 1
        planned.env.side_enabled = False               # another launch sequence: seen 0 times -> eager, then recorded
        for b in bs[3:]:
            planned.training_step(b); opt.step()
    assert planned.plans.recorded == 2 and planned.plans.replayed == 2
    monkeypatch.setattr(PL, "ENABLED", False)
    with torch.no_grad():
        planned.training_step(bs[0])
    assert planned.plans.replayed == 2
    torch.cuda.synchronize()


def test_held_weight_gradients_change_nothing_but_the_schedule(monkeypatch):
    """``FS2_HOLD_WGRADS=1`` moves the PostNet's / decoder's weight-gradient GEMMs into the encoder's backward window
    (``hip.hold_weight_gradients``): the same launches with the same operands at another time -- every gradient, loss and
    weight must come out bit for bit as without it, eagerly and from a recorded plan."""
    from fastspeech2_lightning_amd import model as MM
    ref, opt_r, config = build(plan=False)
    bs = batches(config, 5)
    want = run(ref, opt_r, bs)
    monkeypatch.setattr(MM, "HOLD_WGRADS", True)
    held, opt_h, _ = build(plan=True)
    got = run(held, opt_h, bs)
    assert held.plans.replayed == 3 and H._HELD_WGRADS is None
    for i, (w, g) in enumerate(zip(want, got)):
        assert torch.equal(w, g), i
    assert_same_state(ref, held)


def test_replays_survive_validation_checkpoint_reload_and_a_grad_enabled_loop():
    """What a real training loop does between two steps of one geometry: a validation pass (evaluation forward, its own
    allocations), a checkpoint save + ``load_state_dict`` (weights rewritten in place), Lightning's grad-enabled call with
    ``loss.backward()`` -- the replayed steps must still equal an eager model's bit for bit."""
    eager, opt_e, config = build(plan=False)
    planned, opt_p, _ = build(plan=True)
    bs = batches(config, 6)
    val = batches(config, 1, seed=31)[0]
    rows = {id(eager): [], id(planned): []}
    for model, opt in ((eager, opt_e), (planned, opt_p)):
        for i, b in enumerate(bs):
            if i % 2 == 0:   # the native loop
                with torch.no_grad():
                    model.training_step(b)
            else:            # Lightning's automatic optimization: closure -> zero_grad -> backward -> step
                loss = model.training_step(b)
                opt.zero_grad()
                loss.backward()
            rows[id(model)].append(model._loss_slots.clone())
            opt.step()
            if i == 2:
                v = model.validation_step(val)
                assert torch.isfinite(v["total"])
            if i == 3:
                sd = {k: t.clone() for k, t in model.state_dict().items()}
                model.load_state_dict(sd)
    torch.cuda.synchronize()
    assert planned.plans.replayed == 4
    for i, (w, g) in enumerate(zip(rows[id(eager)], rows[id(planned)])):
        assert torch.equal(w, g), i
    assert_same_state(eager, planned)


def test_logged_loss_tensors_of_a_replayed_step_keep_their_values():
    """Lightning keeps the tensors ``log_dict`` received and reads them later: a replayed step hands out a copy of the
    recorded slot vector, so step n's logged losses are not rewritten by step n + 1 (the big outputs are, by design)."""
    planned, opt, config = build(plan=True)
    bs = batches(config, 5)
    kept = []
    with torch.no_grad():
        for b in bs:
            total = planned.training_step(b)
            kept.append((total, float(total), dict(planned.last_losses), {k: float(v) for k, v in planned.last_losses.items()}))
            opt.step()
    torch.cuda.synchronize()
    assert planned.plans.replayed == 3
    for total, value, losses, values in kept:
        assert float(total) == value
        for k, v in losses.items():
            assert float(v) == values[k], k
    assert len({v for _, v, _, _ in kept}) == len(kept)   # (the steps really differ: fresh contents, optimizer steps)
