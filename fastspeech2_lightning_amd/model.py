"""``FastSpeech2`` -- the reference's module surface (reference ``fs2/model.py:38-549``) over the
fs2hip kernels.

Same constructor, same batch dict in (``collate_method`` output, ``fs2/dataset.py:257-293``), same
16-key dict out (``fs2/model.py:251-268``), same hooks (``training_step``, ``validation_step``,
``predict_step``, ``configure_optimizers``, ``on_save_checkpoint`` / ``on_load_checkpoint``) and the
same state-dict keys, so a reference checkpoint loads and a checkpoint written here loads in the
reference.  Differences, all forced by the design (MI355X-first):

* no autograd tape: ``training_step`` runs forward, loss, backward and leaves the gradients in the
  flat gradient buffer (``model.store.grad``); the optimizer is the fused AdamW of ``optim.py``;
* no host synchronisation inside a step (the reference has a Python loop + ``.item()`` calls);
* the GPU is mandatory: every op raises if the fs2hip library is missing or a tensor is on the CPU.
"""
from __future__ import annotations

import sys
from collections import OrderedDict
from typing import Optional

import torch

from . import hip as H
from . import modules as M
from . import params as P
from . import plan as PL
from .config import (BadDataError, FastSpeech2Config, InferenceControl, N_PHONOLOGICAL_FEATURES, Stats,
                     TargetTrainingTextRepresentationLevel, TextProcessor)

try:  # keep the Lightning surface when Lightning is present (it is not in the build image)
    import pytorch_lightning as pl  # type: ignore

    _Base = pl.LightningModule
except Exception:  # pragma: no cover - depends on the environment
    _Base = torch.nn.Module

LOSS_KEYS = ("pitch", "energy", "duration", "spec", "postnet", "attn_ctc", "attn_bin")
#: "32-true": keep fp32 W^T mirrors of the weights whose data gradient reduces over the model width (one transpose launch
#: per step), so that those GEMMs run in the forward orientation on the streaming kernel.  FS2_FP32_TRANSPOSED=0: off.
import os as _os
FP32_TRANSPOSED = _os.environ.get("FS2_FP32_TRANSPOSED", "1") != "0"
#: FS2_HOLD_WGRADS=1: the PostNet's, mel head's and decoder's weight-gradient GEMMs are held back and released, a few per
#: encoder sub-module, during the encoder's backward pass (single GPU only: under data parallelism a bucket's weight
#: gradients must be enqueued before its hand-off).  See ``hip.hold_weight_gradients``.
HOLD_WGRADS = _os.environ.get("FS2_HOLD_WGRADS", "0") != "0"
#: FS2_EARLY_FLUSH (single GPU): the second-stage sums pending when the encoder's backward pass starts -- every split-K
#: slab set and bias / LayerNorm partial of the PostNet, the decoder and the variance adaptor, ~80 % of the 1.2 GB the
#: final flush reads -- are finished on the side stream beside the encoder's small kernels instead of on the main stream
#: after the last backward kernel, where nothing can hide them (0.2 ms).  Data parallel runs flush per bucket anyway.
EARLY_FLUSH = _os.environ.get("FS2_EARLY_FLUSH", "1") != "0"


#: FS2_PRED_LANES=1 (measurement aid, off by default): the three variance predictors -- independent chains of ~25 small
#: launches each, a training step feeds them targets (fs2/variance_adaptor.py:309-352) -- each get a side stream of their
#: own instead of sharing lane 0 with the weight-gradient GEMMs.  Measured round 5 (same box, alternating, launch plans
#: on): bf16-mixed batch 64 11.13-11.20 ms with lanes against 10.99-11.00 without, fp32 18.53-18.55 against 18.58-18.67:
#: three more chains of small kernels take CU slots from the main chain as often as they fill idle ones.
PRED_LANES = ({"energy": 1, "pitch": 2, "duration": 3} if _os.environ.get("FS2_PRED_LANES", "0") != "0"
              else {"energy": 0, "pitch": 0, "duration": 0})


#: FS2_PRED_GROUP=0 (measurement aid): the variance predictors of a training step run one after the other on the side
#: stream, as in rounds 1-4.  Default: predictors that see the same number of rows walk their layers in lockstep and their
#: pointwise GEMMs -- forward, data gradient, weight gradient -- go out as one grouped launch per layer
#: (``modules.predictors_fwd``, ``fs2hip_gemm_grouped``); the three chains are independent in training because each
#: embedding is looked up from the TARGET (fs2/variance_adaptor.py:309-352).
PRED_GROUP = _os.environ.get("FS2_PRED_GROUP", "1") != "0"


class VarianceAdaptor:
    """reference ``fs2/variance_adaptor.py:84-412`` (order of operations ``:309-397``)."""

    def __init__(self, S: P.ParamStore, env: M.Env, config: FastSpeech2Config, stats: Stats):
        self.S, self.env, self.config = S, env, config
        self.bad_count = None  # set by the model: persistent device word counting duration / mel_lens mismatches
        self.bad_probe = None  # set by the model: called right behind the kernel that bumps bad_count (training steps)
        vp, d = config.model.variance_predictors, config.model.encoder.input_dim
        pre = "variance_adaptor."
        # declaration order = execution order: energy, pitch, duration
        self.energy_predictor = M.VariancePredictor(S, env, pre + "energy_predictor.", d, vp.energy)
        S.add(pre + "energy_embedding.weight", (vp.energy.n_bins, vp.energy.input_dim), "id", P.init_normal)
        self.pitch_predictor = M.VariancePredictor(S, env, pre + "pitch_predictor.", d, vp.pitch)
        S.add(pre + "pitch_embedding.weight", (vp.pitch.n_bins, vp.pitch.input_dim), "id", P.init_normal)
        self.duration_predictor = M.VariancePredictor(S, env, pre + "duration_predictor.", d, vp.duration)
        self.aligner = None
        if config.model.learn_alignment:
            self.aligner = M.Aligner(S, env, pre + "attention.", config.preprocessing.audio.n_mels, d)
        S.add_buffer(pre + "pitch_bins", torch.linspace(stats.pitch.norm_min, stats.pitch.norm_max, vp.pitch.n_bins - 1))
        S.add_buffer(pre + "energy_bins", torch.linspace(stats.energy.norm_min, stats.energy.norm_max, vp.energy.n_bins - 1))
        self.pre = pre

    def _grouping(self) -> bool:
        return PRED_GROUP and not any(PRED_LANES.values())

    def _run_predictors(self, jobs, res):
        """``jobs``: [(name, input, lens)] of a training forward, all inputs made; ``res`` receives name -> (prediction,
        context).  Predictors with equal ``lockstep_key`` run as one lockstep chain on the side stream; the groups are
        returned (the backward pass walks the same groups)."""
        groups = {}
        for name, x, lens in jobs:
            groups.setdefault(getattr(self, f"{name}_predictor").lockstep_key(x), []).append((name, x, lens))
        out = []
        for members in groups.values():
            names = [m[0] for m in members]
            preds = [getattr(self, f"{n}_predictor") for n in names]
            with self.env.side(*[t for m in members for t in m[1:]]):
                ps, ctxs = M.predictors_fwd(preds, [m[1] for m in members], [m[2] for m in members])
            for n, pr, cx in zip(names, ps, ctxs):
                res[n] = (pr, cx)
            out.append(names)
        return out

    def _variance(self, name, x, target, lens, control, inference, jobs=None):
        S, pre = self.S, self.pre
        predictor = getattr(self, f"{name}_predictor")
        if inference:
            pred, pctx = predictor.fwd(x, lens)
        elif jobs is not None:  # training, grouped: run with the other predictors of this level (``_run_predictors``)
            jobs.append((name, x, lens))
            pred = pctx = None
        else:  # training: the prediction feeds only the loss (the embedding uses the target) -> side stream
            with self.env.side(x, lens, lane=PRED_LANES[name]):
                pred, pctx = predictor.fwd(x, lens)
        if inference:
            out, idx = H.bucket_embed_add(pred, S.b(pre + f"{name}_bins"), S.p(pre + f"{name}_embedding.weight"), x, control)
            if control != 1.0:
                pred = H.axpby(pred, None, control, 0.0)
        else:
            out, idx = H.bucket_embed_add(target, S.b(pre + f"{name}_bins"), S.p(pre + f"{name}_embedding.weight"), x)
        return pred, out, (pctx, idx)

    def fwd(self, x, batch, src_lens, table, Tm, control, inference, teacher_forcing, text_emb=None):
        cfg = self.config.model.variance_predictors
        B, Ts, D = x.shape
        c = {}
        energy_t = None if inference else batch["energy"]
        pitch_t = None if inference else batch["pitch"]
        energy_p = pitch_p = None
        attn = dict(attn_logprob=None, attn_soft=None, attn_hard=None)
        dur_aligned = None
        if self.aligner is not None and (teacher_forcing or not inference):
            # fs2/variance_adaptor.py:249-305: soft alignment, MAS, durations, phone-level targets
            mel, mel_lens = batch["mel"], batch["mel_lens"]
            Tm_ = mel.shape[1]
            logprob, soft, hard, hard_idx, dur_aligned, c["align"] = self.aligner.fwd(
                mel, text_emb, batch["duration"], src_lens, mel_lens)
            attn = dict(attn_logprob=logprob.view(B, 1, Tm_, Ts), attn_soft=soft.view(B, 1, Tm_, Ts),
                        attn_hard=hard.view(B, 1, Tm_, Ts))
            c["hard_idx"] = hard_idx
            for name, tgt in (("energy", energy_t), ("pitch", pitch_t)):
                if tgt is not None and tgt.shape[1] == Ts and Ts != Tm_:
                    raise ValueError("pitch/energy targets are already phone-averaged: learned alignment needs "
                                     "frame-level targets (fs2/variance_adaptor.py:269-279)")
            # fs2/variance_adaptor.py:289-305: the aligner's durations must add up to mel_lens; the per-utterance
            # flags stay on the device (FastSpeech2.check_bad_data reads them, no sync in a training step)
            cum_a, _, c["bad"] = H.duration_cumsum(dur_aligned, Tm_, expect=mel_lens, bad_count=self.bad_count)
            if self.bad_probe is not None:
                H.plan_callback(self.bad_probe)  # (host side of the in-step check: FastSpeech2._bad_probe)
            if energy_t is not None and cfg.energy.level.value == "phone":
                energy_t = H.avg_variance(energy_t, cum_a)
            if pitch_t is not None and cfg.pitch.level.value == "phone":
                pitch_t = H.avg_variance(pitch_t, cum_a)
        # training: the predictors of one level are collected and run as lockstep chains (grouped GEMM launches)
        jobs = [] if (not inference and self._grouping()) else None
        res, c["pred_groups"] = {}, []
        if cfg.energy.level.value == "phone":
            energy_p, x, c["energy"] = self._variance("energy", x, energy_t, src_lens, control.energy, inference, jobs)
        if cfg.pitch.level.value == "phone":
            pitch_p, x, c["pitch"] = self._variance("pitch", x, pitch_t, src_lens, control.pitch, inference, jobs)
        if jobs is not None:
            jobs.append(("duration", x, src_lens))
            c["pred_groups"] += self._run_predictors(jobs, res)
            logd, c["duration"] = res["duration"]
        elif dur_aligned is not None or teacher_forcing or not inference:
            # durations come from the batch / the aligner: prediction feeds the loss only
            with self.env.side(x, src_lens, lane=PRED_LANES["duration"]):
                logd, c["duration"] = self.duration_predictor.fwd(x, src_lens)
        else:
            logd, c["duration"] = self.duration_predictor.fwd(x, src_lens)
        if dur_aligned is not None:
            dur = dur_aligned
        elif teacher_forcing or not inference:
            dur = batch["duration"]
        else:
            # fs2/variance_adaptor.py:360-366: clamp(round(exp(logd) - 1) * control, min=0).int()
            dur = H.duration_round(logd, control.duration)
            _, totals = H.duration_cumsum(dur, 1 << 30)
            Tm = int(min(int(totals.max()), int(Tm)))  # host sync (output size): inference only
            Tm = max(Tm, 1)
        frame_level = cfg.energy.level.value == "frame" or cfg.pitch.level.value == "frame"
        x, cum, tgt_lens = H.length_regulate_fwd(x, dur, Tm, None if frame_level else table(Tm))
        c["cum"] = cum
        jobs = [] if jobs is not None else None
        if cfg.energy.level.value == "frame":
            energy_p, x, c["energy"] = self._variance("energy", x, energy_t, tgt_lens, control.energy, inference, jobs)
        if cfg.pitch.level.value == "frame":
            pitch_p, x, c["pitch"] = self._variance("pitch", x, pitch_t, tgt_lens, control.pitch, inference, jobs)
        if jobs:
            c["pred_groups"] += self._run_predictors(jobs, res)
        for name in ("energy", "pitch"):  # grouped predictors: prediction and context arrive here
            if name in res:
                c[name] = (res[name][1], c[name][1])
        energy_p = res["energy"][0] if "energy" in res else energy_p
        pitch_p = res["pitch"][0] if "pitch" in res else pitch_p
        if frame_level:
            x = H.add_posenc(x, table(Tm), tgt_lens, B, Tm)
        return dict(output=x, duration_prediction=logd, duration_target=dur if (teacher_forcing or not inference) else None,
                    duration_rounded=dur, pitch_prediction=pitch_p, pitch_target=pitch_t, energy_prediction=energy_p,
                    energy_target=energy_t, tgt_lens=tgt_lens, Tm=Tm, **attn), c

    def bwd_predictors_early(self, dpred, c):
        """The three predictors' backward chains depend only on the loss gradients and saved activations: they are
        enqueued on the side stream at the very start of the backward pass (small, latency-bound kernels that then
        run under the PostNet / decoder GEMMs); ``bwd`` joins and adds their input gradients where they belong."""
        env = self.env
        if not env.side_enabled and not (self._grouping() and c.get("pred_groups")):
            return
        out = {}
        grouped = set()
        for names in c.get("pred_groups", []):  # the lockstep chains of the forward pass, walked back the same way
            names = [n for n in names if n in c and dpred.get(n) is not None]
            if len(names) < 2:
                continue
            ctxs = [c[n][0] if n != "duration" else c[n] for n in names]
            with env.side(*[dpred[n] for n in names]):
                ds = M.predictors_bwd([getattr(self, f"{n}_predictor") for n in names], [dpred[n] for n in names], ctxs)
            out.update(zip(names, ds))
            grouped.update(names)
        for name in ("duration", "pitch", "energy"):
            if name in grouped or name not in c or dpred.get(name) is None:
                continue
            ctx = c[name][0] if name != "duration" else c[name]
            with env.side(dpred[name], lane=PRED_LANES[name]):
                out[name] = getattr(self, f"{name}_predictor").bwd(dpred[name], ctx)
        if "align" in c:  # the aligner's backward only meets the main chain at the text embedding's gradient
            with env.side(dpred.get("attn_ctc"), dpred.get("attn_bin")):
                out["align"] = self.aligner.bwd(dpred.get("attn_ctc"), dpred.get("attn_bin"), c["align"])
        c["early_bwd"] = out

    def bwd(self, d_dec_in, dpred, c):
        """d_dec_in: gradient of the decoder input; dpred: {'pitch','energy','duration'} loss gradients."""
        S, pre = self.S, self.pre
        cfg = self.config.model.variance_predictors

        early = c.get("early_bwd", {})  # predictor backward chains already computed on the side stream
        if early:
            self.env.join()

        def predictor_bwd(name):
            if name in early:
                return early[name]
            ctx = c[name][0] if name != "duration" else c[name]
            return getattr(self, f"{name}_predictor").bwd(dpred[name], ctx)

        def variance_bwd(name, d):
            pctx, idx = c[name]
            H.embedding_bwd(idx.reshape(-1), d, S.g(pre + f"{name}_embedding.weight"))
            return H.axpby(d, predictor_bwd(name))

        d = d_dec_in
        if cfg.pitch.level.value == "frame":
            d = variance_bwd("pitch", d)
        if cfg.energy.level.value == "frame":
            d = variance_bwd("energy", d)
        d = H.length_regulate_bwd(d, c["cum"])
        d = H.axpby(d, predictor_bwd("duration"))
        if cfg.pitch.level.value == "phone":
            d = variance_bwd("pitch", d)
        if cfg.energy.level.value == "phone":
            d = variance_bwd("energy", d)
        d_text = None
        if "align" in c:
            d_text = early["align"] if "align" in early else self.aligner.bwd(dpred.get("attn_ctc"), dpred.get("attn_bin"),
                                                                              c["align"])
        return d, d_text


class FastSpeech2Loss:
    """reference ``fs2/loss.py:19-126``: ``model.loss(output, batch, current_epoch)``."""

    def __init__(self, model: "FastSpeech2"):
        self.model = model

    def __call__(self, output, batch, current_epoch=0, frozen_components=None):
        with torch.cuda.device(self.model.device_):
            return self._loss(output, batch, current_epoch)

    def _loss(self, output, batch, current_epoch=0):
        m = self.model
        m.env.join()  # variance predictors run on the side stream during a training forward
        cfg, t = m.config.model, m.config.training
        dev = m.device_
        slots = H.zeros(len(LOSS_KEYS) + 1, device=dev)
        want = m.training
        B = output["src_lens"].numel()
        Ts, Tm = output["src_mask"].shape[1], output["tgt_mask"].shape[1]
        src_lens, tgt_lens = output["src_lens"], output["tgt_lens"]
        grads, losses = {}, OrderedDict()

        def term(key, pred, target, lens, T, C, kind, weight):
            i = LOSS_KEYS.index(key)
            grads[key] = H.masked_loss(pred, target, lens, B, T, C, kind=kind, weight=weight,
                                       loss_out=slots[i:i + 1], want_grad=want)
            losses[key] = slots[i]

        for name, w in (("pitch", t.pitch_loss_weight), ("energy", t.energy_loss_weight)):
            if output[f"{name}_target"] is None:
                continue
            c = getattr(cfg.variance_predictors, name)
            phone = c.level.value == "phone"
            term(name, output[f"{name}_prediction"], output[f"{name}_target"], src_lens if phone else tgt_lens,
                 Ts if phone else Tm, 1, c.loss.value, w)
        term("duration", output["duration_prediction"], output["duration_target"], src_lens, Ts, 1,
             cfg.variance_predictors.duration.loss.value, t.duration_loss_weight)
        mel = batch["mel"]
        n_mels = mel.shape[-1]
        term("spec", output["output"], mel, tgt_lens, Tm, n_mels, cfg.mel_loss.value, t.mel_loss_weight)
        if cfg.use_postnet:
            term("postnet", output["postnet_output"], mel, tgt_lens, Tm, n_mels, cfg.mel_loss.value, t.postnet_loss_weight)
        if cfg.learn_alignment and output.get("attn_logprob") is not None:
            # fs2/loss.py:109-122
            lp, soft = output["attn_logprob"], output["attn_soft"]
            Bm, _, Tmm, Tss = lp.shape
            i = LOSS_KEYS.index("attn_ctc")
            early = getattr(m, "_early_ctc", None)
            if early is not None and want:  # computed on the side stream during the forward pass (model.forward)
                H.axpby(early[0], None, 1.0, 0.0, out=slots[i:i + 1])
                grads["attn_ctc"] = early[1]
                m._early_ctc = None
            else:
                grads["attn_ctc"] = H.attn_ctc_loss(lp.view(Bm, Tmm, Tss), batch["src_lens"], batch["mel_lens"],
                                                    t.attn_ctc_loss_weight, slots[i:i + 1], want_grad=want)
            losses["attn_ctc"] = slots[i]
            w = min(current_epoch / t.attn_bin_loss_warmup_epochs, 1.0) * t.attn_bin_loss_weight
            i = LOSS_KEYS.index("attn_bin")
            grads["attn_bin"] = H.attn_bin_loss(soft.view(Bm, Tmm, Tss), m._hard_idx, float(w), slots[i:i + 1])
            losses["attn_bin"] = slots[i]
        H.sum_slots(slots, len(LOSS_KEYS), slots[len(LOSS_KEYS):])
        losses["total"] = slots[len(LOSS_KEYS)]
        m._loss_grads = grads if want else None
        m._loss_slots = slots
        return losses


class _DeliverGrad(torch.autograd.Function):
    """What makes ``training_step``'s result a loss autograd can differentiate (fs2/model.py:384-390 returns the
    loss and Lightning calls ``loss.backward()``).  The backward pass itself has already run inside ``training_step``
    -- explicit launch sequences on two HIP streams, no tape -- and left d(total)/d(parameters) in the flat gradient
    buffer; this node's backward only hands that buffer to autograd as the flat parameter's gradient (an alias, no
    copy) after multiplying it by the upstream gradient (a device scalar: exactly 1 for ``loss.backward()``, 1/N when
    the caller divides the loss for gradient accumulation; the pass is skipped for 1)."""

    @staticmethod
    def forward(ctx, flat_param, total, model):
        ctx.model = model
        return total.view_as(total)

    @staticmethod
    def backward(ctx, grad_out):
        m = ctx.model
        g = m.store.grad
        with torch.cuda.device(g.device):
            H.scale_dev(g, grad_out.to(torch.float32).reshape(1).contiguous())
        p = m.flat_param
        if p.grad is not None and p.grad.untyped_storage().data_ptr() == g.untyped_storage().data_ptr():
            # gradient accumulation without zero_grad(): the previous delivery aliases the buffer the backward pass has
            # just overwritten -- impossible to add to.  training_step() moves it aside before it runs (see there).
            raise RuntimeError("FastSpeech2: the previous step's gradient still aliases the flat gradient buffer")
        opt = getattr(m, "optimizer", None)
        if opt is not None:
            opt.steps_since_delivery = 0
        return g.view(g.shape), None, None


class FastSpeech2(_Base):
    _VERSION: str = "1.2"

    def __init__(self, config, stats=None, lang2id: Optional[dict] = None, speaker2id: Optional[dict] = None,
                 device: Optional[str] = None, seed: int = 1234, precision: Optional[str] = None):
        super().__init__()
        # Lightning's Trainer(precision=...) for this path: "32-true" (the parity path) or "bf16-mixed" (GEMM operands
        # rounded to bf16 for the bf16 MFMA; parameters, activations, accumulation and the optimizer stay fp32).
        # An explicit constructor value (or FS2_PRECISION) outranks a Trainer left at its default -- see
        # ``_adopt_trainer_precision``; "32-split" can only be chosen this way (Trainer rejects the string).
        if precision is None:
            precision = _os.environ.get("FS2_PRECISION") or None
        self._precision_explicit = precision is not None
        H.set_precision(precision or "32-true")
        self.precision = H.get_precision()
        if not isinstance(config, FastSpeech2Config):
            from pydantic import ValidationError
            try:
                config = FastSpeech2Config(**config)
            except ValidationError as e:
                raise TypeError("Unable to load config.  Possible causes: is it really a FastSpeech2Config? "
                                "or the correct version?") from e
        if stats is not None and not isinstance(stats, Stats):
            stats = Stats(**stats)
        H.lib()  # fail loudly when the HIP library has not been built
        if not torch.cuda.is_available():
            raise RuntimeError("FastSpeech2 (MI355X build) needs a GPU: there is no CPU path")
        self.device_ = torch.device(device or f"cuda:{torch.cuda.current_device()}")
        self.config, self.stats = config, stats
        self.batch_size = config.training.batch_size
        self.text_processor = TextProcessor(config.text)
        self.lang2id, self.speaker2id = lang2id or {}, speaker2id or {}
        self.current_epoch_ = 0
        m = config.model
        d = m.encoder.input_dim
        self.step_state = H.new_step_state(self.device_)
        self._seed = int(seed)
        self.env = M.Env(self.step_state, seed)
        S = self.store = P.ParamStore()
        self.padding_idx = self.text_processor.encode_text(self.text_processor._pad_symbol)[0]
        self.use_pfs = m.target_text_representation_level == TargetTrainingTextRepresentationLevel.phonological_features
        if self.use_pfs:  # fs2/model.py:72-81: nn.Linear(N_PHONOLOGICAL_FEATURES, d, bias=False) on batch["pfs"]
            S.add("text_input_layer.weight", (d, N_PHONOLOGICAL_FEATURES), "padk4", P.init_linear_weight)
        else:
            S.add("text_input_layer.weight", (len(self.text_processor.symbols), d), "id", self._init_text_embedding)
        S.add_buffer("position_embedding.inv_freq", 1 / (10000 ** (torch.arange(0.0, d, 2.0) / d)))
        self.encoder = M.Conformer(S, self.env, "encoder.", m.encoder)
        S.next_bucket()
        self._bucket_va = S._bucket
        self.gst = None
        if m.use_global_style_token_module:
            self.gst = M.StyleEncoder(S, self.env, "gst.", config.preprocessing.audio.n_mels)
        if m.multispeaker:
            if len(self.speaker2id) == 0:
                print("Your model is multispeaker but speaker2id LookupTable is empty", file=sys.stderr)
                sys.exit(1)
            S.add("speaker_embedding.weight", (len(self.speaker2id), d), "id", P.init_normal)
        if m.multilingual:
            if len(self.lang2id) == 0:
                print("Your model is multilingual but language2id LookupTable is empty", file=sys.stderr)
                sys.exit(1)
            S.add("language_embedding.weight", (len(self.lang2id), d), "id", P.init_normal)
        if stats is None:
            print("Your model doesn't have a value for self.stats: the variance adaptor cannot be built", file=sys.stderr)
            self.variance_adaptor = None
        else:
            self.variance_adaptor = VarianceAdaptor(S, self.env, config, stats)
        S.next_bucket()
        self.decoder = M.Conformer(S, self.env, "decoder.", m.decoder)
        S.next_bucket()
        self._bucket_head = S._bucket
        n_mels = config.preprocessing.audio.n_mels
        M.decl_linear(S, "mel_linear.", n_mels, m.decoder.input_dim)
        if m.use_postnet:
            self.postnet = M.PostNet(S, self.env, "postnet.", n_mels)
            self.output_key = "postnet_output"
        else:
            self.postnet = None
            self.output_key = "output"
        S.finalize(self.device_, seed)
        # the one nn.Parameter autograd / torch.optim / Lightning see: an alias of the flat buffer (its .grad, once
        # loss.backward() has run, an alias of the flat gradient buffer)
        self.flat_param = torch.nn.Parameter(S.flat)
        self.bad_count = torch.zeros(1, device=self.device_, dtype=torch.int32)
        self._bad_seen, self._pending_bad, self._in_step = 0, [], False
        self._trainer_precision_seen = None
        self.callback_metrics = {}   # (without Lightning: what ``trainer.callback_metrics`` would hold; see ``log_dict``)
        self._val_acc = {}
        if self.variance_adaptor is not None:
            self.variance_adaptor.bad_count = self.bad_count
            self.variance_adaptor.bad_probe = self._bad_probe
        self._bad_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._bad_event, self._bad_probed = None, False
        self._reorder_state_dict_keys()
        self.loss = FastSpeech2Loss(self)
        self._tables = {}
        self._ctx = self._loss_grads = self._loss_slots = self._hard_idx = None
        self.grad_sync = None  # set by parallel.GradSync for data-parallel training
        self.plans = PL.PlanCache()  # recorded launch plans of training steps, per batch geometry (plan.py)
        self.plan_enabled = True     # False: every step of THIS model eager (FS2_PLAN=0: of every model)
        self.training = False
        self.env.training = False

    # ---- helpers ------------------------------------------------------------------------------------
    def data_parallel(self, sync, rank: int):
        """Joins a data-parallel job: ``sync`` (``parallel.GradSync``) exchanges the gradient buckets during the
        backward pass, and the rank enters the DROPOUT seed (not the initialisation seed: every rank builds the same
        weights, rank 0's are broadcast anyway) so that ranks draw independent masks, as N processes each seeding
        their own generator do under Lightning DDP."""
        self.grad_sync = sync
        self.env.seed = self._seed + 1_000_003 * int(rank)

    def _init_text_embedding(self, t):
        torch.nn.init.normal_(t)
        t[self.padding_idx].zero_()

    def _reorder_state_dict_keys(self):
        """state_dict() lists keys in the reference's module registration order."""
        # (a module's own parameters come before its children's: the frozen pitch_bins / energy_bins are
        # nn.Parameters OF the VarianceAdaptor, fs2/variance_adaptor.py:117-148, so they lead its section)
        groups = ["text_input_layer.", "position_embedding.", "gst.", "encoder.",
                  "variance_adaptor.pitch_bins", "variance_adaptor.energy_bins", "variance_adaptor.duration_predictor.",
                  "variance_adaptor.pitch_predictor.", "variance_adaptor.pitch_embedding.",
                  "variance_adaptor.energy_predictor.", "variance_adaptor.energy_embedding.",
                  "variance_adaptor.attention.", "decoder.", "mel_linear.", "postnet.", "speaker_embedding.",
                  "language_embedding."]
        names = self.store.order_hint
        self.store.order_hint = sorted(names, key=lambda n: next((i for i, g in enumerate(groups) if n.startswith(g)), 99))

    @property
    def current_epoch(self):
        try:
            return super().current_epoch  # Lightning
        except Exception:
            return self.current_epoch_

    def train(self, mode: bool = True):
        self.training = mode
        self.env.training = mode
        return self

    def eval(self):
        return self.train(False)

    def parameters(self, recurse=True):  # the flat buffer is the parameter
        return iter([self.flat_param])

    def named_parameters(self, *a, **k):
        return iter([("flat_param", self.flat_param)])

    def state_dict(self, *a, **k):
        return self.store.state_dict()

    def load_state_dict(self, sd, strict=True, **k):
        return self.store.load_state_dict(sd, strict)

    def _table(self, T: int):
        """Positional table [T, D] (fs2/layers.py:132-140), cached per length."""
        tab = self._tables.get("tab")
        if tab is None or tab.shape[0] < T:
            d = self.config.model.encoder.input_dim
            tab = H.posenc_table(self.store.b("position_embedding.inv_freq"), max(T, 256), d)
            self._tables["tab"] = tab
        return tab

    def _dev(self, t, dtype=None):
        if t is None or not torch.is_tensor(t):
            return t
        t = t.to(self.device_, non_blocking=True)
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        return t.contiguous()

    def prepare_batch(self, batch):
        """Moves the collated batch to the GPU with the dtypes the kernels take."""
        b = dict(batch)
        for k in ("text", "src_lens", "mel_lens", "speaker_id", "language_id"):
            if b.get(k) is not None:
                b[k] = self._dev(b[k], torch.int32)
        for k in ("mel", "pitch", "energy"):
            if b.get(k) is not None:
                b[k] = self._dev(b[k], torch.float32)
        if self.use_pfs and b.get("pfs") is not None and b["pfs"].shape[-1] == N_PHONOLOGICAL_FEATURES:
            # feature rows padded with zero columns to a multiple of 4 floats (the GEMM reads 16-byte pieces)
            pfs = self._dev(b["pfs"], torch.float32)
            padded = torch.zeros(*pfs.shape[:-1], (N_PHONOLOGICAL_FEATURES + 3) // 4 * 4, device=pfs.device)
            padded[..., :N_PHONOLOGICAL_FEATURES] = pfs
            b["pfs"] = padded
        if b.get("duration") is not None and torch.is_tensor(b["duration"]):
            # learned alignment: "duration" carries the (B, Tm, Ts) attention prior (fs2/dataset.py:274-281)
            prior = self.config.model.learn_alignment and b["duration"].dim() == 3
            b["duration"] = self._dev(b["duration"], torch.float32 if prior else torch.int32)
        return b

    # ---- per-block precision (measurement aid: tools/bf16_error_budget.py) ---------------------------------
    #: {"encoder" | "decoder" | "postnet" | "adaptor" | "mel_linear" | "gst": precision} -- the named block's GEMMs run in
    #: that precision instead of the model's (forward and backward alike).  Empty in every product path.
    precision_overrides: dict = {}

    def _prec(self, block):
        import contextlib
        over = self.precision_overrides.get(block)
        if over is None:
            return contextlib.nullcontext()

        @contextlib.contextmanager
        def ctx():
            H.set_precision(over)
            try:
                yield
            finally:
                H.set_precision(self.precision)
        return ctx()

    # ---- forward (fs2/model.py:153-268) -------------------------------------------------------------
    def forward(self, batch, control=None, inference=False):
        with torch.cuda.device(self.device_):  # kernels launch on the current device: it must be the model's
            return self._forward(batch, control, inference)

    def _forward(self, batch, control=None, inference=False):
        H.set_precision(self.precision)
        if self.env.stored:
            # the weights as bf16, once per step: every GEMM orientation reads this mirror (the transposed ones only
            # serve data gradients: a training forward)
            self.store.refresh_bf16(transposed=self.training and not inference)
        elif H.GEMM_BF16 == 0 and self.training and not inference and FP32_TRANSPOSED:
            self.store.refresh_transposed_fp32()  # W^T of the K = 256 data-gradient weights (streaming fp32 kernel)
        control = control or InferenceControl()
        if "duration_control" in batch and batch["duration_control"] and batch["duration_control"][0]:
            control.duration = batch["duration_control"][0]
        batch = self.prepare_batch(batch)
        teacher_forcing = bool(inference and batch.get("mel_lens") is not None)
        S, m = self.store, self.config.model
        text, src_lens = (batch["pfs"] if self.use_pfs else batch["text"]), batch["src_lens"]
        B, Ts = text.shape[:2]
        if int(batch["max_src_len"]) != Ts:
            raise ValueError("max_src_len must equal the padded text length")
        save = self.training and not inference
        if self.use_pfs:
            inputs = H.linear_fwd(text, S.p("text_input_layer.weight"))
        else:
            inputs = H.embedding_fwd(text, S.p("text_input_layer.weight"))
        gst_ctx, style = None, None
        if self.gst is not None:
            # fs2/model.py:196-203: a style-reference mel in inference, token 0 in free inference, else the target mel.
            # The style encoder reads only the mel: it runs on the side stream beside the text encoder (whose
            # kernels are small and latency-bound) and is joined where its vector is added.
            ref_mel = batch.get("mel_style_reference")
            with self.env.side(batch.get("mel"), ref_mel if torch.is_tensor(ref_mel) else None):
                if inference and torch.is_tensor(ref_mel):
                    style, _ = self.gst.fwd(self._dev(ref_mel, torch.float32))
                elif inference and not teacher_forcing:
                    style = self.gst.condition_on_gst_tokens(B)
                else:
                    with self._prec("gst"):
                        style, gst_ctx = self.gst.fwd(batch["mel"])
        x = H.add_posenc(inputs, self._table(Ts), src_lens, B, Ts)
        with self._prec("encoder"):
            x, enc_ctx = self.encoder.fwd(x, src_lens)
        if self.gst is not None:
            self.env.join()
            x = H.add_rowvec(x, style, B, Ts)
        if m.multispeaker:
            x = H.add_rowvec(x, H.embedding_fwd(batch["speaker_id"], S.p("speaker_embedding.weight")), B, Ts)
        if m.multilingual:
            x = H.add_rowvec(x, H.embedding_fwd(batch["language_id"], S.p("language_embedding.weight")), B, Ts)
        Tm = batch["max_mel_len"]
        with self._prec("adaptor"):
            va, va_ctx = self.variance_adaptor.fwd(x, batch, src_lens, self._table, int(Tm), control, inference,
                                                   teacher_forcing, text_emb=inputs)
        self._hard_idx = va_ctx.get("hard_idx")
        if va_ctx.get("bad") is not None:
            self._pending_bad.append((va_ctx["bad"], list(batch.get("basename") or [])))
            if len(self._pending_bad) > 4096:
                self.check_bad_data()
        self._early_ctc = None
        if save and va.get("attn_logprob") is not None and self.env.side_enabled:
            # The forward-sum (CTC) loss of the aligner and its gradient are a 2 x Tm-step serial recursion per
            # utterance (1.4 ms on 32 wavefronts): started now on the side stream, it runs under the decoder and
            # PostNet instead of between forward and backward.  fs2/loss.py:109-116 gives weight and inputs.
            lp = va["attn_logprob"]
            slot = H.zeros(1, device=lp.device)
            with self.env.side(lp, slot, batch["src_lens"], batch["mel_lens"]):
                g_ctc = H.attn_ctc_loss(lp.view(lp.shape[0], lp.shape[2], lp.shape[3]), batch["src_lens"],
                                        batch["mel_lens"], self.config.training.attn_ctc_loss_weight, slot, want_grad=True)
            self._early_ctc = (slot, g_ctc)
        Tm, tgt_lens = va["Tm"], va["tgt_lens"]
        if (teacher_forcing or not inference) and batch.get("mel") is not None and batch["mel"].shape[1] != Tm:
            raise ValueError("max_mel_len must equal the padded mel length")
        with self._prec("decoder"):
            y, dec_ctx = self.decoder.fwd(va["output"], tgt_lens)
        with self._prec("mel_linear"):
            output = H.linear_fwd(y, S.p("mel_linear.weight"), S.p("mel_linear.bias"))
        postnet_output, post_ctx = None, None
        if m.use_postnet:
            with self._prec("postnet"):
                post, post_ctx = self.postnet.fwd(output)
            postnet_output = H.axpby(output, post)
        if save:
            self._ctx = dict(text=text, enc=enc_ctx, va=va_ctx, dec=dec_ctx, dec_out=y, post=post_ctx, B=B, Ts=Ts, Tm=Tm,
                             batch=batch, gst=gst_ctx)
        if self.env.training and self.store.bn_counters.numel():
            # every BatchNorm of the model has run once in train mode (an ATen add: order-free, so a recorded launch
            # plan re-runs it after each replay instead of containing it)
            H.plan_host_op(lambda c=self.store.bn_counters: c.add_(1))
        self.env.join()  # variance predictors of a teacher-forced forward ran on the side stream under the decoder
        if not self._in_step:  # called directly (evaluation, teacher forcing): raise at once, as the reference does
            self.check_bad_data()
        return {
            "output": output, "postnet_output": postnet_output,
            "src_mask": H.mask_from_lens(src_lens, Ts), "src_lens": src_lens,
            "tgt_mask": H.mask_from_lens(tgt_lens, Tm), "tgt_lens": tgt_lens,
            "attn_logprob": va["attn_logprob"], "attn_soft": va["attn_soft"], "attn_hard": va["attn_hard"],
            "duration_prediction": va["duration_prediction"], "duration_target": va["duration_target"],
            "energy_prediction": va["energy_prediction"], "energy_target": va["energy_target"],
            "pitch_prediction": va["pitch_prediction"], "pitch_target": va["pitch_target"],
            "text_input": text,
        }

    __call__ = forward

    # ---- backward -----------------------------------------------------------------------------------
    def backward(self):
        """Fills ``store.grad`` with d(total loss)/d(parameters) for the last training forward + loss."""
        with torch.cuda.device(self.device_):
            prev = H.defer_slab_reductions(True)  # split-K finishes of the weight gradients: batched at the flushes
            try:
                return self._backward()
            except BaseException:
                H.drop_pending_reductions()  # (their outputs belong to a backward pass that did not finish)
                raise
            finally:
                H.defer_slab_reductions(prev)
                H.hold_weight_gradients(False)

    def _backward(self):
        if self._ctx is None or self._loss_grads is None:
            raise RuntimeError("backward() needs a training-mode forward() and loss() first")
        H.set_precision(self.precision)
        S, c, g, m = self.store, self._ctx, self._loss_grads, self.config.model
        sync = self.grad_sync
        with self._prec("adaptor"):
            self.variance_adaptor.bwd_predictors_early(g, c["va"])
        hold = HOLD_WGRADS and not sync
        if hold:
            H.hold_weight_gradients(True)
        d_out = g["spec"]
        if m.use_postnet:
            d_post = g["postnet"]
            d_out = H.axpby(d_out, d_post)
            with self._prec("postnet"):
                d_out = H.axpby(d_out, self.postnet.bwd(d_post, c["post"]))
        with self._prec("mel_linear"):
            H.linear_bwd_weight(d_out, c["dec_out"], S.g("mel_linear.weight"), bias_grad=S.g("mel_linear.bias"))
            d = H.linear_bwd_data(d_out, S.p("mel_linear.weight"))
        self._bucket_done(self._bucket_head)                              # mel head + PostNet
        with self._prec("decoder"):
            d = self.decoder.bwd(d, c["dec"], layer_done=self._bucket_done)   # one bucket per decoder layer but the first
        self._bucket_done(self.decoder.buckets[0])
        with self._prec("adaptor"):
            d, d_text = self.variance_adaptor.bwd(d, g, c["va"])
        if c.get("gst") is not None:
            with self.env.side(d):  # parameter gradients only: beside the encoder's backward pass
                with self._prec("gst"):
                    self.gst.bwd(self._rowsum(d), c["gst"])
        if m.multispeaker:
            self._rowvec_embedding_bwd("speaker_embedding.weight", c["batch"]["speaker_id"], d)
        if m.multilingual:
            self._rowvec_embedding_bwd("language_embedding.weight", c["batch"]["language_id"], d)
        self._bucket_done(self._bucket_va)                                # variance adaptor, GST, speaker / language
        if EARLY_FLUSH and not sync and self.env.side_enabled and not hold:
            with self.env.side():   # (after everything enqueued so far on both streams: lane 0 is in order)
                self.env._side_held.extend(H.flush_grad_reductions())
        if hold:
            # from here on weight gradients run as they come (the encoder's own), and the held ones are released in
            # equal shares behind each of the encoder's layers
            n_held = H.held_weight_gradients()
            share = -(-n_held // max(len(self.encoder.layers), 1))
            pending = list(H.hold_weight_gradients(False))

            def release_share(_bucket=None):
                if not pending:
                    return
                jobs = pending[:share]
                del pending[:share]
                with self.env.side(*[t for j in jobs for t in j[:2]]):
                    for dy_, x_, out_, kw_ in jobs:
                        H._linear_bwd_weight(dy_, x_, out_, **kw_)
            release_share()
        with self._prec("encoder"):
            d = self.encoder.bwd(d, c["enc"], layer_done=(release_share if hold else self._bucket_done))
        if hold:
            while pending:
                release_share()
        if d_text is not None:  # the aligner's keys are the raw text embedding (fs2/variance_adaptor.py:254)
            d = H.axpby(d, d_text)
        if self.use_pfs:
            H.linear_bwd_weight(d, c["text"], S.g("text_input_layer.weight"))
        else:
            H.embedding_bwd(c["text"].reshape(-1), d, S.g("text_input_layer.weight"), self.padding_idx)
        self._bucket_done(self.encoder.buckets[0])                        # + text embedding
        self.env.join()
        H.flush_grad_reductions()  # single GPU: everything at once, after the join
        self._ctx = self._loss_grads = None

    def _bucket_done(self, bucket: int):
        """Data parallel: every gradient of ``bucket`` has been enqueued (main chain, side stream and deferred
        second-stage sums).  The hand-off runs ON the side stream, after an event of the main one: the bucket's
        all-reduce then waits for the weight-gradient GEMMs without the main chain ever waiting for them."""
        sync, env = self.grad_sync, self.env
        if not sync:
            return
        if env.side_enabled:
            # the pending second-stage sums include the predictors' (their backward chains run on lanes of their own):
            # the main stream waits for THOSE lanes -- short chains enqueued at the start of the backward pass -- and
            # never for lane 0, where the weight-gradient GEMMs run
            env.join(only=[l for l in set(PRED_LANES.values()) if l != 0])
            with env.side():
                env._side_held.extend(H.flush_grad_reductions())
                H.plan_callback(lambda: sync.bucket_ready(bucket))
        else:
            H.flush_grad_reductions()
            H.plan_callback(lambda: sync.bucket_ready(bucket))

    def _rowsum(self, d):
        """[B, T, D] -> [B, D]: gradient of a per-utterance vector that was broadcast over time (one launch)."""
        B, T, D = d.shape
        return H.segment_colsum(d, torch.empty(B, D, device=d.device, dtype=torch.float32))

    def _rowvec_embedding_bwd(self, name, ids, d):
        H.embedding_bwd(ids, self._rowsum(d), self.store.g(name))

    # ---- Lightning-style hooks ----------------------------------------------------------------------
    def training_step(self, batch, batch_idx=0):
        """fs2/model.py:384-390 -- forward, losses, and (here) the backward pass as well.  Returns the total loss as a
        tensor autograd can differentiate with respect to the flat parameter: ``loss.backward()`` (Lightning's automatic
        optimization, or a plain torch loop) delivers the gradient this call has already computed -- see
        ``_DeliverGrad``.  The native loops (``fs2l train``, ``bench.py``) skip that call and let the optimizer read
        the flat gradient buffer directly."""
        if not self.training:
            raise RuntimeError("training_step() needs model.train()")
        self._adopt_trainer_precision()
        p = self.flat_param
        if p.grad is not None and p.grad.untyped_storage().data_ptr() == self.store.grad.untyped_storage().data_ptr():
            # Lightning calls ``zero_grad`` AFTER ``training_step`` (closure order: training_step -> zero_grad ->
            # backward), so at this point ``p.grad`` normally still aliases the gradient buffer of the previous step.
            # If the optimizer has stepped since that gradient was delivered it has been consumed: the stale alias is
            # dropped (the coming zero_grad would drop it anyway).  Only a gradient that is still waiting for its
            # optimizer step (accumulation across batches without zero_grad) must survive the backward pass below.
            opt = getattr(self, "optimizer", None)
            if opt is not None and getattr(opt, "steps_since_delivery", 0) > 0:
                p.grad = None
            else:
                p.grad = p.grad.clone()
        self._in_step = True
        try:
            losses, output = self._planned_step(batch)
        finally:
            self._in_step = False
        self._raise_if_bad_in_step()  # (models that learn the alignment: BadDataError in the offending step)
        self.last_losses, self.last_output = losses, output
        # fs2/model.py:387-389 -- the reference reads every term back with ``.item()`` (6-8 host syncs per step); here
        # the values stay 0-dim device tensors (views of ONE slot vector): Lightning keeps logged tensors on the device
        # and reads them at its logging / progress-bar interval, the native loops read the whole vector in one copy
        # (``losses_to_host``).  No synchronisation is added to the step.
        self.log_dict({f"training/{k}_loss": v for k, v in losses.items()}, prog_bar=True)
        if torch.is_grad_enabled():
            return _DeliverGrad.apply(p, losses["total"], self)
        return losses["total"]

    # ---- the step itself: eager, recorded, or replayed from its launch plan (plan.py) ------------------------------------
    def _run_step(self, batch):
        output = self(batch)
        losses = self.loss(output, self._ctx["batch"], self.current_epoch)
        self.backward()
        return losses, output

    def _plan_signature(self, batch):
        """Everything a recorded step's launch sequence depends on: the batch's geometry (tensor shapes, which optional
        entries are there), and the model-side switches that choose kernels, streams or arguments.  None: not plannable."""
        if not (PL.ENABLED and self.plan_enabled and H.GEMM_PROFILE is None and not self.precision_overrides and H._REC is None
                and self.variance_adaptor is not None) or torch.cuda.is_current_stream_capturing():
            return None
        geo = []
        for k in sorted(batch):
            v = batch[k]
            if torch.is_tensor(v):
                geo.append((k, tuple(v.shape)) if v.dim() else (k, int(v)))
            elif isinstance(v, (int, float)) or v is None:
                geo.append((k, v))
        t = self.config.training
        bin_w = min(self.current_epoch / t.attn_bin_loss_warmup_epochs, 1.0) if self.config.model.learn_alignment else 0.0
        sync = self.grad_sync
        weights = (t.pitch_loss_weight, t.energy_loss_weight, t.duration_loss_weight, t.mel_loss_weight,
                   t.postnet_loss_weight, t.attn_ctc_loss_weight, t.attn_bin_loss_weight)
        return (tuple(geo), self.precision, bool(self.env.side_enabled), tuple(PRED_LANES.values()), id(sync) if sync else 0, bin_w, weights,
                getattr(self.postnet, "dropout_p", None), M.BF16_CHAIN, M.PRED_STORED, M.POSTNET_IM2COL, FP32_TRANSPOSED, HOLD_WGRADS, EARLY_FLUSH, PRED_GROUP, M.WGRAD_GROUP_ROWS, H.GEMM_GROUP, self.env.seed, H.plan_flags())

    def _planned_step(self, batch):
        sig = self._plan_signature(batch)
        if sig is None:
            return self._run_step(batch)
        plans = self.plans
        plan = plans.lookup(sig)
        if plan is not None:
            with torch.cuda.device(self.device_):
                plan.feed(self.prepare_batch(batch))
                losses, output = plan.replay(self.env.side_streams())
            # the loss terms are handed out as a COPY of the recorded slot vector (one 32-byte ATen copy): Lightning keeps
            # logged tensors and reads them later, and the next replay rewrites the recorded vector in place
            slots = plan.extra["loss_slots"].clone()
            losses = OrderedDict((k, slots[i]) for k, i in plan.extra["loss_index"])
            self._loss_slots, self._hard_idx = slots, plan.extra["hard_idx"]
            if plan.extra["bad"] is not None:
                self._pending_bad.append((plan.extra["bad"], list(batch.get("basename") or [])))
            plans.replayed += 1
            return losses, output
        if not plans.should_record(sig):
            plans.eager += 1
            return self._run_step(batch)
        with torch.cuda.device(self.device_):
            pb = self.prepare_batch(batch)
            inputs = {k: v for k, v in pb.items() if torch.is_tensor(v) and v.is_cuda}
            n_bad = len(self._pending_bad)
            plan, (losses, output) = PL.record(lambda: self._run_step(pb), inputs, self.device_)
        bad = self._pending_bad[-1][0] if len(self._pending_bad) > n_bad else None
        index = [(k, len(LOSS_KEYS) if k == "total" else LOSS_KEYS.index(k)) for k in losses]
        plan.extra = dict(loss_slots=self._loss_slots, loss_index=index, hard_idx=self._hard_idx, bad=bad)
        plans.store(sig, plan)
        plans.recorded += 1
        # (the recorded step's own loss terms are handed out as a copy too: replays rewrite the recorded vector)
        slots = self._loss_slots.clone()
        self._loss_slots = slots
        return OrderedDict((k, slots[i]) for k, i in index), output

    def losses_to_host(self, losses=None) -> dict:
        """Every term of the last ``loss()`` as Python floats with ONE device-to-host copy (the slot vector)."""
        losses = self.last_losses if losses is None else losses
        host = self._loss_slots.detach().cpu().tolist()
        out = {k: host[LOSS_KEYS.index(k)] for k in losses if k != "total"}
        out["total"] = host[len(LOSS_KEYS)]
        return out

    def _bad_probe(self):
        """Host side of the reference's IN-STEP ``BadDataError`` (fs2/variance_adaptor.py:289-305 raises inside the
        step's forward): right behind the kernel that compares the aligner's durations with ``mel_lens``, the device
        counter is copied into pinned host memory and an event is recorded.  ``training_step`` waits for THAT event -- the
        GPU reaches it early in the step -- not for the step, and raises before the optimizer can apply a step computed
        from corrupt data.  A launch-plan callback: replays repeat it at the same place."""
        if not self._in_step:
            return
        if self._bad_event is None:
            self._bad_event = torch.cuda.Event()
        self._bad_host.copy_(self.bad_count, non_blocking=True)
        self._bad_event.record()
        self._bad_probed = True

    def _raise_if_bad_in_step(self):
        if not self._bad_probed:
            return
        self._bad_probed = False
        self._bad_event.synchronize()
        if int(self._bad_host) != self._bad_seen:
            self.check_bad_data()

    def check_bad_data(self):
        """fs2/variance_adaptor.py:289-305: raises ``BadDataError`` naming the utterances whose aligner durations did not
        add up to ``mel_lens``.  One 4-byte read of the device counter; the per-utterance flags are only fetched when it
        moved.  Called where the host synchronises anyway: a direct ``forward()``, ``validation_step``, the trainer's
        logging interval."""
        pend, self._pending_bad = self._pending_bad, []
        if not pend:
            return
        n = int(self.bad_count.cpu())
        if n == self._bad_seen:
            return
        self._bad_seen = n
        mismatches = [name for flags, names in pend for name, f in zip(names or map(str, range(flags.numel())), flags.cpu().tolist()) if f]
        raise BadDataError(f"Something failed with the following items, please check them for errors: {mismatches}")

    def validation_step(self, batch, batch_idx=0):
        """fs2/model.py:515-528 (plots/audio logging are out of scope): forward, losses, and the epoch-mean metrics
        ``validation/{k}_loss`` with ``sync_dist=True`` -- ``validation/total_loss`` is what the reference's
        ``ModelCheckpoint`` monitors (fs2/cli/train.py:37)."""
        self._adopt_trainer_precision()
        was = self.training
        self.eval()
        try:
            output = self(batch)
            losses = self.loss(output, self.prepare_batch(batch), self.current_epoch)
            self.log_dict({f"validation/{k}_loss": v for k, v in losses.items()}, batch_size=self.batch_size,
                          sync_dist=True)
        finally:
            self.train(was)
        return losses

    # ---- metric logging without Lightning ---------------------------------------------------------------------
    # ``LightningModule.log_dict`` feeds ``trainer.callback_metrics``, which ``ModelCheckpoint(monitor=...)`` reads.
    # Lightning is not in the build image, so when ``_Base`` is ``nn.Module`` the same call sites write into
    # ``self.callback_metrics`` with Lightning's default reductions for these two hooks: the latest value of a training
    # step (``on_step``), the batch-size-weighted epoch mean of validation steps (``on_epoch``, summed over ranks when
    # ``sync_dist``).  Values stay device tensors; nothing here synchronises with the host.
    if _Base is torch.nn.Module:
        def log_dict(self, dictionary, prog_bar=False, batch_size=None, sync_dist=False, **kwargs):
            if self.training:
                for k, v in dictionary.items():
                    self.callback_metrics[k] = v.detach() if torch.is_tensor(v) else v
                return
            w = float(batch_size or 1)
            for k, v in dictionary.items():
                v = v.detach().float() if torch.is_tensor(v) else torch.tensor(float(v), device=self.device_)
                acc = self._val_acc.get(k)
                if acc is None:
                    self._val_acc[k] = [v * w, w, bool(sync_dist)]
                else:
                    acc[0] = acc[0] + v * w
                    acc[1] += w

        def on_validation_epoch_end(self):
            """Closes the validation epoch of ``log_dict``: means into ``callback_metrics`` (one all-reduce of the
            stacked sums and the weight under ``sync_dist`` when a process group is up)."""
            import torch.distributed as dist
            acc, self._val_acc = self._val_acc, {}
            if not acc:
                return
            keys = list(acc)
            stat = torch.stack([acc[k][0] for k in keys] + [torch.tensor(acc[keys[0]][1], device=self.device_)])
            if any(a[2] for a in acc.values()) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                dist.all_reduce(stat)
            mean = stat[:-1] / stat[-1]
            for i, k in enumerate(keys):
                self.callback_metrics[k] = mean[i]

    def on_validation_start(self):
        """Lightning's hook in front of a validation pass.  Data parallel through ``parallel.GradSync``: every rank
        evaluates with rank 0's BatchNorm running statistics, as under torch DDP's ``broadcast_buffers=True``
        (``GradSync.broadcast_buffers``; under Lightning's own DDP strategy the wrapper does it and ``grad_sync`` is None)."""
        if self.grad_sync is not None:
            self.grad_sync.broadcast_buffers(0)

    def on_train_batch_end(self, outputs=None, batch=None, batch_idx=0):
        """Lightning's per-step hook: the reference raises ``BadDataError`` inside the step's forward
        (fs2/variance_adaptor.py:289-305); here the check is one 4-byte read after the step has been enqueued (only
        models that learn the alignment ever queue anything to check)."""
        if self._pending_bad:
            self.check_bad_data()

    def _adopt_trainer_precision(self):
        """``Trainer(precision=...)`` reaches a LightningModule as ``self.trainer.precision`` ("32-true", "bf16-mixed",
        ...), not as a constructor argument: take it from there when a trainer is attached (fs2/cli/train.py:33-41
        passes the precision to the Trainer).  A precision this path has no arithmetic for is refused.

        Precedence: ``Trainer()`` defaults to "32-true", so a trainer value of "32-true" cannot be told from
        "nothing was asked for".  A precision given explicitly to the constructor (or through ``FS2_PRECISION``)
        therefore survives a trainer left at that default -- ``FastSpeech2(config, precision="bf16-mixed")`` under a
        plain ``Trainer()`` stays bf16-mixed, "32-split" stays reachable -- with one warning naming both values.  A
        NON-default trainer value is what the user asked the Trainer for and is adopted (warning when it overrides an
        explicit constructor value); a module built without a precision follows the trainer either way."""
        try:
            trainer = self.trainer
        except Exception:  # LightningModule.trainer raises while detached
            trainer = None
        prec = getattr(trainer, "precision", None)
        if prec is None or str(prec) == self._trainer_precision_seen:
            return
        key = str(prec)
        if key not in H.PRECISIONS:
            raise ValueError(f"Trainer(precision={prec!r}): this path runs \"32-true\" and \"bf16-mixed\" "
                             f"(and \"32-split\"); other precisions have no kernels here")
        self._trainer_precision_seen = key
        want = {0: "32-true", 1: "bf16-mixed", 2: "32-split"}[H.PRECISIONS[key]]
        if self._precision_explicit and want != self.precision:
            import warnings
            if want == "32-true":  # the Trainer's default: the explicit constructor value stands
                warnings.warn(f"FastSpeech2 was built with precision={self.precision!r} and the attached Trainer reports "
                              f"its default {key!r}: keeping {self.precision!r} (pass the precision to the Trainer, or "
                              "build the module without one, to follow the Trainer)", stacklevel=2)
                return
            warnings.warn(f"Trainer(precision={key!r}) overrides the module's precision={self.precision!r}", stacklevel=2)
        H.set_precision(want)
        self.precision = H.get_precision()

    def setup(self, stage=None):
        self._adopt_trainer_precision()

    def on_fit_start(self):
        self._adopt_trainer_precision()

    def _apply(self, fn, recurse=True):
        """``nn.Module._apply`` is what ``model.to(...)``, ``.half()``, ``.bfloat16()``, ``.cuda(i)`` and Lightning's
        strategy / precision plugins go through.  It would replace ``flat_param.data`` with a converted COPY and so
        cut the alias with ``store.flat`` that the kernels read and the optimizer writes -- silently.  A dtype change
        is refused (reduced-precision arithmetic is ``precision="bf16-mixed"``, the master weights stay fp32); a move
        to another GPU re-homes every buffer of the model; a no-op cast is a no-op."""
        cur = self.store.flat
        probe = fn(torch.empty(0, dtype=cur.dtype, device=cur.device))
        if probe.dtype != cur.dtype:
            raise RuntimeError(f"FastSpeech2 (MI355X build): cannot cast the module to {probe.dtype} -- the flat fp32 "
                               "parameter buffer is what the kernels and the fused optimizer address; use "
                               "precision=\"bf16-mixed\" (Trainer(precision=...)) for bf16 arithmetic")
        if probe.device != cur.device:
            if probe.device.type != "cuda":
                raise RuntimeError(f"FastSpeech2 (MI355X build): cannot move the module to {probe.device}: there is "
                                   "no CPU path")
            self.move_to(probe.device)
        return self

    def move_to(self, device, force=False):
        """Re-homes the model on another GPU (parameters, gradient, moments, buffers, device records), keeping the
        ``flat_param`` / ``store.flat`` alias.  Optimizers built before the move must be rebuilt
        (``configure_optimizers``): Lightning moves the module before it sets the optimizers up."""
        device = torch.device(device)
        if device == self.device_ and not force:
            return self
        S = self.store
        with torch.no_grad():
            for name in ("flat", "grad", "adam_m", "adam_v"):
                setattr(S, name, getattr(S, name).to(device, copy=True))
            counters = S.bn_counters.to(device, copy=True)
            names = [n for n in S.buffers if n.endswith("num_batches_tracked")]
            for n in list(S.buffers):
                if n not in names:
                    S.buffers[n] = S.buffers[n].to(device, copy=True)
            for i, n in enumerate(names):
                S.buffers[n] = counters[i]
            S.bn_counters = counters
            S.device = device
            S._pviews, S._gviews, S._bviews, S._tviews, S._tviews32, S.flat_bf16 = {}, {}, {}, {}, {}, None
            S.weights_changed()
            self.step_state = self.step_state.to(device, copy=True)
            self.env.step_state = self.step_state
            if getattr(self.env, "_lanes", None):
                self.env.join()
                self.env._lanes, self.env._side_stream = None, None
            self.plans.clear()   # recorded launch plans hold the old device's addresses and streams
            H.release_scratch()  # (scratch keyed by a destroyed side stream's handle would dangle)
            self.bad_count = self.bad_count.to(device, copy=True)
            if self.variance_adaptor is not None:
                self.variance_adaptor.bad_count = self.bad_count
            self.flat_param.data = S.flat
            self.flat_param.grad = None
        self.device_ = device
        self._tables, self._ctx, self._loss_grads, self._loss_slots = {}, None, None, None
        if getattr(self, "optimizer", None) is not None:
            self.optimizer = None  # held the old device record
        return self

    def predict_step(self, batch, batch_idx=0):
        was = self.training
        self.eval()
        out = self(batch, inference=True)
        self.train(was)
        return out

    def configure_optimizers(self):
        """fs2/model.py:530-549: ``([AdamW], [{"scheduler": NoamLR, "interval": "step"}])`` -- here a
        ``torch.optim.Optimizer`` whose ``step()`` is the fused clip + AdamW + schedule launch over the flat buffers and
        a ``torch.optim.lr_scheduler.LRScheduler`` that mirrors the device-resident schedule on the host."""
        from .optim import FusedAdamWNoam, NoamLR
        o = self.config.training.optimizer
        self.optimizer = FusedAdamWNoam(self.store, self.step_state, o.learning_rate, tuple(o.betas), o.eps,
                                        o.weight_decay, o.warmup_steps, param=self.flat_param,
                                        param_names=self.reference_parameter_names())
        self.scheduler = NoamLR(self.optimizer, o.warmup_steps)
        return [self.optimizer], [{"scheduler": self.scheduler, "interval": "step"}]

    def configure_gradient_clipping(self, optimizer, gradient_clip_val=None, gradient_clip_algorithm=None):
        """Lightning's hook for ``Trainer(gradient_clip_val=1.0)`` (fs2/cli/train.py:38), called between backward and
        ``optimizer.step()``.  Global-norm clipping is part of the fused optimizer launch (the norm is one pass over
        the flat gradient, the coefficient goes into the device record): the value is handed over instead of running
        ``torch.nn.utils.clip_grad_norm_``.  Clipping by value has no fused form and takes torch's."""
        from .optim import FusedAdamWNoam
        if gradient_clip_algorithm in (None, "norm") and isinstance(optimizer, FusedAdamWNoam):
            optimizer.max_grad_norm = gradient_clip_val
            return
        if gradient_clip_val:
            if gradient_clip_algorithm == "value":
                torch.nn.utils.clip_grad_value_(self.parameters(), gradient_clip_val)
            else:
                torch.nn.utils.clip_grad_norm_(self.parameters(), gradient_clip_val)

    # ---- checkpoint hooks (fs2/model.py:270-378) ------------------------------------------------------
    def on_save_checkpoint(self, checkpoint):
        checkpoint.setdefault("hyper_parameters", {})
        checkpoint["hyper_parameters"]["config"] = self.config.model_checkpoint_dump()
        if self.stats is not None:
            checkpoint["hyper_parameters"]["stats"] = self.stats.model_dump(mode="json")
        checkpoint["hyper_parameters"]["lang2id"] = self.lang2id
        checkpoint["hyper_parameters"]["speaker2id"] = self.speaker2id
        checkpoint["model_info"] = {"name": self.__class__.__name__, "version": self._VERSION}

    def check_and_upgrade_checkpoint(self, checkpoint):
        from packaging.version import Version
        info = checkpoint.get("model_info", {"name": self.__class__.__name__, "version": "1.0"})
        name = info.get("name", "MISSING_TYPE")
        if name != self.__class__.__name__:
            raise TypeError(f"Wrong model type ({name}), we are expecting a '{self.__class__.__name__}' model")
        version = Version(info.get("version", "0.0"))
        if version > Version(self._VERSION):
            raise ValueError("Your model was created with a newer version of EveryVoice, please update your software.")
        if version < Version("1.0"):
            checkpoint.setdefault("model_info", {})["version"] = "1.0"
        level = checkpoint["hyper_parameters"]["config"]["model"].get("target_text_representation_level")
        if version < Version("1.2") and level == TargetTrainingTextRepresentationLevel.phonological_features.value:
            raise ValueError("There were breaking changes to the handling of phonological features in version 1.2; "
                             f"your model is version {version}.")
        if version < Version("1.2"):
            # fs2/model.py:313-349: before 1.2 the embedding rows followed the old symbol order -- eight hard-coded
            # initial symbols, then the checkpoint's own symbols sorted -- and are moved to this model's symbol table
            remap_pre_1_2_text_embedding(checkpoint, self.text_processor.symbols,
                                         tuple(self.store.entries["text_input_layer.weight"].ref_shape))
        return checkpoint

    def on_load_checkpoint(self, checkpoint):
        checkpoint = self.check_and_upgrade_checkpoint(checkpoint)
        self.config = FastSpeech2Config(**checkpoint["hyper_parameters"]["config"])
        if checkpoint["hyper_parameters"].get("stats") is not None:
            self.stats = Stats(**checkpoint["hyper_parameters"]["stats"])

    # ---- checkpoint files in the layout Lightning writes for the reference --------------------------------------
    def reference_parameter_names(self) -> list:
        """The reference's ``named_parameters()`` order: the state-dict order without buffers, with the frozen
        ``pitch_bins`` / ``energy_bins`` (``nn.Parameter(requires_grad=False)``, fs2/variance_adaptor.py:117-148)
        in their places.  ``torch.optim.AdamW(self.parameters())`` indexes its state by position in this list."""
        frozen = ("variance_adaptor.pitch_bins", "variance_adaptor.energy_bins")
        return [n for n in self.store.order_hint if n in self.store.entries or n in frozen]

    def checkpoint_dict(self, global_step=0, epoch=0, optimizer=None) -> dict:
        """What ``Trainer.save_checkpoint`` hands to ``on_save_checkpoint`` for the reference model, filled from this
        model: weights under the reference's keys, ``hyper_parameters`` = the constructor arguments
        (``save_hyperparameters``, fs2/model.py:69), optimizer / scheduler state in torch's own format."""
        ckpt = {"epoch": int(epoch), "global_step": int(global_step), "pytorch-lightning_version": "2.6.1",
                "state_dict": OrderedDict((k, v.cpu()) for k, v in self.state_dict().items()),
                "hyper_parameters": {"lang2id": dict(self.lang2id), "speaker2id": dict(self.speaker2id)},
                "callbacks": {}}  # (no "loops": Lightning then takes progress from global_step / epoch; an empty dict
        #                          would be indexed for "fit_loop" by Trainer.fit(ckpt_path=...))
        if optimizer is not None:
            ckpt["optimizer_states"] = [optimizer.torch_state_dict(self.reference_parameter_names())]
            ckpt["lr_schedulers"] = [optimizer.torch_scheduler_state_dict()]
        self.on_save_checkpoint(ckpt)
        return ckpt

    def save_checkpoint(self, path, global_step=0, epoch=0, optimizer=None):
        torch.save(self.checkpoint_dict(global_step, epoch, optimizer), path)

    def restore_training_state(self, checkpoint: dict, optimizer) -> tuple:
        """Resume: optimizer moments + step (from torch-format ``optimizer_states`` -- written by the reference through
        Lightning or by ``save_checkpoint`` here), returns ``(global_step, epoch)``.  A checkpoint without optimizer
        state (weights only) is refused: resuming from it would silently restart Adam and the warm-up."""
        if not checkpoint.get("optimizer_states"):
            raise RuntimeError("checkpoint holds no optimizer state: it can initialise weights (finetune) but not resume")
        sched = (checkpoint.get("lr_schedulers") or [None])[0]
        optimizer.load_torch_state_dict(checkpoint["optimizer_states"][0], self.reference_parameter_names(), sched)
        self.current_epoch_ = int(checkpoint.get("epoch", 0))
        return int(checkpoint.get("global_step", 0)), int(checkpoint.get("epoch", 0))

    @classmethod
    def load_from_checkpoint(cls, path, device=None, precision="32-true", return_checkpoint=False):
        ckpt = torch.load(path, map_location="cpu", weights_only=False) if not isinstance(path, dict) else path
        hp = ckpt["hyper_parameters"]
        model = cls(hp["config"], hp.get("stats"), hp.get("lang2id"), hp.get("speaker2id"), device=device,
                    precision=precision)
        model.on_load_checkpoint(ckpt)
        model.load_state_dict(ckpt["state_dict"])
        return (model, ckpt) if return_checkpoint else model


OLD_HARDCODED_SYMBOLS = ("\x80", " ", "<EXCL>", "<QINT>", "<QUOTE>", "<BB>", "<SB>", "<EPS>")  # fs2/model.py:314-323


def symbols_of_checkpoint(symbol_dict: dict) -> list:
    """Every symbol of a checkpoint's ``config.text.symbols`` (restates the parent toolkit's
    ``get_symbols_from_checkpoint_symbol_dict``, which is not part of the reference repository: string values are
    one symbol, lists are taken element by element, nested dicts -- e.g. the punctuation categories -- recursively;
    order of first appearance, duplicates dropped)."""
    out = []

    def walk(v):
        if isinstance(v, str):
            if v not in out:
                out.append(v)
        elif isinstance(v, dict):
            for x in v.values():
                walk(x)
        elif isinstance(v, (list, tuple)):
            for x in v:
                walk(x)

    walk(symbol_dict)
    return out


def old_symbol_order(symbols: list, hardcoded_initial_symbols=OLD_HARDCODED_SYMBOLS) -> list:
    """The pre-1.2 embedding row order (restates ``symbol_sorter``): the hard-coded initial symbols, then the
    remaining symbols sorted."""
    head = list(hardcoded_initial_symbols)
    return head + sorted(set(symbols) - set(head))


def remap_pre_1_2_text_embedding(checkpoint: dict, model_symbols: list, weight_shape: tuple) -> None:
    """fs2/model.py:324-349: row i of the old table belongs to ``old_order[i]``; it moves to that symbol's index in the
    model's table (row 0 -- padding -- when the model does not know the symbol)."""
    old = old_symbol_order(symbols_of_checkpoint(checkpoint["hyper_parameters"]["config"]["text"]["symbols"]))
    if len(old) > len(model_symbols):
        raise AssertionError("unable to update the embedding table automatically: the checkpoint has more symbols "
                             "than the model (fs2/model.py:330-332)")
    w_old = checkpoint["state_dict"]["text_input_layer.weight"]
    if w_old.shape[0] != len(old):
        raise ValueError(f"pre-1.2 checkpoint: embedding has {w_old.shape[0]} rows but its symbol table has {len(old)}")
    idx = torch.tensor([model_symbols.index(c) if c in model_symbols else 0 for c in old], dtype=torch.long)
    new = torch.zeros(weight_shape, dtype=w_old.dtype)
    new[idx] = w_old
    checkpoint["state_dict"]["text_input_layer.weight"] = new
    print(f"checkpoint version < 1.2: text embedding rows moved to the current symbol order "
          f"({len(old)} -> {len(model_symbols)} symbols)", file=sys.stderr)
