"""TEST INFRASTRUCTURE -- definitions of the golden cases shared by
``oracle/make_golden.py`` (which runs the reference) and ``tests/``."""
from __future__ import annotations

from fastspeech2_lightning_amd import config as cfgmod

from . import fs2_oracle as O

STATS = dict(pitch=dict(min=0, max=1, std=1, mean=0, norm_min=-3, norm_max=3),
             energy=dict(min=0, max=1, std=1, mean=0, norm_min=-3, norm_max=3))

N_SYMBOLS = 24  # pad + 23 letters


def small_config(learn_alignment: bool, depthwise: bool = True, level: str = "phone", dropout: float = 0.0,
                 gst: bool = False, multispeaker: bool = False, n_mels: int = 16):
    d = 32
    conf = dict(layers=2, heads=2, input_dim=d, feedforward_dim=64, conv_kernel_size=9, dropout=dropout)
    vp = dict(n_layers=2, kernel_size=3, dropout=dropout, input_dim=d, n_bins=16, depthwise=depthwise)
    return cfgmod.FastSpeech2Config(
        model=dict(encoder=conf, decoder=conf, learn_alignment=learn_alignment,
                   use_global_style_token_module=gst, multispeaker=multispeaker, multilingual=multispeaker,
                   variance_predictors=dict(energy=dict(vp, level=level), pitch=dict(vp, level=level),
                                            duration=vp)),
        preprocessing=dict(audio=dict(n_mels=n_mels)),
        text=dict(symbols=dict(letters=[chr(ord("a") + i) for i in range(N_SYMBOLS - 1)])),
    )


_KW = dict(B=3, ts_lo=6, ts_hi=12, n_symbols=N_SYMBOLS, n_mels=16, dur_hi=4)

#: name -> (config kwargs, batch kwargs, train_mode)
CASES = {
    "e2e_noalign_eval": (dict(learn_alignment=False), dict(seed=11, **_KW), False),
    "e2e_noalign_train": (dict(learn_alignment=False), dict(seed=12, **_KW), True),
    "e2e_align_train": (dict(learn_alignment=True), dict(seed=13, learn_alignment=True, **_KW), True),
    "e2e_fullconv_frame_train": (dict(learn_alignment=False, depthwise=False, level="frame"),
                                 dict(seed=14, frame_level=True, **_KW), True),
    # BASELINE config 5 in miniature: GST reference encoder + speaker and language embeddings.  The reference's
    # GST conv stack needs >= 80-ish mel bins to leave 2 frequency bins after six stride-2 convs and its token
    # layer is fixed at 256 dims, so this case uses d = 256 with 1-layer Conformers.
    "e2e_gst_multispeaker_train": (dict(learn_alignment=False, gst=True, multispeaker=True, n_mels=80),
                                   dict(seed=15, B=3, ts_lo=6, ts_hi=12, n_symbols=N_SYMBOLS, n_mels=80, dur_hi=9), True),
}

SPEAKER2ID = {"spk0": 0, "spk1": 1, "spk2": 2}
LANG2ID = {"l0": 0, "l1": 1}

#: epoch passed to the loss in every case (bin-loss warm-up weight 5/100)
EPOCH = 5


def build(name: str):
    ckw, bkw, train = CASES[name]
    cfg = small_config(**ckw)
    batch = O.synthetic_batch(**bkw)
    if ckw.get("gst"):
        import torch
        d = 256  # the reference's StyleTokenLayer emits 256 dims: the model width must match
        conf = dict(layers=1, heads=2, input_dim=d, feedforward_dim=64, conv_kernel_size=9, dropout=0.0)
        vp = dict(n_layers=1, kernel_size=3, dropout=0.0, input_dim=d, n_bins=16, depthwise=True)
        dump = cfg.model_checkpoint_dump()
        dump["model"].update(encoder=conf, decoder=conf,
                             variance_predictors=dict(energy=dict(vp, level="phone"), pitch=dict(vp, level="phone"),
                                                      duration=vp))
        cfg = cfgmod.FastSpeech2Config(**dump)
        B = batch["text"].shape[0]
        batch["speaker_id"] = torch.arange(B, dtype=torch.int32) % len(SPEAKER2ID)
        batch["language_id"] = torch.arange(B, dtype=torch.int32) % len(LANG2ID)
    return cfg, batch, train
