"""TEST INFRASTRUCTURE -- definitions of the golden cases shared by
``oracle/make_golden.py`` (which runs the reference) and ``tests/``."""
from __future__ import annotations

from fastspeech2_lightning_amd import config as cfgmod

from . import fs2_oracle as O

STATS = dict(pitch=dict(min=0, max=1, std=1, mean=0, norm_min=-3, norm_max=3),
             energy=dict(min=0, max=1, std=1, mean=0, norm_min=-3, norm_max=3))

N_SYMBOLS = 24  # pad + 23 letters


def small_config(learn_alignment: bool, depthwise: bool = True, level: str = "phone", dropout: float = 0.0):
    d = 32
    conf = dict(layers=2, heads=2, input_dim=d, feedforward_dim=64, conv_kernel_size=9, dropout=dropout)
    vp = dict(n_layers=2, kernel_size=3, dropout=dropout, input_dim=d, n_bins=16, depthwise=depthwise)
    return cfgmod.FastSpeech2Config(
        model=dict(encoder=conf, decoder=conf, learn_alignment=learn_alignment,
                   variance_predictors=dict(energy=dict(vp, level=level), pitch=dict(vp, level=level),
                                            duration=vp)),
        preprocessing=dict(audio=dict(n_mels=16)),
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
}

#: epoch passed to the loss in every case (bin-loss warm-up weight 5/100)
EPOCH = 5


def build(name: str):
    ckw, bkw, train = CASES[name]
    return small_config(**ckw), O.synthetic_batch(**bkw), train
