"""ctypes binding of the fs2hip C ABI (include/fs2hip.h).

This is the only way the product reaches the GPU kernels: there is no CPU or
PyTorch fallback.  ``lib()`` raises if ``_fs2hip.so`` is missing, and every
wrapper checks on the host that shapes, strides, dtypes and devices match what
the kernel's grid assumes before launching (a faulting kernel can take the
whole node down).  Launches go to ``torch.cuda.current_stream()``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

# ROCclr maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  Once RCCL has created its streams the
# step's side stream lands on the SAME hardware queue as the main stream and the two-stream step runs as one queue:
# 20.9 ms against 18.7 ms per fp32 step with one RCCL rank (round 5, `FS2_BENCH_FORCE_SYNC=1`; every rank of a
# data-parallel job would pay it).  Eight queues keep them apart.  Read when the HIP runtime initialises, so it is set
# here -- and at the top of bench.py / cli.py, whose first GPU call precedes this import -- unless the user chose a value.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch

_LIB_PATH = Path(__file__).resolve().parent / "_fs2hip.so"
_lib = None

ACT_NONE, ACT_RELU, ACT_SILU, ACT_TANH = 0, 1, 2, 3
EPI_STORE, EPI_ACT, EPI_RESID, EPI_DACT = 0, 1, 2, 3
_ACT = {None: 0, "none": 0, "relu": 1, "silu": 2, "tanh": 3}


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("Mc", C.c_int), ("Nc", C.c_int), ("R", C.c_int),
        ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int),
        ("a_kcontig", C.c_int), ("b_kcontig", C.c_int),
        ("taps", C.c_int), ("T", C.c_int), ("tap_mul", C.c_int), ("tap_add", C.c_int),
        ("shift_operand", C.c_int),
        ("b_tap_stride", C.c_longlong), ("c_tap_stride", C.c_longlong),
        ("bias", C.c_void_p),
        ("epi", C.c_int), ("act", C.c_int),
        ("alpha", C.c_float),
        ("resid", C.c_void_p), ("ldr", C.c_int), ("res_scale", C.c_float),
        ("aux", C.c_void_p), ("ldaux", C.c_int),
        ("out_pre", C.c_void_p), ("ldpre", C.c_int),
        ("drop_p", C.c_float), ("drop_seed", C.c_ulonglong), ("drop_step", C.c_void_p),
        ("splitk", C.c_int), ("workspace", C.c_void_p), ("tile", C.c_int), ("workspace_floats", C.c_longlong),
        ("counters", C.c_void_p),
        ("operand_bf16", C.c_int),
        ("io_bf16", C.c_int),
        ("colsum", C.c_void_p),
    ]


_T = {"p": C.c_void_p, "i": C.c_int, "f": C.c_float, "q": C.c_longlong, "Q": C.c_ulonglong}

#: every symbol include/fs2hip.h declares -> argument type codes (checked by tests/test_abi.py)
SIGNATURES = {
    "fs2hip_version": "",
    "fs2hip_gemm": None,  # (const Fs2GemmArgs*, stream)
    "fs2hip_gemm_grouped": None,  # (const Fs2GemmArgs*, int, stream)
    "fs2hip_reduce_slabs_multi": "pip",
    "fs2hip_reduce_slabs": "ppqiqp",
    "fs2hip_reduce_rows_multi": None,  # (const Fs2ReduceJob*, int, void*): set below
    "fs2hip_colsum_rows": "i",
    "fs2hip_colsum": "piiippp",
    "fs2hip_layernorm_fwd": "ppppppiifp",
    "fs2hip_layernorm_bwd_blocks": "i",
    "fs2hip_layernorm_bwd": "ppppppppppiip",
    "fs2hip_layernorm_bwd_dz": "ppppppppffQppiip",
    "fs2hip_layernorm_fwd_b": "ppppppiifp",
    "fs2hip_layernorm_fwd_drop": "ppppppiiffQpp",
    "fs2hip_layernorm_bwd_pred": "ppppppipiifQpp",
    "fs2hip_layernorm_bwd_x": "ppppppppffQppiiip",
    "fs2hip_dwconv_bwd_b": "ppippipppiiiiip",
    "fs2hip_attention_fwd": "ppppiiiifQpip",
    "fs2hip_attention_bwd": "pppppppiiiifQpip",
    "fs2hip_attention_bwd_spill_supported": "i",
    "fs2hip_attention_bwd_spill": "pppppppqpiiiifQpp",
    "fs2hip_attention_fwd_s": "pppppqiiiifQpip",
    "fs2hip_attention_bwd_spill_s": "ppppppppqpiiiifQpp",
    "fs2hip_attention_fwd_b": "ppppiiiifQpp",
    "fs2hip_attention_bwd_b": "pppppppiiiifQpp",
    "fs2hip_attention_bwd_b_spill": "pppppppqpiiiifQpp",
    "fs2hip_attention_b_supported": "i",
    "fs2hip_dwconv_blocks": "ii",
    "fs2hip_dwconv_part_rows": "",
    "fs2hip_dwconv_fwd": "pippppiiiiiip",
    "fs2hip_dwconv_fwd_b": "pippppiiiiiiip",
    "fs2hip_dwconv_bwd": "ppipppppiiiiip",
    "fs2hip_colstats_parts": "i",
    "fs2hip_colstats_part_rows": "i",
    "fs2hip_colstats": "piipp",
    "fs2hip_colstats_b": "piipip",
    "fs2hip_bn_finalize": "piqiippppffipip",
    "fs2hip_bn_act_fwd": "pppiiifQpp",
    "fs2hip_bn_act_bwd": "ppppppppiiifQpip",
    "fs2hip_bn_act_fwd_b": "ppppiiifQpip",
    "fs2hip_bn_act_bwd_b": "pppppppppiiifQpiip",
    "fs2hip_posenc_table": "ppiip",
    "fs2hip_add_posenc": "ppppiiip",
    "fs2hip_embedding_fwd": "pppiiip",
    "fs2hip_onehot": "ppiiip",
    "fs2hip_bucket_embed_add": "pfpippppiip",
    "fs2hip_length_regulate_fwd": "pppppppiiiip",
    "fs2hip_length_regulate_bwd": "pppiiiip",
    "fs2hip_duration_cumsum": "ppppppiiip",
    "fs2hip_rowdot_fwd": "pppppiiip",
    "fs2hip_rowdot_blocks": "i",
    "fs2hip_rowdot_bwd": "ppppppppiiip",
    "fs2hip_masked_loss": "ppppiiiifpppp",
    "fs2hip_step_advance": "pffffp",
    "fs2hip_grad_clip_coef": "pqffppp",
    "fs2hip_adamw_step": "ppppqpffffp",
    "fs2hip_axpby": "pppqfffQpp",
    "fs2hip_cast_bf16": "ppqp",
    "fs2hip_im2col_taps": "pipiiiiip",
    "fs2hip_transpose_cast_bf16": "piiipiip",
    "fs2hip_transpose_cast_bf16_multi": None,  # (const Fs2TransposeJob*, int, void*): set below
    "fs2hip_add_rowvec": "pppiiip",
    "fs2hip_scale_dev": "pqpp",
    "fs2hip_dact_mul": "pppqip",
    "fs2hip_mask_from_lens": "ppiip",
    "fs2hip_duration_round": "pfpip",
    "fs2hip_sum_slots": "pipp",
    "fs2hip_attn_dist": "pppiiiip",
    "fs2hip_attn_softmax": "pppppiiip",
    "fs2hip_mas": "pippppppiiip",
    "fs2hip_avg_variance": "pppiiip",
    "fs2hip_attn_ctc_loss": "pppppppfpiiip",
    "fs2hip_attn_bin_loss": "pppfppiiip",
    "fs2hip_attn_softmax_bwd": "ppppppiiip",
    "fs2hip_attn_dist_bwd": "pppppiiiip",
    "fs2hip_conv2d_s2_fwd": "pppiiiiip",
    "fs2hip_im2col_s2": "ppiiiip",
    "fs2hip_col2im_s2": "ppiiiip",
    "fs2hip_conv2d_s2_bwd_data": "pppiiiiip",
    "fs2hip_conv2d_s2_wgrad_parts": "iii",
    "fs2hip_conv2d_s2_bwd_weight": "ppppiiiiip",
    "fs2hip_gru_gate_fwd": "pqppppiip",
    "fs2hip_gru_gate_bwd": "ppppqppiip",
    "fs2hip_gst_attn_fwd": "pppppiiip",
    "fs2hip_gst_attn_bwd": "ppppppppiiip",
    "fs2hip_act_apply": "ppqip",
    "fs2hip_memset": "piqp",
    "fs2hip_plan_op_count": "",
    "fs2hip_plan_op_id": None,      # (const char*)
    "fs2hip_plan_events_create": None,   # (void**, int)
    "fs2hip_plan_events_destroy": None,  # (void* const*, int)
    "fs2hip_plan_replay": None,     # (const Fs2PlanCmd*, int, int, void* const*, int, void* const*, int, int*)
}
EXPORTS = list(SIGNATURES)


def lib():
    """The loaded shared library.  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise RuntimeError(
                f"{_LIB_PATH} is missing: build it with `python -m fastspeech2_lightning_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no fallback path.")
        L = C.CDLL(str(_LIB_PATH))
        for name, sig in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = C.c_int
            if sig is not None:
                fn.argtypes = [_T[c] for c in sig]
        L.fs2hip_gemm.argtypes = [C.POINTER(GemmArgs), C.c_void_p]
        L.fs2hip_gemm_grouped.argtypes = [C.POINTER(GemmArgs), C.c_int, C.c_void_p]
        L.fs2hip_reduce_rows_multi.argtypes = [C.POINTER(ReduceJob), C.c_int, C.c_void_p]
        L.fs2hip_reduce_slabs_multi.argtypes = [C.POINTER(SlabJob), C.c_int, C.c_void_p]
        L.fs2hip_transpose_cast_bf16_multi.argtypes = [C.POINTER(TransposeJob), C.c_int, C.c_void_p]
        L.fs2hip_plan_op_id.argtypes = [C.c_char_p]
        L.fs2hip_plan_events_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.fs2hip_plan_events_destroy.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.fs2hip_plan_replay.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_int,
                                         C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
        _lib = L
    # while a launch plan is being recorded (plan.Recorder), every entry-point call also lands in the plan
    return _lib if _REC is None else _REC.lib


def real_lib():
    """The library itself, never the recording proxy (the tuner's timing launches, the plan replayer)."""
    lib()
    return _lib


#: the active ``plan.Recorder`` (None outside a recording step)
_REC = None


def plan_flags() -> tuple:
    """The module-level switches that choose kernels or launch counts: part of a launch plan's signature (tests and
    measurement tools flip them between steps of one model)."""
    return (int(GEMM_BF16), bool(BF16_STORAGE), bool(GEMM_TUNE), TILE_GEN[0], len(_TILE_CACHE), bool(ATTN_SPILL),
            bool(ATTN_SCORES), bool(ATTN_SPILL_B), bool(SPLITK_IN_KERNEL), int(SPLITK_MAX),
            os.environ.get("FS2_GEMM_TILE"), os.environ.get("FS2_DEFER_SLABS"), os.environ.get("FS2_CONV_DW_SPLITK"),
            os.environ.get("FS2_DWCONV_TILE"), os.environ.get("FS2_ATTN_GEN1"))


def plan_callback(fn):
    """Host-side work that belongs at THIS point of the step's launch sequence (a gradient bucket's hand-off to the
    all-reduce): runs now, and at the same point -- under the same current stream -- of every replay of a recorded plan."""
    if _REC is not None:
        # (a callback is HOST work the replayer repeats itself: whatever ATen it runs -- a collective, a pinned-memory
        # copy -- is not what the recording guard is looking for)
        from torch.utils._python_dispatch import _disable_current_modes
        with _disable_current_modes():
            fn()
        _REC.callback(fn, _stream())
    else:
        fn()


def plan_host_op(fn):
    """Order-free host-side work of a step that launches through ATen (not an entry point, so invisible to a recorded
    plan): runs now, or -- while a plan is recorded -- right after the recorded step and after every replay."""
    if _REC is not None:
        _REC.host_op(fn)
    else:
        fn()


_raw_stream = torch._C._cuda_getCurrentRawStream  # (device index) -> hipStream_t of torch's current stream
_current_device = torch._C._cuda_getDevice


def _stream() -> int:
    # ~0.3 us; torch.cuda.current_stream().cuda_stream costs ~3 us and this is called once per launch
    return _raw_stream(_current_device())


def _chk(t: torch.Tensor, dtype=torch.float32, name="tensor"):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"fs2hip: {name} must be a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"fs2hip: {name} must live in GPU memory (got {t.device}); there is no CPU path")
    if t.device.index != _current_device():
        # launches go to the CURRENT device's stream and scratch buffers: a tensor of another GPU would be a
        # cross-device pointer inside a kernel, i.e. a memory fault
        raise RuntimeError(f"fs2hip: {name} lives on cuda:{t.device.index} but the current device is "
                           f"cuda:{_current_device()}; run under torch.cuda.device(...) of the tensors' GPU")
    if t.dtype != dtype:
        raise TypeError(f"fs2hip: {name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"fs2hip: {name} must be contiguous")
    return t


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _ok(code: int, what: str):
    if code != 0:
        raise RuntimeError(f"fs2hip: {what} failed with code {code}"
                           + (" (invalid arguments)" if code == -22 else ""))


def _rows(t: torch.Tensor) -> int:
    return t.numel() // t.shape[-1]


def _req(cond: bool, msg: str):
    if not cond:
        raise ValueError("fs2hip: " + msg)


class Drop:
    """Dropout spec: probability, per-op seed, device step counter (uint64 tensor or None)."""
    __slots__ = ("p", "seed", "step")

    def __init__(self, p: float = 0.0, seed: int = 0, step: Optional[torch.Tensor] = None):
        self.p, self.seed, self.step = float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, step

    @property
    def step_ptr(self):
        return None if self.step is None else self.step.data_ptr()


NO_DROP = Drop()

# ------------------------------------------------------------------------------------------
# workspace (split-K slabs and partial sums): stream-ordered reuse
# ------------------------------------------------------------------------------------------
_WS = {}


#: 512 workgroup slots x 128 x 128 partial sums
HYBRID_WS_FLOATS = 512 * 128 * 128


def reserve_workspace(n: int, device) -> torch.Tensor:
    """Pre-size the scratch buffer (do this before capturing a hipGraph)."""
    key = (device.index if isinstance(device, torch.device) else device, _stream())  # one scratch buffer per stream
    ws = _WS.get(key)
    if ws is None or ws.numel() < n:
        dev = device if isinstance(device, torch.device) else torch.device("cuda", device)
        ws = torch.empty(max(n, 1 << 22), device=dev, dtype=torch.float32)
        _WS[key] = ws
    return ws


def _workspace(n: int, device) -> torch.Tensor:
    return reserve_workspace(n, device)


SPLITK_COUNTERS = 4096  # FS2_SPLITK_COUNTERS
#: "1": the weight-gradient GEMMs finish their split reduction themselves (Fs2GemmArgs.counters; 140 fewer launches per
#: step, bit-identical sums).  Off by default: measured time-neutral for the step (19.78 ms either way) while the GEMM
#: launches get 6 % longer (device-coherent slab stores + the last workgroup's tail)
SPLITK_IN_KERNEL = os.environ.get("FS2_SPLITK_IN_KERNEL", "0") != "0"
SPLITK_MAX = int(os.environ.get("FS2_SPLITK_MAX", 16))
_COUNTERS = {}


def splitk_counters(device) -> torch.Tensor:
    """The per-output-tile arrival counters of the current stream's split-K GEMMs (zero between launches: the
    workgroup that finishes a tile re-arms its counter).  Allocate before capturing a hipGraph."""
    key = (device.index if isinstance(device, torch.device) else device, _stream())
    c = _COUNTERS.get(key)
    if c is None:
        dev = device if isinstance(device, torch.device) else torch.device("cuda", device)
        c = _COUNTERS[key] = torch.zeros(SPLITK_COUNTERS, device=dev, dtype=torch.int32)
    return c


# ------------------------------------------------------------------------------------------
# GEMM family
# ------------------------------------------------------------------------------------------
#: when a list, every GEMM launch appends (start_event, end_event, algorithmic_flops, Mc, Nc, R):
#: bench.py's live roofline measurement of the dominant kernel (HIP events on the launch stream)
GEMM_PROFILE = None


#: workgroup-tile autotuner: (shape signature) -> tile id.  A signature is timed once (3 tile shapes x
#: a few launches with HIP events) the first time it is launched outside a graph capture; the launch
#: is idempotent (outputs are only overwritten), so re-running it for timing is safe.
GEMM_TUNE = os.environ.get("FS2_GEMM_TUNE", "1") != "0"
#: "bf16-mixed": every algorithmic GEMM rounds its operands to bf16 in registers (v_mfma_f32_32x32x16_bf16, fp32
#: accumulate / epilogue / storage).  Set through ``set_precision``; the fp32 path is the parity path and the default.
#: "32-split": fp32 accuracy on the bf16 matrix pipe -- every operand value is cut exactly into three bf16 planes in
#: registers and a product is the six partial products that matter, accumulated in fp32 (csrc/gemm2_core.h); the
#: GEMM family only (attention stays on the fp32 MFMA), same error bound as "32-true" (tests/test_gemm_split_gpu.py).
GEMM_BF16 = 0
#: "bf16-mixed" only: GEMMs whose operands can be had as bf16 in memory without an extra pass (today: the PostNet
#: convolutions, whose inputs / output gradients come from the BatchNorm kernels) read them as such
#: (Fs2GemmArgs.operand_bf16 = 3) instead of rounding fp32 operands in registers.  "0" = measurement aid.
BF16_STORAGE = os.environ.get("FS2_BF16_STORAGE", "1") != "0"
PRECISIONS = {"32-true": 0, "32": 0, "fp32": 0, "bf16-mixed": 1, "bf16": 1, "32-split": 2}


def set_precision(precision) -> None:
    """Mirror of Lightning's ``Trainer(precision=...)`` for this path: "32-true" (default) or "bf16-mixed"."""
    global GEMM_BF16
    key = str(precision)
    _req(key in PRECISIONS, f"precision must be one of {sorted(PRECISIONS)}, got {precision!r}")
    GEMM_BF16 = PRECISIONS[key]


def get_precision() -> str:
    return {0: "32-true", 1: "bf16-mixed", 2: "32-split"}[int(GEMM_BF16)]


GEMM_TILES_B = (20, 21, 22, 23, 24, 25, 26, 30, 31, 33)  # bf16-storage core (operand_bf16 == 4): 128x128, 128x64, 64x64;
#                                      24 / 25: persistent 128x128 / 128x64; 30 / 31: weights-stationary streaming form
#                                      (K = 256, forward orientation: csrc/gemm_ws.hip), 512 / 256 columns per workgroup;
#                                      33: the same for K = 1024 (csrc/gemm_ws4.hip), 128 columns per workgroup
GEMM_TILES = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 32)  # 1-3: register-staged BK=16 core; 4-9: direct-to-LDS BK=32 core;
#                                                      10-12: persistent direct-to-LDS core; 13-15: + split tail;
#                                                      32: weights-stationary streaming form (exact fp32, K = 256, forward
#                                                      orientation: csrc/gemm_ws32.hip)
#: FS2_GEMM_EXCLUDE_TILES=33,30 (measurement aid): tile ids the tuner leaves out -- same-box A/B runs of a kernel family
_EXCLUDED = {int(t) for t in os.environ.get("FS2_GEMM_EXCLUDE_TILES", "").split(",") if t.strip()}
if _EXCLUDED:
    GEMM_TILES_B = tuple(t for t in GEMM_TILES_B if t not in _EXCLUDED)
    GEMM_TILES = tuple(t for t in GEMM_TILES if t not in _EXCLUDED)
_TILE_CACHE = {}
#: bumped by every write to the tile table: a recorded launch plan (plan.py) carries the tiles of its recording and is
#: re-recorded when the table has changed since
TILE_GEN = [0]
#: per signature: [(isolated ms for 4 launches, tile), ...] sorted, and how often the signature was launched --
#: what ``refine_tiles_in_step`` works from
_TILE_TIMINGS = {}
_TILE_CALLS = {}


def refine_tiles_in_step(step, rounds: int = 5, candidates: int = 2, top: int = 16, min_gain: float = 0.004,
                         log=None, settle=None, count_step=None, agree=None, timer=None):
    """Second tuning stage, run once after warm-up: the per-shape tuner times a GEMM alone, back to back, with its
    operands warm in the Infinity Cache; inside the step the same launch sees cold outputs and a second stream.  For
    the ``top`` signatures by time the next-best isolated tiles are therefore tried IN the step (``step()`` = one whole
    training step, timed over ``rounds`` steps) and kept when the step gets faster by more than ``min_gain``.
    With launch plans a changed tile re-keys the plan: ``settle()`` (runs steps until one replays) is called after every
    change, so that what is timed is the replayed step -- whose timing is repeatable to a few hundredths of a millisecond
    -- and ``count_step()`` is an EAGER step (the per-signature launch counts come from the binding).  Data parallel:
    every rank must try the same tiles in the same order and take the same decisions (each trial runs collectives):
    ``agree(ms)`` returns the value all ranks decide on (the maximum over ranks), and the candidate lists must be rank 0's
    (``parallel.share_tile_table`` ships them with the tile table).  ``timer(step, rounds) -> ms per step`` replaces the
    HIP-event clock (tests)."""
    def event_timer(fn, n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n

    def timed():
        if settle is not None:
            settle()
        t = (timer or event_timer)(step, rounds)
        return agree(t) if agree is not None else t

    for v in _TILE_CALLS.values():
        v[0] = 0
    (count_step or step)()  # counts launches per signature
    base = min(timed(), timed())
    order = sorted((k for k in _TILE_TIMINGS if _TILE_CALLS.get(k, [0])[0] > 0),
                   key=lambda k: -_TILE_TIMINGS[k][0][0] * _TILE_CALLS[k][0])[:top]
    changed = 0
    for key in order:
        keep = _TILE_CACHE[key]
        for ms_iso, tile in _TILE_TIMINGS[key][1:1 + candidates]:
            _TILE_CACHE[key] = tile
            TILE_GEN[0] += 1
            t = min(timed(), timed())
            if t < base * (1.0 - min_gain):
                if log:
                    log(f"in-step tile refinement {key[:3]}: tile {keep} -> {tile}, step {base:.3f} -> {t:.3f} ms")
                base, keep, changed = t, tile, changed + 1
        _TILE_CACHE[key] = keep
        TILE_GEN[0] += 1
    if settle is not None:
        settle()
    return base, changed


def _q(n: int) -> int:
    """Long extents in eighths of an octave: training batches differ in their padded length from step to step (the
    row count of every decoder GEMM is B x Tm), and a tile that is fastest at 20 736 rows is fastest at 20 480 too --
    without this every new batch shape would re-run the tuner for every GEMM of the step."""
    if n <= 2048:
        return n
    g = 1 << (n.bit_length() - 4)
    return (n + g - 1) // g * g


def _tile_key(a):
    # everything that decides which tiles are legal for a launch or how fast they are: the shape, the operand
    # layouts, the conv geometry (T only matters to the shifted-operand cores, which refuse T < 32), the epilogue and
    # which of its tensors are present, 16-byte alignment of the output rows (the split-tail tiles need it), device
    flags = ((1 if a.bias else 0) | (2 if a.resid else 0) | (4 if a.aux else 0) | (8 if a.out_pre else 0)
             | (16 if (a.ldc % 4 == 0 and (a.C or 0) % 16 == 0) else 0) | (32 if a.drop_p > 0 else 0)
             | (64 if a.counters else 0) | (128 * a.io_bf16) | (512 if a.colsum else 0))
    # (a cached tile that an exact shape of the bucket does not admit: that launch alone takes the library's choice,
    # see _launch_gemm).  The padded length T of a convolution differs from batch to batch like the row counts do: it
    # is quantised the same way, with the two legality classes of the shifted-operand cores (fp32: T >= 32, bf16
    # storage: T >= 64) kept apart.
    tq = (_q(a.T), a.T >= 32, a.T >= 64) if a.taps > 1 else 0
    return (_q(a.Mc), _q(a.Nc), _q(a.R), a.taps, a.a_kcontig, a.b_kcontig, a.shift_operand, a.splitk, a.epi, a.operand_bf16,
            tq, flags, _current_device())


#: layout version of a tile signature (``_tile_key``): 2 = the convolution length enters as (q(T), T >= 32, T >= 64)
TILE_KEY_FORMAT = 2


def tile_table() -> dict:
    """The tuned tiles as {repr(signature): tile} (JSON-serialisable) plus ``"__format__"``: persist it with
    ``save_tile_cache`` or broadcast it so that every rank of a data-parallel job sums in the same order."""
    t = {repr(k): int(v) for k, v in _TILE_CACHE.items()}
    t["__format__"] = TILE_KEY_FORMAT
    if _GROUP_TILE_CACHE:  # tiles of grouped launches (``gemm_group``), keyed by their members' signatures
        t["__groups__"] = {repr(k): int(v) for k, v in _GROUP_TILE_CACHE.items()}
    return t


def tile_timings() -> dict:
    """The tuner's isolated timings per signature ({repr(signature): [[ms, tile], ...]}): what ``refine_tiles_in_step``
    draws its candidates from -- shipped to every rank with the tile table so that all ranks try the same tiles."""
    return {repr(k): [[float(ms), int(t)] for ms, t in v] for k, v in _TILE_TIMINGS.items()}


def load_tile_timings(table: dict) -> None:
    import ast
    _TILE_TIMINGS.clear()
    _TILE_CALLS.clear()
    for k, v in table.items():
        key = ast.literal_eval(k) if isinstance(k, str) else tuple(k)
        _TILE_TIMINGS[key] = [(float(ms), int(t)) for ms, t in v]
        _TILE_CALLS[key] = [0]


def load_tile_table(table: dict) -> None:
    """Takes a table of ``tile_table()``.  A table written with another signature layout would load and never match a
    launch -- silently, every GEMM then running untuned: it is refused with a warning instead."""
    import ast
    import warnings
    fmt = table.get("__format__")
    if fmt != TILE_KEY_FORMAT:
        warnings.warn(f"fs2hip: tile table in signature format {fmt!r}, this build uses {TILE_KEY_FORMAT}: ignored "
                      "(re-tune, or delete the file FS2_GEMM_TILE_CACHE names)")
        return
    for k, v in table.items():
        if k == "__format__":
            continue
        if k == "__groups__":
            for gk, gv in v.items():
                _GROUP_TILE_CACHE[ast.literal_eval(gk)] = int(gv)
            continue
        _TILE_CACHE[ast.literal_eval(k) if isinstance(k, str) else tuple(k)] = int(v)
    TILE_GEN[0] += 1


def save_tile_cache(path) -> None:
    import json
    Path(path).write_text(json.dumps(tile_table(), indent=0, sort_keys=True))


def load_tile_cache(path) -> bool:
    import json
    pth = Path(path)
    if not pth.exists():
        return False
    load_tile_table(json.loads(pth.read_text()))
    return True


_TILE_CACHE_ENV_LOADED = False


def _tune_tile(a) -> int:
    """Workgroup tile for this launch.  Deterministic modes: ``FS2_GEMM_TUNE=0`` (or ``hip.GEMM_TUNE = False``) always
    takes the library's shape heuristic (tile 0) -- what the parity tests run on, so that a result does not depend
    on which tile won a timing race in that process; ``FS2_GEMM_TILE_CACHE=<file.json>`` replays a saved table."""
    global _TILE_CACHE_ENV_LOADED
    if not _TILE_CACHE_ENV_LOADED:
        _TILE_CACHE_ENV_LOADED = True
        if os.environ.get("FS2_GEMM_TILE_CACHE"):
            load_tile_cache(os.environ["FS2_GEMM_TILE_CACHE"])
    if not GEMM_TUNE and not _TILE_CACHE:
        return 0
    key = _tile_key(a)
    t = _TILE_CACHE.get(key)
    if t is not None:
        c = _TILE_CALLS.get(key)
        if c is not None:
            c[0] += 1
        return t
    if not GEMM_TUNE or _REC is not None or torch.cuda.is_current_stream_capturing():
        return 0
    L, s = real_lib(), _stream()
    best, best_ms = 0, float("inf")
    timings = []
    # the BK=16 core has no bf16 instance (it runs in fp32): last resort in bf16-mixed mode
    order = sorted(GEMM_TILES, key=lambda t: (t < 4, t)) if a.operand_bf16 else GEMM_TILES
    if a.operand_bf16 == 4:
        order = GEMM_TILES_B
    for tile in order:
        if a.operand_bf16 and tile < 4 and best:
            break
        a.tile = tile
        rc = L.fs2hip_gemm(C.byref(a), s)  # warm; -22 = this core/tile does not take the shape
        if rc == -22:
            continue
        _ok(rc, "gemm")
        ms = float("inf")
        for _ in range(2):  # best of two rounds of four launches: single rounds are noisy at 10-100 us per launch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                _ok(L.fs2hip_gemm(C.byref(a), s), "gemm")
            e1.record()
            e1.synchronize()
            ms = min(ms, e0.elapsed_time(e1))
        timings.append((ms, tile))
        if ms < best_ms:
            best, best_ms = tile, ms
    _TILE_CACHE[key] = best
    TILE_GEN[0] += 1
    if timings:
        _TILE_TIMINGS[key] = sorted(timings)
        _TILE_CALLS[key] = [1]
    return best


#: (tile, exact geometry) pairs the library has refused: a cached tile is right for its bucket of quantised shapes, and an
#: exact shape of the bucket that it does not admit must not pay a failing call on every launch
_TILE_REFUSED = set()


def _refusal_key(a):
    return (a.tile, a.Mc, a.Nc, a.R, a.T, a.taps, a.lda, a.ldb, a.ldc, a.epi, a.io_bf16, a.operand_bf16, a.splitk,
            bool(a.colsum), (a.C or 0) % 16)


def _launch_gemm(a):
    if a.tile != 0 and _TILE_REFUSED and _refusal_key(a) in _TILE_REFUSED:
        a.tile = 0
    rc = lib().fs2hip_gemm(C.byref(a), _stream())
    if rc == -22 and a.tile != 0:
        # a cached / replayed tile that this exact geometry does not admit (the key quantises long extents): the
        # library chooses for THIS launch and for every later launch of this exact geometry (noted in _TILE_REFUSED); the
        # cache entry stays -- it is right for the other shapes of its bucket, and dropping it made two alternating
        # shapes re-run the tuner every step
        _TILE_REFUSED.add(_refusal_key(a))
        a.tile = 0
        rc = lib().fs2hip_gemm(C.byref(a), _stream())
    _ok(rc, "gemm")


def _gemm(_algorithmic=True, **kw):
    a = GemmArgs()
    a.taps, a.alpha, a.res_scale, a.splitk = 1, 1.0, 1.0, 1
    for k, v in kw.items():
        setattr(a, k, v)
    if not a.workspace:  # scratch for the split-tail tiles (13-15): at most one slab of partial sums per workgroup slot
        ws = _workspace(HYBRID_WS_FLOATS, _current_device())
        a.workspace, a.workspace_floats = _p(ws), ws.numel()
    if not a.operand_bf16:  # (3 / 4 = bf16 operands in memory: set by the caller that passes bf16 tensors)
        a.operand_bf16 = int(GEMM_BF16) if _algorithmic else 0
    a.tile = _tune_tile(a)
    if _GROUP is not None:  # ``gemm_group()``: launched with the others when the group closes
        _GROUP.entries.append((a, _algorithmic, _stream()))
        return
    if GEMM_PROFILE is None or not _algorithmic:  # (the one-hot embedding GEMM's flops are not algorithmic)
        _launch_gemm(a)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _launch_gemm(a)
    e1.record()
    _profile_gemm(a, e0, e1)


def _gemm_cost(a):
    """(algorithmic flops, algorithmic bytes: every operand element read once, every output element written once)"""
    ntap = a.taps if a.shift_operand == 1 else 1
    ra = a.R // a.taps if (a.taps > 1 and a.shift_operand == 0) else a.R
    esz = 2.0 if a.operand_bf16 >= 3 else 4.0
    osz = 2.0 if (a.io_bf16 & 1) else 4.0
    nbytes = (esz * (a.Mc * ra + a.Nc * a.R) + osz * a.Mc * a.Nc * ntap + 4.0 * (a.Mc * a.Nc if a.resid else 0)
              + (2.0 if (a.io_bf16 & 2) else 4.0) * (a.Mc * a.Nc if a.aux else 0) + osz * (a.Mc * a.Nc if a.out_pre else 0))
    return 2.0 * a.Mc * a.Nc * a.R * ntap, nbytes


def _profile_gemm(a, e0, e1, members=None):
    """One GEMM_PROFILE entry per LAUNCH; a grouped launch carries its members' summed cost under the first member's shape
    and a trailing ("grouped", n) in its kind."""
    ntap = a.taps if a.shift_operand == 1 else 1
    flops, nbytes = _gemm_cost(a)
    kind = (a.a_kcontig, a.b_kcontig, a.taps, a.shift_operand, a.splitk, a.epi, a.tile)
    if members:
        cost = [_gemm_cost(m) for m in members]
        flops, nbytes = sum(c[0] for c in cost), sum(c[1] for c in cost)
        kind += ("grouped", len(members))
    GEMM_PROFILE.append((e0, e1, flops, a.Mc, a.Nc, a.R * ntap, nbytes, kind))


# ---- grouped launches -----------------------------------------------------------------------------------------------
#: FS2_GEMM_GROUP=0 (measurement aid): ``gemm_group()`` sections launch their GEMMs one by one, as before round 5
GEMM_GROUP = os.environ.get("FS2_GEMM_GROUP", "1") != "0"
GROUP_MAX = 8  # FS2_GEMM_GROUP_MAX (include/fs2hip.h)
#: tiles the grouped kernels are built for, first = default: fp32 core / bf16-storage core (csrc/gemm2.hip, gemm_bf16.hip)
GROUP_TILES = {0: (7, 8), 4: (23, 26, 22)}
_GROUP = None
_GROUP_TILE_CACHE = {}
#: how many grouped launches / members went out, and how many members fell back to launches of their own (tests, logs)
GROUP_STATS = {"launches": 0, "members": 0, "single": 0}


class gemm_group:
    """Context manager: the GEMMs the enclosed wrapper calls (``linear_fwd``, ``linear_bwd_data``, ``linear_bwd_weight``)
    would launch are collected and go out together when the section closes -- members that can share a kernel instance
    (``fs2hip_gemm_grouped``: same operand orientations and storage type, no conv taps, no epilogue dropout) in ONE launch
    per up to eight of them, the rest one by one, in call order.  The enclosed calls must be independent of one another
    (no result of one is read by another inside the section), on one stream, and must not need their result before the
    section closes; everything else they do (allocations, deferred second-stage sums) happens at the call.  Each member
    computes exactly what its own launch would: same tile code on the member's own arguments."""

    def __enter__(self):
        global _GROUP
        self.entries, self.after, self.outer = [], [], True
        if GEMM_GROUP and _GROUP is None:
            _GROUP = self
        else:  # switched off, or inside another group (the outer one collects)
            self.outer = False
        return self

    def __exit__(self, et, ev, tb):
        global _GROUP
        if not self.outer:
            return False
        _GROUP = None
        if et is None:
            _launch_group(self.entries)
            for fn in self.after:  # (a wrapper's own follow-up launch: the split-K finish of a direct caller)
                fn()
        return False


def _group_class(a):
    """Members with the same value can share a grouped launch (None: never grouped)."""
    if a.taps > 1 or a.counters or a.drop_p > 0 or a.operand_bf16 not in GROUP_TILES:
        return None
    return (a.operand_bf16, bool(a.a_kcontig), bool(a.b_kcontig))


def _group_tile(arr, n, members, stream):
    """Tile of a grouped launch: the members' common tuned tile when the grouped kernels carry it, else the best of the
    grouped tiles by the tuner's own clock (timed once per combination of member signatures), else the default."""
    tiles = GROUP_TILES[members[0].operand_bf16]
    if not GEMM_TUNE and not _TILE_CACHE:
        return 0
    key = tuple(_tile_key(m) for m in members)
    t = _GROUP_TILE_CACHE.get(key)
    if t is not None:
        return t
    if not GEMM_TUNE or _REC is not None or torch.cuda.is_current_stream_capturing():
        common = {m.tile for m in members}
        return common.pop() if len(common) == 1 and members[0].tile in tiles else 0
    L, best, best_ms = real_lib(), 0, float("inf")
    for tile in tiles:
        arr[0].tile = tile
        if L.fs2hip_gemm_grouped(arr, n, stream) != 0:
            continue
        ms = float("inf")
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                L.fs2hip_gemm_grouped(arr, n, stream)
            e1.record()
            e1.synchronize()
            ms = min(ms, e0.elapsed_time(e1))
        if ms < best_ms:
            best, best_ms = tile, ms
    _GROUP_TILE_CACHE[key] = best
    TILE_GEN[0] += 1
    return best


def _launch_one(a, algorithmic):
    GROUP_STATS["single"] += 1
    if GEMM_PROFILE is None or not algorithmic:
        _launch_gemm(a)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _launch_gemm(a)
    e1.record()
    _profile_gemm(a, e0, e1)


def _launch_group(entries):
    if not entries:
        return
    stream = _stream()
    _req(all(st == stream for _, _, st in entries), "gemm_group: the enclosed calls ran on different streams")
    # partition in call order: runs of one class, at most GROUP_MAX each; members that cannot be grouped go alone
    classes = {}
    order = []
    for a, alg, _ in entries:
        c = _group_class(a)
        if c is None:
            order.append([(a, alg)])
            continue
        run = classes.get(c)
        if run is None or len(run) >= GROUP_MAX:
            run = classes[c] = []
            order.append(run)
        run.append((a, alg))
    for run in order:
        if len(run) == 1:
            _launch_one(*run[0])
            continue
        n = len(run)
        arr = (GemmArgs * n)()
        for i, (a, _) in enumerate(run):
            C.memmove(C.byref(arr[i]), C.byref(a), C.sizeof(GemmArgs))
        members = [a for a, _ in run]
        arr[0].tile = _group_tile(arr, n, members, stream)
        prof = GEMM_PROFILE is not None and all(alg for _, alg in run)
        if prof:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = lib().fs2hip_gemm_grouped(arr, n, stream)
        if rc == -22:  # the members cannot share a launch after all (a shape the grouped tiles refuse): one by one
            for a, alg in run:
                _launch_one(a, alg)
            continue
        _ok(rc, "gemm_grouped")
        GROUP_STATS["launches"] += 1
        GROUP_STATS["members"] += n
        if prof:
            e1.record()
            members[0].tile = arr[0].tile
            _profile_gemm(members[0], e0, e1, members)


def linear_fwd(x, w, bias=None, *, epi=EPI_STORE, act=None, resid=None, res_scale=1.0, out_pre=None,
               drop: Drop = NO_DROP, taps=1, T=0, out=None, out_dtype=torch.float32):
    """y[M, N] = epi(x[M, K*] @ w^T + bias).  ``w`` is [N, K] (taps == 1) or
    [taps, N, Kper] for a k-tap convolution over time (rows of x are (b, t), 'same' padding).
    bf16 ``x`` and ``w`` (operand storage, ``operand_bf16 = 4``): the result is ``out_dtype`` (fp32 or bf16; ``out_pre``
    has the same type); ``bias`` and ``resid`` stay fp32."""
    stored = x.dtype == torch.bfloat16
    _chk(x, x.dtype if stored else torch.float32, "x"); _chk(w, x.dtype if stored else torch.float32, "w")
    M, Kper = _rows(x), x.shape[-1]
    _req(not stored or (Kper % 8 == 0 and (taps == 1 or Kper % 64 == 0)), "linear_fwd: bf16 rows must be whole 16-byte pieces")
    _req(stored or out_dtype == torch.float32, "linear_fwd: bf16 results need bf16 operands")
    if taps == 1:
        _req(w.dim() == 2 and w.shape[1] == Kper, f"linear_fwd: x has {Kper} columns, w is {tuple(w.shape)}")
        N = w.shape[0]
    else:
        _req(w.dim() == 3 and w.shape[0] == taps and w.shape[2] == Kper and T > 0 and M % T == 0,
             "linear_fwd: conv weight must be [taps, N, Kper] and rows a multiple of T")
        N = w.shape[1]
    if out is None:
        out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=out_dtype)
    _chk(out, out.dtype if stored else torch.float32, name="out")
    _req(_rows(out) == M and out.shape[-1] == N, "linear_fwd: bad output shape")
    obf = out.dtype == torch.bfloat16
    kw = dict(A=_p(x), B=_p(w), C=_p(out), Mc=M, Nc=N, R=Kper * taps, lda=Kper, ldb=Kper, ldc=N,
              a_kcontig=1, b_kcontig=1, taps=taps, T=T, tap_mul=1, tap_add=-((taps - 1) // 2), shift_operand=0,
              b_tap_stride=N * Kper, epi=epi, act=_ACT[act], drop_p=drop.p, drop_seed=drop.seed,
              drop_step=drop.step_ptr, operand_bf16=4 if stored else 0, io_bf16=1 if obf else 0)
    if bias is not None:
        _chk(bias, name="bias")
        _req(bias.numel() == N, "linear_fwd: bias size")
        kw["bias"] = _p(bias)
    if epi == EPI_RESID:
        _chk(resid, name="resid")
        _req(resid.shape == out.shape, "linear_fwd: residual shape")
        kw.update(resid=_p(resid), ldr=N, res_scale=float(res_scale))
    if out_pre is not None:
        _chk(out_pre, out.dtype, name="out_pre")
        _req(out_pre.shape == out.shape, "linear_fwd: out_pre shape")
        kw.update(out_pre=_p(out_pre), ldpre=N)
    _gemm(**kw)
    return out


def linear_bwd_data(dy, w, *, epi=EPI_STORE, act=None, aux=None, alpha=1.0, drop: Drop = NO_DROP,
                    taps=1, T=0, out=None, out_dtype=torch.float32, wt=None):
    """dx[M, K] = epi(alpha * dy[M, N] @ w) with w [N, K] (or [taps, N, Kper], transposed conv).
    bf16 ``dy`` and ``w`` (operand storage): the weight is read as it is stored (reduction-major operand of the
    bf16 core); ``aux`` may be fp32 or bf16, the result is ``out_dtype``.
    ``wt`` (bf16 [K, N] = w transposed, from ``ParamStore.pbt``): the product runs in the forward orientation -- both
    operands k-contiguous -- which the weights-stationary streaming kernel takes for N = 256 (csrc/gemm_ws.hip)."""
    if wt is not None and wt.dtype == dy.dtype and taps == 1 and (dy.dtype == torch.bfloat16 or GEMM_BF16 == 0):
        return _bwd_data_transposed(dy, wt, epi, act, aux, alpha, drop, out, out_dtype)
    stored = dy.dtype == torch.bfloat16
    _chk(dy, dy.dtype if stored else torch.float32, "dy"); _chk(w, dy.dtype if stored else torch.float32, "w")
    _req(stored or out_dtype == torch.float32, "linear_bwd_data: bf16 results need bf16 operands")
    M, N = _rows(dy), dy.shape[-1]
    if taps == 1:
        _req(w.dim() == 2 and w.shape[0] == N, "linear_bwd_data: dy columns != w rows")
        K = w.shape[1]
    else:
        _req(w.dim() == 3 and w.shape[0] == taps and w.shape[1] == N and T > 0 and M % T == 0,
             "linear_bwd_data: conv weight must be [taps, N, Kper]")
        K = w.shape[2]
    _req(not stored or (N % 8 == 0 and K % 8 == 0 and (taps == 1 or N % 64 == 0)),
         "linear_bwd_data: bf16 operands need N and K multiples of 8 (tap widths multiples of 64)")
    if out is None:
        out = torch.empty(*dy.shape[:-1], K, device=dy.device, dtype=out_dtype)
    _chk(out, out.dtype if stored else torch.float32, name="out")
    _req(_rows(out) == M and out.shape[-1] == K, "linear_bwd_data: bad output shape")
    io = 1 if out.dtype == torch.bfloat16 else 0
    kw = dict(A=_p(dy), B=_p(w), C=_p(out), Mc=M, Nc=K, R=N * taps, lda=N, ldb=K, ldc=K,
              a_kcontig=1, b_kcontig=0, taps=taps, T=T, tap_mul=-1, tap_add=(taps - 1) // 2, shift_operand=0,
              b_tap_stride=N * K, epi=epi, act=_ACT[act], alpha=float(alpha),
              drop_p=drop.p, drop_seed=drop.seed, drop_step=drop.step_ptr)
    if epi == EPI_DACT:
        _req(aux is not None, "linear_bwd_data: the act' epilogue needs aux")
        _chk(aux, torch.bfloat16 if (stored and aux.dtype == torch.bfloat16) else torch.float32, name="aux")
        _req(aux.shape == out.shape, "linear_bwd_data: aux shape")
        kw.update(aux=_p(aux), ldaux=K)
        if aux.dtype == torch.bfloat16:
            io |= 2
    if stored:
        kw.update(operand_bf16=4, io_bf16=io)
    _gemm(**kw)
    return out


def _bwd_data_transposed(dy, wt, epi, act, aux, alpha, drop, out, out_dtype):
    stored = dy.dtype == torch.bfloat16
    _chk(dy, dy.dtype if stored else torch.float32, "dy"); _chk(wt, dy.dtype, "wt")
    M, N = _rows(dy), dy.shape[-1]
    _req(wt.dim() == 2 and wt.shape[1] == N and N % 8 == 0, "linear_bwd_data: wt must be [K, N] with N a multiple of 8")
    _req(stored or out_dtype == torch.float32, "linear_bwd_data: bf16 results need bf16 operands")
    K = wt.shape[0]
    if out is None:
        out = torch.empty(*dy.shape[:-1], K, device=dy.device, dtype=out_dtype)
    _chk(out, out.dtype, name="out")
    _req(_rows(out) == M and out.shape[-1] == K, "linear_bwd_data: bad output shape")
    io = 1 if out.dtype == torch.bfloat16 else 0
    kw = dict(A=_p(dy), B=_p(wt), C=_p(out), Mc=M, Nc=K, R=N, lda=N, ldb=N, ldc=K, a_kcontig=1, b_kcontig=1, taps=1, T=0,
              tap_mul=1, tap_add=0, shift_operand=0, epi=epi, act=_ACT[act], alpha=float(alpha),
              drop_p=drop.p, drop_seed=drop.seed, drop_step=drop.step_ptr)
    if epi == EPI_DACT:
        _req(aux is not None, "linear_bwd_data: the act' epilogue needs aux")
        _chk(aux, torch.bfloat16 if aux.dtype == torch.bfloat16 else torch.float32, name="aux")
        _req(aux.shape == out.shape, "linear_bwd_data: aux shape")
        kw.update(aux=_p(aux), ldaux=K)
        if aux.dtype == torch.bfloat16:
            _req(stored, "linear_bwd_data: a bf16 aux needs bf16 operands")
            io |= 2
    if stored:
        kw.update(operand_bf16=4, io_bf16=io)
    _gemm(**kw)
    return out


def cast_bf16(x, out=None):
    """bf16 copy (round to nearest even) of an fp32 tensor whose size is a multiple of 8."""
    _chk(x, name="x")
    _req(x.numel() % 8 == 0, "cast_bf16: size must be a multiple of 8")
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    _chk(out, torch.bfloat16, "out")
    _req(out.numel() == x.numel(), "cast_bf16: output size")
    _ok(lib().fs2hip_cast_bf16(_p(x), _p(out), x.numel(), _stream()), "cast_bf16")
    return out


def im2col_taps(x, B, T, taps, direction=1):
    """[B*T, C] (fp32 or bf16) -> bf16 [B*T, taps * C]: row (b, t) holds x[b, t + direction * (tap - (taps - 1) // 2)] for
    every tap, zeros outside the utterance -- a k-tap 'same' convolution becomes one plain GEMM (``matmul_kn``)."""
    xb = x.dtype == torch.bfloat16
    _chk(x, x.dtype if xb else torch.float32, "x")
    Cc = x.shape[-1]
    _req(_rows(x) == B * T and Cc % 8 == 0 and taps % 2 == 1 and direction in (1, -1), "im2col_taps: shape mismatch")
    out = torch.empty(B * T, taps * Cc, device=x.device, dtype=torch.bfloat16)
    _ok(lib().fs2hip_im2col_taps(_p(x), int(xb), _p(out), B, T, Cc, taps, direction, _stream()), "im2col_taps")
    return out


def matmul_kn(a, b_kn, bias=None, out_dtype=torch.float32):
    """out[M, N] = a[M, K] @ b_kn[K, N] (+ bias): bf16 operands, ``b_kn`` reduction-major as stored (the bf16 core's
    transposing reads; K and N multiples of 8, a reduction tail that is not a whole K-tile reads zeros)."""
    _chk(a, torch.bfloat16, "a"); _chk(b_kn, torch.bfloat16, "b_kn")
    M, K = _rows(a), a.shape[-1]
    _req(b_kn.dim() == 2 and b_kn.shape[0] == K and K % 8 == 0 and b_kn.shape[1] % 8 == 0, "matmul_kn: a is [M, K], b_kn [K, N]")
    N = b_kn.shape[1]
    out = torch.empty(*a.shape[:-1], N, device=a.device, dtype=out_dtype)
    kw = dict(A=_p(a), B=_p(b_kn), C=_p(out), Mc=M, Nc=N, R=K, lda=K, ldb=N, ldc=N, a_kcontig=1, b_kcontig=0, taps=1, T=0,
              tap_mul=1, tap_add=0, shift_operand=0, epi=EPI_STORE, operand_bf16=4,
              io_bf16=1 if out_dtype == torch.bfloat16 else 0)
    if bias is not None:
        _chk(bias, name="bias")
        _req(bias.numel() == N, "matmul_kn: bias size")
        kw["bias"] = _p(bias)
    _gemm(**kw)
    return out


def transpose_cast_bf16(w, out=None):
    """[R, C] fp32 -> [C, R] bf16 (R a multiple of 8): the weight as the data-gradient GEMM's k-contiguous operand;
    [taps, R, C] -> [taps, C, R] for a convolution weight (one launch)."""
    _chk(w, name="w")
    _req(w.dim() in (2, 3) and w.shape[-2] % 8 == 0, "transpose_cast_bf16: [R, C] or [taps, R, C] with R a multiple of 8")
    R, Cc = w.shape[-2:]
    batch = w.shape[0] if w.dim() == 3 else 1
    if out is None:
        out = torch.empty(*w.shape[:-2], Cc, R, device=w.device, dtype=torch.bfloat16)
    _chk(out, torch.bfloat16, "out")
    _req(out.numel() == w.numel(), "transpose_cast_bf16: output size")
    _ok(lib().fs2hip_transpose_cast_bf16(_p(w), R, Cc, Cc, _p(out), R, batch, _stream()), "transpose_cast_bf16")
    return out


TRANSPOSE_MAX_JOBS = 64


class TransposeJob(C.Structure):  # mirrors Fs2TransposeJob (include/fs2hip.h)
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("fp32_out", C.c_int),
                ("pad_", C.c_int)]


def transpose_cast_bf16_multi(pairs):
    """``pairs``: [(src fp32 [R, C], dst bf16 [C, R]), ...] -- every dst = src^T rounded to bf16, ``TRANSPOSE_MAX_JOBS``
    matrices per launch."""
    for i in range(0, len(pairs), TRANSPOSE_MAX_JOBS):
        batch = pairs[i:i + TRANSPOSE_MAX_JOBS]
        jobs = (TransposeJob * len(batch))()
        for j, (src, dst) in zip(jobs, batch):
            _chk(src, name="src"); _chk(dst, dst.dtype if dst.dtype == torch.float32 else torch.bfloat16, "dst")
            _req(src.dim() == 2 and tuple(dst.shape) == (src.shape[1], src.shape[0]), "transpose_cast_bf16_multi: dst must be src^T")
            j.src, j.dst, j.rows, j.cols = _p(src), _p(dst), src.shape[0], src.shape[1]
            j.fp32_out = 1 if dst.dtype == torch.float32 else 0
        _ok(lib().fs2hip_transpose_cast_bf16_multi(jobs, len(batch), _stream()), "transpose_cast_bf16_multi")


def pick_splitk(Mc: int, Nc: int, R: int, taps: int = 1) -> int:
    tiles = ((Mc + 127) // 128) * ((Nc + 63) // 64) * taps
    # at most 16 slices: beyond that the slab traffic and the finish outweigh the extra workgroups (tools/splitk_sweep.py,
    # 20736-row reduction: 256x256 output 67 us at 64 slices, 42 at 16; 80x256 57 -> 40; wider outputs unchanged)
    s = max(1, min(SPLITK_MAX, 512 // max(tiles, 1)))
    if taps > 1:
        # conv weight gradients: at least one reduction chunk per XCD -- the slices are ordered (split, tap), so the taps
        # of a chunk (same dY rows, X rows shifted by one) then share that XCD's L2 instead of every (tap, split) slice
        # fetching its own copy (PostNet 512x512x5: 481 -> 286 MB fetched, 118 -> 130 TFLOP/s)
        s = max(s, int(os.environ.get("FS2_CONV_DW_SPLITK", 8)))
    if taps == 1 and Mc * Nc <= 64 * 64 and R >= (1 << 17):
        # a tiny output over a very long reduction -- the GST reference encoder's first convolution (fs2/gst/model.py:
        # 103-139): 9 x 32 weights over B x H x W = 1.65 M rows -- is ONE tile: 16 slices are 16 workgroups on 256 CUs, each
        # walking 100 k rows (2.2 ms, the single longest launch of the configs[4] step).  Its slabs are a few KB, so the
        # 16-slice rule above (slab traffic) does not apply: the slice count follows the reduction, ~4096 rows a slice.
        s = max(s, min(512, R // 4096))
    while s > 1 and R // s < 256:
        s //= 2
    return s


def _linear_bwd_weight(dy, x, out, *, taps=1, T=0, n_valid=None, bias_grad=None):
    """dw[N, K] = dy[M, N]^T @ x[M, K]  (or dw[taps, N, Kper] for the k-tap conv).
    Written into ``out`` (a view of the flat gradient buffer).  ``n_valid``: only the first
    n_valid columns of dy produce output rows (dy's row length may be padded to a multiple of 4).
    ``bias_grad`` [N], when given, receives dy's column sums from the same launch (second stage at the next
    ``flush_grad_reductions()``): the separate column-sum pass over dy disappears.
    bf16 ``dy`` and ``x`` (operand storage): both are read as stored (reduction-major operands of the bf16 core)."""
    stored = dy.dtype == torch.bfloat16
    _chk(dy, dy.dtype if stored else torch.float32, name="dy"); _chk(x, dy.dtype if stored else torch.float32, name="x")
    _chk(out, name="out")
    M, lda, K = _rows(dy), dy.shape[-1], x.shape[-1]
    N = lda if n_valid is None else int(n_valid)
    _req(_rows(x) == M and N <= lda, "linear_bwd_weight: row mismatch")
    _req(out.numel() == taps * N * K, "linear_bwd_weight: bad gradient shape")
    _req(taps == 1 or (T > 0 and M % T == 0), "linear_bwd_weight: rows must be a multiple of T")
    _req(not stored or (lda % 8 == 0 and K % 8 == 0 and N % 8 == 0 and (taps == 1 or T >= 64)),
         "linear_bwd_weight: bf16 operands need row lengths that are multiples of 8 (convolutions: T >= 64)")
    S = pick_splitk(N, K, M, taps)
    kw = dict(A=_p(dy), B=_p(x), C=_p(out), Mc=N, Nc=K, R=M, lda=lda, ldb=K, ldc=K, a_kcontig=0, b_kcontig=0,
              taps=taps, T=T if taps > 1 else 0, tap_mul=1, tap_add=-((taps - 1) // 2),
              shift_operand=1 if taps > 1 else 0, c_tap_stride=N * K, splitk=S)
    if stored:
        kw.update(operand_bf16=4)
    if bias_grad is not None:
        _chk(bias_grad, name="bias_grad")
        _req(bias_grad.numel() == N and n_valid is None, "linear_bwd_weight: bias gradient size")
        part = torch.empty(S * N, device=dy.device, dtype=torch.float32)
        kw["colsum"] = _p(part)
        _defer_reduction(part, S, N, N, bias_grad, N, None)
    if S > 1:
        n = taps * N * K
        ws = _workspace(S * n, dy.device)
        kw["workspace"] = _p(ws)
        kw["workspace_floats"] = ws.numel()
        in_kernel = (not stored) and SPLITK_IN_KERNEL and taps * ((N + 63) // 64) * ((K + 63) // 64) <= SPLITK_COUNTERS
        if in_kernel:  # the last workgroup of every output tile sums the slabs: no second launch
            kw["counters"] = _p(splitk_counters(dy.device))
        defer = _DEFER_SLABS and not in_kernel
        if defer or _GROUP is not None:
            # a buffer of its own: summed with every other pending split at the next flush_grad_reductions() / not shared
            # with the other members of a ``gemm_group`` (their launches come before any of their finishes)
            ws = torch.empty(S * n, device=dy.device, dtype=torch.float32)
            kw["workspace"] = _p(ws)
        _gemm(_algorithmic=n_valid is None, **kw)
        if defer:
            _PENDING_SLABS.append((ws, out, n, S))
        elif not in_kernel:
            def finish(ws=ws):
                _ok(lib().fs2hip_reduce_slabs(_p(ws), _p(out), n, S, n, _stream()), "reduce_slabs")
            if _GROUP is not None:
                _GROUP.after.append(finish)  # behind the group's launches
            else:
                finish()
    else:
        _gemm(_algorithmic=n_valid is None, **kw)
    return out


#: weight-gradient GEMMs held back for a later window of the backward pass (``hold_weight_gradients``): a list while on
_HELD_WGRADS = None


def linear_bwd_weight(dy, x, out, *, taps=1, T=0, n_valid=None, bias_grad=None):
    """dW = dy^T x (see ``_linear_bwd_weight``).  While ``hold_weight_gradients(True)`` is in force the launch is only
    noted -- with its operands, which stay alive -- and enqueued by ``release_weight_gradients``."""
    if _HELD_WGRADS is not None:
        _HELD_WGRADS.append((dy, x, out, dict(taps=taps, T=T, n_valid=n_valid, bias_grad=bias_grad)))
        return out
    return _linear_bwd_weight(dy, x, out, taps=taps, T=T, n_valid=n_valid, bias_grad=bias_grad)


def hold_weight_gradients(on: bool):
    """A weight gradient depends only on tensors that exist once its layer's data gradient has run, and nothing reads it
    before the optimizer: WHEN it runs is free.  Holding the decoder's and PostNet's weight-gradient GEMMs back until
    the encoder's backward pass puts their large, matrix-pipe-bound workgroups beside the encoder's small kernels (which
    leave most of the chip idle) instead of beside the decoder's own large GEMMs (where two matrix-bound kernels only
    slow each other down).  Returns what was still held (a caller that switches holding off must release first)."""
    global _HELD_WGRADS
    left = _HELD_WGRADS or []
    _HELD_WGRADS = [] if on else None
    return left


def holding_weight_gradients() -> bool:
    return _HELD_WGRADS is not None


def held_weight_gradients() -> int:
    return len(_HELD_WGRADS) if _HELD_WGRADS is not None else 0


def release_weight_gradients(n=None):
    """Enqueues the oldest ``n`` held weight-gradient GEMMs (all of them when None) on the current stream; holding stays on
    for later calls.  Returns the operand tensors (a caller on a side stream keeps them until the join)."""
    global _HELD_WGRADS
    if not _HELD_WGRADS:
        return []
    jobs = _HELD_WGRADS[:n] if n is not None else list(_HELD_WGRADS)
    del _HELD_WGRADS[:len(jobs)]
    held, _HELD_WGRADS = _HELD_WGRADS, None   # (the launches below must run, not be noted again)
    used = []
    try:
        for dy, x, out, kw in jobs:
            _linear_bwd_weight(dy, x, out, **kw)
            used += [dy, x]
    finally:
        _HELD_WGRADS = held
    return used


def colsum(x, out):
    """out[N] = sum over rows of x[M, N] (bias gradients)."""
    _chk(x, name="x"); _chk(out, name="out")
    M, N = _rows(x), x.shape[-1]
    _req(out.numel() == N, "colsum: bad output size")
    gy = lib().fs2hip_colsum_rows(M)
    ws = _workspace(gy * N, x.device)
    _ok(lib().fs2hip_colsum(_p(x), N, M, N, _p(ws), _p(out), _stream()), "colsum")
    return out


# ---- deferred second stage of parameter-gradient reductions ----------------------------------------------------
#: bias and LayerNorm parameter gradients are only read by the optimizer / the data-parallel bucket exchange, so
#: their partial sums are kept (each in its own buffer) and finished FS2_REDUCE_MAX_JOBS at a time by one launch:
#: ``flush_grad_reductions()`` -- called by ``FastSpeech2.backward`` before every bucket hand-off and at its end.
REDUCE_MAX_JOBS = 48
_PENDING_REDUCTIONS = []  # (partial tensor, rows, n, stride, out0, n0, out1)


class ReduceJob(C.Structure):  # mirrors Fs2ReduceJob (include/fs2hip.h)
    _fields_ = [("src", C.c_void_p), ("out0", C.c_void_p), ("out1", C.c_void_p), ("stride", C.c_longlong),
                ("rows", C.c_int), ("n", C.c_int), ("n0", C.c_int), ("pad_", C.c_int)]


class SlabJob(C.Structure):  # mirrors Fs2SlabJob (include/fs2hip.h)
    _fields_ = [("slabs", C.c_void_p), ("out", C.c_void_p), ("n", C.c_longlong), ("stride", C.c_longlong),
                ("nslabs", C.c_int), ("vec", C.c_int)]


#: split-K slab sets of weight-gradient GEMMs waiting for their sum.  Only while ``defer_slab_reductions(True)`` is in
#: force (FastSpeech2.backward): direct callers of ``linear_bwd_weight`` get a finished gradient as before.
_PENDING_SLABS = []
_DEFER_SLABS = False


def defer_slab_reductions(on: bool) -> bool:
    """While on, ``linear_bwd_weight`` leaves its split-K slabs unsummed until ``flush_grad_reductions()`` (one launch
    for up to ``REDUCE_MAX_JOBS`` weight gradients instead of one each).  Returns the previous setting."""
    global _DEFER_SLABS
    prev, _DEFER_SLABS = _DEFER_SLABS, bool(on) and os.environ.get("FS2_DEFER_SLABS", "1") != "0"
    return prev


def _defer_reduction(partial, rows, n, stride, out0, n0, out1):
    # never flushed from here: the caller may be on the side stream (modules.Env.side) while other partial sums
    # were produced on the main one; flush_grad_reductions() is called on the main stream after the join
    _PENDING_REDUCTIONS.append((partial, rows, n, stride, out0, n0, out1))


def drop_pending_reductions():
    """Forgets every pending second-stage sum (a backward pass that raised half way)."""
    _PENDING_REDUCTIONS.clear()
    _PENDING_SLABS.clear()


def flush_grad_reductions():
    """Finishes every pending sum on the current stream.  Returns the partial-sum buffers it read: a caller that
    flushes on a stream other than the one they were allocated on must keep them alive until the streams join."""
    used = []
    while _PENDING_REDUCTIONS:
        batch = _PENDING_REDUCTIONS[:REDUCE_MAX_JOBS]
        jobs = (ReduceJob * len(batch))()
        for j, (partial, rows, n, stride, out0, n0, out1) in zip(jobs, batch):
            j.src, j.out0, j.out1 = _p(partial), _p(out0), (_p(out1) if out1 is not None else None)
            j.stride, j.rows, j.n, j.n0 = stride, rows, n, n0
            used.append(partial)
        _ok(lib().fs2hip_reduce_rows_multi(jobs, len(batch), _stream()), "reduce_rows_multi")
        del _PENDING_REDUCTIONS[:len(batch)]
    while _PENDING_SLABS:
        batch = _PENDING_SLABS[:REDUCE_MAX_JOBS]
        jobs = (SlabJob * len(batch))()
        for j, (ws, out, n, S) in zip(jobs, batch):
            j.slabs, j.out, j.n, j.stride, j.nslabs, j.vec = _p(ws), _p(out), n, n, S, 0
            used.append(ws)
        _ok(lib().fs2hip_reduce_slabs_multi(jobs, len(batch), _stream()), "reduce_slabs_multi")
        del _PENDING_SLABS[:len(batch)]
    return used


def segment_colsum(x, out):
    """out[b, :] = sum over t of x[b, t, :] -- the gradient of per-utterance vectors that were broadcast over time
    (GST style vector, speaker / language embeddings).  One job per utterance, ``REDUCE_MAX_JOBS`` jobs per launch."""
    _chk(x, name="x"); _chk(out, name="out")
    B, T, D = x.shape
    _req(out.shape == (B, D), "segment_colsum: bad output shape")
    for b0 in range(0, B, REDUCE_MAX_JOBS):
        nb = min(REDUCE_MAX_JOBS, B - b0)
        jobs = (ReduceJob * nb)()
        for i, j in enumerate(jobs):
            j.src, j.out0, j.out1 = x.data_ptr() + 4 * (b0 + i) * T * D, out.data_ptr() + 4 * (b0 + i) * D, None
            j.stride, j.rows, j.n, j.n0 = D, T, D, D
        _ok(lib().fs2hip_reduce_rows_multi(jobs, nb, _stream()), "reduce_rows_multi")
    return out


def colsum_grad(x, out):
    """``colsum`` for a parameter gradient: first stage now, second stage at the next ``flush_grad_reductions()``."""
    _chk(x, name="x"); _chk(out, name="out")
    M, N = _rows(x), x.shape[-1]
    _req(out.numel() == N, "colsum: bad output size")
    gy = lib().fs2hip_colsum_rows(M)
    if gy == 1:  # nothing to batch: one partial row
        return colsum(x, out)
    part = torch.empty(gy * N, device=x.device, dtype=torch.float32)
    _ok(lib().fs2hip_colsum(_p(x), N, M, N, _p(part), None, _stream()), "colsum")
    _defer_reduction(part, gy, N, N, out, N, None)
    return out


# ------------------------------------------------------------------------------------------
# LayerNorm
# ------------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps=1e-5, out_dtype=torch.float32):
    """``out_dtype=torch.bfloat16``: y is written as bf16 only (the operand of a bf16-storage GEMM)."""
    _chk(x, name="x"); _chk(gamma, name="gamma"); _chk(beta, name="beta")
    M, Cc = _rows(x), x.shape[-1]
    _req(gamma.numel() == Cc and beta.numel() == Cc, "layernorm_fwd: parameter size")
    y = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    mean = torch.empty(M, device=x.device, dtype=torch.float32)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    fn = lib().fs2hip_layernorm_fwd_b if out_dtype == torch.bfloat16 else lib().fs2hip_layernorm_fwd
    _ok(fn(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), M, Cc, eps, _stream()), "layernorm_fwd")
    return y, mean, rstd


def layernorm_fwd_drop(x, gamma, beta, drop: Drop, eps=1e-5):
    """dropout(LayerNorm(x)) in one launch (the variance predictors' layers); the mask is ``axpby``'s over the result."""
    _chk(x, name="x"); _chk(gamma, name="gamma"); _chk(beta, name="beta")
    M, Cc = _rows(x), x.shape[-1]
    _req(gamma.numel() == Cc and beta.numel() == Cc, "layernorm_fwd_drop: parameter size")
    y = torch.empty_like(x)
    mean = torch.empty(M, device=x.device, dtype=torch.float32)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    _ok(lib().fs2hip_layernorm_fwd_drop(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), M, Cc, eps, drop.p,
                                        drop.seed, drop.step_ptr, _stream()), "layernorm_fwd_drop")
    return y, mean, rstd


def layernorm_bwd_pred(dy, x, gamma, mean, rstd, dgamma, dbeta, drop: Drop, out_dtype=torch.float32):
    """relu'(x) * LayerNormBackward(dropmask * dy): the backward of a predictor layer's Dropout, LayerNorm and ReLU in one
    launch (x = the ReLU output).  dgamma / dbeta arrive at the next ``flush_grad_reductions()``.  ``out_dtype`` bf16: the
    result is the bf16 operand of the layer's weight- and data-gradient GEMMs (bf16 operand storage)."""
    for n, t in (("dy", dy), ("x", x), ("gamma", gamma), ("mean", mean), ("rstd", rstd), ("dgamma", dgamma), ("dbeta", dbeta)):
        _chk(t, name=n)
    M, Cc = _rows(x), x.shape[-1]
    _req(dy.shape == x.shape and mean.numel() == M and rstd.numel() == M and dgamma.numel() == Cc and dbeta.numel() == Cc,
         "layernorm_bwd_pred: shape mismatch")
    _req(out_dtype in (torch.float32, torch.bfloat16), "layernorm_bwd_pred: result is fp32 or bf16")
    dx = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    nblk = lib().fs2hip_layernorm_bwd_blocks(M)
    part = torch.empty(nblk * 2 * Cc, device=x.device, dtype=torch.float32)
    _ok(lib().fs2hip_layernorm_bwd_pred(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx),
                                        int(out_dtype == torch.bfloat16), _p(part), M, Cc, drop.p,
                                        drop.seed, drop.step_ptr, _stream()), "layernorm_bwd_pred")
    _defer_reduction(part, nblk, 2 * Cc, 2 * Cc, dgamma, Cc, dbeta)
    return dx


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dx_add=None, defer=False, dz_scale=None,
                  dz_drop: Drop = NO_DROP, dz_colsum=None, dz_dtype=torch.float32):
    """Returns dx (+ dx_add); writes dgamma/dbeta (``defer``: at the next ``flush_grad_reductions()``).
    With ``dz_scale``: returns (dx, dz), dz = dz_scale * dropmask(dz_drop) * dx, and ``dz_colsum`` receives dz's column
    sums at the next ``flush_grad_reductions()`` (like dgamma/dbeta, which are then always deferred).
    bf16 storage: ``dy`` may be bf16 (a data-gradient GEMM's result) and ``dz_dtype`` bf16 (the next GEMMs' operand)."""
    dyb = dy.dtype == torch.bfloat16
    _chk(dy, dy.dtype if dyb else torch.float32, "dy")
    for n, t in (("x", x), ("gamma", gamma), ("mean", mean), ("rstd", rstd), ("dgamma", dgamma), ("dbeta", dbeta)):
        _chk(t, name=n)
    M, Cc = _rows(x), x.shape[-1]
    _req(dy.shape == x.shape and mean.numel() == M and rstd.numel() == M and dgamma.numel() == Cc
         and dbeta.numel() == Cc, "layernorm_bwd: shape mismatch")
    if dx_add is not None:
        _chk(dx_add, name="dx_add")
        _req(dx_add.shape == x.shape, "layernorm_bwd: dx_add shape")
    dx = torch.empty_like(x)
    nblk = lib().fs2hip_layernorm_bwd_blocks(M)
    zb = dz_dtype == torch.bfloat16
    if dz_scale is not None:
        _chk(dz_colsum, name="dz_colsum")
        _req(dz_colsum.numel() == Cc, "layernorm_bwd: dz_colsum size")
        dz = torch.empty(x.shape, device=x.device, dtype=dz_dtype)
        part = torch.empty(nblk * 3 * Cc, device=x.device, dtype=torch.float32)
        if dyb or zb:
            _ok(lib().fs2hip_layernorm_bwd_x(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx_add), _p(dx), _p(dz),
                                             float(dz_scale), dz_drop.p, dz_drop.seed, dz_drop.step_ptr, _p(part), M, Cc,
                                             (1 if dyb else 0) | (2 if zb else 0), _stream()), "layernorm_bwd_x")
        else:
            _ok(lib().fs2hip_layernorm_bwd_dz(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx_add), _p(dx), _p(dz),
                                              float(dz_scale), dz_drop.p, dz_drop.seed, dz_drop.step_ptr, _p(part), M, Cc,
                                              _stream()), "layernorm_bwd_dz")
        _defer_reduction(part, nblk, 2 * Cc, 3 * Cc, dgamma, Cc, dbeta)
        _defer_reduction(part[2 * Cc:], nblk, Cc, 3 * Cc, dz_colsum, Cc, None)
        return dx, dz
    if dyb:
        part = torch.empty(nblk * 2 * Cc, device=x.device, dtype=torch.float32)
        _ok(lib().fs2hip_layernorm_bwd_x(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx_add), _p(dx), None, 0.0,
                                         0.0, 0, None, _p(part), M, Cc, 1, _stream()), "layernorm_bwd_x")
        _defer_reduction(part, nblk, 2 * Cc, 2 * Cc, dgamma, Cc, dbeta)
        return dx
    if defer and nblk > 1:
        part = torch.empty(nblk * 2 * Cc, device=x.device, dtype=torch.float32)
        _ok(lib().fs2hip_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx_add), _p(dx), _p(part),
                                       None, None, M, Cc, _stream()), "layernorm_bwd")
        _defer_reduction(part, nblk, 2 * Cc, 2 * Cc, dgamma, Cc, dbeta)
        return dx
    ws = _workspace(nblk * 2 * Cc, x.device)
    _ok(lib().fs2hip_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx_add), _p(dx), _p(ws),
                                   _p(dgamma), _p(dbeta), M, Cc, _stream()), "layernorm_bwd")
    return dx


# ------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------
#: FS2_ATTN_SCORES=0: the training forward does not keep its scores (the dK/dV kernel recomputes K.Q^T)
ATTN_SCORES = os.environ.get("FS2_ATTN_SCORES", "1") != "0"


def attention_scores_kept(HD: int) -> bool:
    """True when ``attention_fwd(save_scores=True)`` writes the scores out for ``attention_bwd(scores=...)``: exact fp32 or
    "32-split", head dims of the second-generation kernels, spilled-dS backward on."""
    return GEMM_BF16 in (0, 2) and ATTN_SPILL and ATTN_SCORES and bool(lib().fs2hip_attention_bwd_spill_supported(int(HD)))


def attention_fwd(qkv, lens, B, T, H, drop: Drop = NO_DROP, save_scores=False):
    """``save_scores`` (training forward): returns (o, lse, scores) -- scores [B, H, T, T rounded up to 32] for the backward
    pass, or None where ``attention_scores_kept`` says no."""
    _chk(qkv, name="qkv"); _chk(lens, torch.int32, "lens")
    D = qkv.shape[-1] // 3
    _req(_rows(qkv) == B * T and qkv.shape[-1] == 3 * D and D % H == 0 and lens.numel() == B,
         "attention_fwd: shape mismatch")
    o = torch.empty(B, T, D, device=qkv.device, dtype=torch.float32)
    lse = torch.empty(B, H, T, device=qkv.device, dtype=torch.float32)
    if save_scores:
        if not attention_scores_kept(D // H):
            o, lse = attention_fwd(qkv, lens, B, T, H, drop)
            return o, lse, None
        sc = torch.empty(B, H, T, (T + 31) // 32 * 32, device=qkv.device, dtype=torch.float32)
        _ok(lib().fs2hip_attention_fwd_s(_p(qkv), _p(lens), _p(o), _p(lse), _p(sc), sc.numel(), B, T, H, D // H, drop.p, drop.seed,
                                         drop.step_ptr, int(GEMM_BF16), _stream()), "attention_fwd_s")
        return o, lse, sc
    _ok(lib().fs2hip_attention_fwd(_p(qkv), _p(lens), _p(o), _p(lse), B, T, H, D // H, drop.p, drop.seed,
                                   drop.step_ptr, int(GEMM_BF16), _stream()), "attention_fwd")
    return o, lse


#: FS2_ATTN_SPILL=0: the fp32 attention backward recomputes S and dP in both gradient kernels (the round-2 structure)
ATTN_SPILL = os.environ.get("FS2_ATTN_SPILL", "1") != "0"
_SCRATCH = {}


def reserve_scratch(name, floats):
    """A float32 scratch buffer per (name, device, stream) that only grows: contents are dead between the launches of one
    call, and calls on one stream are ordered."""
    key = (name, _current_device(), _stream())
    t = _SCRATCH.get(key)
    if t is None or t.numel() < floats:
        # (a buffer first allocated while a hipGraph is captured would live in the graph's pool and then be used outside
        # it: size the scratch with one eager step before capturing -- ADVICE r4)
        _req(not torch.cuda.is_current_stream_capturing(),
             f"scratch buffer {name!r} must be sized by an eager step before a hipGraph capture")
        t = _SCRATCH[key] = torch.empty(int(floats), device="cuda", dtype=torch.float32)
    return t


def release_scratch():
    """Drops every grow-only scratch / workspace buffer of this process (the dS scratch of the attention backward is
    B*H*T*T floats: 0.9 GB at the configs[4] shape).  They are re-made on demand; ``FastSpeech2.move_to`` calls this, a
    caller that switches to much smaller batches may."""
    _SCRATCH.clear()
    _WS.clear()


def attention_bwd(qkv, lens, o, dout, lse, B, T, H, drop: Drop = NO_DROP, scores=None):
    for n, t in (("qkv", qkv), ("o", o), ("dout", dout), ("lse", lse)):
        _chk(t, name=n)
    _chk(lens, torch.int32, "lens")
    D = qkv.shape[-1] // 3
    _req(_rows(qkv) == B * T and o.numel() == B * T * D and dout.numel() == B * T * D and lse.numel() == B * H * T
         and lens.numel() == B, "attention_bwd: shape mismatch")
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(2 * lse.numel() + 4, device=lse.device, dtype=torch.float32)  # scratch: see fs2hip.h
    if GEMM_BF16 in (0, 2) and ATTN_SPILL and lib().fs2hip_attention_bwd_spill_supported(D // H):
        # exact fp32 and "32-split" (whose dK/dV kernel is the fp32 one): the dK/dV kernel writes dS out and dQ is a product
        # of its own on the fp32 MFMAs (5 products per block instead of 9)
        n = B * H * T * ((T + 31) // 32 * 32)
        ds = reserve_scratch("attn_ds", n)
        if scores is not None:
            _chk(scores, name="scores")
            _req(scores.numel() == n, "attention_bwd: scores must be [B, H, T, T rounded up to 32]")
            _ok(lib().fs2hip_attention_bwd_spill_s(_p(qkv), _p(lens), _p(o), _p(dout), _p(lse), _p(scores), _p(delta), _p(ds), n,
                                                   _p(dqkv), B, T, H, D // H, drop.p, drop.seed, drop.step_ptr, _stream()),
                "attention_bwd_spill_s")
            return dqkv
        _ok(lib().fs2hip_attention_bwd_spill(_p(qkv), _p(lens), _p(o), _p(dout), _p(lse), _p(delta), _p(ds), n, _p(dqkv), B, T, H,
                                             D // H, drop.p, drop.seed, drop.step_ptr, _stream()), "attention_bwd_spill")
        return dqkv
    _ok(lib().fs2hip_attention_bwd(_p(qkv), _p(lens), _p(o), _p(dout), _p(lse), _p(delta), _p(dqkv), B, T, H, D // H,
                                   drop.p, drop.seed, drop.step_ptr, int(GEMM_BF16), _stream()), "attention_bwd")
    return dqkv


def attention_b_supported(HD: int) -> bool:
    """True when the bf16-storage attention kernels take this head dimension."""
    return bool(lib().fs2hip_attention_b_supported(int(HD)))


def attention_fwd_b(qkv, lens, B, T, H, drop: Drop = NO_DROP):
    """Attention on bf16 tensors (precision "bf16-mixed" with bf16 activation storage): qkv bf16 [B*T][3D] -> o bf16,
    lse fp32.  Same dropout mask as ``attention_fwd`` for the same ``drop``."""
    _chk(qkv, torch.bfloat16, "qkv"); _chk(lens, torch.int32, "lens")
    D = qkv.shape[-1] // 3
    _req(_rows(qkv) == B * T and qkv.shape[-1] == 3 * D and D % H == 0 and lens.numel() == B,
         "attention_fwd_b: shape mismatch")
    _req(attention_b_supported(D // H), "attention_fwd_b: head dimension not supported by the bf16-storage kernels")
    o = torch.empty(B, T, D, device=qkv.device, dtype=torch.bfloat16)
    lse = torch.empty(B, H, T, device=qkv.device, dtype=torch.float32)
    _ok(lib().fs2hip_attention_fwd_b(_p(qkv), _p(lens), _p(o), _p(lse), B, T, H, D // H, drop.p, drop.seed,
                                     drop.step_ptr, _stream()), "attention_fwd_b")
    return o, lse


#: FS2_ATTN_SPILL_B=0: the bf16-storage attention backward recomputes S and dP in both gradient kernels
ATTN_SPILL_B = os.environ.get("FS2_ATTN_SPILL_B", "1") != "0"


def attention_bwd_b(qkv, lens, o, dout, lse, B, T, H, drop: Drop = NO_DROP):
    for n, t in (("qkv", qkv), ("o", o), ("dout", dout)):
        _chk(t, torch.bfloat16, n)
    _chk(lse, name="lse"); _chk(lens, torch.int32, "lens")
    D = qkv.shape[-1] // 3
    _req(_rows(qkv) == B * T and o.numel() == B * T * D and dout.numel() == B * T * D and lse.numel() == B * H * T
         and lens.numel() == B, "attention_bwd_b: shape mismatch")
    _req(attention_b_supported(D // H), "attention_bwd_b: head dimension not supported by the bf16-storage kernels")
    dqkv = torch.empty_like(qkv)
    aux = torch.empty(2 * lse.numel() + 4, device=lse.device, dtype=torch.float32)  # {lse', delta'} per row and head
    if ATTN_SPILL_B:
        n = B * H * T * ((T + 31) // 32 * 32)  # bf16 elements
        ds = reserve_scratch("attnb_ds", (n + 1) // 2)
        _ok(lib().fs2hip_attention_bwd_b_spill(_p(qkv), _p(lens), _p(o), _p(dout), _p(lse), _p(aux), _p(ds), n, _p(dqkv), B, T, H,
                                               D // H, drop.p, drop.seed, drop.step_ptr, _stream()), "attention_bwd_b_spill")
        return dqkv
    _ok(lib().fs2hip_attention_bwd_b(_p(qkv), _p(lens), _p(o), _p(dout), _p(lse), _p(aux), _p(dqkv), B, T, H, D // H,
                                     drop.p, drop.seed, drop.step_ptr, _stream()), "attention_bwd_b")
    return dqkv


# ------------------------------------------------------------------------------------------
# depthwise conv (+GLU, +BN statistics) and BatchNorm
# ------------------------------------------------------------------------------------------
class StatParts:
    """Per-part BatchNorm statistics: ``partial`` [nparts][2][C] = (mean, sum of squared deviations) of parts that
    tile ``count`` rows in groups of ``group_rows`` rows, each group in stripes of ``part_rows``."""
    __slots__ = ("partial", "nparts", "part_rows", "group_rows", "count")

    def __init__(self, partial, nparts, part_rows, group_rows, count):
        self.partial, self.nparts, self.part_rows, self.group_rows, self.count = partial, nparts, part_rows, group_rows, count


def dwconv_fwd(x, w, bias, B, T, *, glu=False, stats=False, out_dtype=None):
    """x [B*T, C or 2C] -> y [B, T, C]; w [K, C].  Returns (y, StatParts or None).  A bf16 ``x`` (GLU form; bf16
    activation storage) gives a bf16 ``y`` whose statistics are those of the rounded values.  ``out_dtype`` bf16 with an
    fp32 ``x`` (plain form, no statistics): the result is written as bf16 only -- a bf16-storage GEMM's operand."""
    xb = x.dtype == torch.bfloat16
    _chk(x, x.dtype if xb else torch.float32, "x"); _chk(w, name="w")
    _req(glu or not xb, "dwconv_fwd: bf16 tensors are taken in the GLU form only")
    yb_only = out_dtype == torch.bfloat16 and not xb
    _req(not yb_only or not (glu or stats), "dwconv_fwd: fp32 -> bf16 is the plain form without statistics")
    K, Cc = w.shape
    ldx = x.shape[-1]
    _req(_rows(x) == B * T and ldx == (2 * Cc if glu else Cc), "dwconv_fwd: shape mismatch")
    if bias is not None:
        _chk(bias, name="bias")
        _req(bias.numel() == Cc, "dwconv_fwd: bias size")
    y = torch.empty(B, T, Cc, device=x.device, dtype=torch.bfloat16 if yb_only else x.dtype)
    nparts = lib().fs2hip_dwconv_blocks(B, T)
    partial = torch.empty(nparts, 2, Cc, device=x.device, dtype=torch.float32) if stats else None
    _ok(lib().fs2hip_dwconv_fwd_b(_p(x), ldx, _p(w), _p(bias), _p(y), _p(partial), B, T, Cc, K, int(glu), int(stats),
                                  2 if yb_only else int(xb), _stream()), "dwconv_fwd")
    return y, (StatParts(partial, nparts, lib().fs2hip_dwconv_part_rows(), T, B * T) if stats else None)


def dwconv_bwd(dy, x, w, dw, dbias, B, T, *, glu=False, out_dtype=torch.float32):
    """Returns dx (layout of x; ``out_dtype`` fp32 or bf16); writes dw [K, C] and dbias [C].  ``dy`` and ``x`` may both
    be bf16 tensors (GLU form, bf16 result)."""
    xb = x.dtype == torch.bfloat16
    _chk(dy, x.dtype if xb else torch.float32, "dy"); _chk(x, x.dtype if xb else torch.float32, "x")
    _chk(w, name="w"); _chk(dw, name="dw")
    _req(not xb or (glu and out_dtype == torch.bfloat16), "dwconv_bwd: bf16 inputs need the GLU form and a bf16 result")
    K, Cc = w.shape
    ldx = x.shape[-1]
    _req(_rows(x) == B * T and ldx == (2 * Cc if glu else Cc) and dy.numel() == B * T * Cc and dw.numel() == K * Cc,
         "dwconv_bwd: shape mismatch")
    if dbias is not None:
        _chk(dbias, name="dbias")
        _req(dbias.numel() == Cc, "dwconv_bwd: dbias size")
    dx = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    nblk = lib().fs2hip_dwconv_blocks(B, T)
    stride = (K + 1) * Cc
    if _DEFER_SLABS and dbias is not None and stride <= 16384 and nblk >= 8:
        # inside FastSpeech2.backward: the weight / bias gradient's second stage joins the batched finish (the same
        # row-parallel sum in the same order as the launch it replaces; nothing reads dw / dbias before the flush)
        part = torch.empty(nblk * stride, device=x.device, dtype=torch.float32)
        _ok(lib().fs2hip_dwconv_bwd_b(_p(dy), _p(x), ldx, _p(w), _p(dx), int(out_dtype == torch.bfloat16) | (2 if xb else 0), _p(part),
                                      None, None, B, T, Cc, K, int(glu), _stream()), "dwconv_bwd")
        _defer_reduction(part, nblk, stride, stride, dw, K * Cc, dbias)
        return dx
    ws = _workspace(nblk * stride, x.device)
    _ok(lib().fs2hip_dwconv_bwd_b(_p(dy), _p(x), ldx, _p(w), _p(dx), int(out_dtype == torch.bfloat16) | (2 if xb else 0), _p(ws), _p(dw),
                                  _p(dbias), B, T, Cc, K, int(glu), _stream()), "dwconv_bwd")
    return dx


def colstats(y) -> StatParts:
    yb = y.dtype == torch.bfloat16
    _chk(y, y.dtype if yb else torch.float32, "y")
    M, Cc = _rows(y), y.shape[-1]
    nparts = lib().fs2hip_colstats_parts(M)
    partial = torch.empty(nparts, 2, Cc, device=y.device, dtype=torch.float32)
    _ok(lib().fs2hip_colstats_b(_p(y), M, Cc, _p(partial), int(yb), _stream()), "colstats")
    return StatParts(partial, nparts, lib().fs2hip_colstats_part_rows(M), M, M)


def bn_finalize(parts: Optional[StatParts], gamma, beta, running_mean, running_var, *, momentum=0.1, eps=1e-5,
                training=True):
    """Returns stats [4, C] = (scale, shift, mean, invstd); updates the running buffers when training.
    ``parts``: the batch statistics (``colstats`` / ``dwconv_fwd(stats=True)``); None in evaluation mode."""
    _chk(gamma, name="gamma"); _chk(beta, name="beta")
    Cc = gamma.numel()
    partial, nparts, part_rows, group_rows, count = None, 0, 0, 0, 0
    if training:
        _req(parts is not None, "bn_finalize: training mode needs the batch statistics")
        partial, nparts, part_rows, group_rows, count = (parts.partial, parts.nparts, parts.part_rows, parts.group_rows,
                                                          parts.count)
        _chk(partial, name="partial")
        _req(partial.numel() == nparts * 2 * Cc, "bn_finalize: partial size")
    if running_mean is not None:
        _chk(running_mean, name="running_mean"); _chk(running_var, name="running_var")
        _req(running_mean.numel() == Cc and running_var.numel() == Cc, "bn_finalize: running stats size")
    stats = torch.empty(4, Cc, device=gamma.device, dtype=torch.float32)
    _ok(lib().fs2hip_bn_finalize(_p(partial), nparts, count, part_rows, group_rows, _p(gamma), _p(beta),
                                 _p(running_mean), _p(running_var), momentum, eps, int(training), _p(stats), Cc,
                                 _stream()), "bn_finalize")
    return stats


def bn_act_fwd(y, stats, act=None, drop: Drop = NO_DROP, bf16_copy=False, bf16_only=False):
    """``bf16_copy``: returns (out, out as bf16) -- the form a bf16-operand GEMM (``linear_fwd`` on bf16 tensors) reads;
    ``bf16_only``: returns the bf16 form alone (no fp32 tensor is written).  ``y`` itself may be bf16."""
    yb = y.dtype == torch.bfloat16
    _chk(y, y.dtype if yb else torch.float32, "y"); _chk(stats, name="stats")
    M, Cc = _rows(y), y.shape[-1]
    _req(stats.numel() == 4 * Cc, "bn_act_fwd: stats size")
    out = None if bf16_only else torch.empty(y.shape, device=y.device, dtype=torch.float32)
    out_b = torch.empty(y.shape, device=y.device, dtype=torch.bfloat16) if (bf16_copy or bf16_only) else None
    _ok(lib().fs2hip_bn_act_fwd_b(_p(y), _p(stats), _p(out), _p(out_b), M, Cc, _ACT[act], drop.p, drop.seed,
                                  drop.step_ptr, int(yb), _stream()), "bn_act_fwd")
    if bf16_only:
        return out_b
    return (out, out_b) if bf16_copy else out


def bn_act_bwd(dout, y, stats, dgamma, dbeta, act=None, drop: Drop = NO_DROP, training=True, bf16_copy=False,
               bf16_only=False):
    yb, db = y.dtype == torch.bfloat16, dout.dtype == torch.bfloat16  # (bf16 activation storage: either may be bf16)
    _chk(dout, dout.dtype if db else torch.float32, "dout"); _chk(y, y.dtype if yb else torch.float32, "y")
    for n, t in (("stats", stats), ("dgamma", dgamma), ("dbeta", dbeta)):
        _chk(t, name=n)
    M, Cc = _rows(y), y.shape[-1]
    _req(dout.shape == y.shape and stats.numel() == 4 * Cc and dgamma.numel() == Cc and dbeta.numel() == Cc,
         "bn_act_bwd: shape mismatch")
    nparts = lib().fs2hip_colstats_parts(M)
    ws = _workspace(nparts * 2 * Cc + 2 * Cc, y.device)
    coef_ptr = ws.data_ptr() + 4 * nparts * 2 * Cc
    dy = None if bf16_only else torch.empty(y.shape, device=y.device, dtype=torch.float32)
    dy_b = torch.empty(y.shape, device=y.device, dtype=torch.bfloat16) if (bf16_copy or bf16_only) else None
    _ok(lib().fs2hip_bn_act_bwd_b(_p(dout), _p(y), _p(stats), _p(ws), coef_ptr, _p(dgamma), _p(dbeta), _p(dy), _p(dy_b),
                                  M, Cc, _ACT[act], drop.p, drop.seed, drop.step_ptr, int(training), int(yb) | (2 if db else 0), _stream()),
        "bn_act_bwd")
    if bf16_only:
        return dy_b
    return (dy, dy_b) if bf16_copy else dy


# ------------------------------------------------------------------------------------------
# positional table, embeddings, bucketize
# ------------------------------------------------------------------------------------------
def posenc_table(inv_freq, T, D):
    _chk(inv_freq, name="inv_freq")
    _req(inv_freq.numel() == D // 2, "posenc_table: inv_freq size")
    table = torch.empty(T, D, device=inv_freq.device, dtype=torch.float32)
    _ok(lib().fs2hip_posenc_table(_p(inv_freq), _p(table), T, D, _stream()), "posenc_table")
    return table


def add_posenc(x, table, lens, B, T):
    _chk(x, name="x"); _chk(table, name="table"); _chk(lens, torch.int32, "lens")
    D = x.shape[-1]
    _req(_rows(x) == B * T and table.shape[-1] == D and table.shape[0] >= T and lens.numel() == B,
         "add_posenc: shape mismatch")
    out = torch.empty_like(x)
    _ok(lib().fs2hip_add_posenc(_p(x), _p(table), _p(lens), _p(out), B, T, D, _stream()), "add_posenc")
    return out


def embedding_fwd(idx, W):
    _chk(idx, torch.int32, "idx"); _chk(W, name="W")
    V, D = W.shape
    out = torch.empty(*idx.shape, D, device=W.device, dtype=torch.float32)
    _ok(lib().fs2hip_embedding_fwd(_p(idx), _p(W), _p(out), idx.numel(), V, D, _stream()), "embedding_fwd")
    return out


def embedding_bwd(idx, dy, dW, padding_idx=-1):
    """dW[v] = sum of dy rows whose index is v (row ``padding_idx`` stays 0): one-hot^T @ dy on the MFMA GEMM."""
    _chk(idx, torch.int32, "idx"); _chk(dy, name="dy"); _chk(dW, name="dW")
    V, D = dW.shape
    M = idx.numel()
    _req(_rows(dy) == M and dy.shape[-1] == D, "embedding_bwd: shape mismatch")
    Vp = (V + 3) // 4 * 4
    oh = torch.empty(M, Vp, device=dy.device, dtype=torch.float32)
    _ok(lib().fs2hip_onehot(_p(idx), _p(oh), M, Vp, padding_idx, _stream()), "onehot")
    return linear_bwd_weight(oh, dy, dW, n_valid=V)


def bucket_embed_add(val, bins, W, x, control=1.0):
    """x + W[bucketize(val * control, bins)] ; returns (out, idx int32)."""
    _chk(val, name="val"); _chk(bins, name="bins"); _chk(W, name="W"); _chk(x, name="x")
    M, D = _rows(x), x.shape[-1]
    _req(val.numel() == M and W.shape[1] == D and W.shape[0] >= bins.numel() + 1, "bucket_embed_add: shape mismatch")
    out = torch.empty_like(x)
    idx = torch.empty(val.shape, device=x.device, dtype=torch.int32)
    _ok(lib().fs2hip_bucket_embed_add(_p(val), control, _p(bins), bins.numel(), _p(W), _p(x), _p(out), _p(idx), M, D,
                                      _stream()), "bucket_embed_add")
    return out, idx


# ------------------------------------------------------------------------------------------
# LengthRegulator, predictor head, losses
# ------------------------------------------------------------------------------------------
def length_regulate_fwd(x, dur, Tm, table=None):
    """x [B, Ts, D], dur [B, Ts] int32 -> (out [B, Tm, D], cum, out_lens [B] int32)."""
    _chk(x, name="x"); _chk(dur, torch.int32, "dur")
    B, Ts, D = x.shape
    _req(dur.shape == (B, Ts) and Tm > 0, "length_regulate_fwd: shape mismatch")
    if table is not None:
        _chk(table, name="table")
        _req(table.shape[0] >= Tm and table.shape[1] == D, "length_regulate_fwd: table too short")
    out = torch.empty(B, Tm, D, device=x.device, dtype=torch.float32)
    cum = torch.empty(B, Ts, device=x.device, dtype=torch.int32)
    lens = torch.empty(B, device=x.device, dtype=torch.int32)
    _ok(lib().fs2hip_length_regulate_fwd(_p(x), _p(dur), _p(table), _p(out), _p(cum), _p(lens), None, B, Ts, Tm, D,
                                         _stream()), "length_regulate_fwd")
    return out, cum, lens


def length_regulate_bwd(dy, cum):
    _chk(dy, name="dy"); _chk(cum, torch.int32, "cum")
    B, Tm, D = dy.shape
    Ts = cum.shape[1]
    _req(cum.shape[0] == B, "length_regulate_bwd: shape mismatch")
    dx = torch.empty(B, Ts, D, device=dy.device, dtype=torch.float32)
    _ok(lib().fs2hip_length_regulate_bwd(_p(dy), _p(cum), _p(dx), B, Ts, Tm, D, _stream()), "length_regulate_bwd")
    return dx


def rowdot_fwd(x, w, bias, lens, B, T):
    _chk(x, name="x"); _chk(w, name="w"); _chk(bias, name="bias"); _chk(lens, torch.int32, "lens")
    Cc = x.shape[-1]
    _req(_rows(x) == B * T and w.numel() == Cc and bias.numel() == 1 and lens.numel() == B, "rowdot_fwd: shape mismatch")
    out = torch.empty(B, T, device=x.device, dtype=torch.float32)
    _ok(lib().fs2hip_rowdot_fwd(_p(x), _p(w), _p(bias), _p(lens), _p(out), B * T, T, Cc, _stream()), "rowdot_fwd")
    return out


def rowdot_bwd(dout, x, w, lens, dw, dbias, B, T):
    for n, t in (("dout", dout), ("x", x), ("w", w), ("dw", dw), ("dbias", dbias)):
        _chk(t, name=n)
    _chk(lens, torch.int32, "lens")
    Cc = x.shape[-1]
    _req(_rows(x) == B * T and dout.numel() == B * T and w.numel() == Cc and dw.numel() == Cc and dbias.numel() == 1,
         "rowdot_bwd: shape mismatch")
    dx = torch.empty_like(x)
    nblk = lib().fs2hip_rowdot_blocks(B * T)
    ws = _workspace(nblk * (Cc + 1), x.device)
    _ok(lib().fs2hip_rowdot_bwd(_p(dout), _p(x), _p(w), _p(lens), _p(dx), _p(ws), _p(dw), _p(dbias), B * T, T, Cc,
                                _stream()), "rowdot_bwd")
    return dx


def masked_loss(pred, target, lens, B, T, Cc, *, kind="mse", weight=1.0, loss_out, want_grad=True):
    """loss_out (1-element view) = weight * mean(f((pred - target) * mask)); returns d/dpred or None.
    An int32 ``target`` is a duration: the loss uses log(target + 1)."""
    _chk(pred, name="pred"); _chk(lens, torch.int32, "lens"); _chk(loss_out, name="loss_out")
    _req(pred.numel() == B * T * Cc and target.numel() == B * T * Cc and lens.numel() == B and loss_out.numel() == 1,
         "masked_loss: shape mismatch")
    tf = ti = None
    if target.dtype == torch.int32:
        ti = _chk(target, torch.int32, "target")
    else:
        tf = _chk(target, name="target")
    dpred = torch.empty_like(pred) if want_grad else None
    ws = _workspace(1024, pred.device)
    _ok(lib().fs2hip_masked_loss(_p(pred), _p(tf), _p(ti), _p(lens), B, T, Cc, 0 if kind == "mse" else 1, weight,
                                 _p(dpred), _p(ws), _p(loss_out), _stream()), "masked_loss")
    return dpred


# ------------------------------------------------------------------------------------------
# optimizer and elementwise
# ------------------------------------------------------------------------------------------
def zeros(*shape, device, dtype=torch.float32) -> torch.Tensor:
    """``torch.zeros`` as an entry-point launch (``fs2hip_memset`` on the current stream): inside a training step the
    fill must be part of a recorded launch plan, which an ATen kernel is not."""
    t = torch.empty(*shape, device=device, dtype=dtype)
    _req(t.is_cuda and t.device.index == _current_device(), "zeros: the tensor must live on the current GPU")
    _ok(lib().fs2hip_memset(_p(t), 0, t.numel() * t.element_size(), _stream()), "memset")
    return t


def new_step_state(device) -> torch.Tensor:
    """32-byte device record {uint64 step; float lr, bc1, bc2, clip_coef, grad_norm, pad} as 4 x int64."""
    st = torch.zeros(4, device=device, dtype=torch.int64)
    st.view(torch.float32)[5] = 1.0  # clip_coef
    return st


def step_advance(state, base_lr, warmup, beta1, beta2):
    _chk(state, torch.int64, "state")
    _ok(lib().fs2hip_step_advance(_p(state), base_lr, float(warmup), beta1, beta2, _stream()), "step_advance")


def grad_clip_coef(grad, max_norm, grad_scale, state):
    _chk(grad, name="grad"); _chk(state, torch.int64, "state")
    ws = _workspace(1024, grad.device)
    _ok(lib().fs2hip_grad_clip_coef(_p(grad), grad.numel(), float(max_norm), float(grad_scale), _p(ws), _p(state),
                                    _stream()), "grad_clip_coef")


def adamw_step(p, g, m, v, state, beta1, beta2, eps, weight_decay):
    for n, t in (("p", p), ("g", g), ("m", m), ("v", v)):
        _chk(t, name=n)
    _chk(state, torch.int64, "state")
    _req(p.numel() == g.numel() == m.numel() == v.numel(), "adamw_step: size mismatch")
    _ok(lib().fs2hip_adamw_step(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(state), beta1, beta2, eps, weight_decay,
                                _stream()), "adamw_step")


def axpby(x, y=None, a=1.0, b=1.0, drop: Drop = NO_DROP, out=None):
    _chk(x, name="x")
    if y is not None:
        _chk(y, name="y")
        _req(y.numel() == x.numel(), "axpby: size mismatch")
    if out is None:
        out = torch.empty_like(x)
    _chk(out, name="out")
    _ok(lib().fs2hip_axpby(_p(x), _p(y), _p(out), x.numel(), a, b, drop.p, drop.seed, drop.step_ptr, _stream()),
        "axpby")
    return out


def scale_dev(x, scalar):
    """x *= scalar, ``scalar`` a 1-element fp32 tensor in device memory (nothing is moved when it holds exactly 1)."""
    _chk(x, name="x"); _chk(scalar, name="scalar")
    _req(scalar.numel() == 1, "scale_dev: one scalar")
    _ok(lib().fs2hip_scale_dev(_p(x), x.numel(), _p(scalar), _stream()), "scale_dev")
    return x


def add_rowvec(x, e, B, T):
    _chk(x, name="x"); _chk(e, name="e")
    D = x.shape[-1]
    _req(_rows(x) == B * T and e.numel() == B * D, "add_rowvec: shape mismatch")
    out = torch.empty_like(x)
    _ok(lib().fs2hip_add_rowvec(_p(x), _p(e), _p(out), B, T, D, _stream()), "add_rowvec")
    return out


def dact_mul(dy, aux, act):
    _chk(dy, name="dy"); _chk(aux, name="aux")
    _req(dy.numel() == aux.numel(), "dact_mul: size mismatch")
    out = torch.empty_like(dy)
    _ok(lib().fs2hip_dact_mul(_p(dy), _p(aux), _p(out), dy.numel(), _ACT[act], _stream()), "dact_mul")
    return out


def mask_from_lens(lens, T):
    _chk(lens, torch.int32, "lens")
    B = lens.numel()
    mask = torch.empty(B, T, device=lens.device, dtype=torch.bool)
    _ok(lib().fs2hip_mask_from_lens(_p(lens), _p(mask), B, T, _stream()), "mask_from_lens")
    return mask


def sum_slots(x, n, out):
    _chk(x, name="x"); _chk(out, name="out")
    _req(x.numel() >= n and out.numel() == 1, "sum_slots: size mismatch")
    _ok(lib().fs2hip_sum_slots(_p(x), n, _p(out), _stream()), "sum_slots")


# ------------------------------------------------------------------------------------------
# learned alignment (aligner.hip)
# ------------------------------------------------------------------------------------------
def attn_dist(q, k):
    """logits[b,t1,t2] = -0.0005 * sum_c (q[b,t1,c] - k[b,t2,c])^2."""
    _chk(q, name="q"); _chk(k, name="k")
    B, T1, Cc = q.shape
    _req(k.dim() == 3 and k.shape[0] == B and k.shape[2] == Cc, "attn_dist: shape mismatch")
    T2 = k.shape[1]
    logits = torch.empty(B, T1, T2, device=q.device, dtype=torch.float32)
    _ok(lib().fs2hip_attn_dist(_p(q), _p(k), _p(logits), B, T1, T2, Cc, _stream()), "attn_dist")
    return logits


def attn_softmax(logits, prior, key_lens):
    _chk(logits, name="logits"); _chk(prior, name="prior"); _chk(key_lens, torch.int32, "key_lens")
    B, T1, T2 = logits.shape
    _req(prior.shape == logits.shape and key_lens.numel() == B, "attn_softmax: shape mismatch")
    logprob, soft = torch.empty_like(logits), torch.empty_like(logits)
    _ok(lib().fs2hip_attn_softmax(_p(logits), _p(prior), _p(key_lens), _p(logprob), _p(soft), B, T1, T2, _stream()),
        "attn_softmax")
    return logprob, soft


def mas(x, in_lens, out_lens, is_log=False):
    """Monotonic alignment search.  x [B, Tm, Ts] = attn_soft (or log-probabilities when ``is_log``).
    Returns (hard [B,Tm,Ts] fp32 0/1, hard_idx [B,Tm] int32, dur [B,Ts] int32)."""
    _chk(x, name="x"); _chk(in_lens, torch.int32, "in_lens"); _chk(out_lens, torch.int32, "out_lens")
    B, Tm, Ts = x.shape
    _req(in_lens.numel() == B and out_lens.numel() == B, "mas: lens size")
    hard = torch.empty_like(x)
    hard_idx = torch.empty(B, Tm, device=x.device, dtype=torch.int32)
    dur = torch.empty(B, Ts, device=x.device, dtype=torch.int32)
    W = (Ts + 31) // 32
    dirs = torch.empty(B * Tm * W, device=x.device, dtype=torch.int32)
    _ok(lib().fs2hip_mas(_p(x), int(is_log), _p(in_lens), _p(out_lens), _p(hard), _p(hard_idx), _p(dur), _p(dirs), B, Tm,
                         Ts, _stream()), "mas")
    return hard, hard_idx, dur


def duration_cumsum(dur, Tm, expect=None, bad_count=None):
    """(cum [B, Ts] inclusive cumulative durations, lens [B] = min(total, Tm)).  With ``expect`` [B] int32 a third
    result: mismatch [B] int32 = (total != expect) -- and every mismatch adds 1 to ``bad_count`` (1-element int32)."""
    _chk(dur, torch.int32, "dur")
    B, Ts = dur.shape
    cum = torch.empty(B, Ts, device=dur.device, dtype=torch.int32)
    lens = torch.empty(B, device=dur.device, dtype=torch.int32)
    mismatch = None
    if expect is not None:
        _chk(expect, torch.int32, "expect")
        _req(expect.numel() == B, "duration_cumsum: expect size")
        mismatch = torch.empty(B, device=dur.device, dtype=torch.int32)
        if bad_count is not None:
            _chk(bad_count, torch.int32, "bad_count")
            _req(bad_count.numel() == 1, "duration_cumsum: bad_count is one word")
    _ok(lib().fs2hip_duration_cumsum(_p(dur), _p(cum), _p(lens), _p(expect), _p(mismatch),
                                     _p(bad_count) if expect is not None else None, B, Ts, int(Tm), _stream()),
        "duration_cumsum")
    return (cum, lens) if expect is None else (cum, lens, mismatch)


def avg_variance(var, cum):
    _chk(var, name="var"); _chk(cum, torch.int32, "cum")
    B, Tm = var.shape
    Ts = cum.shape[1]
    _req(cum.shape[0] == B, "avg_variance: shape mismatch")
    out = torch.empty(B, Ts, device=var.device, dtype=torch.float32)
    _ok(lib().fs2hip_avg_variance(_p(var), _p(cum), _p(out), B, Tm, Ts, _stream()), "avg_variance")
    return out


def attn_ctc_loss(logprob, key_lens, query_lens, weight, loss_out, want_grad=True):
    _chk(logprob, name="logprob"); _chk(key_lens, torch.int32, "key_lens"); _chk(query_lens, torch.int32, "query_lens")
    _chk(loss_out, name="loss_out")
    B, Tm, Ts = logprob.shape
    _req(key_lens.numel() == B and query_lens.numel() == B and loss_out.numel() == 1, "attn_ctc_loss: shape mismatch")
    _req(2 * (2 * Ts + 1) * 4 + 16 <= 64 * 1024, "attn_ctc_loss: too many tokens for the on-chip state")
    alpha = torch.empty(B * Tm * (2 * Ts + 1) + B * Tm + B, device=logprob.device, dtype=torch.float32)
    lse = alpha[B * Tm * (2 * Ts + 1):]
    nll = lse[B * Tm:]
    d = torch.empty_like(logprob) if want_grad else None
    _ok(lib().fs2hip_attn_ctc_loss(_p(logprob), _p(key_lens), _p(query_lens), _p(alpha), lse.data_ptr(), nll.data_ptr(),
                                   _p(d), weight, _p(loss_out), B, Tm, Ts, _stream()), "attn_ctc_loss")
    return d


def attn_bin_loss(soft, hard_idx, weight, loss_out):
    """Writes loss_out[0]; returns the 1-element coefficient tensor the softmax backward consumes."""
    _chk(soft, name="soft"); _chk(hard_idx, torch.int32, "hard_idx"); _chk(loss_out, name="loss_out")
    B, Tm, Ts = soft.shape
    _req(hard_idx.shape == (B, Tm) and loss_out.numel() == 1, "attn_bin_loss: shape mismatch")
    ws = _workspace(1024, soft.device)
    coef = torch.empty(1, device=soft.device, dtype=torch.float32)
    _ok(lib().fs2hip_attn_bin_loss(_p(soft), _p(hard_idx), _p(ws), weight, _p(loss_out), _p(coef), B, Tm, Ts, _stream()),
        "attn_bin_loss")
    return coef


def attn_softmax_bwd(logits, soft, dlogprob, hard_idx, bin_coef):
    _chk(logits, name="logits"); _chk(soft, name="soft")
    B, T1, T2 = logits.shape
    if dlogprob is not None:
        _chk(dlogprob, name="dlogprob")
        _req(dlogprob.shape == logits.shape, "attn_softmax_bwd: dlogprob shape")
    if hard_idx is not None:
        _chk(hard_idx, torch.int32, "hard_idx"); _chk(bin_coef, name="bin_coef")
        _req(hard_idx.shape == (B, T1), "attn_softmax_bwd: hard_idx shape")
    d = torch.empty_like(logits)
    _ok(lib().fs2hip_attn_softmax_bwd(_p(logits), _p(soft), _p(dlogprob), _p(hard_idx), _p(bin_coef), _p(d), B, T1, T2,
                                      _stream()), "attn_softmax_bwd")
    return d


def attn_dist_bwd(dlogits, q, k, want_dq=True, want_dk=True):
    _chk(dlogits, name="dlogits"); _chk(q, name="q"); _chk(k, name="k")
    B, T1, Cc = q.shape
    T2 = k.shape[1]
    _req(dlogits.shape == (B, T1, T2) and k.shape == (B, T2, Cc), "attn_dist_bwd: shape mismatch")
    dq = torch.empty_like(q) if want_dq else None
    dk = torch.empty_like(k) if want_dk else None
    _ok(lib().fs2hip_attn_dist_bwd(_p(dlogits), _p(q), _p(k), _p(dq), _p(dk), B, T1, T2, Cc, _stream()), "attn_dist_bwd")
    return dq, dk


def duration_round(logd, control=1.0):
    """Inference durations: int(max(round_half_even(exp(logd) - 1) * control, 0))."""
    _chk(logd, name="logd")
    out = torch.empty(logd.shape, device=logd.device, dtype=torch.int32)
    _ok(lib().fs2hip_duration_round(_p(logd), float(control), _p(out), logd.numel(), _stream()), "duration_round")
    return out


# ------------------------------------------------------------------------------------------
# GST style encoder (gst.hip)
# ------------------------------------------------------------------------------------------
def conv2d_s2_fwd(x, w):
    """x [B, H, W, Cin] channels-last, w [3, 3, Cin, Cout] -> y [B, (H-1)//2+1, (W-1)//2+1, Cout]."""
    _chk(x, name="x"); _chk(w, name="w")
    B, Hh, Ww, Cin = x.shape
    _req(w.shape[:3] == (3, 3, Cin), "conv2d_s2_fwd: weight shape")
    Cout = w.shape[3]
    Ho, Wo = (Hh - 1) // 2 + 1, (Ww - 1) // 2 + 1
    y = torch.empty(B, Ho, Wo, Cout, device=x.device, dtype=torch.float32)
    if Cin % 4 == 0 and Cout % 4 == 0:  # gather the windows, then the MFMA GEMM: y[M, Cout] = col[M, 9 Cin] @ w[9 Cin, Cout]
        col = _im2col_s2(x)
        linear_bwd_data(col, w.view(9 * Cin, Cout), out=y.view(B * Ho * Wo, Cout))
        return y
    _ok(lib().fs2hip_conv2d_s2_fwd(_p(x), _p(w), _p(y), B, Hh, Ww, Cin, Cout, _stream()), "conv2d_s2_fwd")
    return y


def _im2col_s2(x):
    B, Hh, Ww, Cin = x.shape
    Ho, Wo = (Hh - 1) // 2 + 1, (Ww - 1) // 2 + 1
    col = torch.empty(B * Ho * Wo, 9 * Cin, device=x.device, dtype=torch.float32)
    _ok(lib().fs2hip_im2col_s2(_p(x), _p(col), B, Hh, Ww, Cin, _stream()), "im2col_s2")
    return col


def conv2d_s2_bwd(dy, x, w, dw, need_dx=True):
    """Writes dw [3,3,Cin,Cout]; returns dx (or None)."""
    _chk(dy, name="dy"); _chk(x, name="x"); _chk(w, name="w"); _chk(dw, name="dw")
    B, Hh, Ww, Cin = x.shape
    Cout = w.shape[3]
    _req(dy.shape == (B, (Hh - 1) // 2 + 1, (Ww - 1) // 2 + 1, Cout) and dw.numel() == w.numel(), "conv2d_s2_bwd: shapes")
    if Cin % 4 == 0 and Cout % 4 == 0:
        M = dy.numel() // Cout
        dy2 = dy.view(M, Cout)
        col = _im2col_s2(x)  # re-gathered rather than kept since the forward pass: one cheap pass against 9x the input
        linear_bwd_weight(col, dy2, dw)                  # dw[9 Cin, Cout] = col^T @ dy
        if not need_dx:
            return None
        dcol = linear_fwd(dy2, w.view(9 * Cin, Cout))    # dcol[M, 9 Cin] = dy @ w^T
        dx = torch.empty_like(x)
        _ok(lib().fs2hip_col2im_s2(_p(dcol), _p(dx), B, Hh, Ww, Cin, _stream()), "col2im_s2")
        return dx
    if Cin == 1 and Cout % 4 == 0 and not need_dx:  # first layer: taps padded to 12 columns, gradient rows 0..8
        M = dy.numel() // Cout
        col = torch.empty(M, 12, device=x.device, dtype=torch.float32)
        _ok(lib().fs2hip_im2col_s2(_p(x), _p(col), B, Hh, Ww, 1, _stream()), "im2col_s2")
        linear_bwd_weight(col, dy.view(M, Cout), dw, n_valid=9)
        return None
    parts = lib().fs2hip_conv2d_s2_wgrad_parts(B, Hh, Ww)
    ws = _workspace(parts * w.numel(), x.device)
    _ok(lib().fs2hip_conv2d_s2_bwd_weight(_p(x), _p(dy), _p(ws), _p(dw), B, Hh, Ww, Cin, Cout, _stream()), "conv2d_s2_bwd_weight")
    if not need_dx:
        return None
    dx = torch.empty_like(x)
    _ok(lib().fs2hip_conv2d_s2_bwd_data(_p(dy), _p(w), _p(dx), B, Hh, Ww, Cin, Cout, _stream()), "conv2d_s2_bwd_data")
    return dx


def gru_gate_fwd(gi_t, gi_stride, gh, hprev, U, hnew=None):
    """gi_t: view of the step's input projections (first row), rows ``gi_stride`` floats apart."""
    _chk(gh, name="gh"); _chk(hprev, name="hprev")
    B = hprev.shape[0]
    hnew = torch.empty_like(hprev) if hnew is None else _chk(hnew, name="hnew")
    gates = torch.empty(B, 4 * U, device=gh.device, dtype=torch.float32)
    _ok(lib().fs2hip_gru_gate_fwd(gi_t.data_ptr(), gi_stride, _p(gh), _p(hprev), _p(hnew), _p(gates), B, U, _stream()),
        "gru_gate_fwd")
    return hnew, gates


def gru_gate_bwd(dh, gates, hprev, dgi_t, dgi_stride, U, dgh=None):
    _chk(dh, name="dh"); _chk(gates, name="gates"); _chk(hprev, name="hprev")
    B = hprev.shape[0]
    dgh = torch.empty(B, 3 * U, device=dh.device, dtype=torch.float32) if dgh is None else _chk(dgh, name="dgh")
    dhprev = torch.empty_like(hprev)
    _ok(lib().fs2hip_gru_gate_bwd(_p(dh), _p(gates), _p(hprev), dgi_t.data_ptr(), dgi_stride, _p(dgh), _p(dhprev), B, U,
                                  _stream()), "gru_gate_bwd")
    return dgh, dhprev


def gst_attn_fwd(q, k, v, heads):
    _chk(q, name="q"); _chk(k, name="k"); _chk(v, name="v")
    B, F = q.shape
    NT = k.shape[0]
    _req(F == heads * 64 and k.shape == (NT, F) and v.shape == (NT, F), "gst_attn_fwd: shapes")
    p = torch.empty(B, heads, NT, device=q.device, dtype=torch.float32)
    ctx = torch.empty_like(q)
    _ok(lib().fs2hip_gst_attn_fwd(_p(q), _p(k), _p(v), _p(p), _p(ctx), B, NT, heads, _stream()), "gst_attn_fwd")
    return p, ctx


def gst_attn_bwd(dctx, q, k, v, p, heads):
    """Returns dq [B,F] and per-utterance dk, dv [B, NT, F] (sum over the batch with colsum)."""
    for n, t in (("dctx", dctx), ("q", q), ("k", k), ("v", v), ("p", p)):
        _chk(t, name=n)
    B, F = q.shape
    NT = k.shape[0]
    dq = torch.empty_like(q)
    dk = torch.empty(B, NT, F, device=q.device, dtype=torch.float32)
    dv = torch.empty_like(dk)
    _ok(lib().fs2hip_gst_attn_bwd(_p(dctx), _p(q), _p(k), _p(v), _p(p), _p(dq), _p(dk), _p(dv), B, NT, heads, _stream()),
        "gst_attn_bwd")
    return dq, dk, dv


def act_apply(x, act):
    _chk(x, name="x")
    out = torch.empty_like(x)
    _ok(lib().fs2hip_act_apply(_p(x), _p(out), x.numel(), _ACT[act], _stream()), "act_apply")
    return out
