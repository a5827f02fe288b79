"""CPU: the launch-plan table generated from the header is current, and the recorder packs arguments the way the
replayer's generated unpacking lines read them."""
import ctypes as C
import struct
import subprocess
import sys
from pathlib import Path

import torch  # noqa: F401  (before the library: one HIP runtime per process)

REPO = Path(__file__).resolve().parent.parent


def test_generated_thunks_match_the_header():
    r = subprocess.run([sys.executable, str(REPO / "tools" / "gen_plan_thunks.py"), "--check"])
    assert r.returncode == 0, "fastspeech2_lightning_amd/csrc/plan_thunks.inc is stale: run tools/gen_plan_thunks.py"


def test_every_stream_entry_point_is_a_plan_op_and_queries_are_not():
    from fastspeech2_lightning_amd import build, hip, plan
    build.build()
    n = hip.real_lib().fs2hip_plan_op_count()
    ids = {}
    for name, sig in hip.SIGNATURES.items():
        i = plan.op_id(name)
        if name.startswith("fs2hip_plan_") or name in ("fs2hip_version",):
            assert i == -1, name
        elif sig is None or sig.endswith("p") and len(sig) > 1:
            ids[name] = i
    launchers = {k: v for k, v in ids.items() if v >= 0}
    assert len(set(launchers.values())) == len(launchers) == n
    for q in ("fs2hip_dwconv_blocks", "fs2hip_layernorm_bwd_blocks", "fs2hip_colsum_rows", "fs2hip_attention_b_supported"):
        assert plan.op_id(q) == -1
    assert plan.op_id("fs2hip_gemm") >= 0 and plan.op_id("fs2hip_memset") >= 0


def test_recorder_packs_pointers_integers_floats_and_struct_copies():
    from fastspeech2_lightning_amd import hip, plan
    rec = plan.Recorder(main_stream=0)
    # scalar signature: pointer, None pointer, negative int, float, long long, unsigned long long, stream
    rec.add(7, "ppifqQp", (0x7f0000001000, None, -3, 0.2, -5, 0xFFFFFFFFFFFFFFF0, 0), "x")
    op, st, slots = rec.cmds[-1]
    assert (op, st) == (7, 0)
    assert slots[0] == 0x7f0000001000 and slots[1] == 0
    assert C.c_int(slots[2] & 0xFFFFFFFF).value == -3 and C.c_longlong(slots[2]).value == -3
    assert struct.unpack("<f", struct.pack("<I", slots[3]))[0] == struct.unpack("<f", struct.pack("<f", 0.2))[0]
    assert C.c_longlong(slots[4]).value == -5 and slots[5] == 0xFFFFFFFFFFFFFFF0
    # struct argument: a host copy the plan owns, immune to the caller reusing its struct
    a = hip.GemmArgs()
    a.Mc, a.Nc, a.alpha = 11, 22, 0.5
    rec.add(0, None, (C.byref(a), 0x55), "fs2hip_gemm")
    a.Mc = 99
    op, st, slots = rec.cmds[-1]
    assert st == 1 and rec.side == 0x55 and rec.streams == [0, 0x55]
    kept = C.cast(slots[0], C.POINTER(hip.GemmArgs)).contents
    assert (kept.Mc, kept.Nc, kept.alpha) == (11, 22, 0.5)
    jobs = (hip.ReduceJob * 3)()
    jobs[2].rows = 5
    rec.add(1, None, (jobs, 3, 0), "fs2hip_reduce_rows_multi")
    op, st, slots = rec.cmds[-1]
    assert slots[1] == 3 and C.cast(slots[0], C.POINTER(hip.ReduceJob))[2].rows == 5
    # forks / joins number their events; further streams get the next indices, up to the table's size
    rec.sync(0, 0x55)
    rec.sync(0x55, 0)
    assert [c[2] for c in rec.cmds[-2:]] == [[0, 0, 1], [1, 1, 0]] and rec.n_events == 2
    rec.add(3, "pp", (1, 0x66), "y")
    assert rec.cmds[-1][1] == 2 and rec.streams == [0, 0x55, 0x66]
    try:
        for h in range(0x100, 0x100 + plan.MAX_STREAMS):
            rec.stream_index(h)
    except plan.PlanError:
        pass
    else:
        raise AssertionError("more streams than the table holds must be refused")


def test_plan_command_layout_matches_the_header():
    from fastspeech2_lightning_amd import plan
    text = (REPO / "include" / "fs2hip.h").read_text()
    assert f"#define FS2_PLAN_SLOTS {plan.SLOTS}" in text and f"#define FS2_PLAN_SYNC ({plan.SYNC})" in text
    assert C.sizeof(plan.PlanCmd) == 8 + 8 * plan.SLOTS
