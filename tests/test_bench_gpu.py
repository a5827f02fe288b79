"""``python bench.py`` as the driver runs it (one GPU, default flags but a small batch and few steps): the step must come from
its launch plan, the tuner's in-step stage must run and leave a working table, and the line must carry the fields the
contract and DESIGN.md name."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def test_default_line_replays_its_plan_and_refines_tiles_in_the_step():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "FS2_GEMM_TUNE")}
    env.update(FS2_REFINE_TOP="6", FS2_REFINE_CANDIDATES="2")
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--batch", "4", "--steps", "6", "--warmup", "2",
                        "--no-extra-legs", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["steps"] == 6 and line["value"] > 0 and line["scaling"] == "weak"
    assert line["dtype"] == "f32" and line["config"]["precision"] == "32-true" and line["vs_baseline"] is None
    plan = line["launch_plan"]
    assert plan["replayed_steps"] >= 6 and plan["launches_per_step"] > 300 and plan["segments"] == 1
    assert 0 < line["host_enqueue_ms_per_step"] < line["host_loop_ms_per_step"] + 50
    roof = line["roofline"]
    assert roof["bound"] == "mfma" and roof["peak"] == 157.3 and 0 < roof["frac"] < 1 and "time_source" in roof
    assert "in-step tile refinement:" in r.stderr
    assert line["loss_total"] == line["loss_total"]  # finite
