"""MFMA-pipe utilisation per kernel family in the benchmark step, from two rocprofv3 --pmc passes.

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d OUT/mfma -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --graph
  rocprofv3 --pmc GRBM_GUI_ACTIVE          ... -d OUT/active -- (same)
  python tools/pmc_mfma.py OUT/mfma OUT/active

SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of every SIMD's matrix pipe; GRBM_GUI_ACTIVE sums the active cycles of
the 8 XCDs.  utilisation = busy / (active / 8 * 1024 SIMDs).  Only dispatches of the last step are used."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def load(dirname, counter):
    f = glob.glob(f"{dirname}/*/*_counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    last = max(i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"])
    prev = max(i for i, r in enumerate(rows[:last]) if "adamw" in r["Kernel_Name"])
    out = defaultdict(float)
    for r in rows[prev + 1:last + 1]:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"[<(].*", "", name).replace("void ", "")
        out[name] += float(r["Counter_Value"])
    return out


busy, active = load(sys.argv[1], "SQ_VALU_MFMA_BUSY_CYCLES"), load(sys.argv[2], "GRBM_GUI_ACTIVE")
res = {}
for fam, pat in (("gemm (gemm2 / gemm2p / gemm kernels)", r"^gemm"), ("attention (fwd / dQ / dK-dV)", r"^attn[2b]?_(fwd|bwd)"),
                 ("whole step", r".")):
    b = sum(v for k, v in busy.items() if re.search(pat, k))
    a = sum(v for k, v in active.items() if re.search(pat, k))
    res[fam] = {"mfma_busy_cycles": b, "gpu_active_cycles_all_xcds": a, "mfma_utilisation": round(b / (a / 8 * 1024), 4) if a else None}
print(json.dumps(res, indent=1))
