"""Per-launch HBM traffic of the GEMM kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

Usage (on the GPU box, separate passes as MI355X_MICROARCH.md prescribes):
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT/fetch -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-graph --no-roofline
  rocprofv3 --pmc WRITE_SIZE ... -d OUT/write -- (same)
  python tools/pmc_traffic.py OUT/fetch OUT/write N_GEMM_PER_STEP [the bench.py flags of the profiled command]

Only the LAST step's GEMM dispatches are used (earlier ones include the tile autotuner's timing launches).
gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request for wide coalesced
reads -> doubled; WRITE_SIZE is exact; both are in KiB.
"""
import csv
import glob
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def last_step(dirname, counter, n):
    f = glob.glob(f"{dirname}/*/*_counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "gemm" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    rows = rows[-n:]
    return sum(float(r["Counter_Value"]) for r in rows), len(rows)


def whole_step(dirname, counter):
    """Sum over EVERY dispatch of the last step (the dispatches between the last two optimizer launches)."""
    f = glob.glob(f"{dirname}/*/*_counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
    if len(marks) < 2:
        return None, 0
    sel = rows[marks[-2] + 1:marks[-1] + 1]
    return sum(float(r["Counter_Value"]) for r in sel), len(sel)


def main():
    fetch_dir, write_dir, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    fetch_kib, nf = last_step(fetch_dir, "FETCH_SIZE", n)
    write_kib, nw = last_step(write_dir, "WRITE_SIZE", n)
    wf, nwf = whole_step(fetch_dir, "FETCH_SIZE")
    ww, _ = whole_step(write_dir, "WRITE_SIZE")
    # the figure is only valid for the kernel sources AND the configuration it was measured on: both are stamped,
    # and bench.py reports it only for a run whose stamp matches
    from bench import build_parser, config_signature, kernel_source_hash
    out = {
        "kernel_source_hash": kernel_source_hash(),
        "config": config_signature(build_parser().parse_args(sys.argv[4:])),
        "launches": nf,
        "fetch_bytes_per_launch_raw": fetch_kib * 1024 / nf,
        "fetch_bytes_per_launch_corrected": 2 * fetch_kib * 1024 / nf,
        "write_bytes_per_launch": write_kib * 1024 / nw,
        "hbm_bytes_per_launch": (2 * fetch_kib + write_kib) * 1024 / nf,
        "whole_step": None if wf is None else {
            "launches": nwf, "fetch_bytes_corrected": 2 * wf * 1024, "write_bytes": ww * 1024,
            "hbm_bytes": (2 * wf + ww) * 1024},
        "note": "last benchmark step only; FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request); L2-side "
                "counters, Infinity-Cache hits included",
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
