"""Coarse two-stream timeline of ONE training step from a rocprofv3 kernel trace (csv): per time bin and per queue (= HIP
stream) the busy fraction and the kernel that owns most of the bin -- shows where the main chain waits and where the side
stream is the critical path.  The step is the one between the last two AdamW launches.
usage: python tools/timeline.py <..._kernel_trace.csv> [bin_us=250]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
bin_ns = int(float(sys.argv[2]) * 1000) if len(sys.argv) > 2 else 250_000
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
seg = rows[idx[-2] + 1:idx[-1] + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", n).replace("void ", "")[:34]


queues = sorted({r["Queue_Id"] for r in seg}, key=lambda q: -sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg if r["Queue_Id"] == q))
nb = (t1 - t0) // bin_ns + 1
busy = {q: [0] * nb for q in queues}
own = {q: [defaultdict(int) for _ in range(nb)] for q in queues}
for r in seg:
    s, e, q = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Queue_Id"]
    b = s // bin_ns
    while b * bin_ns < e:
        lo, hi = max(s, b * bin_ns), min(e, (b + 1) * bin_ns)
        busy[q][b] += hi - lo
        own[q][b][short(r["Kernel_Name"])] += hi - lo
        b += 1
print(f"step {(t1 - t0) / 1e6:.2f} ms, {len(seg)} launches, queues by busy time: " +
      ", ".join(f"{q}: {sum(busy[q]) / 1e6:.2f} ms" for q in queues))
for b in range(nb):
    cells = []
    for q in queues:
        top = max(own[q][b].items(), key=lambda kv: kv[1])[0] if own[q][b] else "-"
        cells.append(f"{busy[q][b] / bin_ns:4.2f} {top:34s}")
    print(f"{b * bin_ns / 1e6:6.2f} ms | " + " | ".join(cells))
