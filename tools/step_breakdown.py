"""Per-kernel time of ONE training step from a rocprofv3 kernel trace (csv): the step between the last two
AdamW launches, so the autotuner's timing launches and warm-up are excluded.
usage: python tools/step_breakdown.py <..._kernel_trace.csv> [n_steps_back]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
lo, hi = idx[-1 - back], idx[-1]
seg = rows[lo + 1:hi + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
agg = defaultdict(lambda: [0, 0])
busy = 0
for r in seg:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    agg[name][0] += d
    agg[name][1] += 1
    busy += d
print(f"{back} step(s): wall {(t1 - t0) / 1e6 / back:.2f} ms/step, kernel busy {busy / 1e6 / back:.2f} ms/step, "
      f"{len(seg) // back} launches/step")
fam = defaultdict(int)
for k, (d, n) in agg.items():
    fam[k.split("<")[0]] += d
print("-- families")
for k, d in sorted(fam.items(), key=lambda kv: -kv[1])[:25]:
    print(f"{d / 1e6 / back:8.3f} ms {100 * d / busy:5.1f}%  {k}")
print("-- kernels")
for k, (d, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{d / 1e6 / back:8.3f} ms {100 * d / busy:5.1f}%  n={n // back:4d} avg={d / n / 1e3:8.1f} us  {k}")

# ---- concurrency: union of kernel intervals vs sum (two streams) and idle time inside the step
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s_, e_ in iv[1:]:
    if s_ > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
union += cur_e - cur_s
print(f"-- timeline: union of kernel intervals {union / 1e6 / back:.2f} ms/step, idle {(t1 - t0 - union) / 1e6 / back:.2f} ms/step, "
      f"overlapped {(busy - union) / 1e6 / back:.2f} ms/step")

# ---- per queue (= HIP stream): busy time and the largest kernels' share, to see which stream is the critical path
if "Queue_Id" in seg[0]:
    per_q = defaultdict(lambda: [0, 0])
    for r in seg:
        per_q[r["Queue_Id"]][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        per_q[r["Queue_Id"]][1] += 1
    for q_, (d, n) in sorted(per_q.items(), key=lambda kv: -kv[1][0]):
        print(f"-- queue {q_}: busy {d / 1e6 / back:.2f} ms/step in {n // back} launches/step "
              f"(idle inside the step {(t1 - t0 - d) / 1e6 / back:.2f} ms/step)")
    # the largest gaps on the busiest queue (the main chain): is it waiting for the other stream, or for the host?
    main_q = max(per_q.items(), key=lambda kv: kv[1][0])[0]
    mq = [r for r in seg if r["Queue_Id"] == main_q]
    gaps = []
    for a_, b_ in zip(mq[:-1], mq[1:]):
        gap = int(b_["Start_Timestamp"]) - int(a_["End_Timestamp"])
        if gap > 0:
            gaps.append((gap, a_["Kernel_Name"], b_["Kernel_Name"]))
    tot = sum(g[0] for g in gaps)
    print(f"-- queue {main_q}: {len(gaps)} gaps, {tot / 1e6 / back:.2f} ms/step in total; the largest:")
    short = lambda n: re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", n)).replace("void ", "")[:60]
    for gap, a_, b_ in sorted(gaps, reverse=True)[:12]:
        print(f"   {gap / 1e3:8.1f} us  after {short(a_)}  before {short(b_)}")
    hist = defaultdict(int)
    for gap, _, _ in gaps:
        hist[min(int(gap / 1e3), 20)] += 1
    print("   gap histogram (us: count): " + " ".join(f"{k}{'+' if k == 20 else ''}:{v // back}" for k, v in sorted(hist.items())))
