"""Static check of a gfx950 assembly listing: no instruction may touch the destination registers of an inline-asm
vector-memory load (buffer_load_dword* without `lds`) before the s_waitcnt vmcnt that retires it.  Vector-memory
operations retire in issue order; every buffer / global / scratch operation counts.  Linear scan (a loop's back edge is
not followed: a wait inside the loop body is what the kernels rely on).
usage: check_inflight_vmem.py listing.s [kernel-name-substring]"""
import re
import sys

from check_inflight_lds import regs


def main():
    path, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    name, pending, bad, n = None, [], 0, 0
    for ln, line in enumerate(open(path), 1):
        t = line.split(";")[0].strip()
        if t.endswith(":") and not t.startswith(".") and not t.startswith("BB"):
            name, pending = t[:-1], []
            continue
        if not name or want not in name or not t or t.startswith("."):
            continue
        op = t.split()[0]
        if re.match(r"(buffer|global|scratch|flat)_(load|store|atomic)", op):
            if op.startswith("buffer_load") and " lds" not in t:
                pending.append(regs(t.split()[1].rstrip(",")))
                n += 1
            else:
                pending.append(set())
            continue
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", t)
        if m:
            k = int(m.group(1))
            while len(pending) > k:
                pending.pop(0)
            continue
        if op in ("s_barrier", "s_endpgm") or op.startswith("s_waitcnt"):
            continue
        used = regs(t)
        for p in pending:
            if p & used:
                bad += 1
                print(f"{name}: line {ln}: `{t}` touches in-flight {sorted(p & used)}")
                break
    print(f"{n} loads checked, {bad} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
