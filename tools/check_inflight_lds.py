"""Static check of a gfx950 assembly listing: no instruction may touch the destination registers of an inline-asm LDS
read (ds_read_*) before the s_waitcnt lgkmcnt that retires it.  (The compiler believes an asm read's result is there at
once; under register pressure it may copy such a register -- e.g. park it in an AGPR -- while the read is in flight.)
usage: check_inflight_lds.py listing.s [kernel-name-substring]"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]", tok):
        out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    for m in re.finditer(r"\b([va])(\d+)\b", tok):
        out.add((m.group(1), int(m.group(2))))
    return out


def main():
    path, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    name, pending, bad, n = None, [], 0, 0
    for ln, line in enumerate(open(path), 1):
        t = line.split(";")[0].strip()
        if t.endswith(":") and not t.startswith("."):
            name, pending = t[:-1], []
            continue
        if not name or want not in name or not t or t.startswith(";") or t.startswith("."):
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        op = t.split()[0]
        if op.startswith("ds_read"):
            dst = t.split()[1].rstrip(",")
            pending.append(regs(dst))
            n += 1
            continue
        if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
            pending.append(set())  # another lgkm operation: counted, nothing to protect
            continue
        m = re.match(r"s_waitcnt.*lgkmcnt\((\d+)\)", t)
        if m:
            k = int(m.group(1))
            while len(pending) > k:
                pending.pop(0)
            continue
        if op == "s_waitcnt" and "lgkmcnt" not in t and "vmcnt" not in t:
            pending = []
            continue
        if op in ("s_barrier", "s_endpgm"):
            continue
        used = regs(t)
        for p in pending:
            if p & used:
                bad += 1
                print(f"{name}: line {ln}: `{t}` touches in-flight {sorted(p & used)}")
                break
    print(f"{n} LDS reads checked, {bad} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
