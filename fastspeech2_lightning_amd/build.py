"""Builds the fs2hip shared library (C ABI, include/fs2hip.h) in-tree with hipcc for gfx950."""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
INCLUDE = PKG.parent / "include"
LIB = PKG / "_fs2hip.so"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (Path(cand).exists() or cand == "hipcc"):
            return cand
    raise RuntimeError("hipcc not found")


def sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _deps(depfile: Path) -> list[Path]:
    """Headers of this repo that a source included when it was last compiled (the compiler's -MD output); without a
    depfile every header counts."""
    every = list(CSRC.glob("*.h")) + list(INCLUDE.glob("*.h"))
    if not depfile.exists():
        return every
    words = depfile.read_text().replace("\\\n", " ").split()
    mine = [Path(w) for w in words[1:] if w.endswith(".h") and (str(CSRC) in w or str(INCLUDE) in w)]
    return mine or every


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = sources() + list(CSRC.glob("*.h")) + list(INCLUDE.glob("*.h"))
    return any(d.stat().st_mtime > t for d in deps)


def refresh_generated() -> None:
    """csrc/plan_thunks.inc is generated from include/fs2hip.h (tools/gen_plan_thunks.py): rewritten here when the header
    has moved on, so that a changed prototype cannot meet a stale unpacking line (the compiler would refuse it anyway)."""
    gen = PKG.parent / "tools" / "gen_plan_thunks.py"
    if gen.exists() and subprocess.run([sys.executable, str(gen), "--check"]).returncode != 0:
        subprocess.run([sys.executable, str(gen)], check=True, stdout=subprocess.DEVNULL)


def build(force: bool = False, verbose: bool = False) -> Path:
    refresh_generated()
    if not force and not needs_build():
        return LIB
    objs = []
    obj_dir = PKG / "build"
    obj_dir.mkdir(exist_ok=True)
    procs = []
    for src in sources():
        obj = obj_dir / (src.stem + ".o")
        dep = obj_dir / (src.stem + ".d")
        stale = force or not obj.exists() or any(
            not d.exists() or d.stat().st_mtime > obj.stat().st_mtime for d in [src] + _deps(dep))
        objs.append(obj)
        if stale:
            cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", str(INCLUDE),
                   "-I", str(CSRC), "-MD", "-MF", str(dep), "-c", str(src), "-o", str(obj)]
            if os.environ.get("FS2_BUILD_PROBES"):  # phase-ablation instrumentation (FS2_GEMM_PROBE): timing builds only
                cmd.insert(1, "-DFS2_PROBES")
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{out.decode()}")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB)] + [str(o) for o in objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode()}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
