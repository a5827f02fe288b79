"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/fs2hip.h
declares (no compute calls without a GPU); the binding table matches the header."""
import ctypes
import re
from pathlib import Path

import pytest
import torch  # noqa: F401  (before the library: one HIP runtime per process)

REPO = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def built():
    from fastspeech2_lightning_amd import build
    return build.build()


def header_symbols():
    text = (REPO / "include" / "fs2hip.h").read_text()
    return sorted(set(re.findall(r"\bint\s+(fs2hip_\w+)\s*\(", text)))


def test_header_and_binding_agree():
    from fastspeech2_lightning_amd import hip
    assert header_symbols() == sorted(hip.EXPORTS)


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(str(built))
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert lib.fs2hip_version() >= 1


def test_argument_counts_match_header():
    from fastspeech2_lightning_amd import hip
    text = (REPO / "include" / "fs2hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, sig in hip.SIGNATURES.items():
        if sig is None:
            continue
        m = re.search(r"\bint\s+" + name + r"\s*\((.*?)\)\s*;", text, flags=re.S)
        assert m, name
        args = [a.strip() for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        assert len(args) == len(sig), (name, args, sig)
        for a, c in zip(args, sig):
            if c == "p":
                assert "*" in a, (name, a)
            elif c == "f":
                assert a.startswith("float ") and "*" not in a, (name, a)
            elif c == "i":
                assert a.startswith("int ") and "*" not in a, (name, a)
            elif c == "q":
                assert a.startswith("long long "), (name, a)
            elif c == "Q":
                assert a.startswith("unsigned long long ") and "*" not in a, (name, a)


def test_gemm_struct_layout_matches_header():
    from fastspeech2_lightning_amd import hip
    text = (REPO / "include" / "fs2hip.h").read_text()
    body = re.search(r"typedef struct \{(.*?)\} Fs2GemmArgs;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        parts = decl.replace("*", " ").split(",")
        names.append(parts[0].split()[-1])
        names += [p.strip() for p in parts[1:]]
    assert names == [f[0] for f in hip.GemmArgs._fields_]


def test_product_refuses_cpu_tensors(built):
    import torch
    from fastspeech2_lightning_amd import hip
    with pytest.raises(RuntimeError, match="no CPU path"):
        hip.layernorm_fwd(torch.zeros(4, 8), torch.ones(8), torch.zeros(8))


def test_documents_quote_the_real_entry_point_count():
    """the number of C-ABI entry points quoted in DESIGN.md / README.md is ``len(hip.EXPORTS)`` (which
    ``test_library_exports_every_declared_symbol`` ties to the header and the built library), not a hand-kept figure."""
    import re
    from pathlib import Path
    from fastspeech2_lightning_amd import hip
    root = Path(__file__).resolve().parent.parent
    for name in ("DESIGN.md", "README.md"):
        quoted = re.findall(r"(\d+) entry points", (root / name).read_text())
        assert quoted, name
        assert all(int(q) == len(hip.EXPORTS) for q in quoted), (name, quoted, len(hip.EXPORTS))
