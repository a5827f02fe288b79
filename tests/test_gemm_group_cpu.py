"""CPU: the host side of grouped GEMM launches (``hip.gemm_group`` / ``hip._launch_group``) against a stand-in library --
which members share a launch, in what order the launches go out, what happens to members the grouped kernels cannot
take, that a refused group falls back to launches of its own, that the wrappers' follow-up work runs behind the launches,
and that a recorded launch plan keeps its own copy of a group's argument table.  The kernels themselves are
``tests/test_gemm_group_gpu.py``."""
import ctypes as C

import pytest
import torch  # noqa: F401  (before the library: one HIP runtime per process)

from fastspeech2_lightning_amd import hip as H


class FakeLib:
    def __init__(self, refuse_grouped=False):
        self.calls, self.refuse = [], refuse_grouped

    def fs2hip_gemm(self, ref, stream):
        a = ref._obj
        self.calls.append(("one", [(a.Mc, a.Nc, a.R)], a.tile, stream))
        return 0

    def fs2hip_gemm_grouped(self, arr, n, stream):
        if self.refuse:
            return -22
        self.calls.append(("group", [(arr[i].Mc, arr[i].Nc, arr[i].R) for i in range(n)], arr[0].tile, stream))
        return 0


def args(Mc, Nc, R, akc=1, bkc=1, **kw):
    a = H.GemmArgs()
    a.Mc, a.Nc, a.R, a.a_kcontig, a.b_kcontig, a.taps, a.splitk = Mc, Nc, R, akc, bkc, 1, 1
    for k, v in kw.items():
        setattr(a, k, v)
    return a


@pytest.fixture
def fake(monkeypatch):
    lib = FakeLib()
    monkeypatch.setattr(H, "lib", lambda: lib)
    monkeypatch.setattr(H, "_stream", lambda: 7)
    monkeypatch.setattr(H, "GEMM_TUNE", False)
    monkeypatch.setattr(H, "GEMM_GROUP", True)
    monkeypatch.setattr(H, "GEMM_PROFILE", None)
    saved = dict(H._TILE_CACHE)
    H._TILE_CACHE.clear()
    for k in H.GROUP_STATS:
        H.GROUP_STATS[k] = 0
    yield lib
    H._TILE_CACHE.update(saved)


def entries(*members):
    return [(a, True, 7) for a in members]


def test_members_are_partitioned_by_orientation_and_storage_in_call_order(fake):
    fwd = [args(4096, 256, 256) for _ in range(3)]
    wgrad = [args(256, 256, 4096, 0, 0, splitk=16) for _ in range(3)]
    dgrad = [args(4096, 256, 256, 1, 0) for _ in range(3)]
    stored = [args(8192, 256, 256, operand_bf16=4) for _ in range(2)]
    # the backward section of the predictors: weight gradient, data gradient, weight gradient, ...
    mixed = [m for pair in zip(wgrad, dgrad) for m in pair]
    H._launch_group(entries(*fwd, *mixed, *stored))
    kinds = [(c[0], len(c[1])) for c in fake.calls]
    assert kinds == [("group", 3), ("group", 3), ("group", 3), ("group", 2)]
    assert fake.calls[1][1][0] == (256, 256, 4096) and fake.calls[2][1][0] == (4096, 256, 256)
    assert all(c[3] == 7 for c in fake.calls)
    assert H.GROUP_STATS == {"launches": 4, "members": 11, "single": 0}


def test_more_than_eight_members_and_members_that_never_group(fake):
    many = [args(512, 64, 64) for _ in range(11)]
    conv = args(512, 64, 192, taps=3)
    dropped = args(512, 64, 64, drop_p=0.5)
    split_rounded = args(512, 64, 64, operand_bf16=2)   # "32-split" operands: no grouped instance
    counters = args(64, 64, 4096, 0, 0, splitk=8, counters=0x1000)
    H._launch_group(entries(*many[:5], conv, *many[5:], dropped, split_rounded, counters))
    kinds = [(c[0], len(c[1])) for c in fake.calls]
    # call order of the FIRST member of every launch: the run of eight, the convolution, the run of three, the rest
    assert kinds == [("group", 8), ("one", 1), ("group", 3), ("one", 1), ("one", 1), ("one", 1)]
    assert H.GROUP_STATS == {"launches": 2, "members": 11, "single": 4}


def test_a_refused_group_goes_out_one_by_one(fake):
    fake.refuse = True
    H._launch_group(entries(args(100, 64, 64), args(200, 64, 64)))
    assert [(c[0], c[1]) for c in fake.calls] == [("one", [(100, 64, 64)]), ("one", [(200, 64, 64)])]
    assert H.GROUP_STATS == {"launches": 0, "members": 0, "single": 2}


def test_a_single_member_is_a_plain_launch_and_streams_must_agree(fake):
    H._launch_group(entries(args(64, 64, 64)))
    assert [c[0] for c in fake.calls] == ["one"]
    with pytest.raises(Exception):
        H._launch_group([(args(64, 64, 64), True, 7), (args(64, 64, 64), True, 8)])


def test_sections_nest_switch_off_and_run_follow_ups_behind_the_launches(fake, monkeypatch):
    order = []
    monkeypatch.setattr(H, "_launch_group", lambda e: order.append(("launch", len(e))))
    with H.gemm_group() as outer:
        assert H._GROUP is outer
        with H.gemm_group() as inner:          # the outer section collects
            assert H._GROUP is outer and not inner.outer
            H._GROUP.entries.append((args(64, 64, 64), True, 7))
        assert H._GROUP is outer
        H._GROUP.entries.append((args(64, 64, 64), True, 7))
        H._GROUP.after.append(lambda: order.append("finish"))
    assert H._GROUP is None and order == [("launch", 2), "finish"]
    order.clear()
    with pytest.raises(ValueError):            # an exception inside: nothing is launched, the section is closed
        with H.gemm_group():
            H._GROUP.entries.append((args(64, 64, 64), True, 7))
            raise ValueError("x")
    assert H._GROUP is None and order == []
    monkeypatch.setattr(H, "GEMM_GROUP", False)
    with H.gemm_group():
        assert H._GROUP is None                # switched off: every wrapper launches at once


def test_group_tile_without_the_tuner_is_the_library_default(fake):
    members = [args(4096, 256, 256) for _ in range(3)]
    arr = (H.GemmArgs * 3)(*members)
    assert H._group_tile(arr, 3, members, 7) == 0
    assert set(H.GROUP_TILES) == {0, 4} and H.GROUP_MAX == 8


def test_a_recorded_plan_keeps_its_own_copy_of_the_member_table():
    from fastspeech2_lightning_amd import plan
    rec = plan.Recorder(main_stream=0)
    arr = (H.GemmArgs * 3)(args(1, 2, 3), args(4, 5, 6), args(7, 8, 9))
    rec.add(1, None, (arr, 3, 0), "fs2hip_gemm_grouped")
    arr[1].Mc = 99
    op, st, slots = rec.cmds[-1]
    kept = C.cast(slots[0], C.POINTER(H.GemmArgs * 3)).contents
    assert [(m.Mc, m.Nc, m.R) for m in kept] == [(1, 2, 3), (4, 5, 6), (7, 8, 9)] and slots[1] == 3


def test_group_tiles_travel_with_the_tile_table(monkeypatch):
    """``hip.tile_table`` / ``load_tile_table`` (FS2_GEMM_TILE_CACHE files, ``parallel.share_tile_table``): the tiles of
    grouped launches are part of the table, under a key of their own that old readers of single signatures never see."""
    import json
    monkeypatch.setattr(H, "_current_device", lambda: 0)   # (a signature names its device: no GPU on this box)
    saved, saved_g = dict(H._TILE_CACHE), dict(H._GROUP_TILE_CACHE)
    try:
        H._TILE_CACHE.clear(); H._GROUP_TILE_CACHE.clear()
        members = [args(4096, 256, 256) for _ in range(3)]
        key = tuple(H._tile_key(m) for m in members)
        H._TILE_CACHE[H._tile_key(members[0])] = 7
        assert "__groups__" not in H.tile_table()
        H._GROUP_TILE_CACHE[key] = 8
        t = json.loads(json.dumps(H.tile_table()))          # as a file / a broadcast object would carry it
        assert sum(1 for k in t if not k.startswith("__")) == 1 and len(t["__groups__"]) == 1
        H._TILE_CACHE.clear(); H._GROUP_TILE_CACHE.clear()
        H.load_tile_table(t)
        assert H._GROUP_TILE_CACHE == {key: 8} and list(H._TILE_CACHE.values()) == [7]
        arr = (H.GemmArgs * 3)(*members)
        assert H._group_tile(arr, 3, members, 7) == 8       # replayed without the tuner (GEMM_TUNE off in tests)
    finally:
        H._TILE_CACHE.clear(); H._TILE_CACHE.update(saved)
        H._GROUP_TILE_CACHE.clear(); H._GROUP_TILE_CACHE.update(saved_g)
