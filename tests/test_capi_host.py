"""CPU-side checks of the C-ABI library: it loads, exports every symbol that
include/schpf_hip.h declares, refuses to compute without a GPU, and its host-side
plan builder stores every nonzero exactly once in a consistent layout."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, synthetic_counts
from schpf_amd import _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "schpf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(schpf_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), "libschpf_hip.so does not export %s" % name
    # and the binding table covers the header
    bound = set(_lib.SIGNATURES) | set(_lib.STRING_FUNCS)
    assert set(names) == bound


def test_version_and_error_string():
    lib = _lib.load()
    assert b"gfx950" in lib.schpf_version()
    status = lib.schpf_create(None, 0, None, 1, 10, 10, 2)
    assert status != 0
    assert b"NULL" in lib.schpf_last_error()


@pytest.mark.skipif(_lib.device_count() > 0, reason="only meaningful on a box without a GPU")
def test_no_memory_is_a_status_of_its_own():
    """The fall-back chains of the host code ("does not fit: try a smaller layout") react to the C ABI's
    SCHPF_ERR_NO_MEMORY status -- hipErrorOutOfMemory, a plan builder's failed hipMalloc, std::bad_alloc -- and to
    nothing else; the message text does not matter."""
    from schpf_amd import _lib
    header = open(os.path.join(ROOT, "include", "schpf_hip.h")).read()
    assert re.search(r"#define\s+SCHPF_ERR_NO_MEMORY\s+%d\b" % _lib.ERR_NO_MEMORY, header)
    assert _lib.is_out_of_memory(_lib.SchpfHipError("std::bad_alloc", _lib.ERR_NO_MEMORY))
    assert not _lib.is_out_of_memory(_lib.SchpfHipError("hipMalloc failed: out of memory", 1))
    assert not _lib.is_out_of_memory(ValueError("out of memory"))


def test_no_cpu_fallback():
    """Without a GPU the product raises; it never computes on the host."""
    from schpf_amd import hpf_hip, DeviceCAVI
    with pytest.raises(_lib.SchpfHipError):
        DeviceCAVI(10, 10, 2)
    with pytest.raises(_lib.SchpfHipError):
        hpf_hip.psi(1.0)


def expand(major, minor, val, n_major, n_minor, lpc, chunk_len, n_windows):
    lib = _lib.load()
    nnz = len(val)
    major = np.ascontiguousarray(major, np.int32)
    minor = np.ascontiguousarray(minor, np.int32)
    val = np.ascontiguousarray(val, np.float32)
    om = np.empty(nnz, np.int32); on = np.empty(nnz, np.int32); ov = np.empty(nnz, np.float32)
    onat = np.empty(nnz, np.int32); ow = np.empty(nnz, np.int32)
    cptr = np.empty(n_major + 1, np.int32)
    stats = (ctypes.c_int64 * 4)()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    _lib.check(lib.schpf_debug_plan_expand(nnz, p(major), p(minor), p(val), n_major, n_minor, lpc,
                                           chunk_len, n_windows, p(om), p(on), p(ov), p(onat), p(ow),
                                           p(cptr), stats))
    return om, on, ov, onat, ow, cptr, [int(s) for s in stats]


@pytest.mark.parametrize("lpc,chunk_len,n_windows", [(1, 16, 1), (2, 32, 2), (4, 8, 8), (8, 64, 16), (1, 2, 4)])
def test_plan_roundtrip(lpc, chunk_len, n_windows):
    X = synthetic_counts(257, 1031, 0.04, seed=5)
    perm = np.random.RandomState(0).permutation(X.nnz)       # unsorted COO on purpose
    row, col, val = X.row[perm], X.col[perm], X.data[perm].astype(np.float32)
    for major, minor, nM, nm in ((row, col, 257, 1031), (col, row, 1031, 257)):
        om, on, ov, onat, ow, cptr, stats = expand(major, minor, val, nM, nm, lpc, chunk_len, n_windows)
        # every nonzero exactly once
        key_in = np.sort(major.astype(np.int64) * nm + minor)
        key_out = np.sort(om.astype(np.int64) * nm + on)
        assert np.array_equal(key_in, key_out)
        order_in = np.lexsort((minor, major)); order_out = np.lexsort((on, om))
        assert np.array_equal(val[order_in], ov[order_out])
        # partial rows: chunk c belongs to exactly one major, ids of a major are cptr[m]..cptr[m+1]
        assert cptr[0] == 0 and cptr[-1] == stats[0]
        assert np.all(onat >= cptr[om]) and np.all(onat < cptr[om + 1])
        # chunk length bound and one window per chunk
        counts = np.bincount(onat, minlength=stats[0])
        assert counts.max() <= chunk_len and counts.min() >= 1
        wwidth = (nm + n_windows - 1) // n_windows
        win = on // wwidth
        first = np.full(stats[0], -1); first[onat] = win
        assert np.array_equal(first[onat], win)
        # XCD-aware schedule: workgroup b (4 waves) runs on XCD b % 8 and only serves windows
        # congruent to it modulo min(n_windows, 8)
        g = min(n_windows, 8)
        xcd = (ow // 4) % 8
        assert np.array_equal(xcd % g, win % g)
        assert stats[2] % 32 == 0


def test_plan_empty_rows_and_single_nonzero():
    om, on, ov, onat, ow, cptr, stats = expand([3], [2], [7.0], 6, 5, 1, 16, 1)
    assert (om[0], on[0], ov[0]) == (3, 2, 7.0)
    assert list(cptr) == [0, 0, 0, 0, 1, 1, 1] and stats[0] == 1


def expand_tile(major, minor, val, n_major, n_minor, lpc, wpb, win_rows, target_tasks, ring=1, slot_bytes=0):
    lib = _lib.load()
    nnz = len(val)
    major = np.ascontiguousarray(major, np.int32)
    minor = np.ascontiguousarray(minor, np.int32)
    val = np.ascontiguousarray(val, np.float32)
    om = np.empty(nnz, np.int32); on = np.empty(nnz, np.int32); ov = np.empty(nnz, np.float32)
    oprow = np.empty(nnz, np.int32); otask = np.empty(nnz, np.int32)
    pfirst = np.empty(n_major, np.int32); pcount = np.empty(n_major, np.int32)
    stats = (ctypes.c_int64 * 8)()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    _lib.check(lib.schpf_debug_tile_expand(nnz, p(major), p(minor), p(val), n_major, n_minor, lpc, wpb,
                                           win_rows, target_tasks, ring, slot_bytes, p(om), p(on), p(ov), p(oprow), p(otask),
                                           p(pfirst), p(pcount), stats))
    return om, on, ov, oprow, otask, pfirst, pcount, [int(s) for s in stats]


# (lpc, waves per block, rows per window, target tasks, slots, bytes per slot): 1 = window schedule;
# <= -2 = half-window schedule with that many slots, whose rows per sub-window follow from the slot
# (160-byte rows)
@pytest.mark.parametrize("lpc,wpb,win_rows,tasks,ring,slot_bytes",
                         [(4, 8, 64, 64, 1, 0), (2, 4, 37, 1, 1, 0), (1, 1, 1000, 7, 1, 0), (8, 2, 5, 1000, 1, 0),
                          (16, 8, 300, 16, 1, 0),
                          (2, 4, 0, 16, -2, 8192), (1, 1, 0, 1, -2, 1024), (2, 16, 0, 1000, -2, 77824), (4, 8, 0, 64, -3, 4000)])
@pytest.mark.parametrize("coo_order,bank_order,taper", [("shuffled", 2, 0), ("row-major", 2, 40), ("col-major", 2, 0), ("shuffled", 1, 0),
                                                        ("shuffled", 0, 0), ("shuffled", 2, 40), ("col-major", 2, 80)])
def test_tile_plan_roundtrip(lpc, wpb, win_rows, tasks, ring, slot_bytes, coo_order, bank_order, taper, monkeypatch):
    monkeypatch.setenv("SCHPF_BANK_ORDER", str(bank_order))   # order inside a segment: minor / per row / jointly per LDS pass
    monkeypatch.setenv("SCHPF_TAPER", str(taper))             # window ranges of unequal length (per cent)
    X = synthetic_counts(257, 1031, 0.04, seed=5)
    # the plan builder has fast paths for input already sorted by (row, col) / (col, row)
    perm = {"shuffled": np.random.RandomState(0).permutation(X.nnz),
            "row-major": np.lexsort((X.col, X.row)), "col-major": np.lexsort((X.row, X.col))}[coo_order]
    row, col, val = X.row[perm], X.col[perm], X.data[perm].astype(np.float32)
    for major, minor, nM, nm in ((row, col, 257, 1031), (col, row, 1031, 257)):
        om, on, ov, oprow, otask, pfirst, pcount, st = expand_tile(major, minor, val, nM, nm, lpc, wpb, win_rows, tasks,
                                                                   ring, slot_bytes)
        n_tasks, n_blocks, n_windows, pstride, slots, wpt = st[:6]
        if abs(ring) > 1:
            win_rows = (slot_bytes - 64) // 160     # rows of a ring slot (its last 64 bytes stay free); the hook itself checks that every entry of
            #                                  an epoch lies in a slot that is readable during that epoch
        key_in = np.sort(major.astype(np.int64) * nm + minor)
        key_out = np.sort(om.astype(np.int64) * nm + on)
        assert np.array_equal(key_in, key_out)                       # every nonzero exactly once
        oi = np.lexsort((minor, major)); oo = np.lexsort((on, om))
        assert np.array_equal(val[oi], ov[oo])
        # a nonzero's partial row is one of its major's strided partial rows, and the right one
        j = (oprow - pfirst[om]) // pstride
        assert np.array_equal(pfirst[om] + j * pstride, oprow)
        assert np.all(j >= 0) and np.all(j < pcount[om])
        assert n_windows == -(-nm // win_rows)
        n_ranges = -(-n_windows // wpt)
        assert n_tasks == n_blocks * n_ranges
        win = on // win_rows
        if taper == 0:
            assert np.array_equal(j, win // wpt)                     # task range <-> window: equal cuts
        else:                                                        # tapered: ranges are intervals of windows, in order
            assert np.array_equal(j, otask // n_blocks)
            lo = np.full(n_ranges, n_windows); hi = np.full(n_ranges, -1)
            np.minimum.at(lo, j, win); np.maximum.at(hi, j, win)
            seen = hi >= 0
            assert np.all(hi[seen][:-1] < lo[seen][1:])
        # partial rows are exclusive to one major
        owner = np.full(n_tasks * (64 // lpc) * wpb, -1)
        owner[oprow] = om
        assert np.array_equal(owner[oprow], om)


@pytest.mark.parametrize("lpc,wpb,win_rows", [(4, 8, 64), (1, 1, 1000), (4, 16, 9), (2, 4, 37)])
@pytest.mark.parametrize("balanced", [0, 1])
@pytest.mark.parametrize("coo_order", ["shuffled", "col-major"])
def test_single_step_counts_roundtrip(lpc, wpb, win_rows, balanced, coo_order, monkeypatch):
    """`single` step counts (plan.h, round 5): the steps of a (wave, window) count nonzeros, an odd count stops in the
    middle of its last stored slot.  With or without the balancing, through the kernel's own walk (the hook executes
    exactly `steps` slot halves) every nonzero must come back exactly once, in a partial row of its major."""
    monkeypatch.setenv("SCHPF_DEBUG_SINGLE", "1")
    monkeypatch.setenv("SCHPF_DEBUG_BALANCE", str(balanced))
    X = synthetic_counts(257, 1031, 0.04, seed=5)
    perm = {"shuffled": np.random.RandomState(0).permutation(X.nnz), "col-major": np.lexsort((X.row, X.col))}[coo_order]
    row, col, val = X.row[perm], X.col[perm], X.data[perm].astype(np.float32)
    for major, minor, nM, nm in ((row, col, 257, 1031), (col, row, 1031, 257)):
        om, on, ov, oprow, otask, pfirst, pcount, st = expand_tile(major, minor, val, nM, nm, lpc, wpb, win_rows, 16)
        n_tasks, n_blocks, n_windows, pstride, slots = st[:5]
        key_in = major.astype(np.int64) * nm + minor
        key_out = om.astype(np.int64) * nm + on
        oi = np.argsort(key_in, kind="stable"); oo = np.argsort(key_out, kind="stable")
        assert np.array_equal(key_in[oi], key_out[oo])               # every nonzero exactly once
        assert np.array_equal(val[oi], ov[oo])
        j = (oprow - pfirst[om]) // pstride
        assert np.array_equal(pfirst[om] + j * pstride, oprow)
        assert np.all(j >= 0) and np.all(j < pcount[om])
        owner = np.full(n_tasks * (64 // lpc) * wpb, -1)
        owner[oprow] = om
        assert np.array_equal(owner[oprow], om)                      # partial rows are exclusive to one major


@pytest.mark.parametrize("lpc,row_slots", [(2, 10), (1, 5), (4, 28)])
@pytest.mark.parametrize("ring,slot_bytes,win_rows", [(-2, 77824, 0), (1, 0, 900)])
def test_joint_bank_order_has_the_fewest_modelled_conflicts(lpc, row_slots, ring, slot_bytes, win_rows, monkeypatch):
    """The three orders inside a segment hold the same nonzeros (round trip above); under the LDS model of
    plan.cpp (the lane groups of a pass read one table row each, rows of one class are served one after the
    other) the per-row rotation must beat minor order and the joint assignment must beat both -- at the shipped
    shapes: f64 K = 20 (two lanes per 160-byte row), f32 K = 20 (one lane per 80-byte row), f64 K = 50."""
    X = synthetic_counts(6000, 1500, 0.05, seed=8)
    monkeypatch.setenv("SCHPF_DEBUG_ROW_SLOTS", str(row_slots))
    if ring == 1:
        win_rows = win_rows * 10 // row_slots
    extra = {}
    for order in (0, 1, 2):
        monkeypatch.setenv("SCHPF_BANK_ORDER", str(order))
        for major, minor, nM, nm in ((X.row, X.col, 6000, 1500), (X.col, X.row, 1500, 6000)):
            st = expand_tile(major, minor, X.data.astype(np.float32), nM, nm, lpc, 16, win_rows, 4, ring, slot_bytes)[-1]
            extra[order] = extra.get(order, 0.0) + st[7] / st[6] / 2
    assert extra[2] < 0.8 * extra[1] < 0.8 * extra[0], extra


@pytest.mark.parametrize("lpc,wpb,win_rows,tasks", [(4, 8, 64, 64), (2, 4, 37, 1), (1, 1, 1000, 7), (8, 2, 5, 1000), (1, 16, 11, 16)])
@pytest.mark.parametrize("coo_order", ["shuffled", "row-major", "duplicates"])
def test_balanced_windows_roundtrip(lpc, wpb, win_rows, tasks, coo_order, monkeypatch):
    """Balanced windows (plan.h): every block deals the minor rows to its windows itself and the plan is built on that
    virtual numbering.  Through the block's row list every nonzero must come back exactly once, in a partial row of its
    major, and a virtual row must belong to one window of its section only."""
    monkeypatch.setenv("SCHPF_DEBUG_BALANCE", "1")
    X = synthetic_counts(257, 1031, 0.04, seed=5)
    row, col, val = X.row, X.col, X.data.astype(np.float32)
    if coo_order == "shuffled":
        perm = np.random.RandomState(0).permutation(X.nnz)
        row, col, val = row[perm], col[perm], val[perm]
    if coo_order == "duplicates":      # a COO may hold an entry twice (the reference sums them when it converts)
        row = np.concatenate([row, row[:500]]); col = np.concatenate([col, col[:500]]); val = np.concatenate([val, val[:500] + 1])
    for major, minor, nM, nm in ((row, col, 257, 1031), (col, row, 1031, 257)):
        om, on, ov, oprow, otask, pfirst, pcount, st = expand_tile(major, minor, val, nM, nm, lpc, wpb, win_rows, tasks)
        n_tasks, n_blocks, n_windows, pstride = st[:4]
        key_in = major.astype(np.int64) * nm + minor
        key_out = om.astype(np.int64) * nm + on
        oi = np.lexsort((val, key_in)); oo = np.lexsort((ov, key_out))
        assert np.array_equal(key_in[oi], key_out[oo])
        assert np.array_equal(val[oi], ov[oo])
        j = (oprow - pfirst[om]) // pstride
        assert np.array_equal(pfirst[om] + j * pstride, oprow)
        assert np.all(j >= 0) and np.all(j < pcount[om])
        assert n_windows == -(-nm // win_rows)


def test_balanced_windows_store_fewer_slots(monkeypatch):
    """The point of the balancing: at the C5 share's shape (7 nonzeros per row and window, 16 rows per wave) the
    lock-step padding shrinks -- the stored step slots fall by more than a fifth -- and at C3's shape (seven windows
    here: one section) whole balanced windows store about what the half-window schedule stores, with half the barriers."""
    monkeypatch.setenv("SCHPF_DEBUG_ROW_SLOTS", "28")
    X = synthetic_counts(2048, 4000, 0.02, seed=3)
    v = X.data.astype(np.float32)
    slots = {}
    for bal in ("0", "1"):
        monkeypatch.setenv("SCHPF_DEBUG_BALANCE", bal)
        slots[bal] = sum(expand_tile(M, m, v, nM, nm, 4, 16, 347, 1)[-1][4]
                         for M, m, nM, nm in ((X.row, X.col, 2048, 4000), (X.col, X.row, 4000, 2048)))
    assert slots["1"] < 0.8 * slots["0"], slots
    monkeypatch.setenv("SCHPF_DEBUG_ROW_SLOTS", "10")
    X = synthetic_counts(2048, 6000, 0.05, seed=4)
    v = X.data.astype(np.float32)
    monkeypatch.setenv("SCHPF_DEBUG_BALANCE", "0")
    half = expand_tile(X.row, X.col, v, 2048, 6000, 2, 16, 0, 1, -2, 77824)[-1][4]
    whole = expand_tile(X.row, X.col, v, 2048, 6000, 2, 16, 972, 1)[-1][4]
    monkeypatch.setenv("SCHPF_DEBUG_BALANCE", "1")
    balanced = expand_tile(X.row, X.col, v, 2048, 6000, 2, 16, 972, 1)[-1][4]
    assert balanced < 0.87 * whole and balanced < 1.05 * half, (balanced, half, whole)


def test_tile_plan_tiny():
    om, on, ov, oprow, otask, pfirst, pcount, st = expand_tile([3], [2], [7.0], 6, 5, 4, 8, 64, 2048)
    assert (om[0], on[0], ov[0]) == (3, 2, 7.0) and st[0] == 1 and pcount[3] == 1
