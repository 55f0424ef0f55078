#!/usr/bin/env python
"""Simulation gate for the K = 50 proposal of the round-4 verdict (item 3): ONE-nonzero step slots for the wide-row
kernels + balanced HALF windows, the next half window staged under the steps of the current one.

The quantity that decides it is the SLOT FILL: the share of executed step slots that carry a nonzero (a padding
slot executes every instruction of a nonzero).  A (wave, window) lasts as long as its fullest row, so

    fill = nonzeros / sum over (wave, window) of  slots_per_step * steps(wave, window) * rows_per_wave

with steps = ceil(max_row_load / slots_per_step).  Modelled exactly as the plan builders cut the matrix
(schpf_amd/csrc/plan.cpp): major rows sorted by length and dealt to blocks of 256 (16 waves x 16 lane groups at four
lanes per row), windows of `win_rows` minor rows; balanced windows by the builders' greedy (plan.cpp balance_section:
in minor order, every minor row to the window of its section -- at most 32 windows -- where the fullest of the block's
rows that hold it stays lowest, ties by the sum of their loads, then the lowest window).  Variants:

  index / 2   whole 152 KiB windows cut by index, two nonzeros per slot        (round 3; measured fill 0.56)
  bal / 2     balanced whole windows, two per slot                             (round 4, shipped; measured 0.81)
  bal / 1     balanced whole windows, ONE nonzero per slot
  half / 1    balanced HALF windows, one per slot, no work-ahead               (plain double buffering: slot B is
                                                                               being filled while slot A is computed)
  half+ / 1   ... with the work-ahead schedule of the half-window plans: a row that has finished the epoch's own
              half window works ahead in the next one (already resident), every wave its own step count per epoch
              -- an UPPER bound for any schedule in which the next half window is staged under the current steps,
              because then it is not yet readable for (part of) the epoch

Gate (verdict): proceed only if the best half-window variant reaches fill >= 0.72 on the C5 share.

    python tools/sim_k50_half_windows.py [--blocks 16] [--cells 125000 --genes 25000 --density 0.02]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def balance_block(minor_sorted, group_sorted, n_minor, win_rows, gpb, dmax=32):
    """plan.cpp balance_section over the sections of one block.  minor_sorted / group_sorted: the block's nonzeros
    by ascending minor row.  Returns the window of every nonzero."""
    W = (n_minor + win_rows - 1) // win_rows
    n_sections = max(1, (W + dmax - 1) // dmax)
    D = (W + n_sections - 1) // n_sections
    window_of = np.empty(minor_sorted.shape[0], dtype=np.int32)
    starts = np.flatnonzero(np.r_[True, minor_sorted[1:] != minor_sorted[:-1]])
    ends = np.r_[starts[1:], minor_sorted.shape[0]]
    sec_of = (minor_sorted[starts] // win_rows) // D
    for s in range(int(sec_of.max()) + 1 if starts.size else 0):
        sel = np.flatnonzero(sec_of == s)
        w0 = s * D
        Dn = min(W, w0 + D) - w0
        load = np.zeros((gpb, Dn), dtype=np.int32)
        cnt = np.zeros(Dn, dtype=np.int32)
        for i in sel:
            g = group_sorted[starts[i]:ends[i]]
            sub = load[g]                                   # (rows holding it, windows)
            cost = (sub.max(0).astype(np.int64) << 40) | (sub.sum(0).astype(np.int64) << 8) | np.arange(Dn)
            cost[cnt >= win_rows] = np.iinfo(np.int64).max
            c = int(np.argmin(cost))
            load[g, c] += 1
            cnt[c] += 1
            window_of[starts[i]:ends[i]] = w0 + c
    return window_of


def fills(load, gpw, slots_per_step):
    """load: [rows of the block][windows] nonzero counts -> executed slots of the block (lock step per wave)."""
    gpb, W = load.shape
    per_wave = load.reshape(gpb // gpw, gpw, W).max(1)          # fullest row of every (wave, window)
    steps = -(-per_wave // slots_per_step)
    return int(steps.sum()) * slots_per_step * gpw


def work_ahead_slots(load, gpw):
    """One nonzero per slot, two resident half windows: in epoch e a row first does what is left of its own half
    window e, then works ahead in e + 1; a wave's epoch lasts as long as the largest REMAINING need of its rows."""
    gpb, W = load.shape
    total = 0
    for v in range(gpb // gpw):
        n = load[v * gpw:(v + 1) * gpw].astype(np.int64)
        ahead = np.zeros(gpw, dtype=np.int64)
        for e in range(W):
            need = n[:, e] - ahead
            steps = int(need.max()) if need.size else 0
            total += steps * gpw
            spare = steps - need
            ahead = np.minimum(n[:, e + 1], spare) if e + 1 < W else np.zeros(gpw, dtype=np.int64)
    return total


def side_report(name, major, minor, n_major, n_minor, args, rng):
    gpw, wpb = 64 // args.lpc, 16
    gpb = gpw * wpb
    row_bytes = args.row_bytes
    win = (152 * 1024) // row_bytes
    half = (76 * 1024 - 64) // row_bytes
    lengths = np.bincount(major, minlength=n_major)
    order = np.argsort(-lengths, kind="stable")                 # rows by decreasing length, blocks of gpb
    n_blocks = (n_major + gpb - 1) // gpb
    pick = np.sort(rng.choice(n_blocks - 1, size=min(args.blocks, n_blocks - 1), replace=False))
    block_of_row = np.full(n_major, -1, dtype=np.int64)
    group_of_row = np.zeros(n_major, dtype=np.int64)
    for b in pick:
        rows = order[b * gpb:(b + 1) * gpb]
        block_of_row[rows] = b
        group_of_row[rows] = np.arange(rows.shape[0])
    keep = block_of_row[major] >= 0
    mj, mn = major[keep], minor[keep]
    acc = {k: [0, 0] for k in ("index / 2", "bal / 2", "bal / 1", "half-index / 1", "half / 1", "half+ / 1")}
    for b in pick:
        sel = np.flatnonzero(block_of_row[mj] == b)
        g = group_of_row[mj[sel]]
        m = mn[sel]
        o = np.argsort(m, kind="stable")
        g, m = g[o], m[o]
        nnz = m.shape[0]

        def load_matrix(window_of, W):
            ld = np.zeros((gpb, W), dtype=np.int32)
            np.add.at(ld, (g, window_of), 1)
            return ld
        Ww, Wh = (n_minor + win - 1) // win, (n_minor + half - 1) // half
        ld_index = load_matrix(m // win, Ww)
        ld_bal = load_matrix(balance_block(m, g, n_minor, win, gpb), Ww)
        ld_hidx = load_matrix(m // half, Wh)
        ld_half = load_matrix(balance_block(m, g, n_minor, half, gpb), Wh)
        for key, slots in (("index / 2", fills(ld_index, gpw, 2)), ("bal / 2", fills(ld_bal, gpw, 2)),
                           ("bal / 1", fills(ld_bal, gpw, 1)), ("half-index / 1", fills(ld_hidx, gpw, 1)),
                           ("half / 1", fills(ld_half, gpw, 1)), ("half+ / 1", work_ahead_slots(ld_half, gpw))):
            acc[key][0] += nnz
            acc[key][1] += slots
    print("%s side: %d blocks of %d rows sampled, windows of %d rows (half: %d), %.1f nonzeros per row and window (%.1f per half)"
          % (name, len(pick), gpb, win, half, mj.shape[0] / float(len(pick) * gpb) / ((n_minor + win - 1) // win),
             mj.shape[0] / float(len(pick) * gpb) / ((n_minor + half - 1) // half)))
    for key, (nnz, slots) in acc.items():
        print("    %-16s slot fill %.3f" % (key, nnz / float(slots)))
    return {k: v[0] / float(v[1]) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=125000)
    ap.add_argument("--genes", type=int, default=25000)
    ap.add_argument("--density", type=float, default=0.02)
    ap.add_argument("--lpc", type=int, default=4, help="lanes per row (K = 50 f64: 4)")
    ap.add_argument("--row-bytes", type=int, default=448, help="table row in LDS (K = 50 f64: 7 x 4 x 16)")
    ap.add_argument("--blocks", type=int, default=16, help="blocks sampled per orientation")
    args = ap.parse_args()
    from bench import synthetic_block
    X = synthetic_block(args.cells, args.genes, args.density, seed=42)
    print("matrix %d x %d, nnz %d (bench.py generator A, seed 42: the C5 share when left at the defaults)" % (X.shape[0], X.shape[1], X.nnz))
    rng = np.random.RandomState(1)
    c = side_report("cell", X.row.astype(np.int64), X.col.astype(np.int64), X.shape[0], X.shape[1], args, rng)
    g = side_report("gene", X.col.astype(np.int64), X.row.astype(np.int64), X.shape[1], X.shape[0], args, rng)
    best = max(0.5 * (c[k] + g[k]) for k in ("half / 1", "half+ / 1"))
    print("best half-window variant, mean of both orientations: %.3f -> gate (>= 0.72): %s"
          % (best, "PASSED" if best >= 0.72 else "FAILED"))


if __name__ == "__main__":
    main()
