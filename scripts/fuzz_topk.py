#!/usr/bin/env python3
"""Randomised check of the streaming top-k entry points against torch's stable sort (run on an MI355X box):
shapes around the kernel's window / variant boundaries, k on both sides of 256, heavy ties, -inf columns, folds over
random piece boundaries (dprhot_topk_update) and candidate-list merges through dprhot_search.  Prints one line per failure and a
summary; exit status 1 on any mismatch."""
import argparse
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dpr_scale_amd.hotpath import HipKernels, CorpusSearch, sim_score  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    kn = HipKernels()
    g = torch.Generator().manual_seed(a.seed)
    bad = 0

    def rnd(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=g))

    cols_pool = [3, 70, 1000, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 8192, 8200, 12288, 16385, 40000, 70001]
    k_pool = [1, 2, 16, 100, 255, 256, 257, 300, 512, 1000, 1024]
    for case in range(a.cases):
        cols = cols_pool[rnd(0, len(cols_pool) - 1)] if rnd(0, 3) else rnd(1, 20000)
        k = min(k_pool[rnd(0, len(k_pool) - 1)], cols)
        rows = rnd(1, 9)
        mode = rnd(0, 5)
        if mode == 0:
            S = torch.randn(rows, cols, generator=g)
        elif mode == 1:
            S = torch.randint(0, rnd(1, 40), (rows, cols), generator=g).float()
        elif mode == 2:
            S = torch.randn(rows, cols, generator=g).to(torch.bfloat16).float()  # few distinct values, like bf16 scores
        elif mode == 3:
            S = torch.arange(cols, dtype=torch.float32).repeat(rows, 1) * (1 if rnd(0, 1) else -1)
        elif mode == 4:
            S = torch.full((rows, cols), float("-inf"))
            n = rnd(0, min(cols, 2000))
            idx = torch.randperm(cols, generator=g)[:n]
            S[:, idx] = torch.randn(rows, n, generator=g)
        else:
            S = torch.randn(rows, cols, generator=g) * 0.0 + torch.randint(0, 2, (rows, cols), generator=g).float() * 0.0  # +0 / all equal
            S[:, ::3] = -0.0
        S = S.to(dev)
        order = torch.sort(S, dim=1, descending=True, stable=True).indices[:, :k]
        v, i = kn.topk(S, k)
        ok = torch.equal(i, order) and torch.equal(v, S.gather(1, order))
        # folded over random pieces
        cuts = sorted(set([0, cols] + [rnd(0, cols) for _ in range(rnd(0, 4))]))
        v2, i2 = torch.empty_like(v), torch.empty_like(i)
        first = True
        okf = True
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            if hi > lo:
                kn.topk_update(S[:, lo:], hi - lo, 5 + lo, v2, i2, first)
                first = False
        okf = torch.equal(i2, order + 5) and torch.equal(v2, v)
        if not (ok and okf):
            bad += 1
            print(f"MISMATCH case {case}: rows={rows} cols={cols} k={k} mode={mode} whole={ok} folded={okf} cuts={cuts}")
    # candidate-list merges through the search entry point
    for case in range(12):
        nq, d = rnd(1, 300), 64 * rnd(1, 4)
        n = 8 * rnd(200, 9000)
        k = min(k_pool[rnd(3, len(k_pool) - 1)], n)
        chunk = 8 * rnd(max(k // 8 + 1, 64), 4096)
        q = torch.randn(nq, d, generator=g).to(dev)
        c = torch.randn(n, d, generator=g).to(dev)
        c[n // 3: n // 3 + 50] = c[10:60]  # duplicated passages: exact ties
        s = CorpusSearch(q, k, chunk=chunk, kernels=kn)
        s.add(c, 0)
        v, i = s.result()
        S = sim_score(q, c, None, 1.0, kn)
        order = torch.sort(S, dim=1, descending=True, stable=True).indices[:, :k]
        if not torch.equal(i, order):
            bad += 1
            print(f"SEARCH MISMATCH case {case}: nq={nq} n={n} d={d} k={k} chunk={chunk}")
    # k beyond the LDS-resident kernels: dprhot_topk_update_wide (csrc/wideselect.h) folded over random pieces, all value modes of above
    nwide = max(a.cases // 6, 20)
    for case in range(nwide):
        cols = rnd(4100, 90000)
        k = min(rnd(4097, 30000), cols)
        rows = rnd(1, 5)
        mode = rnd(0, 4)
        if mode == 0:
            S = torch.randn(rows, cols, generator=g)
        elif mode == 1:
            S = torch.randint(0, rnd(1, 40), (rows, cols), generator=g).float()  # thousands of ties, also at the k-th value
        elif mode == 2:
            S = torch.randn(rows, cols, generator=g).to(torch.bfloat16).float()
        elif mode == 3:
            S = torch.full((rows, cols), float("-inf"))
            n = rnd(0, cols)
            idx = torch.randperm(cols, generator=g)[:n]
            S[:, idx] = torch.randn(rows, n, generator=g)
        else:
            S = torch.zeros(rows, cols)
            S[:, ::3] = -0.0
        S = S.to(dev)
        order = torch.sort(S, dim=1, descending=True, stable=True).indices[:, :k]
        cuts = sorted(set([0, cols] + [rnd(0, cols) for _ in range(rnd(0, 4))]))
        v2 = torch.empty((rows, k), dtype=torch.float32, device=dev)
        i2 = torch.empty((rows, k), dtype=torch.int64, device=dev)
        ws = kn.topk_wide_workspace(rows, k, S)
        first = True
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            if hi > lo:
                kn.topk_update_wide(S[:, lo:], hi - lo, 7 + lo, v2, i2, first, ws)
                assert not bool(kn.topk_wide_errors(ws, rows).any())
                first = False
        if not (torch.equal(i2, order + 7) and torch.equal(v2, S.gather(1, order))):
            bad += 1
            print(f"WIDE MISMATCH case {case}: rows={rows} cols={cols} k={k} mode={mode} cuts={cuts}")
    print(f"fuzz_topk: {a.cases} top-k cases + 12 search cases + {nwide} wide-k cases, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
