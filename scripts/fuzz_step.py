#!/usr/bin/env python3
"""Randomised check of the training step (hotpath.inbatch_contrastive_loss, forward + backward through autograd) against the
reference's formulation in torch ops (dpr_task.py:197-212 on the bf16-rounded embeddings, fp32 math): shapes on both sides of every
plan boundary (fused small step, skinny step, wide vectors, no-logits forward, phase-interleaved backward), ragged sizes, masks,
temperatures.  Bars as in tests/test_gpu_parity.py: loss <= 1e-3 relative, gradients <= 1e-2 of their maximum.  Run on an MI355X box."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpr_scale_amd.hotpath import inbatch_contrastive_loss  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=120)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--few-rows", action="store_true", help="shapes of the few-rows plan WITHOUT its dScores launch (csrc/skinny.h, round 4), taken "
                                                             "wherever that plan exists (option sk_fused = 2): B in 32..128, thousands of contexts")
    a = ap.parse_args()
    if a.few_rows:
        from dpr_scale_amd import _lib
        _lib.set_option("sk_fused", 2)
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(a.seed)

    def pick(pool):
        return pool[int(torch.randint(0, len(pool), (1,), generator=g))]

    Bs = [1, 3, 8, 31, 32, 33, 64, 100, 128, 129, 256, 300, 512, 1024, 2048]
    Ks = [1, 2, 3, 8, 9, 16, 33, 64]
    ds = [8, 64, 80, 128, 256, 768, 1024, 1032, 4096]
    bad, worst = 0, (0.0, 0.0, 0.0)
    if a.few_rows:
        Bs, Ks, ds = [32, 64, 96, 128], [33, 40, 64, 65, 72, 100, 128], [128, 256, 384, 768, 1024]
    for case in range(a.cases):
        B, K, d = pick(Bs), pick(Ks), pick(ds)
        if a.few_rows and (B * K > 16384 or B * K < 4096):
            continue
        if B * K > 32768 or B * B * K > (1 << 27) or B * K * d > (1 << 27):
            continue
        T = float(pick([0.05, 0.3, 1.0, 2.0]))
        if os.environ.get("FUZZ_VERBOSE"):
            print(f"case {case}: B={B} K={K} d={d} T={T}", flush=True)
        n = B * K
        q = (torch.randn(B, d, generator=g) * d ** -0.25).to(dev)
        c = (torch.randn(n, d, generator=g) * d ** -0.25).to(dev)
        y = (torch.arange(B) * K + torch.randint(0, K, (B,), generator=g)).to(dev)
        m = (torch.rand(n, generator=g) < pick([0.0, 0.05, 0.3])).to(dev)
        m[y] = False
        q1, c1 = q.clone().requires_grad_(True), c.clone().requires_grad_(True)
        loss = inbatch_contrastive_loss(q1, c1, y, m, T, False)
        loss.backward()
        def reference(dt):
            qb = q.to(torch.bfloat16).to(dt).requires_grad_(True)
            cb = c.to(torch.bfloat16).to(dt).requires_grad_(True)
            S = (qb @ cb.t()).masked_fill(m[None, :], float("-inf")) / T
            ref = torch.nn.functional.cross_entropy(S, y)
            ref.backward()
            return ref.item(), qb.grad.double(), cb.grad.double(), S.detach()

        r32, q32, c32, S = reference(torch.float32)
        r64, q64, c64, _ = reference(torch.float64)
        # Bars relative to the fp64 value of the reference's formulation.  Where the softmax saturates (gold logit far above the
        # rest: loss ~ 1e-5, gradients ~ 1e-5 / B) fp32 is cancellation on BOTH sides (lse - gold with |lse| ~ 20-60: one ulp is
        # 2-4e-6) and the reference's own fp32 run is off by up to 100 %: there the bar is a small multiple of the reference's own
        # fp32 error.
        smax = S[torch.isfinite(S)].abs().max().item()
        el = max(abs(loss.item() - r64) - 4e-7 * max(smax, 1.0), 0.0) / max(abs(r64), 1e-6)
        dq_den, dc_den = q64.abs().max().clamp_min(1e-30), c64.abs().max().clamp_min(1e-30)
        eq = ((q1.grad.double() - q64).abs().max() / dq_den).item()
        ec = ((c1.grad.double() - c64).abs().max() / dc_den).item()
        eq = 0.0 if eq <= 4 * ((q32 - q64).abs().max() / dq_den).item() else eq
        ec = 0.0 if ec <= 4 * ((c32 - c64).abs().max() / dc_den).item() else ec
        worst = (max(worst[0], el), max(worst[1], eq), max(worst[2], ec))
        if os.environ.get("FUZZ_VERBOSE") or max(eq, ec) > 5e-3:  # (half the bar: worth a line)
            print(f"  case {case}: B={B} K={K} d={d} T={T}: loss rel {el:.2e}, dQ {eq:.2e}, dC {ec:.2e}", flush=True)
        if not (el <= 1e-3 and eq <= 1e-2 and ec <= 1e-2) or not torch.isfinite(loss):
            bad += 1
            print(f"MISMATCH case {case}: B={B} K={K} d={d} T={T}: loss rel {el:.2e}, dQ {eq:.2e}, dC {ec:.2e}")
    print(f"fuzz_step: {a.cases} cases, {bad} outside the bars; worst loss rel {worst[0]:.2e}, dQ {worst[1]:.2e}, dC {worst[2]:.2e} (of max)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
