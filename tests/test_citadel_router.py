"""the CITADEL router-loss path (dpr_scale_amd/task/citadel_router.py) against a plain-torch
restatement of dpr_scale/task/citadel_task.py:137-153 (sim_score, both `pairwise` modes) and :249-262 (router_loss), and
the ragged multi-GPU gather (:79-135) on two gloo ranks.  CPU tests use the stand-in kernels; the gpu ones the HIP path at
the real router width d = 30522."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def ref_sim_score(query_repr, context_repr, mask=None, pairwise=False):  # citadel_task.py:137-153, verbatim semantics
    if pairwise:
        multiplier = context_repr.shape[0] // query_repr.shape[0]
        query_repr = query_repr.unsqueeze(1)
        mask = mask.view(-1, multiplier)
        context_repr = context_repr.view(-1, multiplier, context_repr.shape[1])
        scores = (query_repr * context_repr).sum(-1)
        if mask is not None:
            scores[mask] = float("-inf")
    else:
        scores = torch.matmul(query_repr, torch.transpose(context_repr, 0, 1))
        if mask is not None:
            scores[mask.repeat(scores.size(0), 1)] = float("-inf")
    return scores


def ref_router_loss(q, c, mask, pos, teacher, in_batch, teacher_coef, tau):  # citadel_task.py:249-262
    from dpr_scale_amd.task.citadel_router import distilled_loss

    loss = 0.0
    if 1 - teacher_coef > 0:
        s = ref_sim_score(q, c, mask, pairwise=not in_batch)
        if not in_batch:
            pos = torch.zeros(len(s), dtype=torch.int64)
        loss = torch.nn.CrossEntropyLoss()(s, pos)
    if teacher_coef > 0:
        ps = ref_sim_score(q, c, mask, pairwise=True)
        loss = (1 - teacher_coef) * loss + teacher_coef * distilled_loss(ps / tau, teacher / tau)
    return loss


def _inputs(B, M, d, seed, dev="cpu"):
    g = torch.Generator().manual_seed(seed)
    # router vectors: sparse non-negative activations over the vocabulary, bf16-representable
    s = 4.0 * d ** -0.5  # logits of order one at any width (a saturated softmax has no gradient to compare)
    q = (torch.relu(torch.randn(B, d, generator=g) - 1.0) * s).to(torch.bfloat16).float()
    c = (torch.relu(torch.randn(B * M, d, generator=g) - 1.0) * s).to(torch.bfloat16).float()
    c[torch.arange(B) * M] += q * 0.5
    c = c.to(torch.bfloat16).float()
    mask = torch.rand(B * M, generator=g) < 0.1
    mask[torch.arange(B) * M] = False
    pos = torch.arange(B) * M
    teacher = torch.randn(B, M, generator=g)
    return q.to(dev), c.to(dev), mask.to(dev), pos.to(dev), teacher.to(dev)


class _Task:
    def __init__(self, kernels, in_batch, teacher_coef, tau=1.0):
        from dpr_scale_amd.task.citadel_router import RouterScoring
        from dpr_scale_amd.task.dpr_task import HotCrossEntropyLoss

        class T(RouterScoring):
            pass

        self.t = T()
        self.t.kernels, self.t.in_batch, self.t.teacher_coef, self.t.tau = kernels, in_batch, teacher_coef, tau
        self.t.loss = HotCrossEntropyLoss(kernels)
        self.t.logged = {}
        self.t.log = lambda k, v, **kw: self.t.logged.__setitem__(k, v)


@pytest.mark.parametrize("in_batch,teacher_coef", [(True, 0.0), (False, 0.0), (True, 0.3), (False, 1.0)])
def test_router_loss_and_gradients_cpu_standin(in_batch, teacher_coef):
    from _oracle_kernels import OracleKernels

    B, M, d = 6, 4, 250
    q, c, mask, pos, teacher = _inputs(B, M, d, 3)
    task = _Task(OracleKernels(), in_batch, teacher_coef, tau=2.0).t
    tq, tc = q.clone().requires_grad_(True), c.clone().requires_grad_(True)
    loss = task.router_loss({"router_repr": tq}, {"router_repr": tc}, mask, pos, teacher)
    loss.backward()
    rq, rc = q.clone().requires_grad_(True), c.clone().requires_grad_(True)
    ref = ref_router_loss(rq, rc, mask, pos, teacher, in_batch, teacher_coef, 2.0)
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-4 * max(1.0, abs(ref.item()))
    assert (tq.grad - rq.grad).abs().max() <= 1e-2 * rq.grad.abs().max() + 1e-7
    assert (tc.grad - rc.grad).abs().max() <= 1e-2 * rc.grad.abs().max() + 1e-7
    assert "train_router_loss" in task.logged


@pytest.mark.gpu
@pytest.mark.parametrize("in_batch,teacher_coef", [(True, 0.0), (False, 0.0), (True, 0.5)])
def test_router_loss_hip_at_vocabulary_width(in_batch, teacher_coef):
    """d = 30522 (not a multiple of 8: zero-padded for the MFMA path, float2 streams for the pairwise kernels)."""
    B, M, d = 16, 4, 30522
    dev = torch.device("cuda:0")
    q, c, mask, pos, teacher = _inputs(B, M, d, 5)
    task = _Task(None, in_batch, teacher_coef, tau=1.5).t
    tq, tc = q.to(dev).requires_grad_(True), c.to(dev).requires_grad_(True)
    loss = task.router_loss({"router_repr": tq}, {"router_repr": tc}, mask.to(dev), pos.to(dev), teacher.to(dev))
    loss.backward()
    rq, rc = q.double().requires_grad_(True), c.double().requires_grad_(True)
    ref = ref_router_loss(rq, rc, mask, pos, teacher.double(), in_batch, teacher_coef, 1.5)
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-3 * max(1.0, abs(ref.item()))
    assert (tq.grad.cpu().double() - rq.grad).abs().max() <= 1e-2 * rq.grad.abs().max()
    assert (tc.grad.cpu().double() - rc.grad).abs().max() <= 1e-2 * rc.grad.abs().max()
    # scores themselves, both modes
    s0 = task.sim_score(tq.detach(), tc.detach(), mask.to(dev), pairwise=False).cpu()
    r0 = ref_sim_score(q.double(), c.double(), mask, pairwise=False)
    fin = torch.isfinite(r0)
    assert torch.equal(fin, torch.isfinite(s0)) and (s0[fin].double() - r0[fin]).abs().max() <= 1e-3 * r0[fin].abs().max()
    s1 = task.sim_score(tq.detach(), tc.detach(), mask.to(dev), pairwise=True).cpu()
    r1 = ref_sim_score(q.double(), c.double(), mask, pairwise=True)
    fin = torch.isfinite(r1)
    assert torch.equal(fin, torch.isfinite(s1)) and (s1[fin].double() - r1[fin]).abs().max() <= 1e-5 * r1[fin].abs().max()


def _gather_worker(rank, W, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from dpr_scale_amd.task.citadel_router import distributed_gather

    g = torch.Generator().manual_seed(10 + rank)
    B, M, V, dim = 3, 2, 11, 4
    Lq, Lc = 5 + rank, 7 - 2 * rank  # ragged token lengths across ranks
    qr = {"router_repr": torch.randn(B, V, generator=g).requires_grad_(True), "expert_repr": torch.randn(B, Lq, dim, generator=g).requires_grad_(True)}
    cr = {"router_repr": torch.randn(B * M, V, generator=g).requires_grad_(True), "expert_repr": torch.randn(B * M, Lc, dim, generator=g).requires_grad_(True)}
    mask = torch.rand(B * M, generator=g) < 0.3
    pos = torch.arange(B) * M
    teacher = torch.randn(B, M, generator=g)
    oq, oc, om, op, ot = distributed_gather(qr, cr, mask, pos, teacher, rank)
    # gradient flows only into this rank's own slices
    (oq["router_repr"].sum() + oc["expert_repr"].sum()).backward()
    q.put((rank, {k: v.detach().numpy() for k, v in oq.items()}, {k: v.detach().numpy() for k, v in oc.items()}, om.numpy(), op.numpy(),
           ot.numpy(), {k: v.detach().numpy() for k, v in qr.items()}, {k: v.detach().numpy() for k, v in cr.items()}, mask.numpy(),
           teacher.numpy(), qr["router_repr"].grad.numpy(), cr["expert_repr"].grad.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_ragged_distributed_gather_two_ranks():
    W = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, W, 29771, q)) for r in range(W)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(W)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    B, M = 3, 2

    def padcat(parts):  # what the reference builds: pad dim 1 to the longest, concatenate along dim 0
        if parts[0].ndim > 2:
            L = max(p.shape[1] for p in parts)
            parts = [np.concatenate([p, np.zeros((p.shape[0], L - p.shape[1]) + p.shape[2:], p.dtype)], 1) for p in parts]
        return np.concatenate(parts, 0)

    for r, oq, oc, om, op, ot, lq, lc, lm, lt, gq, gc in res:
        for k in oq:
            assert np.array_equal(oq[k], padcat([res[i][6][k] for i in range(W)])), k
        for k in oc:
            assert np.array_equal(oc[k], padcat([res[i][7][k] for i in range(W)])), k
        assert np.array_equal(om, np.concatenate([res[i][8] for i in range(W)]))
        assert np.array_equal(op, np.concatenate([np.arange(B) * M + i * B * M for i in range(W)]))  # running context offset
        assert np.array_equal(ot, np.concatenate([res[i][9] for i in range(W)]))
        assert np.all(gq == 1.0) and np.all(gc == 1.0)  # own slices carry gradient
