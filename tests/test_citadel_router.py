"""the CITADEL router-loss path (dpr_scale_amd/task/citadel_router.py) against the reference's own
citadel_task.py -- through tests/golden/router_*.npz, which oracle/make_golden.py wrote by running
/root/reference/dpr_scale/task/citadel_task.py (MultiVecRetrieverTask.sim_score :137-153, router_loss :249-262,
distilled_loss :240-247, distributed_gather :97-135) unmodified, and through oracle/router_oracle.py, the restatement those
fixtures pin.  Nothing expected here is computed by the product.  CPU tests drive the product's orchestration with the
stand-in kernels; the gpu ones the HIP path at the real router width d = 30522."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden
from oracle import router_oracle as R

CASES = ["router_inbatch", "router_pairwise", "router_teacher", "router_teacher_only"]


def _fixture(name):
    meta, z = load_golden(name)
    q, c, mask, pos, teacher = R.synth_router(meta["seed"], meta["B"], meta["M"], meta["d"])
    return meta, z, (q, c, mask, pos, teacher)


def _check_grads(meta, z, dq, dc, tol):
    """dq / dc [rows, d] (numpy) against the fixture's strided columns and full row / column sums; tol relative to max |grad|."""
    st = meta["col_stride"]
    for name, g in (("dq", dq), ("dc", dc)):
        g = np.asarray(g, np.float64)
        ref_cols = z[name + "_cols"].astype(np.float64)
        scale = max(np.abs(ref_cols).max(), 1e-30)
        assert np.abs(g[:, ::st] - ref_cols).max() <= tol * scale, name
        # full row / column sums (every element takes part): tol relative to the largest sum of magnitudes
        assert np.abs(g.sum(1) - z[name + "_rowsum"]).max() <= tol * np.abs(g).sum(1).max(), name
        assert np.abs(g.sum(0) - z[name + "_colsum"]).max() <= tol * np.abs(g).sum(0).max() + 1e-6 * scale, name


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


@pytest.mark.parametrize("name", CASES)
def test_router_oracle_is_pinned_by_the_reference_fixtures(name):
    meta, z, (q, c, mask, pos, teacher) = _fixture(name)
    loss, dq, dc = R.router_step(q, c, mask, pos, teacher, meta["in_batch"], meta["teacher_coef"], meta["tau"])
    assert abs(loss - float(z["loss"])) <= 2e-6 * max(1.0, abs(float(z["loss"])))
    _check_grads(meta, z, dq, dc, 2e-6)
    t = torch.from_numpy
    for pairwise, key in ((False, "S_dense"), (True, "S_pair")):
        S = R.sim_score(t(q).double(), t(c).double(), t(mask), pairwise).numpy()
        fin = np.isfinite(z[key])
        assert np.array_equal(fin, np.isfinite(S)) and np.abs(S[fin] - z[key][fin]).max() <= 2e-6 * np.abs(z[key][fin]).max()


@pytest.mark.parametrize("name", CASES)
def test_router_loss_and_gradients_cpu_standin(name):
    from _oracle_kernels import OracleKernels

    meta, z, (q, c, mask, pos, teacher) = _fixture(name)
    task = _Task(OracleKernels(), meta["in_batch"], meta["teacher_coef"], meta["tau"]).t
    t = torch.from_numpy
    tq, tc = t(q).requires_grad_(True), t(c).requires_grad_(True)
    loss = task.router_loss({"router_repr": tq}, {"router_repr": tc}, t(mask), t(pos), t(teacher))
    loss.backward()
    assert abs(loss.item() - float(z["loss"])) <= 1e-4 * max(1.0, abs(float(z["loss"])))
    _check_grads(meta, z, tq.grad.numpy(), tc.grad.numpy(), 1e-2)  # dScores travel as bf16 on the dense path
    assert "train_router_loss" in task.logged


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_router_loss_hip_at_vocabulary_width(name):
    """d = 30522 (not a multiple of 8: zero-padded for the MFMA path, float2 streams for the pairwise kernels) against
    the reference's own outputs."""
    dev = torch.device("cuda:0")
    meta, z, (q, c, mask, pos, teacher) = _fixture(name)
    task = _Task(None, meta["in_batch"], meta["teacher_coef"], meta["tau"]).t
    t = lambda a: torch.from_numpy(a).to(dev)
    tq, tc = t(q).requires_grad_(True), t(c).requires_grad_(True)
    loss = task.router_loss({"router_repr": tq}, {"router_repr": tc}, t(mask), t(pos), t(teacher))
    loss.backward()
    assert abs(loss.item() - float(z["loss"])) <= 1e-3 * max(1.0, abs(float(z["loss"])))  # north_star: 1e-3 relative
    tol = 1e-2 if meta["teacher_coef"] < 1 else 1e-4   # the cross-entropy's dScores are bf16 (2^-9); the teacher-only loss is fp32 throughout
    _check_grads(meta, z, tq.grad.cpu().numpy(), tc.grad.cpu().numpy(), tol)
    s0 = task.sim_score(tq.detach(), tc.detach(), t(mask), pairwise=False).cpu().numpy()
    fin = np.isfinite(z["S_dense"])
    assert np.array_equal(fin, np.isfinite(s0)) and np.abs(s0[fin] - z["S_dense"][fin]).max() <= 1e-3 * np.abs(z["S_dense"][fin]).max()
    s1 = task.sim_score(tq.detach(), tc.detach(), t(mask), pairwise=True).cpu().numpy()
    fin = np.isfinite(z["S_pair"])
    assert np.array_equal(fin, np.isfinite(s1)) and np.abs(s1[fin] - z["S_pair"][fin]).max() <= 1e-5 * np.abs(z["S_pair"][fin]).max()


@pytest.mark.gpu
def test_router_loss_at_the_bench_shape_against_the_router_oracle():
    """The shape bench.py's `router` block times -- 128 queries x 1024 contexts x 30528 vocabulary columns -- against the pinned
    restatement of citadel_task.py:249-262 (oracle/router_oracle.py, fp64 on the host): in-batch cross-entropy of the router vectors
    and its gradients.  (The four reference-written fixtures are small batches; this is the size the timing is quoted at.)"""
    from dpr_scale_amd import hotpath

    dev = torch.device("cuda:0")
    B, M, d = 128, 8, 30528
    q, c, mask, pos, teacher = R.synth_router(11, B, M, d=d)
    want_loss, want_dq, want_dc = R.router_step(q, c, mask, pos, teacher, in_batch=True, teacher_coef=0.0)
    t = lambda a: torch.from_numpy(a).to(dev)
    tq, tc = t(q).requires_grad_(True), t(c).requires_grad_(True)
    loss = hotpath.inbatch_contrastive_loss(tq, tc, t(pos), t(mask), 1.0, group=False)
    loss.backward()
    assert abs(loss.item() - want_loss) <= 1e-3 * max(1.0, abs(want_loss))
    for got, want in ((tq.grad.cpu().numpy(), want_dq), (tc.grad.cpu().numpy(), want_dc)):
        assert np.abs(got - want).max() <= 1e-2 * np.abs(want).max()


def _gather_worker(rank, W, port, seed, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from dpr_scale_amd.task.citadel_router import distributed_gather
    from oracle.router_oracle import synth_gather_rank

    qr, cr, mask, pos, teacher = synth_gather_rank(seed, rank)
    qr = {k: torch.from_numpy(v).requires_grad_(True) for k, v in qr.items()}
    cr = {k: torch.from_numpy(v).requires_grad_(True) for k, v in cr.items()}
    oq, oc, om, op, ot = distributed_gather(qr, cr, torch.from_numpy(mask), torch.from_numpy(pos), torch.from_numpy(teacher), rank)
    (oq["router_repr"].sum() + oc["expert_repr"].sum()).backward()  # gradient flows only into this rank's own slices
    q.put((rank, {k: v.detach().numpy() for k, v in oq.items()}, {k: v.detach().numpy() for k, v in oc.items()}, om.numpy(),
           op.numpy(), ot.numpy(), qr["router_repr"].grad.numpy(), cr["expert_repr"].grad.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_ragged_distributed_gather_two_ranks_equals_the_reference():
    """citadel_task.py:97-135 run by the reference on 2 gloo ranks (router_gather_w2.npz) == the product's gather, bit for bit."""
    meta, z = load_golden("router_gather_w2")
    W = meta["W"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, W, 29771, meta["seed"], q)) for r in range(W)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(W)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r, oq, oc, om, op, ot, gq, gc in res:
        for k, v in oq.items():
            assert np.array_equal(v, z[f"r{r}_q_{k}"]), k
        for k, v in oc.items():
            assert np.array_equal(v, z[f"r{r}_c_{k}"]), k
        assert np.array_equal(om, z[f"r{r}_mask"]) and np.array_equal(op, z[f"r{r}_pos"]) and np.array_equal(ot, z[f"r{r}_teacher"])
        assert np.all(gq == 1.0) and np.all(gc == 1.0)
