"""N>1 host logic on CPU: world_size 2 and 4 gloo processes run dpr_scale_amd.hotpath.InBatchContrastive
(with the test stand-in for the HIP kernels) and must reproduce, per rank, the loss / q.grad / c.grad that
the REFERENCE's DDP branch produced on real gloo ranks (tests/golden/w2_ddp.npz, w4_ddp.npz)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


def _worker(rank, W, port, meta, q, overlapped=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from _oracle_kernels import OracleKernels
    from oracle.inbatch_oracle import synth_embeddings
    from dpr_scale_amd.hotpath import ContextGather, defer_context_grad, inbatch_contrastive_loss

    qv, cv, y, m = synth_embeddings(meta["seed"] + rank, meta["B"], meta["K"], meta["d"], meta["dist"], meta["ragged"])
    tq = torch.from_numpy(qv).requires_grad_(True)
    tc = torch.from_numpy(cv).requires_grad_(True)
    kn = OracleKernels()
    if overlapped:
        # the order DenseRetrieverTask.training_step uses on several GPUs: context rows first, gather in flight under the
        # "query tower" (here an identity op that stands for it), reduce-scatter waited for by the deferred-grad node
        c1 = tc * 1.0
        c2, pending = defer_context_grad(c1)
        g = ContextGather(c2, torch.from_numpy(m), None, kn)
        q1 = tq * 1.0
        loss = inbatch_contrastive_loss(q1, c2, torch.from_numpy(y), torch.from_numpy(m), meta["T"], None, kn, g, pending)
    else:
        loss = inbatch_contrastive_loss(tq, tc, torch.from_numpy(y), torch.from_numpy(m), meta["T"], None, kn)
    (loss * 3.0).backward()  # grad_output != 1 exercises the device-scalar path
    q.put((rank, loss.item(), tq.grad.numpy() / 3.0, tc.grad.numpy() / 3.0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,port,overlapped", [("w2_ddp", 29711, False), ("w4_ddp", 29712, False), ("w2_ddp", 29713, True),
                                                  ("w4_ddp", 29714, "bf16wire"), ("w4_ddp", 29715, "allpairs"),
                                                  ("w2_ddp", 29716, "allpairs_bf16wire")])
def test_ddp_branch_matches_reference(name, port, overlapped, monkeypatch):
    """overlapped == "bf16wire": DPRHOT_DC_WIRE=bf16 -- the reduce-scatter of the dC partials ships bf16.
    "allpairs": DPRHOT_PATH_COLLECTIVES=allpairs -- all-gather / reduce-scatter as direct all-pairs exchanges."""
    if isinstance(overlapped, str):
        if "bf16wire" in overlapped:
            monkeypatch.setenv("DPRHOT_DC_WIRE", "bf16")  # inherited by the spawned ranks
        if "allpairs" in overlapped:
            monkeypatch.setenv("DPRHOT_PATH_COLLECTIVES", "allpairs")
        overlapped = True
    meta, g = load_golden(name)
    W = meta["W"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, W, port, meta, q, overlapped)) for r in range(W)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=90) for _ in range(W)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r, loss, dq, dc in res:
        assert abs(loss - g["loss_per_rank"][r]) <= 1e-3 * max(1.0, abs(g["loss_per_rank"][r]))
        # G travels as bf16 (2^-9 relative rounding per element), hence the 1e-2-of-max bar on gradients
        assert np.abs(dq - g["dq_per_rank"][r]).max() <= 1e-2 * np.abs(g["dq_per_rank"][r]).max()
        assert np.abs(dc - g["dc_per_rank"][r]).max() <= 1e-2 * np.abs(g["dc_per_rank"][r]).max()


def test_single_process_world():
    """W == 1 path (no process group): same operator, same stand-in, against the cfg2 fixture."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle_kernels import OracleKernels
    from oracle.inbatch_oracle import synth_embeddings
    from dpr_scale_amd.hotpath import inbatch_contrastive_loss

    meta, g = load_golden("cfg2_Ur_T0.05")
    qv, cv, y, m = synth_embeddings(meta["seed"], meta["B"], meta["K"], meta["d"], meta["dist"], meta["ragged"])
    tq = torch.from_numpy(qv).requires_grad_(True)
    tc = torch.from_numpy(cv).requires_grad_(True)
    loss = inbatch_contrastive_loss(tq, tc, torch.from_numpy(y), torch.from_numpy(m), meta["T"], None, OracleKernels())
    loss.backward()
    assert abs(loss.item() - g["loss"]) <= 1e-3 * max(1.0, abs(g["loss"]))
    assert np.abs(tq.grad.numpy() - g["dQ"]).max() <= 1e-2 * np.abs(g["dQ"]).max()
    assert np.abs(tc.grad.numpy() - g["dC"]).max() <= 1e-2 * np.abs(g["dC"]).max()


def test_product_path_refuses_cpu_tensors():
    """No CPU fallback: the default (HIP) kernels must raise on CPU tensors instead of computing."""
    from dpr_scale_amd.hotpath import inbatch_contrastive_loss

    q = torch.zeros(4, 128, requires_grad=True)
    c = torch.zeros(8, 128, requires_grad=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        inbatch_contrastive_loss(q, c, torch.arange(4) * 2, torch.zeros(8, dtype=torch.bool))


def _task_worker(rank, W, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from _oracle_kernels import OracleKernels
    from dpr_scale_amd import hotpath, hydra_compat, lightning_compat
    from dpr_scale_amd.task.dpr_task import DenseRetrieverTask

    cfg = hydra_compat.compose(os.path.join(ROOT, "dpr_scale_amd", "conf"), "tiny_cpu.yaml", ["task.shared_model=false"])
    started = []

    class CountingGather(hotpath.ContextGather):
        def __init__(self, *a, **k):
            started.append(1)
            super().__init__(*a, **k)

    hotpath.ContextGather = CountingGather
    out = {}
    for mode in ("overlapped", "plain", "switch"):
        torch.manual_seed(0)  # same weights on every rank and in both modes
        task = hydra_compat.instantiate(cfg.task, _recursive_=False)
        task.kernels = OracleKernels()
        task.trainer = lightning_compat.Trainer(device="cpu")  # world size 2 -> DDPStrategy marker
        task.setup("fit")
        dm = hydra_compat.instantiate(cfg.datamodule, seed=100 + rank)
        batch = dm.train_dataloader()[0]
        if mode == "plain":  # a subclass overriding forward() takes the reference's call order
            class Plain(DenseRetrieverTask):
                def forward(self, query_ids, contexts_ids):
                    return self.encode_queries(query_ids), self.encode_contexts(contexts_ids)
            task.__class__ = Plain
        if mode == "switch":  # the documented knob for seed-for-seed runs: reference order on the base class
            task.context_tower_first = False
        loss = task.training_step(batch, 0)
        loss.backward()
        grads = torch.cat([p.grad.flatten() for p in task.parameters() if p.grad is not None])
        out[mode] = (loss.item(), grads)
    assert len(started) == 1, started  # the early gather ran in exactly one of the three modes
    assert out["switch"][0] == out["plain"][0] and torch.equal(out["switch"][1], out["plain"][1])
    q.put((rank, out["overlapped"][0], out["plain"][0], (out["overlapped"][1] - out["plain"][1]).abs().max().item(),
           out["plain"][1].abs().max().item()))
    dist.barrier()
    dist.destroy_process_group()


def test_task_training_step_overlapped_order_equals_plain_order():
    """DenseRetrieverTask.training_step on 2 ranks: context tower first + early all-gather + deferred reduce-scatter
    (what runs under DDP) gives the same loss and the same encoder gradients as the reference's call order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_task_worker, args=(r, 2, 29721, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    assert abs(res[0][1] - res[1][1]) <= 1e-6  # the loss is the global mean: identical on both ranks
    for r, lo, lp, gdiff, gmax in res:
        assert abs(lo - lp) <= 1e-6 * max(1.0, abs(lp))
        assert gmax > 0 and gdiff <= 1e-6 * gmax


def _gpu_worker(rank, W, port, meta, q, wire="fp32"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["DPRHOT_DC_WIRE"] = wire
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from oracle.inbatch_oracle import synth_embeddings
    from dpr_scale_amd.hotpath import ContextGather, defer_context_grad, inbatch_contrastive_loss

    dev = torch.device("cuda", 0)  # every rank on the one GPU of the box; gloo moves the bytes (RCCL refuses that)
    qv, cv, y, m = synth_embeddings(meta["seed"] + rank, meta["B"], meta["K"], meta["d"], meta["dist"], meta["ragged"])
    tq = torch.from_numpy(qv).to(dev).requires_grad_(True)
    tc = torch.from_numpy(cv).to(dev).requires_grad_(True)
    ty, tm = torch.from_numpy(y).to(dev), torch.from_numpy(m).to(dev)
    c1 = tc * 1.0
    c2, pending = defer_context_grad(c1)
    g = ContextGather(c2, tm, None)
    q1 = tq * 1.0
    loss = inbatch_contrastive_loss(q1, c2, ty, tm, meta["T"], None, None, g, pending)
    (loss * 3.0).backward()
    torch.cuda.synchronize()
    q.put((rank, loss.item(), tq.grad.cpu().numpy() / 3.0, tc.grad.cpu().numpy() / 3.0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_two_ranks_through_libdprhot_on_one_gpu_match_reference_ddp(wire):
    """The whole multi-rank operator -- pack kernel, early all-gather, one-call step (sim + softmax + both backward GEMMs, gradients
    scaled for the expected grad_output and fixed up in backward), reduce-scatter in either wire format, deferred context gradient
    (half-width wire: widened into the fp32 gradient after the wait) -- with the real HIP kernels, two processes on one device over
    gloo, against the reference's own 2-rank DDP fixture."""
    meta, g = load_golden("w2_ddp")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, 29731 if wire == "fp32" else 29733, meta, q, wire)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r, loss, dq, dc in res:
        assert abs(loss - g["loss_per_rank"][r]) <= 1e-3 * max(1.0, abs(g["loss_per_rank"][r]))
        assert np.abs(dq - g["dq_per_rank"][r]).max() <= 1e-2 * np.abs(g["dq_per_rank"][r]).max()
        assert np.abs(dc - g["dc_per_rank"][r]).max() <= 1e-2 * np.abs(g["dc_per_rank"][r]).max()


def _gpu_worker_cfg3(rank, W, port, meta, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from oracle.inbatch_oracle import synth_embeddings
    from dpr_scale_amd.hotpath import ContextGather, defer_context_grad, inbatch_contrastive_loss

    dev = torch.device("cuda", 0)
    qv, cv, y, m = synth_embeddings(meta["seed"] + rank, meta["B"], meta["K"], meta["d"], meta["dist"], meta["ragged"])
    tq = torch.from_numpy(qv).to(dev).requires_grad_(True)
    tc = torch.from_numpy(cv).to(dev).requires_grad_(True)
    ty, tm = torch.from_numpy(y).to(dev), torch.from_numpy(m).to(dev)
    c1 = tc * 1.0
    c2, pending = defer_context_grad(c1)
    g = ContextGather(c2, tm, None)
    q1 = tq * 1.0
    loss = inbatch_contrastive_loss(q1, c2, ty, tm, meta["T"], None, None, g, pending)
    (loss * 2.0).backward()
    torch.cuda.synchronize()
    own = rank == meta["own_rank"]
    dc = tc.grad.cpu().numpy() / 2.0
    q.put((rank, loss.item(), tq.grad.cpu().numpy() / 2.0 if own else None, dc[:64] if own else None,
           dc.sum(1) if own else None, dc.sum(0)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_eight_ranks_cfg3_operator_on_one_gpu_match_reference():
    """inbatch_contrastive_loss (the autograd operator DenseRetrieverTask.training_step calls under DDP) at cfg3 --
    W8 B128 K8 d768, 1024 x 8192 global, ragged dummy contexts -- eight processes sharing the one GPU of the box over
    gloo, every rank running the production packed step (dprhot_inbatch_step_packed_f32) and the real reduce-scatter,
    against the reference's global step (dpr_task.py:163-212 through oracle/ref_shim.py)."""
    meta, g = load_golden("cfg3_Ur_T1")
    W = meta["W"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker_cfg3, args=(r, W, 29741, meta, q)) for r in range(W)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(W)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
    colsum = np.zeros(meta["d"], np.float64)
    for r, loss, dq, dc_head, dc_rowsum, dc_colsum in res:
        assert abs(loss - g["loss"]) <= 1e-3 * max(1.0, abs(g["loss"]))
        colsum += dc_colsum
        if r == meta["own_rank"]:
            assert np.abs(dq - g["dq_own"]).max() <= 1e-2 * np.abs(g["dq_own"]).max()
            assert np.abs(dc_head - g["dc_own_head"]).max() <= 1e-2 * np.abs(g["dc_own_head"]).max()
            assert np.abs(dc_rowsum - g["dc_own_rowsum"]).max() <= 1e-2 * np.abs(g["dc_own_rowsum"]).max()
            assert np.abs(dc_colsum - g["dc_own_colsum"]).max() <= 1e-2 * np.abs(g["dc_own_colsum"]).max()
    # the global column sum is ~0 (rows of softmax - onehot sum to zero): absolute bar at the scale of one rank's chunk
    assert np.abs(colsum - g["dC_colsum"]).max() <= 1e-2 * W * np.abs(g["dc_own_colsum"]).max()


def _probe_worker(rank, W, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("DPRHOT_PATH_COLLECTIVES", None)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from dpr_scale_amd import dist as D

    res = {}
    # (1) ranks that measured differently must still leave with ONE answer: the slowest rank decides for each form
    local = {0: [(10.0, 5.0), (100.0, 50.0), (10.0, float("inf"))], 1: [(10.0, 20.0), (90.0, 60.0), (10.0, 1.0)]}[rank % 2]
    res["decisions"] = [D.decide_topology(a, b)[0] for a, b in local]
    # (2) the probe itself (CPU tensors over gloo: both forms run), then the collectives follow what it registered
    topo = D.choose_path_collectives(torch.device("cpu"), 16, 32, iters=3)
    res["probe"] = topo
    res["registered"] = D.path_topology()
    send = torch.full((16, 32), float(rank + 1)).to(torch.bfloat16)
    out = torch.empty((W * 16, 32), dtype=torch.bfloat16)
    D.all_gather_rows(send, out)
    res["gather_ok"] = bool(all(torch.all(out[r * 16:(r + 1) * 16].float() == r + 1) for r in range(W)))
    part = torch.arange(W * 16 * 32, dtype=torch.float32).reshape(W * 16, 32) * (rank + 1)
    mine = torch.empty((16, 32))
    D.reduce_scatter_rows(part, mine)
    want = torch.arange(W * 16 * 32, dtype=torch.float32).reshape(W * 16, 32)[rank * 16:(rank + 1) * 16] * sum(range(1, W + 1))
    res["scatter_ok"] = bool(torch.equal(mine, want))
    # (3) an explicit configure() outranks the probe; clearing it gives the probe's answer back
    D.configure(topology="allpairs" if topo == "rccl" else "rccl")
    res["override"] = D.path_topology()
    D.configure(None, None)
    res["back"] = D.path_topology()
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_topology_probe_agrees_across_ranks():
    """dist.decide_topology / choose_path_collectives (what DenseRetrieverTask.on_pretrain_routine_start runs before the first
    step): whatever the ranks measured locally, every rank leaves with the same form of the path's collectives."""
    W = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_probe_worker, args=(r, W, 29751, q)) for r in range(W)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(W))
    for p in procs:
        p.join(timeout=60)
    assert res[0]["decisions"] == res[1]["decisions"] == ["rccl", "allpairs", "rccl"]
    assert res[0]["probe"] == res[1]["probe"] and res[0]["probe"] in ("rccl", "allpairs")
    for r in range(W):
        assert res[r]["registered"] == res[r]["probe"] == res[r]["back"] and res[r]["override"] != res[r]["probe"]
        assert res[r]["gather_ok"] and res[r]["scatter_ok"]


def _probe2_worker(rank, W, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("DPRHOT_PATH_COLLECTIVES", None)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from dpr_scale_amd import dist as D

    issued = {"a2a": 0}
    real_a2a = dist.all_to_all_single

    def counting(*a, **k):
        issued["a2a"] += 1
        return real_a2a(*a, **k)

    dist.all_to_all_single = counting
    res = {}
    cpu = torch.device("cpu")
    # (a) a candidate ONE rank cannot run (its local pre-flight fails there) is abandoned by EVERY rank before anybody issues one of its
    #     collectives: no all-to-all leaves any rank, the form arrives as inf and loses, the ranks agree
    topo = D.choose_path_collectives(cpu, 16, 32, iters=2, preflight=lambda form, w: not (form == "allpairs" and rank == W - 1))
    rec = D._PROBED[D._gkey(None)]
    res["a"] = (topo, issued["a2a"], rec["us"]["allpairs"] == float("inf"), rec["us"]["rccl"] < float("inf"))
    # (c) the probe chooses the WIRE too: both wires timed in both forms, one answer on every rank
    D._PROBED.clear()
    issued["a2a"] = 0
    topo = D.choose_path_collectives(cpu, 16, 32, iters=2, wires=(torch.float32, torch.bfloat16))
    rec = D._PROBED[D._gkey(None)]
    res["c"] = (topo, str(D.path_wire()), sorted(rec["us_by_wire"]), issued["a2a"] > 0)
    # (d) a form pinned by hand is not measured: the other form is never issued
    D._PROBED.clear()
    issued["a2a"] = 0
    D.configure(topology="rccl")
    topo = D.choose_path_collectives(cpu, 16, 32, iters=2)
    res["d"] = (topo, issued["a2a"], D._PROBED[D._gkey(None)]["pinned"])
    D.configure(None, None)
    # (b) the all-pairs reduce-scatter adds the W received chunks in fp32 in RANK ORDER, one rounding at the end -- bit for bit, on a
    #     bf16 wire too (what dprhot_grad_sum_shards does on the device; test_direct_comm.py holds the device-side equality)
    D._PROBED.clear()
    D.configure(topology="allpairs")
    g = torch.Generator().manual_seed(100 + rank)
    for wire in (torch.float32, torch.bfloat16):
        part = (torch.randn(W * 8, 16, generator=g) * 3).to(wire)
        mine = torch.empty((8, 16), dtype=wire)
        D.reduce_scatter_rows(part, mine)
        allp = [torch.empty_like(part) for _ in range(W)]
        half = wire == torch.bfloat16  # (gloo moves no bf16: the bit patterns travel as fp16)
        dist.all_gather([t.view(torch.float16) if half else t for t in allp], part.view(torch.float16) if half else part)
        acc = allp[0][rank * 8:(rank + 1) * 8].float().clone()
        for k in range(1, W):
            acc += allp[k][rank * 8:(rank + 1) * 8].float()
        res["b_" + str(wire)] = bool(torch.equal(mine, acc.to(wire)))
    D.configure(None, None)
    dist.all_to_all_single = real_a2a
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("W,port", [(2, 29761), (4, 29762)])
def test_probe_abandons_a_form_together_chooses_the_wire_and_skips_pinned_forms(W, port):
    """VERDICT r5 #5 / ADVICE r5 (dist.choose_path_collectives): (a) no rank leaves a form while its peers are inside it -- a one-sided
    pre-flight failure is agreed on BEFORE any collective of that form is issued; (b) the torch.distributed all-pairs reduce-scatter is the
    rank-order fp32 sum the C-ABI form computes; (c) the probe chooses the dC wire as well; (d) a pinned form is not measured."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_probe2_worker, args=(r, W, port, q)) for r in range(W)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(W))
    for p in procs:
        p.join(timeout=60)
    for r in range(W):
        assert res[r]["a"] == ("rccl", 0, True, True), res[r]["a"]
        assert res[r]["c"][:3] == res[0]["c"][:3] and res[r]["c"][1] in ("torch.float32", "torch.bfloat16") and res[r]["c"][3]
        assert res[r]["c"][2] == ["bfloat16", "float32"]
        assert res[r]["d"] == ("rccl", 0, True)
        assert res[r]["b_torch.float32"] and res[r]["b_torch.bfloat16"]
