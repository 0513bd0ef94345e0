#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz from the reference's own code.

Run in the build container (needs /root/reference):   python oracle/make_golden.py
Every output below is produced by ``/root/reference/dpr_scale/task/dpr_task.py`` executed unmodified
through ``oracle/ref_shim.py`` (fp32, CPU) on seeded, bf16-representable inputs from
``oracle.inbatch_oracle.synth_embeddings`` -- the fixtures therefore pin the oracle AND the HIP path to the
reference.  Inputs are not stored when they can be regenerated from (seed, B, K, d, dist, ragged); the
numpy PCG64 stream is version-stable.

Cases (SURVEY.md section 8 notation, BASELINE.json configs):
  cfg1  W1 B4  K2 d128        full tensors                      (configs[0] shape)
  cfg2  W1 B32 K8 d768        full tensors, 3 variants           (configs[1], the bench workload)
  cfg3  W8 B128 K8 d768       global step, summaries + samples   (configs[2])
  cfg4  W8 B8 K8 d768         REAL 8-rank gloo DDP branch        (configs[3] as written)
  cfg5  W8 B64 K2 d1024       global step, summaries             (configs[4])
  w2/w4 real 2-/4-rank gloo DDP-branch runs (per-rank loss, q.grad, c.grad)
  ties / nib / topk           rank metrics with ties, in_batch_negatives=False branch, top-k
  router_*                    CITADEL router loss (citadel_task.py:137-153, :240-262) at d = 30522, both sim_score modes
  router_gather_w2            citadel_task.py:97-135 distributed_gather on 2 real gloo ranks, ragged token lengths
  search_ref                  run_retrieval_pytorch.py:141-176 search_index (fp16 einsum + topk, the batched loop and its tail)
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402
from oracle.inbatch_oracle import synth_embeddings, synth_search  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED0 = 1234


def rank_inputs(seed, W, B, K, d, distn, ragged):
    return [synth_embeddings(seed + r, B, K, d, distn, ragged) for r in range(W)]


def global_inputs(seed, W, B, K, d, distn, ragged):
    parts = rank_inputs(seed, W, B, K, d, distn, ragged)
    Q = np.concatenate([p[0] for p in parts])
    C = np.concatenate([p[1] for p in parts])
    y = np.concatenate([p[2] + r * B * K for r, p in enumerate(parts)])
    m = np.concatenate([p[3] for p in parts])
    return Q, C, y, m


def run_global(Q, C, y, m, T):
    """Reference training_step (non-DDP branch) on the concatenated tensors + sim_score + rank metrics."""
    tQ, tC = torch.from_numpy(Q), torch.from_numpy(C)
    ty, tm = torch.from_numpy(y), torch.from_numpy(m)
    loss, dq, dc = ref_shim.reference_training_step(tQ, tC, ty, tm, temperature=T)
    S = ref_shim.reference_sim_score(tQ, tC, tm)  # dpr_task.py:98-105 (before /T)
    rank_sum, mrr_sum, score = ref_shim.reference_rank_metrics(S, ty, k=1)
    # per-row rank under the frozen stable tie rule, from torch itself
    order = torch.sort(S, dim=1, descending=True, stable=True).indices
    ranks = (order == ty[:, None]).nonzero()[:, 1] + 1
    S_T = S / T
    lse = torch.logsumexp(S_T, dim=1)
    return dict(loss=np.float32(loss.item()), dQ=dq.numpy(), dC=dc.numpy(), S=S.numpy(), lse=lse.numpy(),
                ranks=ranks.numpy().astype(np.int64),
                rank_metrics=np.array([rank_sum, mrr_sum, float(score)], dtype=np.float64))


ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]  # optional: fixture-name prefixes to (re)generate


def wanted(name):
    return not ONLY or any(name.startswith(o) for o in ONLY)


def save(name, meta, **arrays):
    if not wanted(name):
        return
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB  {meta}")


def _ddp_worker(rank, W, port, seed, B, K, d, distn, ragged, T, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    qv, cv, y, m = synth_embeddings(seed + rank, B, K, d, distn, ragged)
    loss, dq, dc = ref_shim.reference_training_step(
        torch.from_numpy(qv), torch.from_numpy(cv), torch.from_numpy(y), torch.from_numpy(m),
        temperature=T, distributed=True, rank=rank)
    q.put((rank, loss.item(), dq.numpy(), dc.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def run_ddp(W, seed, B, K, d, distn, ragged, T, port):
    """The reference DDP branch (dpr_task.py:165-195) on W real gloo ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, W, port, seed, B, K, d, distn, ragged, T, q))
             for r in range(W)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(W)], key=lambda t: t[0])
    for p in procs:
        p.join()
    return (np.array([r[1] for r in res], dtype=np.float32),
            np.stack([r[2] for r in res]), np.stack([r[3] for r in res]))


def _gather_worker(rank, W, port, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from oracle.router_oracle import synth_gather_rank

    qr, cr, mask, pos, teacher = synth_gather_rank(seed, rank)
    t = lambda d: {k: torch.from_numpy(v).requires_grad_(True) for k, v in d.items()}
    oq, oc, om, op, ot = ref_shim.reference_distributed_gather(
        t(qr), t(cr), torch.from_numpy(mask), torch.from_numpy(pos), torch.from_numpy(teacher), rank)
    q.put((rank, {k: v.detach().numpy() for k, v in oq.items()}, {k: v.detach().numpy() for k, v in oc.items()},
           om.numpy(), op.numpy(), ot.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def router_fixtures():
    """f4: /root/reference/dpr_scale/task/citadel_task.py (MultiVecRetrieverTask, unmodified) on seeded router-shaped inputs.
    dq / dc are [16, 30522] / [64, 30522]: stored as every 31st column plus full row and column sums."""
    from oracle.router_oracle import ROUTER_D, synth_router

    B, M, d, tau, seed, stride = 16, 4, ROUTER_D, 1.5, SEED0 + 500, 31
    q, c, mask, pos, teacher = synth_router(seed, B, M, d)
    t = torch.from_numpy
    S_dense = ref_shim.reference_citadel_sim_score(t(q), t(c), t(mask), False).numpy()
    S_pair = ref_shim.reference_citadel_sim_score(t(q), t(c), t(mask), True).numpy()
    for name, in_batch, coef in [("router_inbatch", True, 0.0), ("router_pairwise", False, 0.0),
                                 ("router_teacher", True, 0.5), ("router_teacher_only", False, 1.0)]:
        if not wanted(name):
            continue
        loss, dq, dc, logged = ref_shim.reference_router_loss(t(q), t(c), t(mask), t(pos), t(teacher), in_batch, coef, tau)
        assert float(logged["train_router_loss"].detach()) == loss.item()
        dq, dc = dq.numpy(), dc.numpy()
        meta = dict(case=name, B=B, M=M, d=d, seed=seed, tau=tau, in_batch=in_batch, teacher_coef=coef, col_stride=stride,
                    source="reference citadel_task.py MultiVecRetrieverTask.router_loss / sim_score, fp32 CPU; inputs = "
                           "oracle.router_oracle.synth_router(seed, B, M, d)")
        save(name, meta, loss=np.float32(loss.item()), S_dense=S_dense, S_pair=S_pair,
             dq_cols=dq[:, ::stride].copy(), dc_cols=dc[:, ::stride].copy(),
             dq_rowsum=dq.astype(np.float64).sum(1), dc_rowsum=dc.astype(np.float64).sum(1),
             dq_colsum=dq.astype(np.float64).sum(0).astype(np.float32), dc_colsum=dc.astype(np.float64).sum(0).astype(np.float32))
    if wanted("router_gather_w2"):
        W, seed = 2, SEED0 + 600
        ctx = mp.get_context("spawn")
        qq = ctx.Queue()
        procs = [ctx.Process(target=_gather_worker, args=(r, W, 29614, seed, qq)) for r in range(W)]
        for p in procs:
            p.start()
        res = sorted([qq.get(timeout=300) for _ in range(W)], key=lambda x: x[0])
        for p in procs:
            p.join()
        arrays = {}
        for r, oq, oc, om, op, ot in res:
            for k, v in oq.items():
                arrays[f"r{r}_q_{k}"] = v
            for k, v in oc.items():
                arrays[f"r{r}_c_{k}"] = v
            arrays[f"r{r}_mask"], arrays[f"r{r}_pos"], arrays[f"r{r}_teacher"] = om, op, ot
        save("router_gather_w2", dict(case="router_gather_w2", W=W, seed=seed,
                                      source="reference citadel_task.py distributed_gather on 2 gloo ranks (PL 1.6.4 all_gather "
                                             "shim); inputs = oracle.router_oracle.synth_gather_rank(seed, rank)"), **arrays)


def search_fixture():
    """f2: /root/reference/dpr_scale/run_retrieval_pytorch.py search_index, unmodified (oracle/ref_shim.py), on a 20 000-passage index:
    the batched loop (3 full batches of 16 queries) and its tail (5 queries)."""
    if not wanted("search_ref"):
        return
    nq, n, d, k, batch, seed = 53, 20000, 64, 100, 16, SEED0 + 700
    q, c = synth_search(seed, nq, n, d)
    scores, ids = ref_shim.reference_search_index(torch.from_numpy(q), torch.from_numpy(c).to(torch.float16), batch, k)
    save("search_ref", dict(case="search_ref", nq=nq, n=n, d=d, k=k, batch=batch, seed=seed,
                            source="reference run_retrieval_pytorch.py search_index (fp16 einsum + torch.topk on the CPU through the "
                                   "shim); inputs = oracle.inbatch_oracle.synth_search(seed, nq, n, d)"),
         scores=scores.numpy().astype(np.float32), ids=ids.numpy().astype(np.int64))


def main():
    assert ref_shim.reference_available(), "needs /root/reference"
    router_fixtures()
    search_fixture()
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # ---- cfg1 / cfg2: full tensors ------------------------------------------------------------------
    for name, (W, B, K, d), variants in [
        ("cfg1", (1, 4, 2, 128), [("P", False, 1.0), ("U", True, 1.0)]),
        ("cfg2", (1, 32, 8, 768), [("U", False, 1.0), ("P", False, 1.0), ("U", True, 0.05)]),
    ]:
        for vi, (distn, ragged, T) in enumerate(variants):
            seed = SEED0 + 100 * vi
            Q, C, y, m = global_inputs(seed, W, B, K, d, distn, ragged)
            r = run_global(Q, C, y, m, T)
            meta = dict(case=name, W=W, B=B, K=K, d=d, dist=distn, ragged=ragged, T=T, seed=seed,
                        source="reference dpr_task.py training_step/sim_score/compute_rank_metrics, fp32 CPU")
            save(f"{name}_{distn}{'r' if ragged else ''}_T{T:g}", meta, **r)

    # ---- cfg3 / cfg5: global step, summaries ---------------------------------------------------------
    for name, (W, B, K, d), (distn, ragged, T), own in [
        ("cfg3", (8, 128, 8, 768), ("U", True, 1.0), 3),
        ("cfg3", (8, 128, 8, 768), ("P", False, 1.0), 5),
        ("cfg5", (8, 64, 2, 1024), ("U", False, 1.0), 6),
        ("cfg3", (8, 128, 8, 768), ("U", True, 0.05), 1),   # round 2: small temperature through the long-row plan
        ("cfg5", (8, 64, 2, 1024), ("U", True, 1.0), 2),    # round 2: ragged (dummy contexts) at d = 1024
    ]:
        seed = SEED0 + 7
        if not wanted(f"{name}_{distn}{'r' if ragged else ''}_T{T:g}"):
            continue
        Q, C, y, m = global_inputs(seed, W, B, K, d, distn, ragged)
        r = run_global(Q, C, y, m, T)
        rng = np.random.default_rng(99)
        si = rng.integers(0, Q.shape[0], 4096)
        sj = rng.integers(0, C.shape[0], 4096)
        dc_own = r["dC"][own * B * K:(own + 1) * B * K]
        meta = dict(case=name, W=W, B=B, K=K, d=d, dist=distn, ragged=ragged, T=T, seed=seed, own_rank=own,
                    source="reference dpr_task.py (global tensors, non-DDP branch), fp32 CPU")
        save(f"{name}_{distn}{'r' if ragged else ''}_T{T:g}", meta,
             loss=r["loss"], lse=r["lse"], ranks=r["ranks"], rank_metrics=r["rank_metrics"],
             sample_i=si, sample_j=sj, sample_S=r["S"][si, sj],
             dq_own=r["dQ"][own * B:(own + 1) * B],
             dc_own_rowsum=dc_own.sum(1), dc_own_colsum=dc_own.sum(0), dc_own_head=dc_own[:64].copy(),
             dC_colsum=r["dC"].sum(0))

    # ---- real multi-rank DDP-branch runs -------------------------------------------------------------
    for name, (W, B, K, d), (distn, ragged, T), port in [
        ("w2", (2, 4, 2, 128), ("P", False, 1.0), 29611),
        ("w4", (4, 8, 8, 768), ("U", True, 0.5), 29612),
        ("cfg4", (8, 8, 8, 768), ("U", False, 1.0), 29613),
    ]:
        seed = SEED0 + 31
        if not wanted(f"{name}_ddp"):
            continue
        loss, dq, dc = run_ddp(W, seed, B, K, d, distn, ragged, T, port)
        meta = dict(case=name, W=W, B=B, K=K, d=d, dist=distn, ragged=ragged, T=T, seed=seed,
                    source=f"reference dpr_task.py DDP branch on {W} gloo ranks (PL 1.6.4 all_gather shim)")
        save(f"{name}_ddp", meta, loss_per_rank=loss, dq_per_rank=dq, dc_per_rank=dc)

    # ---- ties: quantised logits so that many columns tie with the gold one ---------------------------
    rng = np.random.default_rng(5)
    S = np.round(rng.standard_normal((64, 512)).astype(np.float32) * 2) / 2
    y = rng.integers(0, 512, 64).astype(np.int64)
    S[:, 7] = -np.inf
    y[y == 7] = 8
    tS, ty = torch.from_numpy(S), torch.from_numpy(y)
    order = torch.sort(tS, dim=1, descending=True, stable=True).indices
    ranks = ((order == ty[:, None]).nonzero()[:, 1] + 1).numpy().astype(np.int64)
    tv, ti = torch.topk(tS, 16, dim=1)
    save("ties", dict(case="ties", source="torch.sort(descending=True, stable=True) -- the frozen tie rule"),
         S=S, y=y, ranks=ranks, order16=order[:, :16].numpy().astype(np.int64), topk_values=tv.numpy())

    # ---- in_batch_negatives=False branch (dpr_task.py:198-207) ---------------------------------------
    qv, cv, yv, mv = synth_embeddings(77, 6, 4, 128, "U", True)
    loss, dq, dc = ref_shim.reference_training_step(
        torch.from_numpy(qv), torch.from_numpy(cv), torch.from_numpy(yv), torch.from_numpy(mv),
        temperature=1.0, in_batch_negatives=False)
    save("nib", dict(case="nib", B=6, K=4, d=128, dist="U", ragged=True, seed=77, T=1.0,
                     source="reference training_step with in_batch_negatives=False"),
         loss=np.float32(loss.item()), dQ=dq.numpy(), dC=dc.numpy())


if __name__ == "__main__":
    main()
