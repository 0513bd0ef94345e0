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


def _worker(rank, W, port, meta, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from _oracle_kernels import OracleKernels
    from oracle.inbatch_oracle import synth_embeddings
    from dpr_scale_amd.hotpath import inbatch_contrastive_loss

    qv, cv, y, m = synth_embeddings(meta["seed"] + rank, meta["B"], meta["K"], meta["d"], meta["dist"], meta["ragged"])
    tq = torch.from_numpy(qv).requires_grad_(True)
    tc = torch.from_numpy(cv).requires_grad_(True)
    loss = inbatch_contrastive_loss(tq, tc, torch.from_numpy(y), torch.from_numpy(m), meta["T"], None, OracleKernels())
    (loss * 3.0).backward()  # grad_output != 1 exercises the device-scalar path
    q.put((rank, loss.item(), tq.grad.numpy() / 3.0, tc.grad.numpy() / 3.0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,port", [("w2_ddp", 29711), ("w4_ddp", 29712)])
def test_ddp_branch_matches_reference(name, port):
    meta, g = load_golden(name)
    W = meta["W"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, W, port, meta, q)) for r in range(W)]
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
