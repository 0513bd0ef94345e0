"""the DDP gradient hook (dpr_scale_amd/comm_hooks.py) on world-size-2 gloo ranks (CPU).
The hooked gradients must equal uncompressed DDP's (fp32 all-reduce mean) within the rounding of the wire dtype, for the
all-pairs-exchange decomposition with fp32 accumulation and for the reference's ring decomposition."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, W, port, cfg, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from torch.nn.parallel import DistributedDataParallel as DDP

    from dpr_scale_amd import comm_hooks

    def model():
        torch.manual_seed(7)  # ~1.2 MB of parameters: more than DDP's 1 MiB first bucket
        return torch.nn.Sequential(torch.nn.Linear(37, 530), torch.nn.Tanh(), torch.nn.Linear(530, 511), torch.nn.Tanh(), torch.nn.Linear(511, 3))

    torch.manual_seed(100 + rank)
    x = torch.randn(16, 37)
    ref = DDP(model())
    ref(x).square().sum().backward()
    g_ref = torch.cat([p.grad.flatten() for p in ref.parameters()])
    out = {}
    for name, kw in cfg.items():
        m, st = comm_hooks.wrap_ddp(model(), bucket_cap_mb=kw.pop("bucket_cap_mb", 25), state=comm_hooks.GradCommState(**kw))
        m(x).square().sum().backward()
        out[name] = (torch.cat([p.grad.flatten() for p in m.parameters()]).numpy(), st.buckets, st.wire_bytes)
    q.put((rank, g_ref.numpy(), out))  # numpy: a tensor in the queue needs its producer alive
    dist.barrier()
    dist.destroy_process_group()


def test_hooked_gradients_equal_uncompressed_ddp_within_wire_rounding():
    W = 2
    cfg = {
        "direct_bf16": dict(wire_dtype="bf16", mode="direct"),
        "direct_bf16_small_buckets": dict(wire_dtype="bf16", mode="direct", bucket_cap_mb=0.25),
        "direct_bf16_fp32_return": dict(wire_dtype="bf16", mode="direct", return_dtype="fp32"),
        "direct_fp16": dict(wire_dtype="fp16", mode="direct"),
        "direct_fp32": dict(wire_dtype="fp32", mode="direct"),
        "ring_bf16": dict(wire_dtype="bf16", mode="ring"),
        "ring_fp16": dict(wire_dtype="fp16", mode="ring"),
    }
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, W, 29761, cfg, q)) for r in range(W)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(W)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    g_ref = torch.from_numpy(res[0][1])
    assert torch.equal(g_ref, torch.from_numpy(res[1][1]))  # DDP: identical averaged gradients on both ranks
    gmax = g_ref.abs().max().item()
    eps = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11, "fp32": 2.0 ** -23}
    for name in cfg:
        g0, nb0, wb0 = res[0][2][name]
        g1, nb1, wb1 = res[1][2][name]
        g0, g1 = torch.from_numpy(g0), torch.from_numpy(g1)
        assert torch.equal(g0, g1), name  # every rank ends with the same bits
        wire = "fp32" if "fp32" in name and "return" not in name else ("fp16" if "fp16" in name else "bf16")
        # per rank contribution rounded once (relative eps/2 each), the sum rounded once more on the return leg
        bound = 2.0 * eps[wire] * gmax + 1e-7
        assert (g0 - g_ref).abs().max().item() <= bound, (name, (g0 - g_ref).abs().max().item(), bound)
        assert nb0 == nb1 and nb0 >= 1 and wb0 > 0
    assert res[0][2]["direct_bf16_small_buckets"][1] >= res[0][2]["direct_bf16"][1]
    # fp32 wire reproduces uncompressed DDP to fp32 rounding
    assert (torch.from_numpy(res[0][2]["direct_fp32"][0]) - g_ref).abs().max().item() <= 4e-7 * max(1.0, gmax)
    # fp32 accumulation + fp32 return is at least as close as the half-precision ring
    e_dir = (torch.from_numpy(res[0][2]["direct_bf16_fp32_return"][0]) - g_ref).abs().max().item()
    e_ring = (torch.from_numpy(res[0][2]["ring_bf16"][0]) - g_ref).abs().max().item()
    assert e_dir <= e_ring + 1e-9


def test_time_model_orders_the_decompositions():
    from dpr_scale_amd.comm_hooks import model_allreduce_seconds

    # 2 x bert-base (876 MB of fp32 gradients) and 2 x bert-large (2.7 GB) on 8 GPUs, 7 links x 153 GB/s
    for nbytes in (876e6, 2.7e9):
        direct = model_allreduce_seconds(nbytes, mode="direct", buckets=max(1, int(nbytes / 2 / 64e6)))
        ring = model_allreduce_seconds(nbytes, mode="ring", buckets=max(1, int(nbytes / 2 / 64e6)))
        fp32_ring = model_allreduce_seconds(nbytes, mode="ring", wire_bytes=4)
        assert direct < ring < fp32_ring
    assert 0.5e-3 < model_allreduce_seconds(876e6, mode="direct") < 2e-3
