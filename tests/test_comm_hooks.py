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


# ---- GPU: the HIP legs behind the C ABI, a one-rank RCCL group, and two processes sharing the device over gloo -------------------
_TD = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}


@pytest.mark.gpu
@pytest.mark.parametrize("wire", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("n,W", [(8, 1), (1000003, 8), (4096 * 33 + 5, 4), (37, 2)])
def test_hip_legs_equal_the_torch_formulation_bit_for_bit(wire, n, W):
    """dprhot_grad_pack / _sum_shards / _unpack against the torch ops they replace (comm_hooks._TorchLegs restated here):
    conversions are round-to-nearest-even on both sides and the shard sum runs in rank order in fp32 -> identical bits."""
    from dpr_scale_amd import comm_hooks

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n + W)
    buf = (torch.randn(n, generator=g) * torch.logspace(-6, 3, n)).to(dev)  # spans fp16's subnormal .. overflow range
    legs = comm_hooks._HipLegs()
    shard = ((n + W - 1) // W + 7) // 8 * 8
    npad = shard * W
    send = torch.full((npad,), 7.0, dtype=_TD[wire], device=dev)
    legs.pack(buf, 1.0 / W, send)
    want = torch.zeros(npad, dtype=_TD[wire], device=dev)
    want[:n] = (buf * (1.0 / W)).to(_TD[wire])
    assert torch.equal(send.view(torch.int16 if wire != "fp32" else torch.int32), want.view(torch.int16 if wire != "fp32" else torch.int32))
    recv = (torch.randn(npad, generator=g) * 3).to(dev).to(_TD[wire])
    for out_name in {wire, "fp32"}:
        out = torch.empty(shard, dtype=_TD[out_name], device=dev)
        legs.sum_shards(recv, W, out)
        acc = recv[:shard].float().clone()
        for r in range(1, W):
            acc += recv[r * shard:(r + 1) * shard].float()
        assert torch.equal(out, acc.to(_TD[out_name])), (wire, out_name)
        full = torch.randn(npad, generator=g).to(dev).to(_TD[out_name])
        back = torch.full((n,), -1.0, device=dev)
        legs.unpack(full, back)
        assert torch.equal(back, full[:n].float())


def _tower():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Linear(37, 530), torch.nn.Tanh(), torch.nn.Linear(530, 511), torch.nn.Tanh(), torch.nn.Linear(511, 3))


@pytest.mark.gpu
def test_hook_on_a_one_rank_rccl_group():
    """The nccl branches of the hook (asynchronous all-to-all, all-gather inside the future, HIP legs on the hook's stream) on a
    one-rank RCCL world: W = 1, so the average is the gradient itself, up to the two roundings of the wire."""
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29763"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        from dpr_scale_amd import comm_hooks

        dev = torch.device("cuda:0")
        x = torch.randn(16, 37, generator=torch.Generator().manual_seed(3)).to(dev)
        ref = _tower().to(dev)
        ref(x).square().sum().backward()
        g_ref = torch.cat([p.grad.flatten() for p in ref.parameters()])
        gmax = g_ref.abs().max().item()
        for wire, eps in (("bf16", 2.0 ** -8), ("fp16", 2.0 ** -11), ("fp32", 2.0 ** -23)):
            for cap in (25, 0.25):
                m, st = comm_hooks.wrap_ddp(_tower().to(dev), device_ids=[0], bucket_cap_mb=cap,
                                            state=comm_hooks.GradCommState(wire_dtype=wire, mode="direct"))
                m(x).square().sum().backward()
                torch.cuda.synchronize()
                g = torch.cat([p.grad.flatten() for p in m.parameters()])
                assert (g - g_ref).abs().max().item() <= 2.0 * eps * gmax + 1e-7, (wire, cap)
                assert st.buckets >= 1 and st.wire_bytes == 0  # one rank: nothing travels
        m, st = comm_hooks.wrap_ddp(_tower().to(dev), device_ids=[0], state=comm_hooks.GradCommState(wire_dtype="fp16", mode="ring"))
        m(x).square().sum().backward()
        torch.cuda.synchronize()
        g = torch.cat([p.grad.flatten() for p in m.parameters()])
        assert (g - g_ref).abs().max().item() <= 2.0 ** -10 * gmax + 1e-7
    finally:
        dist.destroy_process_group()


def _shared_device_worker(rank, W, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from torch.nn.parallel import DistributedDataParallel as DDP

    from dpr_scale_amd import comm_hooks

    dev = torch.device("cuda:0")
    torch.manual_seed(100 + rank)
    x = torch.randn(16, 37).to(dev)
    ref = DDP(_tower().to(dev))
    ref(x).square().sum().backward()
    g_ref = torch.cat([p.grad.flatten() for p in ref.parameters()]).cpu()
    out = {}
    for name, kw in {"direct_bf16": dict(wire_dtype="bf16"), "direct_fp16_small": dict(wire_dtype="fp16", bucket_cap_mb=0.25),
                     "direct_bf16_fp32_return": dict(wire_dtype="bf16", return_dtype="fp32")}.items():
        cap = kw.pop("bucket_cap_mb", 25)
        m, st = comm_hooks.wrap_ddp(_tower().to(dev), bucket_cap_mb=cap, state=comm_hooks.GradCommState(mode="direct", **kw))
        m(x).square().sum().backward()
        torch.cuda.synchronize()
        out[name] = (torch.cat([p.grad.flatten() for p in m.parameters()]).cpu().numpy(), st.buckets)
    q.put((rank, g_ref.numpy(), out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_hook_with_device_buckets_on_two_processes_sharing_the_gpu():
    """Two ranks (gloo moves the bytes through host memory, RCCL refuses two ranks on one device), device-resident buckets: the
    HIP legs run on each rank's bucket; averaged gradients must equal uncompressed DDP's within the wire's rounding and be
    identical on both ranks."""
    W = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shared_device_worker, args=(r, W, 29764, q)) for r in range(W)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(W)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    g_ref = torch.from_numpy(res[0][1])
    gmax = g_ref.abs().max().item()
    for name, eps in (("direct_bf16", 2.0 ** -8), ("direct_fp16_small", 2.0 ** -11), ("direct_bf16_fp32_return", 2.0 ** -8)):
        g0, g1 = torch.from_numpy(res[0][2][name][0]), torch.from_numpy(res[1][2][name][0])
        assert torch.equal(g0, g1), name
        assert (g0 - g_ref).abs().max().item() <= 2.0 * eps * gmax + 1e-7, name
    assert res[0][2]["direct_fp16_small"][1] >= res[0][2]["direct_bf16"][1] >= 1
