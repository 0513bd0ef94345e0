"""The optional communicator of the C ABI (dprhot_comm_*, dprhot_allgather_ctx, dprhot_reducescatter_dc,
dprhot_allreduce_sum) on the one GPU of the box: a one-rank RCCL communicator next to a one-rank torch.distributed nccl
group -- API plumbing (id hand-over, init, stream, counts, dtypes) and the guarded factory that compares the two."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_direct_comm_one_rank_group_agrees_with_torch_distributed():
    import torch.distributed as dist

    from dpr_scale_amd import dist as D

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29741"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        comm = D.try_direct_comm(dev)
        assert comm is not None, "the communicator could not be built or failed its self-check"
        send = torch.randn(264, 768, device=dev).to(torch.bfloat16)
        out = torch.empty_like(send)
        comm.all_gather_rows(send, out)
        part = torch.randn(264, 768, device=dev)
        mine = torch.empty_like(part)
        comm.reduce_scatter_rows(part, mine)
        s = torch.tensor([3.5], device=dev)
        comm.all_reduce_sum(s)
        torch.cuda.synchronize()
        assert torch.equal(out, send) and torch.equal(mine, part) and s.item() == 3.5
        comm.close()
        # the product's transport: once enabled for the group, dist.all_gather_rows / reduce_scatter_rows / all_reduce_sum go through
        # the communicator -- synchronously on the caller's stream, or on the side stream behind an event (async_op=True), for
        # both wire formats of the backward
        assert D.direct_comm() is None
        # opt-in: without DPRHOT_DIRECT_RCCL=1 the task's call is a no-op and the path stays on torch.distributed
        os.environ.pop("DPRHOT_DIRECT_RCCL", None)
        assert D.enable_direct_comm(dev) is None and D.direct_comm() is None
        D.disable_direct_comm()
        os.environ["DPRHOT_DIRECT_RCCL"] = "1"
        assert D.enable_direct_comm(dev) is not None and D.direct_comm() is not None  # (set-up on the watchdog's helper thread)
        for dt in (torch.float32, torch.bfloat16):
            part = torch.randn(264, 768, device=dev).to(dt)
            mine = torch.full_like(part, 7.0)
            w = D.reduce_scatter_rows(part, mine, None, async_op=True)
            assert isinstance(w, D._StreamWork)
            w.wait()
            out = torch.empty_like(send)
            w2 = D.all_gather_rows(send, out, None, async_op=True)
            w2.wait()
            mine2 = torch.empty_like(part)
            assert D.reduce_scatter_rows(part, mine2, None) is None
            torch.cuda.synchronize()
            assert torch.equal(mine, part) and torch.equal(mine2, part) and torch.equal(out, send)
        s2 = torch.tensor([1.25], device=dev)
        assert D.all_reduce_sum(s2) is None
        torch.cuda.synchronize()
        assert s2.item() == 1.25
        D.disable_direct_comm()
        assert D.direct_comm() is None
    finally:
        dist.destroy_process_group()
