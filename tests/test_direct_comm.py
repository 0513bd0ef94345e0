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
    finally:
        dist.destroy_process_group()
