"""Collectives of the in-batch contrastive path, one process per GPU (torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" in the CPU tests).

What the reference does (dpr_scale/task/dpr_task.py:166-195): four all-gathers (q, c, labels, mask) in fp32,
then every rank recomputes the whole [W*B, W*B*K] matrix.  What this path does:
  forward   all-gather(bf16 context rows) + all-gather(uint8 column mask); labels need no collective
            (the offset rank * ctx_per_rank is arithmetic); queries are never gathered
  backward  reduce-scatter(sum) of the [Nc, d] dC partials -> this rank's [B*K, d] rows
  logging   all-reduce(sum) of one float (the loss numerator)
"""
import torch
import torch.distributed as dist


def world(group=None):
    """(world_size, rank) of ``group``; (1, 0) when torch.distributed is not initialised."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def _is_nccl(group):
    return dist.get_backend(group) == "nccl"


def all_gather_rows(send: torch.Tensor, out: torch.Tensor, group=None, async_op=False):
    """out[r*n:(r+1)*n] = send from rank r (equal n on every rank -- guaranteed upstream by
    ContiguousDistributedSampler padding, utils.py:48-60, and DPRTransform padding, dpr_transform.py:143-161)."""
    W, _ = world(group)
    assert out.shape[0] == W * send.shape[0], (out.shape, send.shape, W)
    if send.dtype == torch.bfloat16 and not _is_nccl(group):
        # gloo has no bf16: ship the bit patterns in a 2-byte type it knows (all-gather only moves bytes)
        return dist.all_gather_into_tensor(out.view(torch.float16), send.view(torch.float16), group=group,
                                           async_op=async_op)
    return dist.all_gather_into_tensor(out, send, group=group, async_op=async_op)


def reduce_scatter_rows(inp: torch.Tensor, out: torch.Tensor, group=None, async_op=False):
    """out = sum over ranks of inp[r*n:(r+1)*n] for this rank r."""
    W, r = world(group)
    n = out.shape[0]
    assert inp.shape[0] == W * n
    if _is_nccl(group):
        return dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    # gloo (CPU tests): no reduce-scatter -- all-reduce then keep the own slice
    tmp = inp.clone()
    dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
    out.copy_(tmp[r * n:(r + 1) * n])
    return None


def all_reduce_sum(t: torch.Tensor, group=None, async_op=False):
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
