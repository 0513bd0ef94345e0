"""DDP gradient all-reduce of the encoder towers, shaped for the 8-GPU xGMI node.

Reference: dpr_scale/task/dpr_task.py:90-92 registers torch's ``fp16_compress_hook`` on the DDP model when
``fp16_grads`` is set (strategies ``ddp_sharded`` / ``ddp``: conf/msmarco_baseline.yaml:35, conf/nq.yaml:31).  That hook
casts every bucket to fp16 and runs ONE ring all-reduce in fp16: W - 1 sequential half-precision additions per element,
and on MI355X a ring whose every hop is bound by ONE xGMI link (~153 GB/s) although each GPU has seven.

``compressed_allreduce_hook`` keeps the reference's contract -- bucket in, future of the averaged bucket out, half the
bytes on the wire -- with a decomposition that fits a fully connected node:

  1. this rank's bucket, pre-divided by W in fp32, is rounded ONCE to the wire dtype (bf16 by default: fp32's exponent
     range, no loss-scale interplay; fp16 selectable to mirror the reference bit for bit in format);
  2. reduce-scatter as a DIRECT ALL-PAIRS EXCHANGE (``all_to_all_single``): rank r receives everybody's shard r -- on the
     fully connected node every pair of GPUs owns a link, so the W - 1 transfers of a rank use W - 1 links at once;
  3. the W received shards are summed in FP32 (one rounding of the sum instead of W - 1 roundings inside a half-precision
     ring) -- "fp32 accumulation";
  4. all-gather of the reduced shards in the wire dtype (or fp32 with ``return_dtype=torch.float32``: the second
     rounding disappears at 1.5x the total bytes), decompressed into the bucket.

Wire bytes per rank: 2 x (W-1)/W x N x 2 B, the same as the half-precision ring, in 2 collectives per bucket.  Buckets are
DDP's own (``bucket_cap_mb`` of the wrapper; ``wrap_ddp`` below exposes it), reduced in the order they become ready, the
futures run on RCCL's stream underneath the remaining backward.

``mode="ring"`` is the reference's decomposition (one all-reduce in the wire dtype) for A/B runs.
The three local legs (pack, shard sum, unpack) are HIP kernels behind the C ABI (csrc/gradcomm.h): one pass over HBM each
instead of torch's five or six elementwise passes.  Nothing here is measured on more than one GPU (one-GPU boxes):
correctness is covered by the world-size-2 gloo tests in tests/test_comm_hooks.py (CPU buckets, and device buckets of two
processes sharing the GPU) and a one-rank RCCL group on the GPU; bench.py's `grad_hook` block times the legs; DESIGN.md
section 6 models the links.
"""
import os

import torch
import torch.distributed as dist

_WIRE = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}


class GradCommState:
    """State object handed to ``register_comm_hook``.  wire_dtype: bf16 | fp16 | fp32 (torch dtype or name);
    mode: "direct" (all-pairs exchange + fp32 accumulation) | "ring" (one all-reduce in the wire dtype);
    return_dtype: dtype of the all-gather leg (default = wire dtype)."""

    def __init__(self, group=None, wire_dtype="bf16", mode="direct", return_dtype=None):
        self.group = group
        self.wire_dtype = _WIRE[wire_dtype] if isinstance(wire_dtype, str) else wire_dtype
        self.return_dtype = self.wire_dtype if return_dtype is None else (_WIRE[return_dtype] if isinstance(return_dtype, str) else return_dtype)
        assert mode in ("direct", "ring")
        self.mode = mode
        self.buckets = 0      # buckets reduced so far
        self.wire_bytes = 0   # bytes this rank put on the wire so far (both legs)

    @classmethod
    def from_env(cls, group=None, default_wire="bf16"):
        """DPRHOT_GRAD_WIRE=bf16|fp16|fp32, DPRHOT_GRAD_MODE=direct|ring, DPRHOT_GRAD_RETURN=bf16|fp16|fp32."""
        return cls(group, os.environ.get("DPRHOT_GRAD_WIRE", default_wire), os.environ.get("DPRHOT_GRAD_MODE", "direct"),
                   os.environ.get("DPRHOT_GRAD_RETURN"))


_KIND = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}  # wire kinds of include/dprhot.h


def _as_gloo_safe(t):
    """gloo has no bf16: ship the bit patterns in a 2-byte type it knows (exchange / gather only move bytes)."""
    return t.view(torch.float16) if t.dtype == torch.bfloat16 else t


class _HipLegs:
    """The three local legs of the hook on the HIP kernels behind the C ABI (dprhot_grad_pack / _sum_shards / _unpack):
    one pass over HBM each, on the current stream.  Device buckets always take this path; there is no torch fallback."""

    def __init__(self):
        import ctypes

        from . import _lib

        self._lib, self._ct = _lib, ctypes

    def _stream(self):
        return self._ct.c_void_p(torch.cuda.current_stream().cuda_stream)

    def pack(self, buf, scale, send):
        self._lib.check(self._lib.lib.dprhot_grad_pack(buf.data_ptr(), buf.numel(), float(scale), _KIND[send.dtype], send.data_ptr(),
                                                       send.numel(), self._stream()), "dprhot_grad_pack")

    def sum_shards(self, recv, W, out):
        self._lib.check(self._lib.lib.dprhot_grad_sum_shards(recv.data_ptr(), int(W), out.numel(), _KIND[recv.dtype], _KIND[out.dtype],
                                                             out.data_ptr(), self._stream()), "dprhot_grad_sum_shards")

    def unpack(self, full, buf):
        self._lib.check(self._lib.lib.dprhot_grad_unpack(full.data_ptr(), _KIND[full.dtype], buf.data_ptr(), buf.numel(), self._stream()),
                        "dprhot_grad_unpack")


class _TorchLegs:
    """The same three legs in torch ops -- for CPU buckets only (the gloo tests); a device bucket never comes here."""

    def pack(self, buf, scale, send):
        n = buf.numel()
        send[:n].copy_(buf * scale)  # ONE rounding of this rank's contribution (pre-divided in fp32)
        send[n:].zero_()

    def sum_shards(self, recv, W, out):
        out.copy_(recv.view(W, -1).float().sum(dim=0))  # fp32 accumulation of the W shards

    def unpack(self, full, buf):
        buf.copy_(full[:buf.numel()])


_HIP_LEGS = None


def _legs(buf):
    global _HIP_LEGS
    if not buf.is_cuda:
        return _TorchLegs()
    if _HIP_LEGS is None:
        _HIP_LEGS = _HipLegs()  # raises if libdprhot.so is missing: loudly, by design
    return _HIP_LEGS


def _done(value):
    fut = torch.futures.Future()
    fut.set_result(value)
    return fut


def compressed_allreduce_hook(state: GradCommState, bucket: dist.GradBucket) -> torch.futures.Future[torch.Tensor]:
    """DDP communication hook: returns a future whose value is the bucket averaged over the ranks (fp32)."""
    group = state.group if state.group is not None else dist.group.WORLD
    W = dist.get_world_size(group)
    buf = bucket.buffer()  # flat fp32 view of the bucket's gradients
    n = buf.numel()
    nccl = dist.get_backend(group) == "nccl"
    state.buckets += 1

    if state.mode == "ring":
        wire = (buf / W).to(state.wire_dtype)
        state.wire_bytes += 2 * (W - 1) * wire.numel() * wire.element_size() // max(W, 1)
        if wire.dtype == torch.bfloat16 and not nccl:
            # gloo cannot add bf16: emulate the ring's result (sum in bf16) through an fp32 all-reduce of the rounded values
            tmp = wire.float()
            dist.all_reduce(tmp, group=group)
            return _done(buf.copy_(tmp.to(torch.bfloat16)))
        if not nccl:  # callbacks of a gloo future run on gloo's worker threads: keep every collective on this thread
            dist.all_reduce(wire, group=group)
            return _done(buf.copy_(wire))
        fut = dist.all_reduce(wire, group=group, async_op=True).get_future()
        return fut.then(lambda f: buf.copy_(f.value()[0]))

    legs = _legs(buf)
    shard = ((n + W - 1) // W + 7) // 8 * 8  # 16-byte groups per shard
    npad = shard * W
    send = torch.empty(npad, dtype=state.wire_dtype, device=buf.device)
    legs.pack(buf, 1.0 / W, send)
    recv = torch.empty_like(send)
    state.wire_bytes += (W - 1) * shard * (send.element_size() + torch.empty((), dtype=state.return_dtype).element_size())

    def reduce_and_gather(wait2):
        mine = torch.empty(shard, dtype=state.return_dtype, device=buf.device)
        legs.sum_shards(recv, W, mine)
        full = torch.empty(npad, dtype=state.return_dtype, device=buf.device)
        w2 = dist.all_gather_into_tensor(full if nccl else _as_gloo_safe(full), mine if nccl else _as_gloo_safe(mine), group=group,
                                         async_op=wait2)
        if wait2:
            w2.wait()  # nccl: orders the current (hook) stream behind the collective, the host does not block
        legs.unpack(full, buf)
        return buf

    if not nccl:
        # gloo (CPU tests, or two processes sharing one device): a Future.then callback would run on one of gloo's worker threads
        # and issue the second collective from there -- with several buckets in flight its order against the main thread's next
        # all-to-all differs between ranks.  Both legs run here, in bucket order, and the future is already complete.
        if buf.is_cuda:
            r_cpu, s_cpu = torch.empty(npad, dtype=torch.float16 if send.element_size() == 2 else send.dtype), _as_gloo_safe(send).cpu()
            dist.all_to_all_single(r_cpu, s_cpu, group=group)  # gloo moves host memory only
            _as_gloo_safe(recv).copy_(r_cpu)
            mine = torch.empty(shard, dtype=state.return_dtype, device=buf.device)
            legs.sum_shards(recv, W, mine)
            f_cpu = torch.empty(npad, dtype=_as_gloo_safe(mine).dtype)
            dist.all_gather_into_tensor(f_cpu, _as_gloo_safe(mine).cpu(), group=group)
            full = torch.empty(npad, dtype=state.return_dtype, device=buf.device)
            _as_gloo_safe(full).copy_(f_cpu)
            legs.unpack(full, buf)
            return _done(buf)
        dist.all_to_all_single(_as_gloo_safe(recv), _as_gloo_safe(send), group=group)
        return _done(reduce_and_gather(False))
    fut = dist.all_to_all_single(recv, send, group=group, async_op=True).get_future()
    return fut.then(lambda _: reduce_and_gather(True))


def register(ddp_model, state: GradCommState = None):
    """Register the hook on a torch DistributedDataParallel module; returns the state (for its counters)."""
    state = state if state is not None else GradCommState.from_env()
    ddp_model.register_comm_hook(state, compressed_allreduce_hook)
    return state


def wrap_ddp(module, device_ids=None, bucket_cap_mb=64, state: GradCommState = None, **ddp_kwargs):
    """DistributedDataParallel(module) with the hook registered.  bucket_cap_mb: DDP's bucket size -- the unit of overlap
    with the backward and the message size on the links (the all-pairs exchange sends bucket / W per peer: 64 MiB buckets
    on 8 GPUs = 4 MiB bf16 messages, well past the xGMI latency regime; torch's default is 25)."""
    from torch.nn.parallel import DistributedDataParallel as DDP

    ddp = DDP(module, device_ids=device_ids, bucket_cap_mb=bucket_cap_mb, **ddp_kwargs)
    return ddp, register(ddp, state)


def model_allreduce_seconds(param_bytes_fp32, world=8, links=7, link_GBps=153.0, wire_bytes=2, mode="direct", latency_us=12.0, buckets=1):
    """First-order time model: `param_bytes_fp32` of fp32 gradients, `wire_bytes` per element on
    the wire.  direct: each leg moves (W-1)/W of the payload per rank spread over min(links, W-1) links; ring: 2 (W-1)/W
    of the payload through ONE link per hop."""
    n = param_bytes_fp32 / 4.0
    payload = n * wire_bytes
    if mode == "direct":
        per_leg = payload * (world - 1) / world / (min(links, world - 1) * link_GBps * 1e9)
        return 2 * per_leg + 2 * buckets * latency_us * 1e-6
    return 2 * payload * (world - 1) / world / (link_GBps * 1e9) + buckets * 2 * (world - 1) * latency_us * 1e-6 / 4
