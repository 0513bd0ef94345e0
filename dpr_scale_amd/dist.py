"""Collectives of the in-batch contrastive path, one process per GPU (torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" in the CPU tests).

What the reference does (dpr_scale/task/dpr_task.py:166-195): four all-gathers (q, c, labels, mask) in fp32,
then every rank recomputes the whole [W*B, W*B*K] matrix.  What this path does:
  forward   all-gather(bf16 context rows) + all-gather(uint8 column mask); labels need no collective
            (the offset rank * ctx_per_rank is arithmetic); queries are never gathered
  backward  reduce-scatter(sum) of the [Nc, d] dC partials -> this rank's [B*K, d] rows
  logging   all-reduce(sum) of one float (the loss numerator)
"""
import os

import torch
import torch.distributed as dist


# ---- which form of the path's two collectives, and through which transport ---------------------------------------------------------
# topology  "rccl": RCCL's own all-gather / reduce-scatter (rings);  "allpairs": direct all-pairs exchanges (SURVEY.md section 8(e)
#           "Topology": on the fully connected 8-GPU node every pair of GPUs owns an xGMI link, so a rank's W - 1 transfers use W - 1
#           links at once -- all-gather = every rank sends its block to everybody (grouped send/recv), reduce-scatter = every rank
#           sends chunk k to rank k and adds the W chunks it receives in FP32 in rank order).  Modelled on 7 links x ~153 GB/s per
#           direction (cfg3, W = 8): all-gather of 1.5 MiB per rank ~10 us against ~70 us for a ring; reduce-scatter of 25 MB of fp32
#           partials ~20 us + a 6 us local sum against ~140 us.
# Precedence: configure() (bench.py's variants, tests) > DPRHOT_PATH_COLLECTIVES=allpairs|rccl > what choose_path_collectives()
# MEASURED for the group (DenseRetrieverTask runs it once before the first step) > "rccl".
_CFG = {"topology": None, "direct": None}
_PROBED = {}   # id(group) -> {"topology": ..., "us": {...}} (choose_path_collectives)


def configure(topology=None, direct=None):
    """Process-wide override of the product's choices (bench.py's variant matrix, tests): topology "rccl" | "allpairs" | None,
    direct True (use the registered C ABI communicator) | False (torch.distributed even if one is registered) | None."""
    assert topology in (None, "rccl", "allpairs") and direct in (None, True, False)
    _CFG["topology"], _CFG["direct"] = topology, direct


def path_topology(group=None):
    if _CFG["topology"] is not None:
        return _CFG["topology"]
    env = os.environ.get("DPRHOT_PATH_COLLECTIVES", "")
    if env in ("allpairs", "rccl"):
        return env
    p = _PROBED.get(_gkey(group))
    return p["topology"] if p else "rccl"


def _allpairs(group=None):
    return path_topology(group) == "allpairs"


def decide_topology(us_rccl, us_allpairs, group=None, device=None):
    """COLLECTIVE: every rank brings its own two timings; the group agrees on ONE answer -- the slowest rank's time decides for each
    form (all-reduce MAX), and "allpairs" is taken only if EVERY rank's comparison says so (all-reduce MIN of the flag; the MAX
    makes the inputs identical already, the MIN is the belt to those braces: ranks must never split over a collective's form).
    A form that failed on any rank arrives as inf and loses."""
    t = torch.tensor([float(us_rccl), float(us_allpairs)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    rccl_us, ap_us = t.tolist()
    flag = torch.tensor([1 if ap_us < 0.95 * rccl_us else 0], dtype=torch.int32, device=t.device)  # (all-pairs must win by 5 %)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return ("allpairs" if int(flag.item()) == 1 else "rccl"), {"rccl": rccl_us, "allpairs": ap_us}


def path_is_pinned():
    """True when the form of the path's collectives was fixed by hand (configure() or DPRHOT_PATH_COLLECTIVES): nothing to measure."""
    return _CFG["topology"] is not None or os.environ.get("DPRHOT_PATH_COLLECTIVES", "") in ("allpairs", "rccl")


def _agree(flag, group, device):
    """COLLECTIVE: True iff `flag` is true on every rank (all-reduce MIN)."""
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()) == 1)


class probe_watchdog:
    """`with probe_watchdog(seconds, what):` -- a collective start-up measurement that has not come back after `seconds` cannot be
    recovered from inside the process (a rank waiting in a collective its peers never issue waits for ever): the watchdog says so and
    ends the process with exit code 70 instead of leaving a silent hang.  bench.py arms its own (it has a line to print first);
    DenseRetrieverTask arms this one around dist.choose_path_collectives.  seconds <= 0 disarms."""

    def __init__(self, seconds, what="dpr_scale_amd.dist: the collective start-up probe"):
        self.seconds, self.what, self.timer = float(seconds), what, None

    def _fire(self):
        import sys

        sys.stderr.write(f"{self.what} did not come back within {self.seconds:.0f} s: a rank is waiting in a collective its peers never "
                         "issued.  Set DPRHOT_PATH_PROBE=0 (keep RCCL's collectives) or DPRHOT_PATH_COLLECTIVES=rccl|allpairs to skip the "
                         "measurement; exiting (70)\n")
        sys.stderr.flush()
        os._exit(70)

    def __enter__(self):
        if self.seconds > 0:
            import threading

            self.timer = threading.Timer(self.seconds, self._fire)
            self.timer.daemon = True
            self.timer.start()
        return self

    def __exit__(self, *exc):
        if self.timer is not None:
            self.timer.cancel()
        return False


def choose_path_collectives(device, rows_c, d, group=None, iters=20, wire=torch.float32, wires=None, preflight=None):
    """COLLECTIVE, once per group (DenseRetrieverTask.on_pretrain_routine_start): time the path's two collectives at the step's real
    message sizes -- all-gather of one packed block [rows_c, d] bf16 per rank, reduce-scatter of [W * rows_c, d] partials in the wire
    format -- in both forms, `iters` iterations each behind two warm-up rounds, and keep the faster form for this group
    (decide_topology).  `wires`: the dC wire formats to consider (round 6: the probe chooses the WIRE too -- (torch.float32,
    torch.bfloat16): same timing per wire; bf16 is taken only when its best form beats fp32's best form by 5 %, agreed on like the form);
    default: `wire` alone.  The chosen wire is in _PROBED[...]["wire"] / path_wire().
    A form pinned by hand (configure() / DPRHOT_PATH_COLLECTIVES) is not measured at all -- the other form is never issued.
    World size 1: nothing to choose.  `preflight(form, wire) -> bool`: the caller's own LOCAL check of a candidate (no collectives in it).

    No rank may leave a form while its peers are still inside it (ADVICE r5 / VERDICT r5 #9): per candidate (form, wire)
      1. pre-flight, LOCAL: the buffers are allocated and the transport is asked whether it has the form -- the one-sided failures
         (out of memory on the scratch copy, a communicator without send/recv) happen here, before any collective;
      2. the ranks AGREE on the pre-flight (all-reduce MIN); a candidate any rank cannot run is abandoned by every rank, unissued;
      3. ONE trial iteration + synchronise, and the ranks agree again;
      4. the timed loop -- no try/except: a failure now is raised (loudly, under the caller's watchdog), never swallowed."""
    W, _ = world(group)
    k = _gkey(group)
    if W <= 1 or k in _PROBED:
        return _PROBED.get(k, {"topology": "rccl"})["topology"]
    import time

    wires = tuple(wires) if wires else (wire,)
    pinned = path_topology(group) if path_is_pinned() else None
    forms = (pinned,) if pinned else ("rccl", "allpairs")
    sync = torch.cuda.synchronize if torch.device(device).type == "cuda" else (lambda: None)
    saved = _CFG["topology"]
    us = {}
    try:
        for w in wires:
            for form in forms:
                _CFG["topology"] = form
                bufs, ok = None, True
                try:  # 1. local pre-flight: nothing collective in here
                    bufs = (torch.zeros((rows_c, d), dtype=torch.bfloat16, device=device), torch.empty((W * rows_c, d), dtype=torch.bfloat16, device=device),
                            torch.zeros((W * rows_c, d), dtype=w, device=device), torch.empty((rows_c, d), dtype=w, device=device))
                    if form == "allpairs":
                        torch.empty((W * rows_c, d), dtype=w, device=device)  # the exchange's scratch copy must fit too
                        c = direct_comm(group)
                        ok = c is None or getattr(c, "has_allpairs", True)
                    if preflight is not None:  # the caller's own local checks (tests inject a one-sided failure here)
                        ok = ok and bool(preflight(form, w))
                except Exception:
                    ok = False
                if not _agree(ok, group, device):  # 2.
                    us[(form, w)] = float("inf")
                    continue
                send, gathered, part, mine = bufs
                err = None
                try:  # 3. one guarded trial iteration
                    all_gather_rows(send, gathered, group)
                    reduce_scatter_rows(part, mine, group)
                    sync()
                except Exception as e:
                    err = e
                if not _agree(err is None, group, device):
                    us[(form, w)] = float("inf")
                    continue
                for _ in range(2):  # 4. warm-up, then the timed iterations
                    all_gather_rows(send, gathered, group)
                    reduce_scatter_rows(part, mine, group)
                sync()
                t0 = time.perf_counter()
                for _ in range(iters):
                    all_gather_rows(send, gathered, group)
                    reduce_scatter_rows(part, mine, group)
                sync()
                us[(form, w)] = (time.perf_counter() - t0) / iters * 1e6
                del bufs, send, gathered, part, mine
    finally:
        _CFG["topology"] = saved
    per_wire = {}
    for w in wires:
        if pinned:
            t = torch.tensor([us[(pinned, w)]], dtype=torch.float64, device=device if device is not None else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            per_wire[w] = (pinned, {pinned: float(t.item())})
        else:
            per_wire[w] = decide_topology(us[("rccl", w)], us[("allpairs", w)], group, device)
    best = {w: min(per_wire[w][1].values()) for w in wires}  # (identical on every rank: decide_topology's all-reduce MAX)
    chosen = wires[0]
    if len(wires) > 1:
        cand = min(wires[1:], key=lambda w: best[w])
        take = best[cand] < 0.95 * best[wires[0]]  # the narrower wire costs a rounding per partial: it has to win by 5 %
        if _agree(take, group, device):
            chosen = cand
    topo, agreed = per_wire[chosen]
    _PROBED[k] = {"topology": topo, "us": agreed, "wire": chosen, "pinned": bool(pinned),
                  "us_by_wire": {str(w).replace("torch.", ""): per_wire[w][1] for w in wires}}
    return topo


def path_wire(group=None, default=None):
    """The dC wire format choose_path_collectives settled on for `group` (None / `default` when it has not run or had one wire only)."""
    p = _PROBED.get(_gkey(group))
    return p.get("wire", default) if p else default


class _Then:
    """A collective's work handle plus what has to run on the waiting stream once it is done."""

    def __init__(self, work, then):
        self.work, self.then = work, then

    def wait(self):
        if self.work is not None:
            self.work.wait()
        self.then()


def world(group=None):
    """(world_size, rank) of ``group``; (1, 0) when torch.distributed is not initialised."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def force_dist():
    """DPRHOT_FORCE_DIST=1 with an initialised process group: the multi-rank code path (packed layout, ONE all-gather started under the
    query tower, reduce-scatter under the query-tower backward) also at world size 1 -- the way to put the collectives and their
    overlap on a timeline on a one-GPU box (scripts/overlap_trace.py).  Never set in production."""
    return os.environ.get("DPRHOT_FORCE_DIST") == "1" and dist.is_available() and dist.is_initialized()


def _is_nccl(group):
    return dist.get_backend(group) == "nccl"


# ---- the C ABI communicator as the path's transport (SURVEY.md section 8 b3 / e) ------------------------------------------------------
_DIRECT = {}   # id(group) -> DirectComm, or False once its collective set-up has failed for that group
_SIDE = {}     # device index -> the side HIP stream asynchronous collectives are issued on


def _gkey(group):
    return id(group) if group is not None else 0


def direct_comm(group=None):
    """The DirectComm registered for `group` by enable_direct_comm(), or None (lookup only: never collective)."""
    if _CFG["direct"] is False:
        return None
    c = _DIRECT.get(_gkey(group))
    return c if c else None


def enable_direct_comm(device, group=None):
    """COLLECTIVE over `group`, once per group (DenseRetrieverTask calls it on every rank before the first training step): from then
    on the path's all-gather, reduce-scatter and loss all-reduce are issued through the C ABI communicator -- synchronous ones on the
    caller's stream (no hand-over to RCCL's stream and back), asynchronous ones on one side HIP stream with an event either way.
    Returns the communicator, or None when any rank could not build it (every rank then keeps torch.distributed).

    OPT-IN: DPRHOT_DIRECT_RCCL=1.  A second RCCL communicator whose kernels run next to torch.distributed's (DDP's bucket all-reduces
    on its own stream) has only ever run on one physical GPU here; two communicators whose kernels reach the device in a different
    order on different ranks is a known deadlock pattern, so it stays off until it has run on a multi-GPU box.

    The set-up's own handshakes (stage agreements, id broadcast, the self-check against torch.distributed) run on a DEDICATED process
    group created here with the watchdog's time limit as its collective timeout (DPRHOT_DIRECT_TIMEOUT_S, default 60 s) -- never on
    `group`: a set-up that does not come back in time is abandoned with its helper thread, and whatever that thread still does, it
    does on a group the training step never uses (round 4's watchdog only checked a flag between stages; a late helper could queue
    an all-reduce on the training group that paired with DDP's).  The side group is created by every rank (new_group is collective
    over the default group) and destroyed when the set-up is over, either way."""
    k = _gkey(group)
    if k not in _DIRECT:
        on = os.environ.get("DPRHOT_DIRECT_RCCL", "0") == "1"
        comm = None
        if on and dist.is_available() and dist.is_initialized() and _is_nccl(group):
            import datetime

            limit = float(os.environ.get("DPRHOT_DIRECT_TIMEOUT_S", "60"))
            ranks = dist.get_process_group_ranks(group) if group is not None else None
            side = dist.new_group(ranks=ranks, backend="nccl", timeout=datetime.timedelta(seconds=limit))
            comm = _with_watchdog(lambda alive: try_direct_comm(device, group, alive, handshake=side), limit)
            _retire_group(side, late=comm is None)
        _DIRECT[k] = comm or False
    return direct_comm(group)


def _retire_group(pg, late):
    """Destroy a set-up group.  After a timeout (`late`) a collective may still be stuck on it: the destroy then runs on a daemon thread
    so that it cannot hold the caller either."""
    import threading

    def kill():
        try:
            dist.destroy_process_group(pg)
        except Exception:
            pass

    if late:
        threading.Thread(target=kill, name="dprhot-direct-comm-retire", daemon=True).start()
    else:
        kill()


def _with_watchdog(fn, timeout_s):
    """fn(alive) on a helper thread; None if it has not returned after timeout_s.  `alive()` turns False at the timeout: a set-up that
    comes back late releases what it built and stops.  fn must keep its collectives to a group of its own (enable_direct_comm's side
    group): an abandoned helper is not interrupted, it is merely somewhere it can do no harm."""
    import threading

    state = {"alive": True, "out": None}
    dev = torch.cuda.current_device() if torch.cuda.is_available() else None

    def run():
        try:
            if dev is not None:
                torch.cuda.set_device(dev)
            out = fn(lambda: state["alive"])
            if state["alive"]:
                state["out"] = out
            elif out is not None:
                out.close()
        except Exception:
            state["out"] = None

    t = threading.Thread(target=run, name="dprhot-direct-comm-setup", daemon=True)
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        state["alive"] = False
        print(f"dpr_scale_amd.dist: the direct RCCL communicator did not come up within {timeout_s:.0f} s; keeping torch.distributed", flush=True)
        return None
    return state["out"]


def disable_direct_comm(group=None):
    c = _DIRECT.pop(_gkey(group), None)
    if c:
        c.close()


class _StreamWork:
    """Handle of a collective issued on the side stream: wait() makes the CURRENT stream wait for it (the host never blocks)."""

    def __init__(self, event, keep):
        self.event, self.keep = event, keep

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)
        self.keep = None


def _on_side_stream(fn, *tensors):
    dev = tensors[0].device
    side = _SIDE.get(dev.index)
    if side is None:
        # HIGH priority: a priority is a property of the hardware queue, so this stream can never be multiplexed onto the compute
        # stream's queue.  (HIP spreads a process's streams over GPU_MAX_HW_QUEUES = 4 hardware queues; two streams that land on one
        # queue run in order -- profiles/r05_overlap_busy_*: torch.distributed's RCCL stream shared the compute stream's queue on the
        # test box and its 400 us collective ran with the compute stream idle, where this side stream ran 12-35 tower kernels under it.)
        side = _SIDE[dev.index] = torch.cuda.Stream(device=dev, priority=-1)
    side.wait_stream(torch.cuda.current_stream(dev))  # the operands were produced on the compute stream
    with torch.cuda.stream(side):
        fn()
        ev = side.record_event()
    for t in tensors:
        t.record_stream(side)  # the caching allocator must not hand the memory out again before the side stream is done with it
    return _StreamWork(ev, tensors)


# ---- debugging aid for timelines on ONE GPU (scripts/overlap_trace.py --pad-mb): a one-rank world carries each collective as a 2-5 us
# device copy, over before the tower's first kernel has been launched.  DPRHOT_DEBUG_PAD_MB=N issues, right behind every all-gather /
# reduce-scatter of the path and through the same transport, the same collective on a scratch buffer of N MiB per rank, and the handle
# the caller waits on covers both: the collective then TAKES TIME (1 GiB ~ 0.4 ms of device copy), and whether the tower's kernels
# really run underneath it can be read off a rocprofv3 trace.  Never set in production.
_PAD = {}


class _Both:
    def __init__(self, *works):
        self.works = [w for w in works if w is not None]

    def wait(self):
        for w in self.works:
            w.wait()


def _pad_mb():
    try:
        return int(os.environ.get("DPRHOT_DEBUG_PAD_MB", "0") or 0)
    except ValueError:
        return 0


def _pad_bufs(device, W, mb):
    key = (device, W, mb)
    if key not in _PAD:
        rows, d = mb * 1024 * 1024 // (2 * 1024), 1024  # bf16 rows of 2 KiB
        _PAD[key] = (torch.zeros((rows, d), dtype=torch.bfloat16, device=device), torch.empty((W * rows, d), dtype=torch.bfloat16, device=device),
                     torch.zeros((W * rows, d), dtype=torch.bfloat16, device=device), torch.empty((rows, d), dtype=torch.bfloat16, device=device))
    return _PAD[key]


def all_gather_rows(send: torch.Tensor, out: torch.Tensor, group=None, async_op=False):
    """out[r*n:(r+1)*n] = send from rank r (see _all_gather_rows)."""
    w = _all_gather_rows(send, out, group, async_op)
    mb = _pad_mb()
    if mb > 0 and send.is_cuda:
        ps, po, _, _ = _pad_bufs(send.device, world(group)[0], mb)
        w2 = _all_gather_rows(ps, po, group, async_op)
        return _Both(w, w2) if async_op else None
    return w


def reduce_scatter_rows(inp: torch.Tensor, out: torch.Tensor, group=None, async_op=False):
    """out = sum over ranks of inp[r*n:(r+1)*n] for this rank r (see _reduce_scatter_rows)."""
    w = _reduce_scatter_rows(inp, out, group, async_op)
    mb = _pad_mb()
    if mb > 0 and inp.is_cuda:
        _, _, pi, pm = _pad_bufs(inp.device, world(group)[0], mb)
        w2 = _reduce_scatter_rows(pi, pm, group, async_op)
        return _Both(w, w2) if async_op else None
    return w


def _all_gather_rows(send: torch.Tensor, out: torch.Tensor, group=None, async_op=False):
    """out[r*n:(r+1)*n] = send from rank r (equal n on every rank -- guaranteed upstream by
    ContiguousDistributedSampler padding, utils.py:48-60, and DPRTransform padding, dpr_transform.py:143-161)."""
    W, _ = world(group)
    assert out.shape[0] == W * send.shape[0], (out.shape, send.shape, W)
    comm = direct_comm(group)
    if comm is not None and send.is_cuda:
        fn = (lambda: comm.all_gather_allpairs(send, out)) if (_allpairs(group) and comm.has_allpairs) else (lambda: comm.all_gather_rows(send, out))
        if async_op:
            return _on_side_stream(fn, send, out)
        fn()
        return None
    if _allpairs(group):
        s2, o2 = (send.view(torch.float16), out.view(torch.float16)) if (send.dtype == torch.bfloat16 and not _is_nccl(group)) else (send, out)
        if _is_nccl(group):  # grouped send/recv of the ONE send buffer to every peer: no staging copy
            return dist.all_to_all(list(o2.chunk(W, dim=0)), [s2] * W, group=group, async_op=async_op)
        return dist.all_to_all_single(o2, s2.repeat(W, *([1] * (s2.dim() - 1))), group=group, async_op=async_op)  # gloo (CPU tests)
    if send.dtype == torch.bfloat16 and not _is_nccl(group):
        # gloo has no bf16: ship the bit patterns in a 2-byte type it knows (all-gather only moves bytes)
        return dist.all_gather_into_tensor(out.view(torch.float16), send.view(torch.float16), group=group,
                                           async_op=async_op)
    return dist.all_gather_into_tensor(out, send, group=group, async_op=async_op)


def _reduce_scatter_rows(inp: torch.Tensor, out: torch.Tensor, group=None, async_op=False):
    """out = sum over ranks of inp[r*n:(r+1)*n] for this rank r."""
    W, r = world(group)
    n = out.shape[0]
    assert inp.shape[0] == W * n
    comm = direct_comm(group)
    ap_comm = _allpairs(group) and comm is not None and comm.has_allpairs
    if comm is not None and inp.is_cuda and inp.dtype in comm.KINDS and (inp.dtype == out.dtype or (ap_comm and out.dtype == torch.float32)) \
            and (not ap_comm or out.numel() % 8 == 0):
        if ap_comm:
            tmp = torch.empty_like(inp)  # chunk k: what rank k computed for MY columns (summed in fp32, rank order, by the library)
            fn = lambda: comm.reduce_scatter_allpairs(inp, tmp, out)  # noqa: E731
            tensors = (inp, tmp, out)
        else:
            fn = lambda: comm.reduce_scatter_rows(inp, out)  # noqa: E731
            tensors = (inp, out)
        if async_op:
            return _on_side_stream(fn, *tensors)
        fn()
        return None
    if _allpairs(group):
        tmp = torch.empty_like(inp)  # chunk k: what rank k computed for MY columns
        bytes_only = inp.dtype == torch.bfloat16 and not _is_nccl(group)
        work = dist.all_to_all_single(tmp.view(torch.float16) if bytes_only else tmp, inp.view(torch.float16) if bytes_only else inp,
                                      group=group, async_op=async_op)

        def add_up():
            # fp32 accumulation in rank order r = 0 .. W-1, ONE rounding into out's type: the same arithmetic as the C-ABI form
            # (dprhot_grad_sum_shards -- on the device it IS that kernel: no W x n x d fp32 copy of a bf16 wire, no run-dependent order)
            if tmp.is_cuda and tmp.dtype in DirectComm.KINDS and out.dtype in DirectComm.KINDS and out.numel() % 8 == 0 and tmp.is_contiguous() and out.is_contiguous():
                import ctypes

                from . import _lib
                _lib.check(_lib.lib.dprhot_grad_sum_shards(tmp.data_ptr(), int(W), out.numel(), DirectComm.KINDS[tmp.dtype], DirectComm.KINDS[out.dtype],
                                                           out.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "dprhot_grad_sum_shards")
                return
            chunks = tmp.view(W, n, *inp.shape[1:])
            acc = chunks[0].float().clone()
            for kk in range(1, W):
                acc += chunks[kk].float()
            out.copy_(acc)

        if async_op:
            return _Then(work, add_up)
        add_up()
        return None
    if _is_nccl(group):
        return dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    # gloo (CPU tests): no reduce-scatter -- all-reduce then keep the own slice (bf16: gloo cannot add it; the rounded values
    # are summed in fp32 and rounded once more, which bounds what RCCL's bf16 additions do)
    tmp = inp.float() if inp.dtype == torch.bfloat16 else inp.clone()
    dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
    out.copy_(tmp[r * n:(r + 1) * n])
    return None


def all_reduce_sum(t: torch.Tensor, group=None, async_op=False):
    comm = direct_comm(group)
    if comm is not None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and not async_op:
        comm.all_reduce_sum(t)  # on the caller's stream: ordered behind the kernel that wrote it, in front of whoever reads it
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class DirectComm:
    """The path's collectives issued straight on the caller's HIP stream through the C ABI's optional communicator
    (dprhot_comm_* / dprhot_allgather_ctx / dprhot_reducescatter_dc / dprhot_allreduce_sum: RCCL taken from the
    process with dlopen).  No hand-over to RCCL's stream and back, ~3x less host time per call than torch.distributed.
    Build it with try_direct_comm(): construction is COLLECTIVE and falls back as a group."""

    def __init__(self, world_size, rank, unique_id):
        import ctypes

        from . import _lib

        self._lib, self.W, self.rank = _lib, world_size, rank
        self.has_allpairs = True  # try_direct_comm's self-check turns it off, on every rank alike, where the exchanges do not work
        h = ctypes.c_void_p()
        _lib.check(_lib.lib.dprhot_comm_init(unique_id, world_size, rank, ctypes.byref(h)), "dprhot_comm_init")
        self.h = h

    @staticmethod
    def new_unique_id():
        import ctypes

        from . import _lib

        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.lib.dprhot_comm_unique_id(buf), "dprhot_comm_unique_id")
        return bytes(buf.raw)

    @staticmethod
    def _stream():
        import ctypes

        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def all_gather_rows(self, send, out):
        assert send.is_contiguous() and out.is_contiguous() and out.numel() == self.W * send.numel()
        self._lib.check(self._lib.lib.dprhot_allgather_ctx(self.h, send.data_ptr(), out.data_ptr(),
                                                           send.numel() * send.element_size(), self._stream()),
                        "dprhot_allgather_ctx")

    KINDS = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}

    def reduce_scatter_rows(self, inp, out):
        assert inp.dtype == out.dtype and inp.dtype in self.KINDS and inp.numel() == self.W * out.numel()
        assert inp.is_contiguous() and out.is_contiguous()
        self._lib.check(self._lib.lib.dprhot_reducescatter_rows(self.h, inp.data_ptr(), out.data_ptr(), out.numel(), self.KINDS[inp.dtype],
                                                                self._stream()), "dprhot_reducescatter_rows")

    def all_gather_allpairs(self, send, out):
        assert send.is_contiguous() and out.is_contiguous() and out.numel() == self.W * send.numel()
        self._lib.check(self._lib.lib.dprhot_allgather_allpairs(self.h, send.data_ptr(), out.data_ptr(),
                                                                send.numel() * send.element_size(), self._stream()),
                        "dprhot_allgather_allpairs")

    def reduce_scatter_allpairs(self, inp, tmp, out):
        assert inp.dtype in self.KINDS and tmp.dtype == inp.dtype and out.dtype in (inp.dtype, torch.float32)
        assert inp.numel() == self.W * out.numel() and tmp.numel() == inp.numel() and inp.is_contiguous() and tmp.is_contiguous() and out.is_contiguous()
        self._lib.check(self._lib.lib.dprhot_reducescatter_allpairs(self.h, inp.data_ptr(), tmp.data_ptr(), out.data_ptr(), out.numel(),
                                                                    self.KINDS[inp.dtype], self.KINDS[out.dtype], self._stream()),
                        "dprhot_reducescatter_allpairs")

    def all_reduce_sum(self, t):
        assert t.dtype == torch.float32 and t.is_contiguous()
        self._lib.check(self._lib.lib.dprhot_allreduce_sum(self.h, t.data_ptr(), t.numel(), self._stream()),
                        "dprhot_allreduce_sum")

    def close(self):
        if self.h is not None:
            self._lib.lib.dprhot_comm_destroy(self.h)
            self.h = None


def try_direct_comm(device, group=None, alive=None, handshake=None):
    """COLLECTIVE over `group` (an initialised nccl group): every rank gets a DirectComm, or every rank gets None.
    Each stage is agreed on through torch.distributed before the next collective stage starts, and the new
    communicator has to reproduce torch.distributed's all-gather and reduce-scatter on test data -- at the self-check's small size and at
    the size of a cfg3 step's messages, in both forms (RCCL collective, all-pairs) -- before it is used.
    `handshake`: the torch.distributed group every collective of this set-up runs on (same ranks as `group`; enable_direct_comm passes
    a group of its own so that an abandoned set-up can never interleave with the training step's collectives); default `group`.
    `alive` (the watchdog): once it returns False no further collective is issued from here."""
    W, r = world(group)
    if not (dist.is_available() and dist.is_initialized()) or not _is_nccl(group):
        return None
    alive = alive or (lambda: True)
    hs = handshake if handshake is not None else group

    def all_ok(flag):
        if not alive():
            return False
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=hs)
        return bool(t.item()) and alive()

    try:
        uid = DirectComm.new_unique_id()  # every rank probes the library; only rank 0's id is used
    except Exception:
        uid = None
    if not all_ok(uid is not None):
        return None
    box = [uid]
    dist.broadcast_object_list(box, src=dist.get_global_rank(hs, 0) if hs is not None else 0, group=hs)
    comm = None
    try:
        comm = DirectComm(W, r, box[0])
    except Exception:
        comm = None
    if not all_ok(comm is not None):
        if comm is not None:
            comm.close()
        return None
    # self-check against torch.distributed: a small case, then the message sizes of a cfg3 step (1032 packed rows x 768).  RCCL's own
    # collectives decide whether there is a communicator at all; the all-pairs exchanges (ncclSend / ncclRecv groups) are checked
    # SEPARATELY and only decide whether THIS communicator offers that form (a library without send/recv is "no all-pairs",
    # not "no communicator") -- each verdict agreed on by all ranks.
    good, good_ap = True, True
    cases = []
    try:
        g = torch.Generator(device="cpu").manual_seed(77 + r)
        for rows, cols in ((24, 16), (1032, 768)):
            if not alive():
                break
            send = torch.randn(rows, cols, generator=g).to(device).to(torch.bfloat16)
            a, b = (torch.empty((W * rows, cols), dtype=torch.bfloat16, device=device) for _ in range(2))
            comm.all_gather_rows(send, a)
            dist.all_gather_into_tensor(b, send, group=hs)
            part = torch.randn(W * rows, cols, generator=g).to(device)
            m1, m2 = (torch.empty((rows, cols), device=device) for _ in range(2))
            comm.reduce_scatter_rows(part, m1)
            dist.reduce_scatter_tensor(m2, part, op=dist.ReduceOp.SUM, group=hs)
            torch.cuda.synchronize()
            good = good and bool(torch.equal(a, b)) and bool(torch.allclose(m1, m2, rtol=1e-5, atol=1e-5))
            cases.append((send, b, part, m2))
        s1 = torch.full((1,), float(r + 1), device=device)
        comm.all_reduce_sum(s1)
        torch.cuda.synchronize()
        good = good and abs(s1.item() - W * (W + 1) / 2) < 1e-3
    except Exception:
        good = False
    if not all_ok(good):
        comm.close()
        return None
    # (a local pre-flight first: a rank whose library lacks the entry points must say so BEFORE its peers enter a send/recv group)
    try:
        from . import _lib

        pre = bool(_lib.lib.dprhot_comm_has_allpairs(comm.h)) if hasattr(_lib.lib, "dprhot_comm_has_allpairs") else True
    except Exception:
        pre = False
    if all_ok(pre):
        try:
            for send, b, part, m2 in cases:
                if not alive():
                    break
                a2 = torch.empty_like(b)
                comm.all_gather_allpairs(send, a2)
                tmp, m3 = torch.empty_like(part), torch.empty_like(m2)
                comm.reduce_scatter_allpairs(part, tmp, m3)
                torch.cuda.synchronize()
                good_ap = good_ap and bool(torch.equal(a2, b)) and bool(torch.allclose(m3, m2, rtol=1e-5, atol=1e-5))
        except Exception:
            good_ap = False
        comm.has_allpairs = all_ok(good_ap)
    else:
        comm.has_allpairs = False
    if not alive():
        comm.close()
        return None
    return comm
