#!/usr/bin/env python3
"""bench.py -- query-passage pairs/sec of dpr-scale's in-batch contrastive hot path on MI355X.

A "step" is one pass of the hot path over one synthetic NQ-shaped batch whose embeddings are already resident
in HBM: fp32->bf16 cast of q/c -> (W>1: RCCL all-gather of context rows + mask) -> sim (+mask, 1/T, softmax
statistics) -> logsumexp + dScores + loss (-> W>1: all-reduce) -> dC and dQ GEMMs (-> W>1: reduce-scatter of
dC), i.e. everything dpr_task.py:163-212 and its autograd backward do between the encoder outputs and their
gradients.  The step is driven through the C ABI (include/dprhot.h) exactly as a binding would drive it:
dprhot_inbatch_fwd_f32 (fp32 embeddings in, rounded to bf16 while staging) and dprhot_inbatch_bwd -- 3 kernel
launches at N=1.

N=1 workload = BASELINE.json configs[1]: bert-base shapes, batch 32, 1 positive + 7 negatives, d=768, no
all-gather.  N>1: the same per-GPU batch on every rank (weak scaling; the global negatives grow with N).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats R] [--batch B] [--negatives 7] [--dim 768]
                  [--driver graph|graph10|eager] [--no-cpu-baseline] [--no-e2e] [--no-rank-roofline]
--driver eager: one C-ABI call per step; graph: the step's launches captured once into a HIP graph and replayed; graph10: ten
steps per graph (K steps = K // 10 replays + K % 10 eager steps; hipGraphLaunch's fixed cost amortised, pace independent of the
host core); auto (default): an untimed probe picks the fastest of the three on this box (all three rates are reported).
The K-step timed region (barrier + synchronize on both sides) is repeated R times (default 31) and the MEDIAN is
reported (min / max next to it): K = 20 steps of ~10 us are a 0.2 ms region, one measurement of it is noise.
Multi-GPU: `python bench.py --gpus N` launches N ranks itself (torch.distributed.run, one rank per GPU over RCCL) when it
is not already running under a launcher; it refuses to run when fewer than N devices are visible.  Under a launcher
(WORLD_SIZE set) it is one of the ranks:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Kernel arguments in device memory instead of host-coherent memory (a documented ROCm runtime switch, read when the HIP runtime
# initialises): every launch of these microsecond-scale kernels otherwise starts with a scalar load across PCIe -- measured on the
# cfg3-per-rank step, eager C-ABI loop: 31.0 -> 26.0 us (hipGraph replays keep their arguments on the device either way).
# dpr_scale_amd/__init__.py sets the same default for the product; an explicit value in the environment wins.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=32, help="queries per GPU")
    ap.add_argument("--negatives", type=int, default=7)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--driver", choices=["auto", "graph", "graph10", "eager"], default="eager",
                    help="issue mechanism behind `value`: eager (default) = one C-ABI call per step from a prepared argument block, what a C / "
                         "C++ binding of include/dprhot.h pays (the Python autograd operator pays more on the host: `operator` block); "
                         "graph / graph10 = HIP graphs of 1 / 10 steps; auto = the fastest of the three.  The mechanisms not chosen are "
                         "measured too and reported in other_driver")
    ap.add_argument("--only", default="", help="comma list of the extra blocks to run (default: all but e2e5): operator, torch_gpu, scale, rank, "
                                               "router, grad_hook, cpu, e2e, model (scaling_model), e2e5 (BASELINE configs[4]: bert-large "
                                               "towers, seq 512, B 64 -- minutes); `step` = none of them (the contract line alone)")
    ap.add_argument("--no-scale-roofline", action="store_true", help="skip the extra 8192x8192 per-kernel roofline block")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=31, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--e2e", action="store_true", help="(default on; kept for compatibility)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the short end-to-end (bert-base towers) block")
    ap.add_argument("--no-rank-roofline", action="store_true", help="skip the cfg3-per-rank (128 x 8256 x 768) block")
    a = ap.parse_args()
    blocks = {"operator", "torch_gpu", "scale", "rank", "router", "grad_hook", "cpu", "e2e", "model", "e2e5"}
    if a.only:
        want = {x.strip() for x in a.only.split(",") if x.strip()}
        unknown = want - blocks - {"step"}
        if unknown:
            ap.error(f"--only: unknown block(s) {sorted(unknown)}")
        a.blocks = want & blocks
    else:
        a.blocks = set(blocks) - {"e2e5"}
    for flag, names in (("no_scale_roofline", ("scale",)), ("no_cpu_baseline", ("cpu",)), ("no_e2e", ("e2e",)),
                        ("no_rank_roofline", ("rank", "router"))):
        if getattr(a, flag):
            a.blocks -= set(names)
    return a


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class HotPathStep:
    """Pre-allocated buffers + the C-ABI call sequence of one step on one rank."""

    def __init__(self, B, K, d, T, W, r, dev, group=None, comm=None, dist_mode=None):
        from dpr_scale_amd import _lib
        from dpr_scale_amd import dist as D
        from dpr_scale_amd.datamodule.synthetic import unit_logit_embeddings

        self.lib, self._lib, self.D = _lib.lib, _lib, D
        self.B, self.K, self.d, self.T, self.W, self.r, self.group = B, K, d, T, W, r, group
        self.dist = (W > 1) if dist_mode is None else bool(dist_mode)  # packed layout + collectives (also at W = 1 when forced)
        self.comm = comm  # dist.DirectComm (collectives on this stream through the C ABI) or None (torch.distributed)
        self.n_ctx = B * K
        # W>1: every rank ships ONE buffer (context rows + mask bytes in trailing rows, dprhot_pack_ctx); the
        # trailing rows are extra, always-masked columns of the gathered matrix
        self.rows_c = self.n_ctx if not self.dist else _lib.packed_rows(self.n_ctx, d)
        self.Nc = W * self.rows_c
        self.Nq = W * B
        q, c, y, m = unit_logit_embeddings(1234 + r, B, K, d)
        f32, bf16 = torch.float32, torch.bfloat16
        self.q = torch.from_numpy(q).to(dev)
        self.c = torch.from_numpy(c).to(dev)
        self.y = torch.from_numpy(y).to(dev)
        self.m8 = torch.from_numpy(m.astype(np.uint8)).to(dev)
        self.Qb = torch.empty((B, d), dtype=bf16, device=dev)
        self.Cb = torch.empty((self.Nc, d), dtype=bf16, device=dev)
        self.send = self.Cb[:self.n_ctx] if not self.dist else torch.empty((self.rows_c, d), dtype=bf16, device=dev)
        self.mask_all = torch.zeros(self.Nc, dtype=torch.uint8, device=dev)
        self.row_loss = torch.empty(B, dtype=f32, device=dev)
        self.row_lse = torch.empty(B, dtype=f32, device=dev)
        self.loss_sum = torch.empty(1, dtype=f32, device=dev)
        self.G = torch.empty((B, self.Nc), dtype=bf16, device=dev)
        self.dQ = torch.empty((B, d), dtype=f32, device=dev)
        self.dC = torch.empty((self.Nc, d), dtype=f32, device=dev)
        self.dc = self.dC if not self.dist else torch.empty((self.rows_c, d), dtype=f32, device=dev)
        self.go = torch.ones(1, dtype=f32, device=dev)
        self.ws_bytes = _lib.workspace_bytes(B, self.Nc, d)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self.inv_T = 1.0 / T
        self.gscale = self.inv_T / self.Nq
        if not self.dist:
            self.mask_all.copy_(self.m8)
        self.bind_stream()

    def bind_stream(self):
        """(Re)build the argument tuples for the CURRENT torch stream (every buffer is static)."""
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        B, Nc, d, off = self.B, self.Nc, self.d, self.r * self.rows_c
        ws, wsb = P(self.ws), self.ws_bytes
        self.a_prep = (P(self.q), self.q.numel(), P(self.Qb), P(self.c), self.c.numel(), P(self.send), st)
        self.a_fwd = (P(self.Qb), B, P(self.Cb), Nc, d, P(self.y), off, P(self.mask_all), self.inv_T, self.gscale, None,
                      P(self.row_loss), P(self.row_lse), P(self.loss_sum), P(self.G), ws, wsb, st)
        self.a_fwd32 = (P(self.q), P(self.c) if not self.dist else None, P(self.Qb), P(self.Cb), B, Nc, d, P(self.y), off,
                        P(self.mask_all), self.inv_T, self.gscale, None, P(self.row_loss), P(self.row_lse), P(self.loss_sum),
                        P(self.G), ws, wsb, st)
        self.a_sim32 = (P(self.q), P(self.c) if not self.dist else None, P(self.Qb), P(self.Cb), B, Nc, d, P(self.y), off,
                        P(self.mask_all), self.inv_T, None, ws, wsb, st)
        self.a_pack = (P(self.c), P(self.m8), self.n_ctx, d, P(self.send), st)
        self.a_unpack = (P(self.Cb), self.W, self.n_ctx, d, P(self.mask_all), st)
        self.a_bwd = (P(self.G), P(self.Qb), P(self.Cb), B, Nc, d, 1.0, P(self.go), P(self.dQ), P(self.dC), ws, wsb, st)
        # the whole step in one C call (dprhot_inbatch_step_f32): 2 launches at the small shapes, else fwd_f32 + bwd
        # the dScores are an OUTPUT nobody reads in a training step: where the shape's plan can do without them (dprhot_step_wants_g:
        # the few-rows plan, cfg3 per rank) the step is called with G == NULL, as the autograd operator calls it (DPRHOT_BENCH_G=1: A/B)
        self.want_g = os.environ.get("DPRHOT_BENCH_G") == "1" or self._lib.step_wants_g(B, Nc, d)
        g_arg = P(self.G) if self.want_g else None
        self.a_step = (P(self.q), P(self.c) if not self.dist else None, P(self.Qb), P(self.Cb), B, Nc, d, P(self.y), off,
                       P(self.mask_all), self.inv_T, self.gscale, 1.0, P(self.go), None, P(self.row_loss), P(self.row_lse),
                       P(self.loss_sum), g_arg, P(self.dQ), P(self.dC), ws, wsb, st)
        self.small = (B <= 32 and Nc <= 1152 or B <= 64 and Nc <= 256) and d % 16 == 0  # mirrors small_step_ok() in csrc/dprhot.hip
        # N > 1: everything between the all-gather and the reduce-scatter in one call (mask read from the packed buffer,
        # loss numerator riding in dC_part); DPRHOT_UNPACKED=1 keeps the separate unpack launch and the loss all-reduce
        self.packed_step = self.dist and not os.environ.get("DPRHOT_UNPACKED")
        self.a_pstep = (P(self.q), P(self.Cb), P(self.Qb), B, self.W, self.r, self.n_ctx, d, P(self.y), self.inv_T,
                        self.gscale, 1.0, P(self.go), P(self.row_loss), P(self.row_lse), P(self.loss_sum), g_arg, P(self.dQ),
                        P(self.dC), ws, wsb, st)
        self.a_sim = (P(self.Qb), B, P(self.Cb), Nc, d, P(self.y), off, P(self.mask_all), self.inv_T, None, ws, wsb, st)
        self.c_step = (self._prepared(self.lib.dprhot_inbatch_step_packed_f32, self.a_pstep) if self.packed_step
                       else self._prepared(self.lib.dprhot_inbatch_step_f32, self.a_step))
        # no-logits shapes (large B x Nc: the workspace then holds no logit buffer): softmax_finish only derives logsumexp / loss
        # from the strip statistics, the dScores come from a third launch that recomputes the logits (dprhot_dscores)
        self.nl = self.ws_bytes < 4 * B * Nc
        self.a_fin = (None, B, Nc, d, P(self.y), off, self.gscale, P(self.row_loss), P(self.row_lse), P(self.loss_sum),
                      None if self.nl else P(self.G), ws, wsb, st)
        self.a_dsc = (P(self.Qb), B, P(self.Cb), Nc, d, P(self.y), off, P(self.mask_all), self.inv_T, self.gscale, None, P(self.G),
                      ws, wsb, st)
        self.rank = torch.empty(B, dtype=torch.int64, device=self.G.device)
        self.a_rank = (P(self.Qb), B, P(self.Cb), Nc, d, P(self.y), off, P(self.mask_all), self.inv_T, P(self.rank), ws, wsb, st)

    def _call(self, fn, args):
        rc = fn(*args)
        if rc:
            self._lib.check(rc, fn.__name__)

    def _prepared(self, fn, args):
        """The argument block converted ONCE into ctypes instances of the prototype's types, and the same symbol through a handle
        without argtypes (instances pass as they are).  ctypes spends ~3 us per call converting 24 Python objects otherwise -- a
        third of the cfg2 step, which is two kernel launches long; a C or C++ caller of the ABI never pays that."""
        if not hasattr(HotPathStep, "_raw"):
            HotPathStep._raw = ctypes.CDLL(self._lib.LIB_PATH)
        raw = getattr(HotPathStep._raw, fn.__name__)
        raw.restype = ctypes.c_int
        assert len(fn.argtypes) == len(args), fn.__name__
        conv = tuple(t(v.value if isinstance(v, ctypes._SimpleCData) else v) for t, v in zip(fn.argtypes, args))
        name = fn.__name__

        def call():
            rc = raw(*conv)
            if rc:
                self._lib.check(rc, name)
        return call

    # the launches of one step
    def k_prep(self):
        self._call(self.lib.dprhot_prep, self.a_prep)

    def k_fwd(self):
        self._call(self.lib.dprhot_inbatch_fwd, self.a_fwd)

    def k_bwd(self):
        self._call(self.lib.dprhot_inbatch_bwd, self.a_bwd)

    def k_sim(self):
        self._call(self.lib.dprhot_sim_stats, self.a_sim)

    def k_fwd32(self):
        self._call(self.lib.dprhot_inbatch_fwd_f32, self.a_fwd32)

    def k_sim32(self):
        self._call(self.lib.dprhot_sim_stats_f32, self.a_sim32)

    def k_pack(self):
        self._call(self.lib.dprhot_pack_ctx, self.a_pack)

    def k_unpack(self):
        self._call(self.lib.dprhot_unpack_mask, self.a_unpack)

    def k_softmax(self):
        self._call(self.lib.dprhot_softmax_finish, self.a_fin)

    def k_dscores(self):
        self._call(self.lib.dprhot_dscores, self.a_dsc)

    def k_rank(self):
        self._call(self.lib.dprhot_sim_rank, self.a_rank)

    def k_step(self):
        self.c_step()

    def step(self):
        # fp32 encoder outputs go straight into the sim kernel; only the rows that travel over xGMI are cast first.  The collectives
        # are the product's own (dpr_scale_amd.dist): torch.distributed, or -- once dist.enable_direct_comm() has run, exactly as
        # DenseRetrieverTask does before its first step -- the C ABI communicator on this very stream.
        if self.dist:
            self.k_pack()
            self.D.all_gather_rows(self.send, self.Cb, self.group)  # the one forward collective
            if not self.packed_step:
                self.k_unpack()
        self.k_step()  # forward + backward of the local rows: one call into the library
        if self.dist:
            self.D.reduce_scatter_rows(self.dC, self.dc, self.group)  # the one backward collective; dc[n_ctx][0] = global loss numerator
            if not self.packed_step:
                # the loss numerator (one float).  Plain call: enqueued behind the reduce-scatter, nothing on the host waits for it
                self.D.all_reduce_sum(self.loss_sum, self.group)


def _host_identity():
    """CPU model, logical CPUs and the cost of one trivial ctypes call into the library (ns): what the eager driver's step time varies with."""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    from dpr_scale_amd import _lib
    fn = _lib.lib.dprhot_version
    for _ in range(2000):
        fn()
    t0 = time.perf_counter()
    for _ in range(20000):
        fn()
    return {"cpu": model, "logical_cpus": os.cpu_count(), "ctypes_call_ns": round((time.perf_counter() - t0) / 20000 * 1e9, 1)}


def capture(hp, fn, repeat=1):
    """Capture `repeat` back-to-back invocations of fn into one HIP graph; returns the replay callable."""
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        hp.bind_stream()
        fn()  # warm (hipFuncSetAttribute etc. must not happen under capture)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        hp.bind_stream()
        for _ in range(repeat):
            fn()
    hp.bind_stream()
    return g.replay


def time_kernel(hp, fn, reps=50, iters=20, use_graph=True):
    """Average duration of one launch of `fn`, HIP events on the launch stream, launches back to back on the
    device (50 per graph replay, so the host launch rate does not enter the number).  use_graph=False (multi-rank
    runs: no capture next to a live RCCL communicator): plain eager loop, a host-launch-rate upper bound."""
    if use_graph:
        run = capture(hp, fn, reps)
    else:
        def run():
            for _ in range(reps):
                fn()
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * reps)  # us


def cpu_baseline(B, K, d, T, budget_s=10.0):
    """The plain-C port of the same step (oracle/inbatch_oracle.c) on this box's host cores."""
    from oracle import c_oracle
    from oracle.inbatch_oracle import synth_embeddings

    lib = c_oracle.load()
    q, c, y, m = synth_embeddings(1234, B, K, d, "U", False)
    c_oracle.train_step(lib, q, c, y, 0, m, T, B)  # warm
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        c_oracle.train_step(lib, q, c, y, 0, m, T, B)
        n += 1
    el = time.perf_counter() - t0
    return {"value": B * n / el, "unit": "query-passage pairs/s", "cores": int(lib.oracle_num_threads()), "kind": "port",
            "formulation": "plain-C restatement of the step (oracle/inbatch_oracle.c, OpenMP)",
            "sample": f"{n} steps of the same B={B} K={K} d={d} workload in {el:.1f} s"}


def cpu_baseline_reference(B, K, d, T, budget_s=6.0):
    """The reference's own formulation (dpr_task.py:197-212: mask.repeat -> matmul -> masked fill -> /T -> CrossEntropyLoss ->
    backward) in the reference's own library, torch CPU ops, on this box's host cores (oracle/torch_steps.py; /root/reference
    does not exist on the GPU box, tests/test_oracle_golden.py pins the restatement to the reference's fixtures bit for bit)."""
    from oracle.inbatch_oracle import synth_embeddings
    from oracle.torch_steps import time_reference_step

    q, c, y, m = synth_embeddings(1234, B, K, d, "U", False)
    best = None
    for nth in (min(8, torch.get_num_threads()),):  # 8 host threads, stated in `cores`
        torch.set_num_threads(nth)
        med, n = time_reference_step(q, c, y, m, T, budget_s=budget_s / 2)
        if best is None or med < best[0]:
            best = (med, n, nth)
    med, n, nth = best
    return {"value": B / med, "unit": "query-passage pairs/s", "cores": nth, "kind": "reference-ops",
            "formulation": "the reference's own ops in its own library (torch CPU, fp32), dpr_task.py:197-212 line by line: mask.repeat, "
                           "matmul, masked fill, /T, CrossEntropyLoss, autograd backward (oracle/torch_steps.py, pinned to the reference's "
                           "fixtures bit for bit; /root/reference itself does not exist on the GPU box).  SURVEY.md section 6 probed 1.78 ms "
                           "per step for the reference CLASS at this shape: training_step there also builds the [Nq, Nc] mask with a Python "
                           "loop, splices and concatenates the gathered lists and logs -- host bookkeeping around the same ops; this leg "
                           "times the ops alone (the part the hot path replaces), so it is the faster, stricter baseline",
            "sample": f"median of {n} steps of the same B={B} K={K} d={d} workload ({med * 1e3:.3f} ms/step, torch {torch.__version__}, "
                      f"{nth} threads)"}


def _event_us(run, iters):
    """Average microseconds per call of `run`, HIP events on the current stream."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def _graph_of(fn, repeat):
    """`repeat` calls of fn captured into one HIP graph through torch (allocations inside come from the graph's pool)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(repeat):
            fn()
    return g


def _time_python_step(fn, reps=10, iters=30):
    """(graph_us, eager_us) per call of a Python-level step: replayed from a HIP graph holding `reps` calls (device time, the host
    out of the picture) and issued eagerly (what a training loop pays when nothing else hides the host)."""
    for _ in range(5):
        fn()
    eager = _event_us(fn, iters * 4)
    try:
        g = _graph_of(fn, reps)
        for _ in range(3):
            g.replay()
        graph = _event_us(g.replay, iters) / reps
    except Exception as e:  # a step that cannot be captured is reported as such, never hidden
        graph = None
        sys.stderr.write(f"bench.py: graph capture failed: {e!r}\n")
    return graph, eager


def operator_block(dev, d=768):
    """What TRAINING runs (dpr_task.py:153-214 under AMP): hotpath.inbatch_contrastive_loss forward + (loss * 1024).backward() --
    the autograd operator with a non-unit grad_output -- next to the bare C-ABI step of the same shape, at cfg2 (B 32 x 256 contexts)
    and at the cfg3-per-rank shape (B 128 x 8192 contexts).  HIP events; `graph` = ten steps per HIP graph (device time), `eager` =
    one Python call per step (host-bound at these sizes).  `autograd_floor_eager_us`: the same forward + backward(grad) call pattern
    around a Python autograd.Function that launches nothing (`..._cpp_node_...`: around a C++ node that launches nothing -- the floor
    of the operator's own node); `four_torch_launches_eager_us`: four in-place adds on 8 floats."""
    from dpr_scale_amd import hotpath
    from dpr_scale_amd.datamodule.synthetic import unit_logit_embeddings

    out = {"what": "autograd operator forward + backward with grad_output = 1024 (as under AMP) vs the C-ABI step alone; us per step"}
    for name, B, K in (("cfg2", 32, 8), ("cfg3_rank_shape", 128, 64)):
        q, c, y, m = unit_logit_embeddings(1234, B, K, d)
        tq = torch.from_numpy(q).to(dev).requires_grad_(True)
        tc = torch.from_numpy(c).to(dev).requires_grad_(True)
        ty, tm = torch.from_numpy(y).to(dev), torch.from_numpy(m).to(dev)
        scale = torch.full((), 1024.0, device=dev)

        def op_step():
            tq.grad = None
            tc.grad = None
            loss = hotpath.inbatch_contrastive_loss(tq, tc, ty, tm, 1.0)
            loss.backward(scale)

        g_us, e_us = _time_python_step(op_step)

        # What the eager number is made of that no operator can avoid: torch's autograd machinery around a Python Function that launches
        # NOTHING (same call pattern, gradients from the caching allocator), and the host cost of the step's launches by themselves.
        class _Floor(torch.autograd.Function):
            @staticmethod
            def forward(ctx, a, b):
                ctx.shapes = (a.shape, b.shape)
                return torch.empty((), device=a.device)

            @staticmethod
            def backward(ctx, go):
                return torch.empty(ctx.shapes[0], device=go.device), torch.empty(ctx.shapes[1], device=go.device)

        def floor_step():
            tq.grad = None
            tc.grad = None
            _Floor.apply(tq, tc).backward(scale)

        one = torch.zeros(8, device=dev)

        def launches_step(n=4):  # (the operator's step is three or four library launches + one in backward)
            for _ in range(n):
                one.add_(1.0)

        for _ in range(5):
            floor_step()
            launches_step()
        floor_us = _event_us(floor_step, 120)
        cfloor_us = None
        if getattr(hotpath, "_OPX_NODE", False) and hasattr(hotpath._OPX, "floor_loss"):
            def cfloor_step():
                tq.grad = None
                tc.grad = None
                hotpath._OPX.floor_loss(tq, tc).backward(scale)

            for _ in range(5):
                cfloor_step()
            cfloor_us = _event_us(cfloor_step, 120)
        launch_us = _event_us(launches_step, 120)
        hp = HotPathStep(B, K, d, 1.0, 1, 0, dev)
        abi = time_kernel(hp, hp.k_step, reps=10, iters=30)
        out[name] = {"shape": f"B={B} x Nc={B * K} x d={d}", "operator_graph_us": None if g_us is None else round(g_us, 2),
                     "operator_eager_us": round(e_us, 2), "autograd_floor_eager_us": round(floor_us, 2),
                     "autograd_floor_cpp_node_eager_us": None if cfloor_us is None else round(cfloor_us, 2),
                     "operator_node": "C++ (csrc/opx.cpp: InBatchFn)" if getattr(hotpath, "_OPX_NODE", False) else "Python (hotpath.InBatchContrastive)",
                     "four_torch_launches_eager_us": round(launch_us, 2), "c_abi_step_us": round(abi, 2),
                     "operator_minus_c_abi_us": None if g_us is None else round(g_us - abi, 2),
                     "operator_over_c_abi": None if g_us is None else round(g_us / abi, 3)}
        del hp
    torch.cuda.empty_cache()
    return out


def torch_gpu_block(dev, d=768):
    """Context, never `value`: the reference's formulation of the hot path ALONE (dpr_task.py:197-212 + autograd backward) in torch
    ops on this same MI355X -- fp32 as written, and under torch.autocast(bf16) as its AMP recipes run it -- at cfg2 and at
    128 x 8192.  Same timing as operator_block."""
    from dpr_scale_amd.datamodule.synthetic import unit_logit_embeddings

    out = {"what": "reference ops (mask.repeat, matmul, masked fill, /T, CrossEntropyLoss, backward with grad_output = 1024) in torch on "
                   "this GPU; us per step"}
    for name, B, K in (("cfg2", 32, 8), ("cfg3_rank_shape", 128, 64)):
        q, c, y, m = unit_logit_embeddings(1234, B, K, d)
        tq = torch.from_numpy(q).to(dev).requires_grad_(True)
        tc = torch.from_numpy(c).to(dev).requires_grad_(True)
        ty, tm = torch.from_numpy(y).to(dev), torch.from_numpy(m).to(dev)
        scale = torch.full((), 1024.0, device=dev)
        loss_fn = torch.nn.CrossEntropyLoss()
        res = {"shape": f"B={B} x Nc={B * K} x d={d}"}
        for tag, amp in (("fp32", False), ("autocast_bf16", True)):
            def ref_step():
                tq.grad = None
                tc.grad = None
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                    mask = tm.repeat(tq.size(0), 1)
                    scores = torch.matmul(tq, torch.transpose(tc, 0, 1))
                    scores[mask] = float("-inf")
                    scores /= 1.0
                    loss = loss_fn(scores, ty)
                loss.backward(scale)

            g_us, e_us = _time_python_step(ref_step)
            res[tag] = {"graph_us": None if g_us is None else round(g_us, 2), "eager_us": round(e_us, 2)}
        out[name] = res
    torch.cuda.empty_cache()
    return out


def grad_hook_block(dev, n=110_000_000, W=8):
    """the three local legs of the towers' gradient all-reduce (dpr_scale_amd/comm_hooks.py; reference hook
    dpr_task.py:90-92) on one bert-base tower's gradients (110 M fp32 elements = one tower in one bucket), as rank 0 of an 8-rank
    node would run them: pack (fp32 -> wire, x 1/W), sum of the W received shards (fp32 accumulation), unpack (wire -> fp32).  The
    exchange itself needs W GPUs and is not timed here.  HBM roofline against 8 TB/s."""
    from dpr_scale_amd import comm_hooks

    legs = comm_hooks._HipLegs()
    out = {"workload": f"{n} fp32 gradient elements, W = {W} (shard = n / W), legs on one GPU; exchange not included",
           "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    buf = torch.randn(n, device=dev)
    shard = ((n + W - 1) // W + 7) // 8 * 8
    npad = shard * W
    for wire in (torch.float16, torch.bfloat16):
        send = torch.empty(npad, dtype=wire, device=dev)
        recv = torch.empty(npad, dtype=wire, device=dev)
        legs.pack(buf, 1.0 / W, send)
        recv.copy_(send)
        mine = torch.empty(shard, dtype=wire, device=dev)
        full = torch.empty(npad, dtype=wire, device=dev)
        es = wire.itemsize
        rows = (("pack", lambda: legs.pack(buf, 1.0 / W, send), 4.0 * n + es * npad),
                ("sum_shards", lambda: legs.sum_shards(recv, W, mine), es * npad + es * shard),
                ("unpack", lambda: legs.unpack(full, buf), es * n + 4.0 * n))
        res = {}
        for name, fn, by in rows:
            for _ in range(3):
                fn()
            us = _event_us(fn, 20)
            res[name] = {"us": round(us, 1), "achieved": round(by / us * 1e-3, 1), "frac": round(by / us * 1e-3 / HBM_PEAK_GBS, 4)}
        res["legs_total_us"] = round(sum(v["us"] for v in res.values()), 1)
        # the torch formulation these kernels replaced (round 2's comm_hooks): buf / W -> .to(wire); view.float().sum(0).to(); copy back
        def torch_legs():
            s2 = (buf / W).to(wire)
            m2 = recv.view(W, shard).float().sum(dim=0).to(wire)
            buf.copy_(full[:n])
            return s2, m2
        for _ in range(2):
            torch_legs()
        res["torch_ops_total_us"] = round(_event_us(torch_legs, 5), 1)
        out[str(wire).replace("torch.", "")] = res
        del send, recv, mine, full
    # What plain streams reach on THIS box (the library's own copy / fill kernels through torch): the practical ceiling of a leg
    # that is two thirds writes (unpack: 2 bytes read, 4 written per element) is the write stream's, not 8 TB/s.
    dst = torch.empty_like(buf)
    half = torch.empty(n, dtype=torch.bfloat16, device=dev)
    same = {}
    for name, fn, by in (("fill_fp32", lambda: dst.zero_(), 4.0 * n), ("copy_fp32", lambda: dst.copy_(buf), 8.0 * n),
                         ("widen_bf16_to_fp32", lambda: dst.copy_(half), 6.0 * n)):
        for _ in range(3):
            fn()
        us = _event_us(fn, 20)
        same[name] = {"us": round(us, 1), "achieved": round(by / us * 1e-3, 1), "frac": round(by / us * 1e-3 / HBM_PEAK_GBS, 4)}
    out["torch_streams_same_box"] = same
    del buf, dst, half
    torch.cuda.empty_cache()
    return out



def roofline_router(dev, B=128, K=8, d=30528):
    """Extra information (never `value`): the CITADEL router loss (citadel_task.py:249-262) is the same Q x C^T +
    CrossEntropyLoss on vocabulary-wide vectors (d = 30522, zero-padded to 30528): the reference's most arithmetic-heavy use of
    the path (57 flop/byte at B = 128 -- still left of the ridge).  One in-batch step (forward + backward), per launch and as a whole."""
    hp = HotPathStep(B, K, d, 1.0, 1, 0, dev)
    bn = float(B) * hp.Nc
    out = {"workload": f"router vectors: B={B} x Nc={hp.Nc} x d={d} (30522 padded), fp32 in, one in-batch step"}
    tot = 0.0
    for name, fn, fl in (("sim_stats_f32", hp.k_sim32, 2 * bn * d), ("softmax_finish", hp.k_softmax, 0.0), ("bwd_pair", hp.k_bwd, 4 * bn * d)):
        us = time_kernel(hp, fn, reps=10, iters=5)
        tot += us
        out[name] = {"us": round(us, 2), "TFLOPs": round(fl / us * 1e-6, 1), "mfma_frac": round(fl / us * 1e-6 / MFMA_PEAK_TFLOPS, 4)}
    step = time_kernel(hp, hp.k_step, reps=10, iters=5)
    # Which roof: 6 * B * Nc * d flops against fp32 q / c read once, their bf16 copies written and read back by the backward
    # (same bytes as re-reading fp32), fp32 dQ / dC written, logits and G: 12 * (B + Nc) * d + 12 * B * Nc bytes.  At B = 128
    # that is 57 flop/byte, far left of the ridge (312 flop/byte): the step is HBM-bound, the MFMA share is reported beside it.
    algo = 12.0 * (B + hp.Nc) * d + 12.0 * bn
    out.update({"step_us": round(step, 2), "bound": "hbm", "algorithmic_bytes": algo, "achieved": round(algo / step * 1e-3, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algo / step * 1e-3 / HBM_PEAK_GBS, 4),
                "flops": 6 * bn * d, "flop_per_byte": round(6 * bn * d / algo, 1),
                "mfma_frac": round(6 * bn * d / step * 1e-6 / MFMA_PEAK_TFLOPS, 4)})
    del hp
    torch.cuda.empty_cache()
    return out


def roofline_cfg3_rank(dev, d=768, B=128, K=8, W=8):
    """Extra information (never `value`): ONE rank's share of BASELINE configs[2] (8 x MI355X, batch 128 per GPU, 8192 global
    negatives) timed on this GPU -- the gathered packed buffer [W * rows_c, d] is filled locally, then
    dprhot_inbatch_step_packed_f32 (everything between the all-gather and the reduce-scatter) is replayed.  Algorithmic bytes:
    SURVEY.md section 8(d), unfused 3-kernel figure with q read as fp32."""
    hp = HotPathStep(B, K, d, 1.0, W, 0, dev, dist_mode=True)
    hp.k_pack()
    for r in range(W):
        hp.Cb[r * hp.rows_c:(r + 1) * hp.rows_c].copy_(hp.send)
    torch.cuda.synchronize()
    us = time_kernel(hp, hp.k_step, reps=20, iters=10)
    us_with_g = None
    if not hp.want_g:  # the same step with the dScores asked for (the four-launch plan of rounds 2-3), same process, same box
        os.environ["DPRHOT_BENCH_G"] = "1"
        try:
            hp.bind_stream()
            us_with_g = round(time_kernel(hp, hp.k_step, reps=20, iters=10), 2)
        finally:
            del os.environ["DPRHOT_BENCH_G"]
            hp.bind_stream()
    bn, bd, nd = float(B) * hp.Nc, float(B) * d, float(hp.Nc) * d
    algo = (4 * bd + 2 * nd + 4 * bn) + 6 * bn + (2 * bn + 2 * nd + 4 * bd) + (2 * bn + 2 * bd + 4 * nd)
    # the same step as the autograd operator issues it under DDP: dprhot_train_step_packed_f32 (loss mean out of the kernel, dQ left
    # as split-K slabs) + dprhot_rescale_grads (adds the slabs up, checks grad_output) -- with fp32 partials and with the bf16 wire of
    # the reduce-scatter written by the dC epilogue (2 bytes instead of 4 per dC element: 53.8 MB of algorithmic traffic)
    lib, _lib = hp.lib, hp._lib
    nsl = _lib.train_dq_slabs(B, hp.Nc, d)
    part = torch.empty((max(nsl, 1), B, d), dtype=torch.float32, device=dev)
    out2 = torch.empty(2, dtype=torch.float32, device=dev)
    op_us = {}
    for wire, kind in (("fp32_wire", 2), ("bf16_wire", 0)):
        dCw = hp.dC if kind == 2 else torch.empty((hp.Nc, d), dtype=torch.bfloat16, device=dev)

        def train_step():
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            rc = lib.dprhot_train_step_packed_f32(P(hp.q), P(hp.Cb), P(hp.Qb), B, W, 0, hp.n_ctx, d, P(hp.y), hp.inv_T, hp.gscale, 1.0 / hp.Nq,
                                                  P(hp.go), P(hp.row_loss), P(hp.row_lse), P(hp.loss_sum), P(hp.G) if hp.want_g else None, P(hp.dQ),
                                                  P(part) if nsl > 0 else None, P(dCw), kind, P(hp.ws), hp.ws_bytes, st)
            rc = rc or lib.dprhot_rescale_grads(P(hp.dQ), hp.dQ.numel(), P(part) if nsl > 0 else None, nsl, P(dCw), dCw.numel(), kind,
                                                P(hp.go), P(hp.go), P(out2), st)
            if rc:
                _lib.check(rc, "train step")

        try:
            op_us[wire] = round(time_kernel(hp, train_step, reps=20, iters=10), 2)
        except Exception as e:
            op_us[wire] = repr(e)
    out = {"workload": f"cfg3 per rank: B={B} rows x Nc={hp.Nc} gathered columns (W={W} x {hp.rows_c} packed rows) x d={d}, "
                       "bf16 contexts resident, fp32 q in, fp32 dQ / dC_part out",
           "step_us": round(us, 2), "launches": "sim (tile-local softmax out) | dC + dQ units deriving the row logsumexp themselves | dQ slab sum"
           if us_with_g is not None else "sim | dScores | dC + dQ units | dQ slab sum",
           "step_us_with_dscores_launch": us_with_g, "pairs_per_s_per_gpu": round(B / us * 1e6, 1), "bound": "hbm",
           "algorithmic_bytes": algo, "achieved": round(algo / us * 1e-3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(algo / us * 1e-3 / HBM_PEAK_GBS, 4), "flops": 6 * bn * d,
           "mfma_frac": round(6 * bn * d / us * 1e-6 / MFMA_PEAK_TFLOPS, 4),
           "operator_path_step_us": op_us, "operator_path": "dprhot_train_step_packed_f32 (dQ slabs deferred) + dprhot_rescale_grads"}
    if isinstance(op_us.get("fp32_wire"), float):
        out["operator_path_frac_fp32_wire"] = round(algo / op_us["fp32_wire"] * 1e-3 / HBM_PEAK_GBS, 4)
    if isinstance(op_us.get("bf16_wire"), float):
        out["operator_path_frac_bf16_wire"] = round((algo - 2 * nd) / op_us["bf16_wire"] * 1e-3 / HBM_PEAK_GBS, 4)
    out["traffic"], out["traffic_source"] = None, None
    tfile = os.path.join(ROOT, "profiles", "cfg3rank_traffic.json")
    if (B, K, d, W) == (128, 8, 768, 8) and os.path.isfile(tfile):  # PMC bytes of this very call sequence (scripts/gpu_run.sh prof-rank)
        tj = json.load(open(tfile))
        out["traffic"] = tj.get("step_hbm_bytes")
        out["traffic_source"] = f"profiles/cfg3rank_traffic.json ({tj.get('source')}); per kernel: " + ", ".join(
            f"{k} {v['hbm_bytes_per_launch']}" for k, v in tj.get("kernels", {}).items())
        out["traffic_over_algorithmic"] = None if not out["traffic"] else round(out["traffic"] / algo, 3)
    del hp
    torch.cuda.empty_cache()
    return out


XGMI_LINK_GBS = 153.0   # per direction and link, 7 links per GPU (MI355X_MICROARCH.md / BASELINE.md)
XGMI_LAT_US = 5.0       # assumed small-message latency of one RCCL hop (UNMEASURED here: one-GPU boxes)


def scaling_model(dev, d=768, B=128, K=8):
    """A MODEL, not a measurement (hard part 2; section 8 e): what BASELINE configs[2] -- batch 128 per GPU, K = 8,
    RCCL all-gather of the context rows -- does to the hot path at 2 / 4 / 8 GPUs, built from (a) the per-rank step MEASURED on this GPU
    at the column count N ranks produce, (b) the host cost of the path's collectives MEASURED on a one-rank RCCL world (the same
    bench.py with DPRHOT_FORCE_DIST=1 in a child process), (c) the xGMI message model 7 links x 153 GB/s per direction for a ring
    and for the all-pairs exchange.  Every entry says "modelled": true.  What it warns about: the hot-path-only line DROPS per GPU
    beyond one GPU -- every rank scores its 128 rows against N x 1032 columns (10 -> 29 us) and pays two collectives -- while the
    end-to-end step, two bert-base towers of >100 ms, hides both under the towers."""
    import subprocess

    from dpr_scale_amd import _lib
    out = {"modelled": True, "what": "hot path only (no towers), batch 128 per GPU, K = 8, d = 768: per-rank step measured on ONE MI355X, collectives modelled",
           "link_model": f"{XGMI_LINK_GBS} GB/s per direction per xGMI link, 7 links per GPU, {XGMI_LAT_US} us per hop (assumed); ring = N - 1 sequential hops "
                         "over one link, all-pairs = N - 1 links at once (dist.py DPRHOT_PATH_COLLECTIVES=allpairs)"}
    host = None
    try:
        env = dict(os.environ, DPRHOT_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29771")
        base = [sys.executable, os.path.abspath(__file__), "--only", "step", "--steps", "100", "--repeats", "5", "--batch", str(B), "--negatives", str(K - 1)]
        forced = json.loads(subprocess.run(base, env=env, capture_output=True, text=True, timeout=240).stdout.strip().splitlines()[-1])
        env.pop("DPRHOT_FORCE_DIST")
        plain = json.loads(subprocess.run(base, env=env, capture_output=True, text=True, timeout=240).stdout.strip().splitlines()[-1])
        host = {"one_rank_world_step_us": round(forced["ms_per_step"] * 1e3, 2), "single_process_step_us": round(plain["ms_per_step"] * 1e3, 2),
                "collectives": forced["config"].get("collectives")}
        host["launch_cost_of_the_collectives_us"] = round(host["one_rank_world_step_us"] - host["single_process_step_us"], 2)
    except Exception as e:
        host = {"error": repr(e)}
    out["measured_one_rank_collectives"] = host
    coll_host = host.get("launch_cost_of_the_collectives_us", 20.0) if isinstance(host, dict) else 20.0
    rows = []
    for N in (1, 2, 4, 8):
        hp = HotPathStep(B, K, d, 1.0, N, 0, dev, dist_mode=N > 1)
        if N > 1:
            hp.k_pack()
            for r in range(N):
                hp.Cb[r * hp.rows_c:(r + 1) * hp.rows_c].copy_(hp.send)
        torch.cuda.synchronize()
        step_us = time_kernel(hp, hp.k_step, reps=20, iters=10)
        pack_us = time_kernel(hp, hp.k_pack, reps=20, iters=10) if N > 1 else 0.0
        msg_ag = hp.rows_c * d * 2.0                 # one rank's packed block, bf16
        chunk = {"fp32": hp.rows_c * d * 4.0, "bf16": hp.rows_c * d * 2.0}
        link = XGMI_LINK_GBS * 1e3                   # bytes per us
        e = {"n_gpus": N, "modelled": N > 1, "global_batch": N * B, "global_negatives_per_query": N * B * K - 1, "per_rank_columns": hp.Nc,
             "per_rank_step_us_measured": round(step_us, 2), "pack_us_measured": round(pack_us, 2)}
        if N > 1:
            ag = {"ring": (N - 1) * (XGMI_LAT_US + msg_ag / link), "allpairs": XGMI_LAT_US + msg_ag / link}
            rs = {w: {"ring": (N - 1) * (XGMI_LAT_US + chunk[w] / link), "allpairs": XGMI_LAT_US + chunk[w] / link + N * chunk[w] / 5.0e6} for w in chunk}
            e["all_gather_us"] = {k: round(v, 1) for k, v in ag.items()}
            e["reduce_scatter_us"] = {w: {k: round(v, 1) for k, v in rs[w].items()} for w in rs}
            exposed = {f"{kind}_{w}_wire": pack_us + coll_host + ag[kind] + step_us + rs[w][kind] for kind in ("ring", "allpairs") for w in chunk}
            e["hot_path_step_us_collectives_exposed"] = {k: round(v, 1) for k, v in exposed.items()}
            e["hot_path_pairs_per_s_collectives_exposed"] = {k: round(N * B / v * 1e6, 0) for k, v in exposed.items()}
            e["hot_path_pairs_per_s_collectives_hidden_under_the_towers"] = round(N * B / (pack_us + coll_host + step_us) * 1e6, 0)
        else:
            e["hot_path_pairs_per_s"] = round(B / step_us * 1e6, 0)
        rows.append(e)
        del hp
        torch.cuda.empty_cache()
    out["per_n"] = rows
    one = rows[0]["hot_path_pairs_per_s"]
    out["per_gpu_relative_to_one_gpu"] = {str(r["n_gpus"]): round(r["hot_path_pairs_per_s_collectives_hidden_under_the_towers"] / r["n_gpus"] / one, 3)
                                          for r in rows[1:]}
    out["reading"] = ("the hot-path-only weak-scaling line falls per GPU as N grows (each rank's 128 rows meet N x 1032 columns and two collectives); "
                      "the training step it sits in is two encoder towers of > 100 ms per step, under which both collectives run (end_to_end / "
                      "end_to_end_forced_dist, profiles/r04_overlap_*)")
    return out


def hipblaslt_same_box(dev, d=768):
    """Context for roofline_at_scale, never a target: the library GEMM (torch.matmul, bf16, hipBLASLt / rocBLAS) at the same shapes on the
    same box -- a bare C = A x B^T with fp32 accumulate and a bf16 result, no mask, no softmax statistics, no epilogue of the path."""
    out = {}
    for name, M, N in (("8192x8192", 8192, 8192), ("128x8192", 128, 8192)):
        A = torch.randn(M, d, device=dev).to(torch.bfloat16)
        Bm = torch.randn(N, d, device=dev).to(torch.bfloat16)
        Bt = Bm.t()
        for _ in range(5):
            torch.matmul(A, Bt)
        us = _event_us(lambda: torch.matmul(A, Bt), 50)
        fl = 2.0 * M * N * d
        out[name] = {"us": round(us, 2), "tflops": round(fl / us * 1e-6, 1), "frac_of_bf16_peak": round(fl / us * 1e-6 / MFMA_PEAK_TFLOPS, 4),
                     "what": f"torch.matmul bf16 [{M},{d}] x [{d},{N}] -> bf16 (writes {M * N * 2 / 1e6:.0f} MB)"}
    return out


def roofline_at_scale(dev, d, B=8192, Nc=8192):
    """Extra information (never `value`): the same kernel families at a size where a roofline means something -- B x Nc =
    8192 x 8192 logits per rank (one large-batch step on one GPU), per-launch HIP-event timing as above.  At this size the
    library runs the no-logits forward: statistics GEMM (sim_gemm) -> logsumexp (lse_loss) -> dScores GEMM that recomputes the
    logits and writes G as bf16 (dscores_gemm) -> the backward pair (backward_gemms); score_free_rank is the validation-side
    count-greater GEMM (gold mini-GEMM + count + finish)."""
    hp = HotPathStep(B, Nc // B, d, 1.0, 1, 0, dev)
    bn, bd, nd = float(B) * Nc, float(B) * d, float(Nc) * d
    hp.k_prep()
    out = {"workload": f"B={B} x Nc={Nc} x d={d} (bf16 operands resident), per launch",
           "forward_plan": "no-logits (stats GEMM -> lse -> dScores GEMM)" if hp.nl else "logits stored (sim GEMM -> streaming softmax)"}
    if hp.nl:
        rows = (("sim_gemm", hp.k_sim, 2 * (bd + nd) + 8 * bn / 64, 2 * bn * d, "mfma"),
                ("lse_loss", hp.k_softmax, 8 * bn / 64 + 12 * B, 0.0, "hbm"),
                ("dscores_gemm", hp.k_dscores, 2 * (bd + nd) + 2 * bn, 2 * bn * d, "mfma"),
                ("backward_gemms", hp.k_bwd, 4 * bn + 6 * (bd + nd), 4 * bn * d, "mfma"),
                ("score_free_rank", hp.k_rank, 2 * (bd + nd), 2 * bn * d, "mfma"))
    else:
        rows = (("sim_gemm", hp.k_sim, 2 * (bd + nd) + 4 * bn, 2 * bn * d, "mfma"),
                ("softmax_dscores", hp.k_softmax, 6 * bn, 0.0, "hbm"),
                ("backward_gemms", hp.k_bwd, 4 * bn + 6 * (bd + nd), 4 * bn * d, "mfma"))
    # sim_score with the logits wanted (validation through a subclass's own metrics, the head chunk of a retrieval): dprhot_sim_fwd
    S_full = torch.empty((B, Nc), dtype=torch.float32, device=dev)

    def k_simfwd():
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = hp.lib.dprhot_sim_fwd(P(hp.Qb), B, P(hp.Cb), Nc, d, P(hp.mask_all), hp.inv_T, P(S_full), st)
        if rc:
            hp._lib.check(rc, "dprhot_sim_fwd")

    rows = rows + (("sim_store", k_simfwd, 2 * (bd + nd) + 4 * bn, 2 * bn * d, "mfma"),)
    for name, fn, by, fl, bound in rows:
        us = time_kernel(hp, fn, reps=20, iters=3)
        if bound == "mfma":
            ach = fl / us * 1e-6
            out[name] = {"us": round(us, 1), "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4)}
        else:
            ach = by / us * 1e-3
            out[name] = {"us": round(us, 1), "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4)}
    if hp.nl:
        out["forward_us"] = round(out["sim_gemm"]["us"] + out["lse_loss"]["us"] + out["dscores_gemm"]["us"], 1)
        out["step_us"] = round(out["forward_us"] + out["backward_gemms"]["us"], 1)
        out["step_mfma_frac"] = round(8 * bn * d / out["step_us"] * 1e-6 / MFMA_PEAK_TFLOPS, 4)
    del hp, S_full
    torch.cuda.empty_cache()
    return out


def timed_loop(run, steps, W, per_call=1, tail=None):
    """EXACTLY `steps` steps between the two barrier + synchronize pairs.  per_call > 1: `run` is a graph holding per_call steps
    (steps // per_call replays; the remainder, if any, through `tail`, one step per call)."""
    torch.cuda.synchronize()
    if W > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps // per_call):
        run()
    for _ in range(steps % per_call if per_call > 1 else 0):
        tail()
    torch.cuda.synchronize()
    if W > 1:
        dist.barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


class WireStep:
    """One step as the autograd operator issues it under DDP, on HotPathStep's buffers: pack -> all-gather -> dprhot_train_step_packed_f32
    (loss mean out of the kernel, dC partials in the wire format) -> dprhot_rescale_grads -> reduce-scatter in the wire format
    (-> widen to fp32 for a half-width wire) -> all-reduce of the loss means.  The collectives are dpr_scale_amd.dist's, in whatever
    transport / form dist.configure() names."""

    def __init__(self, hp, wire):
        self.hp, self.kind = hp, (2 if wire == "fp32" else 0)
        dev, d = hp.q.device, hp.d
        dt = torch.float32 if self.kind == 2 else torch.bfloat16
        self.dCw = hp.dC if self.kind == 2 else torch.empty((hp.Nc, d), dtype=dt, device=dev)
        self.mine = hp.dc if self.kind == 2 else torch.empty((hp.rows_c, d), dtype=dt, device=dev)
        self.nsl = hp._lib.train_dq_slabs(hp.B, hp.Nc, d)
        self.part = torch.empty((max(self.nsl, 1), hp.B, d), dtype=torch.float32, device=dev)
        self.loss2 = torch.zeros(2, dtype=torch.float32, device=dev)
        self.out2 = torch.empty(2, dtype=torch.float32, device=dev)
        # a plan without a bf16 dC epilogue (the latency-bound shapes: only the few-rows plan has one) writes fp32 partials, and the
        # wire format is produced by one cast launch -- what InBatchContrastive.backward does (hotpath.py: dC_part.to(wire))
        self.native = True
        if self.kind != 2:
            try:
                self._train(self.kind, self.dCw)
            except Exception as e:
                if "dc_kind" not in str(e):
                    raise
                self.native = False

    def _train(self, kind, dC):
        hp, lib = self.hp, self.hp.lib
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        part = P(self.part) if self.nsl > 0 else None
        rc = lib.dprhot_train_step_packed_f32(P(hp.q), P(hp.Cb), P(hp.Qb), hp.B, hp.W, hp.r, hp.n_ctx, hp.d, P(hp.y), hp.inv_T, hp.gscale, 1.0 / hp.Nq,
                                              P(hp.go), P(hp.row_loss), P(hp.row_lse), P(self.loss2), P(hp.G) if hp.want_g else None, P(hp.dQ), part,
                                              P(dC), kind, P(hp.ws), hp.ws_bytes, st)
        rc = rc or lib.dprhot_rescale_grads(P(hp.dQ), hp.dQ.numel(), part, self.nsl, P(dC), dC.numel(), kind, P(hp.go), P(hp.go), P(self.out2), st)
        if rc:
            hp._lib.check(rc, "train step")

    def step(self):
        hp, lib = self.hp, self.hp.lib
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        hp.k_pack()
        hp.D.all_gather_rows(hp.send, hp.Cb, hp.group)
        if self.native:
            self._train(self.kind, self.dCw)
        else:
            self._train(2, hp.dC)
            hp._lib.check(lib.dprhot_cast_bf16(P(hp.dC), P(self.dCw), hp.dC.numel(), st), "dprhot_cast_bf16")
        hp.D.reduce_scatter_rows(self.dCw, self.mine, hp.group)
        if self.kind != 2:
            hp._lib.check(lib.dprhot_grad_unpack(P(self.mine), self.kind, P(hp.dc), hp.dc.numel(), st), "dprhot_grad_unpack")
        hp.D.all_reduce_sum(self.loss2[:1], hp.group)


def variants_block(a, W, rank, dev, backend, B, K, d, T, hp, measure, state):
    """N > 1 (or the forced one-rank world): transport x form x wire, each variant with a cross-rank check BEFORE it is timed -- the loss
    identical on every rank, this rank's c.grad chunk within 8e-3 of max |grad| of the first variant's (torch.distributed, RCCL
    collective, fp32 wire) -- and under its own watchdog: a variant that does not come back in DPRHOT_VARIANT_TIMEOUT seconds (120)
    is reported as timed out, rank 0 prints the line with everything measured so far, and every rank exits."""
    import threading

    from dpr_scale_amd import dist as D

    limit = float(os.environ.get("DPRHOT_VARIANT_TIMEOUT", "120"))
    out, ref = {}, None
    current = {"name": None}

    def give_up():
        if rank == 0 and state.get("line") is not None:
            out[current["name"]] = {"error": f"timed out after {limit:.0f} s"}
            line = dict(state["line"])
            line["variants"] = out
            print(json.dumps(line), flush=True)
        os._exit(0)

    def guarded(name, fn):
        current["name"] = name
        wd = threading.Timer(limit, give_up)
        wd.daemon = True
        wd.start()
        try:
            return fn()
        finally:
            wd.cancel()

    # the C ABI communicator: opt-in in the product, brought up here (collectively, under its own watchdog inside enable_direct_comm
    # plus this block's) so that its four variants can be measured; every rank gets it or none does
    have_direct = False
    env_direct = os.environ.get("DPRHOT_DIRECT_RCCL")
    if backend == "nccl" and os.environ.get("DPRHOT_DIRECT_RCCL", "1") != "0":
        os.environ["DPRHOT_DIRECT_RCCL"] = "1"
        D.configure(None, None)  # (direct=False would hide the communicator from its own set-up's return value)
        have_direct = guarded("direct-comm-setup", lambda: D.enable_direct_comm(dev)) is not None
    reps = max(3, min(7, a.repeats))
    for transport in ("torch.distributed", "c-abi-communicator"):
        for form in ("rccl", "allpairs"):
            for wire in ("fp32", "bf16"):
                name = f"{transport} | {'RCCL collective' if form == 'rccl' else 'all-pairs'} | {wire} dC wire"
                if transport != "torch.distributed" and not have_direct:
                    out[name] = {"skipped": "no C ABI communicator in this world (backend %s)" % backend}
                    continue

                def run_one():
                    nonlocal ref
                    D.configure(topology=form, direct=(transport != "torch.distributed"))
                    ws = WireStep(hp, wire)
                    ws.step()
                    torch.cuda.synchronize()
                    chk = {}
                    loss = ws.loss2[:1].clone()
                    mine_dc = hp.dc[:hp.n_ctx].clone()
                    if W > 1:
                        lc = loss if backend == "nccl" else loss.cpu()  # (gloo gathers host tensors)
                        ls = [torch.empty_like(lc) for _ in range(W)]
                        dist.all_gather(ls, lc)
                        lv = torch.cat(ls)
                        chk["loss_spread"] = float((lv.max() - lv.min()).item())
                    else:
                        chk["loss_spread"] = 0.0
                    chk["loss"] = float(loss.item())
                    if ref is None:
                        ref = (loss.clone(), mine_dc)
                        err = 0.0
                    else:
                        e = (mine_dc - ref[1]).abs().max() / ref[1].abs().max().clamp_min(1e-30)
                        if W > 1:
                            e = e if backend == "nccl" else e.cpu()
                            dist.all_reduce(e, op=dist.ReduceOp.MAX)
                        err = float(e.item())
                    chk["dc_err_vs_first_variant"] = err
                    ok = chk["loss_spread"] <= 1e-6 * max(1.0, abs(chk["loss"])) and err <= 8e-3 and \
                        abs(chk["loss"] - float(ref[0].item())) <= 1e-5 * max(1.0, abs(chk["loss"]))
                    chk["ok"] = bool(ok)
                    if not ok:
                        return {"check": chk, "error": "cross-rank check failed: not timed"}
                    saved = a.repeats
                    a.repeats = reps
                    try:
                        ts = measure(ws.step)
                    finally:
                        a.repeats = saved
                    el = sorted(ts)[len(ts) // 2]
                    return {"check": chk, "ms_per_step": round(el / a.steps * 1e3, 5), "value": round(W * B * a.steps / el, 1), "repeats": len(ts)}

                try:
                    out[name] = guarded(name, run_one)
                except Exception as e:
                    out[name] = {"error": repr(e)}
    D.configure(None, None)
    # The communicator was brought up for this block only.  What runs after it (the end-to-end leg: DDP's own communicator next to the
    # path's collectives) is the product's default configuration, in which the C ABI communicator is opt-in: take it down again, on
    # every rank, and put the environment switch back where the caller had it.
    if have_direct:
        torch.cuda.synchronize()
        D.disable_direct_comm()
    if env_direct is None:
        os.environ.pop("DPRHOT_DIRECT_RCCL", None)
    else:
        os.environ["DPRHOT_DIRECT_RCCL"] = env_direct
    return out


def core_line(a, W, B, K, d, T, hp, els, driver, collectives, backend, DM):
    """The contract's keys from the R timed regions (median); roofline / kernels / extras are added by the caller."""
    es = sorted(els)
    el = es[len(es) // 2]
    return {
        "metric": "query-passage pairs/sec (in-batch contrastive hot path: gather+sim+softmax-CE+dQ/dC)",
        "value": round(W * B * a.steps / el, 1), "unit": "pairs/s", "n_gpus": W, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(el / a.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"cfg2-shaped per GPU: B={B} queries x (1+{a.negatives}) contexts, d={d}, T={T}; "
                               f"global Nq={W * B}, Nc={hp.Nc}; embeddings resident in HBM, step driven through the C ABI",
                   "global_batch": W * B, "global_negatives_per_query": W * hp.n_ctx - 1, "parallelism": f"dp{W}",
                   "driver": driver, "collectives": collectives},
        "roofline": None, "kernels": None, "other_driver": None,
        "timing": {"repeats": len(es), "steps_per_repeat": a.steps, "statistic": "median",
                   "ms_per_step_min": round(es[0] / a.steps * 1e3, 5), "ms_per_step_max": round(es[-1] / a.steps * 1e3, 5),
                   # the eager loop is host-paced (two launches + one ctypes call per step against 8.7 us of kernels): which host
                   # this lease ran on, and how fast it makes a foreign call -- other_driver.graph10 is the number that does not move
                   "host": _host_identity()},
        "rccl_ranks": W if (DM and backend == "nccl") else 0,
        "rccl_version": ".".join(str(x) for x in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
    }


def self_launch(a):
    """`python bench.py --gpus N` outside a launcher: start N ranks, one per GPU, and relay their output."""
    import socket
    import subprocess

    n = torch.cuda.device_count()
    same_dev = bool(os.environ.get("DPRHOT_SAME_DEVICE"))  # debugging aid: every rank on device 0 over gloo
    if n < a.gpus and not same_dev:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} requested but only {n} HIP device(s) are visible; refusing to run fewer ranks "
                         "than asked (a line with a different n_gpus would be a mis-measurement)\n")
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    W = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # debugging aid for boxes with ONE GPU: exercise the N>1 code path with every rank on device 0 over gloo
    # (DPRHOT_DIST_BACKEND=gloo DPRHOT_SAME_DEVICE=1); RCCL itself refuses two ranks on one device
    backend = os.environ.get("DPRHOT_DIST_BACKEND", "nccl")
    if os.environ.get("DPRHOT_SAME_DEVICE"):
        local = 0
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if a.gpus != W:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={W}: the line would report a different n_gpus than was asked for\n")
        sys.exit(2)
    if local >= torch.cuda.device_count():
        sys.stderr.write(f"bench.py: rank {rank} wants device {local} but only {torch.cuda.device_count()} are visible\n")
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # debugging aid for one-GPU boxes: DPRHOT_FORCE_DIST=1 runs the N>1 code path (packed layout, collectives, the C ABI
    # communicator) with a world of ONE rank over real RCCL
    DM = W > 1 or bool(os.environ.get("DPRHOT_FORCE_DIST"))
    if DM:
        if W == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29555")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    B, K, d, T = a.batch, 1 + a.negatives, a.dim, a.temperature
    def measure(fn, per_call=1, tail=None):
        """W warmup steps, then R timed regions of exactly K steps each (barrier + synchronize on both sides), MAX over ranks per
        region; the caller takes the median."""
        for _ in range((a.warmup + per_call - 1) // per_call):
            fn()
        ts = [timed_loop(fn, a.steps, W, per_call, tail) for _ in range(max(1, a.repeats))]
        if W > 1:
            tt = torch.tensor(ts, dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ts = tt.tolist()
        return ts

    comm, torch_coll, variants, collectives_name = None, None, None, "none"
    if DM:
        # N > 1.  `value` is the step in the configuration the PRODUCT defaults to: torch.distributed's own RCCL communicator (the C ABI
        # communicator is opt-in: DPRHOT_DIRECT_RCCL=1) and the FORM of the path's two collectives that dist.choose_path_collectives
        # measures faster on this node (RCCL's all-gather / reduce-scatter, or the direct all-pairs exchange) -- the same probe
        # DenseRetrieverTask.on_pretrain_routine_start runs.  Then, each under its own watchdog, the whole matrix
        #     transport {torch.distributed, C ABI communicator} x form {RCCL collective, all-pairs} x dC wire {fp32, bf16}
        # so that one run on a multi-GPU node yields the comparison, not one number (`variants`).
        import threading

        from dpr_scale_amd import dist as D
        hp = HotPathStep(B, K, d, T, W, rank, dev, comm=None, dist_mode=DM)
        # first RCCL's own collectives through torch.distributed -- the form every PyTorch-ROCm user exercises -- so that a line exists
        # whatever happens next; then the probe (both forms; a watchdog prints that first line if it does not come back); then, if the
        # probe prefers the all-pairs exchange, the step again in that form: `value` is what the product would run on this node
        D.configure(topology="rccl", direct=False)
        els = measure(hp.step)
        name_rccl = "torch.distributed, RCCL all-gather / reduce-scatter, fp32 dC wire"
        state = {"line": core_line(a, W, B, K, d, T, hp, els, "eager", name_rccl + " (the topology probe did not come back)", backend, DM) if rank == 0 else None}
        D.configure(None, False)

        def give_up_probe():
            if rank == 0:
                print(json.dumps(state["line"]), flush=True)
            os._exit(0)

        wdp = threading.Timer(float(os.environ.get("DPRHOT_VARIANT_TIMEOUT", "120")), give_up_probe)
        wdp.daemon = True
        wdp.start()
        topo = D.choose_path_collectives(dev, hp.rows_c, d) if W > 1 else D.path_topology()
        wdp.cancel()
        probe = D._PROBED.get(D._gkey(None))
        note = (f" (probe: RCCL {probe['us']['rccl']:.1f} us, all-pairs {probe['us']['allpairs']:.1f} us per gather + scatter)" if probe else
                (" (DPRHOT_PATH_COLLECTIVES)" if os.environ.get("DPRHOT_PATH_COLLECTIVES") else ""))
        collectives_name = name_rccl + note
        torch_coll = sorted(els)[len(els) // 2]
        if topo == "allpairs":
            els = measure(hp.step)  # (configure(None): the probe's choice is in force)
            collectives_name = "torch.distributed, all-pairs exchange, fp32 dC wire" + note
        state["line"] = core_line(a, W, B, K, d, T, hp, els, "eager", collectives_name, backend, DM) if rank == 0 else None
        if rank == 0 and topo == "allpairs":
            state["line"]["rccl_collective_form"] = {"ms_per_step": round(torch_coll / a.steps * 1e3, 5), "value": round(W * B * a.steps / torch_coll, 1)}
        if not os.environ.get("DPRHOT_NO_VARIANTS"):
            variants = variants_block(a, W, rank, dev, backend, B, K, d, T, hp, measure, state)
        D.configure(None, None)
        driver = "eager"  # the collectives stay outside graphs
        runs = {"eager": (hp.step, 1)}
    else:
        hp = HotPathStep(B, K, d, T, W, rank, dev, comm=None, dist_mode=DM)
        driver = a.driver
        # issue mechanisms of the SAME step: one C call per step (eager), a HIP graph holding one step, a HIP graph holding ten steps
        # (hipGraphLaunch's fixed cost -- more than the two launches it replaces -- amortised; what a caller who captures a whole
        # training iteration gets, and the only one of the three whose pace does not depend on the host core the process landed on:
        # the eager loop is within 1 us of host-bound and flips between 9.9 and 11.5 us per step from run to run)
        runs = {"eager": (hp.c_step, 1)}  # (hp.step without its Python frames: no collectives at N = 1)
        runs["graph"] = (capture(hp, hp.step), 1)
        runs["graph10"] = (capture(hp, hp.step, 10), 10)
        # every mechanism gets the full measurement (W warmup steps, R regions of exactly K steps).  `value` is the one --driver
        # names: eager by default -- one C-ABI call per step is what a binding and the Lightning task do; the others ride in
        # `other_driver` (auto = the best median of the three).
        measured = {name: measure(fn, per, hp.step) for name, (fn, per) in runs.items()}
        med = {name: sorted(v)[len(v) // 2] for name, v in measured.items()}
        if driver == "auto":
            driver = min(med, key=med.get)
        els = measured[driver]

    out = None
    if rank == 0:
        bn, bd, nd = float(B) * hp.Nc, float(B) * d, float(hp.Nc) * d
        sim_bytes = (4 + 2) * bd + (6 * nd if not DM else 2 * nd) + 4 * bn
        if hp.small:
            # two launches: sim (partial-logit slabs), then softmax-CE + dScores + dQ + dC in one kernel.  The second
            # has no entry point of its own: its duration is (both, back to back) - (sim alone)
            kern = {"sim_stats_f32": (hp.k_sim32, sim_bytes, 2 * bn * d),
                    "softmax_bwd_fused": (hp.k_step, 16 * bn + 2 * bn + 6 * (bd + nd), 4 * bn * d)}
        else:
            kern = {  # the launches of one step, in order: (fn, algorithmic HBM bytes, flops) -- DESIGN.md section 4
                "sim_stats_f32": (hp.k_sim32, sim_bytes, 2 * bn * d),
                "softmax_finish": (hp.k_softmax, 6 * bn, 0.0),
                "bwd_pair": (hp.k_bwd, 4 * bn + 6 * (bd + nd), 4 * bn * d),
            }
        ktimes = {}
        for name, (fn, by, fl) in kern.items():
            us = time_kernel(hp, fn, use_graph=not DM)
            if name == "softmax_bwd_fused":
                us = max(us - ktimes["sim_stats_f32"]["us"], 1e-3)
            ktimes[name] = {"us": round(us, 3), "GBps": round(by / us * 1e-3, 1), "TFLOPs": round(fl / us * 1e-6, 2)}
        dom = max(ktimes, key=lambda k: ktimes[k]["us"])
        by = kern[dom][1]
        us = ktimes[dom]["us"]
        traffic, tsrc = None, None
        tfile = os.path.join(ROOT, "profiles", "bench_cfg2_traffic.json")
        if not DM and (B, K, d) == (32, 8, 768) and os.path.isfile(tfile):  # PMC bytes of this very workload (profiles/)
            tj = json.load(open(tfile))
            if dom in tj.get("kernels", {}):
                traffic, tsrc = tj["kernels"][dom]["hbm_bytes_per_launch"], f"profiles/bench_cfg2_traffic.json ({tj['source']})"
        roof = {"kernel": dom, "bound": "hbm", "achieved": round(by / us * 1e-3, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(by / us * 1e-3 / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": tsrc,
                "avg_launch_us": us, "algorithmic_bytes": by,
                "regime": "latency-bound: a whole step moves ~3 MB and 38 MFLOP (0.4 us of HBM time); its cost is the number of "
                          "dependent launches x (kernel boundary + one trip to memory + the dependent work) -- see roofline_at_scale "
                          "for the same kernel families where a roofline applies"}
        alt = None
        if not DM:  # the other issue mechanisms (median of their own R regions; measured once if --driver named a single one)
            alt = {}
            for name, per in (("eager", 1), ("graph", 1), ("graph10", 10)):
                if name == driver:
                    continue
                if name in med:
                    el2 = med[name]
                else:
                    fn = capture(hp, hp.step, per) if name != "eager" else hp.c_step
                    for _ in range(50 // per + 1):
                        fn()
                    el2 = timed_loop(fn, a.steps, 1, per, hp.step)
                alt[name] = {"value": round(B * a.steps / el2, 1), "ms_per_step": round(el2 / a.steps * 1e3, 5)}
        out = core_line(a, W, B, K, d, T, hp, els, driver, collectives_name, backend, DM)
        out.update({"roofline": roof, "kernels": ktimes, "other_driver": alt})
        if DM and state["line"] is not None and "rccl_collective_form" in state["line"]:
            out["rccl_collective_form"] = state["line"]["rccl_collective_form"]
        if variants is not None:
            out["variants"] = variants
        def extra(key, fn):  # extra information only: a failing block is reported, never fatal
            try:
                out[key] = fn()
            except Exception as e:
                out[key] = {"error": repr(e)}

        if not DM and "operator" in a.blocks and d % 8 == 0:
            extra("operator", lambda: operator_block(dev, d))
        if not DM and "torch_gpu" in a.blocks:
            extra("torch_gpu_hot_path", lambda: torch_gpu_block(dev, d))
        if not DM and "grad_hook" in a.blocks:
            extra("grad_hook", lambda: grad_hook_block(dev))
        if not DM and "scale" in a.blocks:
            out["roofline_at_scale"] = roofline_at_scale(dev, d)
            extra("hipblaslt_same_box", lambda: hipblaslt_same_box(dev, d))
        if not DM and "model" in a.blocks and d % 128 == 0 and not os.environ.get("DPRHOT_FORCE_DIST"):
            extra("scaling_model", lambda: scaling_model(dev, d))
        if not DM and "rank" in a.blocks and d % 128 == 0:
            try:
                out["roofline_cfg3_rank"] = roofline_cfg3_rank(dev, d)
            except Exception as e:  # extra info only
                out["roofline_cfg3_rank"] = {"error": repr(e)}
        if not DM and "router" in a.blocks:
            try:
                out["router"] = roofline_router(dev)
            except Exception as e:  # extra info only
                out["router"] = {"error": repr(e)}
        if "cpu" in a.blocks and not DM:
            out["cpu_baseline"] = cpu_baseline_reference(B, K, d, T)   # the faithful one: the reference's ops, 8 threads
            out["cpu_baseline_port"] = cpu_baseline(B, K, d, T, budget_s=6.0)  # the C restatement on every host thread
    if "e2e" in a.blocks and d == 768 and (W == 1 or backend == "nccl"):
        # the END-TO-END number of the north star (bert-base towers, seq_len 256): short, extra information, never `value`.
        # A watchdog prints the line without it if the leg does not come back (a rank lost in a collective must not cost
        # the measured line).
        import threading

        def give_up():
            if rank == 0 and out is not None:
                out["end_to_end"] = {"error": "timed out"}
                print(json.dumps(out), flush=True)
            os._exit(0)

        wd = threading.Timer(240.0, give_up)
        wd.daemon = True
        wd.start()
        try:
            from bench_e2e import end_to_end
            e2e = end_to_end(B, K, d, T, dev, steps=5, warmup=3, world=W, rank=rank)
            if out is not None:
                out["end_to_end"] = e2e
        except Exception as e:  # extra info only
            if out is not None:
                out["end_to_end"] = {"error": repr(e)}
        if W == 1 and not DM and out is not None:
            # the same step through DenseRetrieverTask's MULTI-GPU branch on a one-rank RCCL world (scripts/overlap_trace.py in a child
            # process): packed layout, the all-gather started under the query tower, the reduce-scatter under its backward
            try:
                import subprocess
                # ONE child process, the two branches alternating (three rounds of ten steps each; the first forced round also holds the
                # task's 14-step trial of the tower order): a 125 ms step measured in two different processes differs by more than the
                # +-0.3 ms in question (clock state, allocator history), so both sides of the comparison come from the same process
                r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "forced_dist_breakdown.py"), "--no-prof", "--steps", "10", "--warmup", "16",
                                    "--modes", "forced_auto,single,forced_auto,single,forced_auto,single"],
                                   capture_output=True, text=True, timeout=400, env=dict(os.environ, MASTER_PORT="29773"))
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                if line:
                    fj = json.loads(line[-1])
                    out["end_to_end_forced_dist"] = {
                        "what": "DenseRetrieverTask.training_step through its MULTI-GPU branch on a one-rank RCCL world (packed layout, ContextGather, packed step, "
                                "reduce-scatter, deferred context gradient; tower order chosen by the task's own trial) against the single-device branch, "
                                "alternating in ONE process: " + fj["workload"],
                        "ms_per_step": fj["median_wall_ms_per_step"].get("forced_auto"), "single_branch_ms_per_step_same_process": fj["median_wall_ms_per_step"].get("single"),
                        "forced_minus_single_ms": fj.get("forced_auto_minus_single_ms"), "tower_order_trial": fj.get("tower_order_trial"),
                        "rounds": fj["runs"]}
                else:
                    out["end_to_end_forced_dist"] = {"error": (r.stderr or r.stdout)[-300:]}
            except Exception as e:
                out["end_to_end_forced_dist"] = {"error": repr(e)}
        wd.cancel()
    if "e2e5" in a.blocks and W == 1 and not DM and out is not None:
        # BASELINE configs[4] on one GPU: bert-large towers (d = 1024), seq 512, B = 64, K = 2 (conf/dragon_aws.yaml:19-35), through the
        # task's multi-GPU branch on a one-rank world; the hot path inside it is the 64 x 136 step
        try:
            import subprocess
            r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "overlap_trace.py"), "--large", "--B", "64", "--K", "2", "--seq", "512",
                                "--steps", "3", "--warmup", "2"], capture_output=True, text=True, timeout=900, env=dict(os.environ, MASTER_PORT="29775"))
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out["end_to_end_cfg5"] = json.loads(line[-1]) if line else {"error": (r.stderr or r.stdout)[-300:]}
        except Exception as e:
            out["end_to_end_cfg5"] = {"error": repr(e)}
    # RCCL prints its version banner through C stdio (fully buffered on a pipe: it would surface at process exit, AFTER a
    # line printed from Python).  Every rank pushes its buffers out before the last barrier; rank 0 prints after it, so
    # the JSON line is the last line on the shared stdout.
    if DM:
        torch.cuda.synchronize()
        from dpr_scale_amd import dist as D2
        D2.disable_direct_comm()
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if DM:
        dist.barrier()
        dist.destroy_process_group()
        ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
