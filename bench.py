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
    ap.add_argument("--driver", choices=["auto", "graph", "graph10", "eager"], default="auto",
                    help="issue mechanism behind `value`: auto (default) = the fastest median of the three below, named in config.driver; "
                         "eager = one C-ABI call per step from a prepared argument block, what a C / C++ binding of include/dprhot.h pays "
                         "(the Python autograd operator pays more on the host: `operator` block) -- within 1 us of host-bound at cfg2, it "
                         "moves between 9.2 and 11.5 us per step with the host core the process landed on; graph / graph10 = HIP graphs "
                         "of 1 / 10 steps (graph10 is device-bound: 9.3 us on every box).  The mechanisms not chosen are measured too and "
                         "reported in other_driver")
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
        self.nl = self._lib.fwd_no_logits(B, Nc, d)
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
    sys.modules.setdefault("bench", sys.modules[__name__])  # bench_blocks.py imports the RUNNING bench.py under this name, not a second copy
    import bench_blocks as BB  # the extra-information blocks (never `value`)
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
    import dpr_scale_amd
    dpr_scale_amd.configure_runtime()  # (HIP_FORCE_DEV_KERNARG is set above already; TORCH_NCCL_HIGH_PRIORITY must precede the process group)
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
            variants = BB.variants_block(a, W, rank, dev, backend, B, K, d, T, hp, measure, state)
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
        # names: auto by default = the best median of the three (round 6: the eager loop's pace is the host core's -- 9.2 us on one
        # box, 11.1 on the next, same kernels -- and the 10-step graph's is the device's); the others ride in `other_driver`.
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
            extra("operator", lambda: BB.operator_block(dev, d))
        if not DM and "torch_gpu" in a.blocks:
            extra("torch_gpu_hot_path", lambda: BB.torch_gpu_block(dev, d))
        if not DM and "grad_hook" in a.blocks:
            extra("grad_hook", lambda: BB.grad_hook_block(dev))
        if not DM and "scale" in a.blocks:
            if d % 64 == 0:  # (before the 8192^2 block: these short steps read 3-4 us longer right behind its minutes of power-limited GEMMs)
                extra("band", lambda: BB.band_block(dev, d))
            out["roofline_at_scale"] = BB.roofline_at_scale(dev, d)
            extra("hipblaslt_same_box", lambda: BB.hipblaslt_same_box(dev, d))
        if not DM and "model" in a.blocks and d % 128 == 0 and not os.environ.get("DPRHOT_FORCE_DIST"):
            extra("scaling_model", lambda: BB.scaling_model(dev, d))
        if not DM and "rank" in a.blocks and d % 128 == 0:
            try:
                out["roofline_cfg3_rank"] = BB.roofline_cfg3_rank(dev, d)
            except Exception as e:  # extra info only
                out["roofline_cfg3_rank"] = {"error": repr(e)}
        if not DM and "router" in a.blocks:
            try:
                out["router"] = BB.roofline_router(dev)
            except Exception as e:  # extra info only
                out["router"] = {"error": repr(e)}
        # The driver's record keeps the contract keys only: the figures a reader needs from the extra blocks ride INSIDE `roofline`
        #  -- one rank's share of BASELINE configs[2], and the kernel families at a size where a roofline applies
        r3 = out.get("roofline_cfg3_rank")
        if isinstance(r3, dict) and "step_us" in r3:
            roof["cfg3_rank"] = {"step_us": r3["step_us"], "frac": r3["frac"], "bound": "hbm", "traffic_over_algorithmic": r3.get("traffic_over_algorithmic"),
                                 "dq_reduction": r3.get("dq_reduction")}
        rs, hb = out.get("roofline_at_scale"), out.get("hipblaslt_same_box")
        if isinstance(rs, dict) and "sim_gemm" in rs:
            def f(k, what):
                return rs[k].get(what) if isinstance(rs.get(k), dict) else None
            roof["at_scale"] = {"shape": "8192 x 8192 x %d" % d, "bound": "mfma", "checked": rs.get("checked"), "timing": rs.get("timing"),
                                "sim_gemm_us": f("sim_gemm", "us"), "sim_gemm_frac": f("sim_gemm", "frac"),
                                "dscores_us": f("dscores_gemm", "us"), "dscores_frac": f("dscores_gemm", "frac"),
                                "backward_us": f("backward_gemms", "us"), "backward_frac": f("backward_gemms", "frac"),
                                "forward_one_pass_us": f("forward_one_pass", "us"), "forward_two_pass_us": rs.get("forward_two_pass_us"),
                                "step_us": rs.get("step_us"), "step_mfma_frac": rs.get("step_mfma_frac"),
                                "hipblaslt_us": f("hipblaslt_matmul", "us"), "hipblaslt_frac": f("hipblaslt_matmul", "frac"),
                                "hipblaslt_alone_us": hb["8192x8192"]["us"] if isinstance(hb, dict) and "8192x8192" in hb else None}
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
