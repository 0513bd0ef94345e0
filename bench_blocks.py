"""bench_blocks.py -- the EXTRA-INFORMATION blocks of bench.py's JSON line (never `value`): the autograd operator, the torch hot
path, the gradient hook's local legs, the router width, one rank's share of cfg3, the scaling model, the library GEMM on the same box,
the per-kernel rooflines at 8192 x 8192, and the transport x form x wire matrix of the N > 1 line.  bench.py holds what the contract
times (HotPathStep, the timed loop, the contract keys); everything here runs AFTER `value` was measured and reports, never decides.
"""
import ctypes
import json
import os
import sys
import time

import numpy as np  # noqa: F401
import torch
import torch.distributed as dist  # noqa: F401


from bench import HBM_PEAK_GBS, MFMA_PEAK_TFLOPS, ROOT, HotPathStep, P, capture, core_line, time_kernel, timed_loop  # noqa: F401


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
    out["dq_reduction"] = ("slice-normalised slabs + finishing launch (bit-reproducible); option sk_dq_atomic (fp32 atomics, no finishing launch) "
                           "measured 2 us SLOWER at this shape: profiles/r06_sk_atomic.txt")
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
        base = [sys.executable, os.path.join(ROOT, "bench.py"), "--only", "step", "--steps", "100", "--repeats", "5", "--batch", str(B), "--negatives", str(K - 1)]
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


def band_block(dev, d=768, shapes=((256, 8192), (512, 8192), (1024, 8192), (1024, 4096), (2048, 4096), (256, 32768))):
    """Extra information (never `value`): the one-call training step (dprhot_inbatch_step_f32: fp32 embeddings in, loss / dQ / dC out) at
    the shapes between the BASELINE configs and 8192^2 -- a few hundred to a few thousand query rows against thousands of contexts on one
    GPU (SURVEY 8(d)'s sweep rows; profiles/r06_sweep.jsonl holds the full table).  Each shape is CHECKED before it is timed: loss, dQ and
    dC against fp32 torch autograd on the same bf16-rounded operands (dpr_task.py:197-212); an unchecked shape reports no time."""
    from dpr_scale_amd import _lib

    out = {"timing": "HIP events around 20 steps per graph replay x 5, as time_kernel", "unit": "us per step", "shapes": {}}
    for B, Nc in shapes:
        hp = HotPathStep(B, Nc // B, d, 1.0, 1, 0, dev)
        hp.k_step()
        torch.cuda.synchronize()
        q = hp.q.to(torch.bfloat16).float().requires_grad_(True)
        c = hp.c.to(torch.bfloat16).float().requires_grad_(True)
        S = (q @ c.T).masked_fill(hp.mask_all.bool()[None, :], float("-inf")) * hp.inv_T
        ref = torch.nn.functional.cross_entropy(S, hp.y, reduction="sum")
        (ref * (hp.gscale / hp.inv_T)).backward()
        go = float(hp.go.item())
        err = {"loss": abs(float(hp.loss_sum.item()) - float(ref.item())) / abs(float(ref.item())),
               "dQ": float((hp.dQ - q.grad * go).abs().max() / (q.grad * go).abs().max()),
               "dC": float((hp.dC - c.grad * go).abs().max() / (c.grad * go).abs().max())}
        ok = err["loss"] <= 1e-3 and err["dQ"] <= 1e-2 and err["dC"] <= 1e-2
        row = {"checked": bool(ok), "check_errors": {k: float("%.3g" % v) for k, v in err.items()},
               "forward": ("logits stored", "one pass, 256 x 256 tile", "one pass, 128 x 128 tile")[_lib.fwd_one_pass(B, Nc, d)]}
        if ok:
            us = time_kernel(hp, hp.k_step, reps=20, iters=5)
            bn, bd, nd = float(B) * Nc, float(B) * d, float(Nc) * d
            by = (4 * bd + 4 * nd) + (2 * (bd + nd) + 4 * bn) + 4 * bn + 6 * (bd + nd)  # bench_sweep.py's step bytes (G written once, read twice)
            row.update({"step_us": round(us, 2), "hbm_floor_us": round(by / (HBM_PEAK_GBS * 1e3), 2), "mfma_frac": round(6.0 * B * Nc * d / us * 1e-6 / MFMA_PEAK_TFLOPS, 4)})
        out["shapes"]["%dx%d" % (B, Nc)] = row
        del hp, q, c, S, ref
        torch.cuda.empty_cache()
    return out


def check_at_scale(hp, S_full, k_simfwd):
    """What roofline_at_scale times, verified BEFORE it is timed (the 0.40 / 0.36 / 0.43 figures were of unverified work).
    Every launch of the block runs once on the block's own operands; fp32 torch on the SAME bf16 operands is the checker
    (dpr_task.py:98-105, :209-212 and its autograd): the loss over all rows, and 64 sampled rows of logsumexp, dScores, dQ, dC, stored
    logits; the score-free ranks of the sampled rows bit-exact against a count over the stored logits.  Bars: 1e-3 (loss, logsumexp,
    logits), 1e-2 of max |grad| (gradients: G is bf16, as in tests/test_gpu_parity.py), 2^-7 of max |G|."""
    dev, B, Nc = hp.Qb.device, hp.B, hp.Nc
    hp.k_sim(); hp.k_softmax(); hp.k_dscores(); hp.k_bwd(); hp.k_rank(); k_simfwd()
    torch.cuda.synchronize()
    Qf, Cf = hp.Qb.float(), hp.Cb.float()
    S = (Qf @ Cf.t()) * hp.inv_T
    S.masked_fill_(hp.mask_all.bool()[None, :], float("-inf"))
    lse = torch.logsumexp(S, dim=1)
    rows = torch.arange(B, device=dev)
    loss_ref = (lse - S[rows, hp.y]).sum().item()
    Gref = torch.exp(S - lse[:, None])
    Gref[rows, hp.y] -= 1.0
    Gref *= hp.gscale
    gen = torch.Generator(device="cpu").manual_seed(6)
    ri = torch.randperm(B, generator=gen)[:64].to(dev)
    ci = torch.randperm(Nc, generator=gen)[:64].to(dev)

    def rel(a, b):
        return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()

    fin = torch.isfinite(S[ri])
    Sg = S_full[ri]
    gold = Sg[torch.arange(64, device=dev), hp.y[ri]]
    cols = torch.arange(Nc, device=dev)[None, :]
    rank_ref = 1 + ((Sg > gold[:, None]) | ((Sg == gold[:, None]) & (cols < hp.y[ri][:, None]))).sum(dim=1)
    err = {"loss": abs(hp.loss_sum.item() - loss_ref) / max(1.0, abs(loss_ref)),
           "lse": rel(hp.row_lse[ri], lse[ri]),
           "G": rel(hp.G[ri].float(), Gref[ri]),
           "dQ": rel(hp.dQ[ri], Gref[ri] @ Cf),
           "dC": rel(hp.dC[ci], Gref[:, ci].t() @ Qf),
           "stored_logits": rel(Sg[fin], S[ri][fin]) if bool(torch.equal(torch.isfinite(Sg), fin)) else float("inf"),
           "rank_mismatches": int((hp.rank[ri] != rank_ref).sum().item())}
    # the forward as the step runs it (ONE GEMM pass, dprhot_inbatch_fwd with G wanted): its dScores, logsumexp and loss on the same bars
    hp.k_fwd()
    torch.cuda.synchronize()
    err["G_one_pass"] = rel(hp.G[ri].float(), Gref[ri])
    err["lse_one_pass"] = rel(hp.row_lse[ri], lse[ri])
    err["loss_one_pass"] = abs(hp.loss_sum.item() - loss_ref) / max(1.0, abs(loss_ref))
    bars = {"loss": 1e-3, "lse": 1e-3, "G": 2.0 ** -7, "dQ": 1e-2, "dC": 1e-2, "stored_logits": 1e-3, "rank_mismatches": 0,
            "G_one_pass": 2.0 ** -7, "lse_one_pass": 1e-3, "loss_one_pass": 1e-3}
    ok = all(err[k] <= bars[k] for k in bars)
    del S, Gref, Qf, Cf
    return ok, {k: (round(v, 9) if isinstance(v, float) else v) for k, v in err.items()}, bars


def roofline_at_scale(dev, d, B=8192, Nc=8192):
    """Extra information (never `value`): the same kernel families at a size where a roofline means something -- B x Nc =
    8192 x 8192 logits per rank (one large-batch step on one GPU), per-launch HIP-event timing as above.  At this size the
    library runs the no-logits forward: statistics GEMM (sim_gemm) -> logsumexp (lse_loss) -> dScores GEMM that recomputes the
    logits and writes G as bf16 (dscores_gemm) -> the backward pair (backward_gemms); score_free_rank is the validation-side
    count-greater GEMM (gold mini-GEMM + count + finish).  `checked`: check_at_scale on these very operands; fractions are
    only printed for checked work."""
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
    if hp.nl:
        # the forward as a training step runs it (dprhot_inbatch_fwd with G wanted): since round 6 ONE pass of the GEMM (strip statistics +
        # fp16 softmax numerators, Epi8StatsP) + a row kernel (logsumexp, loss, numerators -> bf16 dScores in place) + the loss sum
        rows = rows + (("forward_one_pass", hp.k_fwd, 2 * (bd + nd) + 6 * bn, 2 * bn * d, "mfma"),)
    # the library GEMM on the same operands rides in the SAME rounds (context, never a target): a bare C = A x B^T, bf16 result
    Bt = hp.Cb.t()
    rows = rows + (("hipblaslt_matmul", lambda: torch.matmul(hp.Qb, Bt), 2 * (bd + nd) + 2 * bn, 2 * bn * d, "mfma"),)
    checked = False
    if hp.nl:
        try:
            checked, out["check_errors"], out["check_bars"] = check_at_scale(hp, S_full, k_simfwd)
        except Exception as e:
            out["check_errors"] = {"error": repr(e)}
    out["checked"] = bool(checked)
    # Timing: every arm N launches back to back between two HIP events, the arms ALTERNATING over R rounds in this one process, median
    # of the rounds after the first (guide 5.4 rule 24).  Until round 5 each arm was timed alone, one after the other, after whatever ran
    # before it: on this power-limited part the clock a launch gets depends on what the chip did in the preceding milliseconds, and the
    # statistics GEMM read 101-104 us next to 92.5 for the library GEMM measured minutes later -- interleaved they are 91.5 against 89.5
    # (scratch/gemm_ab.py, profiles/r06_gemm_ab.txt).
    N, R = 10, 5
    for _, fn, _, _, _ in rows:
        fn()
    torch.cuda.synchronize()
    samples = {name: [] for name, *_ in rows}
    for _ in range(R):
        for name, fn, _, _, _ in rows:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(N):
                fn()
            e1.record()
            e1.synchronize()
            samples[name].append(e0.elapsed_time(e1) * 1e3 / N)
    out["timing"] = f"{N} launches back to back per sample, arms alternating over {R} rounds, median of rounds 2..{R}"
    for name, fn, by, fl, bound in rows:
        ts = sorted(samples[name][1:])
        us = ts[len(ts) // 2]
        if bound == "mfma":
            ach = fl / us * 1e-6
            out[name] = {"us": round(us, 1), "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4) if (checked or name == "hipblaslt_matmul") else None}
        else:
            ach = by / us * 1e-3
            out[name] = {"us": round(us, 1), "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4) if checked else None}
        out[name]["us_rounds"] = [round(x, 1) for x in samples[name]]
    if hp.nl:
        out["forward_two_pass_us"] = round(out["sim_gemm"]["us"] + out["lse_loss"]["us"] + out["dscores_gemm"]["us"], 1)
        out["forward_us"] = out["forward_one_pass"]["us"]
        out["step_us"] = round(out["forward_us"] + out["backward_gemms"]["us"], 1)
        out["step_mfma_frac"] = round(6 * bn * d / out["step_us"] * 1e-6 / MFMA_PEAK_TFLOPS, 4) if checked else None
        out["step_flops"] = "6 B Nc d (one forward GEMM + the two backward GEMMs; the two-pass forward of rounds 2-5 spent 8 B Nc d)"
    del hp, S_full
    torch.cuda.empty_cache()
    return out


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


