#!/usr/bin/env python3
"""Where do the extra milliseconds of DenseRetrieverTask's MULTI-GPU branch go (end_to_end_forced_dist 127.46 ms against
end_to_end 126.02 ms per step)?  One process, one MI355X, a one-rank RCCL world, two random-init bert-base towers (seq 256, B 32, K 8,
bf16 autocast, AdamW), three orders of the SAME step:

  single          training_step's single-device branch: query tower, context tower, operator without collectives
  single_ctx1st   the same operator, context tower FIRST (isolates the order of the towers from the collectives)
  forced_ref      the multi-GPU code path in the reference's tower order (query tower first): collectives exposed, no deferral
  forced          DPRHOT_FORCE_DIST=1: packed layout, ContextGather (all-gather started under the query tower), the packed step,
                  reduce-scatter under the query-tower backward, deferred context gradient (+ widen on a half-width wire)

Per mode: wall time per step over --steps steps, then a torch.profiler pass of --prof steps: device time summed over all kernels, the
compute stream's busy / idle time between a step's first and last kernel, kernels per step, device time of the hot path's own
launches and of the collectives, and HOST time inside the multi-GPU branch's pieces (wrapped with record_function here, not in the
product).  Prints one JSON object (-> profiles/r05_forced_dist_breakdown.json)."""
import argparse
import functools
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prof", type=int, default=3)
    ap.add_argument("--wire", default="bf16")
    ap.add_argument("--modes", default="single,single_ctx1st,forced,single,forced")
    ap.add_argument("--no-prof", action="store_true", help="wall times only (bench.py's end_to_end_forced_dist block: forced_auto against single, one process)")
    a = ap.parse_args()
    os.environ["DPRHOT_DC_WIRE"] = a.wire
    os.environ["DPRHOT_DIRECT_RCCL"] = "0"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29757")
    import torch
    import torch.distributed as dist
    from torch.profiler import ProfilerActivity, profile, record_function

    from bench_e2e import BERT_BASE, _batch
    from dpr_scale_amd import hotpath, lightning_compat
    from dpr_scale_amd.hydra_compat import Conf
    from dpr_scale_amd.task.dpr_task import DenseRetrieverTask

    import dpr_scale_amd

    dpr_scale_amd.configure_runtime()  # before the first HIP call (kernel arguments on the device) and before the process group exists
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    model_conf = Conf({"_target_": "dpr_scale_amd.models.hf_model.HFEncoder", "model_path": dict(BERT_BASE), "dropout": 0.1})
    optim_conf = Conf({"_target_": "torch.optim.AdamW", "lr": 3e-5})
    torch.manual_seed(0)
    task = DenseRetrieverTask(None, model_conf, None, optim_conf, shared_model=False)
    task.trainer = lightning_compat.Trainer(device=dev, max_steps=1000)
    task.trainer.strategy = lightning_compat.DDPStrategy()
    task.setup("fit")
    task.to(dev).train()
    batch = _batch(32, 8, 256, dev)
    opt = torch.optim.AdamW(task.parameters(), lr=3e-5)

    # ---- labels on the branch's pieces (host time of each; the product carries none of this)
    def label(obj, name, tag):
        fn = getattr(obj, name)

        @functools.wraps(fn)
        def wrapped(*args, **kw):
            with record_function(tag):
                return fn(*args, **kw)
        setattr(obj, name, wrapped)

    label(hotpath.ContextGather, "__init__", "mg:ContextGather(pack + async all-gather)")
    label(hotpath.ContextGather, "wait", "mg:ContextGather.wait")
    label(hotpath, "defer_context_grad", "mg:defer_context_grad")
    label(hotpath, "inbatch_contrastive_loss", "op:inbatch_contrastive_loss(forward)")
    label(hotpath.InBatchContrastive, "backward", "op:InBatchContrastive.backward")
    label(hotpath._DeferContextGrad, "backward", "mg:_DeferContextGrad.backward(wait + widen)")
    label(task, "encode_contexts", "tower:context forward")
    label(task, "encode_queries", "tower:query forward")

    def step_single():
        os.environ["DPRHOT_FORCE_DIST"] = "0"
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return task.training_step(batch, 0)

    def step_single_ctx1st():
        os.environ["DPRHOT_FORCE_DIST"] = "0"
        with torch.autocast("cuda", dtype=torch.bfloat16):
            c = task.encode_contexts(batch["contexts_ids"])
            q = task.encode_queries(batch["query_ids"])
            return hotpath.inbatch_contrastive_loss(q, c, batch["pos_ctx_indices"], batch["ctx_mask"], 1.0, False, None)

    os.environ["DPRHOT_TOWER_ORDER"] = "context_first"  # (the orders are chosen per mode here, not by the task's own trial)

    def step_forced():
        os.environ["DPRHOT_FORCE_DIST"] = "1"
        task.context_tower_first = True
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return task.training_step(batch, 0)

    def step_forced_ref():  # the multi-GPU code path in the REFERENCE's tower order: both collectives exposed, big tower's backward first
        os.environ["DPRHOT_FORCE_DIST"] = "1"
        task.context_tower_first = False
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return task.training_step(batch, 0)
        finally:
            task.context_tower_first = True

    def step_forced_auto():  # the product's default: the task's own trial picks the order over its first 14 steps, then keeps it
        os.environ["DPRHOT_FORCE_DIST"] = "1"
        os.environ["DPRHOT_TOWER_ORDER"] = "auto"
        task.context_tower_first = True
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return task.training_step(batch, 0)
        finally:
            os.environ["DPRHOT_TOWER_ORDER"] = "context_first"

    fns = {"single": step_single, "single_ctx1st": step_single_ctx1st, "forced": step_forced, "forced_ref": step_forced_ref, "forced_auto": step_forced_auto}

    def full(fn):
        loss = fn()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    out = {"workload": "2 x bert-base (random init), seq 256, B 32, K 8, bf16 autocast, AdamW; one MI355X, one-rank RCCL world; dC wire " + a.wire,
           "runs": []}
    for mode in a.modes.split(","):
        fn = fns[mode]
        for _ in range(a.warmup):
            full(fn)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            full(fn)
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t0) / a.steps * 1e3
        if a.no_prof:
            out["runs"].append({"mode": mode, "wall_ms_per_step": round(wall_ms, 3)})
            continue
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for _ in range(a.prof):
                full(fn)
            torch.cuda.synchronize()
        dev_ev, host = [], {}
        for e in prof.events():
            if e.device_type == torch.autograd.DeviceType.CUDA:
                dev_ev.append((e.time_range.start, e.time_range.end, e.name, getattr(e, "device_index", 0)))
            elif e.name.startswith(("mg:", "op:", "tower:")):
                host[e.name] = host.get(e.name, 0.0) + (e.time_range.end - e.time_range.start)
        dev_ev.sort()
        busy = sum(e[1] - e[0] for e in dev_ev)
        # union of the intervals (kernels of different streams may overlap) and the idle time inside the profiled window
        union, cur_s, cur_e = 0.0, None, None
        for s, e, _, _ in dev_ev:
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    union += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        if cur_e is not None:
            union += cur_e - cur_s
        span = dev_ev[-1][1] - dev_ev[0][0] if dev_ev else 0.0
        by = {}
        for s, e, n, _ in dev_ev:
            k = n.split("(")[0][:70]
            v = by.setdefault(k, [0, 0.0])
            v[0] += 1
            v[1] += e - s
        hot = {k: {"n_per_step": round(v[0] / a.prof, 2), "us_per_step": round(v[1] / a.prof, 2)} for k, v in by.items()
               if "dprhot" in k or "nccl" in k.lower() or "rccl" in k.lower() or "copyBuffer" in k or "Memcpy" in k}
        out["runs"].append({"mode": mode, "wall_ms_per_step": round(wall_ms, 3), "profiled_steps": a.prof,
                            "device_busy_ms_per_step(sum of kernels)": round(busy / a.prof * 1e-3, 3),
                            "device_union_ms_per_step": round(union / a.prof * 1e-3, 3),
                            "device_idle_ms_per_step(inside the window)": round((span - union) / a.prof * 1e-3, 3),
                            "kernels_per_step": round(len(dev_ev) / a.prof, 1),
                            "hot_path_and_collective_launches": hot,
                            "host_us_per_step": {k: round(v / a.prof, 1) for k, v in sorted(host.items())}})
    by_mode = {}
    for r in out["runs"]:
        by_mode.setdefault(r["mode"], []).append(r)
    if a.no_prof:
        med = lambda rs: sorted(r["wall_ms_per_step"] for r in rs)[len(rs) // 2]  # noqa: E731
        out["median_wall_ms_per_step"] = {m: med(rs) for m, rs in by_mode.items()}
        tr = getattr(task, "_order_trial", None)
        out["tower_order_trial"] = None if not tr or tr.get("decided") is None else {"decided": tr["decided"], "ms_per_step": tr["ms"]}
        if "single" in by_mode and "forced_auto" in by_mode:
            out["forced_auto_minus_single_ms"] = round(med(by_mode["forced_auto"]) - med(by_mode["single"]), 3)
    elif "single" in by_mode and "forced" in by_mode:
        mean = lambda rs, key: sum(r[key] for r in rs) / len(rs)  # noqa: E731
        out["forced_minus_single"] = {
            "wall_ms": round(mean(by_mode["forced"], "wall_ms_per_step") - mean(by_mode["single"], "wall_ms_per_step"), 3),
            "device_busy_ms": round(mean(by_mode["forced"], "device_busy_ms_per_step(sum of kernels)") -
                                    mean(by_mode["single"], "device_busy_ms_per_step(sum of kernels)"), 3),
            "device_idle_ms": round(mean(by_mode["forced"], "device_idle_ms_per_step(inside the window)") -
                                    mean(by_mode["single"], "device_idle_ms_per_step(inside the window)"), 3),
            "kernels": round(mean(by_mode["forced"], "kernels_per_step") - mean(by_mode["single"], "kernels_per_step"), 1)}
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
