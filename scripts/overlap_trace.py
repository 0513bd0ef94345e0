#!/usr/bin/env python3
"""SURVEY.md section 8(e) / north_star "overlapped with the encoder backward on a side HIP stream", on the hardware there is: ONE
MI355X, a ONE-rank RCCL world (DPRHOT_FORCE_DIST=1 makes DenseRetrieverTask.training_step take its multi-GPU branch, dpr_task.py:163-195's
replacement), two random-init bert-base towers, seq_len 256, B = 32, K = 8, bf16 autocast.  Run under rocprofv3:

    rocprofv3 --kernel-trace --memory-copy-trace -d <dir> -o ov -- python scripts/overlap_trace.py [--direct]
    python scripts/overlap_check.py <dir>/.../ov_results.db          # asserts the intervals, writes the summary

and without a profiler it prints the step time (the `end_to_end_forced_dist` figure of bench.py).  --direct: the collectives through
the C ABI communicator on the side HIP stream (DPRHOT_DIRECT_RCCL=1) instead of torch.distributed's RCCL stream."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--direct", action="store_true")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--K", type=int, default=8)
    ap.add_argument("--seq", type=int, default=256)
    ap.add_argument("--large", action="store_true", help="bert-large towers (BASELINE configs[4])")
    ap.add_argument("--order", default="context_first", choices=["context_first", "reference", "auto"],
                    help="DPRHOT_TOWER_ORDER: context_first (default here: the overlap timeline), reference, or auto -- the product's default, "
                         "which times both orders over its first 14 steps (use --warmup >= 15)")
    ap.add_argument("--pad-mb", type=int, default=0, help="DPRHOT_DEBUG_PAD_MB: every collective of the path is followed by the same collective on N MiB "
                                                           "per rank, so that it takes time on a one-rank world (timelines only)")
    a = ap.parse_args()
    os.environ["DPRHOT_FORCE_DIST"] = "1"
    if a.pad_mb > 0:
        os.environ["DPRHOT_DEBUG_PAD_MB"] = str(a.pad_mb)
    os.environ["DPRHOT_TOWER_ORDER"] = a.order
    os.environ.setdefault("DPRHOT_DC_WIRE", "bf16")  # (the widen launch behind the wait is a named landmark on the timeline)
    os.environ["DPRHOT_DIRECT_RCCL"] = "1" if a.direct else "0"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29755")
    import torch
    import torch.distributed as dist

    from bench_e2e import BERT_BASE, _batch
    from dpr_scale_amd import lightning_compat
    from dpr_scale_amd.hydra_compat import Conf
    from dpr_scale_amd.task.dpr_task import DenseRetrieverTask

    import dpr_scale_amd

    dpr_scale_amd.configure_runtime()  # before the first HIP call (kernel arguments on the device) and before the process group exists
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    cfg = dict(BERT_BASE)
    if a.large:
        cfg.update(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
    model_conf = Conf({"_target_": "dpr_scale_amd.models.hf_model.HFEncoder", "model_path": cfg, "dropout": 0.1})
    optim_conf = Conf({"_target_": "torch.optim.AdamW", "lr": 3e-5})
    torch.manual_seed(0)
    task = DenseRetrieverTask(None, model_conf, None, optim_conf, shared_model=False)
    task.trainer = lightning_compat.Trainer(device=dev, max_steps=1000)
    task.trainer.strategy = lightning_compat.DDPStrategy()  # (one rank: the marker the task's isinstance check looks for)
    task.setup("fit")
    task.to(dev).train()
    task.on_pretrain_routine_start()
    from dpr_scale_amd import dist as D

    batch = _batch(a.B, a.K, a.seq, dev)
    opt = torch.optim.AdamW(task.parameters(), lr=3e-5)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = task.training_step(batch, 0)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    print(json.dumps({"workload": f"DenseRetrieverTask.training_step, multi-GPU branch forced on a one-rank RCCL world, 2 x bert-{'large' if a.large else 'base'}, "
                                  f"seq_len={a.seq}, B={a.B}, K={a.K}, bf16 autocast, AdamW",
                      "collectives": "C ABI communicator on the side HIP stream" if D.direct_comm() is not None else "torch.distributed (RCCL's stream)",
                      "ms_per_step": round(ms, 3), "pairs_per_s": round(a.B / ms * 1e3, 1), "loss_last": round(float(loss.detach()), 4),
                      "steps": a.steps, "warmup": a.warmup, "tower_order": a.order,
                      "tower_order_trial": (task._order_trial or {}).get("decided") and {"decided": task._order_trial["decided"], "ms_per_step": task._order_trial["ms"]}}),
          flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
