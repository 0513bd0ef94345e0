"""Extra information for bench.py (`end_to_end` block): the END-TO-END training step of BASELINE.json configs[1] -- two random-init
bert-base towers (PyTorch-ROCm, bf16 autocast), seq_len 256, batch 32, 1 positive + 7 hard negatives, AdamW -- once
with the hand-written hot path (dpr_scale_amd DenseRetrieverTask) and once with the reference's formulation of
the same step written in plain torch ops (matmul, masked_fill, /T, cross_entropy; dpr_task.py:197-212).
The towers are >99.9 % of the FLOPs, so this number is encoder-bound by construction; it is
reported next to, never instead of, the hot-path metric.
"""
import time

import torch

BERT_BASE = {"vocab_size": 30522, "hidden_size": 768, "num_hidden_layers": 12, "num_attention_heads": 12,
             "intermediate_size": 3072, "max_position_embeddings": 512}


def _batch(B, K, T, dev, seed=0):
    g = torch.Generator().manual_seed(seed)

    def tok(rows):
        ids = torch.randint(1000, 30522, (rows, T), generator=g)
        ids[:, 0], ids[:, -1] = 101, 102
        return {"input_ids": ids.to(dev), "token_type_ids": torch.zeros_like(ids).to(dev),
                "attention_mask": torch.ones_like(ids).to(dev)}

    return {"query_ids": tok(B), "contexts_ids": tok(B * K), "pos_ctx_indices": (torch.arange(B) * K).to(dev),
            "ctx_mask": torch.zeros(B * K, dtype=torch.bool, device=dev)}


def end_to_end(B, K, d, T, dev, seq_len=256, steps=5, warmup=3, world=1, rank=0):
    """world == 1: our operator vs the reference's torch ops behind the same two towers.
    world > 1 (one process per GPU, torch.distributed initialised): the towers under DistributedDataParallel with the
    all-pairs gradient hook (dpr_scale_amd.comm_hooks), the context all-gather started under the query tower and the
    reduce-scatter of dC under the query-tower backward -- what DenseRetrieverTask.training_step does under DDP."""
    from dpr_scale_amd import lightning_compat
    from dpr_scale_amd.hydra_compat import Conf
    from dpr_scale_amd.task.dpr_task import DenseRetrieverTask

    assert d == 768, "end-to-end leg is defined for bert-base (d=768)"
    model_conf = Conf({"_target_": "dpr_scale_amd.models.hf_model.HFEncoder", "model_path": dict(BERT_BASE), "dropout": 0.1})
    optim_conf = Conf({"_target_": "torch.optim.AdamW", "lr": 3e-5})
    torch.manual_seed(0)
    task = DenseRetrieverTask(None, model_conf, None, optim_conf, shared_model=False, softmax_temperature=T)
    task.trainer = lightning_compat.Trainer(device=dev, max_steps=1000)
    task.setup("fit")
    task.to(dev).train()
    batch = _batch(B, K, seq_len, dev, seed=rank)
    from dpr_scale_amd.hotpath import ContextGather, defer_context_grad, inbatch_contrastive_loss

    q_enc, c_enc, hook_state = task.query_encoder, task.context_encoder, None
    if world > 1:
        from dpr_scale_amd import comm_hooks

        ids = [dev.index] if torch.distributed.get_backend() == "nccl" else None
        q_enc, hook_state = comm_hooks.wrap_ddp(task.query_encoder, device_ids=ids)
        c_enc, _ = comm_hooks.wrap_ddp(task.context_encoder, device_ids=ids, state=hook_state)
    opt = torch.optim.AdamW(task.parameters(), lr=3e-5)

    def ours():
        if world > 1:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                c = c_enc(batch["contexts_ids"]).float()
            c, pending = defer_context_grad(c)
            gather = ContextGather(c, batch["ctx_mask"], None)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                q = q_enc(batch["query_ids"]).float()
            return inbatch_contrastive_loss(q, c, batch["pos_ctx_indices"], batch["ctx_mask"], T, None, None, gather, pending)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            q, c = task(batch["query_ids"], batch["contexts_ids"])
        return inbatch_contrastive_loss(q.float(), c.float(), batch["pos_ctx_indices"], batch["ctx_mask"], T, False)

    def torch_ref():  # the reference's ops (dpr_task.py:197-212), single rank
        with torch.autocast("cuda", dtype=torch.bfloat16):
            q, c = task(batch["query_ids"], batch["contexts_ids"])
            scores = torch.matmul(q, c.transpose(0, 1))
            scores = scores.masked_fill(batch["ctx_mask"].repeat(q.shape[0], 1), float("-inf")) / T
        return torch.nn.functional.cross_entropy(scores.float(), batch["pos_ctx_indices"])

    def run(fn):
        def step():
            loss = fn()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            return loss
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        return (time.perf_counter() - t0) / steps, float(loss.detach())

    t_ours, l_ours = run(ours)
    out = {"workload": f"2 x bert-base (random init), seq_len={seq_len}, B={B} per GPU, K={K}, bf16 autocast, AdamW, {world} GPU(s)",
           "ms_per_step": round(t_ours * 1e3, 3), "pairs_per_s": round(world * B / t_ours, 1), "loss_last": round(l_ours, 4),
           "steps": steps, "warmup": warmup, "global_negatives_per_query": world * B * K - 1}
    if world == 1:
        t_ref, l_ref = run(torch_ref)
        out["torch_reference_hot_path_ms_per_step"] = round(t_ref * 1e3, 3)
        out["torch_reference_pairs_per_s"] = round(B / t_ref, 1)
    else:
        out["gradient_hook"] = {"mode": hook_state.mode, "wire_dtype": str(hook_state.wire_dtype), "buckets_per_step": hook_state.buckets // (steps + warmup),
                                "wire_MB_per_step_per_rank": round(hook_state.wire_bytes / (steps + warmup) / 1e6, 1)}
    return out
