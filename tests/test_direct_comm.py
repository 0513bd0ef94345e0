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
        # the product's transport: once enabled for the group, dist.all_gather_rows / reduce_scatter_rows / all_reduce_sum go through
        # the communicator -- synchronously on the caller's stream, or on the side stream behind an event (async_op=True), for
        # both wire formats of the backward
        assert D.direct_comm() is None
        # opt-in: without DPRHOT_DIRECT_RCCL=1 the task's call is a no-op and the path stays on torch.distributed
        os.environ.pop("DPRHOT_DIRECT_RCCL", None)
        assert D.enable_direct_comm(dev) is None and D.direct_comm() is None
        D.disable_direct_comm()
        os.environ["DPRHOT_DIRECT_RCCL"] = "1"
        assert D.enable_direct_comm(dev) is not None and D.direct_comm() is not None  # (set-up on the watchdog's helper thread)
        for dt in (torch.float32, torch.bfloat16):
            part = torch.randn(264, 768, device=dev).to(dt)
            mine = torch.full_like(part, 7.0)
            w = D.reduce_scatter_rows(part, mine, None, async_op=True)
            assert isinstance(w, D._StreamWork)
            w.wait()
            out = torch.empty_like(send)
            w2 = D.all_gather_rows(send, out, None, async_op=True)
            w2.wait()
            mine2 = torch.empty_like(part)
            assert D.reduce_scatter_rows(part, mine2, None) is None
            torch.cuda.synchronize()
            assert torch.equal(mine, part) and torch.equal(mine2, part) and torch.equal(out, send)
        s2 = torch.tensor([1.25], device=dev)
        assert D.all_reduce_sum(s2) is None
        torch.cuda.synchronize()
        assert s2.item() == 1.25
        D.disable_direct_comm()
        assert D.direct_comm() is None
    finally:
        dist.destroy_process_group()


def _operator_step(q0, c0, y, m, T, scale=3.0):
    from dpr_scale_amd.hotpath import ContextGather, defer_context_grad, inbatch_contrastive_loss

    tq, tc = q0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
    c1 = tc * 1.0
    c2, pending = defer_context_grad(c1)
    g = ContextGather(c2, m, None)
    q1 = tq * 1.0
    loss = inbatch_contrastive_loss(q1, c2, y, m, T, None, None, g, pending)
    (loss * scale).backward()
    torch.cuda.synchronize()
    return loss.detach().clone(), tq.grad.clone(), tc.grad.clone()


def test_multi_rank_operator_host_side_in_cpp_equals_the_python_one_and_all_pairs_equals_rccl():
    """The multi-GPU branch of the operator on a one-rank RCCL world (DPRHOT_FORCE_DIST=1: packed layout, ContextGather, the packed
    step, asynchronous reduce-scatter, deferred context gradient): (a) the step's host side in C++ (csrc/opx.cpp: packed_train_step /
    packed_backward, round 5) against the Python wrappers -- bit-identical loss and gradients, both wire formats, a shape whose plan
    has a bf16 dC epilogue (128 x 1032) and one that falls back to fp32 partials + one cast launch (32 x 264); (b) the all-pairs form
    of the two collectives, through torch.distributed (grouped all_to_all on nccl) and through the C ABI communicator
    (dprhot_allgather_allpairs / dprhot_reducescatter_allpairs), against RCCL's own collectives."""
    import numpy as np
    import torch.distributed as dist

    from dpr_scale_amd import dist as D
    from dpr_scale_amd import hotpath
    from oracle import inbatch_oracle as O

    if not hotpath._OPX_PACKED:
        pytest.skip("_opx.so not built")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29743"
    os.environ["DPRHOT_FORCE_DIST"] = "1"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        for B, K, d in ((128, 8, 768), (32, 8, 768)):
            qv, cv, y, m = O.synth_embeddings(700 + B, B, K, d, "U", True)
            ref = O.training_step_global(qv, cv, y, m, 0.5)
            q0, c0 = torch.from_numpy(qv).to(dev), torch.from_numpy(cv).to(dev)
            ty, tm = torch.from_numpy(y).to(dev), torch.from_numpy(m).to(dev)
            for wire in ("fp32", "bf16"):
                os.environ["DPRHOT_DC_WIRE"] = wire
                res = {}
                for packed in (True, False, True):
                    hotpath._OPX_PACKED = packed
                    try:
                        res.setdefault(packed, []).append(_operator_step(q0, c0, ty, tm, 0.5))
                    finally:
                        hotpath._OPX_PACKED = True
                for a, b in ((res[True][0], res[False][0]), (res[True][1], res[False][0])):
                    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), (B, wire)
                loss, dq, dc = res[True][0]
                assert abs(loss.item() - ref["loss"]) <= 1e-3 * max(1.0, abs(ref["loss"]))
                assert np.abs(dq.cpu().numpy() / 3.0 - ref["dQ"]).max() <= 1e-2 * np.abs(ref["dQ"]).max()
                assert np.abs(dc.cpu().numpy() / 3.0 - ref["dC"]).max() <= 1e-2 * np.abs(ref["dC"]).max()
                # (b) forms and transports of the collectives: same numbers (one rank: every exchange is the self pair)
                os.environ["DPRHOT_DIRECT_RCCL"] = "1"
                assert D.enable_direct_comm(dev) is not None
                base = None
                for direct in (False, True):
                    for form in ("rccl", "allpairs"):
                        D.configure(topology=form, direct=direct)
                        try:
                            out = _operator_step(q0, c0, ty, tm, 0.5)
                        finally:
                            D.configure(None, None)
                        if base is None:
                            base = out
                        assert torch.equal(out[0], base[0]) and torch.equal(out[1], base[1]), (B, wire, direct, form)
                        assert (out[2] - base[2]).abs().max().item() <= 1e-6 * base[2].abs().max().item(), (B, wire, direct, form)
                D.disable_direct_comm()
    finally:
        os.environ.pop("DPRHOT_FORCE_DIST", None)
        os.environ.pop("DPRHOT_DC_WIRE", None)
        os.environ.pop("DPRHOT_DIRECT_RCCL", None)
        dist.destroy_process_group()


def test_tower_order_of_the_multi_gpu_step_is_measured_and_settles():
    """DenseRetrieverTask._tower_order.  Default: a FIXED order (context_first: the collectives under the query tower), no trial -- a
    seeded run is reproducible.  DPRHOT_TOWER_ORDER=auto (opt-in): over the first 14 steps of the multi-GPU branch the two
    orders of the towers alternate, timed with HIP events; then the step settles on one of them and stays there; with
    accumulate_grad_batches = 2 the orders alternate per accumulation cycle; a run that asked for reproducibility refuses the trial;
    the decision rides with the checkpoint."""
    import torch.distributed as dist

    from dpr_scale_amd import lightning_compat
    from dpr_scale_amd.hydra_compat import Conf
    from dpr_scale_amd.task.dpr_task import DenseRetrieverTask

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29745"
    os.environ["DPRHOT_FORCE_DIST"] = "1"
    os.environ.pop("DPRHOT_TOWER_ORDER", None)
    os.environ.pop("PL_GLOBAL_SEED", None)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        cfg = {"vocab_size": 1000, "hidden_size": 128, "num_hidden_layers": 2, "num_attention_heads": 2, "intermediate_size": 256,
               "max_position_embeddings": 64}
        model_conf = Conf({"_target_": "dpr_scale_amd.models.hf_model.HFEncoder", "model_path": cfg, "dropout": 0.0})
        task = DenseRetrieverTask(None, model_conf, None, Conf({"_target_": "torch.optim.AdamW", "lr": 1e-4}), shared_model=False)
        task.trainer = lightning_compat.Trainer(device=dev, max_steps=100)
        task.trainer.strategy = lightning_compat.DDPStrategy()
        task.setup("fit")
        task.to(dev).train()
        g = torch.Generator().manual_seed(0)
        B, K, T = 8, 4, 32

        def tok(rows):
            ids = torch.randint(5, 1000, (rows, T), generator=g).to(dev)
            return {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": torch.ones_like(ids)}

        batch = {"query_ids": tok(B), "contexts_ids": tok(B * K), "pos_ctx_indices": (torch.arange(B) * K).to(dev),
                 "ctx_mask": torch.zeros(B * K, dtype=torch.bool, device=dev)}
        # default: fixed order, no trial state, nothing timed
        assert [task._tower_order() for _ in range(4)] == ["context_first"] * 4 and task._order_trial is None
        os.environ["DPRHOT_TOWER_ORDER"] = "auto"
        os.environ["PL_GLOBAL_SEED"] = "7"  # seed_everything was called: the trial is refused
        assert task._tower_order() == "context_first" and task._order_trial is None
        os.environ.pop("PL_GLOBAL_SEED")
        orders, losses = [], []
        real = task._tower_order

        def spy():
            o = real()
            orders.append(o)
            return o

        task._tower_order = spy
        for _ in range(20):
            loss = task.training_step(batch, 0)
            loss.backward()
            task.zero_grad(set_to_none=True)
            losses.append(loss.item())
        tr = task._order_trial
        assert tr["decided"] in ("context_first", "reference") and set(tr["ms"]) == {"context_first", "reference"}
        assert orders[:3] == ["context_first"] * 3 and orders[3:14] == ["context_first", "reference"] * 5 + ["context_first"]
        assert orders[14:] == [tr["decided"]] * 6
        assert all(abs(x - losses[0]) <= 1e-4 * max(1.0, abs(losses[0])) for x in losses)  # (dropout 0: the order changes nothing else)
        ck = {}
        task.on_save_checkpoint(ck)
        assert ck["dprhot_runtime"]["tower_order"] == tr["decided"] and ck["dprhot_runtime"]["tower_order_trial_ms"] == tr["ms"]
        # accumulate_grad_batches = 2: whole cycles of warm-up (4 steps), then the orders alternate cycle by cycle
        task._order_trial = None
        task.trainer.accumulate_grad_batches = 2
        seq = []

        def spy2():
            o = real()
            seq.append(o)
            return o

        task._tower_order = spy2
        for _ in range(28):
            task.training_step(batch, 0).backward()
            task.zero_grad(set_to_none=True)
        assert seq[:4] == ["context_first"] * 4 and seq[4:24] == (["context_first"] * 2 + ["reference"] * 2) * 5 and seq[24] == "context_first"
        assert seq[25:] == [task._order_trial["decided"]] * 3
    finally:
        os.environ.pop("DPRHOT_FORCE_DIST", None)
        os.environ.pop("DPRHOT_TOWER_ORDER", None)
        dist.destroy_process_group()
