"""Drop-in boundary: DenseRetrieverTask mirrors the reference class -- constructor kwargs,
hooks, metric names, checkpoint layout, Hydra config schema -- and its training_step reproduces the REFERENCE
training_step on the same encoders and batch (reference run live when /root/reference is mounted).

CPU tests inject the numpy stand-in for the HIP kernels (the product default refuses CPU tensors); the `gpu`
tests run BASELINE.json configs[0] (tiny BERT, batch 4, 1 pos + 1 hard negative) through libdprhot.so.
"""
import inspect
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests"))

from dpr_scale_amd import hydra_compat, lightning_compat  # noqa: E402
from dpr_scale_amd.task.dpr_task import DenseRetrieverTask  # noqa: E402

CONF = os.path.join(ROOT, "dpr_scale_amd", "conf")
REF = "/root/reference"


def make_task(cfg, kernels):
    task = hydra_compat.instantiate(cfg.task, _recursive_=False)  # main.py:25
    task.kernels = kernels
    task.loss.kernels = kernels
    return task


def run_fit(device, kernels, tmp_path, overrides=()):
    cfg = hydra_compat.compose(CONF, "tiny_cpu.yaml", list(overrides))
    torch.manual_seed(0)
    task = make_task(cfg, kernels)
    dm = hydra_compat.instantiate(cfg.datamodule)
    ckpt = hydra_compat.instantiate(cfg.checkpoint_callback, dirpath=str(tmp_path))
    trainer = lightning_compat.Trainer(max_epochs=cfg.trainer.max_epochs, max_steps=cfg.trainer.max_steps,
                                       gradient_clip_val=cfg.trainer.gradient_clip_val, callbacks=[ckpt], device=device)
    trainer.fit(task, dm)
    return task, trainer, ckpt, dm


def test_constructor_signature_matches_reference():
    want = ["transform", "model", "datamodule", "optim", "k", "shared_model", "in_batch_eval", "in_batch_negatives",
            "warmup_steps", "fp16_grads", "pretrained_checkpoint_path", "softmax_temperature"]
    sig = inspect.signature(DenseRetrieverTask.__init__).parameters
    assert [p for p in sig if p != "self"] == want
    assert sig["k"].default == 1 and sig["shared_model"].default is True and sig["softmax_temperature"].default == 1.0
    for m in ["setup", "on_load_checkpoint", "on_pretrain_routine_start", "forward", "training_step", "validation_step",
              "validation_epoch_end", "test_step", "test_epoch_end", "configure_optimizers", "to_torchscript", "sim_score",
              "encode_queries", "encode_contexts", "_encode_sequence", "compute_rank_metrics", "_eval_step", "_eval_epoch_end"]:
        assert callable(getattr(DenseRetrieverTask, m)), m


def test_config_tree_composes_like_hydra():
    cfg = hydra_compat.compose(CONF, "tiny_cpu.yaml", ["task.softmax_temperature=0.5", "datamodule.batch_size=8"])
    assert cfg.task._target_.endswith("task.dpr_task.DenseRetrieverTask")
    assert cfg.task.softmax_temperature == 0.5 and cfg.datamodule.batch_size == 8 and cfg.trainer.max_steps == 3
    assert cfg.task.optim._target_ == "torch.optim.AdamW" and cfg.checkpoint_callback.monitor == "valid_mrr"


def test_config_groups_defaults_overrides_interpolation(tmp_path):
    """The Hydra mechanics the reference's recipes rely on, on a tree generated here: defaults list, `override`,
    `# @package _group_` / `_global_`, `${}` interpolation, scientific-notation floats, k=v and group overrides."""
    def w(rel, text):
        f = tmp_path / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text(text)

    w("config.yaml", "defaults:\n  - _self_\n  - grp: a\n  - grp/sub: x\n  - other: one\nflag: false\n")
    w("grp/a.yaml", "# @package _group_\nname: a\nlr: 1e-3\n")
    w("grp/b.yaml", "# @package _group_\nname: b\nlr: 3e-5\n")
    w("grp/sub/x.yaml", "# @package _group_\npath: base\n")
    w("other/one.yaml", "# @package _group_\nref: ${grp.sub.path}\nmsg: at-${grp.name}\n")
    w("other/two.yaml", "# @package _global_\nflag: true\nother:\n  ref: global\n")
    w("recipe.yaml", "defaults:\n  - config\n  - override grp: b\ngrp:\n  extra: 7\n")
    base = hydra_compat.compose(str(tmp_path), "config")
    assert (base.grp.name, base.grp.lr, base.grp.sub.path, base.other.ref, base.other.msg, base.flag) == ("a", 1e-3, "base", "base", "at-a", False)
    rec = hydra_compat.compose(str(tmp_path), "recipe.yaml", ["grp.sub.path=changed", "+grp.new=2", "other=two"])
    assert (rec.grp.name, rec.grp.lr, rec.grp.extra, rec.grp.new, rec.grp.sub.path) == ("b", 3e-5, 7, 2, "changed")
    assert rec.flag is True and rec.other.ref == "global"
    obj = hydra_compat.instantiate({"_target_": "collections.OrderedDict", "a": 1})
    assert dict(obj) == {"a": 1}


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")
def test_reference_conf_tree_loads_unchanged():
    cfg = hydra_compat.compose(os.path.join(REF, "dpr_scale", "conf"), "msmarco_baseline.yaml")
    assert cfg.task._target_ == "dpr_scale.task.dpr_task.DenseRetrieverTask"
    assert cfg.task.transform.model_path == cfg.task.model.model_path == "bert-base-uncased"  # ${} interpolation
    assert (cfg.datamodule.batch_size, cfg.datamodule.num_negative, cfg.trainer.strategy) == (8, 7, "ddp")
    assert cfg.task.optim.lr == 3e-5 and cfg.task.shared_model is False


def test_cfg1_plumbing_run_cpu(tmp_path):
    """configs[0] end to end on CPU (stand-in kernels): 3 steps, eval, metric names, checkpoint layout, reload."""
    from _oracle_kernels import OracleKernels

    task, trainer, ckpt, dm = run_fit("cpu", OracleKernels(), tmp_path)
    assert trainer.global_step == 3 and all(np.isfinite(trainer.train_losses))
    for name in ["train_loss", "valid_avg_rank", "valid_mrr", "valid_accuracy@1", "valid_ctx_count", "valid_loss"]:
        assert name in task.logged, name
    ck = torch.load(os.path.join(tmp_path, "last.ckpt"), map_location="cpu")
    assert {"state_dict", "hyper_parameters", "epoch", "global_step", "optimizer_states", "lr_schedulers", "callbacks",
            "pytorch-lightning_version"} <= set(ck)
    keys = list(ck["state_dict"])
    assert any(k.startswith("query_encoder.transformer.") for k in keys)
    assert any(k.startswith("context_encoder.transformer.") for k in keys)
    assert ck["hyper_parameters"]["softmax_temperature"] == 1.0 and "model" in ck["hyper_parameters"]
    assert os.path.isfile(os.path.join(tmp_path, "checkpoint_best.ckpt"))
    again = DenseRetrieverTask.load_from_checkpoint(os.path.join(tmp_path, "last.ckpt"))
    for (k1, v1), (k2, v2) in zip(task.state_dict().items(), again.state_dict().items()):
        assert k1 == k2 and torch.equal(v1.cpu(), v2)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")
@pytest.mark.parametrize("in_batch", [True, False])
def test_training_step_equals_reference_training_step(in_batch):
    """Same encoders, same batch: loss and every parameter gradient vs the reference class, run verbatim."""
    from _oracle_kernels import OracleKernels
    from oracle import ref_shim

    cfg = hydra_compat.compose(CONF, "tiny_cpu.yaml", ["datamodule.ragged=true", "task.softmax_temperature=16.0"])
    torch.manual_seed(1)
    mine = make_task(cfg, OracleKernels())
    mine.in_batch_negatives = in_batch
    mine.trainer = lightning_compat.Trainer(device="cpu")
    mine.setup("fit")
    batch = hydra_compat.instantiate(cfg.datamodule).train_dataloader()[0]
    ref = ref_shim.make_reference_task(False, temperature=16.0, in_batch_negatives=in_batch)
    ref.query_encoder, ref.context_encoder = mine.query_encoder, mine.context_encoder  # the very same modules
    mine.eval(), ref.eval()  # (T=16 keeps the softmax of unit-variance LayerNorm outputs out of saturation)
    l_ref = ref.training_step(batch, 0)
    l_ref.backward()
    g_ref = {n: p.grad.clone() for n, p in mine.named_parameters() if p.grad is not None}
    mine.zero_grad()
    l_mine = mine.training_step(batch, 0)
    l_mine.backward()
    assert abs(l_mine.item() - l_ref.item()) <= 1e-3 * max(1.0, abs(l_ref.item()))
    checked = 0
    gmax = max(float(g.abs().max()) for g in g_ref.values())
    for n, p in mine.named_parameters():
        # (attention key biases etc. have analytically zero gradient: only fp noise there, skipped)
        if n in g_ref and g_ref[n].abs().max() > 1e-3 * gmax:
            err = (p.grad - g_ref[n]).abs().max() / g_ref[n].abs().max()
            assert err <= 3e-2, (n, float(err))  # bf16 embeddings + bf16 dScores vs the reference's fp32
            checked += 1
    assert checked > 20


def test_eval_metrics_equal_reference_formulas():
    """compute_rank_metrics / loss on fixed logits vs the fixture the reference produced."""
    from _oracle_kernels import OracleKernels
    from conftest import load_golden

    meta, g = load_golden("cfg2_Ur_T0.05")
    task = DenseRetrieverTask(None, None, None, None, k=1)
    task.kernels = OracleKernels()
    labels = torch.arange(meta["B"]) * meta["K"]
    rank, mrr, score = task.compute_rank_metrics(torch.from_numpy(g["S"]), labels)
    assert (rank, score) == (int(g["rank_metrics"][0]), int(g["rank_metrics"][2])) and abs(mrr - g["rank_metrics"][1]) < 1e-9


@pytest.mark.gpu
def test_cfg1_plumbing_run_gpu(tmp_path):
    """configs[0] through libdprhot.so on the MI355X; same step-0 loss as the CPU stand-in run."""
    task, trainer, ckpt, dm = run_fit("cuda:0", None, tmp_path)
    assert trainer.global_step == 3 and all(np.isfinite(trainer.train_losses))
    assert "valid_mrr" in task.logged and 0 < float(task.logged["valid_mrr"]) <= 1
    from _oracle_kernels import OracleKernels

    _, t2, _, _ = run_fit("cpu", OracleKernels(), tmp_path / "cpu")
    assert abs(trainer.train_losses[0] - t2.train_losses[0]) <= 2e-3 * max(1.0, abs(t2.train_losses[0]))


def test_fp16_grads_registers_the_xgmi_gradient_hook():
    """dpr_task.py:90-92: fp16_grads -> a DDP communication hook on trainer.strategy._model; here the all-pairs-exchange hook
    of dpr_scale_amd.comm_hooks, configured by DPRHOT_GRAD_* like the reference's is by nothing."""
    from types import SimpleNamespace

    from dpr_scale_amd import comm_hooks
    from dpr_scale_amd.task.dpr_task import DenseRetrieverTask

    calls = []
    model = SimpleNamespace(register_comm_hook=lambda state, hook: calls.append((state, hook)))
    task = DenseRetrieverTask(None, None, None, None, fp16_grads=True)
    task.trainer = SimpleNamespace(strategy=SimpleNamespace(_model=model))
    task.on_pretrain_routine_start()
    assert len(calls) == 1 and calls[0][1] is comm_hooks.compressed_allreduce_hook
    st = calls[0][0]
    assert isinstance(st, comm_hooks.GradCommState) and st.mode == "direct" and str(st.wire_dtype) == "torch.float16"  # the reference's wire format
    task2 = DenseRetrieverTask(None, None, None, None, fp16_grads=False)
    task2.trainer = task.trainer
    task2.on_pretrain_routine_start()
    assert len(calls) == 1


def test_sim_score_then_loss_is_differentiable_like_the_reference():
    """dpr_task.py:98-105 is a differentiable torch.matmul and subclasses train through `self.loss(self.sim_score(...))`
    (citadel_task.py:249-262): gradients must reach the encoder outputs (they silently did not)."""
    import numpy as np

    from _oracle_kernels import OracleKernels
    from dpr_scale_amd.task.dpr_task import DenseRetrieverTask
    from oracle import inbatch_oracle as O

    q, c, y, m = O.synth_embeddings(5, 6, 3, 64, "U", True)
    task = DenseRetrieverTask(None, None, None, None)
    task.kernels = OracleKernels()
    task.loss.kernels = task.kernels
    tq, tc = torch.from_numpy(q).requires_grad_(True), torch.from_numpy(c).requires_grad_(True)
    scores = task.sim_score(tq, tc, torch.from_numpy(m))
    assert scores.requires_grad
    loss = task.loss(scores, torch.from_numpy(y))
    loss.backward()
    rq, rc = torch.from_numpy(q).requires_grad_(True), torch.from_numpy(c).requires_grad_(True)
    S = torch.matmul(rq, rc.T)
    S = S.masked_fill(torch.from_numpy(m)[None, :].expand_as(S), float("-inf"))
    ref = torch.nn.CrossEntropyLoss()(S, torch.from_numpy(y))
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    assert tq.grad is not None and tc.grad is not None
    assert np.abs(tq.grad.numpy() - rq.grad.numpy()).max() <= 1e-2 * np.abs(rq.grad.numpy()).max()
    assert np.abs(tc.grad.numpy() - rc.grad.numpy()).max() <= 1e-2 * np.abs(rc.grad.numpy()).max()
    # the [Nq, Nc] mask form the reference passes (:197) works too
    s2 = task.sim_score(tq.detach(), tc.detach(), torch.from_numpy(m).repeat(6, 1))
    assert torch.equal(torch.isinf(s2), torch.from_numpy(m).repeat(6, 1))


def test_to_torchscript_packages_the_encoders_without_the_reference_package(tmp_path):
    """dpr_task.py:325-368 (a SURVEY.md 8-b1 hook): scripted modules texts -> embeddings for the context encoder (plain and int8
    dynamic-quantised) and, with separate towers, the query encoder; the context encoder is saved to file_path."""
    import torch

    from dpr_scale_amd.task.dpr_task import DenseRetrieverTask

    task = DenseRetrieverTask({"_target_": "_script_toys.ToyTransform"}, {"_target_": "_script_toys.ToyEncoder"}, None, None,
                              shared_model=False)
    task.setup("fit")
    path = str(tmp_path / "ctx.pt")
    out = task.to_torchscript(path)
    assert sorted(out) == ["ctx_encoder", "ctx_encoder_qt", "q_encoder", "q_encoder_qt"]
    texts = ["hello", "hi", ""]
    want = task.context_encoder(out["ctx_encoder"].transform(texts)["token_ids"])
    assert torch.allclose(out["ctx_encoder"](texts), want, atol=1e-6)
    assert torch.allclose(torch.jit.load(path)(texts), want, atol=1e-6)
    assert (out["ctx_encoder_qt"](texts) - want).abs().max() < 0.1  # int8 weights
    assert not torch.allclose(out["q_encoder"](texts), want)        # separate towers
    import pytest

    with pytest.raises(ValueError):
        task.to_torchscript(method="trace")
