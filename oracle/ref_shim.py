"""TEST INFRASTRUCTURE ONLY -- runs the *reference's own* hot-path code, unmodified, on CPU.

This module exists only in the build container (it needs ``/root/reference``); it does not travel to
the GPU box and nothing in the product path may import it.  It is used by ``oracle/make_golden.py``
to generate the fixtures under ``tests/golden/`` and by ``tests/test_oracle_vs_reference.py`` (skipped
when ``/root/reference`` is absent) to pin ``oracle/inbatch_oracle.py`` against the reference.

What it does: ``/root/reference/dpr_scale/task/dpr_task.py`` imports ``hydra``, ``pytorch_lightning`` and
``pytorch_lightning.strategies`` at module top (dpr_task.py:4,8,9); none is installed here.  Three stub
modules are placed in ``sys.modules`` so that ``DenseRetrieverTask`` imports *from the reference tree*
and ``training_step`` (dpr_task.py:153-214), ``sim_score`` (:98-105) and ``compute_rank_metrics``
(:235-246) execute verbatim.  The ``all_gather`` stub restates pytorch-lightning==1.6.4
(requirements.txt:4) ``LightningModule.all_gather``: one ``torch.distributed.all_gather`` per tensor of
the tuple, stacked to ``[W, ...]``, under ``no_grad`` (sync_grads=False).
"""
import os
import sys
import types
from types import SimpleNamespace

import torch
import torch.distributed as dist

REFERENCE_ROOT = os.environ.get("DPR_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "dpr_scale", "task", "dpr_task.py"))


def _install_stubs():
    if "pytorch_lightning" in sys.modules and getattr(sys.modules["pytorch_lightning"], "_dprhot_stub", False):
        return
    hydra = types.ModuleType("hydra")
    hydra.utils = types.ModuleType("hydra.utils")

    def _instantiate(*a, **k):  # never reached on the hot path (encoders are bypassed)
        raise RuntimeError("hydra.utils.instantiate stub called")

    hydra.utils.instantiate = _instantiate
    hydra.main = lambda *a, **k: (lambda f: f)
    sys.modules["hydra"] = hydra
    sys.modules["hydra.utils"] = hydra.utils

    pl = types.ModuleType("pytorch_lightning")
    pl._dprhot_stub = True

    class LightningModule(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.trainer = None
            self.global_rank = 0
            self.logged = {}

        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, name, value, **kw):
            self.logged[name] = value

        def log_dict(self, d, **kw):
            self.logged.update(d)

        def all_gather(self, data, group=None, sync_grads=False):
            world = dist.get_world_size()

            def one(t):
                with torch.no_grad():
                    if not isinstance(t, torch.Tensor):  # PL 1.6.4 convert_to_tensors: python numbers become tensors
                        t = torch.tensor(t)             # (citadel_task.py:87 gathers a python int)
                    t = t.contiguous()
                    out = [torch.zeros_like(t) for _ in range(world)]
                    dist.all_gather(out, t)
                    return torch.stack(out, 0)

            if isinstance(data, (tuple, list)):
                return type(data)(one(t) for t in data)
            return one(data)

    pl.LightningModule = LightningModule
    strategies = types.ModuleType("pytorch_lightning.strategies")

    class DDPStrategy:  # marker classes for the isinstance check at dpr_task.py:165
        pass

    class DDPShardedStrategy:
        pass

    strategies.DDPStrategy = DDPStrategy
    strategies.DDPShardedStrategy = DDPShardedStrategy
    pl.strategies = strategies
    sys.modules["pytorch_lightning"] = pl
    sys.modules["pytorch_lightning.strategies"] = strategies


def load_reference_task_class():
    """Import DenseRetrieverTask from the reference tree, unmodified."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from dpr_scale.task.dpr_task import DenseRetrieverTask  # noqa: E402  (the reference's file)

    assert os.path.realpath(sys.modules["dpr_scale.task.dpr_task"].__file__).startswith(
        os.path.realpath(REFERENCE_ROOT)
    ), "dpr_scale resolved to something other than the reference tree"
    return DenseRetrieverTask


def make_reference_task(distributed: bool, rank: int = 0, temperature: float = 1.0, k: int = 1,
                        in_batch_negatives: bool = True):
    cls = load_reference_task_class()
    from pytorch_lightning.strategies import DDPStrategy

    task = cls(transform=None, model=None, datamodule=None, optim=None, k=k,
               softmax_temperature=temperature, in_batch_negatives=in_batch_negatives)
    task.trainer = SimpleNamespace(strategy=DDPStrategy() if distributed else object())
    task.global_rank = rank
    return task


def reference_training_step(q_local, c_local, pos_idx, ctx_mask, temperature=1.0, distributed=False, rank=0,
                            in_batch_negatives=True):
    """Run the reference ``training_step`` + ``loss.backward()`` with the encoders bypassed.

    Returns (loss, dq_local, dc_local) as the reference computes them on this rank.
    """
    task = make_reference_task(distributed, rank, temperature, in_batch_negatives=in_batch_negatives)
    q = q_local.clone().requires_grad_(True)
    c = c_local.clone().requires_grad_(True)
    task.forward = lambda query_ids, contexts_ids: (q, c)
    batch = {"query_ids": None, "contexts_ids": None,
             "pos_ctx_indices": pos_idx.clone(), "ctx_mask": ctx_mask.clone()}
    loss = task.training_step(batch, 0)
    loss.backward()
    return loss.detach(), q.grad.detach(), c.grad.detach()


def reference_sim_score(Q, C, mask_cols=None):
    task = make_reference_task(False)
    m = None if mask_cols is None else mask_cols.repeat(Q.shape[0], 1)  # dpr_task.py:197
    return task.sim_score(Q, C, m)


def reference_rank_metrics(S, labels, k=1):
    task = make_reference_task(False, k=k)
    return task.compute_rank_metrics(S, labels)


# ---- the CITADEL router loss, from the reference's own citadel_task.py ------------------------
def load_reference_citadel_class():
    """Import MultiVecRetrieverTask from /root/reference/dpr_scale/task/citadel_task.py, unmodified (it imports only torch,
    the pytorch_lightning.strategies markers and dpr_task -- citadel_task.py:4-6 -- so the same three stubs suffice)."""
    load_reference_task_class()
    from dpr_scale.task.citadel_task import MultiVecRetrieverTask  # noqa: E402  (the reference's file)

    assert os.path.realpath(sys.modules["dpr_scale.task.citadel_task"].__file__).startswith(
        os.path.realpath(REFERENCE_ROOT)), "citadel_task resolved to something other than the reference tree"
    return MultiVecRetrieverTask


def make_reference_citadel_task(in_batch=True, teacher_coef=0.0, tau=1.0, distributed=False, rank=0):
    cls = load_reference_citadel_class()
    from pytorch_lightning.strategies import DDPStrategy

    task = cls(in_batch=in_batch, teacher_coef=teacher_coef, tau=tau, transform=None, model=None, datamodule=None, optim=None)
    task.trainer = SimpleNamespace(strategy=DDPStrategy() if distributed else object(), max_epochs=1)
    task.global_rank = rank
    return task


def reference_citadel_sim_score(q, c, mask, pairwise):
    """citadel_task.py:137-153, verbatim (note: the pairwise branch dereferences mask before its None check, :141)."""
    return make_reference_citadel_task().sim_score(q, c, mask, pairwise=pairwise)


def reference_router_loss(q, c, mask, pos, teacher, in_batch=True, teacher_coef=0.0, tau=1.0):
    """citadel_task.py:249-262 + backward on {"router_repr": q} / {"router_repr": c}.  Returns (loss, dq, dc, logged)."""
    task = make_reference_citadel_task(in_batch, teacher_coef, tau)
    tq = q.clone().requires_grad_(True)
    tc = c.clone().requires_grad_(True)
    loss = task.router_loss({"router_repr": tq}, {"router_repr": tc}, mask.clone(), pos.clone(), teacher.clone())
    loss.backward()
    return loss.detach(), tq.grad.detach(), tc.grad.detach(), dict(task.logged)


def reference_distributed_gather(query_repr, context_repr, mask, pos, teacher, rank):
    """citadel_task.py:97-135 on the calling gloo rank (an initialised process group is required)."""
    task = make_reference_citadel_task(distributed=True, rank=rank)
    return task.distributed_gather(query_repr, context_repr, mask, pos, teacher)


# ---- brute-force retrieval, from the reference's own run_retrieval_pytorch.py -------------------------------
def load_reference_search_index():
    """Import search_index (run_retrieval_pytorch.py:141-176) from the reference tree, unmodified.  The module pulls in the data
    modules (hydra, ujson, pytorch_lightning ...) at import time although search_index needs none of them: `ujson` is aliased to
    json and `dpr_scale.datamodule.dpr` is replaced by a stub with the three dataset names the script imports.  The function sends
    its operands to `.cuda(0)`; on this GPU-less container torch.Tensor.cuda is patched to the identity for the duration of a call
    (fp16 einsum + topk then run on the CPU: same ops, same dtypes)."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    _install_stubs()
    import json as _json

    sys.modules.setdefault("ujson", _json)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    stub = types.ModuleType("dpr_scale.datamodule.dpr")
    stub.CSVDataset = stub.QueryCSVDataset = stub.QueryTSVDataset = type("_Unused", (), {})
    import dpr_scale.datamodule  # noqa: F401  (the reference's package)

    saved = sys.modules.get("dpr_scale.datamodule.dpr")
    sys.modules["dpr_scale.datamodule.dpr"] = stub
    try:
        import importlib

        mod = importlib.import_module("dpr_scale.run_retrieval_pytorch")
    finally:
        if saved is not None:
            sys.modules["dpr_scale.datamodule.dpr"] = saved
        else:
            sys.modules.pop("dpr_scale.datamodule.dpr", None)
    assert os.path.realpath(mod.__file__).startswith(os.path.realpath(REFERENCE_ROOT))
    return mod.search_index


def reference_search_index(query_embs, corpus_embs, batch, topk):
    """(scores, ids) exactly as the reference's search_index returns them (corpus_embs fp16, as build_index leaves the index)."""
    fn = load_reference_search_index()
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        scores, ids = fn(query_embs, corpus_embs, batch, topk)
    finally:
        torch.Tensor.cuda = real_cuda
    return torch.as_tensor(scores), torch.as_tensor(ids)
