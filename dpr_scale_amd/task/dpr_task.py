"""Drop-in for dpr_scale.task.dpr_task.DenseRetrieverTask (reference: dpr_scale/task/dpr_task.py:17-368).

Same constructor kwargs (:18-32), same Lightning hooks, same overridable methods (`sim_score`, `encode_queries`,
`encode_contexts`, `_encode_sequence`, `self.loss`, `compute_rank_metrics`, `_eval_step`, `_eval_epoch_end`),
same logged metric names and the same checkpoint layout (`query_encoder.*` / `context_encoder.*` +
`hyper_parameters`).  What changes is what happens between the encoder outputs and the loss:

  reference (:163-212)                                   here
  4 fp32 all_gathers, python splice loop, 2 torch.cat    1 bf16 all-gather of context rows (+ mask bytes)
  full [W*B, W*B*K] matmul on EVERY rank                 this rank's B rows only (HIP bf16 MFMA)
  mask.repeat, masked_fill, /= T, log_softmax, nll       fused in the GEMM epilogue + one softmax/dScores pass
  autograd through cat/matmul (7/8 of it discarded)      HIP dQ / dC GEMMs + reduce-scatter of dC

The maths per rank is identical (tests/golden/*_ddp.npz come from the reference's own
DDP branch).  Every device op goes through libdprhot.so (dpr_scale_amd.hotpath); there is no CPU fallback.
"""
import os

import torch
import torch.nn as nn
from torch.optim.lr_scheduler import LambdaLR

from .. import hotpath

try:  # the real runtime when it is installed ...
    from pytorch_lightning import LightningModule
    from pytorch_lightning.strategies import DDPShardedStrategy, DDPStrategy
except ImportError:  # ... the minimal stand-in otherwise (build image: no pytorch_lightning, no network)
    from ..lightning_compat import DDPShardedStrategy, DDPStrategy, LightningModule

try:
    from hydra.utils import instantiate
except ImportError:
    from ..hydra_compat import instantiate


class _CEFunction(torch.autograd.Function):
    """Mean row cross-entropy on existing fp32 logits with gradient (subclasses train through `self.loss`)."""

    @staticmethod
    def forward(ctx, scores, labels, kernels):
        kn = kernels if kernels is not None else hotpath.default_kernels()
        S = scores.detach().float().contiguous()
        row_loss, _, G = kn.softmax_ce(S, labels, 0, 1.0 / S.shape[0], want_G=True)
        ctx.save_for_backward(G)
        ctx.dtype = scores.dtype
        return kn.reduce_sum(row_loss, 1.0 / S.shape[0]).reshape(())

    @staticmethod
    def backward(ctx, go):
        (G,) = ctx.saved_tensors
        return (G.float() * go).to(ctx.dtype), None, None


class HotCrossEntropyLoss(nn.Module):
    """`self.loss` of the task: callable (scores, labels) -> scalar, nn.CrossEntropyLoss() semantics (:46)."""

    def __init__(self, kernels=None):
        super().__init__()
        self.kernels = kernels

    def forward(self, scores, labels):
        labels = torch.as_tensor(labels, dtype=torch.long, device=scores.device)
        if scores.shape[1] % 8 != 0:
            scores = torch.nn.functional.pad(scores, (0, 8 - scores.shape[1] % 8), value=float("-inf"))
        if scores.requires_grad:
            return _CEFunction.apply(scores, labels, self.kernels)
        return hotpath.cross_entropy_mean(scores.float(), labels, self.kernels)


class DenseRetrieverTask(LightningModule):
    def __init__(
        self,
        transform,
        model,
        datamodule,
        optim,
        k=1,
        shared_model: bool = True,
        in_batch_eval: bool = True,
        in_batch_negatives: bool = True,
        warmup_steps: int = 0,
        fp16_grads: bool = False,
        pretrained_checkpoint_path: str = "",
        softmax_temperature: float = 1.0,
    ):
        super().__init__()
        self.save_hyperparameters()
        # the two runtime switches the path was measured with, set explicitly here (no longer by `import dpr_scale_amd`); where their
        # runtime is already up this only warns -- launchers call dpr_scale_amd.configure_runtime() first
        from .. import configure_runtime

        configure_runtime()
        self.transform_conf = getattr(transform, "text_transform", transform)
        self.model_conf, self.optim_conf = model, optim
        self._datamodule_conf = datamodule  # (only read for the size of a step's messages: _path_message_shape)
        self.shared_model = shared_model
        self.k = k
        self.kernels = None  # None = libdprhot.so (the only product path); tests may inject a stand-in
        self.loss = HotCrossEntropyLoss()
        self.in_batch_eval = in_batch_eval
        self.in_batch_negatives = in_batch_negatives
        self.warmup_steps = warmup_steps
        self.fp16_grads = fp16_grads
        self.pretrained_checkpoint_path = pretrained_checkpoint_path
        self.softmax_temperature = softmax_temperature
        self.setup_done = False
        # multi-GPU step: contexts first (collectives hidden under the query tower) unless the reference's order
        # is asked for (dropout RNG stream identical to the reference's, see training_step)
        # DPRHOT_TOWER_ORDER = context_first (default) | reference | auto (the timing trial, opt-in): see _tower_order()
        self.context_tower_first = os.environ.get("DPRHOT_TOWER_ORDER", "") != "reference"
        self._order_trial = None
        self.path_probe = None  # what on_pretrain_routine_start measured for the path's collectives (saved with the checkpoint)

    # ---- model construction / checkpoints (reference :55-92) -------------------------------------------
    def setup(self, stage: str):
        if stage == "test" and self.setup_done:
            return  # keep the restored weights
        self.call_configure_sharded_model_hook = False
        self.query_encoder = instantiate(self.model_conf)
        self.context_encoder = self.query_encoder if self.shared_model else instantiate(self.model_conf)
        if self.pretrained_checkpoint_path:
            ck = torch.load(self.pretrained_checkpoint_path, map_location="cpu", weights_only=False)
            self.load_state_dict(ck["state_dict"])
            print(f"Loaded state dict from {self.pretrained_checkpoint_path}")
        self.setup_done = True

    def on_load_checkpoint(self, checkpoint) -> None:
        self.setup("fit")  # modules must exist before Lightning restores the state dict

    def on_pretrain_routine_start(self):
        """Reference :90-92: `fp16_grads` registers torch's fp16_compress_hook (one half-precision ring all-reduce per
        bucket).  Here the same switch registers dpr_scale_amd.comm_hooks.compressed_allreduce_hook: half the bytes on the
        wire as well, but as an all-pairs exchange with fp32 accumulation (7 xGMI links per GPU instead of a one-link
        ring).  The wire format is the reference's fp16 by default (same flag, same numbers on the wire; the sum itself
        is fp32 here); DPRHOT_GRAD_WIRE=bf16 trades mantissa for fp32's exponent range, DPRHOT_GRAD_MODE=ring selects the
        reference's decomposition."""
        if self._is_distributed() and (hotpath.D.world(None)[0] > 1 or hotpath.D.force_dist()):
            # opt-in (DPRHOT_DIRECT_RCCL=1): the path's three collectives through the C ABI communicator (collective set-up under a
            # watchdog, with a self-check against torch.distributed; every rank keeps torch.distributed if any rank cannot build it)
            try:
                dev = next(self.parameters()).device
            except StopIteration:
                dev = None
            if dev is not None and dev.type == "cuda":
                hotpath.D.enable_direct_comm(dev)
                # which FORM of the path's two collectives on this node -- RCCL's own (rings) or the direct all-pairs exchange -- is
                # measured once, here, at the step's message sizes, and agreed on by all ranks (dist.choose_path_collectives;
                # DPRHOT_PATH_COLLECTIVES=rccl|allpairs pins it, DPRHOT_PATH_PROBE=0 skips the probe and keeps RCCL's)
                # Round 6: the probe also chooses the dC WIRE (fp32 or bf16; bf16 only where its best form wins by 5 %) unless
                # DPRHOT_DC_WIRE names one or the run asked for reproducibility (_wants_reproducible: a format picked by a stopwatch
                # is not reproducible); a form pinned by hand is not measured at all; every candidate is agreed on by all ranks before
                # and after its first iteration, and the whole probe runs under a watchdog (DPRHOT_PROBE_TIMEOUT_S, default 300 s: a
                # probe that does not come back ends the process with a message instead of a silent hang).
                D = hotpath.D
                if os.environ.get("DPRHOT_PATH_PROBE", "1") != "0" and D.world(None)[0] > 1:
                    rows_c, d = self._path_message_shape()
                    env_wire = os.environ.get("DPRHOT_DC_WIRE", "auto")
                    if env_wire in ("bf16", "fp32") or self._wants_reproducible():
                        wires = (torch.bfloat16 if env_wire == "bf16" else torch.float32,)
                    else:
                        wires = (torch.float32, torch.bfloat16)
                    if not (D.path_is_pinned() and len(wires) == 1):
                        with D.probe_watchdog(float(os.environ.get("DPRHOT_PROBE_TIMEOUT_S", "300")), "dpr_scale_amd: the probe of the path's collectives"):
                            self.path_collectives = D.choose_path_collectives(dev, rows_c, d, wires=wires)
                        rec = D._PROBED.get(D._gkey(None))
                        if rec is not None:
                            self.path_probe = {"topology": rec["topology"], "wire": str(rec.get("wire")).replace("torch.", ""), "us": rec.get("us_by_wire")}
                            if getattr(self, "global_rank", 0) == 0:
                                print(f"dpr_scale_amd: collectives of the contrastive path: {self.path_probe}", flush=True)
        if self.fp16_grads:
            from .. import comm_hooks

            self.grad_comm_state = comm_hooks.register(self.trainer.strategy._model, comm_hooks.GradCommState.from_env(default_wire="fp16"))

    def _path_message_shape(self):
        """(packed rows per rank, hidden size) of a training step's all-gather, from the configs where they say it (batch size, contexts
        per query: dpr_transform.py:143-161 pads every query to 1 + num_negative contexts), else BASELINE configs[2]'s 1032 x 768."""
        rows, d = 1032, 768
        try:
            bs = int(self._datamodule_conf.get("batch_size"))
            tr = self.transform_conf
            k = int(tr.get("num_positive", 1)) + int(tr.get("num_negative", 7))
            for p in self.query_encoder.parameters():
                if p.dim() == 2:
                    d = int(p.shape[-1])  # (the last 2-D weight met is the encoder's output-side matrix in every recipe; only a size estimate)
            from .. import _lib

            rows = _lib.packed_rows(bs * k, d)
        except Exception:
            pass
        return rows, d

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, **kwargs):
        base = super()
        if hasattr(base, "load_from_checkpoint"):
            return base.load_from_checkpoint(checkpoint_path, **kwargs)
        from ..lightning_compat import load_from_checkpoint

        return load_from_checkpoint(cls, checkpoint_path, **kwargs)

    # ---- encoders (reference :94-121) -------------------------------------------------------------------
    def _encode_sequence(self, token_ids, encoder_model):
        return encoder_model(token_ids)  # [rows, d]

    def encode_queries(self, query_ids):
        return self._encode_sequence(query_ids, self.query_encoder)

    def encode_contexts(self, contexts_ids):
        return self._encode_sequence(contexts_ids, self.context_encoder)

    def forward(self, query_ids, contexts_ids):
        return self.encode_queries(query_ids), self.encode_contexts(contexts_ids)

    # ---- scoring (reference :98-105) -----------------------------------------------------------------------
    def sim_score(self, query_repr, context_repr, mask=None):
        """fp32 logits [Nq, Nc]; masked entries are -inf.  `mask` is the [Nc] dummy-context mask or, as the
        reference passes it, that row repeated to [Nq, Nc].  Differentiable like the reference's torch.matmul
        (:98-105): subclasses that train through `self.loss(self.sim_score(q, c, mask), y)` (citadel_task.py:249-262)
        get encoder gradients -- the backward runs the HIP dQ / dC GEMMs (hotpath.SimScore)."""
        col = None
        full = None
        if mask is not None:
            if mask.dim() == 1:
                col = mask
            else:
                full = mask
        scores = hotpath.sim_score(query_repr, context_repr, col, 1.0, self.kernels)
        if full is not None:
            scores = scores.masked_fill(full, float("-inf"))
        return scores

    # ---- optimiser (reference :123-151) ----------------------------------------------------------------------
    def configure_optimizers(self):
        self.optimizer = instantiate(self.optim_conf, self.parameters())
        if self.trainer.max_steps and self.trainer.max_steps > 0:
            total = self.trainer.max_steps
        else:
            total = len(self.trainer.datamodule.train_dataloader()) * self.trainer.max_epochs
        warm = self.warmup_steps
        print(f"Configured LR scheduler for total {total} training steps, with {warm} warmup steps.")

        def lr_lambda(step):  # linear warm-up then linear decay to zero
            if step < warm:
                return float(step) / float(max(1, warm))
            return max(0.0, float(total - step) / float(max(1, total - warm)))

        sched = {"scheduler": LambdaLR(self.optimizer, lr_lambda), "name": "learning_rate", "interval": "step",
                 "frequency": 1}
        return [self.optimizer], [sched]

    # ---- the hot path (reference :153-214) -----------------------------------------------------------------
    def _is_distributed(self):
        return isinstance(getattr(self.trainer, "strategy", None), (DDPStrategy, DDPShardedStrategy))

    # The order of the two towers in the multi-GPU step.
    #   context_first  (default) the context tower's rows go into the one all-gather, which runs underneath the query tower; in backward
    #                  the reduce-scatter of dC runs underneath the query-tower backward (hotpath.ContextGather / defer_context_grad).
    #   reference      the reference's order (dpr_task.py:94-101: query tower, context tower); both collectives run exposed; dropout
    #                  masks are drawn in the reference's sequence (seed-for-seed comparison).
    #   auto           OPT-IN (DPRHOT_TOWER_ORDER=auto): the first steps time both orders and training continues in the faster one.
    # Hiding the collectives is not free: autograd runs the NEWEST tower's backward first, so context_first puts the SMALL tower's
    # backward (B rows, ~700 launches of ~10 us) right behind the loss, where nothing is queued ahead of it -- the GPU outruns the host's
    # launches and idles (profiles/r05_forced_dist_breakdown.json: +0.9...1.3 ms of device idle per step at B = 32 on one MI355X, with
    # NO collective involved) -- whereas in the reference's order the big tower's backward goes first and the small one's launches
    # queue up behind it.  Which effect is larger depends on the host, the batch and the node's links: hence the trial.
    # Round 6: the trial is no longer the default -- the two orders draw dropout masks in a different sequence, so a run
    # whose first steps alternate them and whose final order depends on a stopwatch is not reproducible from its seed, which the
    # reference is; it is also refused when the run asked for reproducibility (_wants_reproducible).  When it does run, the orders
    # alternate per ACCUMULATION CYCLE (accumulate_grad_batches steps), so that each order meets the optimizer / DDP-sync step equally
    # often, and the decision is stored with the checkpoint (on_save_checkpoint: "dprhot_runtime").
    _TRIAL_WARM, _TRIAL_N = 3, 5

    def _wants_reproducible(self):
        """The run asked for seed-for-seed reproducibility: deterministic algorithms switched on, or Lightning's seed_everything
        (PL_GLOBAL_SEED) / Trainer(deterministic=True)."""
        if torch.are_deterministic_algorithms_enabled() or os.environ.get("PL_GLOBAL_SEED"):
            return True
        return bool(getattr(getattr(self, "trainer", None), "deterministic", False))

    def _tower_order(self):
        if not self.context_tower_first:
            return "reference"
        mode = os.environ.get("DPRHOT_TOWER_ORDER", "context_first")
        if mode != "auto" or not torch.cuda.is_available() or self._wants_reproducible():
            return "context_first"
        tr = self._order_trial
        if tr is None:
            acc = int(getattr(getattr(self, "trainer", None), "accumulate_grad_batches", 1) or 1)
            tr = self._order_trial = {"step": 0, "events": [], "decided": None, "ms": None, "acc": max(1, acc)}
        if tr["decided"] is not None:
            return tr["decided"]
        i, N, acc = tr["step"], self._TRIAL_N, tr["acc"]
        W0 = -(-self._TRIAL_WARM // acc) * acc  # whole accumulation cycles of warm-up
        tr["step"] += 1
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        tr["events"].append(ev)
        if i < W0:
            return "context_first"
        if i < W0 + 2 * N * acc + 1:
            # the two orders ALTERNATE cycle by cycle (clocks and caches drift over the first steps of a run: two blocks of cycles would
            # time the drift); cycle j's interval runs from the start event of its first step to that of the next cycle's first step
            return "context_first" if ((i - W0) // acc) % 2 == 0 else "reference"
        evs = tr["events"]
        evs[W0 + 2 * N * acc].synchronize()  # (recorded a whole step ago: complete)

        def med(first):
            ts = sorted(evs[W0 + j * acc].elapsed_time(evs[W0 + (j + 1) * acc]) / acc for j in range(first, 2 * N, 2))
            return 0.5 * (ts[(len(ts) - 1) // 2] + ts[len(ts) // 2])

        t = torch.tensor([med(0), med(1)], dtype=torch.float64, device="cuda")
        if hotpath.D.world(None)[0] > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        a_ms, b_ms = t.tolist()
        tr["ms"] = {"context_first": a_ms, "reference": b_ms}
        tr["decided"] = "reference" if b_ms < 0.995 * a_ms else "context_first"
        tr["events"] = []
        if getattr(self, "global_rank", 0) == 0:
            print(f"dpr_scale_amd: tower order of the multi-GPU step: {tr['decided']} (context first {a_ms:.2f} ms / step, reference order {b_ms:.2f} ms / step)",
                  flush=True)
        return tr["decided"]

    def on_save_checkpoint(self, checkpoint) -> None:
        """What this run measured and decided at start-up rides with the checkpoint (an extra key: the reference's layout -- state_dict,
        hyper_parameters -- is untouched): the tower order (fixed, or the trial's decision and its timings) and the probe of the path's
        collectives (form, wire)."""
        tr = self._order_trial or {}
        checkpoint["dprhot_runtime"] = {"tower_order": tr.get("decided") or ("context_first" if self.context_tower_first else "reference"),
                                        "tower_order_trial_ms": tr.get("ms"), "path_collectives": self.path_probe}

    def training_step(self, batch, batch_idx):
        pos, mask = batch["pos_ctx_indices"], batch["ctx_mask"]
        T = self.softmax_temperature
        if (self.in_batch_negatives and self._is_distributed() and (hotpath.D.world(None)[0] > 1 or hotpath.D.force_dist())
                and type(self).forward is DenseRetrieverTask.forward and self._tower_order() == "context_first"):
            # Multi-GPU (reference :163-195).  Context tower FIRST: its rows go into the one all-gather, which then
            # runs on RCCL's stream underneath the query tower; in backward the reduce-scatter of dC overlaps the
            # query-tower backward the same way (hotpath.ContextGather / defer_context_grad).  The encoders are
            # independent, so the order changes q and c only through the dropout RNG stream (the reference draws the
            # query tower's masks first); set `context_tower_first = False` (or DPRHOT_TOWER_ORDER=reference) for a
            # seed-for-seed comparison -- the collectives then run exposed.  A subclass that overrides forward()
            # keeps the reference's call below.
            c = self.encode_contexts(batch["contexts_ids"])
            c, pending = hotpath.defer_context_grad(c)
            gather = hotpath.ContextGather(c, mask, None, self.kernels)
            q = self.encode_queries(batch["query_ids"])
            loss = hotpath.inbatch_contrastive_loss(q, c, pos, mask, T, None, self.kernels, gather, pending)
            self.log("train_loss", loss, prog_bar=True)
            return loss
        q, c = self(batch["query_ids"], batch["contexts_ids"])
        if self.in_batch_negatives:
            group = None if self._is_distributed() else False  # False: never gather (single-device strategies)
            loss = hotpath.inbatch_contrastive_loss(q, c, pos, mask, T, group, self.kernels)
        else:
            loss = hotpath.windowed_contrastive_loss(q, c, pos, mask, T, self.kernels)
        self.log("train_loss", loss, prog_bar=True)
        return loss

    # ---- evaluation (reference :216-310) ---------------------------------------------------------------------
    def _eval_step(self, batch, batch_idx):
        q, c = self(batch["query_ids"], batch["contexts_ids"])
        labels, mask = batch["pos_ctx_indices"], batch["ctx_mask"]
        scores = self.sim_score(q, c, mask)
        loss = self.loss(scores, labels)
        return (self.compute_rank_metrics(scores, labels), q, c, labels, mask, loss)

    def compute_rank_metrics(self, pred_scores, target_labels):
        """(sum of ranks, sum of reciprocal ranks, #rows with rank <= k) -- the reference sorts every row and
        looks the gold index up with one host sync per row (:237-245); here one count kernel, one sync."""
        labels = torch.as_tensor(target_labels, dtype=torch.long, device=pred_scores.device)
        ranks = hotpath.rank_of_gold(pred_scores.float(), labels, self.kernels)
        return self._rank_stats(ranks)

    def _rank_stats(self, ranks):
        r = ranks.double()
        stats = torch.stack([r.sum(), (1.0 / r).sum(), (ranks - 1 < self.k).double().sum()]).tolist()
        return int(stats[0]), stats[1], int(stats[2])

    def _eval_epoch_end(self, outputs, log_prefix="valid"):
        if self.in_batch_eval:
            n = len(outputs)
            rank = sum(o[0][0] for o in outputs)
            mrr = sum(o[0][1] for o in outputs)
            score = sum(o[0][2] for o in outputs)
            count = sum(o[1].size(0) for o in outputs)
            ctx_count = sum(o[2].size(0) - torch.sum(o[4]) for o in outputs) / n
            loss = sum(o[5] for o in outputs) / n
        else:
            q_all = torch.cat([o[1] for o in outputs], dim=0)
            c_all = torch.cat([o[2] for o in outputs], dim=0)
            m_all = torch.cat([o[4] for o in outputs], dim=0)
            labels, offset = [], 0
            for o in outputs:
                labels.extend(int(x) + offset for x in o[3])
                offset += o[2].size(0)
            if self.trainer.world_size > 1:
                c_g, m_g = self.all_gather((c_all, m_all))
                labels = [x + c_g.size(1) * self.global_rank for x in labels]
                c_all, m_all = c_g.flatten(0, 1), m_g.flatten(0, 1)
            count = q_all.size(0)
            ctx_count = c_all.size(0) - torch.sum(m_all)
            cls = type(self)
            stock = (cls.sim_score is DenseRetrieverTask.sim_score and cls.compute_rank_metrics is DenseRetrieverTask.compute_rank_metrics
                     and isinstance(self.loss, HotCrossEntropyLoss) and q_all.is_cuda)
            if stock:
                # the reference's formulation (:291-302) through the score-free kernels: ranks by count-greater inside the
                # similarity GEMM, loss from its online softmax statistics -- the [Nq, Nc] matrix is never stored
                ranks, loss = hotpath.rank_and_loss(q_all.detach(), c_all.detach(), labels, m_all.reshape(-1), 1.0, self.kernels)
                rank, mrr, score = self._rank_stats(ranks)
            else:
                scores = self.sim_score(q_all, c_all, m_all)
                rank, mrr, score = self.compute_rank_metrics(scores, labels)
                loss = self.loss(scores, torch.tensor(labels, dtype=torch.long, device=scores.device))
        self.log_dict({
            f"{log_prefix}_avg_rank": rank / count,
            f"{log_prefix}_mrr": mrr / count,
            f"{log_prefix}_accuracy@{self.k}": score / count,
            f"{log_prefix}_ctx_count": ctx_count,
            f"{log_prefix}_loss": loss,
        }, on_epoch=True, sync_dist=True)

    def validation_step(self, batch, batch_idx):
        return self._eval_step(batch, batch_idx)

    def validation_epoch_end(self, valid_outputs):
        if valid_outputs:
            self._eval_epoch_end(valid_outputs)

    def test_step(self, batch, batch_idx):
        return self._eval_step(batch, batch_idx)

    def test_epoch_end(self, test_outputs):
        if test_outputs:
            self._eval_epoch_end(test_outputs, "test")

    def to_torchscript(self, file_path=None, method="script", example_inputs=None, **kwargs):
        """TorchScript export of the encoders (reference :325-368): {"ctx_encoder", "ctx_encoder_qt"[, "q_encoder", "q_encoder_qt"]},
        the context encoder saved to file_path.  Encoder packaging, outside the hot path: plain PyTorch
        through dpr_scale_amd.utils.script_encoder.ScriptEncoder (same contract as the reference's; no reference package needed)."""
        from ..utils.script_encoder import ScriptEncoder

        if method != "script":
            raise ValueError(f"The 'method' parameter only supports 'script', but value given was: {method}")
        transform = instantiate(self.transform_conf)
        mode = self.training
        with torch.no_grad():
            result = {"ctx_encoder": torch.jit.script(ScriptEncoder(transform, self.context_encoder).eval(), **kwargs),
                      "ctx_encoder_qt": torch.jit.script(ScriptEncoder(transform, self.context_encoder, quantize=True).eval(), **kwargs)}
            if not self.shared_model:
                result["q_encoder"] = torch.jit.script(ScriptEncoder(transform, self.query_encoder).eval(), **kwargs)
                result["q_encoder_qt"] = torch.jit.script(ScriptEncoder(transform, self.query_encoder, quantize=True).eval(), **kwargs)
        self.train(mode)
        if file_path is not None:
            torch.jit.save(result["ctx_encoder"], file_path)
        return result
