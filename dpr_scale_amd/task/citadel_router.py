"""CITADEL router loss on the MI355X hot path: the part of
dpr_scale/task/citadel_task.py that is the SAME dense Q x C^T + CrossEntropyLoss as DPR, on vocabulary-wide router
vectors (d = 30522 -- the one shape of the reference that is genuinely MFMA-bound), plus its ragged multi-GPU gather.

  citadel_task.py:137-153  sim_score(query_repr, context_repr, mask, pairwise)  -> RouterScoring.sim_score
  citadel_task.py:249-262  router_loss(...)                                     -> RouterScoring.router_loss
  citadel_task.py:79-95    evenly_divisible_all_gather                          -> ragged_all_gather
  citadel_task.py:97-135   distributed_gather                                   -> distributed_gather

The rest of the CITADEL task (token-level expert scoring, regularisers, the encoder heads) is outside the path
 and stays the reference's Python; `RouterScoring` is a mix-in that a CITADEL task built on
dpr_scale_amd.task.dpr_task.DenseRetrieverTask inherits its scoring from.
"""
import torch
import torch.distributed as dist

from .. import hotpath


def distilled_loss(input_logits, target_logits):
    """citadel_task.py:240-247 (plain torch: [B, M] tensors, M = contexts per query)."""
    input_logits = input_logits - input_logits.max(-1, True).values.detach()
    target_logits = target_logits - target_logits.max(-1, True).values.detach()
    input_probs = torch.softmax(input_logits, dim=-1)
    target_probs = torch.softmax(target_logits, dim=-1)
    return -(target_probs * torch.log(input_probs + 1e-6)).sum(-1).mean(0)


class RouterScoring:
    """Mix-in: expects `self.loss` (callable (scores, labels) -> scalar), `self.kernels`, `self.in_batch`,
    `self.teacher_coef`, `self.tau` and `self.log` as the reference's CITADELTask has them."""

    def sim_score(self, query_repr, context_repr, mask=None, pairwise=False):
        """citadel_task.py:137-153.  pairwise=False: [Nq, Nc] = Q x C^T on the bf16 MFMA path, masked columns -inf;
        pairwise=True: [B, M] dot products of each query with its own M contexts (fp32 streaming kernels).
        Both are differentiable."""
        kn = getattr(self, "kernels", None)
        if pairwise:
            m = None if mask is None else mask.reshape(-1)
            return hotpath.pairwise_score(query_repr, context_repr, m, kn)
        col = mask
        full = None
        if mask is not None and mask.dim() == 2:
            col, full = None, mask
        scores = hotpath.sim_score(query_repr, context_repr, col, 1.0, kn)
        if full is not None:
            scores = scores.masked_fill(full, float("-inf"))
        return scores

    def _stock_scoring(self):
        """True when neither sim_score nor the loss has been replaced by a subclass / the user: only then may router_loss skip the
        materialised score matrix."""
        from .dpr_task import HotCrossEntropyLoss

        loss = getattr(self, "loss", None)
        if type(self).sim_score is not RouterScoring.sim_score or not isinstance(loss, (HotCrossEntropyLoss, torch.nn.CrossEntropyLoss)):
            return False
        # the fused operator computes the PLAIN mean cross-entropy: a configured torch loss with class weights, label smoothing,
        # another reduction or an ignore_index in use keeps the materialised path (the reference calls self.loss as configured)
        if isinstance(loss, torch.nn.CrossEntropyLoss) and not isinstance(loss, HotCrossEntropyLoss):
            return (loss.weight is None and float(getattr(loss, "label_smoothing", 0.0)) == 0.0 and loss.reduction == "mean"
                    and loss.ignore_index == -100)
        return True

    def router_loss(self, query_repr, context_repr, mask, pos_ctx_indices, teacher_scores):
        """citadel_task.py:249-262."""
        router_loss = 0.0
        if 1 - self.teacher_coef > 0:
            q, c = query_repr["router_repr"], context_repr["router_repr"]
            if self.in_batch and self._stock_scoring() and q.is_cuda and (mask is None or mask.dim() == 1):
                # CrossEntropyLoss(sim_score(q, c, mask), labels) on already-gathered vectors IS the in-batch contrastive step
                # (dpr_task.py:197-212 at T = 1 without a gather): one fused operator -- fp32 vectors read once, no [Nq, Nc] logits
                # in HBM, both backward GEMMs in the forward call -- instead of casts + sim + cross-entropy + two backward launches
                m = mask if mask is not None else torch.zeros(c.shape[0], dtype=torch.bool, device=c.device)
                router_loss = hotpath.inbatch_contrastive_loss(q, c, pos_ctx_indices, m, 1.0, False, getattr(self, "kernels", None))
            else:
                router_scores = self.sim_score(q, c, mask, pairwise=not self.in_batch)
                if not self.in_batch:
                    pos_ctx_indices = torch.zeros(len(router_scores), dtype=torch.int64, device=router_scores.device)
                router_loss = self.loss(router_scores, pos_ctx_indices)
        if self.teacher_coef > 0:
            pairwise_router_scores = self.sim_score(query_repr["router_repr"], context_repr["router_repr"], mask, pairwise=True)
            router_loss = (1 - self.teacher_coef) * router_loss + self.teacher_coef * distilled_loss(
                pairwise_router_scores / self.tau, teacher_scores / self.tau)
        self.log("train_router_loss", router_loss, prog_bar=True)
        return router_loss


# ---- ragged multi-GPU gather ------------------------------------------------------------------------------------------
def _all_gather_stack(src, group):
    """[W, *src.shape] -- one all_gather_into_tensor on flat views (what PL's all_gather returns, without W list entries)."""
    W = dist.get_world_size(group)
    src = src.contiguous()
    out = torch.empty((W * max(src.numel(), 1),), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src.reshape(-1), group=group)
    return out.view((W,) + tuple(src.shape))


def ragged_all_gather(tensors, group=None):
    """citadel_task.py:79-95 (`evenly_divisible_all_gather`): tensors whose dim 1 differs between ranks (token sequences
    padded per rank) are zero-padded to the longest and all-gathered; returns [W, ...] stacks like PL's all_gather.
    The reference issues one length all-gather PER tensor (plus the payload gathers); here every length travels in ONE
    small all-gather, then one payload all-gather per tensor."""
    nd = [i for i, t in enumerate(tensors) if t.dim() > 1]
    max_len = {}
    if nd:
        lens = torch.tensor([tensors[i].shape[1] for i in nd], dtype=torch.int64, device=tensors[nd[0]].device)
        all_lens = _all_gather_stack(lens, group)  # [W, len(nd)]
        mx = all_lens.max(dim=0).values.tolist()   # one host sync for all lengths
        max_len = {i: int(m) for i, m in zip(nd, mx)}
    res = []
    for i, t in enumerate(tensors):
        src = t.detach()
        if i in max_len and src.shape[1] < max_len[i]:
            pad = src.new_zeros([src.shape[0], max_len[i] - src.shape[1]] + list(src.shape[2:]))
            src = torch.cat([src, pad], dim=1)
        is_bool = src.dtype == torch.bool
        g = _all_gather_stack(src.to(torch.uint8) if is_bool else src, group)
        res.append(g.to(torch.bool) if is_bool else g)
    return res


def _pad_dim1(tensor_list):
    """citadel_task.py:68-77 (`pad`)."""
    max_len = max(t.size(1) for t in tensor_list)
    return [torch.cat([t, t.new_zeros([t.shape[0], max_len - t.size(1)] + list(t.shape[2:]))], dim=1) if t.size(1) < max_len else t
            for t in tensor_list]


def distributed_gather(query_repr, context_repr, mask, pos_ctx_indices, teacher_scores, rank, group=None):
    """citadel_task.py:97-135: gather the representation dicts, labels, mask and teacher scores of every rank; this rank's own
    (grad-carrying) tensors are spliced into slot `rank`; labels get the running context offset."""
    q_keys, c_keys = list(query_repr.keys()), list(context_repr.keys())
    q_send, c_send = list(query_repr.values()), list(context_repr.values())
    gathered = ragged_all_gather([*q_send, *c_send, pos_ctx_indices, mask, teacher_scores], group)
    gq, gc = gathered[:len(q_send)], gathered[len(q_send):len(q_send) + len(c_send)]
    all_labels, all_mask, all_teacher = gathered[-3].clone(), gathered[-2], gathered[-1]
    q_lists, c_lists = [[] for _ in q_keys], [[] for _ in c_keys]
    offset = 0
    for i in range(all_labels.size(0)):
        all_labels[i] += offset
        t = None
        for j in range(len(q_keys)):
            q_lists[j].append(q_send[j] if i == rank else gq[j][i])
        for j in range(len(c_keys)):
            t = c_send[j] if i == rank else gc[j][i]
            c_lists[j].append(t)
        offset += t.size(0)
    cat = lambda lst: torch.cat(_pad_dim1(lst) if lst[0].dim() > 1 else lst, dim=0)
    out_q = {k: cat(v) for k, v in zip(q_keys, q_lists)}
    out_c = {k: cat(v) for k, v in zip(c_keys, c_lists)}
    return out_q, out_c, torch.flatten(all_mask), torch.flatten(all_labels), all_teacher.reshape(-1, all_teacher.shape[-1])
