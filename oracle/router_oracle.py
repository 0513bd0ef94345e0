"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the CITADEL router-loss path.

Only ``tests/`` may import this module, and only as the checker; the product (``dpr_scale_amd``) never does.

Pinning: the reference has no tests for ``dpr_scale/task/citadel_task.py``.  This restatement is pinned against that
file itself, imported unmodified through ``oracle/ref_shim.py`` --
  * live in the build container (``tests/test_oracle_vs_reference.py``), and
  * through ``tests/golden/router_*.npz``, written by ``oracle/make_golden.py`` from the reference's outputs
    (``tests/test_citadel_router.py`` runs everywhere, including the GPU box).

Arithmetic in torch float64 on the CPU (autograd gives the gradients); every function cites the lines it restates.
"""
import numpy as np
import torch

from .inbatch_oracle import bf16_round

ROUTER_D = 30522  # bert-base-uncased vocabulary: the width of CITADEL's router vectors (citadel_task.py:249-262)


def synth_router(seed, B, M, d=ROUTER_D, mask_p=0.1):
    """Seeded router-shaped inputs (numpy PCG64: version-stable).  q [B,d], c [B*M,d] fp32, bf16-representable: sparse
    non-negative activations (a ReLU-ed MLM head) scaled so that the logits are of order one; context b*M is query b's
    positive.  mask [B*M] bool (dummy contexts, never a positive), pos [B] int64 = b*M, teacher [B,M] fp32."""
    rng = np.random.default_rng(seed)
    s = 4.0 * d ** -0.5
    q = bf16_round(np.maximum(rng.standard_normal((B, d)).astype(np.float32) - 1.0, 0.0) * s)
    c = np.maximum(rng.standard_normal((B * M, d)).astype(np.float32) - 1.0, 0.0) * s
    c[np.arange(B) * M] += 0.5 * q
    c = bf16_round(c)
    mask = rng.random(B * M) < mask_p
    mask[np.arange(B) * M] = False
    pos = (np.arange(B) * M).astype(np.int64)
    teacher = rng.standard_normal((B, M)).astype(np.float32)
    return q, c, mask, pos, teacher


def sim_score(q, c, mask=None, pairwise=False):
    """citadel_task.py:137-153.  pairwise: [B, M] = <q[b], c[b*M + j]>, -inf at mask[b*M + j]; dense: [Nq, Nc] = q c^T, -inf in
    masked columns (the reference broadcasts the column mask with mask.repeat, :152)."""
    if pairwise:
        M = c.shape[0] // q.shape[0]                                           # :139
        scores = (q.unsqueeze(1) * c.view(-1, M, c.shape[1])).sum(-1)          # :140-143
        if mask is not None:
            scores = scores.masked_fill(mask.view(-1, M), float("-inf"))       # :144-145
    else:
        scores = q @ c.t()                                                     # :147-149
        if mask is not None:
            scores = scores.masked_fill(mask[None, :], float("-inf"))          # :150-152
    return scores


def distilled_loss(input_logits, target_logits):
    """citadel_task.py:240-247: cross-entropy of softmax(input) against softmax(target) with a 1e-6 floor inside the log."""
    x = input_logits - input_logits.max(-1, True).values.detach()              # :241
    t = target_logits - target_logits.max(-1, True).values.detach()            # :242
    return -(torch.softmax(t, -1) * torch.log(torch.softmax(x, -1) + 1e-6)).sum(-1).mean(0)   # :244-246


def router_loss(q, c, mask, pos, teacher, in_batch=True, teacher_coef=0.0, tau=1.0):
    """citadel_task.py:249-262 with self.loss = nn.CrossEntropyLoss() (dpr_task.py:46)."""
    loss = 0.0
    if 1 - teacher_coef > 0:                                                   # :251
        s = sim_score(q, c, mask, pairwise=not in_batch)                       # :252
        if not in_batch:
            pos = torch.zeros(len(s), dtype=torch.int64)                       # :253-254
        loss = torch.nn.functional.cross_entropy(s, pos)                       # :255
    if teacher_coef > 0:                                                       # :256
        ps = sim_score(q, c, mask, pairwise=True)                              # :257-259
        loss = (1 - teacher_coef) * loss + teacher_coef * distilled_loss(ps / tau, teacher / tau)   # :260
    return loss


def router_step(q, c, mask, pos, teacher, in_batch=True, teacher_coef=0.0, tau=1.0, dtype=torch.float64):
    """(loss, dq, dc) as numpy, from numpy inputs."""
    tq = torch.from_numpy(np.asarray(q)).to(dtype).requires_grad_(True)
    tc = torch.from_numpy(np.asarray(c)).to(dtype).requires_grad_(True)
    loss = router_loss(tq, tc, torch.from_numpy(np.asarray(mask)), torch.from_numpy(np.asarray(pos)),
                       torch.from_numpy(np.asarray(teacher)).to(dtype), in_batch, teacher_coef, tau)
    loss.backward()
    return float(loss.item()), tq.grad.numpy(), tc.grad.numpy()


def synth_gather_rank(seed, rank, B=3, M=2, V=11, dim=4):
    """Per-rank inputs of the ragged gather (citadel_task.py:97-135): token sequences whose length differs between ranks."""
    rng = np.random.default_rng(seed + rank)
    Lq, Lc = 5 + rank, 7 - 2 * rank
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    qr = {"router_repr": f(B, V), "expert_repr": f(B, Lq, dim)}
    cr = {"router_repr": f(B * M, V), "expert_repr": f(B * M, Lc, dim)}
    mask = rng.random(B * M) < 0.3
    pos = (np.arange(B) * M).astype(np.int64)
    teacher = f(B, M)
    return qr, cr, mask, pos, teacher
