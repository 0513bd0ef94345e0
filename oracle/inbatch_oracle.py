"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of dpr-scale's in-batch contrastive step.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the *checker*.  The product path (``dpr_scale_amd``) never imports it and has no CPU
fallback.

Pinning: the reference's own tests do not touch ``dpr_scale/task/``, so there are no
reference golden vectors for this path.  This restatement is pinned instead against the reference file
itself, executed unmodified in the build container (``oracle/ref_shim.py``), through
  * ``tests/test_oracle_vs_reference.py`` (live, skipped where /root/reference is absent), and
  * ``tests/golden/*.npz`` written by ``oracle/make_golden.py`` from the reference's outputs
    (``tests/test_oracle_golden.py`` runs everywhere, including the GPU box).

Every function cites the reference lines it restates (paths relative to /root/reference).
Notation: W ranks, B queries per rank, K contexts per query, Nq = W*B, Nc = W*B*K, d hidden size.
"""
import numpy as np

NEG_INF = -np.inf


def bf16_round(x):
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (value-preserving container for parity inputs)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    rounded = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return rounded.astype(np.uint32).view(np.float32).reshape(x.shape)


def to_bf16_bits(x):
    """fp32 -> bf16 bit pattern (uint16), round-to-nearest-even."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16).reshape(x.shape)


def from_bf16_bits(b):
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(b.shape)


def sim_score(Q, C, mask_cols=None, dtype=np.float64):
    """dpr_scale/task/dpr_task.py:98-105 -- scores = Q @ C^T ; scores[mask] = -inf.

    The reference receives a materialised [Nq, Nc] mask built by ``mask.repeat(Nq, 1)`` (:197); every row
    of it is the same column vector, so a column mask is the same thing.
    """
    S = np.asarray(Q, dtype=dtype) @ np.asarray(C, dtype=dtype).T
    if mask_cols is not None:
        S[:, np.asarray(mask_cols, dtype=bool)] = NEG_INF
    return S


def log_softmax_ce(S, labels):
    """dpr_task.py:46,212 -- nn.CrossEntropyLoss() (mean reduction) on rows of S.

    Returns (mean loss, per-row loss, per-row logsumexp).  -inf columns contribute exp(-inf) = 0.
    """
    S = np.asarray(S)
    labels = np.asarray(labels, dtype=np.int64)
    m = np.max(S, axis=1, keepdims=True)
    lse = (m + np.log(np.sum(np.exp(S - m), axis=1, keepdims=True)))[:, 0]
    row_loss = lse - S[np.arange(S.shape[0]), labels]
    return row_loss.mean(), row_loss, lse


def dscores(S, labels, lse, scale):
    """Autograd of dpr_task.py:209-212: G = (softmax(S) - onehot(y)) * scale, masked (-inf) columns -> 0.

    ``scale`` carries 1/Nq (mean reduction), 1/T (the in-place ``scores /= T`` at :211) and the incoming
    grad_output (AMP loss scale).
    """
    G = np.exp(S - lse[:, None])
    G[np.arange(S.shape[0]), np.asarray(labels, dtype=np.int64)] -= 1.0
    return G * scale


def training_step_global(Q, C, labels, mask_cols, temperature=1.0, dtype=np.float64, grad_output=1.0):
    """Single-process restatement of dpr_task.py:153-214 + backward (the non-DDP branch, or the DDP branch
    seen from outside: every rank computes exactly this on the gathered tensors, :181-195).

    Returns dict(loss, S (after /T), lse, row_loss, dQ, dC, G).
    """
    S = sim_score(Q, C, mask_cols, dtype) / dtype(temperature)  # :209-211
    loss, row_loss, lse = log_softmax_ce(S, labels)  # :212
    Nq = S.shape[0]
    G = dscores(S, labels, lse, grad_output / (Nq * temperature))
    dQ = G @ np.asarray(C, dtype=dtype)
    dC = G.T @ np.asarray(Q, dtype=dtype)
    return dict(loss=loss, S=S, lse=lse, row_loss=row_loss, dQ=dQ, dC=dC, G=G)


def gathered_labels(pos_idx_per_rank, ctx_per_rank):
    """dpr_task.py:177-190 -- all_labels[i] += i * (contexts on each earlier rank); flattened at :194.

    ``pos_idx_per_rank``: [W, B] local positive indices; every rank holds ``ctx_per_rank`` contexts (the
    reference assumes equal counts, comment at :168).
    """
    p = np.asarray(pos_idx_per_rank, dtype=np.int64)
    off = (np.arange(p.shape[0], dtype=np.int64) * ctx_per_rank)[:, None]
    return (p + off).reshape(-1)


def training_step_rank(q_r, C_all, labels_r_global, mask_cols, temperature, Nq_global, dtype=np.float64,
                       grad_output=1.0):
    """The local-rows formulation one MI355X rank computes (identity proven there):

    rank r scores only its own B rows against all Nc columns; sum_r(loss_sum_r)/Nq is the reference loss,
    dq_r is the reference's q.grad on rank r (cat backward keeps slot r, dpr_task.py:187,193) and
    sum_r(dC_part_r)[r*B*K:(r+1)*B*K] is the reference's c.grad on rank r (:188,192).
    """
    S = sim_score(q_r, C_all, mask_cols, dtype) / dtype(temperature)
    _, row_loss, lse = log_softmax_ce(S, labels_r_global)
    G = dscores(S, labels_r_global, lse, grad_output / (Nq_global * temperature))
    dq = G @ np.asarray(C_all, dtype=dtype)
    dC_part = G.T @ np.asarray(q_r, dtype=dtype)
    return dict(loss_sum=row_loss.sum(), S=S, lse=lse, row_loss=row_loss, dq=dq, dC_part=dC_part, G=G)


def rank_of_gold(S, labels):
    """dpr_task.py:235-246 restated without the sort: position (1-based) of column y_i in the descending
    order of row i.  Tie rule frozen by SURVEY.md section 8(a11): stable order, lower column index first,
    i.e. rank = 1 + #{j : S_ij > S_iy} + #{j < y : S_ij == S_iy}  (== torch.sort(..., stable=True)).
    """
    S = np.asarray(S)
    y = np.asarray(labels, dtype=np.int64)
    gold = S[np.arange(S.shape[0]), y][:, None]
    cols = np.arange(S.shape[1])[None, :]
    greater = (S > gold).sum(axis=1)
    ties_before = ((S == gold) & (cols < y[:, None])).sum(axis=1)
    return (1 + greater + ties_before).astype(np.int64)


def rank_metrics(S, labels, k=1):
    """dpr_task.py:238-246 -- returns (sum of ranks, sum of reciprocal ranks, #rows with rank <= k)."""
    r = rank_of_gold(S, labels)
    return int(r.sum()), float((1.0 / r).sum()), int((r - 1 < k).sum())


def topk_stable(S, k):
    """run_retrieval_pytorch.py:149-150 (torch.topk) with the same frozen tie rule: descending by score,
    ties by lower column index.  Returns (values, indices)."""
    S = np.asarray(S)
    idx = np.argsort(-S, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(S, idx, axis=1), idx.astype(np.int64)


def topk_merge(state, S, col_offset, k):
    """The shard loop of run_retrieval_pytorch.py:196-243 with the re-merge of :272-277 restated as a fold:
    state = (values [rows,k], ids [rows,k]) of everything scored so far (or None), S = scores of the next
    piece whose columns carry the ids col_offset + j.  Same frozen order (score desc, id asc); unfilled slots
    are (-inf, -1).  Folding pieces in increasing id order equals topk_stable of the concatenation."""
    S = np.asarray(S, dtype=np.float32)
    rows = S.shape[0]
    ids = np.broadcast_to(col_offset + np.arange(S.shape[1], dtype=np.int64), S.shape)
    if state is not None:
        v0, i0 = state
        keep = np.asarray(i0) >= 0
        S = np.concatenate([np.where(keep, v0, -np.inf).astype(np.float32), S], axis=1)
        ids = np.concatenate([np.where(keep, i0, np.iinfo(np.int64).max), ids], axis=1)
    order = np.lexsort((ids, -S), axis=1)[:, :k]  # primary key -S (desc score), secondary id asc
    v = np.take_along_axis(S, order, axis=1)
    i = np.take_along_axis(ids, order, axis=1)
    if v.shape[1] < k:
        pad = k - v.shape[1]
        v = np.concatenate([v, np.full((rows, pad), -np.inf, np.float32)], axis=1)
        i = np.concatenate([i, np.full((rows, pad), -1, np.int64)], axis=1)
    i = np.where(i == np.iinfo(np.int64).max, -1, i)
    return v.astype(np.float32), i.astype(np.int64)


def non_inbatch_query_ctx_mask(pos_idx, ctx_mask, n_queries):
    """dpr_task.py:198-207 -- in_batch_negatives=False: query i sees only its own K contexts."""
    m = np.asarray(ctx_mask, dtype=bool)
    K = int(m.shape[0] / n_queries)
    out = np.ones((n_queries, m.shape[0]), dtype=bool)
    for i, p in enumerate(np.asarray(pos_idx, dtype=np.int64)):
        out[i, p:p + K] = m[p:p + K]
    return out


# ------------------------------------------------------------------------------------------------------
# Seeded synthetic inputs shared by make_golden.py, the tests and bench.py.
# numpy's PCG64 stream is stable across versions, so the GPU box regenerates identical inputs.
# ------------------------------------------------------------------------------------------------------
def synth_embeddings(seed, B, K, d, dist="U", ragged_mask=False):
    """Per-rank synthetic (q [B,d], c [B*K,d], pos_idx [B], ctx_mask [B*K]), bf16-representable fp32.

    dist "U": unit-logit -- q,c ~ N(0,1)*d^-1/4 so S_ij ~ N(0,1); positives c_pos = (4/sqrt(d))*q + noise,
              i.e. S_pos ~ 4 + N(0,1): every one of the Nc terms matters in the softmax and the gradient
              is far from saturation (the numerically hardest regime).
    dist "P": peaky/DPR-like -- q,c ~ N(0,1) so S ~ N(0,d).
    pos_idx[i] = i*K (dpr_transform.py:164-166); ragged_mask marks ~5% of non-positive columns as padded
    dummy contexts (dpr_transform.py:143-157).
    """
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((B, d), dtype=np.float32)
    c = rng.standard_normal((B * K, d), dtype=np.float32)
    if dist == "U":
        s = np.float32(d ** -0.25)
        q *= s
        c *= s
        pos = np.arange(B) * K
        c[pos] = np.float32(4.0 / np.sqrt(d)) * q + c[pos]
    pos_idx = (np.arange(B, dtype=np.int64) * K)
    mask = np.zeros(B * K, dtype=bool)
    if ragged_mask:
        mask = rng.random(B * K) < 0.05
        mask[pos_idx] = False
    return bf16_round(q), bf16_round(c), pos_idx, mask


def synth_search(seed, nq, n, d):
    """Retrieval inputs whose inner products are EXACT in fp16, in bf16-input MFMA arithmetic and in fp32 alike (entries in
    {-1, -0.5, 0, 0.5, 1}: every score is a multiple of 0.25 below 2^6), so the reference's fp16 scores
    (run_retrieval_pytorch.py:149) and the fp32 scores of the MI355X path are the same numbers, and only the order inside a tie
    class is left to the implementation (torch.topk documents none; the frozen rule here is lower id first)."""
    rng = np.random.default_rng(seed)
    q = rng.integers(-2, 3, (nq, d)).astype(np.float32) / 2
    c = rng.integers(-2, 3, (n, d)).astype(np.float32) / 2
    return q, c
