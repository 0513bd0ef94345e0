"""Synthetic NQ-shaped batches in the exact dict layout DPRTransform emits (dpr_transform.py:181-187):
query_ids / contexts_ids (token dicts), pos_ctx_indices [B] = i*K (:164-166), ctx_mask [B*K] bool (:143-157),
scores.  Replaces the reference's mmap JSONL datamodule for measurement and plumbing tests (no datasets offline).
"""
import numpy as np
import torch


def unit_logit_embeddings(seed, B, K, d, ragged=False):
    """Embedding-level synthetic step inputs (distribution "U"): q [B,d], c [B*K,d] ~
    N(0,1)*d^(-1/4) so that logits are ~N(0,1), positives c[i*K] = (4/sqrt(d))*q[i] + noise (every column matters
    in the softmax), values rounded to bf16-representable fp32; pos_idx[i] = i*K (dpr_transform.py:164-166);
    `ragged` marks ~5 % of the non-positive columns as padded dummy contexts (dpr_transform.py:143-157).
    Returns numpy (q, c, pos_idx int64, ctx_mask bool)."""
    rng = np.random.default_rng(seed)
    s = np.float32(d ** -0.25)
    q = rng.standard_normal((B, d), dtype=np.float32) * s
    c = rng.standard_normal((B * K, d), dtype=np.float32) * s
    pos = np.arange(B, dtype=np.int64) * K
    c[pos] += np.float32(4.0 / np.sqrt(d)) * q
    mask = np.zeros(B * K, dtype=bool)
    if ragged:
        mask = rng.random(B * K) < 0.05
        mask[pos] = False

    def to_bf16_grid(x):  # round-to-nearest-even onto the bf16 grid, kept as fp32
        u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)

    return to_bf16_grid(q), to_bf16_grid(c), pos, mask


class SyntheticDPRDataModule:
    def __init__(self, batch_size=4, num_negative=1, num_val_negative=1, seq_len=16, vocab_size=32, n_train_batches=4,
                 n_val_batches=2, ragged=False, seed=0, transform=None, **_):
        self.B, self.K, self.Kv = batch_size, 1 + num_negative, 1 + num_val_negative
        self.T, self.V = seq_len, vocab_size
        self.n_train, self.n_val, self.ragged, self.seed = n_train_batches, n_val_batches, ragged, seed

    def _tokens(self, g, rows):
        ids = torch.randint(5, self.V, (rows, self.T), generator=g)
        ids[:, 0] = 3 if self.V <= 64 else 101
        ids[:, -1] = 4 if self.V <= 64 else 102
        return {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": torch.ones_like(ids)}

    def _batch(self, g, K):
        B = self.B
        mask = torch.zeros(B * K, dtype=torch.bool)
        if self.ragged and K > 1:
            mask = torch.rand(B * K, generator=g) < 0.2
            mask[torch.arange(B) * K] = False
        return {"query_ids": self._tokens(g, B), "contexts_ids": self._tokens(g, B * K),
                "pos_ctx_indices": torch.arange(B, dtype=torch.long) * K, "scores": torch.zeros(B, K),
                "ctx_mask": mask}

    def _loader(self, n, K, offset):
        g = torch.Generator().manual_seed(self.seed + offset)
        return [self._batch(g, K) for _ in range(n)]

    def train_dataloader(self):
        return self._loader(self.n_train, self.K, 0)

    def val_dataloader(self):
        return self._loader(self.n_val, self.Kv, 1000)

    def test_dataloader(self):
        return self._loader(self.n_val, self.Kv, 2000)
