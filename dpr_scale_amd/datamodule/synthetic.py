"""Synthetic NQ-shaped batches in the exact dict layout DPRTransform emits (dpr_transform.py:181-187):
query_ids / contexts_ids (token dicts), pos_ctx_indices [B] = i*K (:164-166), ctx_mask [B*K] bool (:143-157),
scores.  Replaces the reference's mmap JSONL datamodule for measurement and plumbing tests (no datasets offline).
"""
import torch


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
