"""Scriptable stand-ins for the text transform and the encoder (TorchScript needs module-level classes with source)."""
from typing import Any, Dict, List

import torch


class ToyTransform(torch.nn.Module):
    """{"text": [...]} -> {"token_ids": int64 [n, 4]}: byte values of the first four characters (0-padded)."""

    def forward(self, batch: Dict[str, Any]) -> Dict[str, torch.Tensor]:
        texts = batch["text"]
        assert torch.jit.isinstance(texts, List[str])
        ids = torch.zeros((len(texts), 4), dtype=torch.long)
        for i, s in enumerate(texts):
            for j in range(min(len(s), 4)):
                ids[i, j] = ord(s[j]) % 32
        return {"token_ids": ids}


class ToyEncoder(torch.nn.Module):
    def __init__(self, vocab: int = 32, dim: int = 8):
        super().__init__()
        self.emb = torch.nn.Embedding(vocab, dim)
        self.proj = torch.nn.Linear(dim, dim)

    def forward(self, token_ids: torch.Tensor) -> torch.Tensor:
        return self.proj(self.emb(token_ids).mean(1))
