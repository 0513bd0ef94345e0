"""Encoder wrapper with the reference's interface (dpr_scale/models/hf_model.py:12-41): a HuggingFace encoder,
CLS pooling `last_layer[:, 0, :]`, optional Linear + LayerNorm projection, `forward(tokens: dict) -> [B, d]`.
The towers stay on PyTorch-ROCm (north_star); this file only produces the q / c matrices the hot path consumes.

`model_path` may be a directory / hub id (AutoModel.from_pretrained, as in the reference) or -- because the
build image has no network and no weights -- a dict of BertConfig fields, giving a random-init encoder of that
architecture (what bench_e2e.py and the tests use).
"""
from typing import Optional

import torch.nn as nn


class HFEncoder(nn.Module):
    def __init__(self, model_path="roberta-base", dropout: float = 0.1, projection_dim: Optional[int] = None):
        super().__init__()
        from transformers import AutoConfig, AutoModel, BertConfig, BertModel

        if isinstance(model_path, dict):
            cfg = BertConfig(**model_path)
            cfg.attention_probs_dropout_prob = dropout
            cfg.hidden_dropout_prob = dropout
            self.transformer = BertModel(cfg, add_pooling_layer=False)
        else:
            cfg = AutoConfig.from_pretrained(model_path)
            cfg.attention_probs_dropout_prob = dropout
            cfg.hidden_dropout_prob = dropout
            self.transformer = AutoModel.from_pretrained(model_path, config=cfg)
        self.project = nn.Identity()
        if projection_dim == -1:
            projection_dim = cfg.hidden_size
        if projection_dim:
            lin = nn.Linear(cfg.hidden_size, projection_dim)
            lin.weight.data.normal_(mean=0.0, std=0.02)
            self.project = nn.Sequential(lin, nn.LayerNorm(projection_dim))

    def forward(self, tokens):
        hidden = self.transformer(**tokens)[0]  # [B, T, C]
        return self.project(hidden[:, 0, :]).clone()
