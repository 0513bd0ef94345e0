"""The encoder towers (they stay on PyTorch-ROCm, north_star): a HuggingFace transformer, CLS pooling, optional
projection head.  Interface of the reference's `HFEncoder` (dpr_scale/models/hf_model.py:12-41): constructor
`(model_path, dropout, projection_dim)`, `forward(tokens: dict) -> [B, d]`, sub-module names `transformer` /
`project` (they are checkpoint keys: `query_encoder.transformer...`).  This file only produces the q / c matrices
the hot path consumes.

`model_path`: a directory / hub id (loaded with AutoModel.from_pretrained, as the reference does), or -- the build
image has neither network nor weights -- a dict of BertConfig fields for a random-init encoder of that architecture
(what bench_e2e.py and the tests use).
"""
from typing import Optional, Union

import torch.nn as nn


def _with_dropout(config, p):
    for field in ("hidden_dropout_prob", "attention_probs_dropout_prob"):
        setattr(config, field, p)
    return config


def _backbone(model_path, dropout):
    import transformers as tf

    if isinstance(model_path, dict):  # architecture only: random weights
        config = _with_dropout(tf.BertConfig(**model_path), dropout)
        return tf.BertModel(config, add_pooling_layer=False), config
    config = _with_dropout(tf.AutoConfig.from_pretrained(model_path), dropout)
    return tf.AutoModel.from_pretrained(model_path, config=config), config


def _head(hidden_size, projection_dim):
    """None / 0: no head.  -1: a head of the encoder's own width (hf_model.py:30-31).  Linear (N(0, 0.02) weights) + LayerNorm."""
    if not projection_dim:
        return nn.Identity()
    width = hidden_size if projection_dim == -1 else projection_dim
    dense = nn.Linear(hidden_size, width)
    nn.init.normal_(dense.weight, mean=0.0, std=0.02)
    return nn.Sequential(dense, nn.LayerNorm(width))


class HFEncoder(nn.Module):
    def __init__(self, model_path: Union[str, dict] = "roberta-base", dropout: float = 0.1,
                 projection_dim: Optional[int] = None):
        super().__init__()
        self.transformer, config = _backbone(model_path, dropout)
        self.project = _head(config.hidden_size, projection_dim)

    def forward(self, tokens):
        states = self.transformer(**tokens)[0]      # [B, T, C] last layer
        cls = states.select(1, 0)                   # the first token's vector
        return self.project(cls).clone()
