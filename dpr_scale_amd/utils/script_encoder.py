"""TorchScript packaging of an encoder, so that DenseRetrieverTask.to_torchscript works without the reference package.

Contract of the reference (dpr_scale/utils/utils.py:94-121, used by dpr_task.py:325-368): a scripted module whose
``forward(texts: List[str])`` runs the task's text transform on ``{"text": texts}`` and feeds ``batch["token_ids"]`` to a CPU copy
of the encoder (optionally with its Linear layers dynamically quantised to int8); ``encode(token_ids)`` skips the transform.
Encoder packaging is outside the hot path: plain PyTorch.  As with the reference, both the transform
and the encoder must themselves be scriptable (the reference's recipes script its pytext-style encoders; a HuggingFace tower is
scriptable by neither implementation).
"""
import copy
from typing import Any, Dict, List

import torch


class TextToBatch(torch.nn.Module):
    """texts -> the transform's batch dict (the transform sees {"text": texts}, as in the data modules)."""

    def __init__(self, transform):
        super().__init__()
        self.transform = transform

    def forward(self, texts: List[str]) -> Dict[str, torch.Tensor]:
        batch: Dict[str, Any] = {"text": texts}
        return self.transform(batch)


class ScriptEncoder(torch.nn.Module):
    def __init__(self, transform, encoder, quantize: bool = False):
        super().__init__()
        self.transform = TextToBatch(transform)
        enc = copy.deepcopy(encoder).cpu()
        if quantize:
            enc = torch.quantization.quantize_dynamic(enc, {torch.nn.Linear}, dtype=torch.qint8)
        self.encoder = enc
        self.cpu()

    def forward(self, texts: List[str]) -> torch.Tensor:
        return self.encode(self.transform(texts)["token_ids"])

    def encode(self, model_inputs: torch.Tensor) -> torch.Tensor:
        return self.encoder(model_inputs)
