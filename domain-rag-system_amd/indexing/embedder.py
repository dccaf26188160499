"""Bi-encoder text embedder (PyTorch-ROCm forward pass).

Drop-in for the reference ``MedicalEmbedder`` (reference ``src/indexing/embedder.py:8-97``):
same constructor arguments, ``encode`` returns a C-contiguous float32 ``[n, dim]`` array
(L2-normalised by default), ``encode_chunks`` and ``get_model_info`` keep their shapes.
The sentence-transformers pipeline it replaces is restated in ``bert.BiEncoder``: sort the
texts by length, batches of ``batch_size`` padded to the longest member, truncation at
``max_length`` tokens, mean pooling over the attention mask, optional normalisation, output
in input order.

Differences on purpose: a model that cannot be loaded raises (the reference silently
switches to another model), and there is no progress bar.
"""
from typing import Dict, List, Union

import numpy as np
import torch

from . import bert


class MedicalEmbedder:
    def __init__(self, model_name: str = 'pritamdeka/S-PubMedBert-MS-MARCO', batch_size: int = 32,
                 max_length: int = 512, device: str = None, dtype: torch.dtype = torch.float32):
        self.model_name = model_name
        self.batch_size = batch_size
        self.max_length = max_length
        self.device = device or ('cuda' if torch.cuda.is_available() else 'cpu')
        self.model, self.tokenizer, cfg = bert.load_model(model_name, 'bi', self.device, dtype=dtype, native='auto')
        self.max_length = min(max_length, cfg.max_position_embeddings)
        self.embedding_dim = cfg.hidden_size

    def tokenize(self, text: str) -> List[int]:
        """[CLS] tokens [SEP], truncated to ``max_length``."""
        t = self.tokenizer
        ids = t.encode(text)[: max(self.max_length - 2, 0)]
        return [t.cls_id] + ids + [t.sep_id]

    @torch.no_grad()
    def encode_tokens(self, seqs: List[List[int]], normalize: bool = True) -> torch.Tensor:
        """Token id lists -> float32 ``[n, dim]`` tensor on the model's device, input order."""
        n = len(seqs)
        out = torch.empty((n, self.embedding_dim), dtype=torch.float32, device=self.device)
        order = sorted(range(n), key=lambda i: -len(seqs[i]))
        for s in range(0, n, self.batch_size):
            idx = order[s:s + self.batch_size]
            ids, mask = bert.pad_batch([seqs[i] for i in idx], self.tokenizer.pad_id, self.device)
            emb = self.model(ids, None, mask, normalize=normalize)
            out[torch.tensor(idx, device=self.device)] = emb
        return out

    def encode(self, texts: Union[str, List[str]], normalize: bool = True) -> np.ndarray:
        if isinstance(texts, str):
            texts = [texts]
        emb = self.encode_tokens([self.tokenize(t) for t in texts], normalize=normalize)
        return np.ascontiguousarray(emb.cpu().numpy(), dtype=np.float32)

    def encode_chunks(self, chunks: List[Dict], text_key: str = 'text') -> np.ndarray:
        return self.encode([c[text_key] if isinstance(c, dict) else c.text for c in chunks])

    def get_model_info(self) -> Dict:
        return {'model_name': self.model_name, 'embedding_dim': self.embedding_dim,
                'max_length': self.max_length, 'device': self.device}
