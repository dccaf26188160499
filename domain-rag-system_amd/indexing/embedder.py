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
                 max_length: int = 512, device: str = None, dtype: torch.dtype = torch.float32,
                 split_gemm: bool = True, tokens_per_batch: int = 1 << 16):
        """The first four arguments are the reference's.  On a GPU whose kernels cover the model
        (head size 32 or 64) texts are embedded as unpadded token streams of up to
        ``tokens_per_batch`` tokens (``batch_size`` then plays no role); with float32 weights
        and ``split_gemm`` the GEMMs run as split-bf16 products with fp32 accumulation (cosine
        scores within 1e-5 of the plain fp32 forward; pass False for plain fp32 GEMMs)."""
        self.model_name = model_name
        self.batch_size = batch_size
        self.max_length = max_length
        self.tokens_per_batch = int(tokens_per_batch)
        self.device = device or ('cuda' if torch.cuda.is_available() else 'cpu')
        self.model, self.tokenizer, cfg = bert.load_model(model_name, 'bi', self.device, dtype=dtype, native='auto',
                                                          split_gemm=split_gemm and dtype == torch.float32)
        self.max_length = min(max_length, cfg.max_position_embeddings)
        self.embedding_dim = cfg.hidden_size
        self._graphs = bert.GraphCache()          # HIP graphs of the forward for one or two texts (a query at a time)
        self._recent = {}                         # embeddings of the last few single texts (a query is embedded for
        #                                           the search and again by the simple reranker)

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
        if self.model.bert.native and n:
            return self._encode_packed(seqs, out, normalize)
        order = sorted(range(n), key=lambda i: -len(seqs[i]))
        for s in range(0, n, self.batch_size):
            idx = order[s:s + self.batch_size]
            ids, mask = bert.pad_batch([seqs[i] for i in idx], self.tokenizer.pad_id, self.device)
            emb = self.model(ids, None, mask, normalize=normalize)
            out[torch.tensor(idx, device=self.device)] = emb
        return out

    def _encode_packed(self, seqs, out, normalize):
        """Unpadded batches: texts in input order, cut into streams of at most tokens_per_batch
        tokens (an empty text counts as its two special tokens, which tokenize always emits)."""
        lens = np.array([len(s) for s in seqs], dtype=np.int64)
        cu_all = np.concatenate([[0], np.cumsum(lens)])
        p0, n = 0, len(seqs)
        while p0 < n:
            p1 = int(np.searchsorted(cu_all, cu_all[p0] + self.tokens_per_batch, side='right')) - 1
            p1 = min(max(p1, p0 + 1), n)
            flat = np.fromiter((t for s in seqs[p0:p1] for t in s), dtype=np.int32, count=int(cu_all[p1] - cu_all[p0]))
            pos = np.concatenate([np.arange(l, dtype=np.int32) for l in lens[p0:p1]]) if flat.size else flat
            cu = (cu_all[p0:p1 + 1] - cu_all[p0]).astype(np.int32)
            dev = self.device
            ids_t = torch.from_numpy(flat).to(dev)
            longest = max(int(lens[p0:p1].max()), 1)
            emb = None
            if p1 - p0 <= 2 and flat.size <= 1024 and self.model.bert.split_gemm:
                # a query at a time is launch bound: replay the forward from a HIP graph
                emb = bert.forward_packed_graphed(self.model, self._graphs, ids_t, torch.zeros_like(ids_t),
                                                  torch.from_numpy(pos).to(dev), torch.from_numpy(cu).to(dev), longest,
                                                  normalize=normalize)
            if emb is None:
                emb = self.model.forward_packed(ids_t, torch.zeros_like(ids_t), torch.from_numpy(pos).to(dev),
                                                torch.from_numpy(cu).to(dev), longest, normalize=normalize)
            out[p0:p1] = emb
            p0 = p1
        return out

    def encode(self, texts: Union[str, List[str]], normalize: bool = True) -> np.ndarray:
        single = isinstance(texts, str)
        if single:
            hit = self._recent.get((texts, bool(normalize)))
            if hit is not None:
                return hit.copy()
            texts = [texts]
        emb = self.encode_tokens([self.tokenize(t) for t in texts], normalize=normalize)
        out = np.ascontiguousarray(emb.cpu().numpy(), dtype=np.float32)
        if single:
            if len(self._recent) >= 16:
                self._recent.pop(next(iter(self._recent)))
            self._recent[(texts[0], bool(normalize))] = out.copy()
        return out

    def encode_chunks(self, chunks: List[Dict], text_key: str = 'text') -> np.ndarray:
        return self.encode([c[text_key] if isinstance(c, dict) else c.text for c in chunks])

    def get_model_info(self) -> Dict:
        return {'model_name': self.model_name, 'embedding_dim': self.embedding_dim,
                'max_length': self.max_length, 'device': self.device}
