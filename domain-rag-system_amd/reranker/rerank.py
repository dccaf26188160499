"""Second-stage rerankers.

Drop-in for the reference's ``src/reranker/rerank.py``: ``BaseReranker.rerank`` takes the
first ``top_n`` results, scores ``(query, text)`` pairs, adds ``rerank_score`` and appends
``+rerank`` to ``fusion_method`` on copies, orders them by ``rerank_score`` descending with
ties in pool order, and returns ``top_k``.  The ordering runs in ``rag_topk_segmented``.

``SimpleSimilarityReranker`` scores with the bi-encoder's cosine similarity,
``CrossEncoderReranker`` with a one-label BERT cross-encoder (PyTorch-ROCm forward, batches
of 32 pairs).  ``build_reranker`` raises when the cross-encoder cannot be loaded instead of
quietly returning the other reranker.
"""
from typing import Dict, Iterable, List, Literal, Optional, Tuple

import numpy as np
import torch

import rag_hip

try:
    from indexing.embedder import MedicalEmbedder
    from indexing import bert
except ImportError:                     # pragma: no cover - depends on sys.path layout
    from ..indexing.embedder import MedicalEmbedder
    from ..indexing import bert


class BaseReranker:
    def __init__(self, top_n: int = 50, name: str = "base"):
        self.top_n = top_n
        self.name = name

    def score_pairs(self, pairs: Iterable[Tuple[str, str]]) -> List[float]:
        raise NotImplementedError

    def rerank(self, query: str, results: List[Dict], top_k: int = 10) -> List[Dict]:
        if not results:
            return results
        pool = results[: min(self.top_n, len(results))]
        scores = self.score_pairs([(query, c.get("text", "") or "") for c in pool])
        rescored = []
        for c, s in zip(pool, scores):
            r = c.copy()
            r["rerank_score"] = float(s)
            r["fusion_method"] = (r.get("fusion_method") or "pre") + "+rerank"
            rescored.append(r)
        if not rescored or top_k <= 0:
            return []
        order = self._order([r["rerank_score"] for r in rescored], top_k)
        return [rescored[i] for i in order]

    @staticmethod
    def _order(scores: List[float], top_k: int) -> List[int]:
        """Positions of the best ``top_k`` scores, best first, ties in input order."""
        n = len(scores)
        k = min(top_k, rag_hip.MAX_K)
        s32 = np.asarray(scores, dtype=np.float32)
        if np.array_equal(s32.astype(np.float64), np.asarray(scores, dtype=np.float64)):
            idx, cnt = rag_hip.topk_segmented(s32, [0, n], k)
            return [int(i) for i in idx[0, :int(cnt[0])]]
        # scores that are not float32 values are ordered on their fp64 values
        return rag_hip.topk_stable_f64(scores, k)


class SimpleSimilarityReranker(BaseReranker):
    """Cosine similarity between the query embedding and each pool text's embedding."""

    def __init__(self, embedder: Optional[MedicalEmbedder] = None, top_n: int = 50):
        super().__init__(top_n=top_n, name="simple-sim")
        self.embedder = embedder or MedicalEmbedder()
        self.dim = self.embedder.embedding_dim

    def score_pairs(self, pairs: Iterable[Tuple[str, str]]) -> List[float]:
        pairs = list(pairs)
        if not pairs:
            return []
        e = self.embedder
        seqs = [e.tokenize(pairs[0][0])] + [e.tokenize(t) for _, t in pairs]
        emb = e.encode_tokens(seqs, normalize=True)
        q, d = emb[0], emb[1:]
        num = d @ q
        den = torch.linalg.vector_norm(d, dim=1) * torch.linalg.vector_norm(q)
        cos = torch.where(den == 0, torch.zeros_like(num), num / den)
        return [float(x) for x in cos.cpu()]


class CrossEncoderReranker(BaseReranker):
    def __init__(self, model_name: str = "cross-encoder/ms-marco-MiniLM-L-6-v2", top_n: int = 50,
                 device: str = None, dtype: torch.dtype = torch.float32, batch_size: int = 32,
                 max_length: int = 512):
        super().__init__(top_n=top_n, name="cross-encoder")
        self.device = device or ('cuda' if torch.cuda.is_available() else 'cpu')
        # float32 on the GPU: GEMMs and attention as split-fp16 matrix-core products (fp32 round-off class), the
        # reference's precision (CrossEncoder.predict, reference src/reranker/rerank.py:79-84) at matrix-core speed
        on_gpu = torch.device(self.device).type == 'cuda'
        self.model, self.tokenizer, cfg = bert.load_model(model_name, 'cross', self.device, dtype=dtype, native='auto',
                                                          split_gemm=on_gpu and dtype == torch.float32,
                                                          split_dtype=torch.float16)
        self.batch_size = batch_size
        self.max_length = min(max_length, cfg.max_position_embeddings)

    def encode_pair(self, a: str, b: str):
        """[CLS] a [SEP] b [SEP]; the second text is cut so the pair fits ``max_length``."""
        t = self.tokenizer
        ia = t.encode(a)[: max(self.max_length - 3, 0)]
        ib = t.encode(b)[: max(self.max_length - 3 - len(ia), 0)]
        ids = [t.cls_id] + ia + [t.sep_id] + ib + [t.sep_id]
        types = [0] * (len(ia) + 2) + [1] * (len(ib) + 1)
        return ids, types

    @torch.no_grad()
    def score_pairs(self, pairs: Iterable[Tuple[str, str]]) -> List[float]:
        pairs = list(pairs)
        if not pairs:
            return []
        enc = [self.encode_pair(a, b) for a, b in pairs]
        if self.model.bert.native:
            # one unpadded token stream (a query's pool is at most top_n pairs of <= max_length tokens)
            import numpy as np
            lens = np.array([len(e[0]) for e in enc], dtype=np.int64)
            cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            dev = self.device
            ids = torch.tensor([t for e in enc for t in e[0]], dtype=torch.int32, device=dev)
            types = torch.tensor([t for e in enc for t in e[1]], dtype=torch.int32, device=dev)
            pos = torch.from_numpy(np.concatenate([np.arange(l, dtype=np.int32) for l in lens])).to(dev)
            logits = self.model.forward_packed(ids, types, pos, torch.from_numpy(cu).to(dev), int(lens.max()))
            return [float(v) for v in logits.reshape(len(enc), -1)[:, 0].cpu().tolist()]
        order = sorted(range(len(enc)), key=lambda i: -len(enc[i][0]))
        out = [0.0] * len(enc)
        for s in range(0, len(enc), self.batch_size):
            idx = order[s:s + self.batch_size]
            ids, mask = bert.pad_batch([enc[i][0] for i in idx], self.tokenizer.pad_id, self.device)
            types, _ = bert.pad_batch([enc[i][1] for i in idx], 0, self.device)
            logits = self.model(ids, types, mask)
            for i, v in zip(idx, logits.reshape(len(idx), -1)[:, 0].cpu().tolist()):
                out[i] = float(v)
        return out


def build_reranker(kind: Literal["none", "simple", "crossencoder"] = "simple", top_n: int = 50,
                   cross_model: str = "cross-encoder/ms-marco-MiniLM-L-6-v2",
                   embedder: Optional[MedicalEmbedder] = None) -> Optional[BaseReranker]:
    if kind == "none":
        return None
    if kind == "crossencoder":
        return CrossEncoderReranker(model_name=cross_model, top_n=top_n)
    return SimpleSimilarityReranker(embedder=embedder, top_n=top_n)
