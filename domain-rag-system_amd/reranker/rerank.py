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
    """Cosine similarity between the query embedding and each pool text's embedding (reference
    ``src/reranker/rerank.py:46-68``, which embeds the query and every pool text per call).

    With ``vector_indexer`` (a cosine collection built with the same embedder) the pool texts are not embedded
    again: a text the collection holds already has its embedding resident in HBM, so its score is one exact
    float32 dot product with the (normalised) query embedding (``rag_dense_rescore``).  Only the query, and any
    text the collection does not hold, go through the encoder."""

    def __init__(self, embedder: Optional[MedicalEmbedder] = None, top_n: int = 50, vector_indexer=None):
        super().__init__(top_n=top_n, name="simple-sim")
        self.embedder = embedder or MedicalEmbedder()
        self.dim = self.embedder.embedding_dim
        self.vector_indexer = vector_indexer
        if vector_indexer is not None and (vector_indexer.distance_metric != 'cosine'
                                           or vector_indexer.embedder is not self.embedder):
            raise ValueError("the resident-row route needs a cosine collection built with this reranker's embedder")

    def score_pairs(self, pairs: Iterable[Tuple[str, str]]) -> List[float]:
        pairs = list(pairs)
        if not pairs:
            return []
        if self.vector_indexer is not None:
            return self._score_from_rows(pairs)
        return self._score_by_embedding(pairs[0][0], [t for _, t in pairs])

    def _score_from_rows(self, pairs):
        vi = self.vector_indexer
        texts = [t for _, t in pairs]
        vi._attach()
        if not getattr(vi, 'rows_are_embeddings', False):
            # the collection's vectors came from outside (add_points) or from another model: they are not known to be
            # this embedder's embeddings of the texts, and the reference's score is cosine(encode(q), encode(text))
            return self._score_by_embedding(pairs[0][0], texts)
        rows = vi.rows_of_texts(texts)
        out = [0.0] * len(pairs)
        held = [i for i, r in enumerate(rows) if r >= 0]
        foreign = [i for i, r in enumerate(rows) if r < 0]
        if held:
            q = self.embedder.encode(pairs[0][0])[0]
            for i, s in zip(held, vi.cosine_to_rows(q, [rows[i] for i in held])):
                out[i] = float(s)
        if foreign:
            for i, s in zip(foreign, self._score_by_embedding(pairs[0][0], [texts[i] for i in foreign])):
                out[i] = s
        return out

    def _score_by_embedding(self, query: str, texts: List[str]) -> List[float]:
        pairs = [(query, t) for t in texts]
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
        self._graphs = bert.GraphCache()          # HIP graphs of the forward for a query's pool

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
            cu_t = torch.from_numpy(cu).to(dev)
            logits = None
            if len(enc) <= 64 and ids.shape[0] <= 16384 and self.model.bert.split_gemm:
                # one query's pool is launch bound: replay the forward from a HIP graph
                logits = bert.forward_packed_graphed(self.model, self._graphs, ids, types, pos, cu_t, int(lens.max()))
            if logits is None:
                logits = self.model.forward_packed(ids, types, pos, cu_t, int(lens.max()))
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
                   embedder: Optional[MedicalEmbedder] = None, vector_indexer=None) -> Optional[BaseReranker]:
    """The reference's factory (``src/reranker/rerank.py:87-101``) plus ``vector_indexer``: given a cosine
    collection built with ``embedder``, the simple reranker scores pool texts from its resident rows."""
    if kind == "none":
        return None
    if kind == "crossencoder":
        return CrossEncoderReranker(model_name=cross_model, top_n=top_n)
    if vector_indexer is not None and (vector_indexer.distance_metric != 'cosine'
                                       or (embedder is not None and vector_indexer.embedder is not embedder)):
        vector_indexer = None
    return SimpleSimilarityReranker(embedder=embedder or (vector_indexer.embedder if vector_indexer else None),
                                    top_n=top_n, vector_indexer=vector_indexer)
