"""Batched, device-resident form of the query path.

The reference answers one query per call (``scripts/query.py:139-162``: hybrid search with
``fetch_k = max(top_k, rerank_top_n)``, then rerank of the original query).  ``BatchSearcher``
runs the same sequence for a whole batch of queries without leaving the GPU between stages:

    embed (PyTorch)  ->  rag_dense_search_dev   \\
                         rag_bm25_search_dev     -> rag_fuse_rrf_dev / rag_fuse_weighted_dev
    -> rag_ce_pair_len_dev / rag_ce_pack_dev -> cross-encoder forward (PyTorch)
    -> rag_topk_segmented_dev

With ``dist`` set (one process per GPU, corpus rows sharded), each rank searches its own row
range, the per-shard top-K lists are all-gathered (RCCL) and merged with ``rag_merge_topk_*``,
fusion is replicated, the cross-encoder pairs are split across ranks by pair index and the
logits all-gathered.  Global statistics (idf, avgdl, canon) are replicated so every score
equals the unsharded one.
"""
from dataclasses import dataclass
from typing import Optional

import contextlib
import os

import numpy as np
import torch

import rag_hip
from .bert import GraphCache as _GraphCache


@dataclass
class StageTimes:
    embed: float = 0.0
    dense: float = 0.0
    bm25: float = 0.0
    fuse: float = 0.0
    rerank: float = 0.0

    def as_dict(self):
        return dict(embed=self.embed, dense=self.dense, bm25=self.bm25, fuse=self.fuse, rerank=self.rerank)


class TokenStore:
    """CSR token ids of every chunk (global rows), resident on the GPU, for pair packing."""

    def __init__(self, tok: torch.Tensor, off: torch.Tensor, row_base: int = 0):
        assert tok.dtype == torch.int32 and off.dtype == torch.int64
        self.tok, self.off, self.row_base = tok.contiguous(), off.contiguous(), int(row_base)


def pack_static(input_ids, attention_mask):
    """BertEncoder.pack without a host round trip and with static shapes: padded [B, S] ids + left-aligned mask ->
    (ids, types, pos int32 [B * S], cu_seqlens int32 [B + 1]); the stream's unused tail holds token 0 at position 0."""
    b, s = input_ids.shape
    dev = input_ids.device
    t = b * s
    keep = attention_mask.bool()
    cu = torch.zeros((b + 1,), dtype=torch.int32, device=dev)
    cu[1:] = torch.cumsum(keep.sum(dim=1), 0)
    col = torch.arange(s, device=dev, dtype=torch.int64)
    dest = torch.where(keep, cu[:-1, None].long() + col[None, :], torch.full((), t, dtype=torch.int64, device=dev))
    flat = dest.reshape(-1)
    ids = torch.zeros((t + 1,), dtype=torch.int32, device=dev).scatter_(0, flat, input_ids.reshape(-1).to(torch.int32))
    pos = torch.zeros((t + 1,), dtype=torch.int32, device=dev).scatter_(0, flat, col.to(torch.int32).repeat(b))
    return ids[:t], torch.zeros((t,), dtype=torch.int32, device=dev), pos[:t], cu


class BatchSearcher:
    def __init__(self, dense: Optional[rag_hip.DenseIndex], bm25: Optional[rag_hip.Bm25Index],
                 canon: Optional[torch.Tensor] = None, bi_encoder=None, cross_encoder=None,
                 chunk_tokens: Optional[TokenStore] = None, top_k: int = 5, rerank_top_n: int = 50,
                 fusion_method: str = 'rrf', vector_weight: float = 0.5, bm25_weight: float = 0.5,
                 ce_pairs_per_batch: int = 8192, ce_max_len: int = 512, special_ids=(101, 102, 0), dist=None,
                 ce_packed=None, ce_tokens_per_batch: int = 1 << 20, ce_token_bucket: int = 1 << 17,
                 simple_rerank: bool = False, screen_encoder=None, screen_safety: float = 1.5,
                 screen_min_safety: float = 1.25, screen_min_pairs: int = 4096, screen_two_stage: bool = True):
        self.dense, self.bm25, self.canon = dense, bm25, canon
        self.bi_encoder, self.cross_encoder, self.chunk_tokens = bi_encoder, cross_encoder, chunk_tokens
        self.top_k, self.rerank_top_n = int(top_k), int(rerank_top_n)
        self.fusion_method, self.vw, self.bw = fusion_method, float(vector_weight), float(bm25_weight)
        self.ce_pairs_per_batch, self.ce_max_len = int(ce_pairs_per_batch), int(ce_max_len)
        # the reference's "simple" reranker (cosine between the query embedding and each pool text's embedding,
        # src/reranker/rerank.py:46-68) served from the resident rows: a pool entry's embedding IS its row
        self.simple_rerank = bool(simple_rerank) and cross_encoder is None
        if self.simple_rerank and (dense is None or getattr(dense, 'metric', None) not in (rag_hip.METRIC_COSINE, 'cosine')):
            raise ValueError("simple_rerank needs a cosine dense index (rows stored L2-normalised)")
        self._qvec = None
        # unpadded cross-encoder batches when the model can run them (native kernels; bfloat16 weights, or float32
        # weights with split-precision GEMMs: the reference's precision, src/reranker/rerank.py:79-84)
        self.ce_tokens_per_batch, self.ce_token_bucket = int(ce_tokens_per_batch), int(ce_token_bucket)
        self.ce_sort_by_length = True
        self.ce_streams = int(os.environ.get('RAG_CE_STREAMS', '1'))   # HIP streams the cross-encoder mini-batches alternate between
        self._ce_stream_pool = []
        # sharded: each rank embeds a slice of the queries and the slices are all-gathered
        # (RAG_SPLIT_EMBED=0: every rank embeds all of them)
        self.split_embed = False
        self.ce_packed = bool(ce_packed) if ce_packed is not None else (
            cross_encoder is not None and getattr(cross_encoder.bert, 'native', False)
            and (next(cross_encoder.parameters()).dtype in (torch.bfloat16, torch.float16)
                 or (next(cross_encoder.parameters()).dtype == torch.float32
                     and getattr(cross_encoder.bert, 'split_gemm', False))))
        # Screen-then-rescore rerank (see _score_screened): `screen_encoder` is a cheap copy of the cross-encoder (the same
        # weights in float16, one matrix-core product per product) that scores every pair; only the pairs whose cheap
        # logit lies within a margin of their query's cheap k-th best are scored by `cross_encoder`, and the final order
        # and rerank_score come from those exact logits alone.
        self.screen_encoder = screen_encoder if cross_encoder is not None else None
        self.screen_safety, self.screen_min_pairs = float(screen_safety), int(screen_min_pairs)
        self.screen_two_stage = bool(screen_two_stage)
        self.screen_min_safety = min(float(screen_min_safety), self.screen_safety)
        self._screen_err = None                      # running estimate of the screen's error (None: the next batch rescores every pair)
        self.screen_stats = {'pairs': 0, 'rescored': 0, 'widened': 0, 'batches': 0, 'err_max': 0.0}
        self.last_rescored = None
        self.cls_id, self.sep_id, self.pad_id = special_ids
        self.dist = dist                                  # torch.distributed module when sharded
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        if dist is not None:
            self.split_embed = os.environ.get('RAG_SPLIT_EMBED') != '0'
        # the exchanges run whenever there is more than one rank; RAG_DIST_FORCE=1 runs them with a single rank too
        # (every collective is then an identity): the only way to drive the RCCL branch of this file on a one-GPU box
        self.sharded = dist is not None and (self.world > 1 or os.environ.get('RAG_DIST_FORCE') == '1')
        self.times = StageTimes()
        self.timing = False
        # read the device-side consistency flags after every batch and raise if a kernel met inconsistent data
        # (synchronises the device: callers that pipeline batches switch it off and call rag_hip.check_faults()
        # themselves when they collect results)
        self.check_faults = hasattr(rag_hip, 'check_faults')
        self._side_stream = None
        self.overlap_bm25 = os.environ.get('RAG_OVERLAP_BM25', '1') != '0'
        self._bm25_stream = None
        # HIP graphs for the two encoder forwards of very small batches (RAG_GRAPHS=0 switches them off)
        # (measured, 1M rows, 50 pairs per query: the embedder's graph wins up to 32 queries, 2.1 -> 0.9-1.8 ms; the
        # cross-encoder's up to 4, beyond that its forward is no longer launch bound)
        off = os.environ.get('RAG_GRAPHS') == '0'
        self.graph_max_batch = 0 if off else int(os.environ.get('RAG_GRAPH_MAX_BATCH', '32'))
        self.graph_max_batch_ce = 0 if off else int(os.environ.get('RAG_GRAPH_MAX_BATCH_CE', '4'))
        # (bounded: every entry keeps the private memory pool of one captured forward alive; oldest dropped first)
        self._graphs = _GraphCache(limit=int(os.environ.get('RAG_GRAPH_CACHE', '32')))

    # the reference's candidate arithmetic (scripts/query.py:139, hybrid_indexer.py:83,90)
    @property
    def fetch_k(self):
        reranking = self.cross_encoder is not None or self.simple_rerank
        return max(self.top_k, self.rerank_top_n if reranking else self.top_k)

    def _timed(self, name, fn):
        if not self.timing:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        b.synchronize()
        setattr(self.times, name, getattr(self.times, name) + a.elapsed_time(b))
        return out

    # ---- stages ----------------------------------------------------------------------
    @torch.no_grad()
    def embed(self, ids: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """Query embeddings [B, dim].  Sharded: every rank has all the queries, so each embeds one slice of
        them and the slices are all-gathered (B x dim floats) instead of every rank embedding all of them."""
        b = ids.shape[0]
        if self.sharded and self.split_embed and b >= self.world:
            per = (b + self.world - 1) // self.world
            lo, hi = min(b, self.rank * per), min(b, (self.rank + 1) * per)
            mine = self._embed_local(ids[lo:hi].contiguous(), mask[lo:hi].contiguous()) if hi > lo else None
            dim = mine.shape[1] if mine is not None else self._embed_local(ids[:1], mask[:1]).shape[1]
            piece = torch.zeros((per, dim), dtype=torch.float32, device=ids.device)
            if mine is not None:
                piece[:hi - lo] = mine
            gathered = torch.empty((self.world * per, dim), dtype=torch.float32, device=ids.device)
            self._all_gather(gathered, piece)
            return gathered[:b].contiguous()
        return self._embed_local(ids, mask)

    # ---- collectives ----------------------------------------------------------------
    # Over RCCL (backend "nccl") device tensors go straight in.  Any other backend (gloo: CPU tests, and rehearsals
    # of the multi-rank path with several ranks sharing one GPU) gets host copies: gloo's own staging of device
    # tensors hung in such a rehearsal (both ranks inside the same small all_gather_into_tensor, every producer
    # kernel finished), and host tensors are the path gloo is built for.
    def _stage(self, t):
        return t.is_cuda and self.dist.get_backend() != 'nccl'

    def _all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        inp = inp.contiguous()
        if self._stage(inp):
            host = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(host, inp.cpu())
            out.copy_(host)
        else:
            self.dist.all_gather_into_tensor(out, inp)

    def _all_reduce_sum(self, t: torch.Tensor):
        if self._stage(t):
            host = t.cpu()
            self.dist.all_reduce(host)
            t.copy_(host)
        else:
            self.dist.all_reduce(t)

    def _graph_ok(self, b, dev, limit=None):
        return (0 < b <= (self.graph_max_batch if limit is None else limit) and dev.type == 'cuda' and not self.sharded
                and not torch.cuda.is_current_stream_capturing())

    def _graphed(self, key, fn, inputs):
        """The captured form of fn for this key (shapes), or None when capture failed once (then never tried again)."""
        return self._graphs.get(key, fn, inputs)

    def _embed_local(self, ids: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        bert = self.bi_encoder.bert
        if (getattr(bert, 'native', False) and self._graph_ok(ids.shape[0], ids.device)
                and bert.word.weight.dtype == torch.float32 and getattr(bert, 'split_gemm', False)):
            s = (ids.shape[1] + 31) // 32 * 32           # one graph per 32 columns of padded width ...
            b = ids.shape[0]
            b_pad = 1 << (b - 1).bit_length()            # ... and per power of two of queries (empty rows beyond the batch)
            if s != ids.shape[1] or b_pad != b:
                ids = torch.nn.functional.pad(ids, (0, s - ids.shape[1], 0, b_pad - b))
                mask = torch.nn.functional.pad(mask, (0, s - mask.shape[1], 0, b_pad - b))

            def fwd(i, m):
                p_ids, p_types, p_pos, cu = pack_static(i, m)
                return self.bi_encoder.forward_packed(p_ids, p_types, p_pos, cu, s, normalize=True)
            g = self._graphed(('embed', ids.shape[0], s, ids.dtype, mask.dtype), fwd, [ids, mask])
            if g is not None:
                return g(ids, mask)[:b].clone()
            ids, mask = ids[:b], mask[:b]
        if getattr(self.bi_encoder.bert, 'native', False):
            # unpadded: queries differ a lot in length, and padding positions cost full GEMM rows
            p_ids, p_types, p_pos, cu, longest = self.bi_encoder.bert.pack(ids, mask)
            return self.bi_encoder.forward_packed(p_ids, p_types, p_pos, cu, max(longest, 1), normalize=True)
        return self.bi_encoder(ids, None, mask, normalize=True).contiguous()

    def _gather_merge(self, score, row, cnt, k, smaller_is_better=False):
        """all-gather the per-shard lists and merge them to the global top-k (smaller_is_better: lists of
        distances, as a Euclidean index returns them)."""
        if not self.sharded:
            return score, row, cnt
        b = score.shape[0]
        kk = score.shape[1]
        # one exchange per list: scores, rows and counts of a query travel as one row of bytes
        # ([B, kk * (score bytes + 8) + 4]); the gathered tensor is the ranks' pieces along dim 0: [world * B, bytes]
        parts = [score.contiguous(), row.contiguous(), cnt.contiguous().reshape(b, 1)]
        widths = [p.shape[1] * p.element_size() for p in parts]
        mine = torch.cat([p.view(torch.uint8).reshape(b, w) for p, w in zip(parts, widths)], dim=1)
        gathered = torch.empty((self.world * b, mine.shape[1]), dtype=torch.uint8, device=mine.device)
        self._all_gather(gathered, mine)
        gathered = gathered.view(self.world, b, -1).permute(1, 0, 2)           # [B, world, bytes]
        cuts = [0, widths[0], widths[0] + widths[1], sum(widths)]
        gs, gr, gc = (gathered[:, :, lo:hi].contiguous().view(p.dtype)
                      for lo, hi, p in zip(cuts[:-1], cuts[1:], parts))
        return rag_hip.merge_topk_dev(gs, gr, gc.reshape(b, self.world), k, smaller_is_better=smaller_is_better)

    def dense_search(self, qvec: torch.Tensor, k: int):
        s, r, c = self.dense.search_dev(qvec, k)
        euclid = getattr(self.dense, 'metric', None) in (rag_hip.METRIC_EUCLID, 'euclidean', 'euclid')
        return self._gather_merge(s, r, c, k, smaller_is_better=euclid)

    def bm25_search(self, qterms: torch.Tensor, qoff: torch.Tensor, n_terms: int, k: int):
        s, r, c = self.bm25.search_dev(qterms, qoff, n_terms, k)
        return self._gather_merge(s, r, c, k)

    def fuse(self, vec, bm, t):
        vs, vr, vc = vec
        bs, br, bc = bm
        if self.fusion_method == 'rrf':
            return rag_hip.fuse_rrf_dev(vr, vc, br, bc, t, canon=self.canon)
        return rag_hip.fuse_weighted_dev(vr, vs.double(), vc, br, bs, bc, t, self.vw, self.bw, canon=self.canon)

    @torch.no_grad()
    def rerank(self, q_tok: torch.Tensor, q_off: torch.Tensor, fused_row: torch.Tensor, fused_cnt: torch.Tensor):
        """Cross-encoder scores of every (query, fused candidate) pair, then the per-query
        stable top-k.  Returns (positions into the fused list [B, top_k], counts, logits [B, T])."""
        b, t = fused_row.shape
        dev = fused_row.device
        n_pairs = b * t
        pair_query = torch.arange(b, dtype=torch.int32, device=dev).repeat_interleave(t)
        pair_row = fused_row.reshape(-1)
        live = torch.arange(t, device=dev)[None, :] < fused_cnt[:, None]
        screened = None
        if (self.screen_encoder is not None and n_pairs >= self.screen_min_pairs
                and not self._graph_ok(b, dev, self.graph_max_batch_ce)):
            logits, screened = self._score_screened(q_tok, q_off, pair_query, pair_row, live)
        else:
            logits = self._score_pairs(self.cross_encoder, q_tok, q_off, pair_query, pair_row, graph_batch=b)
        self.last_rescored = screened                 # [B, T] bool: which logits are exact ones (None: all of them)
        # pools are the first fused_cnt[q] entries of each row: pack them back to back
        seg_off = torch.zeros((b + 1,), dtype=torch.int32, device=dev)
        seg_off[1:] = torch.cumsum(fused_cnt, 0)
        logits = logits.reshape(b, t)
        # (a pair the screen left out keeps its cheap logit in `logits` for the record, but cannot be chosen)
        ranked = logits if screened is None else torch.where(screened, logits, torch.full_like(logits, float('-inf')))
        if bool((fused_cnt == t).all()):
            flat = ranked.reshape(-1).contiguous()
        else:
            flat = ranked[live].contiguous()
        idx, cnt = rag_hip.topk_segmented_dev(flat, seg_off, self.top_k)
        return idx, cnt, logits

    def _score_pairs(self, model, q_tok, q_off, pair_query, pair_row, graph_batch=0):
        """Logits of the given pairs (any subset, in any order) from `model`, the work split across the ranks by pair
        index (equal shares, the last one padded, so one fixed-size all-gather returns every logit)."""
        dev = pair_row.device
        store = self.chunk_tokens
        n_pairs = pair_row.shape[0]
        share = (n_pairs + self.world - 1) // self.world
        lo, hi = min(n_pairs, share * self.rank), min(n_pairs, share * (self.rank + 1))
        my_q, my_r = pair_query[lo:hi].contiguous(), pair_row[lo:hi].contiguous()
        n_mine = hi - lo
        logits_mine = torch.zeros((share,), dtype=torch.float32, device=dev)
        if n_mine:
            plen = rag_hip.ce_pair_len_dev(q_off, store.off, store.row_base, my_q, my_r, self.ce_max_len)
            mine = logits_mine[:n_mine]               # (the last rank's share may be shorter than the padded piece)
            if (self.ce_packed and model is self.cross_encoder and graph_batch
                    and self._graph_ok(graph_batch, dev, self.graph_max_batch_ce)
                    and self._ce_graphed(q_tok, q_off, my_q, my_r, plen, mine)):
                pass
            elif self.ce_packed:
                self._ce_packed(model, q_tok, q_off, my_q, my_r, plen, mine)
            else:
                self._ce_padded(model, q_tok, q_off, my_q, my_r, plen, mine)
        if self.sharded:
            gathered = torch.empty((share * self.world,), dtype=torch.float32, device=dev)
            self._all_gather(gathered, logits_mine)
            return gathered[:n_pairs]
        return logits_mine

    def _score_screened(self, q_tok, q_off, pair_query, pair_row, live):
        """Screen, then rescore.  Returns (logits [B, T] float32: exact where `rescored`, the screen's elsewhere;
        rescored [B, T] bool).

        Assumption, the same in both forms below: every pair's cheap logit is within E of its exact one up to an offset
        common to the query's pairs (which moves no order).  E is not known beforehand; it is measured on the pairs that
        were rescored — thousands per batch, each with both logits — as the largest deviation e of (cheap - exact) from
        its median over the query's rescored pairs, and E is taken as screen_safety * (running estimate of e).

        One stage (screen_two_stage False): rescore the pairs with cheap >= kth_cheap - 2 E.  A pair left out has an exact
        logit below kth_cheap - E - offset, while each of the k cheap-best has one of at least that.

        Two stages (default): first the k cheap-best of every query (and ties).  Their EXACT logits give T, the k-th best
        exact logit among them — a lower bound of the query's true k-th best, whatever else is rescored later — and the
        query's offset (median of cheap - exact over them).  A pair that was not rescored can only belong to the true
        top k if its exact logit reaches T, hence only if cheap >= T + offset - E: those are rescored in a second forward.
        One E instead of two, because one side of the comparison is now an exact value.

        Either way the batch's own e is compared with the estimate afterwards: if the estimate was below
        screen_min_safety / screen_safety of what it should have been, the bound is recomputed with the batch's e and
        the additional pairs are rescored (after three such rounds: every pair).  Sharded: the cheap logits are gathered
        first, every rank takes the same decisions, and the chosen pairs are split across the ranks."""
        b, t = live.shape
        k = min(self.top_k, t)
        cheap = self._score_pairs(self.screen_encoder, q_tok, q_off, pair_query, pair_row).reshape(b, t)
        neg = torch.full_like(cheap, float('-inf'))
        nan = torch.full_like(cheap, float('nan'))
        cheap_live = torch.where(live, cheap, neg)
        kth = torch.topk(cheap_live, k, dim=1).values[:, k - 1:k]               # -inf where a pool has fewer than k pairs
        exact = cheap.clone()
        rescored = torch.zeros_like(live)
        widened = 0
        any_rescored = [False]

        def rescore(mask):
            sel = torch.nonzero(mask.reshape(-1))[:, 0]
            if sel.numel():
                exact.view(-1)[sel] = self._score_pairs(self.cross_encoder, q_tok, q_off, pair_query[sel].contiguous(),
                                                        pair_row[sel].contiguous())

        def wanted(e_used):
            if e_used is None:                               # first batch: every pair
                return live & ~rescored
            if not self.screen_two_stage:
                return live & ~rescored & ((cheap_live >= kth - 2.0 * self.screen_safety * e_used) | torch.isinf(kth))
            if not any_rescored[0]:                          # stage one: the k cheap-best (whole pools that cannot fill a top k)
                return live & ((cheap_live >= kth) | torch.isinf(kth))
            t_q = torch.topk(torch.where(rescored, exact, neg), k, dim=1).values[:, k - 1:k]
            off = torch.nanmedian(torch.where(rescored, cheap - exact, nan), dim=1, keepdim=True).values
            return live & ~rescored & (cheap_live >= t_q + off - self.screen_safety * e_used)

        e_used = self._screen_err                            # running estimate of e (None: no batch seen yet)
        if e_used is not None and not bool(torch.isfinite(cheap_live[live]).all()):
            e_used = None                                    # a float16 overflow in the screen: it decides nothing, every pair is rescored
        while True:
            first = self.screen_two_stage and e_used is not None and not any_rescored[0]
            want = wanted(e_used)
            rescore(want)
            rescored |= want
            any_rescored[0] = True                           # (stage one is never empty: a live pair is some query's best)
            if first:                                        # stage two, from the exact logits of stage one
                want = wanted(e_used)
                rescore(want)
                rescored |= want
            # what e should have been, from the pairs that now have both logits
            d = torch.where(rescored & torch.isfinite(cheap), cheap - exact, nan)
            resid = (d - torch.nanmedian(d, dim=1, keepdim=True).values).abs()
            err = float(torch.nan_to_num(resid, nan=0.0).max())
            if (e_used is None or self.screen_min_safety * err <= self.screen_safety * e_used
                    or bool((rescored | ~live).all())):
                break
            if widened >= 3:                         # still too small after three rounds: take no chances
                rest = live & ~rescored
                rescore(rest)
                rescored |= rest
                break
            e_used = err
            widened += 1
        self._screen_err = err if self._screen_err is None else max(err, 0.97 * self._screen_err)
        st = self.screen_stats
        st['pairs'] += int(live.sum())
        st['rescored'] += int(rescored.sum())
        st['widened'] += widened
        st['batches'] += 1
        st['err_max'] = max(st['err_max'], err)
        return exact, rescored

    def _ce_graphed(self, q_tok, q_off, my_q, my_r, plen, logits_mine):
        """Packing and the cross-encoder forward of a very small batch as one captured HIP graph.  Static shapes, chosen
        from the batch's pair lengths (one small read-back, as the eager route has): the token stream in steps of 2048
        tokens, the longest pair in steps of 64, the query tokens in steps of 64 per query; sequences are run at their
        real lengths inside these.  Returns False when capture is unavailable (the caller then takes the eager route)."""
        store = self.chunk_tokens
        n = my_r.shape[0]
        b = q_off.shape[0] - 1
        cb = self.cross_encoder.bert
        if not (cb.word.weight.dtype == torch.float32 and getattr(cb, 'split_gemm', False)):
            return False
        total, longest = (int(v) for v in torch.stack([plen.sum(), plen.max()]).cpu().tolist())
        if longest <= 0:
            return False
        # at least one token beyond the stream: a trailing empty slot (fused row -1, no tokens) starts AT the stream's
        # end, and the forward reads the first row of every sequence
        t_pad = (total + 2048) // 2048 * 2048
        max_len = min(self.ce_max_len, (longest + 63) // 64 * 64)
        q_cap = (max(int(q_tok.shape[0]), 1) + 64 * b - 1) // (64 * b) * (64 * b)
        pair_query = my_q
        cut = self.ce_max_len                                # pairs are truncated at the searcher's limit, as always

        def fwd(qt, qo, rows):
            lens = rag_hip.ce_pair_len_dev(qo, store.off, store.row_base, pair_query, rows, cut)
            cu = torch.zeros((n + 1,), dtype=torch.int32, device=rows.device)
            cu[1:] = torch.cumsum(lens, 0)
            ids, types, pos = rag_hip.ce_pack_packed_dev(qt, qo, store.tok, store.off, store.row_base, pair_query, rows, cu,
                                                         t_pad, cut, self.cls_id, self.sep_id, padded_total=t_pad)
            return self.cross_encoder.forward_packed(ids, types, pos, cu, max_len)
        qt = torch.zeros((q_cap,), dtype=torch.int32, device=q_tok.device)
        qt[:q_tok.shape[0]] = q_tok
        g = self._graphed(('rerank', b, n, t_pad, max_len, q_cap, store), fwd, [qt, q_off, my_r])     # (the key holds the store itself:
        #                                                             an id() could be a freed store's, reused)
        if g is None:
            return False
        logits_mine[:n] = g(qt, q_off, my_r)
        return True

    def _ce_padded(self, model, q_tok, q_off, my_q, my_r, plen, logits_mine):
        """Pairs sorted by length, padded mini-batches (any model dtype, PyTorch or native ops)."""
        store = self.chunk_tokens
        n_mine = my_r.shape[0]
        dev = my_r.device
        order = torch.argsort(plen, descending=True)
        sorted_len = plen[order]
        # longest pair of every mini-batch (one small copy to the host sizes the buffers)
        starts = list(range(0, n_mine, self.ce_pairs_per_batch))
        heads = sorted_len[torch.tensor(starts, device=dev)].cpu().tolist()
        for s0, longest in zip(starts, heads):
            s1 = min(s0 + self.ce_pairs_per_batch, n_mine)
            if longest <= 0:
                continue                               # only empty slots (row -1) left
            seq = max(8, (int(longest) + 7) // 8 * 8)
            idx = order[s0:s1]
            ids, types, mask = rag_hip.ce_pack_dev(
                q_tok, q_off, store.tok, store.off, store.row_base, my_q[idx].contiguous(),
                my_r[idx].contiguous(), None, s1 - s0, seq, self.cls_id, self.sep_id, self.pad_id)
            out = model(ids, types, mask)
            logits_mine[idx] = out.float()

    def _ce_packed(self, model, q_tok, q_off, my_q, my_r, plen, logits_mine):
        """Longest pairs first: a mini-batch then holds pairs of similar length, and the attention kernel,
        whose LDS image is sized by the mini-batch's longest pair, fits more workgroups per compute unit on
        all but the first one.  A pair's logit does not depend on its place in the stream."""
        if self._tiles_wanted(model) and my_r.shape[0] >= 1:
            # 16-bit model whose layers run projection and attention in one kernel over 256-token tiles of whole pairs:
            # the pairs go in the order that fills those tiles (one copy of the lengths to the host decides it)
            lens_host = plen.cpu().numpy()
            packed = rag_hip.enc_qkv_attention_pack(lens_host)
            if packed is not None:
                order = torch.from_numpy(packed[0]).to(plen.device)
                sorted_logits = torch.zeros_like(logits_mine)
                self._ce_packed_stream(model, q_tok, q_off, my_q[order].contiguous(), my_r[order].contiguous(),
                                       plen[order].contiguous(), sorted_logits, tile_seq=packed[1],
                                       lens_host=lens_host[packed[0]])
                logits_mine[order] = sorted_logits
                return
        if not self.ce_sort_by_length or my_r.shape[0] < 2:
            return self._ce_packed_stream(model, q_tok, q_off, my_q, my_r, plen, logits_mine)
        order = torch.argsort(plen, descending=True, stable=True)
        sorted_logits = torch.zeros_like(logits_mine)
        self._ce_packed_stream(model, q_tok, q_off, my_q[order].contiguous(), my_r[order].contiguous(),
                               plen[order].contiguous(), sorted_logits)
        logits_mine[order] = sorted_logits

    @staticmethod
    def _tiles_wanted(model):
        b = getattr(model, 'bert', None)
        return (b is not None and getattr(b, 'fused_qkv_attention', False) and getattr(b, 'native_linear', False)
                and getattr(b, 'native', False) and b.word.weight.dtype in (torch.bfloat16, torch.float16)
                and b.config.hidden_size == 384 and b.config.num_attention_heads * 32 == 384)

    def _ce_packed_stream(self, model, q_tok, q_off, my_q, my_r, plen, logits_mine, tile_seq=None, lens_host=None):
        """Pairs in the given order as unpadded token streams: no padding inside a pair is
        computed at all (bfloat16 model, librag_hip encoder kernels).  The stream is cut at
        pair boundaries into mini-batches of at most `ce_tokens_per_batch` tokens, each rounded
        up to a multiple of `ce_token_bucket` with unused tail tokens, so the GEMMs see only a
        handful of distinct shapes (which is what lets hipBLASLt solutions be chosen per shape
        ahead of time)."""
        store = self.chunk_tokens
        n_mine = my_r.shape[0]
        dev = my_r.device
        cu_all = torch.zeros((n_mine + 1,), dtype=torch.int64, device=dev)
        cu_all[1:] = torch.cumsum(plen, 0)
        if lens_host is not None:                        # the caller already has the lengths on the host
            cu_host = np.concatenate([[0], np.cumsum(lens_host, dtype=np.int64)])
        else:
            cu_host = cu_all.cpu().numpy()               # the one host copy of the stage
        budget, bucket = self.ce_tokens_per_batch, self.ce_token_bucket
        cb = model.bert
        own_gemms = (getattr(cb, 'native_linear', False) and getattr(cb, 'fused_mlp', False)
                     and cb.word.weight.dtype in (torch.bfloat16, torch.float16) and cb.config.hidden_size == 384)
        mixed_gemms = cb.word.weight.dtype == torch.float32 and getattr(cb, 'split_gemm', False)
        if own_gemms or mixed_gemms:
            # every GEMM of the layers is one of the library's own kernels, which take any row count, or a
            # 16-bit-in / fp32-out library GEMM, which the per-shape solution table does not cover: the coarse
            # token grid (there for hipBLASLt's per-shape solutions) would only add padding rows
            bucket = min(bucket, 128)
        if os.environ.get('RAG_CE_TOKENS'):
            budget = int(os.environ['RAG_CE_TOKENS'])
        # Mini-batches are independent: with ce_streams > 1 (RAG_CE_STREAMS) they alternate between HIP streams.
        # Measured on the 1M-row workload (fp32 forward: matrix-core-bound GEMMs next to HBM-bound kernels): 2-4
        # streams are 0-4 % slower than one, the kernels of one mini-batch already fill the chip; off by default.
        main = torch.cuda.current_stream() if dev.type == 'cuda' else None
        n_streams = self.ce_streams if main is not None else 1
        if n_streams > 1:
            while len(self._ce_stream_pool) < n_streams:
                self._ce_stream_pool.append(torch.cuda.Stream())
            for st in self._ce_stream_pool[:n_streams]:
                st.wait_stream(main)
        # with a tile list (tile_seq[t] = first pair of tile t, pairs without tokens behind the last tile) the
        # mini-batches are cut at tile boundaries
        tile_tok = cu_host[tile_seq] if tile_seq is not None else None
        # (one upload for the whole stream: a copy from pageable host memory inside the loop would make the host wait for
        # the mini-batch before it, and the GPU for the host)
        tile_seq_dev = torch.from_numpy(np.ascontiguousarray(tile_seq, dtype=np.int32)).to(dev) if tile_seq is not None else None
        t0 = 0
        p0 = i_batch = 0
        while p0 < n_mine:
            # as many whole pairs as fit the token budget (a single pair always fits: budget >> max_len)
            tiles_dev = None
            if tile_seq is not None and t0 < len(tile_seq) - 1:
                t1 = int(np.searchsorted(tile_tok, tile_tok[t0] + budget, side='right')) - 1
                t1 = min(max(t1, t0 + 1), len(tile_seq) - 1)
                p1 = int(tile_seq[t1]) if t1 < len(tile_seq) - 1 else n_mine
                tiles_dev = tile_seq_dev[t0:t1 + 1] - p0
                t0 = t1
            elif tile_seq is not None:
                p1 = n_mine                                  # only pairs without tokens are left
            else:
                p1 = int(np.searchsorted(cu_host, cu_host[p0] + budget, side='right')) - 1
                p1 = min(max(p1, p0 + 1), n_mine)
            total = int(cu_host[p1] - cu_host[p0])
            side = self._ce_stream_pool[i_batch % n_streams] if n_streams > 1 else None
            i_batch += 1
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                if total <= 0:
                    p0 = p1
                    continue
                lens = plen[p0:p1]
                longest = int((cu_host[p0 + 1:p1 + 1] - cu_host[p0:p1]).max())
                cu = (cu_all[p0:p1 + 1] - cu_all[p0]).to(torch.int32)
                padded = (total + bucket - 1) // bucket * bucket
                ids, types, pos = rag_hip.ce_pack_packed_dev(q_tok, q_off, store.tok, store.off, store.row_base,
                                                             my_q[p0:p1], my_r[p0:p1], cu, total, self.ce_max_len,
                                                             self.cls_id, self.sep_id, padded_total=padded)
                kw = {'tiles': tiles_dev} if tiles_dev is not None else {}
                if int(np.count_nonzero(cu_host[p0 + 1:p1 + 1] == cu_host[p0:p1])) == 0:
                    logits_mine[p0:p1] = model.forward_packed(ids, types, pos, cu, longest, **kw)
                else:                                    # empty slots (row -1) have no tokens and no logit
                    # (in the tile order they are the last pairs of the last mini-batch: the live pairs keep their numbers)
                    keep = torch.nonzero(lens > 0)[:, 0]
                    cu_live = torch.cat([cu[:-1][keep], cu[-1:]])
                    logits_mine[p0 + keep] = model.forward_packed(ids, types, pos, cu_live, longest, **kw)
            p0 = p1
        if n_streams > 1:
            for st in self._ce_stream_pool[:n_streams]:
                main.wait_stream(st)

    def reserve(self, max_batch: int, max_terms_total: int):
        """Size the indexes' work buffers for batches of up to max_batch queries (max_terms_total BM25 query tokens in
        all) at this searcher's list length, so that the searches of the path never allocate afterwards."""
        k_list = min(2 * self.fetch_k, rag_hip.MAX_K)
        if self.dense is not None:
            self.dense.reserve(max_batch, k_list)
        if self.bm25 is not None:
            self.bm25.reserve(max_batch, max_terms_total, k_list)

    # ---- whole path ------------------------------------------------------------------
    def retrieve(self, embed_ids, embed_mask, qterms, qoff, n_terms):
        """First half of the path: embed -> dense kNN, BM25 -> fusion.  Returns the fused lists
        (positions, rows [B, T] global, scores f64, counts) as device tensors."""
        t = self.fetch_k
        k_list = min(2 * t, rag_hip.MAX_K)
        vec = bm = None
        # The BM25 search needs nothing from the embedder and is bound by vector-instruction issue, where the embedder's
        # and the dense scan's kernels are matrix-core bound: on a second stream it runs beside them (measured: 32 queries
        # 15.8 -> 14.7 ms end to end, 128 queries -2 %, nothing at 1024, and a loss below four queries).  (One rank only:
        # the sharded searches contain collectives, and one communicator must not be driven from two streams.)
        side = None
        if (self.overlap_bm25 and self.dense is not None and self.bm25 is not None and not self.timing and not self.sharded
                and embed_ids.is_cuda and embed_ids.shape[0] >= 4 and not torch.cuda.is_current_stream_capturing()):
            if self._bm25_stream is None:
                self._bm25_stream = torch.cuda.Stream()
            side, cur = self._bm25_stream, torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                bm = self.bm25_search(qterms, qoff, n_terms, k_list)
        if self.dense is not None:
            qvec = self._timed('embed', lambda: self.embed(embed_ids, embed_mask))
            self._qvec = qvec
            vec = self._timed('dense', lambda: self.dense_search(qvec, k_list))
        if side is not None:
            cur.wait_stream(side)
            for t_ in bm:
                t_.record_stream(cur)
        elif self.bm25 is not None:
            bm = self._timed('bm25', lambda: self.bm25_search(qterms, qoff, n_terms, k_list))
        b = (vec or bm)[1].shape[0]
        dev = (vec or bm)[1].device
        if vec is None:
            vec = (torch.zeros((b, 1), dtype=torch.float32, device=dev), torch.full((b, 1), -1, dtype=torch.int64, device=dev),
                   torch.zeros((b,), dtype=torch.int32, device=dev))
        if bm is None:
            bm = (torch.zeros((b, 1), dtype=torch.float64, device=dev), torch.full((b, 1), -1, dtype=torch.int64, device=dev),
                  torch.zeros((b,), dtype=torch.int32, device=dev))
        return self._timed('fuse', lambda: self.fuse(vec, bm, t))

    def rerank_simple(self, qvec: torch.Tensor, fused_row: torch.Tensor, fused_cnt: torch.Tensor):
        """Cosine of every fused candidate's resident embedding with its query's embedding (exact float32 dot
        products of unit vectors, rag_dense_rescore_dev), then the per-query stable top-k.  Sharded: a row's
        score comes from the rank that holds it (the others report 0 for it), summed over the ranks."""
        b, t = fused_row.shape
        scores = self.dense.rescore_dev(qvec.contiguous(), fused_row.contiguous())
        if self.sharded:
            self._all_reduce_sum(scores)
        seg_off = torch.zeros((b + 1,), dtype=torch.int32, device=fused_row.device)
        seg_off[1:] = torch.cumsum(fused_cnt, 0)
        if bool((fused_cnt == t).all()):
            flat = scores.reshape(-1).contiguous()
        else:
            keep = torch.arange(t, device=fused_row.device)[None, :] < fused_cnt[:, None]
            flat = scores[keep].contiguous()
        idx, cnt = rag_hip.topk_segmented_dev(flat, seg_off, self.top_k)
        return idx, cnt, scores

    def finish(self, fused, ce_q_tok=None, ce_q_off=None, rerank_qvec=None):
        """Second half: rerank of the fused lists (cross-encoder, or the simple reranker from resident rows
        with `rerank_qvec` = embeddings of the ORIGINAL queries when they differ from the retrieval ones,
        scripts/query.py:134-136,160), or the plain cut to top_k."""
        pos, row, score, cnt = fused
        out = {'fused_row': row, 'fused_score': score, 'fused_count': cnt}
        if self.cross_encoder is None and not self.simple_rerank:
            out.update(rows=row[:, :self.top_k], scores=score[:, :self.top_k], count=torch.clamp(cnt, max=self.top_k))
            return out
        if self.cross_encoder is None:
            qv = rerank_qvec if rerank_qvec is not None else self._qvec
            idx, rc, logits = self._timed('rerank', lambda: self.rerank_simple(qv, row, cnt))
        else:
            idx, rc, logits = self._timed('rerank', lambda: self.rerank(ce_q_tok, ce_q_off, row, cnt))
        safe = idx.clamp(min=0).long()
        out.update(rows=torch.where(idx >= 0, torch.gather(row, 1, safe), torch.full_like(safe, -1)),
                   scores=torch.gather(score, 1, safe), rerank_score=torch.gather(logits, 1, safe), count=rc,
                   rerank_pos=idx, logits=logits)
        return out

    def search(self, embed_ids, embed_mask, qterms, qoff, n_terms, ce_q_tok=None, ce_q_off=None):
        """One batch through the whole path.  Returns dict of device tensors:
        rows [B, top_k] (global), fused_score f64 [B, top_k], rerank_score f32 (if reranked)."""
        out = self.finish(self.retrieve(embed_ids, embed_mask, qterms, qoff, n_terms), ce_q_tok, ce_q_off)
        if self.check_faults:
            rag_hip.check_faults()
        return out

    def search_many(self, batches):
        """Several batches, software-pipelined over two HIP streams: while the cross-encoder
        reranks batch i on the current stream, the retrieval half of batch i+1 (small kernels
        that leave most of the chip idle) runs on a side stream.  ``batches`` is a list of
        argument tuples of ``search``; returns the list of its result dicts.  With more than one
        rank the batches run one after the other: the retrieval half contains collectives, and
        one communicator must not be driven from two streams at once."""
        if self.sharded or self.cross_encoder is None or len(batches) < 2 or self.timing:
            return [self.search(*b) for b in batches]
        main = torch.cuda.current_stream()
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream()
        side = self._side_stream
        outs, ready = [], None

        def launch(b):
            side.wait_stream(main)                    # inputs prepared on the current stream
            with torch.cuda.stream(side):
                fused = self.retrieve(*b[:5])
                ev = torch.cuda.Event()
                ev.record(side)
            for t in fused:
                t.record_stream(main)                 # consumed on the current stream later
            return fused, ev
        nxt = launch(batches[0])
        for i, b in enumerate(batches):
            fused, ev = nxt
            if i + 1 < len(batches):
                nxt = launch(batches[i + 1])
            main.wait_event(ev)
            outs.append(self.finish(fused, *b[5:7]))
        if self.check_faults:
            rag_hip.check_faults()
        return outs
