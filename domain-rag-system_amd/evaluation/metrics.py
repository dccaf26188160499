"""Retrieval metrics and the timed evaluation loop of the reference's evaluation harness.

``calculate_retrieval_metrics`` follows reference ``evaluation/comprehensive_evaluation.py:25-70``
(pinned by ``tests/golden/retrieval_metrics.json``, outputs of the reference's own function):
the retrieved ids are de-duplicated in order first (several chunks of one document count once),
``k`` is the number of distinct retrieved ids, precision = hits / k, recall = hits / distinct
ground-truth ids, hit@k = 1 if anything was found, MRR = reciprocal rank of the first hit.

``evaluate_retrieval`` is the loop of reference ``evaluation/comprehensive_evaluation.py:298-371``
reduced to what the retrieval path needs: per query, search (and rerank when a reranker is given),
map the results to document ids, score them, time the retrieval.
"""
import statistics
import time
from typing import Callable, Dict, Iterable, List, Optional


def calculate_retrieval_metrics(ground_truth: List[str], retrieved: List[str]) -> Dict[str, float]:
    wanted = set(ground_truth)
    distinct = list(dict.fromkeys(retrieved))
    k = len(distinct)
    if k == 0:
        return {'precision': 0.0, 'recall': 0.0, 'f1': 0.0, 'hit@k': 0.0, 'mrr': 0.0}
    hits = sum(1 for r in distinct if r in wanted)
    precision = hits / k
    recall = hits / len(wanted) if wanted else 0.0
    f1 = 2 * precision * recall / (precision + recall) if precision + recall > 0 else 0.0
    mrr = 0.0
    for rank, r in enumerate(distinct, start=1):
        if r in wanted:
            mrr = 1.0 / rank
            break
    return {'precision': precision, 'recall': recall, 'f1': f1, 'hit@k': 1.0 if hits else 0.0, 'mrr': mrr}


def evaluate_retrieval(indexer, queries: Iterable[Dict], top_k: int = 5, reranker=None, fetch_k: Optional[int] = None,
                       doc_id_of: Callable[[Dict], str] = lambda r: r.get('doc_id', ''),
                       preprocess: Optional[Callable[[str], str]] = None) -> Dict:
    """queries: dicts with 'query' and 'ground_truth' (document ids).  Returns the mean of every
    metric, the mean / median retrieval time in seconds and the per-query records."""
    records = []
    for item in queries:
        text = item['query']
        asked = preprocess(text) if preprocess is not None else text
        t0 = time.perf_counter()
        n_fetch = fetch_k if fetch_k is not None else (max(top_k, reranker.top_n) if reranker is not None else top_k)
        results = indexer.search(asked, top_k=n_fetch)
        if reranker is not None:
            results = reranker.rerank(text, results, top_k=top_k)
        else:
            results = results[:top_k]
        elapsed = time.perf_counter() - t0
        retrieved = [doc_id_of(r) for r in results]
        records.append({'query': text, 'retrieved_docs': retrieved, 'retrieval_time': elapsed,
                        'metrics': calculate_retrieval_metrics(list(item.get('ground_truth', [])), retrieved)})
    out = {'num_queries': len(records), 'per_query': records}
    if records:
        for key in ('precision', 'recall', 'f1', 'hit@k', 'mrr'):
            out[key] = statistics.mean(r['metrics'][key] for r in records)
        times = [r['retrieval_time'] for r in records]
        out['mean_retrieval_time'] = statistics.mean(times)
        out['median_retrieval_time'] = statistics.median(times)
    return out
