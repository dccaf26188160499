"""Retrieval metrics against the outputs of the reference's own function (tests/golden)."""
import json
import os

from evaluation import calculate_retrieval_metrics, evaluate_retrieval

HERE = os.path.dirname(os.path.abspath(__file__))


def test_metrics_equal_the_reference_outputs():
    with open(os.path.join(HERE, 'golden', 'retrieval_metrics.json')) as f:
        cases = json.load(f)
    assert len(cases) >= 8
    for c in cases:
        assert calculate_retrieval_metrics(c['ground_truth'], c['retrieved']) == c['metrics']


def test_evaluation_loop_scores_and_times_each_query():
    class Index:
        def search(self, query, top_k=5):
            docs = {'q1': ['d1', 'd1', 'd2'], 'q2': ['d9']}[query]
            return [{'doc_id': d, 'text': d} for d in docs][:top_k]

    class Rerank:
        top_n = 3

        def rerank(self, query, results, top_k=5):
            return list(reversed(results))[:top_k]

    qs = [{'query': 'q1', 'ground_truth': ['d2']}, {'query': 'q2', 'ground_truth': ['d1']}]
    out = evaluate_retrieval(Index(), qs, top_k=2)
    assert out['num_queries'] == 2 and out['per_query'][0]['retrieved_docs'] == ['d1', 'd1']
    assert out['recall'] == 0.0
    out = evaluate_retrieval(Index(), qs, top_k=2, reranker=Rerank())
    assert out['per_query'][0]['retrieved_docs'] == ['d2', 'd1'] and out['per_query'][0]['metrics']['mrr'] == 1.0
    assert out['recall'] == 0.5 and out['mean_retrieval_time'] >= 0
