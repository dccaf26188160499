from .metrics import calculate_retrieval_metrics, evaluate_retrieval

__all__ = ['calculate_retrieval_metrics', 'evaluate_retrieval']
