#!/usr/bin/env python3
"""Generate tests/golden/retrieval_metrics.json (run in the build container only): inputs and the
outputs of the reference's own ``calculate_retrieval_metrics``
(reference evaluation/comprehensive_evaluation.py:25-70) for them.  Third-party modules that the
reference imports at module level are replaced by inert placeholders, as in make_golden.py.

Usage:  python tests/golden/make_golden_metrics.py [/root/reference]
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402  (placeholders + reference import path)


def main():
    make_golden.import_reference()
    sys.path.insert(0, os.path.join(make_golden.REF, 'evaluation'))
    sys.path.insert(0, os.path.join(make_golden.REF, 'scripts'))
    from comprehensive_evaluation import calculate_retrieval_metrics
    cases = [
        (['a', 'b'], ['a', 'c', 'b', 'd', 'e']),
        (['a'], ['x', 'y', 'a']),
        (['a', 'b', 'c'], []),
        ([], ['a', 'b']),
        (['a'], ['a', 'a', 'a', 'b']),                 # duplicates in the retrieved list
        (['a', 'a', 'b'], ['b', 'z']),                 # duplicates in the ground truth
        (['d1', 'd2', 'd3', 'd4'], ['d9', 'd3', 'd3', 'd1', 'd8']),
        (['q'], ['w', 'e', 'r', 't', 'y']),
    ]
    out = [{'ground_truth': g, 'retrieved': r, 'metrics': calculate_retrieval_metrics(list(g), list(r))} for g, r in cases]
    with open(os.path.join(HERE, 'retrieval_metrics.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print(f'{len(out)} cases written')


if __name__ == '__main__':
    main()
