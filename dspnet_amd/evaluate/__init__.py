"""Evaluation read-outs (numerics contract of multi_eval.py / evaluate/eval_metric.py)."""
