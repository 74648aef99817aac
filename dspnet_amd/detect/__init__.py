"""Inference wrappers (numerics contract of detect/multitask_detector.py)."""
