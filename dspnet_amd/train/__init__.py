"""Training-step driver and metric readouts (numerics contract of multi_solver.py / train/metric.py)."""
