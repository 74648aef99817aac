"""Data path of the reference (dataset/iterator.py): RecordIO reader and the multi-task batch iterator."""
