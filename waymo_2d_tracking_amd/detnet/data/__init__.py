"""Evaluation side of the detection stage (reference: detnet/data/metric.py, data/__init__.py, detnet/data/coco.py)."""
