"""Prediction store of the inference / evaluation / export stage (reference: detnet/trainer/predictions.py)."""
from .predictions import Predictions      # noqa: F401
