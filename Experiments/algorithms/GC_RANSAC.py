"""`from algorithms.GC_RANSAC import GC_RANSAC` -- same call shape as the reference (Experiments/algorithms/GC_RANSAC.py)."""
from lidarregistration_amd.ransac import GC_RANSAC  # noqa: F401
