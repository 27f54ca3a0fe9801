"""`from algorithms.FR import FR` -- same import path as the reference (Experiments/algorithms/FR.py)."""
from lidarregistration_amd.FR import FR, PointCloud, make_open3d_point_cloud, pair_params  # noqa: F401
from lidarregistration_amd.ransac import RANSAC_registration  # noqa: F401
