"""`import pygcransac` as the reference's Experiments/algorithms/GC_RANSAC.py:2-5 writes it (run from this directory)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lidarregistration_amd.pygcransac import *          # noqa: F401,F403,E402
from lidarregistration_amd import pygcransac as _impl   # noqa: E402

findRigidTransform = _impl.findRigidTransform
