"""`from algorithms.matching import ...` -- same names as the reference (Experiments/algorithms/matching.py)."""
from lidarregistration_amd.matching import (Grid_Prioritized_Filter, calc_distance_ratio_in_feature_space, find_2nn,  # noqa: F401
                                            find_nn, mark_best_buddies, measure_inlier_ratio, nn_to_mutual)
