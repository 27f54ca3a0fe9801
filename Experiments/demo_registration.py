"""BASELINE config #1: one synthetic 5k-point pair through the whole path (plumbing demo).

    python demo_registration.py --algo RANSAC --mode MMN --iters 1000

The reference's demo_registration.py is a PointDSC viewer without RANSAC flags (Experiments/demo_registration.py:61-67);
this one exercises FR() with the reference's flag names and prints the recovered motion next to the ground truth.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--algo", default="RANSAC", choices=["RANSAC"])
    p.add_argument("--codebase", default="GC", choices=["open3D", "GC"])
    p.add_argument("--mode", default="MMN")
    p.add_argument("--iters", type=int, default=1000)
    p.add_argument("--n", type=int, default=5000)
    p.add_argument("--seed", type=int, default=51)
    p.add_argument("--fast_rejection", default="ELC")
    p.add_argument("--GPF_factor", type=float, default=2.0)
    p.add_argument("--GPF_grid_wid", type=int, default=10)
    args = p.parse_args(argv)
    import torch
    from lidarregistration_amd import metrics, synth
    from algorithms.FR import FR
    pr = synth.make_pair(N=args.n, rho=0.5, s=0.9, seed=args.seed)
    t = torch.from_numpy
    T, elapsed, _, _, n_init, ir_init, n_filt, ir_filt = FR(t(pr["xyz0"]), t(pr["xyz1"]), t(pr["feats0"]), t(pr["feats1"]), args, pr["T_gt"])
    np.set_printoptions(precision=4, suppress=True)
    print(f"{n_init} nn pairs ({ir_init:.3f} inliers), {n_filt} filtered pairs ({ir_filt:.3f} inliers), {elapsed*1e3:.2f} ms")
    print("estimated motion:\n", T, "\nground truth:\n", pr["T_gt"])
    print(f"RE = {metrics.rotation_error_deg(T, pr['T_gt']):.4f} deg, TE = {metrics.translation_error_cm(T, pr['T_gt']):.3f} cm")
    return T


if __name__ == "__main__":
    main()
