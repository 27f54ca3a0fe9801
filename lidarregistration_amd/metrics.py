"""Registration metric and run summary of the reference harness, evaluated in float64.

RE = acos(clamp((tr(R^T R_gt) - 1) / 2)) in degrees, TE = |t - t_gt| in cm, success iff RE < re_thre and TE < te_thre
(Experiments/libs/loss.py:44-50; thresholds 5 deg / 60 cm, Experiments/test.py:330-331).
"""
import numpy as np

RE_THRE_DEG, TE_THRE_CM = 5.0, 60.0


def rotation_error_deg(T, T_gt):
    R, Rg = np.asarray(T, np.float64)[:3, :3], np.asarray(T_gt, np.float64)[:3, :3]
    return float(np.degrees(np.arccos(np.clip((np.trace(R.T @ Rg) - 1) / 2.0, -1, 1))))


def translation_error_cm(T, T_gt):
    return float(np.linalg.norm(np.asarray(T, np.float64)[:3, 3] - np.asarray(T_gt, np.float64)[:3, 3]) * 100)


def is_success(T, T_gt, re_thre=RE_THRE_DEG, te_thre=TE_THRE_CM):
    return rotation_error_deg(T, T_gt) < re_thre and translation_error_cm(T, T_gt) < te_thre


def summarize(all_stats, algo="RANSAC"):
    """The summary block of Experiments/test.py:65-84 (same wording, same columns)."""
    s = np.asarray(all_stats, np.float64)
    avg = s.mean(0)
    ok = s[s[:, 0] == 1]
    okavg = ok.mean(0) if len(ok) else np.full(s.shape[1], np.nan)
    t99 = np.quantile(s[:, 9], 0.99)
    n = s.shape[0]
    out = "\n"
    out += f"{avg[15]:.0f} nn pairs ({avg[16]:.3f} inliers), {avg[17]:.0f} filtered pairs ({avg[18]:.3f} inliers)\n"
    out += (f"{algo}     | recall: {100*avg[0]:.2f}%, #failed/#total: {int((s[:,0]==0).sum())}/{n}, TE(cm): {okavg[2]:.3f}, "
            f"RE(deg): {okavg[1]:.3f}, mean reg time(s): {avg[9]:.3f}, 99% reg time(s): {t99:.3f}\n")
    okicp = s[s[:, 12] == 1]
    icpavg = okicp.mean(0) if len(okicp) else np.full(s.shape[1], np.nan)
    out += (f"{algo}+ICP | recall: {100*avg[12]:.2f}%, #failed/#total: {int((s[:,12]==0).sum())}/{n}, TE(cm): {icpavg[14]:.3f}, "
            f"RE(deg): {icpavg[13]:.3f}, ICP time(s): {avg[11]:.3f}, Total time(s) {avg[9]+avg[11]:.3f}\n")
    # (not in the reference's block) what "reg time" covers here, so that the two are not compared line by line
    out += ("note: reg time = what FR.py:117 bills (filter incl. the reverse NN + RANSAC + the 2nd neighbour's surcharge) as an ESTIMATE: "
            "a pair's share of its window's registration time (batched calls) or its call's device time (--serial), minus the first "
            "neighbour's part of the forward NN from the library's stage events and a share calibrated once per cloud size; "
            "ICP is timed on its own\n")
    return out
