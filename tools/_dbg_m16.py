import sys, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from lidarregistration_amd import synth, matching, _ext

from oracle import oracle
_ext.lib()
F0, F1 = synth.make_features(2048, 2048, 32, 0.5, 1.0, 1)
i1, i2, s1, s2 = matching.nn_top2_dev(F0, F1, want_2nd=True, want_dist=True)
o1, o2, os1, os2 = oracle.nn_top2(F0, F1)
i1 = i1.cpu().numpy(); i2 = i2.cpu().numpy()
bad1 = np.nonzero(i1 != o1)[0]; bad2 = np.nonzero(i2 != o2)[0]
print("bad1", len(bad1), bad1[:40], "mod64", np.bincount(bad1 % 64, minlength=64))
print("bad2", len(bad2), bad2[:40], "mod64", np.bincount(bad2 % 64, minlength=64))
print("cols of missed best mod 32:", np.bincount(o1[bad1] % 32, minlength=32))
