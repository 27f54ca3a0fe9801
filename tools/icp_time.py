import sys, time, numpy as np, torch
sys.path.insert(0,'.')
from lidarregistration_amd import synth, ransac
p = synth.make_pair(N=30000, seed=51)
T0 = p["T_gt"].copy(); T0[:3,3] += [0.2,-0.1,0.05]
x0 = torch.from_numpy(p["xyz0"]).cuda(); x1 = torch.from_numpy(p["xyz1"]).cuda()
for _ in range(3): T, info = ransac.icp_dev(x0, x1, T0)
torch.cuda.synchronize(); t=time.time()
for _ in range(20): T, info = ransac.icp_dev(x0, x1, T0)
torch.cuda.synchronize(); print('icp 30k: %.3f ms per call' % ((time.time()-t)/20*1e3), info)
