import numpy as np, sys
sys.path.insert(0,'.')
from lidarregistration_amd import synth, matching
from oracle import oracle as orc
F0,F1 = synth.make_features(2048,2048,32,0.5,1.0,1)
i1,i2,s1,s2 = matching.nn_top2_dev(F0,F1,True,True)
o1,o2,os1,os2 = orc.nn_top2(F0,F1)
s1=s1.cpu().numpy(); s2=s2.cpu().numpy()
bad = np.nonzero(s1.view(np.uint32)!=os1.view(np.uint32))[0]
print('mismatch s1', len(bad), 'of', len(s1), ' s2', (s2.view(np.uint32)!=os2.view(np.uint32)).sum())
d = s1.view(np.int32).astype(np.int64)-os1.view(np.int32).astype(np.int64)
print('ulp diff hist', np.unique(d, return_counts=True))
# host recomputation variants for the first few bad rows
import ctypes
def fma32(a,b,c):
    # exact fmaf via float64 is not exact; use np.longdouble (80-bit) -> product of two 24-bit is exact in 64-bit mantissa, sum rounding once to fp32 nearly always exact
    return np.float32(np.longdouble(a)*np.longdouble(b)+np.longdouble(c))
for r in bad[:5]:
    a=F0[r]; b=F1[o1[r]]
    n0=np.float32(0); n1=np.float32(0)
    for k in range(32): n0=fma32(a[k],a[k],n0); n1=fma32(b[k],b[k],n1)
    acc=np.float32(0)
    for k in range(32): acc=fma32(a[k],b[k],acc)
    d2=fma32(np.float32(-2),acc,np.float32(n0+n1))
    # variant: pairwise per mfma: (a0*b0 + a1*b1) then + C
    acc2=np.float32(0)
    for m in range(16):
        acc2 = np.float32(np.longdouble(a[2*m])*np.longdouble(b[2*m]) + np.longdouble(a[2*m+1])*np.longdouble(b[2*m+1]) + np.longdouble(acc2))
    d2b=fma32(np.float32(-2),acc2,np.float32(n0+n1))
    print(r, 'gpu s', s1[r], 'orc s', os1[r], 'chain sqrt', np.sqrt(np.float32(max(d2,1e-30))), 'fused-pair sqrt', np.sqrt(np.float32(max(d2b,1e-30))), 'gpu^2', s1[r]*s1[r], 'd2',d2,'d2b',d2b)
