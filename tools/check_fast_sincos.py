"""Numpy (float32-emulated) validation of the range reduction + Taylor kernels behind fast_sin / fast_cos in\npn_chain.hip: prints the low part of 1/2pi and the maximum error over 2^-5 <= |y| < 2^19."""
import numpy as np
# emulate in numpy float32 the reduction to validate constants/quadrant logic
def f32(x): return np.float32(x)
c_hi=f32(0.15915494); c_lo=f32(1/(2*np.pi)-float(c_hi))
print("c_lo", c_lo)
def red(y):
    y=y.astype(np.float32)
    p=(y*c_hi).astype(np.float32)
    e=(y.astype(np.float64)*float(c_hi)-p.astype(np.float64)).astype(np.float32)  # fma exact error
    e=(y.astype(np.float64)*float(c_lo)+e.astype(np.float64)).astype(np.float32)
    f=((p-np.rint(p)).astype(np.float32)+e).astype(np.float32)
    q4=(f*f32(4)).astype(np.float32); qn=np.rint(q4)
    return ((q4-qn)*f32(1.5707963267948966)).astype(np.float32), qn.astype(np.int64)
def ksin(x):
    z=x*x; p=z*f32(2.7557319e-6)+f32(-1.9841270e-4); p=z*p+f32(8.3333333e-3); p=z*p+f32(-1.6666667e-1); return (x*z)*p+x
def kcos(x):
    z=x*x; p=z*f32(-2.7557319e-7)+f32(2.4801587e-5); p=z*p+f32(-1.3888889e-3); p=z*p+f32(4.1666667e-2); p=z*p+f32(-0.5); return z*p+f32(1)
rng=np.random.default_rng(0)
y=(rng.uniform(-1,1,2000000)*2.0**rng.integers(-5,19,2000000)).astype(np.float32)
r,q=red(y); s,c=ksin(r),kcos(r)
v=np.where(q&1,c,s); sn=np.where(q&2,-v,v)
v2=np.where(q&1,s,c); cs=np.where((q+1)&2,-v2,v2)
print("sin err", np.abs(sn-np.sin(y.astype(np.float64))).max(), "cos err", np.abs(cs-np.cos(y.astype(np.float64))).max())
