import numpy as np
from scipy.optimize import minimize
from _pregrasp_fk import *
rng=np.random.RandomState(0)
lo=m['jnt_range'].reshape(-1,2)[:,0]; hi=m['jnt_range'].reshape(-1,2)[:,1]
def cost(q5, target, open_dir, jaw):
    q=np.r_[q5,jaw]
    pf,Rf=geom_world(q,'fixed_jaw_pad_3'); pm,Rm=geom_world(q,'moving_jaw_pad_3')
    pf2,_=geom_world(q,'fixed_jaw_pad_1'); 
    mid=0.5*(pf+pm)
    c=np.sum((mid-target)**2)*1e4
    nf=Rf[:,0]                               # fixed pad outward normal
    c+= (1-abs(nf@open_dir))*10              # opening direction
    down=(pf2-pf); down/=np.linalg.norm(down)   # pad1 is nearer the tip than pad3 -> finger direction
    c+= (1+down[2])*10                        # fingers point down
    c+= 1e-3*np.sum(np.maximum(0,q5-hi[:5]+0.1)**2+np.maximum(0,lo[:5]+0.1-q5)**2)*1e3
    return c
def solve(target, open_dir, jaw=0.9):
    best=None
    for _ in range(40):
        q0=rng.uniform(lo[:5],hi[:5])
        r=minimize(cost,q0,args=(target,open_dir,jaw),method='Nelder-Mead',options=dict(maxiter=4000,xatol=1e-5,fatol=1e-8))
        if best is None or r.fun<best.fun: best=r
    return best
t=np.array([0.25,0.0,0.4217+0.0178]); od=np.array([1.0,0,0])
r=solve(t,od); q=np.r_[r.x,0.9]
print("cost",r.fun,"q",np.round(q,4))
for n in ('fixed_jaw_pad_1','fixed_jaw_pad_3','fixed_jaw_pad_4','moving_jaw_pad_1','moving_jaw_pad_3','moving_jaw_pad_4'):
    p,R=geom_world(q,n); print(n,np.round(p,4),np.round(R[:,0],2))
P,Q=fk(q); print("links z", np.round(P[2:8,2],3))
import numpy as np

def cost2(q5, target, open_dir, jaw):
    c=cost(q5,target,open_dir,jaw)
    q=np.r_[q5,jaw]
    zmin=min(geom_world(q,n)[0][2] for n in ('fixed_jaw_pad_1','moving_jaw_pad_1'))
    return c+1e4*max(0,0.4275-zmin)**2

best=None
t=np.array([0.25,0.0,0.4217+0.0178+0.008])
for _ in range(60):
    q0=rng.uniform(lo[:5],hi[:5])
    r=minimize(cost2,q0,args=(t,np.array([1.,0,0]),0.9),method='Nelder-Mead',options=dict(maxiter=6000,xatol=1e-6,fatol=1e-9))
    if best is None or r.fun<best.fun: best=r
q=np.r_[best.x,0.9]
print("cost",best.fun,"q",repr(np.round(q,4)))
for jaw in (0.9,0.5,0.2,0.0):
    qq=q.copy(); qq[5]=jaw
    print("jaw",jaw,[ (n,np.round(geom_world(qq,n)[0],3)) for n in ('fixed_jaw_pad_3','moving_jaw_pad_3','moving_jaw_pad_1')])
for n in ('fixed_jaw_pad_1','fixed_jaw_pad_4','moving_jaw_pad_1','moving_jaw_pad_4','Fixed_Jaw_Collision_2'):
    p,R=geom_world(q,n); print(n,np.round(p,4))
