import numpy as np
from so101_sim_amd.model import scenes, blob as blobfmt
raw,meta=scenes.load_blob("banana","f64")
m=blobfmt.unpack(raw)
def q2m(q):
    w,x,y,z=q
    return np.array([[1-2*(y*y+z*z),2*(x*y-w*z),2*(x*z+w*y)],[2*(x*y+w*z),1-2*(x*x+z*z),2*(y*z-w*x)],[2*(x*z-w*y),2*(y*z+w*x),1-2*(x*x+y*y)]])
def qmul(a,b):
    return np.array([a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3], a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2], a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1], a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0]])
bp=m['body_pos'].reshape(-1,3); bq=m['body_quat'].reshape(-1,4); par=m['body_parent']; axis=m['jnt_axis'].reshape(-1,3)
arm=list(m['arm_body'])
def fk(q):
    nb=len(par); P=np.zeros((nb,3)); Q=np.zeros((nb,4)); Q[0]=[1,0,0,0]
    for b in range(1,nb):
        if b in (11,12): continue
        p=par[b]; R=q2m(Q[p]); P[b]=P[p]+R@bp[b]; Q[b]=qmul(Q[p],bq[b])
        if b in arm:
            k=arm.index(b); h=0.5*q[k]; Q[b]=qmul(Q[b],np.r_[np.cos(h),np.sin(h)*axis[k]])
    return P,Q
gname=meta['geom_names']; gb=m['geom_body']; gp=m['geom_pos'].reshape(-1,3); gq=m['geom_quat'].reshape(-1,4); gs=m['geom_size'].reshape(-1,3)
def geom_world(q,name):
    P,Q=fk(q); g=gname.index(name); b=gb[g]; R=q2m(Q[b]); return P[b]+R@gp[g], R@q2m(gq[g])
if __name__=="__main__":
    for q in (np.zeros(6), np.array([0,-1.57079,1.57079,1.57079,-1.57079,0.])):
        P,Q=fk(q)
        print("q",q)
        for b,n in zip(arm,meta['body_names'][2:8]): print("  ",n,np.round(P[b],3))
        for n in ('fixed_jaw_pad_3','moving_jaw_pad_3'):
            p,R=geom_world(q,n); print("  ",n,np.round(p,3),"normal(x axis)",np.round(R[:,0],2))
    print(meta['body_names'], m['jnt_range'].reshape(-1,2))

def prop_cloud(body):
    pts=[]
    mv=m['mesh_vert'].reshape(-1,3)
    for g in range(len(gb)):
        if gb[g]==body and m['geom_vertnum'][g]>0:
            v=mv[m['geom_vertadr'][g]:m['geom_vertadr'][g]+m['geom_vertnum'][g]]
            pts.append(v@q2m(gq[g]).T+gp[g])
    return np.vstack(pts)
