"""CPU experiment (test infrastructure, numpy): how far a pattern search on the unit sphere, started at the MPR normal of every non-flat
contacting pair of the twelve contact-rich fixture states, moves the reported depth towards the brute-forced minimum translation
(oracle/geomcheck.py), as a function of the number of support-function pairs it spends.  Result in DESIGN.md section 4."""
import json, numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import geomcheck as gc
from oracle.oracle import Oracle
from so101_sim_amd.model import blob as blobfmt, scenes
raw = scenes.load_blob("banana","f64")[0]; model = blobfmt.unpack(raw)
states = json.load(open(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'tests', 'golden', 'contact_rich_states.json')))["states"]
def tangent(u):
    a = np.array([0,1.0,0]) if abs(u[1]) < 0.5 else np.array([0,0,1.0])
    t1 = a - (a@u)*u; t1 /= np.linalg.norm(t1); return t1, np.cross(u, t1)
def refine(sc, g1, g2, n, iters, r0=0.3, ndir=4):
    u = n/np.linalg.norm(n); best = sc.overlap(g1,g2,u[None])[0]; r = r0; evals = 1
    for _ in range(iters):
        t1,t2 = tangent(u)
        dirs = [t1,-t1,t2,-t2] if ndir==4 else [t1,-t1,t2,-t2,(t1+t2)/1.414,(t1-t2)/1.414,(-t1+t2)/1.414,(-t1-t2)/1.414]
        C = np.array([u + r*d for d in dirs]); C /= np.linalg.norm(C,axis=1,keepdims=True)
        o = sc.overlap(g1,g2,C); evals += len(C)
        k = int(np.argmin(o))
        if o[k] < best: best, u = o[k], C[k]
        else: r *= 0.5
    return best, u, evals
rows=[]
for st in states:
    o = Oracle(raw); o.set_state(np.array(st["qpos"]), np.array(st["qvel"]), np.array(st["warm"])); o.set_ctrl(np.array(st["action"])); o.forward()
    sc = gc.Scene.from_oracle(model, o)
    for r in gc.check_contacts(sc, o.contacts()):
        if r["plane"]: continue
        g1,g2 = r["pair"]
        c = [c for c in o.contacts() if (c["geom1"],c["geom2"])==(g1,g2)]
        n = np.asarray(c[0]["normal"])
        out = {}
        for iters in (4, 8, 12):
            b,u,ev = refine(sc,g1,g2,n,iters)
            out[iters] = b/r["mtd"]
        rows.append((r["minimality"], r["consistency"], out))
m0 = np.array([r[0] for r in rows]); c0=np.array([r[1] for r in rows])
print("pairs", len(rows), "MPR: <=1.02 %.3f <=1.25 %.3f max %.2f | along-n overlap/mtd: <=1.02 %.3f max %.2f" % (np.mean(m0<=1.02), np.mean(m0<=1.25), m0.max(), np.mean(m0*c0<=1.02), (m0*c0).max()))
for iters in (4,8,12):
    m = np.array([r[2][iters] for r in rows])
    print("refined %2d iters (%d support pairs): <=1.02 %.3f <=1.25 %.3f max %.3f" % (iters, 1+4*iters, np.mean(m<=1.02), np.mean(m<=1.25), m.max()))
