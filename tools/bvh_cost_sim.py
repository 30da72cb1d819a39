"""CPU model of the closest-hit walk on the Cornell scene: inner-node visits and triangle tests per
ray for (a) a sweep-SAH tree over the triangles, (b) the same after early split clipping (each
triangle cut along its box's longest axis until the pieces' boxes are small; pieces refer to the
whole triangle), (c) the product's builder through tests/native/bvh_probe.cpp.
Result (600 random rays inside the room): 7.5 node visits + 2.5 triangle tests per ray for (a) and
for (c) with leaves of two; splitting only adds node visits (8.3-11.8) for 2.7-2.0 triangle tests.
usage: python tools/bvh_cost_sim.py"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from raytracer_amd import scenes
rng=np.random.default_rng(1)

def tri_boxes(tris):
    v=tris.reshape(-1,3,3); return v.min(1), v.max(1)

def clip_poly(poly, axis, pos, keep_low):
    out=[]
    n=len(poly)
    for i in range(n):
        a=poly[i]; b=poly[(i+1)%n]
        ina = a[axis]<=pos if keep_low else a[axis]>=pos
        inb = b[axis]<=pos if keep_low else b[axis]>=pos
        if ina: out.append(a)
        if ina!=inb:
            t=(pos-a[axis])/(b[axis]-a[axis]); p=a+t*(b-a); p[axis]=pos; out.append(p)
    return out

def fragments(tris, max_extent):
    """early split clipping: split each triangle's polygon along its box's longest axis until box extent<=max_extent"""
    frags=[]  # (lo,hi,tri)
    for ti,t in enumerate(tris.reshape(-1,3,3)):
        stack=[[p.astype(np.float64) for p in t]]
        while stack:
            poly=stack.pop()
            P=np.array(poly); lo=P.min(0); hi=P.max(0); ext=hi-lo
            ax=int(np.argmax(ext))
            if ext[ax]<=max_extent or len(poly)<3:
                frags.append((lo,hi,ti)); continue
            mid=0.5*(lo[ax]+hi[ax])
            a=clip_poly(poly,ax,mid,True); b=clip_poly(poly,ax,mid,False)
            if len(a)>=3: stack.append(a)
            if len(b)>=3: stack.append(b)
    return frags

class Node: pass
def build(lo,hi,ids,max_leaf=2):
    idx=np.arange(len(ids))
    def rec(idx):
        n=Node(); n.lo=lo[idx].min(0); n.hi=hi[idx].max(0)
        if len(idx)<=max_leaf: n.leaf=idx; return n
        c=0.5*(lo[idx]+hi[idx]); best=None
        for ax in range(3):
            o=idx[np.argsort(c[:,ax],kind='stable')]
            L=[];  
            l0=np.minimum.accumulate(lo[o],0); h0=np.maximum.accumulate(hi[o],0)
            l1=np.minimum.accumulate(lo[o][::-1],0)[::-1]; h1=np.maximum.accumulate(hi[o][::-1],0)[::-1]
            def ha(l,h):
                d=h-l; return d[:,0]*d[:,1]+d[:,1]*d[:,2]+d[:,2]*d[:,0]
            k=np.arange(1,len(o))
            cost=ha(l0[:-1],h0[:-1])*k+ha(l1[1:],h1[1:])*(len(o)-k)
            j=int(np.argmin(cost))
            if best is None or cost[j]<best[0]: best=(cost[j],o,j+1)
        _,o,m=best
        n.leaf=None; n.l=rec(o[:m]); n.r=rec(o[m:]); return n
    return rec(idx)

def tri_t(tri,o,d):
    a,b,c=tri; ba=b-a; ca=c-a; n=np.cross(ba,ca); roa=o-a; q=np.cross(roa,d)
    dn=np.dot(d,n)
    if dn==0: return np.inf
    inv=1/dn; u=inv*-np.dot(q,ca); v=inv*np.dot(q,ba)
    if u<0 or u>1 or v<0 or (u+v)>1: return np.inf
    return inv*-np.dot(n,roa)

def slab(lo,hi,o,inv,tmax):
    t0=(lo-o)*inv; t1=(hi-o)*inv
    tn=max(np.minimum(t0,t1).max(),0); tf=min(np.maximum(t0,t1).min(),tmax)
    return tn<=tf, tn

def trace(root,tris3,ids,o,d):
    inv=1/np.where(np.abs(d)<1e-20,1e-20,d)
    best=np.inf; nn=0; nt=0; stack=[root]
    while stack:
        n=stack.pop()
        if n.leaf is not None:
            for k in n.leaf:
                nt+=1; t=tri_t(tris3[ids[k]],o,d)
                if 1e-3<t<best: best=t
            continue
        nn+=1
        hl,tl=slab(n.l.lo,n.l.hi,o,inv,best); hr,tr=slab(n.r.lo,n.r.hi,o,inv,best)
        if hl and hr:
            if tr<tl: stack.append(n.l); stack.append(n.r)
            else: stack.append(n.r); stack.append(n.l)
        elif hl: stack.append(n.l)
        elif hr: stack.append(n.r)
    return nn,nt,best

sc=scenes.cornell()
tris=sc.triangles.astype(np.float64)
tris3=tris.reshape(-1,3,3)
# rays: random origins on surfaces via first tracing random rays, then bounce rays from hit points
rays=scenes.random_rays(600,seed=5,origin_box=((-0.9,0.1,-0.9),(0.9,1.9,0.9))).astype(np.float64)
def evaluate(frs,label):
    lo=np.array([f[0] for f in frs]); hi=np.array([f[1] for f in frs]); ids=np.array([f[2] for f in frs])
    pad=1e-5; root=build(lo-pad,hi+pad,ids)
    def count(n): return 1 if n.leaf is not None else 1+count(n.l)+count(n.r)
    tn=tt=0; bests=[]
    for r in rays:
        a,b,t=trace(root,tris3,ids,r[:3],r[3:]); tn+=a; tt+=b; bests.append(t)
    print(f"{label:28s} refs {len(frs):4d} treenodes {count(root):4d} nodes/ray {tn/len(rays):6.2f} tris/ray {tt/len(rays):6.2f} cost {(45*tn+60*tt)/len(rays):7.1f}")
    return np.array(bests)
lo,hi=tri_boxes(tris)
base=evaluate([(lo[i],hi[i],i) for i in range(len(lo))],"no split")
for me in (1.5,1.0,0.7,0.5,0.35):
    b=evaluate(fragments(tris,me),f"split extent<={me}")
    assert np.array_equal(b,base)

# the product's builder through the test probe
import ctypes as C, subprocess, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); src=ROOT+'/raytracer_amd/csrc'; out='/tmp/libbvh_probe_sim.so'
subprocess.check_call(["g++","-O2","-std=c++17","-ffp-contract=off","-fPIC","-shared","-I",src,ROOT+"/tests/native/bvh_probe.cpp",src+"/bvh_build.cpp","-o",out])
lib=C.CDLL(out); lib.bvh_probe.argtypes=[C.c_void_p,C.c_size_t,C.c_uint,C.c_uint,C.c_void_p,C.c_void_p,C.c_void_p,C.c_void_p]
def pbuild(tris,max_leaf):
    t=np.ascontiguousarray(tris,np.float32).reshape(-1,9)
    n=lib.bvh_probe(t.ctypes.data,len(t),max_leaf,2,None,None,None,None)
    nodes=np.zeros((n,16),np.float32); order=np.zeros(len(t),np.uint32); d=C.c_uint32(); s=C.c_uint32()
    lib.bvh_probe(t.ctypes.data,len(t),max_leaf,2,nodes.ctypes.data,order.ctypes.data,C.byref(d),C.byref(s))
    return nodes,order
def ptrace(nodes,order,o,d):
    kids=nodes[:,12:14].copy().view(np.int32)
    inv=1/np.where(np.abs(d)<1e-20,1e-20,d)
    best=np.inf; nn=nt=0; stack=[0]
    while stack:
        c=stack.pop()
        if c<0:
            v=~c; first,cnt=v>>3,(v&7)+1
            for k in range(first,first+cnt):
                nt+=1; t=tri_t(tris3[order[k]],o,d)
                if 1e-3<t<best: best=t
            continue
        q=nodes[c].astype(np.float64); nn+=1
        h0,t0=slab(np.array([q[0],q[2],q[8]]),np.array([q[1],q[3],q[9]]),o,inv,best)
        h1,t1=slab(np.array([q[4],q[6],q[10]]),np.array([q[5],q[7],q[11]]),o,inv,best)
        c0,c1=int(kids[c,0]),int(kids[c,1])
        if h0 and h1:
            if t1<t0: stack.append(c0); stack.append(c1)
            else: stack.append(c1); stack.append(c0)
        elif h0: stack.append(c0)
        elif h1: stack.append(c1)
    return nn,nt,best
for ml in (1,2,4):
    nodes,order=pbuild(tris,ml)
    tn=tt=0
    for r in rays:
        a,b,t=ptrace(nodes,order,r[:3],r[3:]); tn+=a; tt+=b
    print(f"product builder max_leaf {ml}: treenodes {len(nodes)} nodes/ray {tn/len(rays):.2f} tris/ray {tt/len(rays):.2f} cost {(45*tn+60*tt)/len(rays):.1f}")
