import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools'); sys.path.insert(0,'/root/repo/tests')
import synth, helpers
from oracle import binding as ob
N=1_000_000; sh,cov=3,0
g=synth.scene(N); pods=ob.pack(sh,cov,g)
cam=helpers.default_camera(ob,1920,1080)
gt,mt=ob.gaussian_transform(sh_deg=0),ob.model_transform()
proj,tiles=ob.preprocess(sh,cov,pods,gt,mt,cam)
keys,idx=ob.build_keys(proj,tiles,120)
keys,idx=ob.sort_pairs(keys,idx)
tile=(keys>>np.uint64(32)).astype(np.int64)
D=len(keys); print('D',D)
p=proj[idx]
tx=(tile%120); ty=(tile//120)
mx=p['mx'].astype(np.float64); my=p['my'].astype(np.float64)
ca=p['ca'].astype(np.float64); cb=p['cb'].astype(np.float64); cc=p['cc'].astype(np.float64)
op=p['opacity'].astype(np.float64)
pmin=np.maximum(-np.log(255*op)-1e-3,-5.6)
def pmax(q2,q1,q0,lo,hi):
    t=np.clip(-0.5*q1/q2,lo,hi); return (q2*t+q1)*t+q0
def touches(rx0,rx1,ry0,ry1):
    dxl=mx-rx1; dxh=mx-rx0; dyl=my-ry1; dyh=my-ry0
    inx=(dxl<=0)&(dxh>=0); iny=(dyl<=0)&(dyh>=0)
    m0=pmax(cc,cb*dxl,ca*dxl*dxl,dyl,dyh); m1=pmax(cc,cb*dxh,ca*dxh*dxh,dyl,dyh)
    m2=pmax(ca,cb*dyl,cc*dyl*dyl,dxl,dxh); m3=pmax(ca,cb*dyh,cc*dyh*dyh,dxl,dxh)
    m=np.maximum(np.maximum(m0,m1),np.maximum(m2,m3))
    return (inx&iny)|~(m<pmin-0.1)
x0=tx*16+0.5; y0=ty*16+0.5
# position within tile list -> batch
start=np.searchsorted(tile,np.arange(8160)); pos=np.arange(D)-start[tile]; batch=pos//128
gid=tile*1000+batch   # (tile,batch) group id
def grp_sum(mask):
    u,inv=np.unique(gid,return_inverse=True); return np.bincount(inv,weights=mask,minlength=len(u))
# halves (current)
h=[touches(x0,x0+15,y0+8*k,y0+8*k+7) for k in range(2)]
cur=sum(grp_sum(m).sum() for m in h)
print('current wave-iterations (halves):',cur, 'per pair',cur/D)
# quadrants: wave w has quadrants (left,right) of half w
q={}
for qy in range(2):
    for qx in range(2):
        q[(qy,qx)]=touches(x0+8*qx,x0+8*qx+7,y0+8*qy,y0+8*qy+7)
new=0
for qy in range(2):
    a=grp_sum(q[(qy,0)]); b=grp_sum(q[(qy,1)]); new+=np.maximum(a,b).sum()
print('quadrant half-waves (max of 2 lists):',new,'ratio',new/cur, ' ideal sum/2:',sum(grp_sum(m).sum() for m in q.values())/2/cur)
# 8x4 blocks, quarter waves: wave w covers half w: 4 blocks: (bx in 0,1) x (by in 0,1) of size 8x4
e={}
new8=0
for hy in range(2):
    lists=[]
    for by in range(2):
        for bx in range(2):
            m=touches(x0+8*bx,x0+8*bx+7,y0+8*hy+4*by,y0+8*hy+4*by+3)
            lists.append(grp_sum(m))
    new8+=np.maximum.reduce(lists).sum()
    e[hy]=sum(l.sum() for l in lists)/4
print('8x4 quarter-waves (max of 4 lists):',new8,'ratio',new8/cur,' ideal:',sum(e.values())/cur)
# 4x4 blocks, eighth-waves (8 lanes x 2 px): wave w covers half w: 8 blocks (4 across x 2 down)
new16=0; ideal16=0
for hy in range(2):
    lists=[]
    for by in range(2):
        for bx in range(4):
            m=touches(x0+4*bx,x0+4*bx+3,y0+8*hy+4*by,y0+8*hy+4*by+3)
            lists.append(grp_sum(m))
    new16+=np.maximum.reduce(lists).sum(); ideal16+=sum(l.sum() for l in lists)/8
print('4x4 eighth-waves (max of 8 lists):',new16,'ratio',new16/cur,' ideal:',ideal16/cur)
# 8x2 blocks, eighth-waves
new82=0; ideal82=0
for hy in range(2):
    lists=[]
    for by in range(4):
        for bx in range(2):
            m=touches(x0+8*bx,x0+8*bx+7,y0+8*hy+2*by,y0+8*hy+2*by+1)
            lists.append(grp_sum(m))
    new82+=np.maximum.reduce(lists).sum(); ideal82+=sum(l.sum() for l in lists)/8
print('8x2 eighth-waves (max of 8 lists):',new82,'ratio',new82/cur,' ideal:',ideal82/cur)
