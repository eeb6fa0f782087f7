import sys, subprocess, re
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np, oracle_lib as ol
pa=ol.pa
subprocess.check_call([sys.executable,'/root/repo/tools/gen_scenes.py','bathroom','--tris','60000','--res','192','108','--spp','16','--out','/tmp/bath.pbrt'],stdout=subprocess.DEVNULL)
text=open('/tmp/bath.pbrt').read().replace('bath_geo/','/tmp/bath_geo/')
ys,xs=np.mgrid[0:108,0:192]; xy=np.stack([xs.ravel(),ys.ravel()],1).astype(np.int32)
for md in (1,2,3,4,6,30):
    t=re.sub(r'"integer maxdepth" \[\d+\]','"integer maxdepth" [%d]'%md,text)
    sc=pa.Scene(text=t); ctx=pa.Context(sc)
    nb=0; nz=0; tot=0
    for s in range(4):
        sn=np.full(len(xy),s,dtype=np.int32)
        d=ctx.li(xy,sn); r=ol.li(sc,xy,sn)
        e=np.linalg.norm(d-r,axis=1); bad=e>1e-4*(1+np.linalg.norm(r,axis=1))
        nb+=bad.sum(); nz+=(e>0).sum(); tot+=len(e)
        if md==2 and s==0:
            for i in np.nonzero(bad)[0][:6]: print("   md2 px",xy[i],"dev",d[i],"ref",r[i])
    print("maxdepth",md,"bad",nb,"nonzero-diff",nz,"of",tot)
    ctx.close()
# first-hit + secondary hit exactness on this scene
sc=pa.Scene(text=text); ctx=pa.Context(sc)
rng=np.random.default_rng(3); n=60000
pxy=np.stack([rng.integers(0,192,n),rng.integers(0,108,n)],1).astype(np.int32); ss=rng.integers(0,16,n).astype(np.int32)
rays,_=ol.camera_rays(sc,pxy,ss); dh=ctx.intersect(rays); rh,_=ol.intersect(sc,rays)
print("primary prim mismatches", (dh['prim']!=rh['prim']).sum(), "t mismatches", (dh['t']!=rh['t']).sum())
hit=rh['prim']>=0
o=rays['o'][hit]+rays['d'][hit]*rh['t'][hit][:,None]+rh['n'][hit]*1e-3
d=rng.standard_normal(o.shape).astype(np.float32); d/=np.linalg.norm(d,axis=1)[:,None]
r2=np.zeros(len(o),dtype=pa.RAY_DTYPE); r2['o']=o; r2['d']=d; r2['tmax']=np.inf
dh2=ctx.intersect(r2); rh2,_=ol.intersect(sc,r2)
mm=dh2['prim']!=rh2['prim']; print("secondary prim mismatches", mm.sum(), "of", len(r2), "t mismatches", (dh2['t']!=rh2['t']).sum())
for i in np.nonzero(mm)[0][:5]: print("   ", dh2[i], rh2[i])
