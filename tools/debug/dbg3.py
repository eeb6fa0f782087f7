import sys, subprocess
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np, oracle_lib as ol
pa=ol.pa
subprocess.check_call([sys.executable,'/root/repo/tools/gen_scenes.py','bathroom','--tris','60000','--res','192','108','--spp','16','--out','/tmp/bath.pbrt'],stdout=subprocess.DEVNULL)
sc=pa.Scene('/tmp/bath.pbrt'); ctx=pa.Context(sc)
ctx.render(count_work=True); img=sc.film_image(ctx.film()); cnt=ctx.counters()
rgbw,rcnt,_=ol.render(sc); ref=sc.film_image(rgbw)
print("image", ol.image_metrics(img,ref), "counts", {k:(cnt[k],rcnt.get(k)) for k in ('camera_rays','closest_rays','shadow_rays')})
ys,xs=np.mgrid[0:sc.height,0:sc.width]
xy=np.stack([xs.ravel(),ys.ravel()],1).astype(np.int32)
for s in (0,5):
    sn=np.full(len(xy),s,dtype=np.int32)
    d=ctx.li(xy,sn); r=ol.li(sc,xy,sn)
    e=np.linalg.norm(d-r,axis=1); bad=e>1e-4*(1+np.linalg.norm(r,axis=1))
    print("sample",s,"bad",bad.sum(),"of",len(bad),"mean dev",d.mean(),"mean ref",r.mean())
    idx=np.nonzero(bad)[0][:8]
    for i in idx: print("  px",xy[i],"dev",d[i],"ref",r[i])
