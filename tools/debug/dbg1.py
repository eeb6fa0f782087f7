import sys, os
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np, oracle_lib as ol
pa=ol.pa
sc=pa.Scene('/root/repo/scenes/materials.pbrt'); ctx=pa.Context(sc)
def whole():
    ctx.film_clear(); ctx.render(); return ctx.film()
a=whole(); b=whole()
print("whole vs whole equal:", np.array_equal(a.view(np.uint32), b.view(np.uint32)), np.abs(a-b).max(), (a!=b).any(-1).sum())
acc=np.zeros_like(a)
for r in range(3):
    ctx.film_clear(); ctx.render(rank=r, world=3); acc+=ctx.film()
d=(acc!=a).any(-1); print("sharded vs whole differing px", d.sum(), np.abs(acc-a).max())
ys,xs=np.nonzero(d); print(list(zip(xs[:10],ys[:10])))
if d.sum():
    y,x=ys[0],xs[0]; print(a[y,x], acc[y,x])
ref,_,_=ol.render(sc); print("vs oracle", ol.image_metrics(sc.film_image(a), sc.film_image(ref)))
# per-sample li determinism
rng=np.random.default_rng(1); xy=np.stack([rng.integers(0,sc.width,20000),rng.integers(0,sc.height,20000)],1).astype(np.int32); s=rng.integers(0,16,20000).astype(np.int32)
l1=ctx.li(xy,s); l2=ctx.li(xy,s); print("li repeat equal", np.array_equal(l1,l2), np.abs(l1-l2).max(), (l1!=l2).any(-1).sum())
lo=ol.li(sc,xy,s); e=np.linalg.norm(l1-lo,axis=1); print("li vs oracle bad", (e>1e-4*(1+np.linalg.norm(lo,axis=1))).sum())
