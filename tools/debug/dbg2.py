import sys
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np, oracle_lib as ol
pa=ol.pa
sc=pa.Scene('/root/repo/scenes/cornell.pbrt'); ctx=pa.Context(sc)
ctx.film_clear(); ctx.render(); a=ctx.film()
print("weights unique", np.unique(a[...,3], return_counts=True))
ref,_,_=ol.render(sc); print("vs oracle", ol.image_metrics(sc.film_image(a), sc.film_image(ref)))
acc=np.zeros_like(a)
for r in range(3):
    ctx.film_clear(); ctx.render(rank=r, world=3); acc+=ctx.film()
print("sharded weights", np.unique(acc[...,3], return_counts=True), "diff px", (acc!=a).any(-1).sum())
