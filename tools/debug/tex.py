#!/usr/bin/env python3
"""GPU box: the texture path (row f2) scene by scene -- stage-level Texture::Evaluate vs the oracle for every node, then the
rendered image vs the reference fixture and vs the oracle.  One scene per process (tools/debug/tex.sh) so that a fault in one
does not hide the others.  usage: tex.py <scene name>"""
import os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
import edge_scenes as es

pa = ol.pa


def queries(sc, n, seed):
    rng = np.random.default_rng(seed)
    q = np.zeros(n, dtype=pa.TEX_QUERY_DTYPE)
    q["p"] = rng.uniform(-4, 4, (n, 3)); q["uv"] = rng.uniform(-1, 3, (n, 2))
    s = 10.0 ** rng.uniform(-4, -0.5, (n, 1))
    q["dpdx"] = rng.normal(size=(n, 3)) * s; q["dpdy"] = rng.normal(size=(n, 3)) * s
    for k in ("dudx", "dvdx", "dudy", "dvdy"):
        q[k] = rng.normal(size=n) * s[:, 0]
    z = rng.random(n) < 0.15   # no differentials (every bounce after the first, alpha tests)
    for k in ("dpdx", "dpdy", "dudx", "dvdx", "dudy", "dvdy"):
        q[k][z] = 0
    return q


def main():
    name = sys.argv[1]
    sc = pa.Scene(text=es.scene(name))
    print("== %s: %s" % (name, {k: v for k, v in sc.info.items() if k in ("n_tris", "n_materials", "n_textures", "n_images", "n_textured_materials", "n_masked_meshes")}), flush=True)
    ctx = pa.Context(sc)
    q = queries(sc, 2048, 1)
    for node in range(sc.info["n_textures"]):
        dev = ctx.texture_eval(node, q)
        ref = ol.texture_eval(sc, node, q)
        close = np.isclose(dev, ref, rtol=2e-4, atol=2e-6).all(axis=1)
        print("  node %2d: exact %.4f close %.4f maxabs %.3e" % (node, np.mean(np.all(dev.view(np.uint32) == ref.view(np.uint32), axis=1)), close.mean(), float(np.abs(dev - ref).max())), flush=True)
    ctx.timing_enable(True)
    t0 = time.time()
    ctx.render()
    img = sc.film_image(ctx.film())
    t1 = time.time()
    ref = pa.read_pfm(os.path.join(ROOT, "tests", "golden", "edge_%s.pfm" % name))
    frac, relmse = ol.image_metrics(img, ref)
    d = np.abs(img - ref).max(-1)
    print("  image vs reference: frac %.4f relmse %.3e exact %.4f maxdiff %.3e mean %.5f/%.5f (%.2f s)" % (frac, relmse, np.mean(d == 0), d.max(), img.mean(), ref.mean(), t1 - t0), flush=True)
    print("  timing:", {k: round(v[0], 2) for k, v in ctx.timing().items()} if hasattr(ctx, "timing") else "", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
