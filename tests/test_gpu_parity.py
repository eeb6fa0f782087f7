"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs."""
import os

import numpy as np
import pytest

import oracle_lib as ol

pa = ol.pa
ROOT = ol.ROOT
pytestmark = pytest.mark.gpu

SCENES = ["cornell.pbrt", "materials.pbrt"]
# every traversal kernel instance the library ships runs the parity tests (the variables are read by mi_scene_upload):
# traversal layouts that ship: bvh4q = the default (64-byte quantised BVH4 over the library's own topology of the reference's leaves, hot nodes in LDS); general = full-precision
# 128-byte nodes; bvh4q-cold = reference node order, nothing in LDS, AND the reference's own interior nodes (PBRT_AMD_TREE=reference: the tree exactly as handed over)
#  bvh4q-reftree = the reference's own interior nodes WITH hot nodes in LDS, the combination bvh4q-cold left out
TRACE_MODES = {"bvh4q": {}, "general": {"PBRT_AMD_TRACE": "general"}, "bvh4q-cold": {"PBRT_AMD_HOT": "0", "PBRT_AMD_TREE": "reference"}, "bvh4q-reftree": {"PBRT_AMD_TREE": "reference"}}


def _report(key, **values):
    """measured agreement figures (fractions of bit-identical values etc.) -> one JSON line each in $PBRT_AMD_PARITY_REPORT: the numbers DESIGN.md quotes"""
    path = os.environ.get("PBRT_AMD_PARITY_REPORT")
    if path:
        import json
        with open(path, "a") as f:
            f.write(json.dumps({"test": key, **{k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in values.items()}}) + "\n")


# (not part of the parametrised matrix) the default topology without hot nodes: what test_hot_nodes_in_lds_change_nothing compares the default with
EXTRA_MODES = {"bvh4q-nohot": {"PBRT_AMD_HOT": "0"}}


def make_ctx(sc, mode="bvh4q", **kw):
    env = TRACE_MODES[mode] if mode in TRACE_MODES else EXTRA_MODES[mode]
    saved = {k: os.environ.get(k) for k in ("PBRT_AMD_TRACE", "PBRT_AMD_TREE", "PBRT_AMD_HOT")}
    for k in saved:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        return pa.Context(sc, **kw)
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


@pytest.fixture(scope="module", params=[(s, m) for s in SCENES for m in TRACE_MODES], ids=lambda p: "%s-%s" % (p[0].split(".")[0], p[1]))
def pair(request):
    sc = pa.Scene(os.path.join(ROOT, "scenes", request.param[0]))
    ctx = make_ctx(sc, request.param[1])
    yield sc, ctx
    ctx.close()


def _pixels(sc, n, seed=7):
    rng = np.random.default_rng(seed)
    xy = np.stack([rng.integers(0, sc.width, n), rng.integers(0, sc.height, n)], 1).astype(np.int32)
    s = rng.integers(0, sc.info["spp"], n).astype(np.int32)
    return xy, s


def test_sobol_bit_exact(pair):
    sc, ctx = pair
    for (px, py) in [(0, 0), (1, 0), (sc.width - 1, sc.height - 1), (17, 5)]:
        dev, didx = ctx.sobol(px, py, sc.info["spp"], 64)
        ref, ridx = ol.sobol(sc, px, py, sc.info["spp"], 64)
        assert np.array_equal(didx, ridx)
        assert np.array_equal(dev.view(np.uint32), ref.view(np.uint32))


def test_halton_bit_exact():
    """HaltonSampler on the device: sample indices and the first 200 dimensions vs the oracle, and the Get1D stream of the
    reference's own class (tests/golden/ref_vectors.npz: halton_sampler) -- bit for bit."""
    sc = pa.Scene(text='Film "image" "integer xresolution" [400] "integer yresolution" [300] "string filename" "x.pfm"\n'
                       'Sampler "halton" "integer pixelsamples" [6]\nWorldBegin\nWorldEnd\n')
    ctx = pa.Context(sc)
    rows = np.load(os.path.join(ROOT, "tests", "golden", "ref_vectors.npz"))["halton_sampler"]
    for (px, py) in sorted(set(zip(rows["px"].tolist(), rows["py"].tolist()))):
        sel = rows[(rows["px"] == px) & (rows["py"] == py)]
        dev, didx = ctx.sobol(px, py, 6, 200)
        ref, ridx = ol.sobol(sc, px, py, 6, 200)
        assert np.array_equal(didx, ridx)
        assert np.array_equal(dev.view(np.uint32), ref.view(np.uint32))
        assert np.array_equal(dev[sel["s"], :24].view(np.uint32), sel["u"].view(np.uint32)), (px, py)
    ctx.close()


def test_camera_rays_bit_exact(pair):
    sc, ctx = pair
    xy, s = _pixels(sc, 4096)
    rd, pfd = ctx.camera_rays(xy, s)
    rr, pfr = ol.camera_rays(sc, xy, s)
    assert np.array_equal(pfd.view(np.uint32), pfr.view(np.uint32))
    for k in ("o", "d"):
        assert np.array_equal(rd[k].view(np.uint32), rr[k].view(np.uint32)), k


def test_hot_nodes_in_lds_change_nothing():
    """The traversal blocks keep the scene's most visited BVH4Q nodes (hot-node probe at upload: nodes renumbered, nodesq[0 .. n_hot) staged into LDS) --
    that decides where a node is read from, never what is read: per-sample radiance, hits and work counters equal the reference-order / no-LDS run
    bit for bit (the film up to the order of its few atomic adds: box-filter samples that land exactly on a pixel edge)."""
    sc = pa.Scene(os.path.join(ROOT, "scenes", "materials.pbrt"))
    res = {}
    for mode in ("bvh4q", "bvh4q-nohot"):
        ctx = make_ctx(sc, mode)
        ti = ctx.trace_info()
        ctx.counters_reset()
        ctx.render(count_work=True)
        film = ctx.film().copy()
        cnt = ctx.counters()
        xy, s = _pixels(sc, 20000, seed=5)
        rays, _ = ol.camera_rays(sc, xy, s)
        hits = ctx.intersect(rays)
        li = ctx.li(xy[:6000], s[:6000])
        res[mode] = (ti, film, cnt, hits, li)
        ctx.close()
    (ti, film, cnt, hits, li), (ti0, film0, cnt0, hits0, li0) = res["bvh4q"], res["bvh4q-nohot"]
    assert ti["mode"] == 5 and ti0["mode"] == 5
    assert ti["hot_nodes"] > 0 and 0.2 < ti["hot_probe_share"] <= 1.0 and ti0["hot_nodes"] == 0
    assert cnt["nodes_hot_closest"] > 0 and cnt["nodes_hot_any"] > 0 and cnt0["nodes_hot_closest"] == 0 and cnt0["nodes_hot_any"] == 0
    for k in ("camera_rays", "closest_rays", "shadow_rays", "trace_guard_trips"):
        assert cnt[k] == cnt0[k], k
    # node / triangle fetch counts: a lane with a parked leaf (PT_PEND_LEAF) steps through nodes with the tMax of the moment, so a few of its
    # visits depend on when the wave ran its leaf phase, i.e. on which rays shared the wave -- the counts wobble in the fourth digit, the hits never
    # (any-hit rays end at whichever occluder they meet first, so their counts wobble more: 0.58 % measured on the MI355X in round 5, profiles/r05_v_*)
    for k, tol in (("nodes_closest", 5e-3), ("tris_closest", 5e-3), ("nodes_any", 2e-2), ("tris_any", 2e-2)):
        assert abs(cnt[k] - cnt0[k]) <= tol * cnt0[k], k
    assert np.array_equal(li.view(np.uint32), li0.view(np.uint32))
    assert np.allclose(film, film0, rtol=1e-6, atol=0)
    for k in hits.dtype.names:
        assert np.array_equal(hits[k], hits0[k]), k


def test_closest_hit_matches_reference_traversal(pair):
    sc, ctx = pair
    xy, s = _pixels(sc, 20000, seed=3)
    rays, _ = ol.camera_rays(sc, xy, s)
    dh = ctx.intersect(rays)
    rh, _ = ol.intersect(sc, rays)
    assert np.array_equal(dh["prim"], rh["prim"])
    hit = rh["prim"] >= 0
    assert hit.sum() > 1000
    for k in ("t", "b0", "b1", "b2"):
        assert np.array_equal(dh[k][hit].view(np.uint32), rh[k][hit].view(np.uint32)), k
    assert np.array_equal(dh["n"][hit].view(np.uint32), rh["n"][hit].view(np.uint32))
    # secondary (incoherent) rays: bounce off the first hit along the oriented normal hemisphere
    rng = np.random.default_rng(11)
    o = rays["o"][hit] + rays["d"][hit] * rh["t"][hit][:, None] + rh["n"][hit] * 1e-2
    d = rng.standard_normal(o.shape).astype(np.float32)
    d /= np.linalg.norm(d, axis=1)[:, None]
    r2 = np.zeros(len(o), dtype=pa.RAY_DTYPE)
    r2["o"] = o; r2["d"] = d; r2["tmax"] = np.inf
    dh2 = ctx.intersect(r2)
    rh2, _ = ol.intersect(sc, r2)
    assert np.array_equal(dh2["prim"], rh2["prim"])
    h2 = rh2["prim"] >= 0
    assert np.array_equal(dh2["t"][h2].view(np.uint32), rh2["t"][h2].view(np.uint32))
    # any-hit with finite segments
    r2["tmax"] = rng.uniform(0.5, 600.0, len(o)).astype(np.float32)
    assert np.array_equal(ctx.intersect_p(r2), ol.intersect_p(sc, r2)[0])
    assert ctx.counters()["trace_guard_trips"] == 0
    # rays with exactly-zero direction components (1/d = +-inf in Bounds3::IntersectP; Sobol' values like 0.5 produce them on
    # axis-aligned surfaces) incl. -0 and origins that sit exactly on bounding planes (vertex coordinates of the scene)
    n3 = 6000
    r3 = np.zeros(n3, dtype=pa.RAY_DTYPE)
    d3 = rng.standard_normal((n3, 3)).astype(np.float32)
    kill = rng.integers(0, 3, n3)
    d3[np.arange(n3), kill] = np.where(rng.random(n3) < 0.5, np.float32(0.0), np.float32(-0.0))
    two = rng.random(n3) < 0.3   # a third of them: two zero components (axis-parallel rays)
    d3[np.arange(n3)[two], (kill[two] + 1) % 3] = 0
    d3 /= np.linalg.norm(d3, axis=1)[:, None]
    o3 = o[rng.integers(0, len(o), n3)].copy()
    r3["o"] = o3; r3["d"] = d3; r3["tmax"] = np.inf
    dh3 = ctx.intersect(r3)
    rh3, _ = ol.intersect(sc, r3)
    assert np.array_equal(dh3["prim"], rh3["prim"]), int((dh3["prim"] != rh3["prim"]).sum())
    h3 = rh3["prim"] >= 0
    assert h3.sum() > 500
    assert np.array_equal(dh3["t"][h3].view(np.uint32), rh3["t"][h3].view(np.uint32))
    # the same directions from origins that lie exactly ON bounding planes (coordinates of hit points on the axis-aligned walls):
    # there the reference's slab test evaluates 0 * inf = NaN, and a NaN on the x axis makes Bounds3::IntersectP REJECT the box
    # (geometry.h:1412-1438: every comparison with the NaN tMin is false, the final tMin < ray.tMax too) although the ray runs
    # inside its face -- the reference then misses triangles the device (whose box test skips a NaN axis: a superset) still
    # finds.  Stated deviation: in this measure-zero configuration the device may report a hit where the
    # reference reports none or a farther one; never the other way round.
    src = rays["o"][hit] + rays["d"][hit] * rh["t"][hit][:, None]
    r3["o"] = src[rng.integers(0, len(src), n3)]
    dh4 = ctx.intersect(r3)
    rh4, _ = ol.intersect(sc, r3)
    differ = dh4["prim"] != rh4["prim"]
    tdev = np.where(dh4["prim"] >= 0, dh4["t"], np.inf); tref = np.where(rh4["prim"] >= 0, rh4["t"], np.inf)
    assert np.all(tdev[differ] <= tref[differ]), "device missed a hit the reference finds"
    assert differ.mean() < 0.05, float(differ.mean())
    same = ~differ & (rh4["prim"] >= 0)
    assert np.array_equal(dh4["t"][same].view(np.uint32), rh4["t"][same].view(np.uint32))


def test_li_per_sample(pair):
    """PathIntegrator::Li per camera sample.  Tolerance: |dL| <= 1e-4 * (1 + |L|) per sample for >= 99.5 % of samples
    (discontinuous decisions may flip on a few paths: libm last-ulp differences)."""
    sc, ctx = pair
    xy, s = _pixels(sc, 6000, seed=5)
    dev = ctx.li(xy, s)
    ref = ol.li(sc, xy, s)
    err = np.linalg.norm(dev - ref, axis=1)
    ok = err <= 1e-4 * (1 + np.linalg.norm(ref, axis=1))
    _report("li_per_sample", samples=len(ref), within_tol=ok.mean(), bit_identical=(dev.view(np.uint32) == ref.view(np.uint32)).all(1).mean(), max_err=err.max())
    assert ok.mean() >= 0.995, (ok.mean(), err.max())
    assert abs(dev.mean() - ref.mean()) <= 2e-3 * ref.mean()


def test_render_image_vs_oracle(pair):
    """Whole image, stated tolerance: per-pixel L2 <= 1e-3 (1 + |ref|) for >= 99.5 % of pixels, relMSE <= 1e-4."""
    sc, ctx = pair
    ctx.film_clear(); ctx.counters_reset()
    ctx.render(count_work=True)
    img = sc.film_image(ctx.film())
    ref_rgbw, rcnt, _ = ol.render(sc)
    ref = sc.film_image(ref_rgbw)
    frac, relmse = ol.image_metrics(img, ref)
    _report("render_image_vs_oracle", pixels=int(img.shape[0] * img.shape[1]), within_tol=frac, relMSE=relmse, bit_identical_pixels=(img.view(np.uint32) == ref.view(np.uint32)).all(-1).mean())
    assert frac >= 0.995 and relmse <= 1e-4, (frac, relmse)
    cnt = ctx.counters()
    assert cnt["camera_rays"] == rcnt["camera_rays"]
    assert cnt["trace_guard_trips"] == 0
    # ray counts follow the reference's issue conditions up to decision flips
    assert abs(cnt["closest_rays"] - rcnt["closest_rays"]) <= 2e-3 * rcnt["closest_rays"]
    assert abs(cnt["shadow_rays"] - rcnt["shadow_rays"]) <= 2e-3 * rcnt["shadow_rays"]


def test_tile_sharding_is_exact(pair):
    """Rendering the tiles of rank r of N on separate films and summing == single-GPU film, bit for bit (box filter)."""
    sc, ctx = pair
    ctx.film_clear(); ctx.render()
    whole = ctx.film()
    acc = np.zeros_like(whole)
    for r in range(3):
        ctx.film_clear(); ctx.render(rank=r, world=3)
        acc += ctx.film()
    # pixels that only received their own samples (filterWeightSum == spp) must agree bit for bit; a pixel that also
    # got a sample landing exactly on its edge from a neighbour is summed through float atomics (order not fixed)
    own_only = whole[..., 3] == sc.info["spp"]
    assert own_only.mean() > 0.98
    assert np.array_equal(acc[own_only].view(np.uint32), whole[own_only].view(np.uint32))
    assert np.allclose(acc, whole, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("flt", ["box", "gaussian"])
def test_film_gather_of_two_contexts_equals_the_single_render(flt):
    """mi_film_gather (the exchange step of the tile-sharded render): contexts render the tiles of rank r of N -- the calls return before the GPU
    has finished (mi_render is asynchronous), so the renders overlap -- and the gather adds what each context's samples can REACH (its tiles + the
    filter's ring, packed; round 5) into the root film, in context order.  On this one-GPU box the contexts share device 0, so the packed lists are
    handed over directly (RCCL refuses duplicate devices); with one context per GPU the same call issues one group of ncclSend / ncclRecv.  A wide
    filter (gaussian, radius 2: overlapping footprints across tile borders) and three contexts cover the ring and the order of the adds."""
    text = open(os.path.join(ROOT, "scenes", "materials.pbrt")).read()
    if flt == "gaussian":
        text = text.replace("WorldBegin", 'PixelFilter "gaussian" "float xwidth" [2] "float ywidth" [2]\nWorldBegin', 1)
    sc = pa.Scene(text=text)
    whole = pa.Context(sc); whole.render(); ref = whole.film(); whole.close()
    world = 2 if flt == "box" else 3
    ctxs = [pa.Context(sc) for _ in range(world)]
    for r, c in enumerate(ctxs):
        c.render(rank=r, world=world, sync=False)
    pa.film_gather(ctxs, root=0)
    got = ctxs[0].film()
    if flt == "box":
        own_only = ref[..., 3] == sc.info["spp"]
        assert np.array_equal(got[own_only].view(np.uint32), ref[own_only].view(np.uint32))
    assert np.allclose(got, ref, rtol=1e-6, atol=1e-7) if flt == "box" else np.allclose(got, ref, rtol=1e-5, atol=1e-6)   # (wide filters: every pixel is a float-atomic sum)
    assert np.array_equal(got[..., 3] != 0, ref[..., 3] != 0)   # no pixel's contributions were left behind
    # later frames of the same sharding: the reach lists, index and exchange buffers stay in the senders' contexts -- no rebuild, allocation or upload (round 5 redid
    # all of it every frame); the counter moves again only when the sharding changes
    assert [c.counters()["film_gather_builds"] for c in ctxs] == [0] + [1] * (world - 1)
    for _ in range(2):
        for r, c in enumerate(ctxs):
            c.film_clear(); c.render(rank=r, world=world, sync=False)
        pa.film_gather(ctxs, root=0)
        again = ctxs[0].film()
        assert np.allclose(again, ref, rtol=1e-5, atol=1e-6) and np.array_equal(again[..., 3] != 0, ref[..., 3] != 0)   # and the frame is the same frame
        if flt == "box":
            assert np.array_equal(again[own_only].view(np.uint32), ref[own_only].view(np.uint32))
    assert [c.counters()["film_gather_builds"] for c in ctxs] == [0] + [1] * (world - 1)
    for c in ctxs:
        c.close()


def test_film_gather_adds_the_whole_film_when_a_context_holds_more_than_one_shard():
    """ The sparse gather sends what ONE sharding can reach.  A context that accumulated two shards into its film (no clear in between), or whose sharding
    changed, must not lose the pixels outside its last shard: the library tracks what each film holds and adds such a film whole."""
    sc = pa.Scene(os.path.join(ROOT, "scenes", "cornell.pbrt"))
    whole = pa.Context(sc); whole.render(); ref = whole.film(); whole.close()
    a, b = pa.Context(sc), pa.Context(sc)
    a.render(rank=0, world=3, sync=False)
    b.render(rank=1, world=3, sync=False)
    b.render(rank=2, world=3, sync=False)   # a second shard into the same film: its last mi_render alone says "rank 2 of 3"
    pa.film_gather([a, b], root=0)
    got = a.film()
    assert np.array_equal(got[..., 3] != 0, ref[..., 3] != 0)
    own_only = ref[..., 3] == sc.info["spp"]
    assert np.array_equal(got[own_only].view(np.uint32), ref[own_only].view(np.uint32))
    # after a clear the context is a single shard again and goes back to the sparse form (one build), with the same result
    for c in (a, b):
        c.film_clear()
    a.render(rank=0, world=2, sync=False); b.render(rank=1, world=2, sync=False)
    pa.film_gather([a, b], root=0)
    assert b.counters()["film_gather_builds"] == 1
    got = a.film()
    assert np.array_equal(got[own_only].view(np.uint32), ref[own_only].view(np.uint32)) and np.array_equal(got[..., 3] != 0, ref[..., 3] != 0)
    a.close(); b.close()


# ---------------------------------------------------------------- against the committed reference fixtures
G = os.path.join(ROOT, "tests", "golden")


def test_device_triangle_intersect_matches_reference_vectors():
    """The device watertight test on the reference's Triangle.* unit-test constructions (BadCases, Reintersect, vertex/edge rays)."""
    t = np.load(os.path.join(G, "ref_vectors.npz"))["triangles"]
    rays = np.zeros(len(t), dtype=pa.RAY_DTYPE)
    rays["o"] = t["o"]; rays["d"] = t["d"]; rays["tmax"] = t["tmax"]
    h = pa.triangle_intersect(t["p"], rays)
    assert np.array_equal(h["prim"] >= 0, t["hit"] == 1)
    hit = t["hit"] == 1
    assert np.array_equal(h["t"][hit].view(np.uint32), t["t"][hit].view(np.uint32))
    assert np.all(h["b2"][hit] == t["uv"][hit, 1]) and np.all((h["b1"][hit] + h["b2"][hit]).astype(np.float32) == t["uv"][hit, 0])
    assert h["prim"][0] == -1    # Triangle.BadCases known answer


def test_device_bxdfs_match_reference_classes():
    """Rows a15 / a16 at stage level on the DEVICE: f, Pdf and Sample_f of every lobe class (Lambertian R / T, OrenNayar, SpecularReflection /
    Transmission, FresnelSpecular, MicrofacetReflection / Transmission, FresnelBlend; Fresnel NoOp / dielectric / conductor) replayed on the
    2400 records ref_probe dumped from the reference's own classes.  Everything that is +, -, *, /, sqrt is bit-exact (the build has no FMA
    contraction) and the lobes that go through libm (sin / cos / tan / atan of the microfacet sampling, FrConductor) use csrc/pt_libm.h, which returns
    glibc's own bits since round 3: EVERY value must be bit-identical (measured 26 400 of 26 400 since round 3; the gate was rtol 5e-6 / 97 % until
    round 5 -- VERDICT r4 item 6).  The sampled lobe type must always agree."""
    rows = np.load(os.path.join(G, "ref_vectors.npz"))["bxdfs"]
    out = pa.bxdf_eval(rows)
    assert np.array_equal(out["type_s"], rows["type_s"])
    exact, total = 0, 0
    for k in ("f", "pdf", "wi_s", "pdf_s", "f_s"):
        a, b = out[k], rows[k]
        same = (a.view(np.uint32) == b.view(np.uint32)) | (a == b)
        exact += int(same.sum()); total += same.size
        assert same.all(), (k, int((~same).sum()), float(np.abs(a - b).max()))
    _report("device_bxdfs_vs_reference_classes", values=total, bit_identical=exact / total)
    assert exact == total, exact / total


def test_sobol_index_33_bit_regime_c5():
    """configs[4] (3840 x 2160, 512 spp): log2 resolution m = 12, Sobol' indices need 33 bits.  The device's SobolIntervalToIndex against the
    known answers dumped from the reference (ref_vectors.npz: sobol_index rows with m = 12, sample numbers up to ~4000) and the sample values of
    the first dimensions against the oracle."""
    sc = pa.Scene(text='Film "image" "integer xresolution" [3840] "integer yresolution" [2160] "string filename" "x.pfm"\n'
                       'Sampler "sobol" "integer pixelsamples" [512]\nWorldBegin\nWorldEnd\n')
    assert sc.info["sobol_log2_resolution"] == 12 and sc.info["spp"] == 512
    ctx = pa.Context(sc)
    rows = np.load(os.path.join(G, "ref_vectors.npz"))["sobol_index"]
    rows = rows[(rows["m"] == 12) & (rows["px"] < 3840) & (rows["py"] < 2160)]
    assert len(rows) > 100 and (rows["idx"] >= 2 ** 32).any()
    for r in rows[:160]:
        n = int(r["frame"]) + 1
        dev, didx = ctx.sobol(int(r["px"]), int(r["py"]), n, 8)
        assert int(didx[-1]) == int(r["idx"]), (r, int(didx[-1]))
    for (px, py) in [(0, 0), (3839, 2159), (1234, 2001)]:
        dev, didx = ctx.sobol(px, py, 512, 24)
        ref, ridx = ol.sobol(sc, px, py, 512, 24)
        assert np.array_equal(didx, ridx) and np.array_equal(dev.view(np.uint32), ref.view(np.uint32))
    ctx.close()


def test_device_sphere_intersect_matches_reference_vectors():
    """Device Sphere::Intersect on the reference's FullSphere / PartialSphere test constructions (+ transformed spheres): hit decision,
    tHit, p, pError AND the normal bit for bit.  (The normal goes through acos / sin of libm; with round 2's correctly rounded device routines
    16 of 429 records differed by <= 6 ulp and the gate was 1e-5 absolute; csrc/pt_libm.h has returned glibc's bits since round 3 -- VERDICT r4 item 6.)"""
    rows = np.load(os.path.join(G, "ref_vectors.npz"))["spheres"]
    sp, rays = ol.sphere_records(rows)
    h = pa.sphere_intersect(sp, rays)
    assert np.array_equal(h["hit"], rows["hit"])
    hit = rows["hit"] == 1
    for k in ("t", "p", "p_error"):
        assert np.array_equal(h[k][hit].view(np.uint32), rows[k][hit].view(np.uint32)), k
    same = (h["n"][hit].view(np.uint32) == rows["n"][hit].view(np.uint32)) | (h["n"][hit] == rows["n"][hit])
    assert same.all(), (int((~same).sum()), float(np.abs(h["n"][hit] - rows["n"][hit]).max()))


@pytest.mark.parametrize("name,w,h,spp,strategy", [("cornell", 64, 64, 1, None), ("cornell", 64, 64, 8, None), ("materials", 96, 72, 1, None), ("materials", 96, 72, 16, None),
                                                    ("cornell", 64, 64, 4, "spatial"), ("materials", 96, 72, 4, "spatial"),
                                                    ("cornell", 64, 48, 4, "gaussian"), ("cornell", 64, 48, 4, "mitchell"),
                                                    ("cornell", 64, 48, 4, "triangle"), ("cornell", 64, 48, 4, "sinc"),
                                                    ("cornell", 64, 64, 6, "halton"), ("materials", 96, 72, 5, "halton")])
@pytest.mark.parametrize("mode", list(TRACE_MODES))
def test_render_vs_reference_fixture(name, w, h, spp, strategy, mode):
    """GPU image vs the REAL reference's render (tests/golden/*.pfm).  Stated tolerance: per-pixel L2 <= 1e-3 (1 + |ref|)
    for >= 99.5 % of the pixels and relMSE <= 1e-4; at 1 spp each pixel is one camera sample's radiance."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(ROOT, "tools", "gen_golden.py"))
    gg = importlib.util.module_from_spec(spec); spec.loader.exec_module(gg)
    sc = pa.Scene(text=gg.scene_text(name, w, h, spp, strategy))
    ctx = make_ctx(sc, mode)
    ctx.render()
    img = sc.film_image(ctx.film())
    ref = pa.read_pfm(os.path.join(G, "%s_%dx%d_%dspp%s.pfm" % (name, w, h, spp, "_" + strategy if strategy else "")))
    frac, relmse = ol.image_metrics(img, ref)
    _report("render_vs_reference_fixture", scene="%s_%dx%d_%dspp%s" % (name, w, h, spp, "_" + strategy if strategy else ""), mode=mode, within_tol=frac, relMSE=relmse,
            bit_identical_pixels=(img.view(np.uint32) == ref.view(np.uint32)).all(-1).mean())
    assert frac >= 0.995 and relmse <= 1e-4, (frac, relmse)
    ctx.close()


# ---------------------------------------------------------------- the BASELINE.json configs as parity cases (reduced sizes)
def _config_scene(name, tmp):
    import subprocess, sys, re
    gen = os.path.join(ROOT, "tools", "gen_scenes.py")
    if name == "killeroo":       # configs[1]: killeroo-simple, shading-bound small BVH
        text = open(os.path.join(ROOT, "scenes", "killeroo.pbrt")).read()
        text = text.replace('[700] "integer yresolution" [700]', '[192] "integer yresolution" [192]').replace("killeroo_geo/", os.path.join(ROOT, "scenes", "killeroo_geo") + "/")
        return pa.Scene(text=text)
    out = os.path.join(tmp, name + ".pbrt")
    if name == "sanmiguel":      # configs[2]/[4]: San-Miguel-class stand-in, many lights + materials
        subprocess.check_call([sys.executable, gen, "sanmiguel", "--tris", "200000", "--res", "240", "136", "--spp", "8", "--out", out], stdout=subprocess.DEVNULL)
    elif name == "sanmiguel_leafmask":   # the same with its leaf quads as alpha-masked meshes (bench.py --leafmask): the traversal's wave-wide alpha phases (PT_ALPHA_DEFER)
        subprocess.check_call([sys.executable, gen, "sanmiguel", "--tris", "200000", "--res", "240", "136", "--spp", "8", "--leafmask", "--out", out], stdout=subprocess.DEVNULL)
    elif name == "sanmiguel_smokebox":   # the same with a heterogeneous (grid) medium behind a BSDF-less box (bench.py --smokebox): the split form of k_shade_vol -- ratio tracking inside the walked transmittance queries, k_vol_continue
        subprocess.check_call([sys.executable, gen, "sanmiguel", "--tris", "200000", "--res", "240", "136", "--spp", "8", "--smokebox", "--out", out], stdout=subprocess.DEVNULL)
    elif name == "sanmiguel_subsurface":   # the same with three kdsubsurface materials (bench.py --subsurface): BSSRDF probe chains walked through the queues (k_sss_probe_step / k_sss_entry)
        subprocess.check_call([sys.executable, gen, "sanmiguel", "--tris", "200000", "--res", "240", "136", "--spp", "8", "--subsurface", "--out", out], stdout=subprocess.DEVNULL)
    else:                        # configs[3]: bathroom-class, glass + mirror + deep paths (maxdepth 30)
        subprocess.check_call([sys.executable, gen, "bathroom", "--tris", "60000", "--res", "192", "108", "--spp", "16", "--out", out], stdout=subprocess.DEVNULL)
    return pa.Scene(out)


@pytest.mark.parametrize("name,mode", [("killeroo", "general"), ("sanmiguel", "general"), ("bathroom", "general"),
                                       ("killeroo", "bvh4q"), ("sanmiguel", "bvh4q"), ("bathroom", "bvh4q"), ("sanmiguel_leafmask", "bvh4q"), ("sanmiguel_leafmask", "general")])
def test_baseline_configs_reduced(name, mode, tmp_path):
    """GPU vs oracle on reduced-size versions of the BASELINE.json configs + ray accounting + a per-sample criterion
    (killeroo-simple has a Sphere light: it always runs the general kernel instance)."""
    sc = _config_scene(name, str(tmp_path))
    ctx = make_ctx(sc, mode)
    ctx.render(count_work=True)
    img = sc.film_image(ctx.film())
    cnt = ctx.counters()
    ref_rgbw, rcnt, _ = ol.render(sc)
    ref = sc.film_image(ref_rgbw)
    frac, relmse = ol.image_metrics(img, ref)
    # One criterion for every config -- the bathroom's private, wider one went away in round 3 together with its cause: the
    # device's float libm now returns glibc's own bits (csrc/pt_libm.h), so 30 specular bounces have no last-ulp differences left to amplify.
    min_frac, max_relmse = 0.995, 1e-4
    print("config %s/%s: pixels within tolerance %.5f, relMSE %.3g" % (name, mode, frac, relmse))
    _report("baseline_configs_reduced", config=name, mode=mode, within_tol=frac, relMSE=relmse, bit_identical_pixels=(img.view(np.uint32) == ref.view(np.uint32)).all(-1).mean())
    assert frac >= min_frac and relmse <= max_relmse, (name, frac, relmse)
    assert cnt["camera_rays"] == rcnt["camera_rays"] and cnt["trace_guard_trips"] == 0
    if name == "sanmiguel_subsurface":   # probe chains with more than PT_SSS_KEEP counted hits (the quad soups) are walked a second time up to the chosen hit; the oracle keeps a list
        assert -3e-3 * rcnt["closest_rays"] <= cnt["closest_rays"] - rcnt["closest_rays"] <= 2e-2 * rcnt["closest_rays"]
    else:
        assert abs(cnt["closest_rays"] - rcnt["closest_rays"]) <= 3e-3 * rcnt["closest_rays"]
    assert abs(cnt["shadow_rays"] - rcnt["shadow_rays"]) <= 3e-3 * rcnt["shadow_rays"]
    # per camera sample: >= 99.9 % within 1e-4 (1 + |L|)
    ys, xs = np.mgrid[0:sc.height, 0:sc.width]
    xy = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
    sn = np.full(len(xy), 3, dtype=np.int32)
    d, r = ctx.li(xy, sn), ol.li(sc, xy, sn)
    ok = np.linalg.norm(d - r, axis=1) <= 1e-4 * (1 + np.linalg.norm(r, axis=1))
    _report("baseline_configs_reduced_li", config=name, mode=mode, within_tol=ok.mean(), bit_identical=(d.view(np.uint32) == r.view(np.uint32)).all(1).mean())
    assert ok.mean() >= 0.999, (name, ok.mean())
    ctx.close()


# ---------------------------------------------------------------- many lights: light CDF outside the LDS, 3-level CDF search
def _many_lights_scene(strategy, nx=36, nz=35):
    """A ceiling of nx*nz*2 emissive triangles of varying size (2520 lights > the 2047 that fit the LDS copy of the CDF)."""
    rng = np.random.RandomState(7)
    xs = np.sort(rng.uniform(-1, 1, nx + 1)); zs = np.sort(rng.uniform(-1, 1, nz + 1))
    xs[0], xs[-1], zs[0], zs[-1] = -1, 1, -1, 1
    P = [(x, 1.9, z) for z in zs for x in xs]
    idx = []
    for j in range(nz):
        for i in range(nx):
            a = j * (nx + 1) + i; b = a + 1; c = a + nx + 1; d = c + 1
            idx += [a, b, d, a, d, c]     # facing down (-y)
    s = 'LookAt 0 1 -3.4  0 0.9 0  0 1 0\nCamera "perspective" "float fov" [45]\nSampler "sobol" "integer pixelsamples" [8]\nPixelFilter "box"\n'
    s += 'Integrator "path" "integer maxdepth" [3] "string lightsamplestrategy" "%s"\n' % strategy
    s += 'Film "image" "integer xresolution" [64] "integer yresolution" [48] "string filename" "ml.pfm"\nWorldBegin\n'
    s += 'AttributeBegin\n AreaLightSource "diffuse" "rgb L" [6 5 4]\n Material "matte" "rgb Kd" [0 0 0]\n'
    s += ' Shape "trianglemesh" "integer indices" [%s] "point P" [%s]\nAttributeEnd\n' % (" ".join(map(str, idx)), " ".join("%.9g %.9g %.9g" % p for p in P))
    s += 'Material "plastic" "rgb Kd" [.6 .5 .4] "rgb Ks" [.2 .2 .2] "float roughness" [.2]\n'
    s += 'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-2 0 -2  2 0 -2  2 0 2  -2 0 2]\n'
    s += 'Material "matte" "rgb Kd" [.3 .5 .7]\nShape "trianglemesh" "integer indices" [0 1 2 0 2 3 4 6 5 4 7 6 0 4 5 0 5 1 1 5 6 1 6 2 2 6 7 2 7 3 3 7 4 3 4 0] '
    s += '"point P" [-.5 0 -.5  .5 0 -.5  .5 0 .5  -.5 0 .5  -.5 .8 -.5  .5 .8 -.5  .5 .8 .5  -.5 .8 .5]\nWorldEnd\n'
    return s


@pytest.mark.parametrize("strategy", ["power", "spatial"])
def test_many_lights(strategy):
    sc = pa.Scene(text=_many_lights_scene(strategy))
    assert sc.info["n_lights"] == 2520
    ctx = pa.Context(sc)
    ctx.render()
    img = sc.film_image(ctx.film())
    ref_rgbw, _, _ = ol.render(sc)
    frac, relmse = ol.image_metrics(img, sc.film_image(ref_rgbw))
    assert frac >= 0.995 and relmse <= 1e-4, (strategy, frac, relmse)
    ys, xs = np.mgrid[0:sc.height, 0:sc.width]
    xy = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
    sn = np.full(len(xy), 5, dtype=np.int32)
    d, r = ctx.li(xy, sn), ol.li(sc, xy, sn)
    ok = np.linalg.norm(d - r, axis=1) <= 1e-4 * (1 + np.linalg.norm(r, axis=1))
    assert ok.mean() >= 0.999, (strategy, ok.mean())
    ctx.close()


# ---------------------------------------------------------------- edge cases (tests/edge_scenes.py), against the reference's renders
import edge_scenes


@pytest.mark.parametrize("name", edge_scenes.NAMES + edge_scenes.TEX_ORACLE_ONLY + ["instances2"])
def test_edge_cases_vs_reference_fixture(name):
    sc = pa.Scene(text=edge_scenes.scene(name))
    ctx = pa.Context(sc)
    ctx.render()
    img = sc.film_image(ctx.film())
    ref = pa.read_pfm(os.path.join(G, "edge_%s.pfm" % name))
    assert img.shape == ref.shape
    frac, relmse = ol.image_metrics(img, ref)
    assert frac >= 0.995 and relmse <= 1e-4, (name, frac, relmse)
    # split renders accumulate to the same film: samples [0,1) then [1,spp)
    if sc.info["spp"] > 1:
        ctx.film_clear()
        ctx.render(spp_begin=0, spp_end=1)
        ctx.render(spp_begin=1, spp_end=sc.info["spp"])
        img2 = sc.film_image(ctx.film())
        assert np.allclose(img2, img, rtol=1e-5, atol=1e-6)
    ctx.close()


@pytest.mark.parametrize("name", edge_scenes.INSTANCE_NAMES)
def test_two_level_instancing_vs_reference_fixture(name, monkeypatch):
    """Row f3 on the device: with PBRT_AMD_INSTANCING=1 the host hands over the reference's own structure (one BVH per object,
    TransformedPrimitive leaves) and k_trace / k_shade <..., INST> traverse it -- ray into the object's space with the reference's
    error-bounded origin (core/primitive.cpp:76-111, transform.h:252-264), interaction transformed back.  The oracle reproduces the
    reference's render bit for bit in this mode; first GPU run (round 2): every pixel within tolerance, relMSE 1e-16...1e-17,
    99.6-99.9 % of the pixels bit-identical."""
    monkeypatch.delenv("PBRT_AMD_INSTANCING", raising=False)   # two-level is the host's default (round 2); =0 flattens
    sc = pa.Scene(text=edge_scenes.scene(name))
    ctx = pa.Context(sc)
    ctx.render()
    img = sc.film_image(ctx.film())
    ref = pa.read_pfm(os.path.join(G, "edge_%s.pfm" % name))
    frac, relmse = ol.image_metrics(img, ref)
    assert frac >= 0.999 and relmse <= 1e-8, (name, frac, relmse)
    assert float(np.mean(np.abs(img - ref).max(-1) == 0)) >= 0.98
    ctx.close()
    # the flattening option still renders the same surfaces (different roundings: the image criterion)
    monkeypatch.setenv("PBRT_AMD_INSTANCING", "0")
    flat = pa.Scene(text=edge_scenes.scene(name))
    assert flat.info["n_tris"] > sc.info["n_tris"]
    ctx = pa.Context(flat)
    ctx.render()
    frac, relmse = ol.image_metrics(flat.film_image(ctx.film()), ref)
    assert frac >= 0.995 and relmse <= 1e-4, (name, frac, relmse)
    ctx.close()


# ---------------------------------------------------------------- textures
def _tex_queries(n, seed):
    rng = np.random.default_rng(seed)
    q = np.zeros(n, dtype=pa.TEX_QUERY_DTYPE)
    q["p"] = rng.uniform(-4, 4, (n, 3)); q["uv"] = rng.uniform(-1, 3, (n, 2))
    s = 10.0 ** rng.uniform(-4, -0.5, (n, 1))
    q["dpdx"] = rng.normal(size=(n, 3)) * s; q["dpdy"] = rng.normal(size=(n, 3)) * s
    for k in ("dudx", "dvdx", "dudy", "dvdy"):
        q[k] = rng.normal(size=n) * s[:, 0]
    z = rng.random(n) < 0.15   # interactions without differentials: every bounce after the first, alpha tests
    for k in ("dpdx", "dpdy", "dudx", "dvdx", "dudy", "dvdy"):
        q[k][z] = 0
    return q


@pytest.mark.parametrize("name", edge_scenes.TEX_NAMES)
def test_texture_evaluation_matches_oracle(name):
    """Texture<T>::Evaluate on the device (mi_texture_eval) against the oracle's restatement for EVERY node of the scene's texture
    table at random interactions: image maps (EWA / trilinear, three wrap modes, non-power-of-two sources), procedural classes,
    the four 2D mappings and the 3D one.  libm differences move values by ulps and flip a checker / dot / level decision on a
    handful of inputs, hence: within 2e-4 relative on >= 99 % of the evaluations of every node."""
    sc = pa.Scene(text=edge_scenes.scene(name))
    ctx = pa.Context(sc)
    q = _tex_queries(4096, 11)
    for node in range(sc.info["n_textures"]):
        dev = ctx.texture_eval(node, q)
        ref = ol.texture_eval(sc, node, q)
        close = np.isclose(dev, ref, rtol=2e-4, atol=2e-6).all(axis=1)
        assert close.mean() >= 0.99, (name, node, float(close.mean()))
    ctx.close()


@pytest.mark.parametrize("name", edge_scenes.TEX_NAMES)
def test_textured_scenes_vs_reference_fixture(name):
    """Rendered images of the textured scenes (per-hit material evaluation, ray differentials of camera rays, bump mapping,
    alpha / shadow-alpha masks in both traversal kernels) against the reference's own renders."""
    sc = pa.Scene(text=edge_scenes.scene(name))
    assert sc.info["n_textured_materials"] > 0 or sc.info["n_masked_meshes"] > 0
    ctx = pa.Context(sc)
    ctx.render()
    img = sc.film_image(ctx.film())
    ref = pa.read_pfm(os.path.join(G, "edge_%s.pfm" % name))
    frac, relmse = ol.image_metrics(img, ref)
    assert frac >= 0.99 and relmse <= 5e-4, (name, frac, relmse)
    ctx.close()


# ---------------------------------------------------------------- the command-line renderer end to end (parser -> BVH -> device -> Film -> file)
@pytest.mark.launches_processes
def test_cli_render_matches_reference_fixture(tmp_path):
    import subprocess, importlib.util
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(ROOT, "tools", "gen_golden.py"))
    gg = importlib.util.module_from_spec(spec); spec.loader.exec_module(gg)
    exe = os.path.join(ROOT, "pbrt-v3-distributed_amd", "bin", "pbrt_amd")
    f = tmp_path / "c.pbrt"
    f.write_text(gg.scene_text("cornell", 64, 64, 8))
    out = tmp_path / "c.pfm"
    r = subprocess.run([exe, "--quiet", "--outfile", str(out), str(f)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and out.exists(), r.stderr
    img = pa.read_pfm(str(out))
    ref = pa.read_pfm(os.path.join(G, "cornell_64x64_8spp.pfm"))
    frac, relmse = ol.image_metrics(img, ref)
    assert frac >= 0.995 and relmse <= 1e-4, (frac, relmse)
    # --cropwindow and EXR output (pbrt's default output format)
    out2 = tmp_path / "crop.exr"
    r = subprocess.run([exe, "--quiet", "--cropwindow", "0.25", "0.75", "0.5", "1", "--outfile", str(out2), str(f)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and out2.exists() and open(out2, "rb").read(4) == bytes([0x76, 0x2F, 0x31, 0x01]), r.stderr


def test_killeroo_simple_vs_the_reference_scene():
    """The reference's example scene as shipped (Sphere area light, Halton, Loop subdivision) -- GPU render of scenes/killeroo.pbrt vs the
    reference's render of its own scenes/killeroo-simple.pbrt."""
    text = open(os.path.join(ROOT, "scenes", "killeroo.pbrt")).read()
    text = text.replace('[700] "integer yresolution" [700]', '[96] "integer yresolution" [96]').replace("killeroo_geo/", os.path.join(ROOT, "scenes", "killeroo_geo") + "/")
    sc = pa.Scene(text=text)
    ctx = pa.Context(sc)
    ctx.render()
    img = sc.film_image(ctx.film())
    ref = pa.read_pfm(os.path.join(G, "killeroo_simple_96x96_reference.pfm"))
    frac, relmse = ol.image_metrics(img, ref)
    assert frac >= 0.995 and relmse <= 1e-4, (frac, relmse)
    ctx.close()


@pytest.mark.parametrize("name", edge_scenes.VOL_NAMES + edge_scenes.SSS_NAMES)
def test_volpath_and_subsurface_scenes_vs_reference_fixture(name):
    """Row f4 on the device (k_shade_vol, csrc/pt_volpath.h): VolPathIntegrator::Li (integrators/volpath.cpp:55-190) with homogeneous
    and grid media, medium interfaces with and without a BSDF, the phase-function branch of EstimateDirect, VisibilityTester::Tr /
    Scene::IntersectTr, and the BSSRDF branch (path.cpp:153-174) with TabulatedBSSRDF probe chains.  Image against the REFERENCE's render
    of the same file (fixtures edge_vol_*.pfm / edge_sss_*.pfm), per-sample radiance against the oracle (which reproduces those fixtures
    bit for bit): |dL| <= 1e-4 (1 + |L|) for >= 99.5 % of the samples."""
    sc = pa.Scene(text=edge_scenes.scene(name))
    ctx = pa.Context(sc)
    rng = np.random.default_rng(11)
    n = 3000
    xy = np.stack([rng.integers(0, sc.width, n), rng.integers(0, sc.height, n)], axis=1).astype(np.int32)
    s = rng.integers(0, sc.info["spp"], n).astype(np.int32)
    dev, ref = ctx.li(xy, s), ol.li(sc, xy, s)
    ok = np.linalg.norm(dev - ref, axis=1) <= 1e-4 * (1 + np.linalg.norm(ref, axis=1))
    assert ok.mean() >= 0.995, (name, ok.mean())
    ctx.render()
    img = sc.film_image(ctx.film())
    fx = pa.read_pfm(os.path.join(G, "edge_%s.pfm" % name))
    frac, relmse = ol.image_metrics(img, fx)
    assert frac >= 0.995 and relmse <= 1e-4, (name, frac, relmse)
    assert ctx.counters()["trace_guard_trips"] == 0
    ctx.close()


@pytest.mark.parametrize("name", ["vol_fog", "vol_glass", "vol_none", "vol_inst", "vol_alpha_fog"])
def test_volpath_general_form_on_homogeneous_scenes(name, monkeypatch):
    """Scenes whose media are all homogeneous (and that have no masks or BSSRDFs) send their shadow / MIS rays through the wavefront queues
    (k_shade_vol<WAVE = true>; closed-form transmittance, interfaces walked: test_walked_interfaces_...); PBRT_AMD_VOL_INLINE=1 runs them through the general form
    (every lane traces its own transmittance rays) -- both must reproduce the reference's render."""
    monkeypatch.setenv("PBRT_AMD_VOL_INLINE", "1")
    sc = pa.Scene(text=edge_scenes.scene(name))
    ctx = pa.Context(sc)
    ctx.render()
    frac, relmse = ol.image_metrics(sc.film_image(ctx.film()), pa.read_pfm(os.path.join(G, "edge_%s.pfm" % name)))
    assert frac >= 0.995 and relmse <= 1e-4, (name, frac, relmse)
    ctx.close()


@pytest.mark.parametrize("name,flatten", [("vol_inst", False), ("vol_inst", True), ("vol_alpha_fog", False)])
def test_walked_interfaces_match_the_general_form(name, flatten, monkeypatch):
    """BSDF-less interfaces between homogeneous media in wavefront form (round 3): the shadow and MIS rays are walked through the interfaces segment by
    segment through the queues (k_trace<..., TR> + k_vol_tr_step) instead of being traced by the shading lanes (PBRT_AMD_VOL_TR_QUEUES=0: the general
    form).  vol_inst has instanced and top-level volumes behind BSDF-less boundaries, vol_alpha_fog alpha-masked quads in fog in front of one (the walk's
    segments evaluate alphaMask like Scene::Intersect, shapes/triangle.cpp:333-338): both forms reproduce the reference's render, with the same rays."""
    monkeypatch.setenv("PBRT_AMD_INSTANCING", "0" if flatten else "1")
    fx = pa.read_pfm(os.path.join(G, "edge_%s.pfm" % name))
    out = {}
    for form in ("walked", "general"):
        if form == "general":
            monkeypatch.setenv("PBRT_AMD_VOL_TR_QUEUES", "0")
        else:
            monkeypatch.delenv("PBRT_AMD_VOL_TR_QUEUES", raising=False)
        sc = pa.Scene(text=edge_scenes.scene(name))
        ctx = pa.Context(sc)
        ctx.timing_enable(True); ctx.counters_reset()
        ctx.render()
        t, cnt = ctx.timing(), ctx.counters()
        img = sc.film_image(ctx.film())
        frac, relmse = ol.image_metrics(img, fx)
        assert (frac >= 0.995 and relmse <= 1e-4) if not flatten else (frac >= 0.99 and relmse <= 5e-4), (form, frac, relmse)
        out[form] = (img, {k: v[1] for k, v in t.items() if v[1]}, cnt)
        ctx.close()
    assert "anyhit" in out["walked"][1] and "mis_closest" in out["walked"][1] and "anyhit" not in out["general"][1]   # the walk ran / the lanes traced their own rays
    assert out["walked"][2]["closest_rays"] == out["general"][2]["closest_rays"] and out["walked"][2]["trace_guard_trips"] == 0   # segment for segment the same queries
    assert np.allclose(out["walked"][0], out["general"][0], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", ["vol_inst", "sss_inst", "vol_glass"])
def test_volpath_and_subsurface_flattened_instances(name, monkeypatch):
    """the same scenes with PBRT_AMD_INSTANCING=0 (instances flattened on the host): single-level k_shade_vol instances"""
    monkeypatch.setenv("PBRT_AMD_INSTANCING", "0")
    sc = pa.Scene(text=edge_scenes.scene(name))
    ctx = pa.Context(sc)
    ctx.render()
    frac, relmse = ol.image_metrics(sc.film_image(ctx.film()), pa.read_pfm(os.path.join(G, "edge_%s.pfm" % name)))
    assert frac >= 0.99 and relmse <= 5e-4, (name, frac, relmse)   # flattening: the image criterion of the device fuzzer
    ctx.close()


def test_media_under_path_are_ignored():
    """PathIntegrator passes handleMedia = false: the same media under Integrator "path" do nothing, in the reference and here."""
    sc2 = pa.Scene(text=edge_scenes.scene("vol_fog").replace('Integrator "volpath" "integer maxdepth" [6]', 'Integrator "path" "integer maxdepth" [5]'))
    ctx = pa.Context(sc2)
    ctx.render()
    img = sc2.film_image(ctx.film())
    ref = sc2.film_image(ol.render(sc2, nthreads=4)[0])
    frac, relmse = ol.image_metrics(img, ref)
    assert frac >= 0.995 and relmse <= 1e-4, (frac, relmse)
    ctx.close()


# ---------------------------------------------------------------- deep traversal stacks (round 4: TravStackB's generic tail + the spill slices)
@pytest.mark.gpu
def test_deep_stacks_take_the_generic_tail_and_spill():
    """6 000 random triangles that all span the same cube: every child box of every node is entered, three pushes per level, and the host emulation of the
    kernel's state machine (mi_bvh4q_validate) needs 24 stack entries -- more than the 15 a lane of the 768-thread traversal shape holds in LDS.  The
    interior step's fast tail is only valid while no lane of the wave is within three entries of its LDS part; here waves go through the per-wave branch into
    the generic tail and through the HBM spill slices all the time.  Hits and occlusion flags must equal the oracle's BVH2 walk bit for bit."""
    rng = np.random.default_rng(5)
    n = 6000
    P = rng.uniform(-1, 1, (3 * n, 3)).astype(np.float32)
    text = ('LookAt 0 0 5  0 0 0  0 1 0\nCamera "perspective" "float fov" [40]\nSampler "sobol" "integer pixelsamples" [1]\n'
            'Film "image" "integer xresolution" [32] "integer yresolution" [32] "string filename" ["x.pfm"]\nWorldBegin\nMaterial "matte"\n'
            'Shape "trianglemesh" "integer indices" [%s] "point P" [%s]\nWorldEnd\n' % (" ".join(str(i) for i in range(3 * n)), " ".join("%.9g" % v for v in P.reshape(-1))))
    sc = pa.Scene(text=text)
    m = 4000
    o = rng.uniform(-1, 1, (m, 3)).astype(np.float32)
    d = rng.standard_normal((m, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1)[:, None]
    rays = np.zeros(m, dtype=pa.RAY_DTYPE)
    rays["o"] = 3 * o
    rays["d"] = -o / np.linalg.norm(o, axis=1)[:, None] + 0.3 * d
    rays["tmax"] = np.inf
    _, st = pa.bvh4q_validate(sc, rays, want_hits=False)
    ctx = pa.Context(sc)
    ti = ctx.trace_info()
    assert st["max_stack"] > ti["lds_stack_entries"], (st, ti)   # the workload does leave the LDS part
    dh = ctx.intersect(rays)
    rh, _ = ol.intersect(sc, rays)
    assert np.array_equal(dh["prim"], rh["prim"])
    hit = rh["prim"] >= 0
    assert hit.sum() > 3000
    assert np.array_equal(dh["t"][hit].view(np.uint32), rh["t"][hit].view(np.uint32))
    rays["tmax"] = rng.uniform(0.5, 4.0, m).astype(np.float32)
    assert np.array_equal(ctx.intersect_p(rays), ol.intersect_p(sc, rays)[0])
    assert ctx.counters()["trace_guard_trips"] == 0
    ctx.close()


# ---------------------------------------------------------------- the reference's own host driving the device
REF_STUB = os.path.join(ROOT, "oracle", "_ref", "pbrt_ref_wavefront")


@pytest.mark.launches_processes
@pytest.mark.parametrize("name", ["infinite", "spheres", "instances2", "tex_materials", "tex_bump", "tex_alpha", "vol_smoke", "vol_inst", "sss_coeff", "sss_inst"])
def test_reference_host_drives_the_device(name, tmp_path):
    """The drop-in itself: pbrt-v3's unmodified main / parser / API state machine / shape and material factories / BVH build (libpbrt_ref.a) with
    `Integrator "path"` / `"volpath"` bound to the reference-side stub of INTEGRATION.md s.2 (integration/wavefrontpath.cpp), which flattens
    the reference's own Scene to a mi_scene_desc and calls mi_ctx_create / mi_scene_upload / mi_render / mi_film_download of libpbrt_amd.so; the
    film goes back through the reference's Film::MergeFilmTile / WriteImage.  Compared with the reference's own render of the same file
    (committed fixture): the image criterion of this suite."""
    import subprocess
    if not os.access(REF_STUB, os.X_OK):
        pytest.skip("oracle/_ref/pbrt_ref_wavefront not built (it is built where /root/reference is present and travels with the snapshot)")
    scene = str(tmp_path / "s.pbrt")
    open(scene, "w").write(edge_scenes.scene(name))
    out = str(tmp_path / "o.pfm")
    env = {k: v for k, v in os.environ.items() if not k.startswith("PBRT_AMD_BACKEND")}   # the binding knows no backend switch: it binds mi_* or fails
    env["PBRT_AMD_DEVICE_LIB"] = pa.DEVICE_LIB
    r = subprocess.run([REF_STUB, "--quiet", "--nthreads", "4", "--outfile", out, scene], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and os.path.exists(out), (r.stdout[-400:], r.stderr[-800:])
    img, ref = pa.read_pfm(out), pa.read_pfm(os.path.join(G, "edge_%s.pfm" % name))
    assert img.shape == ref.shape
    frac, relmse = ol.image_metrics(img, ref)
    assert frac >= 0.995 and relmse <= 1e-4, (name, frac, relmse)


@pytest.mark.launches_processes
def test_reference_host_shards_tiles_over_gpu_contexts(tmp_path):
    """`PBRT_AMD_GPUS=2` in the reference-side binding: one mi_ctx per rank, mi_render(rank r, world 2) each, one mi_film_gather onto rank 0 (the
    reference's tile loop core/integrator.cpp:228-339 + the merge of film.cpp:117-130 across devices).  On a one-GPU box both contexts sit on
    device 0 (PBRT_AMD_GPU_MAP=0,0) and the gather is the library's same-device sum; the image must equal the one-context render bit for bit
    (box filter: disjoint tiles) and the reference's fixture by the suite's criterion."""
    import subprocess
    if not os.access(REF_STUB, os.X_OK):
        pytest.skip("oracle/_ref/pbrt_ref_wavefront not built (it is built where /root/reference is present and travels with the snapshot)")
    scene = str(tmp_path / "s.pbrt")
    open(scene, "w").write(edge_scenes.scene("infinite"))
    imgs = []
    for gpus in ("1", "2"):
        out = str(tmp_path / ("o%s.pfm" % gpus))
        env = {k: v for k, v in os.environ.items() if not k.startswith("PBRT_AMD_BACKEND")}
        env.update(PBRT_AMD_DEVICE_LIB=pa.DEVICE_LIB, PBRT_AMD_GPUS=gpus, PBRT_AMD_GPU_MAP="0,0")
        r = subprocess.run([REF_STUB, "--quiet", "--nthreads", "4", "--outfile", out, scene], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and os.path.exists(out), (r.stdout[-400:], r.stderr[-800:])
        imgs.append(pa.read_pfm(out))
    assert np.array_equal(imgs[0], imgs[1])
    frac, relmse = ol.image_metrics(imgs[1], pa.read_pfm(os.path.join(G, "edge_infinite.pfm")))
    assert frac >= 0.995 and relmse <= 1e-4, (frac, relmse)


# ---------------------------------------------------------------- stage-level and scene tests added after the last GPU call of round 2
# (kernels byte-identical to the measured build; these were exercised on tools/hostemu only and are confirmed on the MI355X by the round-end run;
#  they sit at the end of the file so that `pytest -x` reaches every hardware-validated test first)
def _ulp_distance(a, b):
    ia = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    ib = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia); ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def test_device_light_sampling_matches_reference_classes():
    """mi_light_sample (the SampleLi / PdfLi routines k_shade and k_shade_vol call) against the reference's own Light classes: the 3 840 records of
    tests/golden/light_vectors.npz (DiffuseAreaLight over triangles -- tests/shapes.cpp Triangle.Sampling geometry -- and spheres, point, spot;
    surface and medium reference points).  Stated tolerance: <= 16 ulp (or 2e-6 (1 + |ref|) for components next to zero) on wi, pdf, Li and the shadow ray's origin (the sphere's cone sampling goes
    through the device's own sin / cos), 1e-5 (1 + |d|) on the shadow ray's direction (a difference of two offset points), Pdf_Li <= 16 ulp where
    both hit and identical hit / miss decisions otherwise."""
    recs = np.load(os.path.join(G, "light_vectors.npz"))["light_samples"]
    sc = pa.Scene(text=edge_scenes.light_kat_scene(recs))
    q = np.zeros(len(recs), dtype=pa.LIGHT_QUERY_DTYPE)
    q["light"] = np.arange(len(recs)) // 24
    q["p"] = recs["p"]; q["n"] = recs["n"]; q["u"] = recs["u"]; q["wi"] = recs["wi2"]
    ctx = pa.Context(sc)
    o = ctx.light_sample(q)
    ctx.close()
    def close(a, b):   # <= 16 ulp, or (components next to zero) <= 2e-6 (1 + |ref|) absolute
        return (_ulp_distance(a, b) <= 16) | (np.abs(a - b) <= 2e-6 * (1 + np.abs(b)))
    for k in ("wi", "pdf", "Li", "ray_o"):
        assert close(o[k], recs[k]).all(), (k, int(_ulp_distance(o[k], recs[k]).max()))
    assert (np.abs(o["ray_d"] - recs["ray_d"]) <= 1e-5 * (1 + np.abs(recs["ray_d"]))).all()
    assert o["ray_tmax"].tobytes() == recs["ray_tmax"].tobytes()
    assert ((o["pdf_wi"] > 0) == (recs["pdf_b"] > 0)).mean() >= 0.999     # a grazing direction may flip between hit and miss
    both = (o["pdf_wi"] > 0) & (recs["pdf_b"] > 0)
    assert close(o["pdf_wi"][both], recs["pdf_b"][both]).all()
    assert (o["delta"] == (recs["kind"] >= 2)).all()
    # the triangle, point and spot records involve nothing beyond +, -, *, /, sqrt: expected bit-exact (>= 99 % required, as for the BxDF vectors)
    ex = recs["kind"] != 1
    same = sum(int((o[k][ex] == recs[k][ex]).sum()) for k in ("wi", "pdf", "Li", "ray_o", "ray_d"))
    total = sum(o[k][ex].size for k in ("wi", "pdf", "Li", "ray_o", "ray_d"))
    _report("device_light_sampling_vs_reference_classes", values=total, bit_identical=same / total,
            bit_identical_all_kinds=sum(int((o[k] == recs[k]).sum()) for k in ("wi", "pdf", "Li", "ray_o", "ray_d")) / sum(o[k].size for k in ("wi", "pdf", "Li", "ray_o", "ray_d")))
    assert same / total >= 0.99, same / total


def test_device_scene_dependent_lights_match_reference_classes():
    """mi_light_sample on distant and infinite lights (constant and with a radiance map under a rotation) against the reference's own classes
    (light_vectors.npz 'scene_lights', 768 records): Sample_Li, Pdf_Li, and Le of an escaped ray.  Stated tolerance: relative 2e-5 on pdf, Li, Pdf_Li
    and Le (angles go through the device's own acos / atan2 / sin; the constant light is the documented <= 1 ulp deviation of DESIGN.md s.0),
    2e-6 absolute on wi, 1e-5 (1 + |d|) on the shadow ray's direction, the distant lights bit-exact."""
    recs = np.load(os.path.join(G, "light_vectors.npz"))["scene_lights"]
    sc = pa.Scene(text=edge_scenes.scene_light_kat_scene(recs))
    q = np.zeros(len(recs), dtype=pa.LIGHT_QUERY_DTYPE)
    q["light"] = np.arange(len(recs)) // 64
    q["p"] = recs["p"]; q["n"] = recs["n"]; q["u"] = recs["u"]; q["wi"] = recs["wi2"]
    ctx = pa.Context(sc)
    o = ctx.light_sample(q)
    ctx.close()
    for a, b in ((o["pdf"], recs["pdf"]), (o["Li"], recs["Li"]), (o["pdf_wi"], recs["pdf_b"]), (o["le_wi"], recs["le"])):
        assert np.allclose(a, b, rtol=2e-5, atol=1e-7), float(np.abs(a - b).max())
    assert np.abs(o["wi"] - recs["wi"]).max() <= 2e-6
    assert (np.abs(o["ray_o"] - recs["ray_o"]) <= 2e-6 * (1 + np.abs(recs["ray_o"]))).all()
    assert (np.abs(o["ray_d"] - recs["ray_d"]) <= 1e-5 * (1 + np.abs(recs["ray_d"]))).all()
    d = recs["kind"] == 4
    same = sum(int((o[k][d] == recs[k][d]).sum()) for k in ("wi", "pdf", "Li", "ray_o", "ray_d"))
    assert same / sum(o[k][d].size for k in ("wi", "pdf", "Li", "ray_o", "ray_d")) >= 0.99


def test_device_bssrdf_profile_and_phase_function_match_reference_classes():
    """Row f4 at stage level on the device against the reference's own classes (tests/golden/bssrdf_tables.npz): mi_bssrdf_eval -- TabulatedBSSRDF::Sr,
    Sample_Sr, Pdf_Sr and SubsurfaceFromDiffuse, the out-of-line routines k_shade_vol calls -- on 2 000 coefficient triples, mi_phase_hg --
    HenyeyGreenstein::p / Sample_p -- on 4 000 records.  Everything but the sampled direction is +, -, *, /, sqrt: bit-exact expected (>= 99 % required,
    1e-5 relative stated); the sampled direction goes through the device's sin / cos: 2e-6 absolute."""
    d = np.load(os.path.join(G, "bssrdf_tables.npz"))
    r = d["radial"]
    q = np.zeros(len(r), pa.BSSRDF_QUERY_DTYPE)
    for k in ("sigma_a", "sigma_s", "ch", "r", "u", "kd", "mfp"):
        q[k] = r[k]
    o = pa.bssrdf_eval(d["tables"][0], float(d["tables"][0]["eta"]), q)
    same = total = 0
    for a, b in ((o["sr"], r["sr"]), (o["sample_sr"], r["sample_sr"]), (o["pdf_sr"], r["pdf_sr"]), (o["sigma_a"], r["out_sigma_a"]), (o["sigma_s"], r["out_sigma_s"])):
        assert np.allclose(a, b, rtol=1e-5, atol=1e-9), float(np.abs(a - b).max())
        same += int((a == b).sum()); total += a.size
    h = d["hg"]
    qh = np.zeros(len(h), pa.HG_QUERY_DTYPE)
    for k in ("g", "wo", "wi", "u"):
        qh[k] = h[k]
    oh = pa.phase_hg(qh)
    for a, b in ((oh["p"], h["p"]), (oh["p_s"], h["p_s"])):
        assert np.allclose(a, b, rtol=1e-5, atol=1e-9)
        same += int((a == b).sum()); total += a.size
    assert np.abs(oh["wi_s"] - h["wi_s"]).max() <= 2e-6
    assert same / total >= 0.99, same / total


@pytest.mark.parametrize("name", edge_scenes.FURNACE_NAMES)
def test_furnace_scenes(name):
    """The reference's analytic scenes (src/tests/analytic_scenes.cpp:71-203, CheckSceneAverage :55-68) on the device: mean radiance inside the closed unit
    sphere 1.0 +- 0.02 for Sobol' / Halton x path / volpath at 256 spp, depth 8; the Sobol' / path image also against the reference's own render (fixture)."""
    for sampler in ("sobol", "halton"):
        for integrator in ("path", "volpath"):
            sc = pa.Scene(text=edge_scenes.furnace_scene(name, sampler, integrator))
            ctx = pa.Context(sc)
            ctx.render()
            img = sc.film_image(ctx.film())
            ctx.close()
            assert abs(float(img.mean()) - 1.0) <= 0.02, (name, sampler, integrator, float(img.mean()))
            if (sampler, integrator) == ("sobol", "path"):
                frac, relmse = ol.image_metrics(img, pa.read_pfm(os.path.join(G, "%s.pfm" % name)))
                assert frac >= 0.995 and relmse <= 1e-4, (name, frac, relmse)


def test_contexts_sharing_a_device_render_concurrently():
    """the texture / alpha / instance tables live in per-DEVICE __constant__ symbols that every pass rewrites.  Two contexts on
    one device, rendering different textured / instanced scenes from two host threads at once, take turns on them (TableTurn in run_pass):
    each image must equal the one the same context renders alone (bit for bit wherever the film sum has a fixed order)."""
    import threading
    names = ["tex_materials", "tex_alpha", "instances2"]
    scenes = [pa.Scene(text=edge_scenes.scene(n)) for n in names]
    ctxs = [pa.Context(s) for s in scenes]
    alone = []
    for c in ctxs:
        c.film_clear(); c.render(); alone.append(c.film().copy())
    rounds, got, errs = 4, [[] for _ in ctxs], []

    def work(i):
        try:
            for _ in range(rounds):
                ctxs[i].film_clear(); ctxs[i].render(); got[i].append(ctxs[i].film().copy())
        except Exception as e:   # noqa: BLE001
            errs.append((i, e))
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(ctxs))]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    for i, n in enumerate(names):
        assert alone[i].any()
        # pixels that only received their own samples agree bit for bit between two renders; one that also got a neighbour's sample landing
        # exactly on its edge is summed through float atomics in no fixed order (see test_tile_sharding_is_exact)
        own_only = alone[i][..., 3] == scenes[i].info["spp"]
        assert own_only.mean() > 0.9
        for g in got[i]:
            assert np.array_equal(g[own_only].view(np.uint32), alone[i][own_only].view(np.uint32)), n
            assert np.allclose(g, alone[i], rtol=1e-6, atol=1e-7), n
    for c in ctxs: c.close()


def test_camera_rays_match_reference_classes():
    """mi_camera_rays against the reference's own SobolSampler + PerspectiveCamera objects (tests/golden/camera_vectors.npz: pinhole / thin lens,
    non-square film, crop window, frame aspect ratio): pFilm and the pinhole rays bit for bit, thin-lens rays within 1e-6 (bit-exact expected)."""
    recs = np.load(os.path.join(ROOT, "tests", "golden", "camera_vectors.npz"))["camera_rays"]
    import edge_scenes as es
    for c in range(4):
        r = recs[recs["cfg"] == c]
        sc = pa.Scene(text=es.camera_kat_scene(r[0]))
        ctx = pa.Context(sc)
        rays, pf = ctx.camera_rays(np.stack([r["px"], r["py"]], 1).astype(np.int32), r["s"])
        ctx.close()
        assert np.array_equal(pf.view(np.uint32), r["p_film"].view(np.uint32)), c
        for k in ("o", "d"):
            a = np.ascontiguousarray(rays[k])
            if r["lensr"][0] == 0:
                assert np.array_equal(a.view(np.uint32), r[k].view(np.uint32)), (c, k)
            else:   # the lens sample goes through the device's own sin / cos (ConcentricSampleDisk): bit-exact expected, 1e-6 stated
                assert (a == r[k]).mean() >= 0.9 and np.allclose(a, r[k], rtol=1e-6, atol=1e-6), (c, k)



def test_camera_differentials_match_reference_classes():
    """The offset rays rx / ry of PerspectiveCamera::GenerateRayDifferential (perspective.cpp:118-141) as the shading kernels rebuild them at the first
    hit of a textured scene (CameraDifferentials), against the same reference-class records: the fixture holds the camera's own rx / ry, the device
    hands out what Render makes of them, ScaleDifferentials(1 / sqrt(spp)) (integrator.cpp:262-263) -- applied here to the records in float32.
    Pinhole: bit for bit; thin lens (the lens point goes through the device's sin / cos): >= 90 % bit for bit, 1e-6 stated."""
    recs = np.load(os.path.join(ROOT, "tests", "golden", "camera_vectors.npz"))["camera_rays"]
    import edge_scenes as es
    for c in range(4):
        r = recs[recs["cfg"] == c]
        sc = pa.Scene(text=es.camera_kat_scene(r[0]))
        ctx = pa.Context(sc)
        got = ctx.camera_differentials(np.stack([r["px"], r["py"]], 1).astype(np.int32), r["s"])
        ctx.close()
        s = np.float32(1) / np.sqrt(np.float32(sc.info["spp"]))
        for j, (k, base) in enumerate((("rx_o", "o"), ("rx_d", "d"), ("ry_o", "o"), ("ry_d", "d"))):
            want = (r[base] + (r[k] - r[base]) * s).astype(np.float32)
            a = np.ascontiguousarray(got[:, j])
            if r["lensr"][0] == 0:
                assert np.array_equal(a.view(np.uint32), want.view(np.uint32)), (c, k)
            else:
                assert (a == want).mean() >= 0.9 and np.allclose(a, want, rtol=1e-6, atol=1e-6), (c, k)


def test_c5_regime_crop_tile_sharded():
    """configs[4]'s regime through the whole pipeline: a 24 x 21 pixel crop of a 3840 x 2160 film at 512 spp (log2 resolution 12: the Sobol' index of a
    sample needs 33 bits; test_sobol_index_33_bit_regime_c5 covers the index alone) against the reference's own render of the same crop, and the
    8-rank tile-sharded form of it (rank r of 8 rendered separately, films summed) against the single render."""
    sc = pa.Scene(text=edge_scenes.scene("c5_crop"))
    assert sc.info["spp"] == 512
    ctx = pa.Context(sc)
    ctx.render()
    whole = ctx.film()
    frac, relmse = ol.image_metrics(sc.film_image(whole), pa.read_pfm(os.path.join(G, "edge_c5_crop.pfm")))
    assert frac >= 0.995 and relmse <= 1e-4, (frac, relmse)
    acc = np.zeros_like(whole)
    for r in range(8):
        ctx.film_clear(); ctx.render(rank=r, world=8)
        acc += ctx.film()
    ctx.close()
    own_only = whole[..., 3] == sc.info["spp"]   # at 512 spp about a fifth of the pixels also get a neighbour's sample landing exactly on their edge
    assert own_only.mean() > 0.6
    assert np.array_equal(acc[own_only].view(np.uint32), whole[own_only].view(np.uint32))
    assert np.allclose(acc, whole, rtol=1e-6, atol=1e-7)


# ---------------------------------------------------------------- added after the last GPU call of round 3 (no GPU minutes left): the wavefront forms of BSSRDF probe
# chains and of grid media were verified on tools/hostemu against pbrt_ref's fixtures; these tests are confirmed on the MI355X by the round-end run and sit at the
# end of the file so that `pytest -x` reaches every hardware-validated test first
@pytest.mark.parametrize("name", ["vol_smoke", "vol_alpha"])
def test_split_form_matches_the_general_form_on_grid_media(name, monkeypatch):
    """Grid media in wavefront form (round 3): GridDensityMedium::Tr draws a data-dependent number of sampler dimensions (ratio tracking, media/grid.cpp:89-118)
    between a vertex's light sample and its continuation sample.  k_shade_vol<WAVE> stops such a vertex after the light sample, its shadow ray is walked to
    the end, then its MIS ray (k_trace<..., TR> + k_vol_tr_step with the path's sampler), then k_vol_continue samples the continuation from the dimension the
    walks left behind.  PBRT_AMD_VOL_SPLIT=0: the general form (every lane traces its own rays).  Both reproduce the reference's render with the same
    queries; a wrong order of draws would show at once (every later dimension of the path would change)."""
    fx = pa.read_pfm(os.path.join(G, "edge_%s.pfm" % name))
    out = {}
    for form in ("split", "general"):
        if form == "general":
            monkeypatch.setenv("PBRT_AMD_VOL_SPLIT", "0")
        else:
            monkeypatch.delenv("PBRT_AMD_VOL_SPLIT", raising=False)
        sc = pa.Scene(text=edge_scenes.scene(name))
        ctx = pa.Context(sc)
        ctx.timing_enable(True); ctx.counters_reset()
        ctx.render()
        t, cnt = ctx.timing(), ctx.counters()
        img = sc.film_image(ctx.film())
        frac, relmse = ol.image_metrics(img, fx)
        assert frac >= 0.995 and relmse <= 1e-4, (form, frac, relmse)
        out[form] = (img, {k: v[1] for k, v in t.items() if v[1]}, cnt)
        ctx.close()
    assert "anyhit" in out["split"][1] and "mis_closest" in out["split"][1] and "anyhit" not in out["general"][1]   # the walks ran / the lanes traced their own rays
    assert out["split"][2]["closest_rays"] == out["general"][2]["closest_rays"] and out["split"][2]["trace_guard_trips"] == 0   # segment for segment the same queries
    assert np.allclose(out["split"][0], out["general"][0], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name,flatten", [("sss_named", False), ("sss_coeff", False), ("sss_kd", False), ("sss_vol_iface", False), ("sss_vol_smoke", False), ("sss_inst", False), ("sss_inst", True)])
def test_walked_bssrdf_probes_match_the_general_form(name, flatten, monkeypatch):
    """Subsurface materials under Integrator "path" in wavefront form (round 3): the vertex's shadow / MIS rays take the plain traversals, the path parks,
    its probe chain is walked hit by hit through the queues (k_sss_probe_step + k_trace<2, ..., TR>: count, choose, walk again up to the chosen hit --
    SeparableBSSRDF::Sample_Sp, core/bssrdf.cpp:249-326) and k_sss_entry shades the entry vertex (path.cpp:160-174).  PBRT_AMD_VOL_INLINE=1: the
    per-lane form (every lane traces its own rays inside k_shade_vol).  Both reproduce the reference's render with the same rays; sss_inst walks
    its chains through TransformedPrimitives; sss_kd is a KdSubsurfaceMaterial under "volpath" in haze (homogeneous media, no interfaces: closed-form
    transmittance on the queued rays, the chain carries its media for the entry vertex); sss_vol_iface adds a bank of fog behind a BSDF-less box that cuts
    through the subsurface object (the direct-lighting rays of both vertices are walked through the interface, the chains cross it); sss_vol_smoke puts a GRID
    medium around it: the subsurface vertex and the entry vertex are both shaded in two stages around their walks (k_vol_continue), the chain in between."""
    monkeypatch.setenv("PBRT_AMD_INSTANCING", "0" if flatten else "1")
    fx = pa.read_pfm(os.path.join(G, "edge_%s.pfm" % name))
    out = {}
    for form in ("walked", "general"):
        if form == "general":
            monkeypatch.setenv("PBRT_AMD_VOL_INLINE", "1")
        else:
            monkeypatch.delenv("PBRT_AMD_VOL_INLINE", raising=False)
        sc = pa.Scene(text=edge_scenes.scene(name))
        ctx = pa.Context(sc)
        ctx.timing_enable(True); ctx.counters_reset()
        ctx.render()
        t, cnt = ctx.timing(), ctx.counters()
        img = sc.film_image(ctx.film())
        frac, relmse = ol.image_metrics(img, fx)
        assert (frac >= 0.995 and relmse <= 1e-4) if not flatten else (frac >= 0.99 and relmse <= 5e-4), (form, frac, relmse)
        out[form] = (img, {k: v[1] for k, v in t.items() if v[1]}, cnt)
        ctx.close()
    assert "anyhit" in out["walked"][1] and "mis_closest" in out["walked"][1] and "anyhit" not in out["general"][1]   # the queues ran / the lanes traced their own rays
    w, g = out["walked"][2], out["general"][2]
    # the same rays -- minus the second walk of a chain whose chosen hit is among the first PT_SSS_KEEP counted ones (the walked form keeps those; the
    # per-lane form always walks twice, the reference once with a list).  Under "volpath" the per-lane form's visibility queries are closest-hit queries.
    wt, gt = w["closest_rays"] + w["shadow_rays"], g["closest_rays"] + g["shadow_rays"]
    assert w["camera_rays"] == g["camera_rays"] and 0.9 * gt <= wt <= gt, (wt, gt)
    if name != "sss_kd":
        assert w["shadow_rays"] == g["shadow_rays"]
    assert out["walked"][2]["trace_guard_trips"] == 0
    assert np.allclose(out["walked"][0], out["general"][0], rtol=1e-4, atol=1e-5)


def _sss_mix_scene():
    """sss_coeff with its second subsurface object under a MixMaterial of two subsurface materials: MixMaterial evaluates m1 on *si itself (mixmat.cpp:45-64), so the
    mix's interaction carries m1's BSSRDF although the mix is no subsurface material -- its vertices belong to k_shade_vol too (found by the device fuzzer in round 4)"""
    text = edge_scenes.scene("sss_coeff")
    a = 'Material "subsurface" "rgb sigma_a" [.05 .2 .4] "rgb sigma_s" [6 5 3] "float g" [.3] "float eta" [1.5] "float uroughness" [.2] "float vroughness" [.1] "float scale" [3]\nShape "trianglemesh" "integer indices" [0 1 2] '
    assert text.count(a) == 1
    b = ('MakeNamedMaterial "sa" "string type" "subsurface" "rgb sigma_a" [.05 .2 .4] "rgb sigma_s" [6 5 3] "float g" [.3] "float eta" [1.5] "float uroughness" [.2] "float vroughness" [.1] "float scale" [3]\n'
         'MakeNamedMaterial "sb" "string type" "kdsubsurface" "rgb Kd" [.6 .3 .2] "rgb mfp" [.8 .5 .9] "float eta" [1.4] "float uroughness" [.3] "float vroughness" [.3]\n'
         'Material "mix" "rgb amount" [.4 .4 .4] "string namedmaterial1" "sa" "string namedmaterial2" "sb"\nShape "trianglemesh" "integer indices" [0 1 2] ')
    return text.replace(a, b)


@pytest.mark.parametrize("name", ["sss_named", "sss_coeff", "sss_inst", "sss_mix"])
def test_only_the_vertices_on_bssrdf_materials_go_to_the_volumetric_shading_kernel(name, monkeypatch):
    """Round 4: under Integrator "path" the material sort puts the BSSRDF materials' keys last and the sorted queue is shaded in two launches -- k_shade for the
    ordinary vertices (PathIntegrator::Li's loop body, path.cpp:64-188), k_shade_vol for the vertices on BSSRDF materials (path.cpp:153-174).  PBRT_AMD_SSS_ROUTE=0
    sends every vertex through k_shade_vol as before: the same samples, the same rays, the same image."""
    text = _sss_mix_scene() if name == "sss_mix" else edge_scenes.scene(name)
    fx = None if name == "sss_mix" else pa.read_pfm(os.path.join(G, "edge_%s.pfm" % name))
    if fx is None:   # (no pbrt_ref fixture of this one: the oracle, which reproduces the other three, is the judge)
        sc0 = pa.Scene(text=text)
        fx = sc0.film_image(ol.render(sc0, nthreads=4)[0])
    out = {}
    for form in ("routed", "all_vol"):
        if form == "all_vol":
            monkeypatch.setenv("PBRT_AMD_SSS_ROUTE", "0")
        else:
            monkeypatch.delenv("PBRT_AMD_SSS_ROUTE", raising=False)
        sc = pa.Scene(text=text)
        ctx = pa.Context(sc)
        ctx.timing_enable(True); ctx.counters_reset()
        ctx.render()
        t, cnt = ctx.timing(), ctx.counters()
        img = sc.film_image(ctx.film())
        frac, relmse = ol.image_metrics(img, fx)
        assert frac >= 0.995 and relmse <= 1e-4, (form, frac, relmse)
        out[form] = (img, {k: v[1] for k, v in t.items() if v[1]}, cnt)
        ctx.close()
    assert out["routed"][1]["shade"] > out["all_vol"][1]["shade"]   # two shading launches per bounce instead of one
    for k in ("camera_rays", "closest_rays", "shadow_rays", "path_segments"):
        assert out["routed"][2][k] == out["all_vol"][2][k], k
    assert out["routed"][2]["trace_guard_trips"] == 0
    assert np.allclose(out["routed"][0], out["all_vol"][0], rtol=1e-5, atol=1e-6)
    print("routed vs all-vol: %.4f of the pixels bit-identical" % float((out["routed"][0].view(np.uint32) == out["all_vol"][0].view(np.uint32)).all(-1).mean()))


def test_the_tail_of_the_probe_walk_lists_its_hits_instead_of_walking_twice(tmp_path, monkeypatch):
    """k_sss_probe_tail keeps the counted hits of a chain's first walk in a per-thread list (SssLog) and steps to the chosen one, as the reference's linked list does
    (bssrdf.cpp:285-314; the default since round 5, measured in profiles/r05_a_*); with PBRT_AMD_SSS_LOG=0 it walks the chain a second time up to the chosen hit.  Same image bit for bit, fewer probe segments -- on the reduced
    subsurface stand-in, whose quad soups give chains of tens of hits."""
    out = {}
    for form in ("list", "twice"):
        monkeypatch.setenv("PBRT_AMD_SSS_LOG", "1" if form == "list" else "0")
        sc = _config_scene("sanmiguel_subsurface", str(tmp_path))
        ctx = pa.Context(sc)
        ctx.counters_reset()
        ctx.render()
        out[form] = (ctx.film().copy(), ctx.counters())
        ctx.close()
    assert np.array_equal(out["list"][0].view(np.uint32), out["twice"][0].view(np.uint32))
    assert out["list"][1]["closest_rays"] < out["twice"][1]["closest_rays"]
    assert out["list"][1]["trace_guard_trips"] == 0 and out["twice"][1]["trace_guard_trips"] == 0
    for k in ("camera_rays", "shadow_rays", "path_segments"):
        assert out["list"][1][k] == out["twice"][1][k], k


@pytest.mark.parametrize("name", ["sanmiguel_subsurface", "sanmiguel_smokebox"])
def test_baseline_config_reduced_with_subsurface_materials_and_with_a_grid_medium(name, tmp_path):
    """the reduced C3 stand-in with three kdsubsurface materials (bench.py --subsurface: walked probe chains) and with a heterogeneous medium behind a
    BSDF-less box (bench.py --smokebox: the split form), against the oracle -- same checks as test_baseline_configs_reduced"""
    test_baseline_configs_reduced(name, "bvh4q", tmp_path)


@pytest.mark.launches_processes
def test_reference_host_drives_the_device_with_the_maxmindist_sampler(tmp_path):
    """Sampler "maxmindist" (ABI v12, MI_SAMPLER_MAXMIN): the reference-side binding hands the generator matrix of the reference's own MaxMinDistSampler
    (CMaxMinDist[log2 spp], samplers/maxmin.h:74-77) over with the scene; the device draws the first 2D dimension from it and the rest like 02sequence
    (k_pix_start_pixel), tile-serially.  Fixture: pbrt_ref's render (tools/gen_golden.py)."""
    test_reference_host_drives_the_device("sampler_maxmin", tmp_path)
