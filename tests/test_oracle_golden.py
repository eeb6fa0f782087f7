"""CPU: pin the oracle (CPU restatement) against the REAL reference -- golden vectors dumped from the reference's own
classes (tools/gen_golden.py + oracle/ref_build/ref_probe.cpp) and reference renders -- and against the
known-answer properties the reference's unit tests state (src/tests/sampling.cpp, src/tests/shapes.cpp)."""
import os
import struct

import numpy as np
import pytest

import oracle_lib as ol

pa = ol.pa
G = os.path.join(ol.ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def vec(built):
    return np.load(os.path.join(G, "ref_vectors.npz"))


def test_sobol_sample_float_matches_reference(vec):
    s = vec["sobol_samples"]
    got = np.array([ol.lib().oracle_sobol_sample_float(int(i), int(d)) for i, d in zip(s["i"], s["d"])], dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), s["v"].view(np.uint32))


def test_sobol_known_answer_reverse_bits(built):
    """LowDiscrepancy.Sobol (tests/sampling.cpp:120-136): dimension 0 is the bit-reversed index * 2^-32."""
    for i in range(8192):
        rev = int("{:032b}".format(i)[::-1], 2)
        assert ol.lib().oracle_sobol_sample_float(i, 0) == np.float32(min(rev * 2.0 ** -32, float(np.nextafter(np.float32(1), np.float32(0)))))


def test_sobol_interval_to_index_matches_reference(vec):
    s = vec["sobol_index"]
    got = np.array([ol.lib().oracle_sobol_interval_to_index(int(m), int(f), int(x), int(y)) for m, f, x, y in zip(s["m"], s["frame"], s["px"], s["py"])],
                   dtype=np.uint64)
    assert np.array_equal(got, s["idx"])


def test_sobol_sampler_stream_matches_reference(vec):
    """SobolSampler (samplers/sobol.cpp) Get1D stream incl. the pixel remap of dims 0/1, 400x300 film, 16 spp."""
    sc = pa.Scene(text='Film "image" "integer xresolution" [400] "integer yresolution" [300] "string filename" "x.pfm"\n'
                       'Sampler "sobol" "integer pixelsamples" [16]\nWorldBegin\nWorldEnd\n')
    rows = vec["sobol_sampler"]
    for (px, py) in sorted(set(zip(rows["px"].tolist(), rows["py"].tolist()))):
        sel = rows[(rows["px"] == px) & (rows["py"] == py)]
        got, _ = ol.sobol(sc, px, py, 16, 24)
        assert np.array_equal(got[sel["s"]].view(np.uint32), sel["u"].view(np.uint32)), (px, py)


def test_halton_sampler_stream_matches_reference(vec):
    """HaltonSampler (samplers/halton.cpp, pbrt's default sampler): Get1D stream of the class itself -- pixel offsets by the
    Chinese-remainder construction, base-2/3 pixel dimensions, PCG32-shuffled digit permutations for dims >= 2; 6 spp."""
    sc = pa.Scene(text='Film "image" "integer xresolution" [400] "integer yresolution" [300] "string filename" "x.pfm"\n'
                       'Sampler "halton" "integer pixelsamples" [6]\nWorldBegin\nWorldEnd\n')
    rows = vec["halton_sampler"]
    assert len(rows) == 36
    for (px, py) in sorted(set(zip(rows["px"].tolist(), rows["py"].tolist()))):
        sel = rows[(rows["px"] == px) & (rows["py"] == py)]
        got, _ = ol.sobol(sc, px, py, 6, 24)   # "sobol" = the scene's GlobalSampler
        assert np.array_equal(got[sel["s"]].view(np.uint32), sel["u"].view(np.uint32)), (px, py)


def test_sobol_elementary_intervals(built):
    """ElementaryIntervals (tests/sampling.cpp:139-188): the first 2^k samples of a pixel stratify every 2^i x 2^j grid."""
    sc = pa.Scene(text='Film "image" "integer xresolution" [64] "integer yresolution" [64] "string filename" "x.pfm"\n'
                       'Sampler "sobol" "integer pixelsamples" [64]\nWorldBegin\nWorldEnd\n')
    u, _ = ol.sobol(sc, 5, 9, 64, 2)
    for k in range(0, 7):
        n = 1 << k
        for i in range(k + 1):
            nx, ny = 1 << i, 1 << (k - i)
            cells = set((int(x * nx), int(y * ny)) for x, y in u[:n])
            assert len(cells) == n, (k, i)


def test_triangle_intersect_matches_reference(vec):
    """Triangle::Intersect on the reference's unit-test constructions (BadCases, Reintersect, vertex/edge-aimed rays)."""
    t = vec["triangles"]
    assert t["hit"][0] == 0   # Triangle.BadCases: known answer = miss
    nh = 0
    for r in t:
        p = r["p"].reshape(3, 3)
        hit, th, b = ol.triangle_intersect(p[0], p[1], p[2], r["o"], r["d"], float(r["tmax"]))
        assert hit == bool(r["hit"])
        if hit:
            nh += 1
            assert np.float32(th).view(np.uint32) == r["t"].view(np.uint32)
            assert b[2] == r["uv"][1]                                            # v = b0*0 + b1*0 + b2*1 = b2 (default uvs; -0 == +0)
            assert np.float32(np.float32(b[1]) + np.float32(b[2])) == r["uv"][0]   # u = b1 + b2
    assert nh > 500


def test_reintersect_property(vec):
    """Triangle.Reintersect (tests/shapes.cpp:154-205): rays spawned from a hit (SpawnRay/SpawnRayTo with pError and
    OffsetRayOrigin) never re-hit the triangle -- the reference dump must say so, and the oracle agrees above."""
    t = vec["triangles"]
    spawned = t[(t["tmax"] > 0.99) & (t["tmax"] < 1.0)]   # SpawnRayTo rays: tMax = 1 - ShadowEpsilon
    assert len(spawned) > 300 and spawned["hit"].sum() == 0


def test_distribution1d_matches_reference(vec):
    raw = vec["distribution1d"].tobytes()
    off = 0
    for _ in range(8):
        (n,) = struct.unpack_from("<i", raw, off); off += 4
        func = np.frombuffer(raw, "<f4", n, off); off += 4 * n
        cdf = np.frombuffer(raw, "<f4", n + 1, off); off += 4 * (n + 1)
        (fi,) = struct.unpack_from("<f", raw, off); off += 4
        (m,) = struct.unpack_from("<i", raw, off); off += 4
        c2, fi2 = ol.distribution1d(func)
        assert np.array_equal(c2.view(np.uint32), cdf.view(np.uint32)) and np.float32(fi2) == np.float32(fi)
        for _k in range(m):
            u, idx, pdf = struct.unpack_from("<fif", raw, off); off += 12
            i2, p2 = ol.sample_discrete(func, cdf, fi, u)
            assert i2 == idx and np.float32(p2) == np.float32(pdf)


@pytest.mark.parametrize("name,w,h,spp,strategy", [("cornell", 64, 64, 1, None), ("cornell", 64, 64, 8, None), ("materials", 96, 72, 1, None), ("materials", 96, 72, 16, None),
                                                    ("cornell", 64, 64, 4, "spatial"), ("materials", 96, 72, 4, "spatial"),
                                                    ("cornell", 64, 48, 4, "gaussian"), ("cornell", 64, 48, 4, "mitchell"),
                                                    ("cornell", 64, 48, 4, "triangle"), ("cornell", 64, 48, 4, "sinc"),
                                                    ("cornell", 64, 64, 6, "halton"), ("materials", 96, 72, 5, "halton")])
def test_oracle_render_matches_reference_image(built, name, w, h, spp, strategy):
    """Whole pipeline vs the reference's own render (lossless PFM fixture).  1 spp = per-camera-sample radiance.
    Tolerance: max |d| <= 2e-6 (1 + |ref|): the only differences are last-ulp film-sum / libm effects."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(ol.ROOT, "tools", "gen_golden.py"))
    gg = importlib.util.module_from_spec(spec); spec.loader.exec_module(gg)
    sc = pa.Scene(text=gg.scene_text(name, w, h, spp, strategy))
    rgbw, cnt, _ = ol.render(sc, nthreads=4)
    img = sc.film_image(rgbw)
    ref = pa.read_pfm(os.path.join(G, "%s_%dx%d_%dspp%s.pfm" % (name, w, h, spp, "_" + strategy if strategy else "")))
    assert img.shape == ref.shape
    # wide filters: a pixel sums contributions of several tiles, in a different order than the reference's tile merge
    wide = strategy in ("gaussian", "mitchell", "triangle", "sinc")
    tol = 1e-5 if wide else 2e-6
    assert np.all(np.abs(img - ref) <= tol * (1 + np.abs(ref))), float(np.abs(img - ref).max())
    if not wide: assert cnt["camera_rays"] == w * h * spp   # wide filters: sample bounds exceed the image


def test_oracle_vs_live_reference_binary(built, tmp_path):
    """When oracle/_ref/pbrt_ref exists (built here from /root/reference), render a scene variant no fixture covers."""
    if not ol.have_ref():
        pytest.skip("oracle/_ref/pbrt_ref not built in this environment")
    text = open(os.path.join(ol.ROOT, "scenes", "cornell.pbrt")).read().replace("[400] \"integer yresolution\" [400]", "[80] \"integer yresolution\" [48]")
    text = text.replace('"integer pixelsamples" [8]', '"integer pixelsamples" [4]').replace('"uniform"', '"power"')
    f = tmp_path / "c.pbrt"; f.write_text(text)
    ref = ol.run_ref(str(f), str(tmp_path / "ref.pfm"), nthreads=4)
    sc = pa.Scene(str(f))
    rgbw, _, _ = ol.render(sc, nthreads=4)
    img = sc.film_image(rgbw)
    assert np.all(np.abs(img - ref) <= 2e-6 * (1 + np.abs(ref)))


import edge_scenes


@pytest.mark.parametrize("name", edge_scenes.NAMES + edge_scenes.C5_NAMES + edge_scenes.TEX_NAMES + edge_scenes.TEX_ORACLE_ONLY)
def test_oracle_edge_cases_match_reference(built, name):
    """Edge cases of the path (tests/edge_scenes.py): constant infinite light (escaped rays, light sampling, single-light ->
    uniform substitution), thin lens, crop window + pixel bounds, luminance clamp, empty world, single-leaf BVH with a
    degenerate triangle, a crop of configs[4]'s 3840 x 2160 / 512 spp frame (33-bit Sobol' indices end to end); and the textured scenes of row f2 (PNG / TGA / PFM image maps with EWA and trilinear filtering and
    all wrap modes, every procedural texture class, the four 2D mappings, bump maps, alpha / shadow-alpha masks, textured
    parameters of all nine materials incl. mix) -- oracle vs the reference's render."""
    sc = pa.Scene(text=edge_scenes.scene(name))
    rgbw, _, _ = ol.render(sc, nthreads=4)
    img = sc.film_image(rgbw)
    ref = pa.read_pfm(os.path.join(G, "edge_%s.pfm" % name))
    assert img.shape == ref.shape
    # "instances": flattened by default -- same surfaces, other roundings (the two-level form is pinned bit for bit below); the other four:
    # <= 0.2 % of the pixels differ in the last place (order of the film sums), everything else is bit-identical
    # "c5_crop": at 512 spp a fifth of the pixels also receive a neighbour's sample that lands exactly on their edge; where that neighbour sits in
    # another tile the reference adds the two tiles' sums after converting each to XYZ (Film::MergeFilmTile), this film adds them as RGB: last place
    if name in ("instances", "dof", "clamp", "onetri", "tex_dof", "c5_crop"):
        assert np.all(np.abs(img - ref) <= 2e-6 * (1 + np.abs(ref))), float(np.abs(img - ref).max())
    else:
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), float(np.abs(img - ref).max())   # bit for bit


@pytest.mark.parametrize("name", edge_scenes.SAMPLER_NAMES)
def test_oracle_tile_serial_samplers_match_reference(built, name):
    """Sampler "random" / "stratified" / "02sequence" ("lowdiscrepancy") -- ABI v11.  Their values come from ONE PCG32 stream per 16 x 16 tile
    (sampler->Clone(seed = tile index), integrator.cpp:246-248; core/sampler.cpp:100-135, samplers/{random,stratified,zerotwosequence}.cpp,
    core/rng.h): what a sample receives depends on how many numbers every earlier sample of its tile drew.  The oracle walks the tiles in the
    reference's order; the renders (partial tiles, depth of field, crop window + pixel bounds, volpath in fog, one precomputed dimension only,
    no jitter, 3 -> 4 samples) equal pbrt_ref's bit for bit -- including the order in which PixelSampler::Get2D's `Point2f(rng.UniformFloat(),
    rng.UniformFloat())` draws its two numbers in a g++ build (second argument first)."""
    sc = pa.Scene(text=edge_scenes.scene(name))
    rgbw, _, _ = ol.render(sc, nthreads=4)
    img = sc.film_image(rgbw)
    ref = pa.read_pfm(os.path.join(G, "edge_%s.pfm" % name))
    assert img.shape == ref.shape
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), float(np.abs(img - ref).max())


@pytest.mark.parametrize("name", edge_scenes.VOL_NAMES)
def test_oracle_volpath_matches_reference(built, name):
    """SURVEY.md s.8 row f4, Integrator "volpath" (integrators/volpath.cpp) with participating media: a chromatic homogeneous medium
    around camera and scene (HG g = .4), a heterogeneous grid medium inside a box without a BSDF (delta tracking, ratio tracking with
    its Russian roulette, the reference's medium-space ray) next to an infinite light, a preset-coefficient medium inside a glass box
    (an interface with a BSDF) in haze, and volpath without any medium (unconditional light sample, Intersect-based visibility).
    Host (MakeNamedMedium / MediumInterface / camera medium) + oracle against the reference's renders, BIT FOR BIT.  The device refuses
    "volpath" scenes until it has the medium kernels (tested in test_host.py)."""
    sc = pa.Scene(text=edge_scenes.scene(name))
    img = sc.film_image(ol.render(sc, nthreads=4)[0])
    ref = pa.read_pfm(os.path.join(G, "edge_%s.pfm" % name))
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), float(np.abs(img - ref).max())


@pytest.mark.parametrize("name", edge_scenes.SSS_NAMES)
def test_oracle_subsurface_matches_reference(built, name):
    """SURVEY.md s.8 row f4, the BSSRDF branch (path.cpp:153-174, volpath.cpp:153-180): SubsurfaceMaterial by measured name and by explicit
    coefficients (smooth and rough boundary, two material objects of equal parameters = two different materials for the probe rays),
    KdSubsurfaceMaterial with a textured reflectance inverted per hit, under "path" and under "volpath" in a medium.  Host (beam-diffusion
    table of the material constructor, one slot per material object) + oracle (TabulatedBSSRDF: spline sampling of the radius, the three
    projection axes, the probe-segment hit chain, the adapter lobe at the exit point; sampler dimensions in the order g++ evaluates the
    reference's call arguments) against the reference's renders, BIT FOR BIT.  The device refuses such scenes so far."""
    sc = pa.Scene(text=edge_scenes.scene(name))
    img = sc.film_image(ol.render(sc, nthreads=4)[0])
    ref = pa.read_pfm(os.path.join(G, "edge_%s.pfm" % name))
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), float(np.abs(img - ref).max())


@pytest.mark.parametrize("name", edge_scenes.INSTANCE_NAMES)
def test_oracle_object_instancing_both_ways(built, name, monkeypatch):
    """ObjectBegin / ObjectInstance.  Two-level mode (PBRT_AMD_INSTANCING=1: the reference's own structure -- one
    BVHAccel per object, TransformedPrimitive leaves in the top-level BVH, rays transformed into the object's space, interactions
    transformed back): the oracle must reproduce the reference's render BIT FOR BIT, including a sphere and an alpha-masked, textured,
    shading-normal mesh inside an object, a mirroring instance transform and a one-triangle object.  Default mode (instances flattened
    to world-space copies, what the device renders): the same surfaces with different roundings -- image criterion of the GPU tests."""
    ref = pa.read_pfm(os.path.join(G, "edge_%s.pfm" % name))
    monkeypatch.delenv("PBRT_AMD_INSTANCING", raising=False)   # two-level is the default
    sc = pa.Scene(text=edge_scenes.scene(name))
    img = sc.film_image(ol.render(sc, nthreads=4)[0])
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    monkeypatch.setenv("PBRT_AMD_INSTANCING", "0")              # the flattening option
    flat = pa.Scene(text=edge_scenes.scene(name))
    assert flat.info["n_tris"] > sc.info["n_tris"]        # copies instead of references
    img2 = flat.film_image(ol.render(flat, nthreads=4)[0])
    frac, relmse = ol.image_metrics(img2, ref)
    assert frac >= 0.995 and relmse <= 1e-4, (frac, relmse)


def _killeroo_96():
    text = open(os.path.join(ol.ROOT, "scenes", "killeroo.pbrt")).read()
    text = text.replace('[700] "integer yresolution" [700]', '[96] "integer yresolution" [96]')
    return text.replace("killeroo_geo/", os.path.join(ol.ROOT, "scenes", "killeroo_geo") + "/")


def test_oracle_killeroo_simple_matches_the_reference_scene(built):
    """scenes/killeroo.pbrt (PLY geometry subdivided by this host) against the reference's UNMODIFIED killeroo-simple.pbrt rendered by the
    reference: Sphere area light, Halton sampler, Loop subdivision, plastic + matte."""
    sc = pa.Scene(text=_killeroo_96())
    img = sc.film_image(ol.render(sc, nthreads=4)[0])
    ref = pa.read_pfm(os.path.join(G, "killeroo_simple_96x96_reference.pfm"))
    assert np.all(np.abs(img - ref) <= 2e-6 * (1 + np.abs(ref))), float(np.abs(img - ref).max())


def test_sphere_intersect_matches_reference(vec):
    """Sphere::Intersect (EFloat quadratic, clipping, interaction through ObjectToWorld) on the FullSphere / PartialSphere constructions
    of the reference's tests (tests/shapes.cpp:376-497) plus transformed spheres: hit decision, tHit, p, pError, n -- bit for bit; and the
    property those tests assert: rays spawned from a hit into the normal's hemisphere of a convex sphere do not re-intersect it."""
    rows = vec["spheres"]
    assert len(rows) > 1000 and rows["hit"].sum() > 200
    sp, rays = ol.sphere_records(rows)
    h = ol.sphere_intersect(sp, rays)
    assert np.array_equal(h["hit"], rows["hit"])
    hit = rows["hit"] == 1
    for k in ("t", "p", "p_error", "n"):
        assert np.array_equal(h[k][hit].view(np.uint32), rows[k][hit].view(np.uint32)), k


def test_bxdfs_match_reference_classes(vec):
    """f, Pdf and Sample_f of every lobe the path carries (core/reflection.{h,cpp}: Lambertian R/T, OrenNayar, SpecularReflection /
    Transmission, FresnelSpecular, MicrofacetReflection / Transmission with TrowbridgeReitz, FresnelBlend; Fresnel NoOp / dielectric /
    conductor) against the reference classes on random directions (2400 records dumped by ref_probe) -- bit for bit."""
    rows = vec["bxdfs"]
    assert len(rows) == 2400
    out = ol.bxdf_eval(rows)
    for k in ("f", "pdf", "wi_s", "pdf_s", "f_s", "type_s"):
        a, b = out[k], rows[k]
        same = (a.view(np.uint32) == b.view(np.uint32)) | (a == b)   # -0 == +0
        assert same.all(), (k, int((~same).sum()))


@pytest.mark.parametrize("name", edge_scenes.FURNACE_NAMES)
def test_oracle_furnace_scenes(name, built):
    """The reference's analytic scenes (src/tests/analytic_scenes.cpp:71-203; CheckSceneAverage :55-68): inside a closed unit sphere the radiance is 1, so the
    mean over all pixels and channels must be 1.0 +- 0.02 -- with the samplers (Sobol' / Halton, 256 spp) and integrators (path / volpath, depth 8) this path
    carries.  The Sobol' / path render is also compared with the reference's own image of the same file (fixture): bit-exact."""
    for sampler in ("sobol", "halton"):
        for integrator in ("path", "volpath"):
            sc = pa.Scene(text=edge_scenes.furnace_scene(name, sampler, integrator))
            img = sc.film_image(ol.render(sc, nthreads=4)[0])
            assert abs(float(img.mean()) - 1.0) <= 0.02, (name, sampler, integrator, float(img.mean()))
            if (sampler, integrator) == ("sobol", "path"):
                ref = pa.read_pfm(os.path.join(G, "%s.pfm" % name))
                assert np.array_equal(img, ref), name


def _light_queries(recs):
    q = np.zeros(len(recs), dtype=pa.LIGHT_QUERY_DTYPE)
    q["light"] = np.arange(len(recs)) // 24
    q["p"] = recs["p"]; q["n"] = recs["n"]; q["u"] = recs["u"]; q["wi"] = recs["wi2"]
    return q


def test_light_sampling_matches_reference_classes(built):
    """Light::Sample_Li / Pdf_Li against the reference's own classes (oracle/ref_build/ref_probe.cpp -> tests/golden/light_vectors.npz): 160 lights
    (DiffuseAreaLight over random triangles -- the geometry of tests/shapes.cpp Triangle.Sampling :210-269 -- and over spheres, point and spot
    lights) x 24 reference points (surface points with a normal and medium points without): wi, pdf, Li, the VisibilityTester's shadow ray
    (SpawnRayTo -> OffsetRayOrigin -> NextFloatUp/Down, the arithmetic tests/fp_tests.cpp NextUpDownFloat :29-47 pins) and Pdf_Li of a second
    direction -- bit for bit."""
    recs = np.load(os.path.join(G, "light_vectors.npz"))["light_samples"]
    assert len(recs) == 3840
    sc = pa.Scene(text=edge_scenes.light_kat_scene(recs))
    assert sc.info["n_lights"] == len(recs) // 24
    o = ol.light_sample(sc, _light_queries(recs))
    assert (recs["pdf"] > 0).all() and (recs["pdf_b"][recs["kind"] == 1] > 0).all() and (recs["pdf_b"][recs["kind"] == 0] > 0).mean() > .5
    for k in ("wi", "pdf", "Li", "ray_o", "ray_d", "ray_tmax"):
        assert o[k].tobytes() == recs[k].tobytes(), k
    assert o["pdf_wi"].tobytes() == recs["pdf_b"].tobytes()
    assert (o["delta"] == (recs["kind"] >= 2)).all()
    # Triangle.Sampling's own property on the same records: 1 / pdf averaged over the light's samples estimates the solid angle of the
    # triangle, which must agree with Pdf_Li's view of it (pdf of the sampled direction == Pdf_Li of that direction)
    q2 = _light_queries(recs); q2["wi"] = recs["wi"]
    o2 = ol.light_sample(sc, q2)
    area = recs["kind"] < 2
    assert o2["pdf_wi"][area].tobytes() == recs["pdf_a"][area].tobytes()
    tri = recs["kind"] == 0
    assert np.allclose(o2["pdf_wi"][tri], recs["pdf"][tri], rtol=2e-3)


def test_next_float_up_down_known_answers(built):
    """tests/fp_tests.cpp NextUpDownFloat :29-47 on the restated NextFloatUp / NextFloatDown (through OffsetRayOrigin: a reference point with
    pError = 0 moves by exactly one ulp away from the surface in every component the offset touches -- geometry.h:1452-1459)."""
    recs = np.load(os.path.join(G, "light_vectors.npz"))["light_samples"]
    r = recs[(recs["kind"] == 2) & (np.abs(recs["n"]).sum(axis=1) > 0)]
    # pError = 0 -> offset 0 -> po = p, then each component with offset == 0 is left alone: the origin must be p itself
    assert r["ray_o"].tobytes() == r["p"].tobytes()
    for v, up, down in [(np.float32(1.0), np.float32(1.0000001), np.float32(0.99999994)), (np.float32(-0.0), np.float32(1e-45), np.float32(-1e-45)),
                        (np.float32(np.inf), np.float32(np.inf), np.float32(3.4028235e38)), (np.float32(-np.inf), np.float32(-3.4028235e38), np.float32(-np.inf))]:
        assert np.nextafter(v, np.float32(np.inf)) == up or v == np.inf
        assert ol.next_float(v, +1) == up and ol.next_float(v, -1) == down
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2**32, 100000, dtype=np.uint64).astype(np.uint32)
    f = bits.view(np.float32)
    f = f[np.isfinite(f)]
    upv, dnv = ol.next_float_array(f, +1), ol.next_float_array(f, -1)
    assert np.array_equal(upv, np.nextafter(f, np.float32(np.inf))) and np.array_equal(dnv, np.nextafter(f, np.float32(-np.inf)))


def test_scene_dependent_lights_match_reference_classes(built):
    """DistantLight and InfiniteAreaLight (constant -- the reference's 1 x 1 map -- and with a radiance map: MIPMap::Lookup + Distribution2D,
    lights/infinite.cpp:43-137), which need Light::Preprocess's world bound: Sample_Li (wi, pdf, Li, shadow ray to the scene's bounding
    sphere), Pdf_Li and Le of an escaped ray, 12 lights x 64 reference points from the reference's own classes -- bit for bit."""
    recs = np.load(os.path.join(G, "light_vectors.npz"))["scene_lights"]
    assert len(recs) == 768
    sc = pa.Scene(text=edge_scenes.scene_light_kat_scene(recs))
    assert sc.info["n_lights"] == 12
    q = np.zeros(len(recs), dtype=pa.LIGHT_QUERY_DTYPE)
    q["light"] = np.arange(len(recs)) // 64
    q["p"] = recs["p"]; q["n"] = recs["n"]; q["u"] = recs["u"]; q["wi"] = recs["wi2"]
    o = ol.light_sample(sc, q)
    assert (recs["pdf"] > 0).all() and (recs["pdf_b"][recs["kind"] >= 5] > 0).all() and (recs["le"][recs["kind"] >= 5] > 0).all()
    for k in ("wi", "pdf", "Li", "ray_o", "ray_d", "ray_tmax"):
        assert o[k].tobytes() == recs[k].tobytes(), k
    assert o["pdf_wi"].tobytes() == recs["pdf_b"].tobytes() and o["le_wi"].tobytes() == recs["le"].tobytes()


def test_camera_rays_match_reference_classes(built):
    """Sampler::GetCameraSample + PerspectiveCamera::GenerateRayDifferential (core/sampler.cpp:46-52, cameras/perspective.cpp:95-144) against the
    reference's own SobolSampler + PerspectiveCamera objects (ref_probe -> tests/golden/camera_vectors.npz): pinhole and thin-lens cameras, a
    non-square film, a crop window, a frame aspect ratio; 400 (pixel, sample) pairs each -- pFilm and the main ray bit for bit, weight 1."""
    recs = np.load(os.path.join(G, "camera_vectors.npz"))["camera_rays"]
    assert len(recs) == 1600 and (recs["weight"] == 1).all()
    for c in range(4):
        r = recs[recs["cfg"] == c]
        sc = pa.Scene(text=edge_scenes.camera_kat_scene(r[0]))
        rays, pf = ol.camera_rays(sc, np.stack([r["px"], r["py"]], 1).astype(np.int32), r["s"])
        assert pf.tobytes() == r["p_film"].tobytes(), c
        assert np.ascontiguousarray(rays["o"]).tobytes() == r["o"].tobytes() and np.ascontiguousarray(rays["d"]).tobytes() == r["d"].tobytes(), c


def test_bssrdf_radial_profile_and_phase_function_match_reference_classes(built):
    """Row f4 at stage level against the reference's own classes (ref_probe -> tests/golden/bssrdf_tables.npz): TabulatedBSSRDF::Sr, Sample_Sr, Pdf_Sr
    (core/bssrdf.cpp:199-233, 353-390: CatmullRomWeights, SampleCatmullRom2D) for 2 000 random coefficient triples on the (g = 0, eta = 1.33) table incl.
    a channel with sigma_t = 0; SubsurfaceFromDiffuse (InvertCatmullRom, :178-188); HenyeyGreenstein::p / Sample_p (core/medium.cpp:189-213) for 4 000
    (g, wo, wi, u) incl. |g| < 1e-3 -- every value bit for bit."""
    d = np.load(os.path.join(G, "bssrdf_tables.npz"))
    t = ol.bssrdf_table(d["tables"][0])
    assert float(d["tables"][0]["g"]) == 0.0
    r = d["radial"]
    sr, smp, pdf = ol.bssrdf_radial(t, float(d["tables"][0]["eta"]), r)
    assert (r["sr"] > 0).mean() > .5 and (r["sample_sr"] > 0).mean() > .9 and (r["pdf_sr"] > 0).mean() > .5
    assert sr.tobytes() == r["sr"].tobytes() and smp.tobytes() == r["sample_sr"].tobytes() and pdf.tobytes() == r["pdf_sr"].tobytes()
    sa, ss = ol.subsurface_from_diffuse(t, r["kd"], r["mfp"])
    assert sa.tobytes() == r["out_sigma_a"].tobytes() and ss.tobytes() == r["out_sigma_s"].tobytes()
    h = d["hg"]
    p, ws, ps = ol.hg(h)
    assert p.tobytes() == h["p"].tobytes() and ws.tobytes() == h["wi_s"].tobytes() and ps.tobytes() == h["p_s"].tobytes()
