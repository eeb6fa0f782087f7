"""CPU: the hand-over half of the reference-side binding of INTEGRATION.md s.2, compiled for real (SURVEY.md s.8 row b).

integration/flatten.h is the FlattenScene a pbrt-v3 maintainer would add (with integration/wavefrontpath.cpp, the `WavefrontPathIntegrator :
public Integrator` that renders the flattened description on the MI355X through the mi_* entry points -- exercised on the GPU box by
tests/test_gpu_parity.py::test_reference_host_drives_the_device).  Here its TEST-ONLY twin oracle/ref_build/flatcheck.cpp, built against the
UNMODIFIED reference (libpbrt_ref.a) into oracle/_ref/pbrt_ref_flatcheck, runs on the CPU: the reference's own main, parser, API state machine,
shape factories, Loop subdivision and BVH build run as they are, FlattenScene flattens the reference's `Scene` / `BVHAccel` /
`GeometricPrimitive` / `Material` (through the BxDFs its ComputeScatteringFunctions builds) / `Light` / camera / film / sampler objects to a
`mi_scene_desc`, and the CPU oracle renders THAT description.  The resulting image goes through the reference's own Film::MergeFilmTile / WriteImage
and must equal what `pbrt_ref` renders for the same file up to the rounding of the film sum (a pixel that also receives an edge sample
of a neighbour adds it in a different order: 1 ulp of the sum in < 0.1 % of the pixels)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol

pa = ol.pa
ROOT = ol.ROOT
REF = os.path.join(ROOT, "oracle", "_ref", "pbrt_ref")
STUB = os.path.join(ROOT, "oracle", "_ref", "pbrt_ref_flatcheck")   # TEST-ONLY twin of the binding (oracle/ref_build/flatcheck.cpp): the same FlattenScene, rendered by the CPU checker
ORACLE = os.path.join(ROOT, "oracle", "liboracle.so")

CASES = [("cornell", os.path.join(ROOT, "scenes", "cornell.pbrt"), []),
         ("materials", os.path.join(ROOT, "scenes", "materials.pbrt"), []),
         # the reference's own example scene, unmodified: Sphere area light, Halton sampler, Loop subdivision surfaces
         ("killeroo-simple", "/root/reference/scenes/killeroo-simple.pbrt", ["--cropwindow", "0.3", "0.7", "0.3", "0.7"])]


@pytest.mark.parametrize("name,scene,extra", CASES, ids=[c[0] for c in CASES])
def test_reference_scene_through_the_stub_equals_pbrt_ref(name, scene, extra, tmp_path):
    if not (os.access(REF, os.X_OK) and os.access(STUB, os.X_OK)):
        pytest.skip("oracle/_ref/pbrt_ref[_flatcheck] not built here (needs /root/reference)")
    if not os.path.exists(scene):
        pytest.skip("scene file not present: %s" % scene)
    a, b = str(tmp_path / "stub.pfm"), str(tmp_path / "ref.pfm")
    env = dict(os.environ, PBRT_AMD_BACKEND_LIB=ORACLE)
    cwd = os.path.dirname(scene)
    r1 = subprocess.run([STUB, "--quiet", "--nthreads", "4"] + extra + ["--outfile", a, scene], cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0 and os.path.exists(a), r1.stderr[-800:]
    r2 = subprocess.run([REF, "--quiet", "--nthreads", "4"] + extra + ["--outfile", b, scene], cwd=cwd, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0 and os.path.exists(b), r2.stderr[-800:]
    ia, ib = pa.read_pfm(a), pa.read_pfm(b)
    assert ia.shape == ib.shape and ib.mean() > 1e-3
    d = np.abs(ia - ib).max(-1)
    assert d.max() <= 5e-7, float(d.max())
    assert (d == 0).mean() >= 0.998, float((d == 0).mean())


# the same check over the edge-case scenes of tests/edge_scenes.py: what the stub hands over beyond triangles + area lights -- spot and
# infinite lights (the reference's own Lmap texels + Distribution2D, constant lights as its 1 x 1 map, transformed lights), spheres, thin
# lens, crop windows, luminance clamp, and row f4's participating media (HomogeneousMedium / GridDensityMedium, per-primitive
# MediumInterfaces incl. BSDF-less boundaries, the camera medium) with `Integrator "volpath"` bound to the stub as well
import edge_scenes

EDGE_CASES = ["infinite", "infinite_only", "infinite_xf", "envmap", "envmap_power", "spot", "spheres", "dof", "crop", "clamp", "onetri",
              # ABI v11: the reference's RandomSampler / StratifiedSampler / ZeroTwoSequenceSampler objects handed over (one PCG32 stream per tile)
              "sampler_random", "sampler_stratified", "sampler_strat_d1", "sampler_02sequence", "sampler_lowdisc_vol", "sampler_maxmin",
              "vol_fog", "vol_smoke", "vol_glass", "vol_none",
              # two-level instancing: the reference's TransformedPrimitives and per-object BVHAccels handed over as mi_instance / mi_object
              "instances", "vol_inst", "instances_one",
              # rows f2 / f4: the reference's Texture / MIPMap / TextureMapping objects as mi_texture nodes and mi_image pyramids, material
              # parameters as mi_material_desc, TriangleMesh alpha masks, SubsurfaceMaterial / KdSubsurfaceMaterial with their BSSRDFTable
              "tex_imagemap", "tex_procedural", "tex_noise", "tex_mappings", "tex_bump", "tex_alpha", "tex_materials", "tex_spheres", "tex_dof",
              "instances2", "vol_alpha", "sss_named", "sss_coeff", "sss_kd", "sss_inst", "sss_vol_iface", "sss_vol_smoke"]


def _edge_text(name):
    if name == "instances_one":   # + an object with a single primitive (no accelerator in the reference, api.cpp:1572-1580), mirrored
        one = ('ObjectBegin "one"\nMaterial "matte" "rgb Kd" [.2 .3 .8]\nShape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0  .8 0 0  .4 .9 0]\nObjectEnd\n'
               'AttributeBegin\nTranslate -.6 .05 -1.2\nRotate 10 0 1 0\nObjectInstance "one"\nAttributeEnd\n'
               'AttributeBegin\nTranslate 1.1 .05 -1.0\nRotate -35 0 1 0\nScale -1.5 .7 1\nObjectInstance "one"\nAttributeEnd\n')
        t = edge_scenes.scene("instances")
        k = t.index("ObjectBegin")
        return t[:k] + one + t[k:]
    return edge_scenes.scene(name)


@pytest.mark.parametrize("name", EDGE_CASES)
def test_edge_scene_through_the_stub_equals_pbrt_ref(name, tmp_path):
    if not (os.access(REF, os.X_OK) and os.access(STUB, os.X_OK)):
        pytest.skip("oracle/_ref/pbrt_ref[_flatcheck] not built here (needs /root/reference)")
    scene = str(tmp_path / "s.pbrt")
    open(scene, "w").write(_edge_text(name))
    a, b = str(tmp_path / "stub.pfm"), str(tmp_path / "ref.pfm")
    env = dict(os.environ, PBRT_AMD_BACKEND_LIB=ORACLE)
    r1 = subprocess.run([STUB, "--quiet", "--nthreads", "4", "--outfile", a, scene], env=env, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0 and os.path.exists(a), r1.stderr[-800:]
    r2 = subprocess.run([REF, "--quiet", "--nthreads", "4", "--outfile", b, scene], capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0 and os.path.exists(b), r2.stderr[-800:]
    ia, ib = pa.read_pfm(a), pa.read_pfm(b)
    assert ia.shape == ib.shape
    d = np.abs(ia - ib).max(-1)
    if name == "sampler_maxmin":
        # MaxMinDistSampler's first film sample of every pixel sits EXACTLY on the pixel's left edge (i / spp with i = 0): under the box filter it also counts for
        # the pixel to the left -- for the first column of a 16 x 16 tile that is a pixel of the NEIGHBOUR tile, whose FilmTile the reference converts to XYZ and
        # merges separately (film.cpp:117-130: XYZ(a) + XYZ(b)), where one film of RGB sums converts once (XYZ(a + b)): the last column and the last row of each
        # tile may differ by one ulp, every other pixel is bit-identical
        border = (np.arange(ia.shape[1]) % 16 == 15)[None, :] | (np.arange(ia.shape[0]) % 16 == 15)[:, None]   # (the sample sits on the pixel's CORNER: C * 0 = 0 too)
        assert (d[~border] == 0).all() and d.max() <= 1e-6 * max(1.0, float(np.abs(ib).max())), (float(d.max()), float((d[~border] == 0).mean()))
        return
    assert d.max() <= 5e-7, float(d.max())
    assert (d == 0).mean() >= 0.995, float((d == 0).mean())


def test_fast_samplers_through_the_stub_equals_pbrt_ref_with_sobol(tmp_path):
    """PBRT_AMD_FAST_SAMPLERS=1 in the reference-side binding (integration/wavefrontpath.cpp): a scene that names the stratified sampler is handed over with
    the reference's own SobolSampler at the same sample count -- the image equals pbrt_ref's render of the scene with Sampler "sobol" written into it."""
    if not (os.access(REF, os.X_OK) and os.access(STUB, os.X_OK)):
        pytest.skip("oracle/_ref/pbrt_ref[_flatcheck] not built here (needs /root/reference)")
    import re
    base = edge_scenes.scene("sampler_stratified")
    m = re.search(r'Sampler "stratified"[^\n]*', base)
    nx, ny = (int(v) for v in re.findall(r'"integer [xy]samples" \[(\d+)\]', m.group(0)))
    s1, s2 = str(tmp_path / "strat.pbrt"), str(tmp_path / "sobol.pbrt")
    open(s1, "w").write(base)
    open(s2, "w").write(base.replace(m.group(0), 'Sampler "sobol" "integer pixelsamples" [%d]' % (nx * ny)))
    a, b = str(tmp_path / "stub.pfm"), str(tmp_path / "ref.pfm")
    env = dict(os.environ, PBRT_AMD_BACKEND_LIB=ORACLE, PBRT_AMD_FAST_SAMPLERS="1")
    r1 = subprocess.run([STUB, "--quiet", "--nthreads", "4", "--outfile", a, s1], env=env, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0 and os.path.exists(a), r1.stderr[-800:]
    r2 = subprocess.run([REF, "--quiet", "--nthreads", "4", "--outfile", b, s2], capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0 and os.path.exists(b), r2.stderr[-800:]
    ia, ib = pa.read_pfm(a), pa.read_pfm(b)
    d = np.abs(ia - ib).max(-1)
    assert ia.shape == ib.shape and d.max() <= 5e-7 and (d == 0).mean() >= 0.995, (float(d.max()), float((d == 0).mean()))


@pytest.mark.parametrize("name", edge_scenes.TEX_NAMES + ["tex_dof"])
def test_texture_nodes_equal_the_reference_classes(name, tmp_path):
    """Row f2 at stage level against the reference's OWN texture classes: with PBRT_AMD_TEX_PROBE set the stub evaluates every Texture object the
    scene reaches with the reference's Texture<T>::Evaluate (ImageTexture + MIPMap EWA / trilinear, the procedural classes, the 2D / 3D mappings)
    at 2048 random interactions (sub-texel to many-texel footprints) and the oracle evaluates the node that object was flattened to
    (oracle_texture_eval) at the same interactions: every value bit for bit."""
    if not os.access(STUB, os.X_OK):
        pytest.skip("oracle/_ref/pbrt_ref_flatcheck not built here (needs /root/reference)")
    scene = str(tmp_path / "s.pbrt")
    open(scene, "w").write(edge_scenes.scene(name))
    rep = str(tmp_path / "probe.txt")
    env = dict(os.environ, PBRT_AMD_BACKEND_LIB=ORACLE, PBRT_AMD_TEX_PROBE=rep)
    r = subprocess.run([STUB, "--quiet", "--nthreads", "4", "--outfile", str(tmp_path / "o.pfm"), scene], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and os.path.exists(rep), r.stderr[-800:]
    rows = [l.split() for l in open(rep)]
    assert len(rows) >= 8
    for node, typ, spectrum, n, same, worst in rows:
        assert n == same, (name, "node", node, "type", typ, "identical", same, "of", n, "largest difference", worst)


HIT_PROBE_SCENES = (edge_scenes.NAMES + edge_scenes.TEX_NAMES + edge_scenes.VOL_NAMES + edge_scenes.SSS_NAMES + ["instances2", "heightfield", "nurbs"]
                    + ["file:cornell", "file:materials", "file:killeroo", "gen:sanmiguel:200000", "gen:bathroom:60000"])


@pytest.mark.parametrize("name", HIT_PROBE_SCENES)
def test_traversal_equals_the_reference_scene_intersect(name, tmp_path):
    """Rows a7-a12 against the reference's OWN Scene::Intersect / IntersectP on multi-primitive trees (SURVEY.md s.8c lists these as unpinned by any reference
    test): with PBRT_AMD_HIT_PROBE set the reference-side binding shoots 20 000 random rays (from outside and inside the bounds, finite and infinite
    tMax, normalised and unnormalised directions) through BVHAccel / TransformedPrimitive / Triangle / Sphere with alpha masks, and the oracle traverses
    the flattened description with the same rays: hit or miss, the hit distance, the geometric normal and the occlusion flag all bit for bit -- on every
    edge scene (two-level instancing, spheres, masks, tessellated height fields / NURBS) and on the 66 k-triangle killeroo, a 200 k-triangle
    San-Miguel-class and a 60 k-triangle bathroom-class scene."""
    if not os.access(STUB, os.X_OK):
        pytest.skip("oracle/_ref/pbrt_ref_flatcheck not built here (needs /root/reference)")
    if name.startswith("file:"):
        scene = os.path.join(ROOT, "scenes", name[5:] + ".pbrt")
    elif name.startswith("gen:"):
        _, kind, tris = name.split(":")
        scene = str(tmp_path / "g.pbrt")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_scenes.py"), kind, "--tris", tris, "--res", "64", "36", "--spp", "1", "--out", scene],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    else:
        scene = str(tmp_path / "s.pbrt")
        open(scene, "w").write(edge_scenes.scene(name))
    rep = str(tmp_path / "hits.txt")
    env = dict(os.environ, PBRT_AMD_BACKEND_LIB=ORACLE, PBRT_AMD_HIT_PROBE=rep)
    r = subprocess.run([STUB, "--quiet", "--quick", "--nthreads", "4", "--outfile", str(tmp_path / "o.pfm"), scene], env=env, capture_output=True, text=True, timeout=900)
    assert os.path.exists(rep), r.stderr[-800:]
    n, hits, same_flag, same_t, same_n, same_occ = [int(v) for v in open(rep).read().split()]
    assert n == 20000 and (hits > 1000 or name == "empty")
    assert same_flag == n and same_occ == n and same_t == hits and same_n == hits, (name, n, hits, same_flag, same_t, same_n, same_occ)


@pytest.mark.parametrize("name", edge_scenes.NAMES + edge_scenes.TEX_NAMES + edge_scenes.VOL_NAMES + edge_scenes.SSS_NAMES + ["instances2", "file:materials", "file:killeroo", "file:cornell"])
def test_bsdfs_at_hits_equal_the_reference_material_classes(name, tmp_path):
    """Rows a13-a16 against the reference's OWN Material / BSDF classes: with PBRT_AMD_BSDF_PROBE set the reference-side binding builds the BSDF at the first hit
    of 20 000 random rays with SurfaceInteraction::ComputeScatteringFunctions (texture evaluation, bump mapping, the material's lobe list, the shading
    frame; every material incl. mix, uber with opacity, subsurface boundaries) and evaluates BSDF::f, Pdf and Sample_f for a random direction / sample;
    the oracle does the same on the flattened description (oracle_bsdf_at_hit): hit state (miss / BSDF / null BSDF), number of components, f, Pdf and the
    sampled direction, pdf, value and lobe type -- all bit for bit."""
    if not os.access(STUB, os.X_OK):
        pytest.skip("oracle/_ref/pbrt_ref_flatcheck not built here (needs /root/reference)")
    if name.startswith("file:"):
        scene = os.path.join(ROOT, "scenes", name[5:] + ".pbrt")
    else:
        scene = str(tmp_path / "s.pbrt")
        open(scene, "w").write(edge_scenes.scene(name))
    rep = str(tmp_path / "bsdf.txt")
    env = dict(os.environ, PBRT_AMD_BACKEND_LIB=ORACLE, PBRT_AMD_BSDF_PROBE=rep)
    r = subprocess.run([STUB, "--quiet", "--quick", "--nthreads", "4", "--outfile", str(tmp_path / "o.pfm"), scene], env=env, capture_output=True, text=True, timeout=900)
    assert os.path.exists(rep), r.stderr[-800:]
    n, with_bsdf, same_state, same_count, same_f, same_pdf, same_sample = [int(v) for v in open(rep).read().split()]
    assert n == 20000 and same_state == n and (with_bsdf > 1000 or name == "empty")
    assert same_count == with_bsdf and same_f == with_bsdf and same_pdf == with_bsdf and same_sample == with_bsdf, (name, with_bsdf, same_count, same_f, same_pdf, same_sample)


def test_the_binding_itself_has_no_cpu_render_path(tmp_path):
    """integration/wavefrontpath.cpp binds the mi_* entry points and nothing else: handed the CPU checker as its library it refuses
    ("lacks mi_ctx_create"), handed libpbrt_amd.so on a box without a GPU it reports mi_ctx_create's error; no image is written either way.  And
    the sources under integration/ do not mention the checker at all."""
    binding = os.path.join(ROOT, "oracle", "_ref", "pbrt_ref_wavefront")
    if not os.access(binding, os.X_OK):
        pytest.skip("oracle/_ref/pbrt_ref_wavefront not built here (needs /root/reference)")
    for f in os.listdir(os.path.join(ROOT, "integration")):
        assert "oracle" not in open(os.path.join(ROOT, "integration", f)).read(), f
    scene = os.path.join(ROOT, "scenes", "cornell.pbrt")
    out = str(tmp_path / "o.pfm")
    env = {k: v for k, v in os.environ.items() if not k.startswith("PBRT_AMD_")}
    r = subprocess.run([binding, "--quiet", "--outfile", out, scene], env=dict(env, PBRT_AMD_DEVICE_LIB=ORACLE), capture_output=True, text=True, timeout=300)
    assert "lacks mi_ctx_create" in (r.stderr + r.stdout) and not os.path.exists(out)
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([binding, "--quiet", "--outfile", out, scene], env=dict(env, PBRT_AMD_DEVICE_LIB=pa.DEVICE_LIB), capture_output=True, text=True, timeout=300)
        assert "no HIP device available" in (r.stderr + r.stdout) and not os.path.exists(out)


def _host_bvh(sc):
    """(nodes as raw bytes, ordered triangles as 9 floats each) of a scene the host library built"""
    import ctypes as C
    raw = (C.c_uint64 * 12).from_address(sc.desc)   # [abi|n_verts] P N UV [n_tris] idx mesh light [n_meshes] meshes [n_nodes] nodes
    n_verts, n_tris, n_nodes = raw[0] >> 32, raw[4] & 0xffffffff, raw[10] & 0xffffffff
    P = np.ctypeslib.as_array((C.c_float * (3 * n_verts)).from_address(raw[1])).reshape(-1, 3)
    idx = np.ctypeslib.as_array((C.c_uint32 * (3 * n_tris)).from_address(raw[5]))
    nodes = bytes((C.c_uint8 * (32 * n_nodes)).from_address(raw[11]))
    return n_nodes, nodes, P[idx].reshape(n_tris, 9).copy()


@pytest.mark.parametrize("method,maxprims", [("sah", 4), ("hlbvh", 4), ("hlbvh", 1), ("hlbvh", 9), ("middle", 4), ("equal", 2)])
def test_host_bvh_build_equals_the_reference_build_node_for_node(method, maxprims, tmp_path):
    """Every `Accelerator "bvh" "string splitmethod"` the reference knows (accelerators/bvh.cpp:740-760), built by the host library and by the reference's own
    BVHAccel (one thread: HLBVH hands out leaf offsets from an atomic counter inside a ParallelFor) on the same scene -- two meshes of scattered triangles, one
    with many coinciding centroids -- and compared as it crosses the boundary: the LinearBVHNode array byte for byte (bounds, child / primitive offsets, counts,
    split axes) and the ordered primitive list triangle for triangle.  Round 5 built SAH for "hlbvh" with a warning: same hits except for ties, different tree."""
    if not os.access(STUB, os.X_OK):
        pytest.skip("oracle/_ref/pbrt_ref_flatcheck not built here (needs /root/reference)")
    rng = np.random.default_rng(11)
    def soup(n, spread, size):
        c = rng.uniform(-spread, spread, (n, 1, 3)).astype(np.float32)
        return (c + rng.uniform(-size, size, (n, 3, 3)).astype(np.float32)).reshape(-1, 3)
    a = soup(3000, 4.0, 0.15)
    b = np.concatenate([soup(400, 1.0, 0.02), np.tile(soup(1, 0.5, 0.3), (40, 1))])   # 40 identical triangles: equal Morton codes down to the last bit
    def mesh(v):
        return 'Shape "trianglemesh" "integer indices" [%s] "point P" [%s]\n' % (" ".join(str(i) for i in range(len(v))), " ".join("%.9g" % x for x in v.reshape(-1)))
    text = ('LookAt 0 0 -12 0 0 0 0 1 0\nCamera "perspective"\nSampler "sobol" "integer pixelsamples" [1]\nIntegrator "path"\n'
            'Film "image" "integer xresolution" [8] "integer yresolution" [8] "string filename" "o.pfm"\n'
            'Accelerator "bvh" "string splitmethod" "%s" "integer maxnodeprims" [%d]\nWorldBegin\nLightSource "point" "point from" [0 0 -10]\n%s%sWorldEnd\n'
            % (method, maxprims, mesh(a), mesh(b)))
    scene = tmp_path / "s.pbrt"
    scene.write_text(text)
    dump = str(tmp_path / "bvh.bin")
    env = dict(os.environ, PBRT_AMD_BACKEND_LIB=ORACLE, PBRT_AMD_BVH_DUMP=dump, PBRT_AMD_BVH_DUMP_ONLY="1")
    r = subprocess.run([STUB, "--quiet", "--nthreads", "1", "--outfile", str(tmp_path / "o.pfm"), str(scene)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and os.path.exists(dump), r.stderr[-800:]
    blob = open(dump, "rb").read()
    n_nodes, n_tris = np.frombuffer(blob[:8], np.uint32)
    ref_nodes = blob[8:8 + 32 * n_nodes]
    ref_tris = np.frombuffer(blob[8 + 32 * n_nodes:], np.float32).reshape(n_tris, 9)
    sc = pa.Scene(str(scene), strict=True)
    h_nodes, nodes, tris = _host_bvh(sc)
    assert (h_nodes, len(tris)) == (n_nodes, n_tris) and n_tris == 3000 + 440
    dt = np.dtype([("bmin", "f4", 3), ("bmax", "f4", 3), ("offset", "i4"), ("n", "u2"), ("axis", "u1"), ("pad", "u1")])
    R, H = np.frombuffer(ref_nodes, dt), np.frombuffer(nodes, dt)
    # the DFS layout is the same array position by position (bvh.cpp:640-658); everything but a leaf's primitivesOffset must be equal byte for byte.  The offsets
    # themselves are handed out in construction order, which the reference leaves to the compiler (`InitInterior(dim, recursiveBuild(..), recursiveBuild(..))`:
    # g++ evaluates the second argument first, so its ordered list fills from the right) -- what is pinned is what a leaf HOLDS, in order.
    for f in ("bmin", "bmax", "n", "axis"):
        assert np.ascontiguousarray(R[f]).tobytes() == np.ascontiguousarray(H[f]).tobytes(), f
    interior = R["n"] == 0
    assert np.array_equal(R["offset"][interior], H["offset"][interior])   # secondChildOffset
    leaves = np.flatnonzero(~interior)
    seen = 0
    for i in leaves:
        a, b, n = int(R["offset"][i]), int(H["offset"][i]), int(R["n"][i])
        assert np.array_equal(ref_tris[a:a + n].view(np.uint32), tris[b:b + n].view(np.uint32)), (i, a, b, n)
        seen += n
    assert seen == n_tris
    if method == "hlbvh":
        assert R["n"].max() > 1 or maxprims == 1   # the treelets did produce multi-primitive leaves (40 coinciding triangles run out of Morton bits)
