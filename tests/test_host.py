"""CPU: host-side logic of the drop-in (parser, pbrt API state machine, BVH build, flattening, film/image IO)."""
import os, sys
import subprocess

import numpy as np
import pytest

import oracle_lib as ol

pa = ol.pa
ROOT = ol.ROOT

MIN = ('Film "image" "integer xresolution" [32] "integer yresolution" [16] "string filename" "x.pfm"\n'
       'Sampler "sobol" "integer pixelsamples" [3]\n')


def test_scene_info_and_sampler_rounding(built):
    sc = pa.Scene(text=MIN + 'WorldBegin\nShape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]\nWorldEnd\n')
    assert (sc.width, sc.height) == (32, 16)
    assert sc.info["spp"] == 4                      # SobolSampler rounds up to a power of two (sobol.h:52)
    assert sc.info["sobol_resolution"] == 32 and sc.info["sobol_log2_resolution"] == 5
    assert sc.info["n_tris"] == 1 and sc.info["n_bvh_nodes"] == 1 and sc.info["max_depth"] == 5


def test_parser_features(built, tmp_path):
    """comments, bracketless single values, Include, named materials, attribute / transform stacks, instancing"""
    (tmp_path / "inc.pbrt").write_text('Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0] # a comment\n')
    (tmp_path / "m.pbrt").write_text(MIN + '''
# options
LookAt 0 0 -5  0 0 0  0 1 0
Camera "perspective" "float fov" 30
WorldBegin
MakeNamedMaterial "red" "string type" "matte" "rgb Kd" [.8 .1 .1]
AttributeBegin
  NamedMaterial "red"
  Translate 1 0 0
  Include "inc.pbrt"
AttributeEnd
TransformBegin
  Scale 2 2 2
  Include "inc.pbrt"
TransformEnd
ObjectBegin "o"
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [0 0 0 1 0 0 1 1 0 0 1 0]
ObjectEnd
ObjectInstance "o"
Translate 0 3 0
ObjectInstance "o"
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [1 1 1]
  Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 5 0 1 5 0 0 5 1]
AttributeEnd
LightSource "point" "point from" [0 2 0] "rgb I" [3 3 3]
WorldEnd
''')
    sc = pa.Scene(str(tmp_path / "m.pbrt"))
    assert sc.info["n_tris"] == 1 + 1 + 2 + 2 + 1
    assert sc.info["n_lights"] == 2            # one emissive triangle + the point light, in file order
    assert sc.info["n_materials"] == 2         # default matte and "red" (identical BxDF lists are merged)


def test_bvh_invariants(built):
    """every triangle in exactly one leaf; children inside their parent's box; DFS layout (bvh.cpp:640-658)"""
    import ctypes as C
    sc = pa.Scene(os.path.join(ROOT, "scenes", "materials.pbrt"))

    class Node(C.Structure):
        _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("offset", C.c_int32), ("n_prims", C.c_uint16), ("axis", C.c_uint8), ("pad", C.c_uint8)]
    assert C.sizeof(Node) == 32
    # mi_scene_desc prefix: u32 abi, u32 n_verts, P, N, UV, u32 n_tris(+pad), tri_indices, tri_mesh, tri_light, u32 n_meshes(+pad), meshes, u32 n_nodes(+pad), nodes
    raw = (C.c_uint64 * 12).from_address(sc.desc)   # 8-byte words: [abi|n_verts] P N UV [n_tris] idx mesh light [n_meshes] meshes [n_nodes] nodes
    n_nodes = raw[10] & 0xffffffff
    nodes = (Node * n_nodes).from_address(raw[11])
    assert n_nodes == sc.info["n_bvh_nodes"]
    seen = np.zeros(sc.info["n_tris"], dtype=np.int32)
    stack = [0]
    while stack:
        i = stack.pop()
        nd = nodes[i]
        if nd.n_prims > 0:
            seen[nd.offset:nd.offset + nd.n_prims] += 1
        else:
            for c in (i + 1, nd.offset):
                ch = nodes[c]
                assert all(ch.bmin[a] >= nd.bmin[a] and ch.bmax[a] <= nd.bmax[a] for a in range(3))
                stack.append(c)
    assert np.all(seen == 1)


def test_film_pfm_round_trip_and_xyz_semantics(built, tmp_path):
    sc = pa.Scene(text=MIN + "WorldBegin\nWorldEnd\n")
    rng = np.random.default_rng(0)
    rgbw = rng.random((sc.height, sc.width, 4)).astype(np.float32)
    rgbw[..., 3] = 4
    img = sc.film_image(rgbw)
    # Film::MergeFilmTile + WriteImage: RGB -> XYZ -> RGB in fp32, / weight (film.cpp:117-130,168-210)
    c = rgbw[..., :3]
    xyz = np.stack([np.float32(0.412453) * c[..., 0] + np.float32(0.357580) * c[..., 1] + np.float32(0.180423) * c[..., 2],
                    np.float32(0.212671) * c[..., 0] + np.float32(0.715160) * c[..., 1] + np.float32(0.072169) * c[..., 2],
                    np.float32(0.019334) * c[..., 0] + np.float32(0.119193) * c[..., 1] + np.float32(0.950227) * c[..., 2]], -1)
    rgb = np.stack([np.float32(3.240479) * xyz[..., 0] - np.float32(1.537150) * xyz[..., 1] - np.float32(0.498535) * xyz[..., 2],
                    np.float32(-0.969256) * xyz[..., 0] + np.float32(1.875991) * xyz[..., 1] + np.float32(0.041556) * xyz[..., 2],
                    np.float32(0.055648) * xyz[..., 0] - np.float32(0.204043) * xyz[..., 1] + np.float32(1.057311) * xyz[..., 2]], -1)
    want = np.maximum(np.float32(0), rgb * (np.float32(1) / np.float32(4)))
    assert np.allclose(img, want, rtol=2e-7, atol=1e-7)
    out = str(tmp_path / "o.pfm")
    sc.write_image(rgbw, out)
    back = pa.read_pfm(out)
    assert np.array_equal(back.view(np.uint32), img.view(np.uint32))     # PFM is lossless (tests/imageio.cpp:41-48)
    exr = str(tmp_path / "o.exr")
    sc.write_image(rgbw, exr)
    head = open(exr, "rb").read(8)
    assert head[:4] == bytes([0x76, 0x2F, 0x31, 0x01]) and head[4] == 2   # OpenEXR magic, version 2


def test_cli_reports_missing_gpu_loudly(built, tmp_path):
    """No CPU fallback: without a GPU the CLI must say so and write no image."""
    exe = os.path.join(ROOT, "pbrt-v3-distributed_amd", "bin", "pbrt_amd")
    out = tmp_path / "x.pfm"
    f = tmp_path / "s.pbrt"
    f.write_text(MIN + "WorldBegin\nWorldEnd\n")
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, "--quiet", "--outfile", str(out), str(f)], capture_output=True, text=True)
    assert "Error" in r.stderr and not out.exists()
    assert r.returncode != 0   # a render that did not happen is a failed run (never a silent success)


def test_bvh4_collapse_invariants(built, tmp_path):
    """The BVH2 -> BVH4 collapse of mi_scene_upload, checked on the host (mi_bvh4_validate, no GPU): every primitive in exactly one leaf
    reference of 1..16 primitives, child boxes = the reference nodes' boxes and nested, depth / stack bound consistent -- on the parity
    scenes, on degenerate ones (single leaf, empty world, spheres, instances) and with reference leaves far beyond 16 primitives
    (maxnodeprims 200 -> chained leaf nodes), under every split method."""
    import subprocess, sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import edge_scenes as es
    for f in ("cornell.pbrt", "materials.pbrt"):
        sc = pa.Scene(os.path.join(ROOT, "scenes", f))
        st = pa.bvh4_validate(sc)
        assert st["prims"] == sc.info["n_tris"] and st["nodes"] >= 1
    for n in ("onetri", "empty", "spheres", "instances"):
        sc = pa.Scene(text=es.scene(n))
        assert pa.bvh4_validate(sc)["prims"] == sc.info["n_tris"]
    out = str(tmp_path / "sm.pbrt")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_scenes.py"), "sanmiguel", "--tris", "60000", "--res", "64", "36", "--spp", "1", "--out", out], stdout=subprocess.DEVNULL)
    text = open(out).read()
    for acc in ('', 'Accelerator "bvh" "integer maxnodeprims" [200]\n', 'Accelerator "bvh" "string splitmethod" "middle" "integer maxnodeprims" [40]\n',
                'Accelerator "bvh" "string splitmethod" "equal" "integer maxnodeprims" [1]\n'):
        t = text.replace("WorldBegin", acc + "WorldBegin", 1).replace("geo/", os.path.join(str(tmp_path), "geo") + "/")
        sc = pa.Scene(text=t) if acc else pa.Scene(out)
        st = pa.bvh4_validate(sc)
        assert st["prims"] == sc.info["n_tris"] and st["stack_need"] == 3 * (st["depth"] + 1) + 1, (acc, st)


@pytest.mark.parametrize("tree", ["own", "reference"])
def test_own_topology_over_the_reference_leaves(built, tree, tmp_path, monkeypatch):
    """Round 5: single-level scenes traverse the library's own topology over the reference's leaves (csrc/pt_treebuild.h); PBRT_AMD_TREE=reference
    keeps the tree as handed over.  Host only: under either, the collapsed tree passes its structural checks (every primitive in exactly one leaf,
    boxes nested) with the SAME leaf references -- the leaves are the contract -- and the per-ray state machine of the quantised traversal gives the
    hits of the oracle's BVH2 traversal bit for bit (closest and any hit)."""
    import subprocess, sys
    monkeypatch.setenv("PBRT_AMD_TREE", tree)
    out = str(tmp_path / "sm.pbrt")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_scenes.py"), "sanmiguel", "--tris", "150000", "--res", "96", "54", "--spp", "1", "--out", out], stdout=subprocess.DEVNULL)
    for sc in (pa.Scene(os.path.join(ROOT, "scenes", "materials.pbrt")), pa.Scene(out)):
        st = pa.bvh4_validate(sc)
        assert st["prims"] == sc.info["n_tris"] and st["own_topology"] == (tree == "own")
        monkeypatch.setenv("PBRT_AMD_TREE", "reference")
        st_ref = pa.bvh4_validate(sc)
        monkeypatch.setenv("PBRT_AMD_TREE", tree)
        assert st["leaf_refs"] == st_ref["leaf_refs"]   # same leaves, however the interior is arranged
        rays = _study_rays(sc, 20000, 3)
        ref, _ = ol.intersect(sc, rays)
        h, _ = pa.bvh4q_validate(sc, rays)
        assert np.array_equal(h["prim"], ref["prim"])
        for k in ("t", "b1", "b2"):
            assert np.array_equal(h[k].view(np.uint32), ref[k].view(np.uint32)), k
        occ, _ = ol.intersect_p(sc, rays)
        h2, _ = pa.bvh4q_validate(sc, rays, any_hit=True)
        assert np.array_equal((h2["prim"] >= 0).astype(np.uint8), occ)


@pytest.mark.parametrize("tree", ["own", "reference"])
def test_coplanar_duplicated_triangles_resolve_as_in_the_reference(built, tree, monkeypatch):
    """Ties at exactly equal t: a wall of 64 quads, every quad present TWICE as separate shapes in shuffled order -- each ray
    that hits the wall hits two coincident triangles at the same t, bit for bit.  The reference keeps the one it tests last (`t > ray.tMax` rejects,
    triangle.cpp:258-261); the per-ray state machine of the quantised traversal must report the same primitive under either topology."""
    import random
    monkeypatch.setenv("PBRT_AMD_TREE", tree)
    random.seed(1)
    quad = lambda x0, y0: ('Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [%g %g 1 %g %g 1 %g %g 1 %g %g 1]\n' % (x0, y0, x0 + .5, y0, x0 + .5, y0 + .5, x0, y0 + .5))
    order = [(-2 + .5 * i, -2 + .5 * j) for i in range(8) for j in range(8)] * 2
    random.shuffle(order)
    text = ('LookAt 0 0 -5 0 0 0 0 1 0\nCamera "perspective" "float fov" [40]\nSampler "sobol" "integer pixelsamples" [1]\n'
            'Film "image" "integer xresolution" [64] "integer yresolution" [64] "string filename" "t.pfm"\nWorldBegin\nMaterial "matte"\n' + "".join(quad(*q) for q in order) +
            'AttributeBegin\nAreaLightSource "diffuse" "rgb L" [5 5 5]\nShape "trianglemesh" "integer indices" [0 1 2] "point P" [-1 3 0 1 3 0 0 3 1]\nAttributeEnd\nWorldEnd\n')
    sc = pa.Scene(text=text)
    rng = np.random.default_rng(3)
    n = 20000
    rays = np.zeros(n, dtype=pa.RAY_DTYPE)
    rays["o"] = np.stack([rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), np.full(n, -5.0)], 1).astype(np.float32)
    rays["d"] = np.stack([rng.uniform(-.05, .05, n), rng.uniform(-.05, .05, n), np.ones(n)], 1).astype(np.float32)
    rays["tmax"] = np.inf
    ref, _ = ol.intersect(sc, rays)
    assert (ref["prim"] >= 0).sum() > n // 2
    h, _ = pa.bvh4q_validate(sc, rays)
    assert np.array_equal(h["prim"], ref["prim"])
    assert np.array_equal(h["t"].view(np.uint32), ref["t"].view(np.uint32))


@pytest.mark.parametrize("name", ["instances", "instances2"])
def test_bvh4_collapse_two_level(built, name, monkeypatch):
    """Two-level scenes (PBRT_AMD_INSTANCING=1): mi_scene_upload collapses the top-level BVH2 and every instanced object's own BVH2 into
    one BVH4 node array (object leaves carry GLOBAL primitive offsets).  Same invariants over all the trees together: every primitive --
    top-level ones, the TransformedPrimitive records, the objects' own -- in exactly one leaf reference."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import edge_scenes as es
    monkeypatch.setenv("PBRT_AMD_INSTANCING", "1")
    sc = pa.Scene(text=es.scene(name))
    st = pa.bvh4_validate(sc)
    assert st["objects"] >= 1 and st["prims"] == sc.info["n_tris"]
    assert st["stack_need"] > 3 * (st["depth"] + 1) + 1    # room for the rest of a leaf, the sentinel and the object's tree


# ---------------------------------------------------------------- image readers (core/imageio.cpp:216-290)
def _png_bytes(samples, w, h, ctype, depth, palette=None, filters=None):
    """encode rows of integer samples (h x w*channels) as a PNG with the given per-row filter types (PNG spec s.9)"""
    import struct, zlib
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    rows = []
    for y in range(h):
        r = [int(v) for v in samples[y]]
        if depth == 16:
            b = b"".join(struct.pack(">H", v) for v in r)
        elif depth == 8:
            b = bytes(r)
        else:
            bits = "".join(format(v, "0%db" % depth) for v in r)
            bits += "0" * (-len(bits) % 8)
            b = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
        rows.append(b)
    bpp = max(1, ch * depth // 8)
    out = b""
    prev = bytes(len(rows[0]))
    for y, row in enumerate(rows):
        ft = filters[y % len(filters)] if filters else 0
        enc = bytearray()
        for i, x in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0: p = 0
            elif ft == 1: p = a
            elif ft == 2: p = b
            elif ft == 3: p = (a + b) >> 1
            else:
                pp = a + b - c
                pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            enc.append((x - p) & 255)
        out += bytes([ft]) + bytes(enc)
        prev = row

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if palette is not None:
        data += chunk(b"PLTE", bytes(palette))
    half = len(out) // 2
    comp = zlib.compress(out, 6)
    return data + chunk(b"IDAT", comp[:half]) + chunk(b"tEXt", b"k\0v") + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b"")   # split IDAT + ancillary chunk


@pytest.mark.parametrize("ctype,depth", [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 4), (3, 8), (4, 8), (4, 16), (6, 8), (6, 16)])
def test_png_reader_all_layouts(built, tmp_path, ctype, depth):
    """every colour type / bit depth of non-interlaced PNG, all five scanline filters, split IDAT: the host reader must give what
    lodepng_decode24 gives the reference (8-bit RGB: 16-bit samples keep their high byte, sub-byte greys scale to 255, alpha dropped) / 255"""
    rng = np.random.default_rng(ctype * 100 + depth)
    w, h = 13, 9
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    s = rng.integers(0, 1 << depth, (h, w * ch))
    pal = [int(v) for v in rng.integers(0, 256, 3 * (1 << depth))] if ctype == 3 else None
    f = tmp_path / "t.png"
    f.write_bytes(_png_bytes(s, w, h, ctype, depth, pal, filters=[0, 1, 2, 3, 4]))
    img = pa.read_image(str(f))
    px = s.reshape(h, w, ch)
    to8 = (lambda v: v >> 8) if depth == 16 else ((lambda v: v) if depth == 8 else (lambda v: (v * 255) // ((1 << depth) - 1)))
    if ctype == 3:
        want = np.array(pal, dtype=np.int64).reshape(-1, 3)[px[..., 0]]
    elif ctype in (0, 4):
        want = np.repeat(to8(px[..., :1]), 3, axis=2)
    else:
        want = to8(px[..., :3])
    assert img.shape == (h, w, 3)
    assert np.array_equal(img, (want.astype(np.float32) / np.float32(255)))


def test_tga_reader_variants(built, tmp_path):
    """TGA: 24 / 32-bit true colour, uncompressed and RLE, 8-bit mono, colour-mapped; all four origin conventions"""
    import struct
    rng = np.random.default_rng(5)
    w, h = 7, 5
    rgb = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)        # row 0 = top, what ReadImage must return (/255)

    def hdr(typ, bits, desc, cmap=(0, 0, 0, 0)):
        return struct.pack("<BBBHHBHHHHBB", 0, cmap[0], typ, cmap[1], cmap[2], cmap[3], 0, 0, w, h, bits, desc)

    def rle(pixels, pb):   # alternate raw packets and run packets
        out, i = b"", 0
        n = len(pixels) // pb
        while i < n:
            k = min(3, n - i)
            if (i // 3) % 2 == 0:
                out += bytes([k - 1]) + pixels[i * pb:(i + k) * pb]
                i += k
            else:
                out += bytes([0x80]) + pixels[i * pb:(i + 1) * pb]
                i += 1
        return out
    for desc in (0x00, 0x10, 0x20, 0x30):
        src = rgb[:, ::-1] if desc & 0x10 else rgb
        src = src if desc & 0x20 else src[::-1]
        bgr = np.ascontiguousarray(src[..., ::-1])
        bgra = np.concatenate([bgr, np.full((h, w, 1), 200, np.uint8)], -1)
        cases = {"t24": hdr(2, 24, desc) + bgr.tobytes(), "t32": hdr(2, 32, desc | 8) + bgra.tobytes(),
                 "r24": hdr(10, 24, desc) + rle(bgr.tobytes(), 3), "r32": hdr(10, 32, desc | 8) + rle(bgra.tobytes(), 4)}
        for name, data in cases.items():
            f = tmp_path / ("%s_%02x.tga" % (name, desc))
            f.write_bytes(data)
            assert np.array_equal(pa.read_image(str(f)), rgb.astype(np.float32) / np.float32(255)), (name, desc)
    mono = rng.integers(0, 256, (h, w)).astype(np.uint8)
    f = tmp_path / "m.tga"; f.write_bytes(hdr(3, 8, 0x20) + mono.tobytes())
    assert np.array_equal(pa.read_image(str(f)), np.repeat(mono[..., None], 3, 2).astype(np.float32) / np.float32(255))
    pal = rng.integers(0, 256, (16, 3)).astype(np.uint8)            # BGR entries
    idx = rng.integers(0, 16, (h, w)).astype(np.uint8)
    f = tmp_path / "c.tga"; f.write_bytes(hdr(1, 8, 0x20, cmap=(1, 0, 16, 24)) + pal.tobytes() + idx.tobytes())
    assert np.array_equal(pa.read_image(str(f)), pal[idx][..., ::-1].astype(np.float32) / np.float32(255))


def _study_rays(sc, n, seed):
    """camera rays and incoherent secondary rays leaving the first hit points in random directions"""
    rng = np.random.default_rng(seed)
    px = np.stack([rng.integers(0, sc.width, n // 2), rng.integers(0, sc.height, n // 2)], 1).astype(np.int32)
    cam, _ = ol.camera_rays(sc, px, np.zeros(n // 2, np.int32))
    hits, _ = ol.intersect(sc, cam)
    sec = np.zeros(n // 2, dtype=pa.RAY_DTYPE)
    t = np.where(hits["prim"] >= 0, hits["t"], 1.0)[:, None]
    sec["o"] = cam["o"] + cam["d"] * t * 0.999
    d = rng.normal(size=(n // 2, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
    d[::7, 1] = 0                      # some axis-parallel components (the 1/d = inf case of the slab test)
    d[::11, 0] = 0
    sec["d"] = d; sec["tmax"] = np.inf
    return np.concatenate([cam, sec])


def test_quantised_bvh4_gives_the_reference_hits(built, tmp_path, monkeypatch):
    """The default traversal layout (csrc/pt_bvh4q.h, mi_bvh4q_validate; host only): the reference's BVH2 collapsed to 4-wide nodes with
    16-bit quantised child boxes on one grid and the folded, slack-padded box test must (a) pass its structural checks (every primitive in
    one leaf, every quantised box a superset of its reference box in exact arithmetic) and (b) give, through the per-ray state machine
    the kernel runs, exactly the hits of the oracle's BVH2 traversal -- primitive, t and barycentrics bit for bit, closest and any
    hit -- on camera rays and incoherent secondary rays, including rays with zero direction components."""
    import subprocess, sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import edge_scenes as es
    monkeypatch.setenv("PBRT_AMD_INSTANCING", "0")   # the quantised layout covers single-level scenes: the instanced test scene in its flattened form
    out = str(tmp_path / "sm.pbrt")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_scenes.py"), "sanmiguel", "--tris", "150000", "--res", "96", "54", "--spp", "1", "--out", out], stdout=subprocess.DEVNULL)
    scenes = [pa.Scene(os.path.join(ROOT, "scenes", "cornell.pbrt")), pa.Scene(os.path.join(ROOT, "scenes", "materials.pbrt")),
              pa.Scene(text=es.scene("onetri")), pa.Scene(text=es.scene("empty")), pa.Scene(text=es.scene("instances")), pa.Scene(out)]
    for sc in scenes:
        rays = _study_rays(sc, 20000 if sc.info["n_tris"] > 1000 else 4000, 3)
        ref, cnt = ol.intersect(sc, rays)
        if sc.info.get("n_instances", 0):
            continue
        h, st = pa.bvh4q_validate(sc, rays)
        assert st["prims"] == sc.info["n_tris"]
        assert np.array_equal(h["prim"], ref["prim"])
        for k in ("t", "b1", "b2"):
            assert np.array_equal(h[k].view(np.uint32), ref[k].view(np.uint32)), k
        occ, _ = ol.intersect_p(sc, rays)
        h2, _ = pa.bvh4q_validate(sc, rays, any_hit=True)
        assert np.array_equal((h2["prim"] >= 0).astype(np.uint8), occ)


def test_png_and_tga_output_match_the_reference_writer(built, tmp_path):
    """WriteImage for the 8-bit formats (imageio.cpp:90-117): gamma-encoded bytes as PNG / TGA.  The files written by this host must decode
    (through the host's own readers) to the bytes the formula gives, and -- when the reference binary is built here -- to the very bytes
    the reference writes for the same scene."""
    text = open(os.path.join(ROOT, "scenes", "cornell.pbrt")).read()
    text = text.replace('[400] "integer yresolution" [400]', '[48] "integer yresolution" [40]').replace('"integer pixelsamples" [8]', '"integer pixelsamples" [2]')
    sc = pa.Scene(text=text)
    rgbw, _, _ = ol.render(sc, nthreads=2)
    img = sc.film_image(rgbw)
    g = np.where(img <= np.float32(0.0031308), np.float32(12.92) * img, np.float32(1.055) * np.power(img, np.float32(1 / 2.4), dtype=np.float32) - np.float32(0.055))
    want = np.clip(np.float32(255) * g + np.float32(0.5), 0, 255).astype(np.uint8)
    for ext in ("png", "tga"):
        out = str(tmp_path / ("o." + ext))
        sc.write_image(rgbw, out)
        got = np.round(pa.read_image(out) * 255).astype(np.int32)
        assert got.shape == want.shape
        assert np.abs(got - want.astype(np.int32)).max() <= 1 and np.mean(got != want) < 2e-3   # powf vs numpy's float power at a rounding boundary
        if ol.have_ref():
            f = tmp_path / "s.pbrt"; f.write_text(text)
            ref_out = str(tmp_path / ("r." + ext))
            subprocess.check_call([ol.PBRT_REF, "--quiet", "--nthreads", "1", "--outfile", ref_out, str(f)])
            ref = np.round(pa.read_image(ref_out) * 255).astype(np.int32)
            assert np.abs(got - ref).max() <= 1 and np.mean(got != ref) < 2e-3   # the oracle's film differs from the reference's by <= 1 ulp


def _exr_bytes(chans, w, h, compression, line_order=0, y_origin=3, x_origin=2):
    """assemble a single-part scan-line OpenEXR file: chans = {name: (type, array[h, w])}, type 1 = HALF (float16), 2 = FLOAT"""
    import struct, zlib
    names = sorted(chans)

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(data)) + data
    chl = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", chans[n][0], 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    hdr = struct.pack("<II", 20000630, 2)
    hdr += attr("channels", "chlist", chl) + attr("compression", "compression", bytes([compression]))
    box = struct.pack("<iiii", x_origin, y_origin, x_origin + w - 1, y_origin + h - 1)
    hdr += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", bytes([line_order]))
    hdr += attr("pixelAspectRatio", "float", struct.pack("<f", 1)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0))
    hdr += attr("screenWindowWidth", "float", struct.pack("<f", 1)) + b"\0"
    lpb = 16 if compression == 3 else 1
    blocks = []
    for y0 in range(0, h, lpb):
        raw = b""
        for y in range(y0, min(h, y0 + lpb)):
            for n in names:
                t, a = chans[n]
                raw += a[y].astype("<f2" if t == 1 else "<f4").tobytes()
        data = raw
        if compression in (1, 2, 3):
            b = np.frombuffer(raw, np.uint8)
            t = np.concatenate([b[0::2], b[1::2]]).astype(np.int32)            # even / odd byte split
            d = t.copy(); d[1:] = (t[1:] - t[:-1] + 128 + 256) % 256           # byte predictor
            pre = d.astype(np.uint8).tobytes()
            if compression == 1:                                               # RLE with signed counts (runs of >= 3, else literals)
                out, i = bytearray(), 0
                while i < len(pre):
                    j = i
                    while j + 1 < len(pre) and pre[j + 1] == pre[i] and j - i < 126: j += 1
                    if j - i >= 2:
                        out += bytes([j - i]) + pre[i:i + 1]; i = j + 1
                    else:
                        k = i
                        while k < len(pre) and k - i < 127 and not (k + 2 < len(pre) and pre[k] == pre[k + 1] == pre[k + 2]): k += 1
                        out += struct.pack("b", -(k - i)) + pre[i:k]; i = k
                comp = bytes(out)
            else:
                comp = zlib.compress(pre, 6)
            data = comp if len(comp) < len(raw) else raw                       # stored raw when compression does not pay
        blocks.append(struct.pack("<ii", y_origin + y0, len(data)) + data)
    order = list(range(len(blocks)))
    if line_order == 1: order.reverse()                                        # decreasing y: blocks stored bottom-up
    table_pos = len(hdr)
    offs, pos = [0] * len(blocks), table_pos + 8 * len(blocks)
    body = b""
    for i in order:
        offs[i] = pos; body += blocks[i]; pos += len(blocks[i])
    return hdr + b"".join(struct.pack("<Q", o) for o in offs) + body


@pytest.mark.parametrize("compression", [0, 1, 2, 3])
def test_exr_reader(built, tmp_path, compression):
    """scan-line OpenEXR input (radiance maps / textures of the public scenes): NONE / RLE / ZIPS / ZIP, half and float channels,
    RGBA and luminance-only files, both line orders, a data window that does not start at (0,0).  No OpenEXR library in this image: checked
    against files assembled here from the format specification and against this host's own EXR writer (the uncompressed layout is pinned on a file
    OpenEXR wrote: test_exr_reader_on_a_file_written_by_openexr)."""
    rng = np.random.default_rng(compression)
    w, h = 21, 37
    R, G, B = [(rng.random((h, w)) * 4).astype(np.float32) for _ in range(3)]
    R[5:9] = 0.5                                                               # long runs for the RLE path
    f = tmp_path / "a.exr"
    f.write_bytes(_exr_bytes({"R": (1, R), "G": (1, G), "B": (2, B), "A": (1, np.ones((h, w), np.float32))}, w, h, compression))
    img = pa.read_image(str(f))
    want = np.stack([R.astype(np.float16).astype(np.float32), G.astype(np.float16).astype(np.float32), B], -1)
    assert img.shape == (h, w, 3) and np.array_equal(img, want)
    f = tmp_path / "y.exr"
    f.write_bytes(_exr_bytes({"Y": (2, R)}, w, h, compression, line_order=1))
    assert np.array_equal(pa.read_image(str(f)), np.repeat(R[..., None], 3, 2))
    if compression == 0:   # and the file this host writes itself (uncompressed half RGBA)
        sc = pa.Scene(text=MIN + "WorldBegin\nWorldEnd\n")
        rgbw = rng.random((sc.height, sc.width, 4)).astype(np.float32)
        rgbw[..., 3] = 1
        out = str(tmp_path / "o.exr")
        sc.write_image(rgbw, out)
        back = pa.read_image(out)
        assert np.array_equal(back, sc.film_image(rgbw).astype(np.float16).astype(np.float32))


@pytest.mark.parametrize("case", ["piz_smooth", "piz_wide_noise", "piz_no_runs", "piz_tiled", "pxr24", "pxr24_tiled", "none_tiled_mip"])
def test_exr_reader_piz_pxr24_and_tiles(built, tmp_path, case):
    """Round 6 (VERDICT r5 'missing' 5): PIZ -- what most published pbrt-v3 radiance maps are stored with -- PXR24 and tiled files.  UNPINNED against OpenEXR
    (no library, no such file in this image): the files come from tests/exr_codec.py, a restatement of the format's ENCODING side (forward LUT, wenc14 / wenc16, packed
    code lengths with zero runs, the run-length symbol), the reader restates the DECODING side; each case must come back bit for bit.
    piz_smooth: few distinct values (14-bit wavelet), constant areas (run-length symbol), odd sizes (the wavelet's odd rows / columns), half + float + uint channels, a
    last block shorter than 32 lines.  piz_wide_noise: > 16384 distinct values in a block (16-bit wavelet, codes longer than the decoder's 14-bit table)."""
    import exr_codec as ec
    rng = np.random.default_rng(7)
    info = {}
    if case == "piz_wide_noise":
        w, h = 331, 35
        R = np.exp(rng.uniform(-8, 8, (h, w))).astype(np.float32)                # many binades: > 2^14 distinct half values in one 32-line block
        G = (-np.exp(rng.uniform(-8, 8, (h, w)))).astype(np.float32)
        B = np.exp(rng.uniform(-3, 10, (h, w))).astype(np.float32)
        f = ec.exr_bytes({"R": (1, R), "G": (1, G), "B": (1, B)}, w, h, 4, info=info)
        assert info["max_value"] >= 1 << 14 and info["longest_code"] > 14, info
        want = np.stack([R, G, B], -1).astype(np.float16).astype(np.float32)
    else:
        w, h = (45, 71) if "tiled" in case else (37, 45)
        yy, xx = np.mgrid[0:h, 0:w]
        R = (np.round((np.sin(xx * 0.3) + np.cos(yy * 0.2)) * 8) / 8 + 3).astype(np.float32)
        R[10:30, 5:30] = 0.75                                                   # long runs
        G = np.round(rng.random((h, w)) * 64).astype(np.float32) / 4            # 65 distinct halves
        B = (np.round(rng.random((h, w)) * 2048) / 256).astype(np.float32)      # floats whose low mantissa bits are zero (PXR24 keeps 24 bits)
        U = rng.integers(0, 70000, (h, w)).astype(np.float32)
        chans = {"R": (1, R), "G": (1, G), "B": (2, B), "A": (0, U)}
        comp = 4 if case.startswith("piz") else (5 if case.startswith("pxr24") else 0)
        tiles = (16, 24) if "tiled" in case else None
        f = ec.exr_bytes(chans, w, h, comp, tiles=tiles, piz_runs=case != "piz_no_runs", info=info, level_mode=1 if case == "none_tiled_mip" else 0)
        if comp == 4: assert info["max_value"] < 1 << 14
        want = np.stack([R.astype(np.float16).astype(np.float32), G.astype(np.float16).astype(np.float32), B], -1)
    path = tmp_path / "a.exr"
    path.write_bytes(f)
    img = pa.read_image(str(path))
    assert img.shape == want.shape and np.array_equal(img, want)
    if case in ("piz_smooth", "pxr24"):   # damaged data is an error, never an image: cut the file, flip bytes in the payload
        cut = tmp_path / "cut.exr"
        cut.write_bytes(f[:len(f) - 40])
        with pytest.raises(Exception):
            pa.read_image(str(cut))
        bad = bytearray(f)
        for k in range(len(f) - 300, len(f) - 100, 7): bad[k] ^= 0x5a
        (tmp_path / "bad.exr").write_bytes(bytes(bad))
        try:
            got = pa.read_image(str(tmp_path / "bad.exr"))                      # (a flipped payload may still decode: then it must at least keep the shape)
            assert got.shape == want.shape
        except Exception:
            pass


def test_media_declarations_reach_the_scene_description(built):
    """MakeNamedMedium / MediumInterface / Integrator "volpath" (core/api.cpp:685-731,1093-1121,1496-1516): media in
    definition order, the MediumInterface of each GeometricPrimitive, the camera medium = the OUTSIDE medium of the graphics state at
    WorldEnd (api.cpp:793), measured-coefficient presets, and -- for Integrator "path", which never looks at ray.medium -- the same
    declarations carried along without changing what is rendered."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import edge_scenes as es
    fog = pa.Scene(text=es.scene("vol_fog"))
    assert (fog.info["integrator"], fog.info["n_media"], fog.info["camera_medium"]) == ("volpath", 1, 0)
    assert fog.info["n_medium_transitions"] >= 4          # every surface: vacuum inside, fog outside
    smoke = pa.Scene(text=es.scene("vol_smoke"))
    assert (smoke.info["n_media"], smoke.info["camera_medium"], smoke.info["n_medium_transitions"]) == (1, -1, 1)
    glass = pa.Scene(text=es.scene("vol_glass"))
    assert glass.info["n_media"] == 3 and glass.info["camera_medium"] == 2   # "haze" defined twice: the later definition is the one named afterwards
    as_path = pa.Scene(text=es.scene("vol_fog").replace('Integrator "volpath"', 'Integrator "path"'))
    assert (as_path.info["integrator"], as_path.info["n_media"]) == ("path", 1)
    plain = pa.Scene(os.path.join(ROOT, "scenes", "cornell.pbrt"))
    assert (plain.info["integrator"], plain.info["n_media"], plain.info["camera_medium"]) == ("path", 0, -1)


def test_blackbody_and_sampled_spectrum_parameters_vs_reference(built):
    """"blackbody" / "spectrum" parameter values -> RGB (host/spectrum.cpp) against the reference's own conversion
    (ParamSet::AddBlackbodySpectrum / AddSampledSpectrum -> RGBSpectrum::FromSampled, recorded by oracle/ref_build/ref_probe.cpp into
    tests/golden/spectra_vectors.npz): bit-exact, including unsorted samples and samples outside / inside the CIE range only."""
    recs = np.load(os.path.join(ROOT, "tests", "golden", "spectra_vectors.npz"))["spectra"]
    assert len(recs) == 27 + 240
    lights = []
    for r in recs:
        vals = " ".join("%.9g" % v for v in r["vals"][:r["n"]])
        lights.append('LightSource "point" "%s I" [%s]' % ("blackbody" if r["kind"] == 0 else "spectrum", vals))
    sc = pa.Scene(text=MIN + "WorldBegin\n" + "\n".join(lights) + '\nShape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]\nWorldEnd\n')
    assert sc.info["n_lights"] == len(recs)
    for i, r in enumerate(recs):
        _t, rgb = sc.light(i)
        assert rgb.tobytes() == r["rgb"].tobytes(), (i, int(r["kind"]), rgb, r["rgb"])


def test_spectrum_files(built, tmp_path):
    """SPD files ("spectrum" with string values, paramset.cpp:171-205): comments, exponents, unsorted samples, the reader's dropped last
    number when the file does not end in whitespace (floatfile.cpp:52-79), a missing file -> black with a warning; several files = several spectra."""
    (tmp_path / "a.spd").write_text("# c\n400 .2\n500 3.5e-1 # t\n700 .75\n600 .7\n")
    (tmp_path / "b.spd").write_text("400 1 500 2 600 3 700 9")           # the trailing 9 is never stored: 7 values, the odd one ignored -> 3 pairs
    (tmp_path / "c.spd").write_text("400 1 500 2 600 3\n")
    t = (MIN + 'WorldBegin\nLightSource "point" "spectrum I" "%s/a.spd"\nLightSource "point" "spectrum I" [400 .2 500 .35 600 .7 700 .75]\n'
         'LightSource "point" "spectrum I" "%s/b.spd"\nLightSource "point" "spectrum I" "%s/c.spd"\nLightSource "point" "spectrum I" "%s/missing.spd"\n'
         'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]\nWorldEnd\n') % ((tmp_path,) * 4)
    (tmp_path / "s.pbrt").write_text(t.replace(str(tmp_path) + "/", ""))   # relative names resolve against the scene file's directory
    sc = pa.Scene(str(tmp_path / "s.pbrt"))
    L = [sc.light(i)[1] for i in range(5)]
    assert L[0].tobytes() == L[1].tobytes() and L[0].min() > 0
    assert L[2].tobytes() == L[3].tobytes()
    assert not L[4].any()


@pytest.mark.parametrize("edit", [('Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]', 'Shape "disk" "float radius" [1]'),
                                  ('WorldBegin\n', 'WorldBegin\nLightSource "projection" "rgb I" [1 1 1] "string mapname" "x.png"\n'),
                                  ('WorldBegin\n', 'Camera "orthographic"\nWorldBegin\n'),
                                  ('WorldBegin\n', 'Integrator "bdpt"\nWorldBegin\n'),
                                  ('WorldBegin\n', 'Accelerator "kdtree"\nWorldBegin\n'),
                                  ('Shape "trianglemesh"', 'ActiveTransform EndTime\nTranslate 1 0 0\nActiveTransform All\nShape "trianglemesh"'),
                                  ('WorldBegin\n', 'ActiveTransform EndTime\nTranslate 0 0 1\nActiveTransform All\nCamera "perspective"\nWorldBegin\n')])
def test_scene_content_without_a_counterpart_is_refused_not_skipped(edit, tmp_path):
    """ADVICE r1 (plausible-but-wrong images): shapes / lights / cameras of the reference that this path does not carry used to be skipped
    with a warning, so the scene rendered without them.  Now the scene is refused: pbrt_amd_scene_load returns NULL and the command-line
    renderer exits non-zero without an image.  Names the reference does not know either keep its behaviour (warning, skipped)."""
    base = MIN + 'WorldBegin\nLightSource "point" "rgb I" [1 1 1]\nShape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]\nWorldEnd\n'
    assert edit[0] in base
    with pytest.raises(RuntimeError):
        pa.Scene(text=base.replace(edit[0], edit[1], 1))
    f = tmp_path / "r.pbrt"
    f.write_text(base.replace(edit[0], edit[1], 1))
    out = tmp_path / "r.pfm"
    r = subprocess.run([os.path.join(ROOT, "pbrt-v3-distributed_amd", "bin", "pbrt_amd"), "--quiet", "--outfile", str(out), str(f)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and not out.exists() and "unsupported" in r.stderr
    pa.Scene(text=base)                                                              # the count is per pbrtInit
    pa.Scene(text=base.replace('Shape "trianglemesh"', 'Shape "nosuchshape"\nShape "trianglemesh"'))   # unknown to the reference too: skipped


def test_beam_diffusion_table_matches_reference(built):
    """host/bssrdf.cpp (the table the Subsurface / KdSubsurface material constructors compute) against the reference's ComputeBeamDiffusionBSSRDF
    (core/bssrdf.cpp:113-160; tests/golden/bssrdf_tables.npz from ref_probe) for three (g, eta) pairs: albedo and radius samples, the 100 x 64
    profile, the effective albedo and the profile CDF -- 13 064 values each, bit for bit."""
    import ctypes as C
    L = pa.host_lib()
    L.pbrt_amd_bssrdf_table.argtypes = [C.c_float, C.c_float, C.c_void_p]
    T = np.load(os.path.join(ROOT, "tests", "golden", "bssrdf_tables.npz"))["tables"]
    assert len(T) == 3
    for t in T:
        out = np.zeros(100 + 64 + 6400 + 100 + 6400, np.float32)
        assert L.pbrt_amd_bssrdf_table(float(t["g"]), float(t["eta"]), out.ctypes.data) == 0
        ref = np.concatenate([t["rho_samples"], t["radius_samples"], t["profile"], t["rho_eff"], t["profile_cdf"]])
        assert out.tobytes() == ref.tobytes(), (float(t["g"]), float(t["eta"]))


def test_fast_samplers_option_renders_tile_serial_sampler_names_with_sobol(built, monkeypatch):
    """--fast-samplers / PBRT_AMD_FAST_SAMPLERS=1 (host/api.cpp): scenes that name random / stratified / 02sequence / lowdiscrepancy are rendered with
    sobol at the same sample count -- the user's choice of wavefront speed over the reference's pixel values.  Default: the tile-serial samplers."""
    import edge_scenes
    base = edge_scenes.scene("sampler_stratified")   # Sampler "stratified" xsamples x ysamples
    import re
    m = re.search(r'Sampler "stratified"[^\n]*', base)
    nx, ny = (int(v) for v in re.findall(r'"integer [xy]samples" \[(\d+)\]', m.group(0)))
    as_sobol = base.replace(m.group(0), 'Sampler "sobol" "integer pixelsamples" [%d]' % (nx * ny))
    want = ol.render(pa.Scene(text=as_sobol), nthreads=4)[0]
    default = ol.render(pa.Scene(text=base), nthreads=4)[0]
    monkeypatch.setenv("PBRT_AMD_FAST_SAMPLERS", "1")
    fast = ol.render(pa.Scene(text=base), nthreads=4)[0]
    assert np.array_equal(fast, want) and not np.array_equal(default, want)


@pytest.mark.parametrize("keep", [0.999, 0.6, 0.05])
def test_a_cut_off_ply_file_is_an_error_never_a_smaller_mesh(built, tmp_path, keep):
    """CreatePLYMesh on a file that ends before the vertices / faces its header declares (cut off, or caught while another process is still writing it): every
    read is checked against the end of the file, the shape is dropped with the reference's own error (plymesh.cpp:196-202: rply's ply_read fails), the scene's
    error count says so and a strict load raises -- round 5's reader ran past the buffer and built a mesh of whatever it found."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_scenes", os.path.join(ol.ROOT, "tools", "gen_scenes.py"))
    gs = importlib.util.module_from_spec(spec); spec.loader.exec_module(gs)
    rng = np.random.default_rng(3)
    v = rng.uniform(-1, 1, (3000, 3)); f = rng.integers(0, 3000, (5000, 3))
    ply = tmp_path / "m.ply"
    gs.write_ply(str(ply), v, f, normals=v, uvs=v[:, :2])
    text = ('LookAt 0 0 -5 0 0 0 0 1 0\nCamera "perspective"\nFilm "image" "integer xresolution" [16] "integer yresolution" [16]\nWorldBegin\n'
            'LightSource "point"\nShape "plymesh" "string filename" "%s"\nShape "sphere"\nWorldEnd\n' % ply)
    sc = ol.pa.Scene(text=text)
    assert sc.errors == 0 and sc.info["n_tris"] == 5001   # + the sphere
    size = ply.stat().st_size
    with open(ply, "r+b") as fh:
        fh.truncate(int(size * keep))
    sc = ol.pa.Scene(text=text)
    assert sc.errors >= 1 and sc.info["n_tris"] == 1   # the reference's behaviour: an Error, the shape dropped, the rest of the scene kept
    with pytest.raises(RuntimeError, match="strict mode"):
        ol.pa.Scene(text=text, strict=True)


def test_exr_reader_on_a_file_written_by_openexr(built):
    """The one pin of the EXR reader against the real library that this image allows: tests/golden/openexr_written_16x16_rgba_half.exr is a file WRITTEN BY OpenEXR (CPython's
    Lib/test/imghdrdata/python.exr, 16 x 16, channels A B G R as half, no compression, increasing y; copied as data) -- its header attributes, offset table and scan-line blocks are
    OpenEXR's own, not this repository's idea of them.  Decoded here independently with numpy (the format's uncompressed layout: per scan line the channels in alphabetical order, each
    a row of little-endian halfs) and compared with the host reader's result, bit for bit.  PIZ / PXR24 / tiled files: test_exr_reader_piz_pxr24_and_tiles (against a restatement of the encoding side: no writer of them anywhere in the image)."""
    import struct
    path = os.path.join(ROOT, "tests", "golden", "openexr_written_16x16_rgba_half.exr")
    d = open(path, "rb").read()
    assert d[:4] == bytes([0x76, 0x2F, 0x31, 0x01])
    i, attrs = 8, {}
    while d[i] != 0:
        j = d.index(b"\0", i); k = d.index(b"\0", j + 1)
        size = struct.unpack("<i", d[k + 1:k + 5])[0]
        attrs[d[i:j].decode()] = (d[j + 1:k].decode(), d[k + 5:k + 5 + size])
        i = k + 5 + size
    i += 1
    assert attrs["compression"][1] == b"\0" and attrs["lineOrder"][1] == b"\0"
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    chans, c = [], attrs["channels"][1]
    p = 0
    while c[p] != 0:
        q = c.index(b"\0", p)
        chans.append((c[p:q].decode(), struct.unpack("<i", c[q + 1:q + 5])[0]))
        p = q + 17
    assert [n for n, _ in chans] == ["A", "B", "G", "R"] and all(t == 1 for _, t in chans) and (w, h) == (16, 16)
    offsets = struct.unpack("<%dQ" % h, d[i:i + 8 * h])
    want = np.zeros((h, w, 3), np.float32)
    for y in range(h):
        yy, size = struct.unpack("<ii", d[offsets[y]:offsets[y] + 8])
        row = np.frombuffer(d[offsets[y] + 8:offsets[y] + 8 + size], "<f2").reshape(4, w).astype(np.float32)   # A, B, G, R
        want[yy - y0] = np.stack([row[3], row[2], row[1]], -1)
    got = pa.read_image(path)
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert want.max() > 0.5 and len(np.unique(want)) > 20   # a picture, not a constant
