#!/usr/bin/env python3
"""Generate tests/golden/ from the REAL reference (oracle/_ref/pbrt_ref + ref_probe, built from /root/reference by
oracle/ref_build/Makefile).  Run here (where /root/reference exists); the fixtures are committed because the reference
cannot travel to the GPU box.

  golden/ref_vectors.npz    known-answer vectors from the reference's own classes (Sobol, SobolSampler, Triangle::Intersect
                            on the unit-test ray constructions, Distribution1D::SampleDiscrete)
  golden/<scene>_<res>_<spp>spp.pfm   reference renders (float32, lossless) of the small parity scenes
"""
import os, subprocess, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(ROOT, "tests", "golden")


LIGHT_SAMPLES = np.dtype([("kind", "<i4"), ("two_sided", "<i4"), ("geom", "<f4", 12), ("L", "<f4", 3), ("p", "<f4", 3), ("n", "<f4", 3), ("u", "<f4", 2),
                          ("wi", "<f4", 3), ("pdf", "<f4"), ("Li", "<f4", 3), ("ray_o", "<f4", 3), ("ray_d", "<f4", 3), ("ray_tmax", "<f4"),
                          ("wi2", "<f4", 3), ("pdf_a", "<f4"), ("pdf_b", "<f4"), ("ok", "<i4")])
SCENE_LIGHTS = np.dtype([("kind", "<f4"), ("geom", "<f4", 6), ("L", "<f4", 3), ("p", "<f4", 3), ("n", "<f4", 3), ("u", "<f4", 2), ("wi", "<f4", 3), ("pdf", "<f4"),
                         ("Li", "<f4", 3), ("ray_o", "<f4", 3), ("ray_d", "<f4", 3), ("ray_tmax", "<f4"), ("wi2", "<f4", 3), ("pdf_b", "<f4"), ("le", "<f4", 3)])
CAMERA_RAYS = np.dtype([("cfg", "<i4"), ("px", "<i4"), ("py", "<i4"), ("s", "<i4"), ("eye", "<f4", 3), ("look", "<f4", 3), ("up", "<f4", 3), ("fov", "<f4"), ("lensr", "<f4"),
                        ("focald", "<f4"), ("aspect", "<f4"), ("xres", "<i4"), ("yres", "<i4"), ("crop", "<f4", 4), ("spp", "<i4"),
                        ("p_film", "<f4", 2), ("p_lens", "<f4", 2), ("time", "<f4"), ("o", "<f4", 3), ("d", "<f4", 3), ("weight", "<f4"),
                        ("rx_o", "<f4", 3), ("rx_d", "<f4", 3), ("ry_o", "<f4", 3), ("ry_d", "<f4", 3)])
BSSRDF_TABLES = np.dtype([("g", "<f4"), ("eta", "<f4"), ("rho_samples", "<f4", 100), ("radius_samples", "<f4", 64), ("profile", "<f4", 6400), ("rho_eff", "<f4", 100),
                          ("profile_cdf", "<f4", 6400)])
BSSRDF_RADIAL = np.dtype([("sigma_a", "<f4", 3), ("sigma_s", "<f4", 3), ("ch", "<i4"), ("r", "<f4"), ("u", "<f4"), ("sr", "<f4", 3), ("sample_sr", "<f4"), ("pdf_sr", "<f4"),
                          ("kd", "<f4", 3), ("mfp", "<f4", 3), ("out_sigma_a", "<f4", 3), ("out_sigma_s", "<f4", 3)])
HG = np.dtype([("g", "<f4"), ("wo", "<f4", 3), ("wi", "<f4", 3), ("u", "<f4", 2), ("p", "<f4"), ("wi_s", "<f4", 3), ("p_s", "<f4")])
SPECTRA = np.dtype([("kind", "<i4"), ("n", "<i4"), ("vals", "<f4", 80), ("rgb", "<f4", 3)])


def main():
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp()
    subprocess.check_call([os.path.join(REF, "ref_probe"), tmp, os.path.join(ROOT, "scenes", "envmap_40x20.pfm")])
    np.savez_compressed(os.path.join(OUT, "spectra_vectors.npz"), spectra=np.fromfile(os.path.join(tmp, "spectra.bin"), dtype=SPECTRA))
    np.savez_compressed(os.path.join(OUT, "light_vectors.npz"), light_samples=np.fromfile(os.path.join(tmp, "light_samples.bin"), dtype=LIGHT_SAMPLES),
                        scene_lights=np.fromfile(os.path.join(tmp, "light_samples_scene.bin"), dtype=SCENE_LIGHTS))
    np.savez_compressed(os.path.join(OUT, "bssrdf_tables.npz"), tables=np.fromfile(os.path.join(tmp, "bssrdf_tables.bin"), dtype=BSSRDF_TABLES),
                        radial=np.fromfile(os.path.join(tmp, "bssrdf_radial.bin"), dtype=BSSRDF_RADIAL), hg=np.fromfile(os.path.join(tmp, "hg.bin"), dtype=HG))
    np.savez_compressed(os.path.join(OUT, "camera_vectors.npz"), camera_rays=np.fromfile(os.path.join(tmp, "camera_rays.bin"), dtype=CAMERA_RAYS))
    if "--only-spectra" in sys.argv or "--only-kat" in sys.argv:
        return
    ss = np.fromfile(os.path.join(tmp, "sobol_samples.bin"), dtype=np.dtype([("i", "<i8"), ("d", "<i4"), ("v", "<f4")]))
    si = np.fromfile(os.path.join(tmp, "sobol_index.bin"), dtype=np.dtype([("m", "<u4"), ("frame", "<u8"), ("px", "<i4"), ("py", "<i4"), ("idx", "<u8")]))
    sp = np.fromfile(os.path.join(tmp, "sobol_sampler.bin"), dtype=np.dtype([("px", "<i4"), ("py", "<i4"), ("s", "<i4"), ("u", "<f4", 24)]))
    hp = np.fromfile(os.path.join(tmp, "halton_sampler.bin"), dtype=np.dtype([("px", "<i4"), ("py", "<i4"), ("s", "<i4"), ("u", "<f4", 24)]))
    sph = np.fromfile(os.path.join(tmp, "spheres.bin"), dtype=np.dtype([("o2w", "<f4", 16), ("w2o", "<f4", 16), ("radius", "<f4"), ("zmin", "<f4"), ("zmax", "<f4"),
                      ("theta_min", "<f4"), ("theta_max", "<f4"), ("phi_max", "<f4"), ("area", "<f4"), ("flags", "<i4"), ("o", "<f4", 3), ("d", "<f4", 3), ("tmax", "<f4"),
                      ("hit", "<i4"), ("t", "<f4"), ("p", "<f4", 3), ("p_error", "<f4", 3), ("n", "<f4", 3)]))
    bx = np.fromfile(os.path.join(tmp, "bxdfs.bin"), dtype=np.dtype([("bxdf", "V100"), ("wo", "<f4", 3), ("wi", "<f4", 3), ("u", "<f4", 2), ("f", "<f4", 3), ("pdf", "<f4"),
                     ("wi_s", "<f4", 3), ("pdf_s", "<f4"), ("f_s", "<f4", 3), ("type_s", "<i4")]))
    tri = np.fromfile(os.path.join(tmp, "triangles.bin"), dtype=np.dtype([("p", "<f4", 9), ("o", "<f4", 3), ("d", "<f4", 3), ("tmax", "<f4"), ("hit", "<i4"),
                                                                           ("t", "<f4"), ("uv", "<f4", 2), ("b1", "<f4"), ("b2", "<f4"), ("n", "<f4", 3)]))
    keep = np.zeros(len(tri), dtype=bool); keep[0] = True; keep[1::4] = True   # BadCases record + every 4th
    tri = tri[keep]
    raw = np.fromfile(os.path.join(tmp, "distribution1d.bin"), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "ref_vectors.npz"), sobol_samples=ss, sobol_index=si, sobol_sampler=sp, halton_sampler=hp, spheres=sph, bxdfs=bx, triangles=tri, distribution1d=raw)
    print("ref_vectors.npz:", len(ss), "sobol samples,", len(si), "indices,", len(sp), "sampler rows,", len(tri), "triangle records")

    # strategy None = as the scene file says (cornell: uniform, materials: power); "spatial" = the reference's default
    scenes = [("cornell", 64, 64, 1, None), ("cornell", 64, 64, 8, None), ("materials", 96, 72, 1, None), ("materials", 96, 72, 16, None),
              ("cornell", 64, 64, 4, "spatial"), ("materials", 96, 72, 4, "spatial"),
              # wide pixel filters (overlapping footprints, sample bounds larger than the image): variant = key of FILTERS
              ("cornell", 64, 48, 4, "gaussian"), ("cornell", 64, 48, 4, "mitchell"), ("cornell", 64, 48, 4, "triangle"), ("cornell", 64, 48, 4, "sinc"),
              # HaltonSampler (pbrt's default), sample counts that are not powers of two
              ("cornell", 64, 64, 6, "halton"), ("materials", 96, 72, 5, "halton")]
    for name, w, h, spp, strategy in scenes:
        text = scene_text(name, w, h, spp, strategy)
        f = os.path.join(tmp, "s.pbrt"); open(f, "w").write(text)
        out = os.path.join(OUT, "%s_%dx%d_%dspp%s.pfm" % (name, w, h, spp, "_" + strategy if strategy else ""))
        subprocess.check_call([os.path.join(REF, "pbrt_ref"), "--quiet", "--nthreads", "1", "--outfile", out, f])   # one thread: the tile merge order (wide filters) is then fixed
        print("rendered", out)
    edge_fixtures(tmp)
    # the reference's own example scene, UNMODIFIED except for resolution and output name (scenes/killeroo-simple.pbrt: Sphere area
    # light, Halton sampler, loopsubdiv geometry) -- the tests render this repository's scenes/killeroo.pbrt against it
    text = open("/root/reference/scenes/killeroo-simple.pbrt").read()
    text = text.replace('"integer xresolution" [700] "integer yresolution" [700]', '"integer xresolution" [96] "integer yresolution" [96]')
    text = text.replace('"string filename" "killeroo-simple.exr"', '"string filename" "k.pfm"').replace("geometry/killeroo.pbrt", "/root/reference/scenes/geometry/killeroo.pbrt")
    assert "[96]" in text
    f = os.path.join(tmp, "ks.pbrt"); open(f, "w").write(text)
    out = os.path.join(OUT, "killeroo_simple_96x96_reference.pfm")
    subprocess.check_call([os.path.join(REF, "pbrt_ref"), "--quiet", "--nthreads", "1", "--outfile", out, f])
    print("rendered", out)


FILTERS = {"gaussian": 'PixelFilter "gaussian"', "mitchell": 'PixelFilter "mitchell" "float xwidth" [2.5] "float ywidth" [1.5]',
           "triangle": 'PixelFilter "triangle" "float xwidth" [1.5] "float ywidth" [2.25]', "sinc": 'PixelFilter "sinc" "float xwidth" [3] "float ywidth" [3] "float tau" [2.5]'}


def edge_fixtures(tmp):
    """tests/edge_scenes.py variants (infinite light, thin lens, crop window + pixel bounds, luminance clamp, empty world, ...)
    and its textured scenes (image / procedural textures, mappings, bump maps, alpha masks, textured material parameters)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import edge_scenes
    for name in edge_scenes.NAMES + edge_scenes.C5_NAMES + edge_scenes.TEX_NAMES + edge_scenes.TEX_ORACLE_ONLY + ["instances2"] + edge_scenes.VOL_NAMES + edge_scenes.SSS_NAMES + edge_scenes.SAMPLER_NAMES + edge_scenes.STUB_ONLY_SAMPLER_NAMES:
        f = os.path.join(tmp, "e.pbrt"); open(f, "w").write(edge_scenes.scene(name))
        out = os.path.join(OUT, "edge_%s.pfm" % name)
        subprocess.check_call([os.path.join(REF, "pbrt_ref"), "--quiet", "--nthreads", "1", "--outfile", out, f])
        print("rendered", out)
    for name in edge_scenes.FURNACE_NAMES:   # the reference's analytic furnace scenes (src/tests/analytic_scenes.cpp), Sobol' 256, path depth 8
        f = os.path.join(tmp, "e.pbrt"); open(f, "w").write(edge_scenes.furnace_scene(name))
        out = os.path.join(OUT, "%s.pfm" % name)
        subprocess.check_call([os.path.join(REF, "pbrt_ref"), "--quiet", "--nthreads", "1", "--outfile", out, f])
        print("rendered", out)


def scene_text(name, w, h, spp, strategy=None):
    """the scene edits behind each fixture (also used by the tests)"""
    import re
    text = open(os.path.join(ROOT, "scenes", name + ".pbrt")).read()
    text = re.sub(r'"integer xresolution" \[\d+\] "integer yresolution" \[\d+\]', '"integer xresolution" [%d] "integer yresolution" [%d]' % (w, h), text)
    if strategy == "halton":
        text = re.sub(r'Sampler "sobol"', 'Sampler "halton"', text)
    elif strategy in FILTERS:
        assert 'PixelFilter "box"' in text
        text = text.replace('PixelFilter "box"', FILTERS[strategy])
    elif strategy:
        text = re.sub(r'"string lightsamplestrategy" "\w+"', '"string lightsamplestrategy" "%s"' % strategy, text)
    return re.sub(r'"integer pixelsamples" \[\d+\]', '"integer pixelsamples" [%d]' % spp, text)


if __name__ == "__main__":
    main()
