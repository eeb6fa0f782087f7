"""One scene build per node (SURVEY.md s.8 row e): Scene.save_blob writes the flattened scene -- the mi_scene_desc and every array it points to, nested
ones included -- into one file; Scene(blob=...) maps it in another rank.  The mapped description must be the builder's, byte for byte: checked here end
to end through the CPU oracle (the film it renders from the mapped description equals the film from the parsed scene bit for bit) over scenes that
cover every nested array of the description: radiance maps, image pyramids, grid media, BSSRDF tables, two-level instancing, spheres, alpha masks."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol
import edge_scenes as es

pa = importlib.import_module("pbrt-v3-distributed_amd")
ROOT = ol.ROOT

NAMES = ["cornell", "envmap", "tex_imagemap", "tex_alpha", "vol_smoke", "vol_inst", "sss_kd", "instances2", "spheres", "empty"]


def _text(name):
    if name == "cornell":
        t = open(os.path.join(ROOT, "scenes", "cornell.pbrt")).read()
        return t.replace('"integer xresolution" [400] "integer yresolution" [400]', '"integer xresolution" [48] "integer yresolution" [48]').replace('"integer pixelsamples" [8]', '"integer pixelsamples" [2]')
    return es.scene(name)


@pytest.mark.parametrize("name", NAMES)
def test_mapped_blob_renders_the_builders_film(name, tmp_path):
    sc = pa.Scene(text=_text(name))
    path = str(tmp_path / (name + ".blob"))
    sc.save_blob(path)
    m = pa.Scene(blob=path)
    assert m.mapped and not sc.mapped
    assert m.info == sc.info and (m.width, m.height) == (sc.width, sc.height)
    for i in range(sc.info["n_lights"]):
        a, b = sc.light(i), m.light(i)
        assert a[0] == b[0] and np.array_equal(a[1], b[1])
    spp = min(2, sc.info["spp"])
    ref, cref, _ = ol.render(sc, 0, spp, 4)
    got, cgot, _ = ol.render(m, 0, spp, 4)
    assert cref == cgot
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))
    with pytest.raises(RuntimeError):
        m.film_image(got)          # the Film stays with the process that built the scene
    m.close(); sc.close()


def test_blob_is_published_atomically_and_rejects_damage(tmp_path):
    sc = pa.Scene(text=_text("cornell"))
    path = str(tmp_path / "c.blob")
    sc.save_blob(path)
    assert [f for f in os.listdir(tmp_path) if ".tmp." in f] == []   # written under a temporary name, then renamed
    data = open(path, "rb").read()
    with pytest.raises(RuntimeError):
        pa.Scene(blob=str(tmp_path / "missing.blob"))
    open(str(tmp_path / "short.blob"), "wb").write(data[: len(data) // 2])
    with pytest.raises(RuntimeError):
        pa.Scene(blob=str(tmp_path / "short.blob"))
    bad = bytearray(data); bad[8] ^= 0xFF                                # the ABI version field
    open(str(tmp_path / "abi.blob"), "wb").write(bytes(bad))
    with pytest.raises(RuntimeError):
        pa.Scene(blob=str(tmp_path / "abi.blob"))
    sc.close()


def test_blob_mapped_in_another_process(tmp_path):
    """the use case: rank 0 builds, another PROCESS maps and gets the same description (checked through the oracle's camera rays + first hits)"""
    sc = pa.Scene(text=_text("instances2"))
    path = str(tmp_path / "i.blob")
    sc.save_blob(path)
    xy = np.stack([np.arange(200) % sc.width, (np.arange(200) * 7) % sc.height], 1).astype(np.int32)
    rays, _ = ol.camera_rays(sc, xy, np.zeros(200, np.int32))
    hits, _ = ol.intersect(sc, rays)
    np.save(str(tmp_path / "prim.npy"), hits["prim"]); np.save(str(tmp_path / "t.npy"), hits["t"])
    code = ("import sys, importlib, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import oracle_lib as ol; pa = importlib.import_module('pbrt-v3-distributed_amd');"
            "m = pa.Scene(blob=%r); xy = np.stack([np.arange(200) %% m.width, (np.arange(200) * 7) %% m.height], 1).astype(np.int32);"
            "rays, _ = ol.camera_rays(m, xy, np.zeros(200, np.int32)); h, _ = ol.intersect(m, rays);"
            "assert np.array_equal(h['prim'], np.load(%r)) and np.array_equal(h['t'].view(np.uint32), np.load(%r).view(np.uint32)); print('same')"
            % (ROOT, os.path.join(ROOT, "tests"), path, str(tmp_path / "prim.npy"), str(tmp_path / "t.npy")))
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "same" in r.stdout, r.stdout[-800:]
    sc.close()
