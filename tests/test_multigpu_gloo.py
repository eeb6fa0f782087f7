"""CPU, world_size 2, gloo: the N>1 path -- tile ownership (mi_tile_owner) + the film exchange of parallel.py, dense and sparse -- with the
oracle standing in for the device renderer (the sharding rule is the same function of (tile, rank, world) on both)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

import oracle_lib as ol

pa = ol.pa


def _worker(rank, world, port, scene_text, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = importlib.import_module("pbrt-v3-distributed_amd.parallel")
    # one scene build per node: rank 0 parses + builds and publishes the flattened scene, rank 1 maps it (parallel.node_scene)
    sc, _, how = par.node_scene(lambda: pa.Scene(text=scene_text), os.path.join(out_dir, "scene_%d.blob" % os.getppid()), rank)
    assert how == ("built" if rank == 0 else "mapped") and sc.mapped == (rank != 0)
    rgbw, cnt, _ = ol.render(sc, nthreads=2, rank=rank, world=world)
    ntx, nty = (sc.width + 15) // 16, (sc.height + 15) // 16
    mine = par.owned_tiles(rank, world, ntx, nty)
    # the rank's film is zero outside its tiles (up to edge-spill pixels)
    mask = np.zeros((sc.height, sc.width), dtype=bool)
    for t in mine:
        ty, tx = divmod(t, ntx)
        mask[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] = True
    assert cnt["camera_rays"] == int(mask.sum()) * sc.info["spp"]
    outside = rgbw[~mask]
    assert (outside[..., 3] != 0).mean() < 0.02
    # everything a rank's samples touched lies inside reach_pixels (its tiles + the filter's ring): the sparse exchange moves only those
    reach = par.reach_pixels(rank, world, sc.width, sc.height, sc.info["sample_bounds"], (sc.info["crop_x0"], sc.info["crop_y0"]), sc.info["filter_radius"])
    touched = np.flatnonzero((rgbw.reshape(-1, 4) != 0).any(1))
    assert np.isin(touched, reach).all() and len(reach) < 0.75 * sc.width * sc.height
    film = torch.from_numpy(rgbw.reshape(-1).copy())
    sparse = film.clone()
    par.combine_films(film, dst=0)                                  # dense: SUM-reduce of the whole films
    xchg = par.FilmExchange(sc, rank, world, torch.device("cpu"))   # sparse: reachable pixels only, added on rank 0 (what ShardedFrame.step does)
    xchg.finish(xchg.start(sparse))
    if rank == 0:
        if world == 2:
            assert np.array_equal(sparse.numpy().view(np.uint32), film.numpy().view(np.uint32))   # (two ranks: one addition per pixel either way)
        else:   # more senders: rank 0's N - 1 receives are one group, added in rank order; a pixel three ranks reach may round differently from gloo's reduction tree
            assert np.allclose(sparse.numpy(), film.numpy(), rtol=1e-6, atol=1e-7)
            assert (sparse.numpy().view(np.uint32) == film.numpy().view(np.uint32)).mean() > 0.999
        np.save(os.path.join(out_dir, "combined.npy"), sparse.numpy().reshape(sc.height, sc.width, 4))
    dist.barrier()
    dist.destroy_process_group()


def _scene_text(which):
    if which == "cornell":
        text = open(os.path.join(ol.ROOT, "scenes", "cornell.pbrt")).read()
        return text.replace('[400] "integer yresolution" [400]', '[80] "integer yresolution" [48]').replace('"integer pixelsamples" [8]', '"integer pixelsamples" [4]')
    sys.path.insert(0, os.path.join(ol.ROOT, "tests"))
    import edge_scenes
    return edge_scenes.scene(which)   # a textured scene: the image pyramids travel inside the blob


@pytest.mark.parametrize("which,world", [("cornell", 2), ("tex_imagemap", 2), ("cornell", 4)])
def test_two_rank_tile_sharding_and_film_reduce(built, tmp_path, which, world):
    """(world 4: three senders -- the grouped receives of FilmExchange and the rank-order adds with more than one source, the shape of the driver's N = 4 / 8 runs)"""
    import torch.multiprocessing as mp
    text = _scene_text(which)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(world, port, text, str(tmp_path)), nprocs=world, join=True)
    combined = np.load(tmp_path / "combined.npy")
    sc = pa.Scene(text=text)
    whole, _, _ = ol.render(sc, nthreads=2)
    own_only = whole[..., 3] == sc.info["spp"]
    assert np.array_equal(combined[own_only].view(np.uint32), whole[own_only].view(np.uint32))   # N-rank image == 1-rank image, bit for bit
    assert np.allclose(combined, whole, rtol=1e-6, atol=1e-7)


def _node_scene_worker(rank, port, texts, out_dir):
    os.environ["MASTER_PORT"] = str(port)
    par = importlib.import_module("pbrt-v3-distributed_amd.parallel")
    base = os.path.join(out_dir, "scene.blob")
    hows = []
    for k, text in enumerate(texts):   # two calls of one job under the same base name: the second must never map the first's blob
        sc, _, how = par.node_scene(lambda: pa.Scene(text=text), base, rank)
        hows.append((how, sc.width, sc.height))
    # third call: rank 0's load fails -- it says why to the waiting rank (which fails with that reason instead of loading what rank 0 could not) and re-raises
    if rank == 0:
        def boom():
            raise ValueError("no such scene")
        with pytest.raises(ValueError):
            par.node_scene(boom, base, rank)
        hows.append(("raised", 0, 0))
    else:
        with pytest.raises(RuntimeError, match="local rank 0 could not load the scene.*no such scene"):
            par.node_scene(lambda: pa.Scene(text=texts[0]), base, rank, timeout_s=120.0)
        hows.append(("told", 0, 0))
    np.save(os.path.join(out_dir, "dims_%d.npy" % rank), np.array([[h[1], h[2]] for h in hows]))
    np.save(os.path.join(out_dir, "hows_%d.npy" % rank), np.array([h[0] for h in hows]))
    if rank == 0:   # the publisher's exit removes its blobs (parallel.node_scene registers the cleanup): like every job, stay until the other rank has mapped them
        import time
        t0 = time.time()
        while not os.path.exists(os.path.join(out_dir, "hows_1.npy")) and time.time() - t0 < 120:
            time.sleep(0.05)


def test_node_scene_names_are_per_job_and_per_call_and_failures_are_announced(built, tmp_path):
    """parallel.node_scene: the published file's name carries the launcher's pid + start time, the rendezvous port and the call number, and the waiting ranks test for
    existence only.  Two calls with different scenes under one base name: the waiting rank maps each call's own blob (a stale file of another job under the OLD naming
    scheme, planted here, is ignored); a load that fails on rank 0 reaches the waiting rank as a '.failed' note -- it raises with rank 0's reason -- instead of a timeout."""
    import torch.multiprocessing as mp
    a = _scene_text("cornell")
    b = a.replace('[80] "integer yresolution" [48]', '[64] "integer yresolution" [32]')
    open(tmp_path / "scene.blob", "wb").write(b"stale")   # what an earlier job might have left under the bare base name
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_node_scene_worker, args=(port, [a, b], str(tmp_path)), nprocs=2, join=True)
    h0, h1 = np.load(tmp_path / "hows_0.npy"), np.load(tmp_path / "hows_1.npy")
    d0, d1 = np.load(tmp_path / "dims_0.npy"), np.load(tmp_path / "dims_1.npy")
    assert list(h0) == ["built", "built", "raised"]
    assert list(h1) == ["mapped", "mapped", "told"]
    assert d0[:2].tolist() == [[80, 48], [64, 32]] and d1[:2].tolist() == [[80, 48], [64, 32]]
    par = importlib.import_module("pbrt-v3-distributed_amd.parallel")
    assert par._launcher_id().startswith("%d_" % os.getppid())
    left = [f for f in os.listdir(tmp_path) if f.startswith("scene.blob.")]
    assert all(f.endswith(".failed") for f in left), left   # the publisher's exit removed its blobs; only the note for waiting ranks stays
    open(tmp_path / "scene.blob.999999999_123.0_none_0_1", "wb").write(b"left behind by a job that was killed")
    par._sweep_stale(str(tmp_path / "scene.blob"))
    assert not os.path.exists(tmp_path / "scene.blob.999999999_123.0_none_0_1") and os.path.exists(tmp_path / "scene.blob")
