"""-m gpu, LAST in a session (conftest.py orders `multirank` tests after everything else): bench.py's N > 1 launch path end to end on a one-GPU box -- two ranks
under torch.distributed.run, both on GPU 0 (--one-device moves the device ordinal only), backend gloo (RCCL refuses two ranks on one device).

Every job here runs in its own session with a hard limit; at the limit the whole process GROUP is killed and the test fails with the ranks' phase log and
stacks (bench.py prints one line per rank and phase and dumps every rank's stack periodically), so a stall names the call it stalled in."""
import json
import os
import signal
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol

pa = ol.pa
ROOT = ol.ROOT
pytestmark = [pytest.mark.gpu, pytest.mark.multirank]

BENCH = os.path.join(ROOT, "bench.py")
SMALL = ["--steps", "1", "--warmup", "1", "--tris", "200000", "--res", "320", "192", "--spp", "4", "--cpu-seconds", "0", "--traffic", "none"]
SCENE_KEY = "sanmiguel_synth_200k_320x192_4spp"


def run_job(args, tmp_path, limit_s=300, extra_env=None):
    """`python bench.py args` in its own process group, stderr to a file; (rc, stdout, stderr) -- rc None = killed at the limit"""
    env = dict(os.environ, PBRT_AMD_BENCH_DIR=str(tmp_path), PBRT_AMD_BENCH_STACKS_S="60", PBRT_AMD_BENCH_WAIT_S="120", PBRT_AMD_PG_TIMEOUT_S="120")
    env.update(extra_env or {})
    err_path, out_path = str(tmp_path / "job.err"), str(tmp_path / "job.out")
    with open(err_path, "w") as ferr, open(out_path, "w") as fout:
        p = subprocess.Popen([sys.executable, BENCH] + args, stdout=fout, stderr=ferr, env=env, start_new_session=True)
        try:
            rc = p.wait(timeout=limit_s)
        except subprocess.TimeoutExpired:
            rc = None
        finally:
            try:
                os.killpg(p.pid, signal.SIGKILL)   # launcher, torchrun and every rank: nothing stays on the GPU
            except (ProcessLookupError, PermissionError):
                pass
            p.wait()
    return rc, open(out_path).read(), open(err_path).read()


def test_bench_two_ranks_self_launch_end_to_end(tmp_path):
    """`python bench.py --gpus 2` starts its own two ranks (one process per GPU), local rank 0 generates the scene, builds it and publishes the blob, rank 1 MAPS
    it; the tiles are sharded, rank 1's reachable FilmTilePixels are added into rank 0's film, ONE JSON line for the whole job.  Checked: the line; the
    whole-job sample count (every pixel exactly once across the ranks); how each rank got its scene; and the IMAGE -- rank 0's combined film equals the
    one-process render of the same scene file, bit for bit on the pixels that only hold their own samples (film.cpp:117-130: a merge of disjoint tiles)."""
    film2 = str(tmp_path / "film2.npy")
    rc, out, err = run_job(["--gpus", "2", "--one-device", "--backend", "gloo", "--dump-film", film2] + SMALL, tmp_path)
    assert rc == 0, "rc %s (None = killed at the limit)\n%s" % (rc, err[-6000:])
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "strong" and d["unit"] == "Msamples/s"
    assert abs(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 - 320 * 192 * 4) <= 0.01 * 320 * 192 * 4   # all samples of the frame, once
    assert d["setup_s"]["scene_by_rank"] == ["built", "mapped"], d["setup_s"]
    assert err.count("wrote ") and err.count("sanmiguel_synth.pbrt") and len([l for l in err.splitlines() if l.startswith("wrote ") and l.endswith(".pbrt")]) == 1, "the scene was generated more than once"
    # the image: one process, one context, the same file
    got = np.load(film2)
    sc = pa.Scene(os.path.join(str(tmp_path), SCENE_KEY, "sanmiguel_synth.pbrt"), strict=True)
    ctx = pa.Context(sc)
    ctx.render()
    ref = ctx.film()
    ctx.close()
    assert got.shape == ref.shape == (192, 320, 4)
    own_only = ref[..., 3] == sc.info["spp"]
    assert own_only.mean() > 0.98
    assert np.array_equal(got[own_only].view(np.uint32), ref[own_only].view(np.uint32))
    assert np.allclose(got, ref, rtol=1e-6, atol=1e-7)
    assert np.array_equal(got[..., 3] != 0, ref[..., 3] != 0)
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("pbrt_amd_scene_") and ("%d" % os.getpid()) in f]   # the blob is gone


def test_bench_one_rank_film_equals_the_library_render(tmp_path):
    """the N = 1 line of the same command dumps the same film the library renders in-process (the dump is what the 2-rank test compares)"""
    film1 = str(tmp_path / "film1.npy")
    rc, out, err = run_job(["--gpus", "1", "--dump-film", film1, "--secondary", "off"] + SMALL, tmp_path)
    assert rc == 0, err[-3000:]
    sc = pa.Scene(os.path.join(str(tmp_path), SCENE_KEY, "sanmiguel_synth.pbrt"), strict=True)
    ctx = pa.Context(sc)
    ctx.render()
    own_only = ctx.film()[..., 3] == sc.info["spp"]
    assert np.array_equal(np.load(film1)[own_only].view(np.uint32), ctx.film()[own_only].view(np.uint32))
    ctx.close()


def test_a_cut_off_scene_file_fails_both_ranks_loudly(tmp_path):
    """A PLY file shorter than its header promises (cut off, or caught half-written) is an ERROR on the rank that reads it -- never a smaller scene: local rank 0
    fails its strict load and announces why, rank 1 stops waiting and fails with that reason (or is ended by the launcher first), the job ends non-zero well
    inside the limit and prints no line."""
    d = tmp_path / SCENE_KEY
    d.mkdir()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_scenes.py"), "sanmiguel", "--tris", "200000", "--res", "320", "192", "--spp", "4",
                           "--out", str(d / "sanmiguel_synth.pbrt")], stdout=subprocess.DEVNULL)
    ply = d / "sanmiguel_synth_geo" / "m03.ply"
    size = ply.stat().st_size
    with open(ply, "r+b") as f:
        f.truncate(size - size // 3)
    (d / ".done").write_text("ok")
    rc, out, err = run_job(["--gpus", "2", "--one-device", "--backend", "gloo"] + SMALL, tmp_path, limit_s=240)
    assert rc is not None and rc != 0, "rc %s\n%s" % (rc, err[-3000:])
    assert not [l for l in out.splitlines() if l.startswith("{")]
    assert "Unable to read the contents of PLY file" in err and "strict mode" in err, err[-3000:]
    assert "Traceback" in err and "timed out" not in err
