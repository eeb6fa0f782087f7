"""-m gpu: Sampler "random" / "stratified" / "02sequence" ("lowdiscrepancy") on the device (ABI v11).

Their values come from ONE PCG32 stream per 16 x 16 tile (sampler->Clone(seed = tile index), integrator.cpp:246-248; core/sampler.cpp:100-135,
samplers/{random,stratified,zerotwosequence}.cpp), so what a sample receives depends on how many numbers every earlier sample of its tile drew.
mi_render walks the tiles' pixels and samples in the reference's order, one path per tile in flight (tile-serial rounds: k_pix_seed,
k_pix_start_pixel, the SMP = 2 instances of k_shade, Sampler::PixGet1D / PixGet2D).  The fixtures are pbrt_ref's renders of the same scenes
(tests/golden/edge_sampler_*.pfm, tools/gen_golden.py); the oracle equals them bit for bit (tests/test_oracle_golden.py)."""
import os

import numpy as np
import pytest

import edge_scenes
import oracle_lib as ol

pa = ol.pa
G = os.path.join(ol.ROOT, "tests", "golden")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", edge_scenes.SAMPLER_NAMES)
def test_tile_serial_samplers_match_reference(name):
    """partial tiles, depth of field (the lens sample is 2D dimension 1), ONE precomputed dimension + no jitter (everything after the film sample
    from the stream), crop window + pixel bounds (StartPixel also runs for the pixels outside the bounds), volpath in fog with 3 -> 4 samples"""
    sc = pa.Scene(text=edge_scenes.scene(name))
    ctx = pa.Context(sc)
    ctx.counters_reset()
    ctx.render()
    img = sc.film_image(ctx.film())
    ref = pa.read_pfm(os.path.join(G, "edge_%s.pfm" % name))
    assert img.shape == ref.shape
    frac, relmse = ol.image_metrics(img, ref)
    print("%s: pixels within tolerance %.5f, relMSE %.3g, bit-identical pixels %.4f" % (name, frac, relmse, (img.view(np.uint32) == ref.view(np.uint32)).all(-1).mean()))
    assert frac >= 0.995 and relmse <= 1e-4, (name, frac, relmse)
    assert ctx.counters()["trace_guard_trips"] == 0
    ctx.close()


def test_tile_serial_frames_shard_by_tile_and_refuse_partial_work():
    """tiles own their streams: rank r of 2 renders exactly its tiles' pixels of the one-rank frame; a sample range or a per-sample query has no
    meaning for these samplers and is refused"""
    sc = pa.Scene(text=edge_scenes.scene("sampler_stratified"))
    one = pa.Context(sc)
    one.render()
    full = one.film().copy()
    acc = np.zeros_like(full)
    for r in range(2):
        ctx = pa.Context(sc)
        ctx.render(rank=r, world=2)
        acc += ctx.film()
        ctx.close()
    assert np.array_equal(acc.view(np.uint32), full.view(np.uint32))   # box filter: every pixel's samples come from its own tile
    with pytest.raises(RuntimeError):
        one.render(spp_begin=0, spp_end=2)
    with pytest.raises(RuntimeError):
        one.li(np.array([[3, 3]], dtype=np.int32), np.array([0], dtype=np.int32))
    one.close()
