"""Multi-GPU plumbing of the tile-sharded render (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards naturally: image tiles (the reference's 16x16 tiles, core/integrator.cpp:233-240) are independent given
the replicated scene, and the Sobol' sample of (pixel, k) is a pure function of the pixel and k (samplers/sobol.cpp:42-45).
Tile t = ty * nTilesX + tx belongs to rank t % world (mi_render applies the same rule on the device).  The only exchange
is the FilmTilePixel buffers at the end of a frame: every rank holds a full-size film that is zero outside its tiles
(plus, rarely, a neighbour pixel that a sample landing exactly on a pixel edge also contributes to), so a SUM reduction to
rank 0 is exactly the gather of the owned tiles -- and stays exact for those edge pixels, which a plain gather would drop.
"""


def owned_tiles(rank, world, n_tiles_x, n_tiles_y):
    """tile ids of `rank`: round-robin over the row-major tile index (same rule as mi_render / oracle_render_sharded)"""
    return list(range(rank, n_tiles_x * n_tiles_y, world))


def combine_films(film, dst=0):
    """In-place SUM-reduce of the per-rank film tensors (float32, 4 per cropped pixel) onto rank `dst`.
    Works for CPU tensors (gloo) and for device tensors (RCCL over xGMI)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film
