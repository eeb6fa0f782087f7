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


def node_scene(load, blob_path, local_rank, timeout_s=3600.0):
    """One scene build per NODE: local rank 0 calls `load()` (parse + the reference's BVH build, with every CPU of the node: the other ranks only
    wait) and publishes the flattened scene with Scene.save_blob(blob_path); the other local ranks map that file (Scene(blob=...): shared page
    cache, no second copy of the geometry in host memory, no second BVH build).  `blob_path` must be unique per job (callers put the launcher's
    pid into it) so that a blob left behind by an earlier job is never mapped.  Returns (scene, seconds, "built" | "mapped")."""
    import importlib, os, time
    pa = importlib.import_module(__package__)
    t0 = time.time()
    if local_rank == 0:
        sc = load()
        try:
            sc.save_blob(blob_path)
        except Exception as e:   # no room in /dev/shm, read-only directory ...: the other ranks build the scene themselves instead of waiting for nothing
            open(blob_path + ".failed", "w").write(str(e))
        return sc, time.time() - t0, "built"
    while not os.path.exists(blob_path):
        if os.path.exists(blob_path + ".failed"):
            return load(), time.time() - t0, "built (rank 0 could not publish the scene)"
        if time.time() - t0 > timeout_s:
            raise RuntimeError("node_scene: %s was not published within %.0f s" % (blob_path, timeout_s))
        time.sleep(0.1)
    return pa.Scene(blob=blob_path), time.time() - t0, "mapped"


def launch_ranks(n_ranks, script, argv, backend_env=None):
    """Start `script argv` as n_ranks processes of ONE node under torch.distributed.run (one process per GPU, rendezvous on
    127.0.0.1 with a free port -- the launch line the round driver uses) and return the job's exit code.  Used by
    `python bench.py --gpus N` when it was not started by a launcher itself."""
    import os, socket, subprocess, sys
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(script)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    env.update(backend_env or {})
    return subprocess.call(cmd, env=env)


class ShardedFrame:
    """One rank of the tile-sharded frame: the rank's device context renders the tiles t with t % world == rank into a film that
    lives in a torch tensor, and `step()` ends with the only exchange of the path -- the SUM-reduce of the films onto rank 0
    (RCCL ncclReduce over xGMI when the backend is "nccl").  world == 1: no torch, no collective.

        frame = ShardedFrame(ctx, scene, backend="nccl", one_device=False)   # reads RANK / WORLD_SIZE / LOCAL_RANK
        frame.step(); frame.sync_all()
    """

    def __init__(self, ctx, scene, rank, world, local_rank, backend="nccl", one_device=False):
        self.ctx, self.scene, self.rank, self.world = ctx, scene, rank, world
        self.torch = self.dist = self.film = None
        if world > 1:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            dev = 0 if one_device else local_rank
            torch.cuda.set_device(dev)
            if not dist.is_initialized():
                if backend == "nccl":
                    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
                else:
                    dist.init_process_group(backend=backend)
            self.film = torch.zeros(scene.height * scene.width * 4, dtype=torch.float32, device="cuda")
            ctx.film_bind(self.film.data_ptr())   # mi_render accumulates straight into the tensor the collective reduces

    def step(self, count_work=False, max_paths=0):
        self.ctx.film_clear()
        self.ctx.render(rank=self.rank, world=self.world, count_work=count_work, max_paths=max_paths, sync=False)
        if self.world > 1:
            self.ctx.sync()                    # the ctx stream is not torch's current stream
            combine_films(self.film, dst=0)
            self.torch.cuda.synchronize()      # the reduction reads the film: done before the next step clears it

    def sync_all(self):
        self.ctx.sync()
        if self.world > 1:
            self.torch.cuda.synchronize()
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, value):
        if self.world == 1:
            return float(value)
        t = self.torch.tensor([value], dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, values):
        import numpy as np
        v = np.asarray(values, dtype=np.float64)
        if self.world == 1:
            return v
        t = self.torch.tensor(v, dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def root_film(self):
        """rank 0: the combined FilmTilePixel array (H, W, 4) after a step"""
        if self.world == 1:
            return self.ctx.film()
        return self.film.cpu().numpy().reshape(self.scene.height, self.scene.width, 4)

    def close(self):
        if self.world > 1 and self.dist.is_initialized():
            self.dist.barrier()
            self.dist.destroy_process_group()
