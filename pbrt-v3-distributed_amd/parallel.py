"""Multi-GPU plumbing of the tile-sharded render (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards naturally: image tiles (the reference's 16x16 tiles, core/integrator.cpp:233-240) are independent given
the replicated scene, and the Sobol' sample of (pixel, k) is a pure function of the pixel and k (samplers/sobol.cpp:42-45).
Tile (tx, ty) belongs to rank mi_tile_owner(tx, ty, world) of include/pbrt_amd.h -- a skewed 2-D lattice, so that every rank's tiles are
spread over the whole image (mi_render applies the same rule on the device).  The only exchange
is the FilmTilePixel buffers at the end of a frame: every rank holds a full-size film that is zero outside its tiles
(plus, rarely, a neighbour pixel that a sample landing exactly on a pixel edge also contributes to), so a SUM reduction to
rank 0 is exactly the gather of the owned tiles -- and stays exact for those edge pixels, which a plain gather would drop.
"""


def owned_tiles(rank, world, n_tiles_x, n_tiles_y):
    """row-major tile ids of `rank`: mi_tile_owner of include/pbrt_amd.h (a skewed 2-D lattice over the tile grid), through the library's own
    mi_owned_tiles so that mi_render, the reference-side bindings, the test checker and this module share ONE definition"""
    import ctypes
    import numpy as np
    f = _owned_tiles_fn()
    n = f(n_tiles_x, n_tiles_y, rank, world, None)
    out = np.zeros(max(1, n), dtype=np.uint32)
    f(n_tiles_x, n_tiles_y, rank, world, out.ctypes.data_as(ctypes.c_void_p))
    return [int(t) for t in out[:n]]


_OWNED_TILES = None


def _owned_tiles_fn():
    """mi_owned_tiles of the device library, bound once"""
    global _OWNED_TILES
    if _OWNED_TILES is None:
        import ctypes, importlib
        L = importlib.import_module(__package__).device_lib()
        L.mi_owned_tiles.restype = ctypes.c_int64
        L.mi_owned_tiles.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _OWNED_TILES = L.mi_owned_tiles
    return _OWNED_TILES


def combine_films(film, dst=0):
    """In-place SUM-reduce of the per-rank film tensors (float32, 4 per cropped pixel) onto rank `dst`: the dense form of the exchange
    (every rank moves the whole film).  Works for CPU tensors (gloo) and for device tensors (RCCL over xGMI).  FilmExchange below is the sparse
    form ShardedFrame uses."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film


def reach_pixels(rank, world, width, height, sample_bounds, crop_origin, filter_radius):
    """Flat indices (row-major over the cropped film, ascending) of every film pixel the samples of `rank`'s tiles can contribute to: its tiles,
    each grown by floor(radius + 1/2) pixels per axis and clipped to the film (Film::GetFilmTile core/film.cpp:95-106 computes the same reach for
    one tile).  For the box filter (radius 1/2) that is the tile plus a one-pixel ring -- a sample that lands exactly on a pixel edge also counts
    for the neighbour."""
    import numpy as np
    sx0, sy0, sx1, sy1 = sample_bounds
    ntx, nty = (sx1 - sx0 + 15) // 16, (sy1 - sy0 + 15) // 16
    hx, hy = int(np.floor(filter_radius[0] + 0.5)), int(np.floor(filter_radius[1] + 0.5))
    mask = np.zeros((height, width), dtype=bool)
    cx0, cy0 = crop_origin
    for t in owned_tiles(rank, world, ntx, nty):
        ty, tx = divmod(t, ntx)
        x0, y0 = sx0 + 16 * tx - hx - cx0, sy0 + 16 * ty - hy - cy0
        x1, y1 = min(sx0 + 16 * tx + 16, sx1) + hx - cx0, min(sy0 + 16 * ty + 16, sy1) + hy - cy0
        mask[max(0, y0):max(0, min(height, y1)), max(0, x0):max(0, min(width, x1))] = True
    return np.flatnonzero(mask.reshape(-1))


class FilmExchange:
    """The one exchange of the path, sparse: rank r sends the FilmTilePixels its samples can reach (reach_pixels: its own tiles + the filter's
    ring, 16.6 MB x 1.27 per rank for a 4K film on 8 ranks under the box filter) to rank `dst`, which ADDS them into its film -- equal to the
    SUM-reduce of the full films (133 MB per rank) because a rank's film is zero everywhere else, exact for every filter, and deterministic
    (contributions are added in rank order).  ONE group of point-to-point operations per frame -- dist.batch_isend_irecv, i.e. ncclGroupStart / End with
    backend "nccl": rank 0's N - 1 receives are posted together and proceed concurrently over the N - 1 xGMI links into its GPU (separate irecv calls
    ran them one after the other on RCCL's stream: ADVICE r4); gloo moves the packed buffers through host memory."""

    def __init__(self, scene, rank, world, device, dst=0):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.rank, self.world, self.dst = torch, dist, rank, world, dst
        self.via_host = dist.get_backend() == "gloo"
        info = scene.info
        args = (scene.width, scene.height, info["sample_bounds"], (info["crop_x0"], info["crop_y0"]), info["filter_radius"])
        ranks = [r for r in range(world) if r != dst] if rank == dst else [rank]
        self.idx = {r: torch.from_numpy(reach_pixels(r, world, *args)).to(device) for r in ranks}
        comm_dev = "cpu" if self.via_host else device
        self.buf = {r: torch.empty((len(self.idx[r]), 4), dtype=torch.float32, device=comm_dev) for r in ranks}
        self.bytes_moved = sum(b.numel() * 4 for b in self.buf.values())

    def start(self, film):
        """Enqueue the exchange of `film` (flat float32 tensor, 4 per pixel) and return what finish() waits for.  With RCCL nothing here blocks the
        host: packing, the transfers and rank `dst`'s adds are stream-ordered on torch's current stream / RCCL's own (Work.wait() makes the
        current STREAM wait), so the caller goes on -- ShardedFrame renders the next frame meanwhile.  gloo completes inside this call."""
        torch, dist = self.torch, self.dist
        px = film.view(-1, 4)
        if self.rank != self.dst:
            packed = px.index_select(0, self.idx[self.rank])
            if self.via_host:
                self.buf[self.rank].copy_(packed)
                packed = self.buf[self.rank]
            for work in dist.batch_isend_irecv([dist.P2POp(dist.isend, packed, self.dst)]):
                work.wait()
            self._keep = packed   # alive until the send has been consumed (finish)
        else:
            srcs = sorted(self.buf)
            for work in dist.batch_isend_irecv([dist.P2POp(dist.irecv, self.buf[r], r) for r in srcs]):   # one group: the receives run concurrently
                work.wait()
            for r in srcs:   # rank order: the sum is deterministic
                px.index_add_(0, self.idx[r], self.buf[r].to(px.device) if self.via_host else self.buf[r])
        if film.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            return ev
        return None

    def finish(self, pending):
        if pending is not None:
            pending.synchronize()


def node_scene(load, blob_path, local_rank, timeout_s=1800.0):
    """One scene build per NODE: local rank 0 calls `load()` (parse + the reference's BVH build, with every CPU of the node: the other ranks only
    wait) and publishes the flattened scene with Scene.save_blob(blob_path); the other local ranks map that file (Scene(blob=...): shared page
    cache, no second copy of the geometry in host memory, no second BVH build).  The file name gets a per-job, per-call suffix (below), so that a blob left behind by
    an earlier job is never mapped.  Returns (scene, seconds, "built" | "mapped")."""
    import importlib, os, time
    pa = importlib.import_module(__package__)
    t0 = time.time()
    #  the name carries a nonce every rank of THIS job derives alike -- the launcher's pid AND start time (a reused pid gets another start time), the
    # rendezvous' port and run id, and the number of the call within the job -- so that a waiting rank can never map the blob of an earlier node_scene call, of an
    # earlier job whose launcher pid was reused, or of another job started from the same long-lived parent process.  Whatever exists under this name is this call's
    # blob: the waiting ranks test for existence only (no comparison of file times with process start times, which a stepped clock could turn into an hour's wait).
    global _NODE_SCENE_CALLS
    _NODE_SCENE_CALLS += 1
    blob_base = blob_path
    blob_path = "%s.%s.%s_%s_%s_%d" % (blob_path, _launcher_id(), os.environ.get("MASTER_PORT", "0"), "".join(ch for ch in os.environ.get("TORCHELASTIC_RUN_ID", "none") if ch.isalnum())[:24],
                                       os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), _NODE_SCENE_CALLS)
    launcher = os.getppid()
    if local_rank == 0:
        if os.path.isdir("/proc/self"):
            _sweep_stale(blob_base)
        _NODE_SCENE_FILES.append(blob_path)
        global _CLEANUP_REGISTERED
        if not _CLEANUP_REGISTERED:   #  an exit that skips node_scene_cleanup -- exception, SIGTERM from a launcher's timeout -- must not leave a multi-GB blob in /dev/shm
            import atexit, signal
            _CLEANUP_REGISTERED = True
            atexit.register(node_scene_cleanup)
            try:
                prev = signal.getsignal(signal.SIGTERM)
                def _on_term(signum, frame):
                    node_scene_cleanup()
                    if callable(prev):
                        prev(signum, frame)
                    raise SystemExit(128 + signum)
                signal.signal(signal.SIGTERM, _on_term)
            except ValueError:   # not the main thread
                pass
        for f in (blob_path, blob_path + ".failed"):   # leftovers of a crashed job with the same launcher pid
            try:
                os.remove(f)
            except OSError:
                pass
        try:
            sc = load()
        except BaseException as e:   # parse error, out of memory ...: tell the waiting ranks before re-raising, or they poll until the timeout
            _write_failed(blob_path, "load failed: %r" % (e,))
            raise
        try:
            sc.save_blob(blob_path + ".tmp")
            os.rename(blob_path + ".tmp", blob_path)   # published complete or not at all
        except Exception as e:   # no room in /dev/shm, read-only directory ...: the other ranks build the scene themselves instead of waiting for nothing
            _write_failed(blob_path, str(e))
        return sc, time.time() - t0, "built"
    while not os.path.exists(blob_path):   # (published by rename: complete or absent)
        if os.path.exists(blob_path + ".failed"):
            try:
                why = open(blob_path + ".failed").read()
            except OSError:
                why = ""
            if why.startswith("load failed"):   # the scene itself is bad: this rank would fail the same way (or, worse, load something different)
                raise RuntimeError("node_scene: local rank 0 could not load the scene -- %s" % why)
            return load(), time.time() - t0, "built (rank 0 could not publish the scene)"
        if time.time() - t0 > timeout_s:
            raise RuntimeError("node_scene: %s was not published within %.0f s" % (blob_path, timeout_s))
        if os.getppid() != launcher:   # the launcher is gone (killed at a timeout): nobody will publish anything
            raise RuntimeError("node_scene: the launcher (pid %d) exited while waiting for %s" % (launcher, blob_path))
        time.sleep(0.1)
    return pa.Scene(blob=blob_path), time.time() - t0, "mapped"


_NODE_SCENE_CALLS = 0
_NODE_SCENE_FILES = []
_CLEANUP_REGISTERED = False


def node_scene_cleanup():
    """remove the files node_scene published in this process (local rank 0, once every rank has mapped its scene: mappings stay valid; also registered for the
    exits that skip the explicit call).  A '.failed' note (a few bytes) stays for the ranks still waiting; the next job's sweep removes it."""
    import os
    for f in _NODE_SCENE_FILES:
        for g in (f, f + ".tmp"):
            try:
                os.remove(g)
            except OSError:
                pass
    del _NODE_SCENE_FILES[:]


def _sweep_stale(blob_base):
    """remove what node_scene calls of DEAD jobs left under this base name (a crash or SIGKILL skips every cleanup): files whose name carries a launcher pid that no
    longer exists, or exists with another start time"""
    import glob, os, re
    for f in glob.glob(glob.escape(blob_base) + ".*"):
        m = re.match(r"\.(\d+)_(\d+)\.", f[len(blob_base):])
        if not m:
            continue
        try:
            with open("/proc/%s/stat" % m.group(1)) as st:
                alive = m.group(2) == "0" or st.read().rsplit(")", 1)[1].split()[19] == m.group(2)
        except Exception:
            alive = False
        if not alive:
            try:
                os.remove(f)
            except OSError:
                pass


def _write_failed(blob_path, text):
    try:
        with open(blob_path + ".failed", "w") as f:
            f.write(text)
    except OSError:
        pass


def _launcher_id():
    """pid and start time (clock ticks since boot, /proc/<pid>/stat field 22) of this process's parent -- the launcher all local ranks share: unique per job on a node,
    also when the pid is reused.  Without /proc: the pid alone."""
    import os
    ppid = os.getppid()
    try:
        with open("/proc/%d/stat" % ppid) as f:
            return "%d_%s" % (ppid, f.read().rsplit(")", 1)[1].split()[19])
    except Exception:
        return "%d_0" % ppid


def die_with_parent(sig=None):
    """prctl(PR_SET_PDEATHSIG): this process gets `sig` (default SIGKILL) when its parent dies -- a rank never outlives its launcher on the GPU, whatever killed the
    launcher.  Linux only; elsewhere a no-op."""
    import ctypes, signal
    try:
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, int(sig if sig is not None else signal.SIGKILL), 0, 0, 0)
    except Exception:
        pass


def launch_ranks(n_ranks, script, argv, backend_env=None, timeout_s=None, after=None):
    """Start `script argv` as n_ranks processes of ONE node under torch.distributed.run (one process per GPU, rendezvous on
    127.0.0.1 with a free port -- the launch line the round driver uses) and return the job's exit code.  Used by
    `python bench.py --gpus N` when it was not started by a launcher itself.  The job runs in its own session (process group): at `timeout_s`
    ($PBRT_AMD_LAUNCH_TIMEOUT_S, default none), on SIGTERM / SIGINT to this process and on any exit of this function the WHOLE group is killed, so no rank
    outlives its launcher on the GPU (a timed-out job left two ranks behind and the next process waited 171 s for the device).  `after(pid)` is called
    with the launcher's pid (= the ranks' parent) once the job is gone, whatever ended it -- bench.py removes what node_scene left under that pid there."""
    import os, signal, socket, subprocess, sys
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(script)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    env.update(backend_env or {})
    if timeout_s is None and os.environ.get("PBRT_AMD_LAUNCH_TIMEOUT_S"):
        timeout_s = float(os.environ["PBRT_AMD_LAUNCH_TIMEOUT_S"])
    # torch.distributed.run starts every rank in a session of its own (elastic/multiprocessing/subprocess_handler), so no process-group kill from outside reaches
    # them; the chain that does: this process dies -> the launcher gets SIGTERM (parent-death signal, below) -> it shuts its workers down; the launcher is
    # killed outright -> every rank gets SIGKILL (die_with_parent(), called by the ranks themselves).
    p = subprocess.Popen(cmd, env=env, start_new_session=True, preexec_fn=lambda: die_with_parent(signal.SIGTERM))

    def kill_group(sig=signal.SIGKILL):
        try:
            os.killpg(p.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass

    def stop_job():
        if p.poll() is None:
            kill_group(signal.SIGTERM)   # the launcher's handler ends the ranks
            try:
                p.wait(timeout=15)
            except subprocess.TimeoutExpired:
                pass
        kill_group()

    def on_signal(signum, frame):
        stop_job()
        raise SystemExit(128 + signum)

    saved = {}
    for sg in (signal.SIGTERM, signal.SIGINT):
        try:
            saved[sg] = signal.signal(sg, on_signal)
        except ValueError:
            pass
    try:
        try:
            return p.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            sys.stderr.write("launch_ranks: the %d-rank job did not finish within %.0f s -- killing its process group\n" % (n_ranks, timeout_s))
            return 124
    finally:
        stop_job()
        try:
            p.wait(timeout=10)
        except Exception:
            pass
        for sg, h in saved.items():
            signal.signal(sg, h)
        if after is not None:
            try:
                after(p.pid)
            except Exception:
                pass


class ShardedFrame:
    """One rank of the tile-sharded frame: the rank's device context renders its tiles (mi_tile_owner) into a film that lives in a torch
    tensor, and `step()` ends by STARTING the only exchange of the path -- FilmExchange: every rank's reachable pixels added into rank 0's film,
    RCCL send / recv over xGMI when the backend is "nccl" -- which then runs while the next step renders into the second film buffer (the two
    buffers alternate; a buffer is reused only after its exchange has finished).  world == 1: no torch, no exchange.
    `exchange="reduce"` keeps round 3's dense form (SUM-reduce of the whole film, synchronous) as the A/B partner.

        frame = ShardedFrame(ctx, scene, rank, world, local_rank, backend="nccl")
        frame.step(); frame.step(); frame.sync_all(); img = frame.root_film()
    """

    def __init__(self, ctx, scene, rank, world, device, backend="nccl", exchange="sparse", trace=None, timeout_s=None):
        """`device` = this rank's GPU ordinal (its local rank; every rank 0 only to exercise the N > 1 path on a one-GPU box, backend "gloo").  Every wait of the
        process group is bounded by `timeout_s` ($PBRT_AMD_PG_TIMEOUT_S, default 600: ranks of a fresh box import torch minutes apart): a rank that never arrives fails the job instead of hanging it."""
        self.ctx, self.scene, self.rank, self.world = ctx, scene, rank, world
        self.trace = trace or (lambda s: None)
        self.torch = self.dist = self.film = None
        self.films, self.pending, self.k, self.xchg, self.dense = [], [None, None], 0, None, exchange != "sparse"
        if world > 1:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            import datetime, os
            dev = device
            torch.cuda.set_device(dev)
            self.trace("torch imported, device %d selected" % dev)
            if not dist.is_initialized():
                if timeout_s is None:
                    timeout_s = float(os.environ.get("PBRT_AMD_PG_TIMEOUT_S", "600"))
                to = datetime.timedelta(seconds=timeout_s)
                if backend == "nccl":
                    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev), timeout=to)
                else:
                    if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
                        # one node: gloo's pairs go over loopback, not over whatever address the container's hostname happens to resolve to (it may not resolve at all)
                        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
                    dist.init_process_group(backend=backend, timeout=to)
            self.trace("init_process_group(%s) returned" % backend)
            n = scene.height * scene.width * 4
            self.films = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(1 if self.dense else 2)]
            self.film = self.films[0]
            if not self.dense:
                self.xchg = FilmExchange(scene, rank, world, torch.device("cuda", dev))
            ctx.film_bind(self.film.data_ptr())   # mi_render accumulates straight into the tensor the exchange reads
            self.trace("film buffers and exchange lists ready")

    def _finish(self, slot):
        if self.pending[slot] is not None:
            self.xchg.finish(self.pending[slot])   # the adds on rank 0 / the packing on the others: done before the buffer is cleared again
            self.pending[slot] = None

    def step(self, count_work=False, max_paths=0):
        if self.world > 1 and not self.dense:
            slot = self.k % 2
            self._finish(slot)                 # the exchange that last used this buffer (two steps ago)
            self.film = self.films[slot]
            self.ctx.film_bind(self.film.data_ptr())
        self.ctx.film_clear()
        self.ctx.render(rank=self.rank, world=self.world, count_work=count_work, max_paths=max_paths, sync=False)
        if self.world > 1:
            self.trace("step %d: render enqueued" % self.k)
            self.ctx.sync()                    # the ctx stream is not torch's current stream
            self.trace("step %d: rendered, exchange starts" % self.k)
            if self.dense:
                combine_films(self.film, dst=0)
                self.torch.cuda.synchronize()  # the reduction reads the film: done before the next step clears it
            else:
                self.pending[self.k % 2] = self.xchg.start(self.film)   # runs while the next step renders into the other buffer
            self.trace("step %d: exchange enqueued" % self.k)
        self.k += 1

    def sync_all(self):
        self.ctx.sync()
        if self.world > 1:
            if not self.dense:
                for slot in ((self.k) % 2, (self.k + 1) % 2):   # older exchange first
                    self._finish(slot)
            self.torch.cuda.synchronize()
            self.trace("sync_all: device idle, barrier")
            self.dist.barrier()
            self.torch.cuda.synchronize()
            self.trace("sync_all: barrier passed")

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
        if not self.dense:
            self._finish((self.k + 1) % 2)   # the last step's exchange
        return self.film.cpu().numpy().reshape(self.scene.height, self.scene.width, 4)

    def close(self):
        if self.world > 1 and self.dist.is_initialized():
            self.dist.barrier()
            self.dist.destroy_process_group()
