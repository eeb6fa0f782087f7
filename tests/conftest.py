import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")
    config.addinivalue_line("markers", "launches_processes: starts child processes that use the GPU (runs after every in-process test)")
    config.addinivalue_line("markers", "multirank: starts a multi-rank job under torch.distributed.run (runs last)")


# ---- order of a session: stage-level tests against the reference's vectors first, scene fixtures next, tests that start GPU-using child
# processes after those, multi-rank jobs last -- `pytest -x` has then reached everything else before the most fragile tests run.
_STAGE_PREFIXES = ("test_sobol", "test_halton", "test_camera_", "test_device_", "test_closest_hit", "test_hot_nodes", "test_deep_stacks", "test_texture_evaluation")


def _group(item):
    if item.get_closest_marker("multirank"):
        return 3
    if item.get_closest_marker("launches_processes"):
        return 2
    if item.get_closest_marker("gpu") and item.name.startswith(_STAGE_PREFIXES):
        return 0
    return 1


@pytest.hookimpl(trylast=True)
def pytest_collection_modifyitems(session, config, items):
    items.sort(key=_group)   # stable: the files' own order within a group


def _gpu_session_guard():
    """Which library do the -m gpu tests run?  The in-tree HIP build for gfx950, on a visible HIP device, with the header's ABI -- or the session stops (VERDICT r5
    item 9).  $PBRT_AMD_DEVICE_LIB (kernel-variant A/B runs, the x86 emulator of tools/hostemu) is refused unless PBRT_AMD_TEST_EMULATOR=1 says the run is a
    developer's emulator run, which is then labelled as such.  Returns the text of the banner."""
    import ctypes, hashlib, importlib, re
    emu = os.environ.get("PBRT_AMD_TEST_EMULATOR") == "1"
    if os.environ.get("PBRT_AMD_DEVICE_LIB") and not emu:
        pytest.exit("PBRT_AMD_DEVICE_LIB=%s is set: the -m gpu tests only run the in-tree lib/libpbrt_amd.so (PBRT_AMD_TEST_EMULATOR=1 labels an emulator run)"
                    % os.environ["PBRT_AMD_DEVICE_LIB"], returncode=3)
    pa = importlib.import_module("pbrt-v3-distributed_amd")
    lib = os.path.realpath(pa.DEVICE_LIB)
    if not os.path.exists(lib):
        pytest.exit("%s missing: the HIP extension was not built -- there is no fallback" % lib, returncode=3)
    blob = open(lib, "rb").read()
    build_id = hashlib.sha256(blob).hexdigest()[:16]
    if emu:
        return "[gpu tests] EMULATOR RUN (not a GPU result): %s sha256 %s" % (lib, build_id)
    if lib != os.path.realpath(os.path.join(ROOT, "pbrt-v3-distributed_amd", "lib", "libpbrt_amd.so")) or b"gfx950" not in blob:
        pytest.exit("%s is not the in-tree gfx950 build of the device library" % lib, returncode=3)
    want = int(re.search(r"#define MI_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "pbrt_amd.h")).read()).group(1))
    L = pa.device_lib()
    if L.mi_abi_version() != want:
        pytest.exit("libpbrt_amd.so speaks ABI %d, include/pbrt_amd.h declares %d" % (L.mi_abi_version(), want), returncode=3)
    try:
        hip = ctypes.CDLL("libamdhip64.so")
    except OSError:
        hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
    n = ctypes.c_int(0)
    rc = hip.hipGetDeviceCount(ctypes.byref(n))
    if rc != 0 or n.value < 1:
        pytest.exit("no HIP device visible (hipGetDeviceCount: rc %d, %d devices): the -m gpu tests need a real MI355X" % (rc, n.value), returncode=3)
    ctx = ctypes.c_void_p()
    if L.mi_ctx_create(0, None, ctypes.byref(ctx)) != 0:
        pytest.exit("mi_ctx_create(0) failed: %s" % L.mi_last_error().decode(), returncode=3)
    L.mi_ctx_destroy(ctx)
    return "[gpu tests] device library %s sha256 %s, ABI %d, %d HIP device(s)" % (lib, build_id, want, n.value)


# ---- the multi-rank tests need torch (torch.distributed under bench.py --gpus 2).  Its first import and first device tensor on a fresh box page in gigabytes of the
# image -- minutes, and very variable -- which is what round 5's driver run died of.  When such tests are selected the warm-up starts NOW, in a detached process, and
# proceeds while the two hundred in-process tests (which never touch torch) run; tests/test_zz_multirank_gpu.py waits for it before any job's limit starts counting.
TORCH_WARMUP = None
TORCH_WARMUP_CODE = ("import torch, torch.distributed as dist, torch.distributed.run; t = torch.zeros(1 << 20, device='cuda'); t += 1; i = torch.arange(8, device='cuda'); "
                     "t.view(-1, 4).index_select(0, i); t.view(-1, 4).index_add_(0, i, t.view(-1, 4)[:8].clone()); torch.cuda.synchronize(); print(float(t.sum().cpu()))")


def pytest_collection_finish(session):
    global TORCH_WARMUP
    if not session.config.option.collectonly and any(it.get_closest_marker("multirank") and it.get_closest_marker("gpu") for it in session.items):
        import subprocess
        TORCH_WARMUP = subprocess.Popen([sys.executable, "-c", TORCH_WARMUP_CODE], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
    if not session.config.option.collectonly and any(it.get_closest_marker("gpu") for it in session.items):
        banner = _gpu_session_guard()
        tr = session.config.pluginmanager.get_plugin("terminalreporter")
        if tr:
            tr.write_line(banner)
        else:
            print(banner)


@pytest.fixture(scope="session")
def built():
    """Make sure the host library and the oracle are built (CPU-only artefacts)."""
    import __graft_entry__ as ge
    ge.build(device=False)
    return True
