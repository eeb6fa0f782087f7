"""CPU: the C-ABI library loads and exports exactly the symbols include/pbrt_amd.h declares (no compute calls)."""
import os
import re
import subprocess

import oracle_lib as ol

pa = ol.pa


def _header_symbols():
    text = open(os.path.join(ol.ROOT, "include", "pbrt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"static inline[^\n]*\{[^\n]*\}\n", "", text)                  # header-only helpers (mi_tile_owner / mi_tile_skew) are not exported symbols:
    text = re.sub(r"static inline[^;{]*\{.*?\n\}", "", text, flags=re.S)   # one-line bodies first, then bodies that end with a brace in column 0
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text)))


def test_header_matches_binding_list():
    assert _header_symbols() == sorted(pa.DEVICE_SYMBOLS)


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build(device=True)
    assert os.path.exists(pa.DEVICE_LIB)
    out = subprocess.run(["nm", "-D", "--defined-only", pa.DEVICE_LIB], capture_output=True, text=True, check=True).stdout
    exported = set(l.split()[-1] for l in out.splitlines() if l.strip())
    missing = [s for s in _header_symbols() if s not in exported]
    assert not missing, missing
    L = pa.device_lib()          # loads without a GPU (libamdhip64 is present)
    assert L.mi_abi_version() == int(re.search(r"#define MI_ABI_VERSION (\d+)", open(os.path.join(ol.ROOT, "include", "pbrt_amd.h")).read()).group(1)) == 14


def test_no_cpu_fallback_without_gpu():
    import torch, pytest
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    sc = pa.Scene(os.path.join(ol.ROOT, "scenes", "cornell.pbrt"))
    try:
        pa.Context(sc)
    except RuntimeError as e:
        assert "no HIP device" in str(e) or "mi_ctx_create" in str(e)
    else:
        raise AssertionError("Context creation must fail loudly without a GPU")
