"""csrc/pt_libm.h == the glibc the reference links against.

CPU (`-m "not gpu"`): tools/libm_check/check.cpp compiles the device header for the host and compares it with the host's libm over ALL 2^32
inputs of sinf / cosf / sincosf / expf / logf / acosf / atanf and over 10^9+ pairs for atan2f (about a minute on 8 cores); the header's tables
are looked up byte for byte in the installed libm.so.6.
GPU (`-m gpu`): the device build of the same header, through the C ABI's stage entry mi_libm_eval, against the GPU box's own glibc."""
import ctypes as C
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "libm_check")
BUILD = os.path.join(TOOL, "_build")


def _glibc_is_the_pinned_one():
    v = os.confstr("CS_GNU_LIBC_VERSION") if hasattr(os, "confstr") else ""
    return v.strip() == "glibc 2.35" and os.uname().machine == "x86_64" and "fma" in open("/proc/cpuinfo").read() and "avx2" in open("/proc/cpuinfo").read()


pinned = pytest.mark.skipif(not _glibc_is_the_pinned_one(), reason="pt_libm.h restates glibc 2.35's x86-64 FMA builds (the reference's libm in this image)")


def _build_checker():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "check")
    src = os.path.join(TOOL, "check.cpp")
    hdr = os.path.join(ROOT, "pbrt-v3-distributed_amd", "csrc", "pt_libm.h")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", src, "-o", exe, "-lpthread", "-lm"])
    return exe


@pinned
def test_tables_are_the_installed_libms():
    r = subprocess.run([sys.executable, os.path.join(TOOL, "verify_tables.py")], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("found at") == 10


@pinned
def test_exhaustive_against_host_glibc():
    """every float bit pattern of every one-argument routine; 10^9 random + 2.1e8 structured pairs for atan2f.  PT_LIBM_STRIDE=k thins the sweep."""
    exe = _build_checker()
    stride = os.environ.get("PT_LIBM_STRIDE", "1")
    r = subprocess.run([exe, "--stride", stride, "--pairs", "1000000000"], stdout=subprocess.PIPE, text=True, timeout=3000)
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert [x["routine"] for x in rows] == ["sinf", "cosf", "expf", "logf", "acosf", "atanf", "sincosf", "atan2f"], r.stdout
    for x in rows:
        assert x["mismatches"] == 0, x
        assert x["routine"] == "atan2f" or x["tested"] == (1 << 32) // int(stride) + (0 if (1 << 32) % int(stride) == 0 else 1) or int(stride) > 1
    assert r.returncode == 0


def _host_libm():
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "host_libm.so")
    src = os.path.join(TOOL, "host_libm.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O1", "-fno-builtin", "-shared", "-fPIC", src, "-o", so, "-lm"])
    L = C.CDLL(so)
    L.host_libm_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
    return L


def _host_eval(L, fn, a, b=None):
    out = np.zeros(len(a), np.float32)
    out2 = np.zeros(len(a), np.float32) if fn == 2 else None
    L.host_libm_eval(fn, a.ctypes.data, None if b is None else b.ctypes.data, len(a), out.ctypes.data, None if out2 is None else out2.ctypes.data)
    return out, out2


def _same(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


SPECIAL = np.array([0, 0x80000000, 1, 0x80000001, 0x007fffff, 0x00800000, 0x3f800000, 0xbf800000, 0x3f000000, 0xbf000000, 0x3effffff, 0x3f000001,
                    0x32800000, 0x32800001, 0x3f7fffff, 0x3f800001, 0x7f7fffff, 0xff7fffff, 0x7f800000, 0xff800000, 0x7fc00000, 0x7f800001,
                    0x3f490fdb, 0x3f3fffff, 0x3f400000, 0x397fffff, 0x39800000, 0x42efffff, 0x42f00000, 0x42b17217, 0x42b17218, 0xc2cff1b4, 0xc2cff1b5,
                    0xc2ce8ecf, 0xc2ce8ed0, 0x42afffff, 0x42b00000, 0x4bffffff, 0x4c000000, 0x30ffffff, 0x31000000, 0x3edfffff, 0x3ee00000, 0x3f2fffff,
                    0x3f300000, 0x3f97ffff, 0x3f980000, 0x401bffff, 0x401c0000, 0x3f330000, 0x3f32ffff], dtype=np.uint32)


def _inputs(rng, n):
    raw = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    return {
        "raw": raw,
        "angles": rng.uniform(-4 * np.pi, 4 * np.pi, n).astype(np.float32),                 # what the samplers hand over
        "unit": rng.uniform(-1, 1, n).astype(np.float32),                                    # acos of clamped cosines
        "open01": (1 - rng.random(n, dtype=np.float32)).astype(np.float32),                  # log(1 - u)
        "neg": (-rng.exponential(8.0, n)).astype(np.float32),                                # exp(-optical depth)
        "big": (rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(0, 12, n)).astype(np.float32),   # large-argument reduction
        "tiny": (rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-44, -3, n)).astype(np.float32),
        "special": SPECIAL.view(np.float32),
    }


@pytest.mark.gpu
@pinned
def test_device_libm_matches_host_glibc():
    """bit for bit (NaN == NaN) on 8 x 2^21 inputs per routine incl. subnormal results, both reduction paths of sin / cos, every branch
    boundary of the fdlibm routines; atan2f on raw pairs, comparable-magnitude pairs and the special-value grid"""
    sys.path.insert(0, ROOT)
    pa = importlib.import_module("pbrt-v3-distributed_amd")
    L = _host_libm()
    rng = np.random.default_rng(20260921)
    n = 1 << 21
    sets = _inputs(rng, n)
    report = {}
    for name in ("sinf", "cosf", "sincosf", "expf", "logf", "acosf", "atanf"):
        fn = pa.LIBM_FUNCS[name]
        bad = total = 0
        for key, a in sets.items():
            a = np.ascontiguousarray(a)
            want, want2 = _host_eval(L, fn, a)
            got = pa.libm_eval(name, a)
            if name == "sincosf":
                ok = _same(got[0], want) & _same(got[1], want2)
            else:
                ok = _same(got, want)
            bad += int((~ok).sum())
            total += len(a)
            assert ok.all(), (name, key, a[~ok][:4].view(np.uint32), (got[0] if name == "sincosf" else got)[~ok][:4].view(np.uint32), want[~ok][:4].view(np.uint32))
        report[name] = (total, bad)
    ys = [sets["raw"], sets["angles"], sets["unit"], np.repeat(SPECIAL.view(np.float32), len(SPECIAL))]
    xs = [np.roll(sets["raw"], 1), sets["unit"] * np.float32(3), sets["big"], np.tile(SPECIAL.view(np.float32), len(SPECIAL))]
    for y, x in zip(ys, xs):
        y = np.ascontiguousarray(y); x = np.ascontiguousarray(x)
        want, _ = _host_eval(L, 7, y, x)
        got = pa.libm_eval("atan2f", y, x)
        ok = _same(got, want)
        assert ok.all(), ("atan2f", y[~ok][:4].view(np.uint32), x[~ok][:4].view(np.uint32), got[~ok][:4].view(np.uint32), want[~ok][:4].view(np.uint32))
    print("device libm == host glibc:", report)
