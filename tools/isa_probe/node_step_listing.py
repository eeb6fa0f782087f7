#!/usr/bin/env python3
"""ISA of the interior-node phase and the triangle phase of the headline traversal kernel k_trace<0, false, false, false, false, true, false>
(closest hit, all-triangle scene, quantised nodes, 768-thread shape), read off the PRODUCT source compiled for gfx950:

    python tools/isa_probe/node_step_listing.py [old.s]      (old.s: an earlier build's assembly of the same kernel, for the before / after table)

Instruction classes: VALU by the two issue classes tools/valu_probe/issue_probe measured on the MI355X (2.3 and 4.1 cycles per wave64 instruction per SIMD),
transcendental, scalar ALU, branches, LDS, vector memory, s_waitcnt / s_nop.  Regions are cut at the landmarks of the loop (first hot-node LDS read, the
triangle record loads, the leaf bookkeeping's v_bfe_u32); "executed" = the regions a scheduling round runs through when no lane is deep in its stack."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KERNEL = "_Z7k_traceILi0ELb0ELb0ELb0ELb0ELb1ELb0EEv8DevScene9PathStatej"
FAST = {"v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_f32", "v_sub_f32", "v_subrev_f32",
        "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_not_b32"}


def kernel_lines(path):
    out, on = [], False
    for l in open(path):
        if l.startswith(KERNEL + ":"):
            on = True
        if on:
            out.append(l.rstrip("\n"))
            if l.startswith(".Lfunc_end"):
                break
    return out


def classify(lines):
    c = dict(valu_fast=0, valu_slow=0, trans=0, salu=0, branch=0, lds=0, vmem=0, wait_nop=0)
    for l in lines:
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
            continue
        op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", t[0])
        if op.startswith("s_cbranch") or op == "s_branch": c["branch"] += 1
        elif op in ("s_waitcnt", "s_nop"): c["wait_nop"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith(("global_", "flat_", "buffer_", "scratch_")): c["vmem"] += 1
        elif op in ("v_rcp_f32", "v_sqrt_f32", "v_rsq_f32"): c["trans"] += 1
        elif op in FAST: c["valu_fast"] += 1
        elif op.startswith("v_"): c["valu_slow"] += 1
    c["valu"] = c["valu_fast"] + c["valu_slow"] + c["trans"]
    c["all"] = sum(v for k, v in c.items() if k != "valu")
    return c


def regions(L):
    """(start, end) of the innermost interior-node loop (every block whose comment names the loop header that precedes the first hot-node LDS read) and
    (start, end) of the triangle phase (the block with the triangle record loads up to the end of the leaf bookkeeping)"""
    first_lds = next(i for i, l in enumerate(L) if "ds_read_b128" in l)
    hdr = max(i for i, l in enumerate(L[:first_lds]) if l.startswith(".LBB") and "Parent Loop" in l)
    name = L[hdr].split(":")[0].lstrip(".L")
    members = [i for i, l in enumerate(L) if ("Header=%s " % name) in l or ("Header=%s\t" % name) in l or l.rstrip().endswith("Header=%s Depth=3" % name)]
    a, b = min(members + [hdr]), max(members)
    b = next(i for i, l in enumerate(L) if i > b and l.startswith(".LBB"))   # to the end of the last member block
    loads = [i for i, l in enumerate(L) if "global_load_dwordx4" in l]
    tri_first = loads[-3]
    ts = max(i for i, l in enumerate(L[:tri_first]) if l.startswith(".LBB"))
    bfe = next(i for i, l in enumerate(L) if i > tri_first and "v_bfe_u32" in l)
    te = next(i for i, l in enumerate(L) if i > bfe + 25 and l.startswith(".LBB"))
    return a, b, ts, te


def main():
    tmp = os.path.join(tempfile.mkdtemp(), "dev.s")
    pkg = os.path.join(ROOT, "pbrt-v3-distributed_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(pkg, "csrc"), "--cuda-device-only", "-S", os.path.join(pkg, "csrc", "pbrt_amd.hip"), "-o", tmp],
                          stderr=subprocess.DEVNULL)
    builds = [("this build", kernel_lines(tmp))]
    if len(sys.argv) > 1:
        builds.append((os.path.basename(sys.argv[1]), [l.rstrip("\n") for l in open(sys.argv[1])]))
    for name, L in builds:
        a, b, ts, te = regions(L)
        print("== %s: %d lines of kernel" % (name, len(L)))
        print("  interior-node loop, all of it (static; includes the deep-stack / spill paths a round normally skips)", classify(L[a:b]))
        print("  triangle phase (static)                                                                          ", classify(L[ts:te]))
        mad = [i for i in range(a, b) if "v_mad_u32_u24" in L[i]]
        scc = [i for i in range(a, b) if "s_cbranch_scc" in L[i]]
        if mad and scc and scc[-1] < mad[0]:   # round 4's step: the deep-stack tail lies between the wave-uniform branch and the block of the fast tail
            fast0 = max(i for i in range(a, mad[0]) if L[i].startswith(".LBB"))
            print("    of which the deep-stack tail (skipped unless some lane is within 3 entries of its LDS part)   ", classify(L[scc[-1] + 1:fast0]))
            print("    => one round without deep lanes runs through                                                   ", classify(L[a:scc[-1] + 1] + L[fast0:b]))
    name, L = builds[0]
    a, b, ts, te = regions(L)
    print("\n== listing: node phase of this build (one scheduling round of the interior-node loop; the block after `s_cbranch_scc0` .. the unconditional ds_write_b32 triple is the deep-stack tail, skipped unless a lane is within 3 entries of its LDS part)")
    for l in L[a:b]:
        if l.strip() and not l.strip().startswith(";"):
            print(l.split(";")[0].rstrip() if not l.startswith(".LBB") else l)


if __name__ == "__main__":
    main()
