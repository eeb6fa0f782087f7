#!/usr/bin/env python3
"""Generate the Sobol' generator matrices used by the hot path from first principles.

What the reference tabulates in core/sobolmatrices.cpp:69 (SobolMatrices32, 1024 dims x 52
columns), :26700 (VdCSobolMatrices) and :26826 (VdCSobolMatricesInv) is *derived data*:
  * SobolMatrices32 = Joe & Kuo (2008) direction numbers ("new-joe-kuo-6.21201"), column j of
    dimension d being the 32-bit direction number v_{d,j} (j >= 32 columns shifted out).
    The primitive polynomials / initial numbers come from scipy's bundled copy of that
    public table (scipy/stats/_sobol_direction_numbers.npz); the recurrence is implemented here.
  * VdCSobolMatrices[m-1][c]   = the (x,y) pixel bits (2^m x 2^m grid over dims 0,1) contributed
    by index bit 2m+c;  VdCSobolMatricesInv[m-1] = GF(2) inverse of the map from the low 2m
    index bits to the pixel bits  (lowdiscrepancy.h:229-249 consumes both).
Output: pbrt-v3-distributed_amd/csrc/sobol_tables.inc  (checked in; regenerate with this script).
With --verify REF the three tables are compared word-for-word with the reference .cpp.
"""
import os, re, sys, numpy as np

NDIM, NCOL, NRES = 1024, 52, 26   # NRES rows of the VdC tables (m = 1..26, 2m <= 52)

def direction_numbers():
    import scipy.stats
    z = np.load(os.path.join(os.path.dirname(scipy.stats.__file__), "_sobol_direction_numbers.npz"))
    poly, vinit = z["poly"], z["vinit"]
    M = np.zeros((NDIM, NCOL), dtype=np.uint64)
    for j in range(NCOL):                      # dimension 0: van der Corput (identity, bit-reversed)
        M[0, j] = (1 << (31 - j)) if j < 32 else 0
    for d in range(1, NDIM):
        p = int(poly[d]); s = p.bit_length() - 1          # degree
        a = [(p >> (s - k)) & 1 for k in range(1, s)]     # interior coefficients a_1..a_{s-1}
        m = [int(vinit[d, k]) for k in range(s)]
        for k in range(s, NCOL):
            v = m[k - s] ^ (m[k - s] << s)
            for i in range(1, s):
                if a[i - 1]:
                    v ^= m[k - i] << i
            m.append(v)
        for j in range(NCOL):                  # v_j = m_j / 2^(j+1) as a 32-bit fixed-point fraction
            M[d, j] = (m[j] << (31 - j)) & 0xFFFFFFFF if j <= 31 else (m[j] >> (j - 31)) & 0xFFFFFFFF
    return M

def gf2_inverse(cols, n):
    """cols[j] = image (n-bit int) of unit vector j; returns columns of the inverse map."""
    A = [[(cols[j] >> i) & 1 for j in range(n)] for i in range(n)]      # A[i][j]
    I = [[int(i == j) for j in range(n)] for i in range(n)]
    for c in range(n):
        piv = next(r for r in range(c, n) if A[r][c])
        A[c], A[piv] = A[piv], A[c]; I[c], I[piv] = I[piv], I[c]
        for r in range(n):
            if r != c and A[r][c]:
                A[r] = [x ^ y for x, y in zip(A[r], A[c])]
                I[r] = [x ^ y for x, y in zip(I[r], I[c])]
    return [sum(I[i][j] << i for i in range(n)) for j in range(n)]

def vdc_tables(M):
    V = np.zeros((NRES, NCOL), dtype=np.uint64); Vi = np.zeros((NRES, NCOL), dtype=np.uint64)
    for m in range(1, NRES + 1):
        def pix(j):   # pixel bits (x << m | y) produced by index bit j
            x = int(M[0, j]) >> (32 - m) if j < NCOL else 0
            y = int(M[1, j]) >> (32 - m) if j < NCOL else 0
            return (x << m) | y
        for c in range(NCOL):
            V[m - 1, c] = pix(2 * m + c) if 2 * m + c < NCOL else 0
        inv = gf2_inverse([pix(j) for j in range(2 * m)], 2 * m)
        for c in range(2 * m):
            Vi[m - 1, c] = inv[c]
    return V, Vi

def parse_ref(path):
    src = open(path).read()
    a = src.index("SobolMatrices32["); b = src.index("SobolMatrices64[")
    r32 = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", src[a:b])]
    a = src.index("VdCSobolMatrices["); b = src.index("VdCSobolMatricesInv[")
    def rows(txt):   # rows carry only their non-zero-padded prefix; the rest is zero-initialised
        out = {}
        for m, body in re.findall(r"\{// m = (\d+)([^}]*)\}", txt):
            out[int(m)] = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", body)]
        return out
    return r32, rows(src[a:b]), rows(src[b:])

def main():
    M = direction_numbers(); V, Vi = vdc_tables(M)
    if len(sys.argv) > 2 and sys.argv[1] == "--verify":
        r32, rv, rvi = parse_ref(sys.argv[2])
        assert len(r32) == NDIM * NCOL, len(r32)
        assert np.array_equal(np.array(r32, dtype=np.uint64), M.reshape(-1)), "SobolMatrices32 mismatch"
        for name, ref, mine in (("VdC", rv, V), ("VdCInv", rvi, Vi)):
            for m, vals in sorted(ref.items()):
                got = [int(x) for x in mine[m - 1]]
                assert got[:len(vals)] == vals and not any(got[len(vals):]), (name, m, vals[:6], got[:6])
        print("verify OK: SobolMatrices32 and %d/%d VdC rows identical to the reference" % (len(rv), len(rvi)))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pbrt-v3-distributed_amd", "csrc", "sobol_tables.inc")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_sobol_tables.py -- do not edit.\n")
        f.write("// Joe-Kuo direction numbers -> 32-bit generator matrices [%d dims][%d cols] and the\n" % (NDIM, NCOL))
        f.write("// pixel<->index maps for dims (0,1) consumed by SobolIntervalToIndex.\n")
        f.write("#define PBRT_AMD_SOBOL_NDIM %d\n#define PBRT_AMD_SOBOL_NCOL %d\n#define PBRT_AMD_SOBOL_NRES %d\n" % (NDIM, NCOL, NRES))
        f.write("static const uint32_t kSobolMatrices32[%d] = {\n" % (NDIM * NCOL))
        flat = M.reshape(-1)
        for i in range(0, len(flat), 13):
            f.write(",".join("0x%xu" % int(x) for x in flat[i:i + 13]) + ",\n")
        f.write("};\n")
        for name, T in (("kVdCSobolMatrices", V), ("kVdCSobolMatricesInv", Vi)):
            f.write("static const uint64_t %s[%d][%d] = {\n" % (name, NRES, NCOL))
            for m in range(NRES):
                f.write("{" + ",".join("0x%xull" % int(x) for x in T[m]) + "},\n")
            f.write("};\n")
    print("wrote", os.path.normpath(out))

if __name__ == "__main__":
    main()
