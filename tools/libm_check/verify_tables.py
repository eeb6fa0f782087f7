"""Check that every table / constant of csrc/pt_libm.h is present, byte for byte, in the installed glibc's libm.so.6
(test infrastructure; run by tests/test_libm.py).  The header's numbers were written from glibc's published sources
(sysdeps/ieee754/flt-32/{s_sincosf_data.c,e_exp2f_data.c,e_logf_data.c,e_acosf.c,s_atanf.c,e_atan2f.c}); this script is the
"tables read from the installed libm" step: it parses the header and looks each group up in the binary's read-only data.

    python tools/libm_check/verify_tables.py [path/to/libm.so.6]  -> prints one line per group, exit 1 if one is missing
"""
import os
import re
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "..", "pbrt-v3-distributed_amd", "csrc", "pt_libm.h")
LIBM_CANDIDATES = ["/lib/x86_64-linux-gnu/libm.so.6", "/usr/lib/x86_64-linux-gnu/libm.so.6", "/lib64/libm.so.6"]


def _hexf(tok):
    return float.fromhex(tok)


def groups_from_header(text):
    """-> list of (name, bytes): consecutive runs of the binary's data that the header must reproduce"""
    g = []
    m = re.search(r"pt_lm_inv_pio4\[24\] = \{(.*?)\};", text, re.S)
    g.append(("__inv_pio4[24]", struct.pack("<24I", *[int(t, 16) for t in re.findall(r"0x[0-9a-fA-F]+", m.group(1))])))
    m = re.search(r"pt_lm_exp2f_tab\[32\] = \{(.*?)\};", text, re.S)
    g.append(("__exp2f_data.tab[32]", struct.pack("<32Q", *[int(t, 16) for t in re.findall(r"0x[0-9a-fA-F]+", m.group(1))])))
    m = re.search(r"pt_lm_logf_tab\[32\] = \{.*?\n(.*?)\};", text, re.S)
    vals = [_hexf(t) for t in re.findall(r"-?0x[0-9a-fA-F.]+p[+-]?\d+", m.group(1))]
    assert len(vals) == 32, len(vals)
    g.append(("__logf_data.tab[16]", struct.pack("<32d", *vals)))

    def dconsts(fn, names):
        body = text[text.index(fn):]
        out = []
        for n in names:
            mm = re.search(r"\b%s = (-?0x[0-9a-fA-F.]+p[+-]?\d+)" % re.escape(n), body)
            out.append(_hexf(mm.group(1)))
        return out
    # sincosf table 0 in the binary's order: hpi_inv (2^24-scaled), hpi, c0, c1, s1, c2, s2, c3, s3, c4
    s = dconsts("pt_lm_sin_poly", ["s1", "s2", "s3"])
    c = dconsts("pt_lm_cos_poly", ["c0", "c1", "c2", "c3", "c4"])
    g.append(("__sincosf_table[0] (hpi_inv, hpi, polynomials)",
              struct.pack("<10d", float.fromhex("0x1.45F306DC9C883p+23"), float.fromhex("0x1.921FB54442D18p0"), c[0], c[1], s[0], c[2], s[1], c[3], s[2], c[4])))
    for need in ("0x1.45F306DC9C883p+23", "0x1.921FB54442D18p0", "0x1.921FB54442D18p-62"):
        assert need in text, need
    e = dconsts("pt_expf", ["C0", "C1", "C2"])
    g.append(("__exp2f_data: shift, invln2_scaled, poly_scaled", struct.pack("<5d", float.fromhex("0x1.8p+52"), float.fromhex("0x1.71547652b82fep+5"), *e)))
    for need in ("InvLn2N = 0x1.71547652b82fep+5", "SHIFT = 0x1.8p+52"):
        assert need in text, need
    l = dconsts("pt_logf", ["Ln2", "A0", "A1", "A2"])
    g.append(("__logf_data: ln2, poly", struct.pack("<4d", *l)))

    def fconsts(fn, names):
        body = text[text.index(fn):]
        return [int(re.search(r"\b%s = pt_lm_asf32\((0x[0-9a-fA-F]+)\)" % re.escape(n), body).group(1), 16) for n in names]
    # e_acosf.c's constants as gcc laid them out (descending polynomial order, signs folded into sub instructions)
    a = fconsts("PT_DEV float pt_acosf", ["pS5", "pS4", "pS3", "pS2", "pS1", "pS0", "qS4", "qS3", "qS2", "qS1"])
    g.append(("acosf pS5..pS0, qS4..qS1 (magnitudes)", struct.pack("<10I", *[v & 0x7fffffff for v in a])))
    t = fconsts("PT_DEV float pt_atanf", ["aT10", "aT8", "aT6", "aT4", "aT2", "aT9", "aT7", "aT5", "aT3", "aT1"])
    g.append(("atanf aT10, aT8 .. aT2 | aT9 .. aT1 (as laid out)", struct.pack("<10I", t[0], t[1], t[2], t[3], t[4], t[5], t[6] & 0x7fffffff, t[7] & 0x7fffffff, t[8] & 0x7fffffff, t[9] & 0x7fffffff)))
    g.append(("atanf atanlo[2], atanhi[2], atanlo[0], atanhi[0]", struct.pack("<4I", 0x33140fb4, 0x3f7b985e, 0x31ac3769, 0x3eed6338)))
    for need in ("0x3f490fda", "0x33222168", "0x3fc90fda", "0x33a22168", "0x33140fb4", "0x3f7b985e", "0x31ac3769", "0x3eed6338"):
        assert need in text, need
    p = fconsts("PT_DEV float pt_atan2f", ["pi_o_4", "pi_o_2", "pi", "pi_lo"])
    g.append(("atan2f pi_o_4 .. pi, tiny", struct.pack("<2I", p[0], 0x80000000) + struct.pack("<2I", p[2], 0x0da24260)))
    assert p[1] == 0x3fc90fdb and p[3] == 0xb3bbbd2e
    return g


def main(argv):
    path = argv[1] if len(argv) > 1 else next((p for p in LIBM_CANDIDATES if os.path.exists(p)), None)
    if not path:
        print("no libm.so.6 found")
        return 2
    blob = open(path, "rb").read()
    text = open(HEADER).read()
    bad = 0
    for name, pat in groups_from_header(text):
        off = blob.find(pat)
        print("%-60s %s" % (name, "found at 0x%x" % off if off >= 0 else "MISSING"))
        bad += off < 0
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
