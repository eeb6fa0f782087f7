// ReadImage (core/imageio.cpp:60-79): the texture / radiance-map readers of the host.  Same contract as the
// reference: RGB float texels, row 0 = top of the image; 8-bit formats map through v / 255.f (imageio.cpp:243-249,
// 274-280), gamma is the caller's business (ImageTexture::convertIn).  PNG is decoded here with zlib's inflate
// (the reference goes through lodepng_decode24_file: any colour type / bit depth -> 8-bit RGB, 16-bit samples keep
// their high byte, alpha is dropped); TGA covers the uncompressed / RLE, true-colour / mono / colour-mapped variants
// the reference's targa.c reads.  OpenEXR is absent from this image (and from oracle/_ref): ".exr" goes through a reader written
// from the format specification (scan-line and tiled files, NONE / RLE / ZIPS / ZIP / PIZ / PXR24), pinned on one uncompressed file written by OpenEXR itself (tests/golden/openexr_written_16x16_rgba_half.exr), otherwise against files assembled from the specification (see DecodeEXR).
#include <zlib.h>

#include <cstdio>
#include <cstring>

#include "scene.h"

namespace pbrt_amd {
namespace {

bool ReadFile(const std::string &name, std::vector<uint8_t> *out) {
    FILE *f = std::fopen(name.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out->resize(n > 0 ? (size_t)n : 0);
    size_t got = n > 0 ? std::fread(out->data(), 1, (size_t)n, f) : 0;
    std::fclose(f);
    return got == out->size();
}
inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// ---- PNG (ISO/IEC 15948): non-interlaced, colour types 0/2/3/4/6, bit depths 1..16 -> 8-bit RGB
bool DecodePNG(const std::string &name, std::vector<uint8_t> *rgb, int *w, int *h) {
    std::vector<uint8_t> file;
    if (!ReadFile(name, &file) || file.size() < 33) { Error("Error reading PNG \"%s\": cannot open / too short", name.c_str()); return false; }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (std::memcmp(file.data(), sig, 8) != 0) { Error("Error reading PNG \"%s\": bad signature", name.c_str()); return false; }
    uint32_t width = 0, height = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    size_t pos = 8;
    while (pos + 12 <= file.size()) {
        uint32_t len = be32(&file[pos]);
        const uint8_t *type = &file[pos + 4], *data = &file[pos + 8];
        if (pos + 12 + len > file.size()) break;
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            width = be32(data); height = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
        } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + len;
    }
    int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!width || !height || !channels || interlace != 0 || (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16)) {
        Error("Error reading PNG \"%s\": unsupported layout (colour type %d, depth %d, interlace %d)", name.c_str(), ctype, depth, interlace);
        return false;
    }
    size_t bpp = (size_t)channels * depth;           // bits per pixel
    size_t stride = (width * bpp + 7) / 8, fbytes = (bpp + 7) / 8;
    std::vector<uint8_t> raw((stride + 1) * height);
    uLongf rawLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) {
        Error("Error reading PNG \"%s\": inflate failed", name.c_str());
        return false;
    }
    // undo the scanline filters (sect. 9)
    std::vector<uint8_t> img(stride * height);
    for (uint32_t y = 0; y < height; ++y) {
        const uint8_t *src = &raw[(stride + 1) * y + 1];
        uint8_t *dst = &img[stride * y];
        const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        int ft = raw[(stride + 1) * y];
        for (size_t i = 0; i < stride; ++i) {
            int a = i >= fbytes ? dst[i - fbytes] : 0, b = up ? up[i] : 0, c = (up && i >= fbytes) ? up[i - fbytes] : 0;
            int pred = 0;
            switch (ft) {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
            default: Error("Error reading PNG \"%s\": bad filter type", name.c_str()); return false;
            }
            dst[i] = (uint8_t)(src[i] + pred);
        }
    }
    rgb->resize((size_t)width * height * 3);
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x) {
            const uint8_t *row = &img[stride * y];
            auto sample = [&](int ch) -> int {   // channel ch of pixel x as an 8-bit value
                if (depth == 16) return row[((size_t)x * channels + ch) * 2];   // high byte
                if (depth == 8) return row[(size_t)x * channels + ch];
                size_t bit = ((size_t)x * channels + ch) * depth;
                int v = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
                return ctype == 3 ? v : (v * 255) / ((1 << depth) - 1);
            };
            uint8_t *o = &(*rgb)[((size_t)y * width + x) * 3];
            if (ctype == 3) {
                size_t idx = (size_t)sample(0) * 3;
                for (int c = 0; c < 3; ++c) o[c] = idx + c < plte.size() ? plte[idx + c] : 0;
            } else if (ctype == 0 || ctype == 4) o[0] = o[1] = o[2] = (uint8_t)sample(0);
            else for (int c = 0; c < 3; ++c) o[c] = (uint8_t)sample(c);
        }
    *w = (int)width; *h = (int)height;
    return true;
}

// ---- TGA (Truevision TGA 2.0): image types 1/2/3 and their RLE forms 9/10/11
bool DecodeTGA(const std::string &name, std::vector<Float> *out, int *w, int *h) {
    std::vector<uint8_t> f;
    if (!ReadFile(name, &f) || f.size() < 18) { Error("Unable to read from TGA file \"%s\"", name.c_str()); return false; }
    int idLen = f[0], cmapType = f[1], type = f[2];
    int cmapFirst = f[3] | (f[4] << 8), cmapLen = f[5] | (f[6] << 8), cmapBits = f[7];
    int width = f[12] | (f[13] << 8), height = f[14] | (f[15] << 8), bits = f[16], desc = f[17];
    bool rle = type >= 9;
    int base = rle ? type - 8 : type;
    if ((base != 1 && base != 2 && base != 3) || width <= 0 || height <= 0 || (bits != 8 && bits != 24 && bits != 32) ||
        (base == 1 && (cmapType != 1 || (cmapBits != 24 && cmapBits != 32) || bits != 8)) ||   // colour-mapped: 8-bit indices
        (base == 2 && bits != 24 && bits != 32) ||                                             // true colour: 3 or 4 bytes per pixel
        (base == 3 && bits != 8)) {                                                            // mono: one byte
        Error("Unable to read from TGA file \"%s\" (unsupported image type %d / %d bpp)", name.c_str(), type, bits);
        return false;
    }
    size_t pos = 18 + idLen;
    const uint8_t *cmap = &f[std::min(pos, f.size())];
    int cmapBytes = cmapType ? (cmapBits + 7) / 8 : 0;
    if (pos + (size_t)cmapLen * cmapBytes > f.size()) { Error("Unable to read from TGA file \"%s\" (truncated colour map)", name.c_str()); return false; }
    pos += (size_t)cmapLen * cmapBytes;
    int pb = bits / 8;
    std::vector<uint8_t> px((size_t)width * height * pb);
    if (!rle) {
        if (pos + px.size() > f.size()) { Error("Unable to read from TGA file \"%s\" (truncated)", name.c_str()); return false; }
        std::memcpy(px.data(), &f[pos], px.size());
    } else {
        size_t o = 0;
        while (o < px.size()) {
            if (pos >= f.size()) { Error("Unable to read from TGA file \"%s\" (truncated)", name.c_str()); return false; }
            int hd = f[pos++], cnt = (hd & 127) + 1;
            if (hd & 128) {
                if (pos + pb > f.size()) return false;
                for (int i = 0; i < cnt && o < px.size(); ++i, o += pb) std::memcpy(&px[o], &f[pos], pb);
                pos += pb;
            } else {
                size_t n = (size_t)cnt * pb;
                if (pos + n > f.size() || o + n > px.size()) return false;
                std::memcpy(&px[o], &f[pos], n);
                pos += n; o += n;
            }
        }
    }
    bool rightToLeft = desc & 0x10, topToBottom = desc & 0x20;
    out->resize((size_t)width * height * 3);
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            int sx = rightToLeft ? width - 1 - x : x, sy = topToBottom ? y : height - 1 - y;
            const uint8_t *src = &px[((size_t)sy * width + sx) * pb];
            Float *o = &(*out)[((size_t)y * width + x) * 3];
            if (base == 3) o[0] = o[1] = o[2] = *src / 255.f;   // mono (imageio.cpp:243-244)
            else {
                if (base == 1) {   // colour-mapped: unmap first (tga_color_unmap)
                    int idx = (int)*src - cmapFirst;
                    src = (idx >= 0 && idx < cmapLen) ? cmap + (size_t)idx * cmapBytes : cmap;
                }
                o[2] = src[0] / 255.f; o[1] = src[1] / 255.f; o[0] = src[2] / 255.f;   // BGR(A)
            }
        }
    *w = width; *h = height;
    return true;
}

// ---- OpenEXR 2 (single-part scan-line and tiled files -- level (0, 0) of the latter; compression NONE / RLE / ZIPS / ZIP / PIZ / PXR24; HALF / FLOAT / UINT channels R G B or Y).
// The reference reads EXR through the OpenEXR library (imageio.cpp:124-161, Imf::RgbaInputFile); that library is not in this image
// (nor in oracle/_ref, which therefore cannot read EXR either), so this reader is written from the file-format specification and is
// UNPINNED against the reference: tests/test_host.py checks it against files assembled independently in Python and against this
// host's own writer.  B44 / DWA compression and multi-part or deep files are reported as unsupported.
float HalfToFloat(uint16_t hbits) {
    uint32_t sign = (uint32_t)(hbits >> 15) << 31, e = (hbits >> 10) & 31, m = hbits & 1023;
    uint32_t out;
    if (e == 0) {
        if (m == 0) out = sign;
        else {   // subnormal half -> normal float
            int sh = 0;
            while (!(m & 1024)) { m <<= 1; ++sh; }
            out = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((m & 1023) << 13);
        }
    } else if (e == 31) out = sign | 0x7f800000u | (m << 13);
    else out = sign | ((e + 112) << 23) | (m << 13);
    float f;
    std::memcpy(&f, &out, 4);
    return f;
}
// ---- PIZ blocks (OpenEXR's ImfPizCompressor: a 16-bit value-range LUT, a 2-D Haar-like wavelet per channel, Huffman coding with one run-length
// symbol), restated from the published description of the format (OpenEXR "Technical Introduction" + the layout the library documents in
// ImfPizCompressor.cpp / ImfHuf.cpp / ImfWav.cpp's header comments).  Round 6: most published pbrt-v3 radiance maps are PIZ files.  UNPINNED against the
// library (none here, no PIZ file in this image): tests/test_host.py checks it against blocks encoded by a Python restatement of the
// ENCODING side (forward LUT, wenc14 / wenc16, code-table packing, run-length symbol) -- two independent restatements that must invert each other.
// Every read is bounds-checked; malformed data is an error, never an image.
struct PizChan { int nx, ny, size; size_t start; };   // size: 16-bit words per sample (HALF 1, FLOAT / UINT 2)
namespace piz {
constexpr int ENCSIZE = (1 << 16) + 1, DECBITS = 14, DECSIZE = 1 << DECBITS, DECMASK = DECSIZE - 1;
constexpr int SHORT_ZEROCODE_RUN = 59, LONG_ZEROCODE_RUN = 63, SHORTEST_LONG_RUN = 2 + LONG_ZEROCODE_RUN - SHORT_ZEROCODE_RUN;
struct Bits {   // most significant bit first
    const uint8_t *in, *end;
    uint64_t c = 0; int lc = 0; bool overrun = false;
    void byte() { uint8_t b = 0; if (in < end) b = *in; else overrun = true; ++in; c = (c << 8) | b; lc += 8; }
    uint32_t get(int n) { while (lc < n) byte(); lc -= n; return (uint32_t)((c >> lc) & ((1ull << n) - 1)); }
};
// code lengths (6 bits each; 59..62 = 2..5 zero lengths, 63 + 8 bits = 6..261 zero lengths) for symbols im..iM -> canonical codes, longest codes first
bool UnpackTable(Bits &b, int im, int iM, std::vector<uint64_t> &h) {
    h.assign(ENCSIZE, 0);
    for (; im <= iM; ++im) {
        uint32_t l = b.get(6);
        if (b.overrun) return false;
        h[im] = l;
        if (l == (uint32_t)LONG_ZEROCODE_RUN || l >= (uint32_t)SHORT_ZEROCODE_RUN) {
            int zerun = l == (uint32_t)LONG_ZEROCODE_RUN ? (int)b.get(8) + SHORTEST_LONG_RUN : (int)l - SHORT_ZEROCODE_RUN + 2;
            if (b.overrun || im + zerun > iM + 1) return false;
            while (zerun--) h[im++] = 0;
            --im;
        }
    }
    uint64_t n[59] = {0};
    for (int i = 0; i < ENCSIZE; ++i) n[h[i]] += 1;
    uint64_t c = 0;
    for (int i = 58; i > 0; --i) { uint64_t nc = (c + n[i]) >> 1; n[i] = c; c = nc; }
    for (int i = 0; i < ENCSIZE; ++i) { int l = (int)h[i]; if (l > 0) h[i] = (uint64_t)l | (n[l]++ << 6); }
    return true;
}
struct Dec { uint8_t len = 0; uint32_t lit = 0; int longs = -1; };   // primary table entry: a short code (len, symbol) or the list of long codes that share these 14 bits
bool Decode(const uint8_t *src, size_t nSrc, uint16_t *out, size_t nOut) {
    if (nSrc == 0) return nOut == 0;
    if (nSrc < 20) return false;
    auto u32 = [&](size_t at) { return (uint32_t)src[at] | ((uint32_t)src[at + 1] << 8) | ((uint32_t)src[at + 2] << 16) | ((uint32_t)src[at + 3] << 24); };
    const uint32_t im = u32(0), iM = u32(4), nBits = u32(12);   // (bytes 8..11: the table's packed length, 16..19: reserved)
    if (im >= (uint32_t)ENCSIZE || iM >= (uint32_t)ENCSIZE) return false;
    Bits tb{src + 20, src + nSrc};
    std::vector<uint64_t> h;
    if (!UnpackTable(tb, (int)im, (int)iM, h)) return false;
    const uint8_t *data = tb.in;   // the table ends at a byte boundary of its own reader: what it has buffered beyond is dropped
    if (data > src + nSrc || (uint64_t)nBits > 8ull * (uint64_t)(src + nSrc - data)) return false;
    std::vector<Dec> dec(DECSIZE);
    std::vector<std::vector<int>> longs;
    for (uint32_t s = im; s <= iM; ++s) {
        const uint64_t c = h[s] >> 6;
        const int l = (int)(h[s] & 63);
        if (l == 0) continue;
        if (c >> l) return false;
        if (l > DECBITS) {
            Dec &d = dec[c >> (l - DECBITS)];
            if (d.len) return false;
            if (d.longs < 0) { d.longs = (int)longs.size(); longs.emplace_back(); }
            longs[d.longs].push_back((int)s);
        } else {
            Dec *d = &dec[c << (DECBITS - l)];
            for (uint64_t i = 1ull << (DECBITS - l); i > 0; --i, ++d) {
                if (d->len || d->longs >= 0) return false;
                d->len = (uint8_t)l; d->lit = s;
            }
        }
    }
    Bits b{data, data + (nBits + 7) / 8};
    size_t o = 0;
    const uint32_t rlc = iM;   // the run-length symbol: followed by 8 bits = how often the previous value repeats
    auto emit = [&](uint32_t sym) -> bool {
        if (sym == rlc) {
            if (b.lc < 8) { if (b.in >= src + nSrc) return false; b.end = std::max(b.end, b.in + 1); b.byte(); }
            b.lc -= 8;
            uint32_t cs = (uint32_t)(b.c >> b.lc) & 0xff;
            if (o == 0 || o + cs > nOut) return false;
            const uint16_t v = out[o - 1];
            while (cs--) out[o++] = v;
            return true;
        }
        if (o >= nOut) return false;
        out[o++] = (uint16_t)sym;
        return true;
    };
    const uint8_t *ie = b.end;
    while (b.in < ie) {
        b.byte();
        while (b.lc >= DECBITS) {
            const Dec &d = dec[(b.c >> (b.lc - DECBITS)) & DECMASK];
            if (d.len) { b.lc -= d.len; if (!emit(d.lit)) return false; }
            else {
                if (d.longs < 0) return false;
                bool found = false;
                for (int s : longs[d.longs]) {
                    const int l = (int)(h[s] & 63);
                    while (b.lc < l && b.in < ie) b.byte();
                    if (b.lc >= l && (h[s] >> 6) == ((b.c >> (b.lc - l)) & ((1ull << l) - 1))) { b.lc -= l; if (!emit((uint32_t)s)) return false; found = true; break; }
                }
                if (!found) return false;
            }
        }
    }
    const int pad = (8 - (int)nBits) & 7;   // the bits of the last byte beyond nBits
    b.c >>= pad; b.lc -= pad;
    while (b.lc > 0) {
        const Dec &d = dec[(b.c << (DECBITS - b.lc)) & DECMASK];
        if (!d.len || d.len > b.lc) return false;
        b.lc -= d.len;
        if (!emit(d.lit)) return false;
    }
    return o == nOut;
}
// the inverse of the 2-D wavelet: one level = 2 x 2 butterflies, coarsest level first; 14-bit data (max value < 2^14) in plain signed arithmetic, 16-bit data modulo 2^16
inline void wdec14(uint16_t l, uint16_t hh, uint16_t &a, uint16_t &b) {
    const int ls = (int16_t)l, hi = (int16_t)hh;
    const int ai = ls + (hi & 1) + (hi >> 1);
    a = (uint16_t)(int16_t)ai; b = (uint16_t)(int16_t)(ai - hi);
}
inline void wdec16(uint16_t l, uint16_t hh, uint16_t &a, uint16_t &b) {
    const int m = l, d = hh;
    const int bb = (m - (d >> 1)) & 0xffff;
    const int aa = (d + bb - 0x8000) & 0xffff;
    b = (uint16_t)bb; a = (uint16_t)aa;
}
void WavDecode(uint16_t *in, int nx, int ox, int ny, int oy, uint16_t mx) {
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx;
    int p = 1, p2;
    while (p <= n) p <<= 1;
    p >>= 1; p2 = p; p >>= 1;
    auto dec = [&](uint16_t l, uint16_t hh, uint16_t &a, uint16_t &b) { if (w14) wdec14(l, hh, a, b); else wdec16(l, hh, a, b); };
    while (p >= 1) {
        uint16_t *py = in, *ey = in + (ptrdiff_t)oy * (ny - p2);
        const ptrdiff_t oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2, ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t *px = py, *ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
                dec(*px, *p10, i00, i10);
                dec(*p01, *p11, i01, i11);
                dec(i00, i01, *px, *p01);
                dec(i10, i11, *p10, *p11);
            }
            if (nx & p) {   // an odd column at this level: the vertical butterfly alone
                uint16_t *p10 = px + oy1;
                dec(*px, *p10, i00, *p10);
                *px = i00;
            }
        }
        if (ny & p) {   // an odd row: the horizontal butterflies alone
            uint16_t *px = py, *ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1;
                dec(*px, *p01, i00, *p01);
                *px = i00;
            }
        }
        p2 = p; p >>= 1;
    }
}
}  // namespace piz
// one PIZ block -> the block's scan lines in the uncompressed layout (per line: channel after channel); chans[] carries nx / ny / size of this block
bool PizBlock(const uint8_t *src, size_t n, std::vector<PizChan> &chans, int nLines, uint8_t *dst, size_t raw) {
    size_t total = 0;
    for (PizChan &c : chans) { c.start = total; total += (size_t)c.nx * c.ny * c.size; }
    if (total * 2 != raw || n < 4) return false;
    const uint32_t minNonZero = src[0] | (src[1] << 8), maxNonZero = src[2] | (src[3] << 8);
    constexpr uint32_t BITMAP = 8192;
    if (maxNonZero >= BITMAP) return false;
    std::vector<uint8_t> bitmap(BITMAP, 0);
    size_t p = 4;
    if (minNonZero <= maxNonZero) {
        const size_t nb = maxNonZero - minNonZero + 1;
        if (p + nb > n) return false;
        std::memcpy(&bitmap[minNonZero], src + p, nb);
        p += nb;
    }
    std::vector<uint16_t> lut(65536, 0);   // k-th value present in the block -> the value (0 is always present)
    uint32_t k = 0;
    for (uint32_t i = 0; i < 65536; ++i) if (i == 0 || (bitmap[i >> 3] & (1 << (i & 7)))) lut[k++] = (uint16_t)i;
    const uint16_t maxValue = (uint16_t)(k - 1);
    if (p + 4 > n) return false;
    const uint32_t length = (uint32_t)src[p] | ((uint32_t)src[p + 1] << 8) | ((uint32_t)src[p + 2] << 16) | ((uint32_t)src[p + 3] << 24);
    p += 4;
    if (length > n - p) return false;
    std::vector<uint16_t> tmp(total);
    if (!piz::Decode(src + p, length, tmp.data(), total)) return false;
    for (const PizChan &c : chans)
        for (int j = 0; j < c.size; ++j) piz::WavDecode(tmp.data() + c.start + j, c.nx, c.size, c.ny, c.nx * c.size, maxValue);
    for (uint16_t &v : tmp) v = lut[v];
    std::vector<size_t> at(chans.size());
    for (size_t c = 0; c < chans.size(); ++c) at[c] = chans[c].start;
    size_t o = 0;
    for (int y = 0; y < nLines; ++y)
        for (size_t c = 0; c < chans.size(); ++c) {
            const size_t cnt = (size_t)chans[c].nx * chans[c].size;
            for (size_t i = 0; i < cnt; ++i) { const uint16_t v = tmp[at[c] + i]; dst[o++] = (uint8_t)(v & 0xff); dst[o++] = (uint8_t)(v >> 8); }
            at[c] += cnt;
        }
    return o == raw;
}
// ---- PXR24 blocks (ImfPxr24Compressor): inflate, then per line and channel 2 (HALF) / 3 (FLOAT: the low 8 mantissa bits were dropped by the writer) / 4 (UINT) byte planes,
// most significant first, each sample a running sum of differences.  types[]: 0 UINT, 1 HALF, 2 FLOAT
bool Pxr24Block(const uint8_t *src, size_t n, const std::vector<int> &types, int width, int nLines, uint8_t *dst, size_t raw) {
    size_t need = 0;
    for (int t : types) need += (size_t)width * (t == 1 ? 2 : (t == 2 ? 3 : 4));
    need *= nLines;
    std::vector<uint8_t> tmp(need);
    uLongf got = (uLongf)need;
    if (uncompress(tmp.data(), &got, src, (uLong)n) != Z_OK || got != need) return false;
    const uint8_t *t = tmp.data();
    size_t o = 0;
    for (int y = 0; y < nLines; ++y)
        for (int ty : types) {
            const int planes = ty == 1 ? 2 : (ty == 2 ? 3 : 4);
            uint32_t pixel = 0;
            for (int x = 0; x < width; ++x) {
                uint32_t diff = 0;
                for (int k = 0; k < planes; ++k) diff = (diff << 8) | t[(size_t)k * width + x];
                if (ty == 2) diff <<= 8;
                pixel += diff;
                if (ty == 1) { dst[o++] = (uint8_t)(pixel & 0xff); dst[o++] = (uint8_t)((pixel >> 8) & 0xff); }
                else { dst[o++] = (uint8_t)pixel; dst[o++] = (uint8_t)(pixel >> 8); dst[o++] = (uint8_t)(pixel >> 16); dst[o++] = (uint8_t)(pixel >> 24); }
            }
            t += (size_t)planes * width;
        }
    return o == raw;
}
bool DecodeEXR(const std::string &name, std::vector<Float> *out, int *w, int *h) {
    std::vector<uint8_t> f;
    if (!ReadFile(name, &f) || f.size() < 16) { Error("Unable to read image file \"%s\"", name.c_str()); return false; }
    auto rd32 = [&](size_t p) { uint32_t v; std::memcpy(&v, &f[p], 4); return v; };
    if (rd32(0) != 20000630u) { Error("\"%s\" is not an OpenEXR file", name.c_str()); return false; }
    uint32_t version = rd32(4);
    if ((version & 0xff) != 2 || (version & 0x1800)) { Error("EXR file \"%s\": multi-part / deep files are not supported", name.c_str()); return false; }
    const bool tiled = (version & 0x200) != 0;
    int tileW = 0, tileH = 0;
    struct Chan { std::string name; int type; };
    std::vector<Chan> chans;
    int compression = -1, lineOrder = 0;
    int32_t dw[4] = {0, 0, -1, -1};
    size_t p = 8;
    // every string of the header must end inside the file: strnlen over what is left
    auto cstr = [&](size_t at, std::string *s) { size_t n = at < f.size() ? strnlen((const char *)&f[at], f.size() - at) : 0; if (at + n >= f.size()) return false; s->assign((const char *)&f[at], n); return true; };
    while (p < f.size() && f[p] != 0) {   // attributes: name\0 type\0 size data
        std::string an, at;
        if (!cstr(p, &an)) { Error("EXR file \"%s\": truncated header", name.c_str()); return false; }
        p += an.size() + 1;
        if (!cstr(p, &at)) { Error("EXR file \"%s\": truncated header", name.c_str()); return false; }
        p += at.size() + 1;
        if (p + 4 > f.size()) { Error("EXR file \"%s\": truncated header", name.c_str()); return false; }
        uint32_t sz = rd32(p); p += 4;
        if (p + sz > f.size()) { Error("EXR file \"%s\": truncated header", name.c_str()); return false; }
        if (an == "channels") {
            size_t q = p;
            while (q < p + sz && f[q] != 0) {
                Chan c;
                if (!cstr(q, &c.name) || q + c.name.size() + 1 + 16 > p + sz) { Error("EXR file \"%s\": truncated channel list", name.c_str()); return false; }
                q += c.name.size() + 1;
                c.type = (int)rd32(q);
                uint32_t xs = rd32(q + 8), ys = rd32(q + 12);
                q += 16;
                if (xs != 1 || ys != 1) { Error("EXR file \"%s\": subsampled channels are not supported", name.c_str()); return false; }
                chans.push_back(c);
            }
        } else if (an == "compression" && sz >= 1) compression = f[p];
        else if (an == "dataWindow" && sz >= 16) std::memcpy(dw, &f[p], 16);
        else if (an == "lineOrder" && sz >= 1) lineOrder = f[p];
        else if (an == "tiles" && sz >= 9) { tileW = (int)rd32(p); tileH = (int)rd32(p + 4); }   // (the level mode byte: level (0, 0) is stored first in every mode)
        p += sz;
    }
    ++p;   // end of header
    int width = dw[2] - dw[0] + 1, height = dw[3] - dw[1] + 1;
    if (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 28 || chans.empty()) { Error("EXR file \"%s\": bad header", name.c_str()); return false; }
    if (compression < 0 || compression > 5) { Error("EXR file \"%s\": compression method %d is not supported (NONE, RLE, ZIPS, ZIP, PIZ, PXR24 are)", name.c_str(), compression); return false; }
    const int linesPerBlock = compression == 4 ? 32 : (compression == 3 || compression == 5 ? 16 : 1);
    for (const Chan &c : chans) if (c.type < 0 || c.type > 2) { Error("EXR file \"%s\": channel \"%s\" has an unknown pixel type", name.c_str(), c.name.c_str()); return false; }
    size_t pixBytes = 0;
    for (const Chan &c : chans) pixBytes += c.type == 1 ? 2 : 4;
    int ir = -1, ig = -1, ib = -1, iy = -1;
    for (size_t c = 0; c < chans.size(); ++c) {
        if (chans[c].name == "R") ir = (int)c; else if (chans[c].name == "G") ig = (int)c; else if (chans[c].name == "B") ib = (int)c; else if (chans[c].name == "Y") iy = (int)c;
    }
    if ((ir < 0 || ig < 0 || ib < 0) && iy < 0) { Error("EXR file \"%s\": no R, G, B (or Y) channels", name.c_str()); return false; }
    out->assign((size_t)width * height * 3, 0.f);
    std::vector<uint8_t> buf, tmp;
    // one chunk of pixel data (a block of scan lines, or one tile): bw x nl pixels at (x0, y0) of the data window, stored line after line, each line channel after channel
    auto chunk = [&](const uint8_t *src, uint32_t dsz, int x0, int y0, int bw, int nl) -> bool {
        const size_t lineBytes = pixBytes * (size_t)bw, raw = lineBytes * nl;
        buf.resize(raw);
        if (compression == 0 || dsz >= raw) { if (dsz < raw) return false; std::memcpy(buf.data(), src, raw); }   // stored uncompressed when that is not larger
        else if (compression == 4) {
            std::vector<PizChan> pc(chans.size());
            for (size_t c = 0; c < chans.size(); ++c) pc[c] = PizChan{bw, nl, chans[c].type == 1 ? 1 : 2, 0};
            if (!PizBlock(src, dsz, pc, nl, buf.data(), raw)) { Error("EXR file \"%s\": bad PIZ data at line %d", name.c_str(), y0); return false; }
        } else if (compression == 5) {
            std::vector<int> types(chans.size());
            for (size_t c = 0; c < chans.size(); ++c) types[c] = chans[c].type;
            if (!Pxr24Block(src, dsz, types, bw, nl, buf.data(), raw)) { Error("EXR file \"%s\": bad PXR24 data at line %d", name.c_str(), y0); return false; }
        } else {
            tmp.resize(raw);
            if (compression == 1) {   // RLE: signed run lengths
                size_t o = 0, i = 0;
                while (i < dsz && o < raw) {
                    int c = (int8_t)src[i++];
                    if (c < 0) { size_t n = (size_t)(-c); if (i + n > dsz || o + n > raw) break; std::memcpy(&tmp[o], &src[i], n); i += n; o += n; }
                    else { size_t n = (size_t)c + 1; if (i >= dsz || o + n > raw) break; std::memset(&tmp[o], src[i++], n); o += n; }
                }
                if (o != raw) { Error("EXR file \"%s\": bad RLE data", name.c_str()); return false; }
            } else {
                uLongf got = (uLongf)raw;
                if (uncompress(tmp.data(), &got, src, dsz) != Z_OK || got != raw) { Error("EXR file \"%s\": inflate failed", name.c_str()); return false; }
            }
            for (size_t i = 1; i < raw; ++i) tmp[i] = (uint8_t)(tmp[i - 1] + tmp[i] - 128);   // undo the byte predictor ...
            size_t half = (raw + 1) / 2;
            for (size_t i = 0; i < raw; ++i) buf[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];   // ... and the even / odd byte split
        }
        std::vector<size_t> chanOff(chans.size());
        size_t o = 0;
        for (size_t c = 0; c < chans.size(); ++c) { chanOff[c] = o; o += (size_t)bw * (chans[c].type == 1 ? 2 : 4); }
        for (int l = 0; l < nl; ++l) {
            const uint8_t *line = &buf[lineBytes * l];
            auto value = [&](int c, int x) -> Float {
                const uint8_t *q = line + chanOff[c];
                if (chans[c].type == 1) { uint16_t hb; std::memcpy(&hb, q + 2 * (size_t)x, 2); return HalfToFloat(hb); }
                if (chans[c].type == 2) { float v; std::memcpy(&v, q + 4 * (size_t)x, 4); return v; }
                uint32_t u; std::memcpy(&u, q + 4 * (size_t)x, 4); return (Float)u;
            };
            Float *dst = &(*out)[((size_t)(y0 + l) * width + x0) * 3];
            for (int x = 0; x < bw; ++x) {
                if (ir >= 0 && ig >= 0 && ib >= 0) { dst[3 * x] = value(ir, x); dst[3 * x + 1] = value(ig, x); dst[3 * x + 2] = value(ib, x); }
                else dst[3 * x] = dst[3 * x + 1] = dst[3 * x + 2] = value(iy, x);
            }
        }
        return true;
    };
    if (tiled) {   // level (0, 0) of a tiled file: the first numX x numY entries of the offset table, whatever the level mode; Imf::RgbaInputFile reads the same level
        if (tileW <= 0 || tileH <= 0) { Error("EXR file \"%s\": tiled file without a valid \"tiles\" attribute", name.c_str()); return false; }
        const int ntx = (width + tileW - 1) / tileW, nty = (height + tileH - 1) / tileH;
        if (p + (size_t)ntx * nty * 8 > f.size()) { Error("EXR file \"%s\": truncated offset table", name.c_str()); return false; }
        for (int t = 0; t < ntx * nty; ++t) {
            uint64_t off;
            std::memcpy(&off, &f[p + (size_t)t * 8], 8);
            if (off > f.size() || off + 20 > f.size()) { Error("EXR file \"%s\": bad tile offset", name.c_str()); return false; }
            const int32_t tx = (int32_t)rd32(off), ty = (int32_t)rd32(off + 4), lx = (int32_t)rd32(off + 8), ly = (int32_t)rd32(off + 12);
            const uint32_t dsz = rd32(off + 16);
            if (lx != 0 || ly != 0 || tx < 0 || ty < 0 || tx >= ntx || ty >= nty || dsz > f.size() - (off + 20)) { Error("EXR file \"%s\": bad tile", name.c_str()); return false; }
            const int x0 = tx * tileW, y0 = ty * tileH;
            if (!chunk(&f[off + 20], dsz, x0, y0, std::min(tileW, width - x0), std::min(tileH, height - y0))) { Error("EXR file \"%s\": bad tile data", name.c_str()); return false; }
        }
    } else {
        const int nBlocks = (height + linesPerBlock - 1) / linesPerBlock;
        if (p + (size_t)nBlocks * 8 > f.size()) { Error("EXR file \"%s\": truncated offset table", name.c_str()); return false; }
        for (int b = 0; b < nBlocks; ++b) {
            uint64_t off;
            std::memcpy(&off, &f[p + (size_t)b * 8], 8);
            if (off > f.size() || off + 8 > f.size()) { Error("EXR file \"%s\": bad block offset", name.c_str()); return false; }
            const int32_t y0 = (int32_t)rd32(off) - dw[1];
            const uint32_t dsz = rd32(off + 4);
            if (dsz > f.size() - (off + 8) || y0 < 0 || y0 >= height || y0 % linesPerBlock != 0) { Error("EXR file \"%s\": bad block", name.c_str()); return false; }
            if (!chunk(&f[off + 8], dsz, 0, y0, width, std::min(linesPerBlock, height - y0))) { Error("EXR file \"%s\": bad block data", name.c_str()); return false; }
        }
    }
    (void)lineOrder;   // every block carries its own y coordinate
    *w = width; *h = height;
    return true;
}

bool HasExt(const std::string &n, const char *ext) {
    size_t l = std::strlen(ext);
    if (n.size() < l) return false;
    for (size_t i = 0; i < l; ++i) if (std::tolower((unsigned char)n[n.size() - l + i]) != ext[i]) return false;
    return true;
}

}  // namespace

bool HasExtension(const std::string &name, const char *ext) { return HasExt(name, ext); }

bool ReadImage(const std::string &name, std::vector<Float> *rgb, int *w, int *h) {   // imageio.cpp:60-79
    if (HasExt(name, ".pfm")) return ReadImagePFM(name, rgb, w, h);
    if (HasExt(name, ".png")) {
        std::vector<uint8_t> b;
        if (!DecodePNG(name, &b, w, h)) return false;
        rgb->resize(b.size());
        for (size_t i = 0; i < b.size(); ++i) (*rgb)[i] = b[i] / 255.f;
        return true;
    }
    if (HasExt(name, ".tga")) return DecodeTGA(name, rgb, w, h);
    if (HasExt(name, ".exr")) return DecodeEXR(name, rgb, w, h);
    Error("Unable to load image stored in format \"%s\" for filename \"%s\" (this host reads .pfm, .png, .tga and scan-line .exr).",
          name.rfind('.') != std::string::npos ? name.c_str() + name.rfind('.') : "(none)", name.c_str());
    return false;
}

}  // namespace pbrt_amd
