// ReadImage (core/imageio.cpp:60-79): the texture / radiance-map readers of the host.  Same contract as the
// reference: RGB float texels, row 0 = top of the image; 8-bit formats map through v / 255.f (imageio.cpp:243-249,
// 274-280), gamma is the caller's business (ImageTexture::convertIn).  PNG is decoded here with zlib's inflate
// (the reference goes through lodepng_decode24_file: any colour type / bit depth -> 8-bit RGB, 16-bit samples keep
// their high byte, alpha is dropped); TGA covers the uncompressed / RLE, true-colour / mono / colour-mapped variants
// the reference's targa.c reads.  OpenEXR is absent from this image (and from oracle/_ref), so ".exr" textures are an
// Error exactly like an unreadable file.
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
        (base == 1 && (cmapType != 1 || (cmapBits != 24 && cmapBits != 32)))) {
        Error("Unable to read from TGA file \"%s\" (unsupported image type %d / %d bpp)", name.c_str(), type, bits);
        return false;
    }
    size_t pos = 18 + idLen;
    const uint8_t *cmap = &f[std::min(pos, f.size())];
    int cmapBytes = cmapType ? (cmapBits + 7) / 8 : 0;
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
    Error("Unable to load image stored in format \"%s\" for filename \"%s\" (this host reads .pfm, .png and .tga; OpenEXR is not in the image).",
          name.rfind('.') != std::string::npos ? name.c_str() + name.rfind('.') : "(none)", name.c_str());
    return false;
}

}  // namespace pbrt_amd
