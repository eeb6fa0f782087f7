// Film, reconstruction filters and image output (host).  Reference: core/film.{h,cpp},
// filters/*.cpp, core/imageio.cpp.  The device returns FilmTilePixel{rgb contribSum,
// filterWeightSum} per cropped pixel; MergeFilm + WriteImage below perform exactly the
// RGB->XYZ->RGB, divide-by-weight, clamp and scale steps of film.cpp:117-130,168-210
// (the XYZ round trip is not the identity in fp32).
#include <cstdio>

#include <zlib.h>

#include "scene.h"

namespace pbrt_amd {

// ------------------------------------------------------------------ filters
Float Filter::Evaluate(Float x, Float y) const {
    if (name == "box") return 1.;   // filters/box.cpp:41
    if (name == "gaussian") {       // filters/gaussian.h:62-64
        auto g = [&](Float d, Float expv) { return std::max((Float)0, Float(std::exp(-p0 * d * d) - expv)); };
        return g(x, expX) * g(y, expY);
    }
    if (name == "mitchell") {       // filters/mitchell.h:53-63
        Float B = p0, C = p1;
        auto m1 = [&](Float v) {
            v = std::abs(2 * v);
            if (v > 1)
                return ((-B - 6 * C) * v * v * v + (6 * B + 30 * C) * v * v + (-12 * B - 48 * C) * v + (8 * B + 24 * C)) *
                       (1.f / 6.f);
            return ((12 - 9 * B - 6 * C) * v * v * v + (-18 + 12 * B + 6 * C) * v * v + (6 - 2 * B)) * (1.f / 6.f);
        };
        return m1(x * (1 / rx)) * m1(y * (1 / ry));
    }
    if (name == "triangle")         // filters/triangle.cpp:41-44
        return std::max((Float)0, rx - std::abs(x)) * std::max((Float)0, ry - std::abs(y));
    if (name == "sinc") {           // filters/sinc.h:52-62
        Float tau = p0;
        auto sinc = [](Float v) -> Float {
            v = std::abs(v);
            if (v < 1e-5) return 1;
            return std::sin(kPi * v) / (kPi * v);
        };
        auto ws = [&](Float v, Float r) -> Float {
            v = std::abs(v);
            if (v > r) return 0;
            Float lanczos = sinc(v / tau);
            return sinc(v) * lanczos;
        };
        return ws(x, rx) * ws(y, ry);
    }
    return 1.;
}

std::unique_ptr<Filter> MakeFilter(const std::string &name, const ParamSet &ps) {   // api.cpp:842-861
    std::unique_ptr<Filter> f(new Filter);
    f->name = name;
    if (name == "box") {
        f->rx = ps.FindOneFloat("xwidth", 0.5f); f->ry = ps.FindOneFloat("ywidth", 0.5f);
    } else if (name == "gaussian") {
        f->rx = ps.FindOneFloat("xwidth", 2.f); f->ry = ps.FindOneFloat("ywidth", 2.f);
        f->p0 = ps.FindOneFloat("alpha", 2.f);
        f->expX = std::exp(-f->p0 * f->rx * f->rx); f->expY = std::exp(-f->p0 * f->ry * f->ry);
    } else if (name == "mitchell") {
        f->rx = ps.FindOneFloat("xwidth", 2.f); f->ry = ps.FindOneFloat("ywidth", 2.f);
        f->p0 = ps.FindOneFloat("B", 1.f / 3.f); f->p1 = ps.FindOneFloat("C", 1.f / 3.f);
    } else if (name == "sinc") {
        f->rx = ps.FindOneFloat("xwidth", 4.); f->ry = ps.FindOneFloat("ywidth", 4.);
        f->p0 = ps.FindOneFloat("tau", 3.f);
    } else if (name == "triangle") {
        f->rx = ps.FindOneFloat("xwidth", 2.f); f->ry = ps.FindOneFloat("ywidth", 2.f);
    } else {
        Error("Filter \"%s\" unknown.", name.c_str());
        return nullptr;
    }
    ps.ReportUnused();
    return f;
}

// ------------------------------------------------------------------ film
Film::Film(int xres, int yres, const Float crop[4], std::unique_ptr<Filter> filt, Float diag,
           const std::string &fn, Float sc, Float maxLum)
    : filter(std::move(filt)), filename(fn), diagonal(diag * .001), scale(sc), maxSampleLuminance(maxLum) {
    fullResolution[0] = xres; fullResolution[1] = yres;
    // film.cpp:55-60 (crop = {x0, x1, y0, y1})
    cropMin[0] = (int)std::ceil(xres * crop[0]); cropMin[1] = (int)std::ceil(yres * crop[2]);
    cropMax[0] = (int)std::ceil(xres * crop[1]); cropMax[1] = (int)std::ceil(yres * crop[3]);
    pixels.assign((size_t)std::max(0, (cropMax[0] - cropMin[0]) * (cropMax[1] - cropMin[1])), Pixel{{0, 0, 0}, 0});
    int offset = 0;   // film.cpp:69-77
    const int W = MI_FILTER_TABLE_WIDTH;
    for (int y = 0; y < W; ++y)
        for (int x = 0; x < W; ++x, ++offset) {
            Float px = (x + 0.5f) * filter->rx / W, py = (y + 0.5f) * filter->ry / W;
            filterTable[offset] = filter->Evaluate(px, py);
        }
}

void Film::GetSampleBounds(int mn[2], int mx[2]) const {   // film.cpp:80-86
    mn[0] = (int)std::floor((Float)cropMin[0] + 0.5f - filter->rx);
    mn[1] = (int)std::floor((Float)cropMin[1] + 0.5f - filter->ry);
    mx[0] = (int)std::ceil((Float)cropMax[0] - 0.5f + filter->rx);
    mx[1] = (int)std::ceil((Float)cropMax[1] - 0.5f + filter->ry);
}

void Film::Clear() { for (auto &p : pixels) p = Pixel{{0, 0, 0}, 0}; }

void Film::MergeFilm(const float *rgbw) {
    for (size_t i = 0; i < pixels.size(); ++i) {
        const float *c = rgbw + 4 * i;
        Float xyz[3];   // RGBToXYZ, spectrum.h:62-66
        xyz[0] = 0.412453f * c[0] + 0.357580f * c[1] + 0.180423f * c[2];
        xyz[1] = 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2];
        xyz[2] = 0.019334f * c[0] + 0.119193f * c[1] + 0.950227f * c[2];
        for (int k = 0; k < 3; ++k) pixels[i].xyz[k] += xyz[k];
        pixels[i].filterWeightSum += c[3];
    }
}

std::vector<Float> Film::FinalRGB() const {   // film.cpp:171-203 (no splats on this path)
    std::vector<Float> rgb(3 * pixels.size());
    for (size_t i = 0; i < pixels.size(); ++i) {
        const Float *xyz = pixels[i].xyz;
        Float *o = &rgb[3 * i];   // XYZToRGB, spectrum.h:56-60
        o[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
        o[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
        o[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
        Float w = pixels[i].filterWeightSum;
        if (w != 0) {
            Float invWt = (Float)1 / w;
            for (int k = 0; k < 3; ++k) o[k] = std::max((Float)0, o[k] * invWt);
        }
        for (int k = 0; k < 3; ++k) { o[k] += 0.f; o[k] *= scale; }   // splat term is 1 * XYZToRGB(0) = 0
    }
    return rgb;
}

void Film::WriteImage(Float) {
    std::vector<Float> rgb = FinalRGB();
    pbrt_amd::WriteImage(filename, rgb.data(), cropMin, cropMax, fullResolution);
}

extern std::string g_imageFileOverride;   // --outfile (api.cpp)
extern Float g_cropWindow[4];
extern bool g_quickRender;

Film *CreateFilm(const ParamSet &ps, std::unique_ptr<Filter> filter) {   // film.cpp:212-256
    std::string filename;
    if (g_imageFileOverride != "") {
        filename = g_imageFileOverride;
        std::string pf = ps.FindOneString("filename", "");
        if (pf != "")
            Warning("Output filename supplied on command line, \"%s\" is overriding filename provided in scene "
                    "description file, \"%s\".", filename.c_str(), pf.c_str());
    } else
        filename = ps.FindOneString("filename", "pbrt.exr");
    int xres = ps.FindOneInt("xresolution", 1280), yres = ps.FindOneInt("yresolution", 720);
    if (g_quickRender) { xres = std::max(1, xres / 4); yres = std::max(1, yres / 4); }
    Float crop[4];
    int cwi;
    const Float *cr = ps.FindFloat("cropwindow", &cwi);
    if (cr && cwi == 4) {
        crop[0] = Clamp(std::min(cr[0], cr[1]), 0.f, 1.f); crop[1] = Clamp(std::max(cr[0], cr[1]), 0.f, 1.f);
        crop[2] = Clamp(std::min(cr[2], cr[3]), 0.f, 1.f); crop[3] = Clamp(std::max(cr[2], cr[3]), 0.f, 1.f);
    } else {
        if (cr) Error("%d values supplied for \"cropwindow\". Expected 4.", cwi);
        for (int i = 0; i < 4; ++i) crop[i] = Clamp(g_cropWindow[i], 0, 1);
    }
    Float scale = ps.FindOneFloat("scale", 1.);
    Float diagonal = ps.FindOneFloat("diagonal", 35.);
    Float maxLum = ps.FindOneFloat("maxsampleluminance", kInfinity);
    return new Film(xres, yres, crop, std::move(filter), diagonal, filename, scale, maxLum);
}

// ------------------------------------------------------------------ image output
static bool WritePFM(const std::string &fn, const Float *rgb, int w, int h) {   // imageio.cpp:437-482
    FILE *fp = std::fopen(fn.c_str(), "wb");
    if (!fp) { Error("Unable to open output PFM file \"%s\"", fn.c_str()); return false; }
    std::fprintf(fp, "PF\n%d %d\n%f\n", w, h, -1.f);   // little endian host
    for (int y = h - 1; y >= 0; --y) std::fwrite(rgb + (size_t)y * w * 3, sizeof(float), (size_t)w * 3, fp);
    std::fclose(fp);
    return true;
}

bool ReadImagePFM(const std::string &fn, std::vector<Float> *rgb, int *w, int *h) {   // core/imageio.cpp:350-425
    FILE *fp = std::fopen(fn.c_str(), "rb");
    if (!fp) return false;
    char tag[8];
    float sc;
    if (std::fscanf(fp, "%7s %d %d %f", tag, w, h, &sc) != 4 || (std::string(tag) != "PF" && std::string(tag) != "Pf")) { std::fclose(fp); return false; }
    std::fgetc(fp);
    if (*w <= 0 || *h <= 0 || (int64_t)*w * *h > (int64_t)1 << 28) { std::fclose(fp); return false; }   // header values come from the file
    int nc = std::string(tag) == "PF" ? 3 : 1;
    std::vector<float> data((size_t)*w * *h * nc);
    for (int y = *h - 1; y >= 0; --y)   // flip in Y: P*M has the origin at the lower left
        if (std::fread(data.data() + (size_t)y * *w * nc, sizeof(float), (size_t)*w * nc, fp) != (size_t)*w * nc) { std::fclose(fp); return false; }
    std::fclose(fp);
    if (!(sc < 0.f))   // big-endian file on a little-endian host
        for (float &v : data) { unsigned char b[4]; std::memcpy(b, &v, 4); std::swap(b[0], b[3]); std::swap(b[1], b[2]); std::memcpy(&v, b, 4); }
    if (std::abs(sc) != 1.f) for (float &v : data) v *= std::abs(sc);
    rgb->resize((size_t)*w * *h * 3);
    for (size_t i = 0; i < (size_t)*w * *h; ++i)
        for (int c = 0; c < 3; ++c) (*rgb)[3 * i + c] = nc == 3 ? data[3 * i + c] : data[i];
    return true;
}

static uint16_t FloatToHalf(float f) {   // round-to-nearest-even binary32 -> binary16
    uint32_t x; std::memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t m = x & 0x7fffffu;
    if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0));
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        int shift = 14 - e;
        uint32_t h = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = (uint32_t)(e << 10) | (m >> 13), rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
    return (uint16_t)(sign | h);
}

// Minimal OpenEXR 2 scan-line writer: uncompressed, half RGBA -- what the reference asks
// OpenEXR for (Imf::RgbaOutputFile, WRITE_RGBA, imageio.cpp:164-189) minus compression.
static bool WriteEXR(const std::string &fn, const Float *rgb, int w, int h, int totalX, int totalY, int xOff, int yOff) {
    FILE *fp = std::fopen(fn.c_str(), "wb");
    if (!fp) { Error("Unable to open output EXR file \"%s\"", fn.c_str()); return false; }
    std::vector<uint8_t> hdr;
    auto put = [&](const void *p, size_t n) { hdr.insert(hdr.end(), (const uint8_t *)p, (const uint8_t *)p + n); };
    auto puts = [&](const char *s) { put(s, std::strlen(s) + 1); };
    auto puti = [&](int32_t v) { put(&v, 4); };
    auto putf = [&](float v) { put(&v, 4); };
    uint32_t magic = 20000630, version = 2;
    put(&magic, 4); put(&version, 4);
    puts("channels"); puts("chlist"); puti(4 * 18 + 1);
    for (const char *c : {"A", "B", "G", "R"}) { puts(c); puti(1 /*HALF*/); uint8_t z[4] = {0, 0, 0, 0}; put(z, 4); puti(1); puti(1); }
    { uint8_t z = 0; put(&z, 1); }
    puts("compression"); puts("compression"); puti(1); { uint8_t z = 0; put(&z, 1); }
    puts("dataWindow"); puts("box2i"); puti(16); puti(xOff); puti(yOff); puti(xOff + w - 1); puti(yOff + h - 1);
    puts("displayWindow"); puts("box2i"); puti(16); puti(0); puti(0); puti(totalX - 1); puti(totalY - 1);
    puts("lineOrder"); puts("lineOrder"); puti(1); { uint8_t z = 0; put(&z, 1); }
    puts("pixelAspectRatio"); puts("float"); puti(4); putf(1.f);
    puts("screenWindowCenter"); puts("v2f"); puti(8); putf(0.f); putf(0.f);
    puts("screenWindowWidth"); puts("float"); puti(4); putf(1.f);
    { uint8_t z = 0; put(&z, 1); }
    std::fwrite(hdr.data(), 1, hdr.size(), fp);
    uint64_t lineBytes = 8 + (uint64_t)w * 4 * 2, base = hdr.size() + (uint64_t)h * 8;
    for (int y = 0; y < h; ++y) { uint64_t off = base + (uint64_t)y * lineBytes; std::fwrite(&off, 8, 1, fp); }
    std::vector<uint16_t> line((size_t)w * 4);
    for (int y = 0; y < h; ++y) {
        int32_t yy = yOff + y, sz = w * 4 * 2;
        std::fwrite(&yy, 4, 1, fp); std::fwrite(&sz, 4, 1, fp);
        for (int x = 0; x < w; ++x) {
            const Float *p = rgb + 3 * ((size_t)y * w + x);
            line[x] = FloatToHalf(1.f); line[w + x] = FloatToHalf(p[2]);
            line[2 * w + x] = FloatToHalf(p[1]); line[3 * w + x] = FloatToHalf(p[0]);
        }
        std::fwrite(line.data(), 2, line.size(), fp);
    }
    std::fclose(fp);
    return true;
}

// ---- 8-bit outputs (imageio.cpp:90-117): gamma-encoded bytes, then PNG (the reference: lodepng_encode24_file) or TGA
// (tga_write_bgr: type 2, 24 bit, uncompressed, top-to-bottom)
static inline Float GammaCorrect(Float value) {   // core/pbrt.h:289-292
    if (value <= 0.0031308f) return 12.92f * value;
    return 1.055f * std::pow(value, (Float)(1.f / 2.4f)) - 0.055f;
}
static std::vector<uint8_t> ToBytes(const Float *rgb, int w, int h) {
    std::vector<uint8_t> out((size_t)3 * w * h);
    for (size_t i = 0; i < out.size(); ++i) out[i] = (uint8_t)Clamp(255.f * GammaCorrect(rgb[i]) + 0.5f, 0.f, 255.f);   // TO_BYTE imageio.cpp:98
    return out;
}
static bool WritePNG(const std::string &name, const uint8_t *rgb8, int w, int h) {   // 8-bit RGB, filter type 0, one IDAT (zlib)
    std::vector<uint8_t> raw((size_t)h * (3 * w + 1));
    for (int y = 0; y < h; ++y) {
        raw[(size_t)y * (3 * w + 1)] = 0;
        std::memcpy(&raw[(size_t)y * (3 * w + 1) + 1], rgb8 + (size_t)y * 3 * w, (size_t)3 * w);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { Error("Error writing PNG \"%s\": deflate failed", name.c_str()); return false; }
    FILE *f = std::fopen(name.c_str(), "wb");
    if (!f) { Error("Error writing PNG \"%s\": cannot open", name.c_str()); return false; }
    auto be32 = [](uint32_t v, uint8_t *p) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; };
    auto chunk = [&](const char *type, const uint8_t *data, uint32_t len) {
        uint8_t hd[8];
        be32(len, hd); std::memcpy(hd + 4, type, 4);
        std::fwrite(hd, 1, 8, f);
        if (len) std::fwrite(data, 1, len, f);
        uLong crc = crc32(0L, (const Bytef *)type, 4);
        if (len) crc = crc32(crc, data, len);
        uint8_t c[4]; be32((uint32_t)crc, c);
        std::fwrite(c, 1, 4, f);
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::fwrite(sig, 1, 8, f);
    uint8_t ihdr[13];
    be32((uint32_t)w, ihdr); be32((uint32_t)h, ihdr + 4);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", comp.data(), (uint32_t)clen);
    chunk("IEND", nullptr, 0);
    std::fclose(f);
    return true;
}
static bool WriteTGA(const std::string &name, const uint8_t *rgb8, int w, int h) {
    FILE *f = std::fopen(name.c_str(), "wb");
    if (!f) { Error("Unable to write output file \"%s\"", name.c_str()); return false; }
    uint8_t hd[18] = {0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, (uint8_t)(w & 255), (uint8_t)(w >> 8), (uint8_t)(h & 255), (uint8_t)(h >> 8), 24, 0x20};
    std::fwrite(hd, 1, 18, f);
    std::vector<uint8_t> bgr((size_t)3 * w * h);
    for (size_t i = 0; i < (size_t)w * h; ++i) { bgr[3 * i] = rgb8[3 * i + 2]; bgr[3 * i + 1] = rgb8[3 * i + 1]; bgr[3 * i + 2] = rgb8[3 * i]; }
    std::fwrite(bgr.data(), 1, bgr.size(), f);
    std::fclose(f);
    return true;
}

bool WriteImage(const std::string &name, const Float *rgb, const int cropMin[2], const int cropMax[2], const int fullRes[2]) {
    int w = cropMax[0] - cropMin[0], h = cropMax[1] - cropMin[1];
    auto ends = [&](const char *suf) { size_t n = std::strlen(suf); return name.size() >= n && name.compare(name.size() - n, n, suf) == 0; };
    if (ends(".pfm")) return WritePFM(name, rgb, w, h);
    if (ends(".exr")) return WriteEXR(name, rgb, w, h, fullRes[0], fullRes[1], cropMin[0], cropMin[1]);
    if (ends(".png") || ends(".tga")) {
        std::vector<uint8_t> b = ToBytes(rgb, w, h);
        return ends(".png") ? WritePNG(name, b.data(), w, h) : WriteTGA(name, b.data(), w, h);
    }
    Error("Can't determine image file type from suffix of filename \"%s\"", name.c_str());
    return false;
}

}  // namespace pbrt_amd
