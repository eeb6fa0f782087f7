// Textures of the host: the "Texture" directive's factories (MakeFloatTexture / MakeSpectrumTexture, api.cpp:613-683 and
// textures/*.cpp Create*Texture) and the MIPMap constructor (core/mipmap.h:101-199).  Nothing is evaluated here except
// constants: every texture becomes a POD mi_texture node (include/pbrt_amd.h) whose per-hit evaluation is the device's
// job.  Parameters, defaults and their lookup order follow the reference factory cited at each branch.
#include <cmath>
#include <cstring>

#include "scene.h"

namespace pbrt_amd {

static std::shared_ptr<TextureStore> g_store;
std::shared_ptr<TextureStore> CurrentTextures() {
    if (!g_store) g_store = std::make_shared<TextureStore>();
    return g_store;
}
void ResetTextures() { g_store = std::make_shared<TextureStore>(); }

static mi_texture BlankNode(int type, bool spectrum) {
    mi_texture t;
    std::memset(&t, 0, sizeof(t));
    t.type = type; t.spectrum = spectrum ? 1 : 0;
    t.tex1 = t.tex2 = t.amount = t.image = -1;
    t.su = t.sv = 1;
    return t;
}
static int AddNode(const mi_texture &t) {
    auto st = CurrentTextures();
    st->nodes.push_back(t);
    return (int)st->nodes.size() - 1;
}
int ConstantTextureNode(bool spectrum, const RGB &v) {   // ConstantTexture<T> (textures/constant.h:48-57)
    auto st = CurrentTextures();
    for (size_t i = 0; i < st->nodes.size(); ++i) {   // share identical constants (keeps the node table small)
        const mi_texture &n = st->nodes[i];
        if (n.type == MI_TEX_CONSTANT && n.spectrum == (spectrum ? 1 : 0) && std::memcmp(n.value, v.c, 3 * sizeof(float)) == 0) return (int)i;
    }
    mi_texture t = BlankNode(MI_TEX_CONSTANT, spectrum);
    for (int c = 0; c < 3; ++c) t.value[c] = v.c[c];
    return AddNode(t);
}

// Constant folding: CONSTANT, and SCALE / MIX of foldable children, evaluated with the reference's own expressions
// (scale.h:57-59, mix.h:58-62).  Everything else depends on the hit.
bool TextureStore::Fold(int node, RGB *out) const {
    if (node < 0 || node >= (int)nodes.size()) return false;
    const mi_texture &t = nodes[node];
    switch (t.type) {
    case MI_TEX_CONSTANT: *out = RGB(t.value[0], t.value[1], t.value[2]); return true;
    case MI_TEX_SCALE: {
        RGB a, b;
        if (!Fold(t.tex1, &a) || !Fold(t.tex2, &b)) return false;
        *out = a * b;
        return true;
    }
    case MI_TEX_MIX: {
        RGB a, b, amt;
        if (!Fold(t.tex1, &a) || !Fold(t.tex2, &b) || !Fold(t.amount, &amt)) return false;
        *out = a * (1 - amt.c[0]) + b * amt.c[0];
        return true;
    }
    default: return false;
    }
}

// ------------------------------------------------------------------------------------------------ MIPMap
namespace {
inline int ModI(int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; }   // core/pbrt.h:310-313
inline int RoundUpPow2I(int v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }
inline bool IsPow2(int v) { return v && !(v & (v - 1)); }
inline int Log2IntU(uint32_t v) { return 31 - __builtin_clz(v); }
Float LanczosW(Float x, Float tau = 2) {   // core/texture.cpp:254-262
    x = std::abs(x);
    if (x < 1e-5f) return 1;
    if (x > 1.f) return 0;
    x *= kPi;
    Float s = std::sin(x * tau) / (x * tau);
    Float lanczos = std::sin(x) / x;
    return s * lanczos;
}
struct ResampleWeight { int firstTexel; Float weight[4]; };
std::vector<ResampleWeight> resampleWeights(int oldRes, int newRes) {   // mipmap.h:77-96
    std::vector<ResampleWeight> wt(newRes);
    Float filterwidth = 2.f;
    for (int i = 0; i < newRes; ++i) {
        Float center = (i + .5f) * oldRes / newRes;
        wt[i].firstTexel = (int)std::floor((center - filterwidth) + 0.5f);
        for (int j = 0; j < 4; ++j) {
            Float pos = wt[i].firstTexel + j + .5f;
            wt[i].weight[j] = LanczosW((pos - center) / filterwidth);
        }
        Float invSumWts = 1 / (wt[i].weight[0] + wt[i].weight[1] + wt[i].weight[2] + wt[i].weight[3]);
        for (int j = 0; j < 4; ++j) wt[i].weight[j] *= invSumWts;
    }
    return wt;
}
inline Float InverseGammaCorrect(Float value) {   // core/pbrt.h:294-297
    if (value <= 0.04045f) return value * 1.f / 12.92f;
    return std::pow((value + 0.055f) * 1.f / 1.055f, (Float)2.4f);
}
}  // namespace

// MIPMap<T>::MIPMap (mipmap.h:101-199), T = Float (channels 1) or RGBSpectrum (channels 3; all its operators are
// componentwise, so one per-channel implementation serves both).  `img` is channels*w*h, row 0 first.
std::shared_ptr<ImagePyramid> BuildMIPMap(int w, int h, int channels, std::vector<Float> img, bool doTrilinear, Float maxAniso, int wrap) {
    auto pyr = std::make_shared<ImagePyramid>();
    const int C = channels;
    if (!IsPow2(w) || !IsPow2(h)) {   // :111-176 resample to a power-of-two resolution, s then t
        int pw = RoundUpPow2I(w), ph = RoundUpPow2I(h);
        std::vector<ResampleWeight> sW = resampleWeights(w, pw);
        std::vector<Float> res((size_t)pw * ph * C, 0.f);
        for (int t = 0; t < h; ++t)
            for (int s = 0; s < pw; ++s)
                for (int c = 0; c < C; ++c) {
                    Float acc = 0.f;
                    for (int j = 0; j < 4; ++j) {
                        int origS = sW[s].firstTexel + j;
                        if (wrap == 0) origS = ModI(origS, w);
                        else if (wrap == 2) origS = Clamp(origS, 0, w - 1);
                        if (origS >= 0 && origS < w) acc += sW[s].weight[j] * img[((size_t)t * w + origS) * C + c];
                    }
                    res[((size_t)t * pw + s) * C + c] = acc;
                }
        std::vector<ResampleWeight> tW = resampleWeights(h, ph);
        std::vector<Float> work((size_t)ph * C);
        for (int s = 0; s < pw; ++s) {
            for (int t = 0; t < ph; ++t)
                for (int c = 0; c < C; ++c) {
                    Float acc = 0.f;
                    for (int j = 0; j < 4; ++j) {
                        int offset = tW[t].firstTexel + j;
                        if (wrap == 0) offset = ModI(offset, h);
                        else if (wrap == 2) offset = Clamp(offset, 0, h - 1);
                        if (offset >= 0 && offset < h) acc += tW[t].weight[j] * res[((size_t)offset * pw + s) * C + c];
                    }
                    work[(size_t)t * C + c] = acc;
                }
            for (int t = 0; t < ph; ++t)
                for (int c = 0; c < C; ++c) res[((size_t)t * pw + s) * C + c] = Clamp(work[(size_t)t * C + c], 0.f, kInfinity);   // clamp() :97-101
        }
        img.swap(res);
        w = pw; h = ph;
    }
    int nLevels = 1 + Log2IntU((uint32_t)std::max(w, h));
    pyr->width = w; pyr->height = h; pyr->levels = nLevels; pyr->channels = C;
    pyr->trilinear = doTrilinear; pyr->maxAniso = maxAniso; pyr->wrap = wrap;
    pyr->texels = std::move(img);
    // :184-199 each level = box filter of four texels of the finer one, fetched through Texel() (wrap mode applies)
    size_t prevOff = 0;
    int pwid = w, phei = h;
    for (int i = 1; i < nLevels; ++i) {
        int sRes = std::max(1, pwid / 2), tRes = std::max(1, phei / 2);
        size_t off = pyr->texels.size();
        pyr->texels.resize(off + (size_t)sRes * tRes * C);
        auto texel = [&](int s, int t, int c) -> Float {
            if (wrap == 0) { s = ModI(s, pwid); t = ModI(t, phei); }
            else if (wrap == 2) { s = Clamp(s, 0, pwid - 1); t = Clamp(t, 0, phei - 1); }
            else if (s < 0 || s >= pwid || t < 0 || t >= phei) return 0.f;
            return pyr->texels[prevOff + ((size_t)t * pwid + s) * C + c];
        };
        for (int t = 0; t < tRes; ++t)
            for (int s = 0; s < sRes; ++s)
                for (int c = 0; c < C; ++c)
                    pyr->texels[off + ((size_t)t * sRes + s) * C + c] =
                        .25f * (texel(2 * s, 2 * t, c) + texel(2 * s + 1, 2 * t, c) + texel(2 * s, 2 * t + 1, c) + texel(2 * s + 1, 2 * t + 1, c));
        prevOff = off; pwid = sRes; phei = tRes;
    }
    return pyr;
}

// ImageTexture<Tmemory,Treturn>::GetTexture (imagemap.cpp:53-98): cache by TexInfo, read, flip in y, convertIn, MIPMap
static int GetImage(const std::string &filename, bool spectrum, bool doTrilinear, Float maxAniso, int wrap, Float scale, bool gamma) {
    auto st = CurrentTextures();
    char key[64];
    std::snprintf(key, sizeof(key), "|%d|%d|%a|%d|%a|%d", spectrum ? 3 : 1, (int)doTrilinear, (double)maxAniso, wrap, (double)scale, (int)gamma);
    std::string k = filename + key;
    auto it = st->imageCache.find(k);
    if (it != st->imageCache.end()) return it->second;
    std::vector<Float> texels;
    int w = 0, h = 0;
    if (!ReadImage(filename, &texels, &w, &h)) {
        Warning("Creating a constant grey texture to replace \"%s\".", filename.c_str());
        w = h = 1;
        texels.assign(3, 0.5f);
    }
    for (int y = 0; y < h / 2; ++y)   // (0,0) of texture space is the lower left corner
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c) std::swap(texels[((size_t)y * w + x) * 3 + c], texels[((size_t)(h - 1 - y) * w + x) * 3 + c]);
    int C = spectrum ? 3 : 1;
    std::vector<Float> conv((size_t)w * h * C);
    for (size_t i = 0; i < (size_t)w * h; ++i) {   // convertIn imagemap.h:101-109
        if (spectrum)
            for (int c = 0; c < 3; ++c) conv[3 * i + c] = scale * (gamma ? InverseGammaCorrect(texels[3 * i + c]) : texels[3 * i + c]);
        else {
            Float y = RGB(texels[3 * i], texels[3 * i + 1], texels[3 * i + 2]).y();
            conv[i] = scale * (gamma ? InverseGammaCorrect(y) : y);
        }
    }
    st->images.push_back(BuildMIPMap(w, h, C, std::move(conv), doTrilinear, maxAniso, wrap));
    st->imageCache[k] = (int)st->images.size() - 1;
    return (int)st->images.size() - 1;
}

// ------------------------------------------------------------------------------------------------ factories
static void copyM(float dst[16], const Matrix4x4 &m) { std::memcpy(dst, m.m, 16 * sizeof(float)); }
// "Initialize 2D texture mapping map from tp" -- the block every 2D texture factory repeats (e.g. imagemap.cpp:105-128)
static void Mapping2D(mi_texture &t, const Transform &tex2world, const TextureParams &tp) {
    std::string type = tp.FindString("mapping", "uv");
    if (type == "uv") {
        t.mapping = MI_MAP_UV;
        t.su = tp.FindFloat("uscale", 1.); t.sv = tp.FindFloat("vscale", 1.);
        t.du = tp.FindFloat("udelta", 0.); t.dv = tp.FindFloat("vdelta", 0.);
    } else if (type == "spherical") { t.mapping = MI_MAP_SPHERICAL; copyM(t.w2t, Inverse(tex2world).m); }
    else if (type == "cylindrical") { t.mapping = MI_MAP_CYLINDRICAL; copyM(t.w2t, Inverse(tex2world).m); }
    else if (type == "planar") {
        t.mapping = MI_MAP_PLANAR;
        Vec3 v1 = tp.FindVector3f("v1", Vec3(1, 0, 0)), v2 = tp.FindVector3f("v2", Vec3(0, 1, 0));
        for (int c = 0; c < 3; ++c) { t.vs[c] = v1[c]; t.vt[c] = v2[c]; }
        t.du = tp.FindFloat("udelta", 0.f); t.dv = tp.FindFloat("vdelta", 0.f);
    } else {
        Error("2D texture mapping \"%s\" unknown", type.c_str());
        t.mapping = MI_MAP_UV;   // UVMapping2D() defaults (texture.h:60)
    }
}
// IdentityMapping3D(tex2world): the 3D texture factories hand tex2world itself to the mapping (fbm.cpp:43 etc.)
static void Mapping3D(mi_texture &t, const Transform &tex2world) { t.mapping = MI_MAP_IDENTITY3D; copyM(t.w2t, tex2world.m); }

static int MakeTexture(const std::string &name, bool spectrum, const Transform &tex2world, const TextureParams &tp) {
    auto child = [&](const char *n, Float def) { return spectrum ? tp.GetSpectrumTexture(n, RGB(def)) : tp.GetFloatTexture(n, def); };
    int node = -1;
    if (name == "constant") {          // constant.cpp:40-49
        node = ConstantTextureNode(spectrum, spectrum ? tp.FindSpectrum("value", RGB(1.f)) : RGB(tp.FindFloat("value", 1.f)));
    } else if (name == "scale") {      // scale.cpp:40-51
        mi_texture t = BlankNode(MI_TEX_SCALE, spectrum);
        t.tex1 = child("tex1", 1.f); t.tex2 = child("tex2", 1.f);
        node = AddNode(t);
    } else if (name == "mix") {        // mix.cpp:40-52
        mi_texture t = BlankNode(MI_TEX_MIX, spectrum);
        t.tex1 = child("tex1", 0.f); t.tex2 = child("tex2", 1.f);
        t.amount = tp.GetFloatTexture("amount", 0.5f);
        node = AddNode(t);
    } else if (name == "bilerp") {     // bilerp.cpp:40-97
        mi_texture t = BlankNode(MI_TEX_BILERP, spectrum);
        Mapping2D(t, tex2world, tp);
        const char *nm[4] = {"v00", "v01", "v10", "v11"};
        float *dst[4] = {t.v00, t.v01, t.v10, t.v11};
        Float def[4] = {0.f, 1.f, 0.f, 1.f};
        for (int i = 0; i < 4; ++i) {
            RGB v = spectrum ? tp.FindSpectrum(nm[i], RGB(def[i])) : RGB(tp.FindFloat(nm[i], def[i]));
            for (int c = 0; c < 3; ++c) dst[i][c] = v.c[c];
        }
        node = AddNode(t);
    } else if (name == "imagemap") {   // imagemap.cpp:100-187
        mi_texture t = BlankNode(MI_TEX_IMAGEMAP, spectrum);
        Mapping2D(t, tex2world, tp);
        Float maxAniso = tp.FindFloat("maxanisotropy", 8.f);
        bool trilerp = tp.FindBool("trilinear", false);
        std::string wrap = tp.FindString("wrap", "repeat");
        int wrapMode = wrap == "black" ? 1 : wrap == "clamp" ? 2 : 0;
        Float scale = tp.FindFloat("scale", 1.f);
        std::string filename = tp.FindFilename("filename");
        bool gamma = tp.FindBool("gamma", HasExtension(filename, ".tga") || HasExtension(filename, ".png"));
        t.image = GetImage(filename, spectrum, trilerp, maxAniso, wrapMode, scale, gamma);
        node = AddNode(t);
    } else if (name == "uv") {         // uv.cpp:40-70 (no Float form)
        if (!spectrum) return -1;
        mi_texture t = BlankNode(MI_TEX_UV, true);
        Mapping2D(t, tex2world, tp);
        node = AddNode(t);
    } else if (name == "checkerboard") {   // checkerboard.cpp:40-154
        int dim = tp.FindInt("dimension", 2);
        if (dim != 2 && dim != 3) { Error("%d dimensional checkerboard texture not supported", dim); return -1; }
        mi_texture t = BlankNode(MI_TEX_CHECKERBOARD, spectrum);
        t.tex1 = child("tex1", 1.f); t.tex2 = child("tex2", 0.f);
        t.dim = dim;
        if (dim == 2) {
            Mapping2D(t, tex2world, tp);
            std::string aa = tp.FindString("aamode", "closedform");
            if (aa == "none") t.aa = 0;
            else {
                if (aa != "closedform") Warning("Antialiasing mode \"%s\" not understood by Checkerboard2DTexture; using \"closedform\"", aa.c_str());
                t.aa = 1;
            }
        } else Mapping3D(t, tex2world);
        node = AddNode(t);
    } else if (name == "dots") {       // dots.cpp:40-96: DotsTexture(map, outsideDot = "inside", insideDot = "outside") -- argument order as written there
        mi_texture t = BlankNode(MI_TEX_DOTS, spectrum);
        Mapping2D(t, tex2world, tp);
        t.tex1 = child("inside", 1.f); t.tex2 = child("outside", 0.f);
        node = AddNode(t);
    } else if (name == "fbm" || name == "wrinkled") {   // fbm.cpp:40-54, wrinkled.cpp:40-55
        mi_texture t = BlankNode(name == "fbm" ? MI_TEX_FBM : MI_TEX_WRINKLED, spectrum);
        Mapping3D(t, tex2world);
        t.octaves = tp.FindInt("octaves", 8); t.omega = tp.FindFloat("roughness", .5f);
        node = AddNode(t);
    } else if (name == "marble") {     // marble.cpp:40-53 (no Float form)
        if (!spectrum) return -1;
        mi_texture t = BlankNode(MI_TEX_MARBLE, true);
        Mapping3D(t, tex2world);
        t.octaves = tp.FindInt("octaves", 8); t.omega = tp.FindFloat("roughness", .5f);
        t.scale = tp.FindFloat("scale", 1.f); t.variation = tp.FindFloat("variation", .2f);
        node = AddNode(t);
    } else if (name == "windy") {      // windy.cpp:40-52
        mi_texture t = BlankNode(MI_TEX_WINDY, spectrum);
        Mapping3D(t, tex2world);
        node = AddNode(t);
    } else if (name == "ptex") {
        Warning("Ptex textures are outside this path's scope (the reference build here has no Ptex either).");
        return -1;
    } else {
        Warning("%s texture \"%s\" unknown.", spectrum ? "Spectrum" : "Float", name.c_str());
        return -1;
    }
    tp.ReportUnused();
    return node;
}
int MakeFloatTexture(const std::string &name, const Transform &tex2world, const TextureParams &tp) { return MakeTexture(name, false, tex2world, tp); }
int MakeSpectrumTexture(const std::string &name, const Transform &tex2world, const TextureParams &tp) { return MakeTexture(name, true, tex2world, tp); }

}  // namespace pbrt_amd
