// InfiniteAreaLight with a radiance map: what the reference's constructor builds (lights/infinite.cpp:43-84) --
// ReadImage (core/imageio.cpp:60-79; PFM here), texels * L, MIPMap<RGBSpectrum> (core/mipmap.h:101-199: Lanczos
// resampling to power-of-two sizes, box-filtered pyramid), the 2w x 2h luminance * sin(theta) image and its
// Distribution2D (core/sampling.cpp:159-171).  The device gets level 0 and the distribution arrays (mi_envmap);
// the pyramid is only needed for Power() (infinite.cpp:86-90).
#include <cmath>
#include <cstring>

#include "scene.h"

namespace pbrt_amd {
namespace {

inline int ModI(int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; }   // core/pbrt.h:310-313
inline Float Log2F(Float x) { const Float invLog2 = 1.442695040888963387004650940071; return std::log(x) * invLog2; }   // pbrt.h:324-327
inline int Log2IntU(uint32_t v) { return 31 - __builtin_clz(v); }
inline int RoundUpPow2I(int v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }

Float Lanczos(Float x, Float tau = 2) {   // core/texture.cpp:254-262
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
            wt[i].weight[j] = Lanczos((pos - center) / filterwidth);
        }
        Float invSumWts = 1 / (wt[i].weight[0] + wt[i].weight[1] + wt[i].weight[2] + wt[i].weight[3]);
        for (int j = 0; j < 4; ++j) wt[i].weight[j] *= invSumWts;
    }
    return wt;
}

struct Level { int w, h; std::vector<RGB> px; const RGB &at(int s, int t) const { return px[(size_t)ModI(t, h) * w + ModI(s, w)]; } };   // ImageWrap::Repeat

RGB triangle(const Level &l, Float s_, Float t_) {   // mipmap.h:263-275
    Float s = s_ * l.w - 0.5f, t = t_ * l.h - 0.5f;
    int s0 = (int)std::floor(s), t0 = (int)std::floor(t);
    Float ds = s - s0, dt = t - t0;
    return l.at(s0, t0) * ((1 - ds) * (1 - dt)) + l.at(s0, t0 + 1) * ((1 - ds) * dt) + l.at(s0 + 1, t0) * (ds * (1 - dt)) + l.at(s0 + 1, t0 + 1) * (ds * dt);
}
// MIPMap::Lookup(st, width) mipmap.h:244-261
RGB Lookup(const std::vector<Level> &pyr, Float s, Float t, Float width) {
    int levels = (int)pyr.size();
    Float level = levels - 1 + Log2F(std::max(width, (Float)1e-8));
    if (level < 0) return triangle(pyr[0], s, t);
    if (level >= levels - 1) return pyr[levels - 1].at(0, 0);
    int iLevel = (int)std::floor(level);
    Float delta = level - iLevel;
    return triangle(pyr[iLevel], s, t) * (1 - delta) + triangle(pyr[iLevel + 1], s, t) * delta;   // Lerp core/pbrt.h:351
}

void Distribution1D(const Float *f, int n, Float *func, Float *cdf, Float *funcInt) {   // core/sampling.h:55-70
    for (int i = 0; i < n; ++i) func[i] = f[i];
    cdf[0] = 0;
    for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / n;
    *funcInt = cdf[n];
    if (*funcInt == 0) for (int i = 1; i < n + 1; ++i) cdf[i] = Float(i) / Float(n);
    else for (int i = 1; i < n + 1; ++i) cdf[i] /= *funcInt;
}

}  // namespace

std::shared_ptr<EnvMap> CreateEnvMap(const std::string &filename, const RGB &L) {
    std::vector<Float> texels;
    int w = 0, h = 0;
    if (filename.empty()) {   // no map: the constructor's 1x1 image of ones (infinite.cpp:58-62) -- used for constant lights under a transform
        texels.assign(3, 1.f);
        w = h = 1;
    } else if (!ReadImage(filename, &texels, &w, &h)) {   // .pfm, .png, .tga, .exr (host/imageread.cpp); the radiance map is used as read: no gamma (infinite.cpp:51)
        Error("Unable to read environment map \"%s\"", filename.c_str());
        return nullptr;
    }
    Level l0;
    l0.w = w; l0.h = h;
    l0.px.resize((size_t)w * h);
    for (size_t i = 0; i < l0.px.size(); ++i) l0.px[i] = RGB(texels[3 * i], texels[3 * i + 1], texels[3 * i + 2]) * L;   // infinite.cpp:54-56
    // ---- MIPMap constructor: resample to power-of-two resolution (mipmap.h:120-182)
    if (w != RoundUpPow2I(w) || h != RoundUpPow2I(h)) {
        int pw = RoundUpPow2I(w), ph = RoundUpPow2I(h);
        std::vector<ResampleWeight> sW = resampleWeights(w, pw);
        std::vector<RGB> tmp((size_t)pw * ph, RGB(0.f));
        for (int t = 0; t < h; ++t)
            for (int s = 0; s < pw; ++s) {
                RGB acc(0.f);
                for (int j = 0; j < 4; ++j) {
                    int origS = ModI(sW[s].firstTexel + j, w);
                    if (origS >= 0 && origS < w) acc = acc + l0.px[(size_t)t * w + origS] * sW[s].weight[j];
                }
                tmp[(size_t)t * pw + s] = acc;
            }
        std::vector<ResampleWeight> tW = resampleWeights(h, ph);
        std::vector<RGB> work(ph);
        for (int s = 0; s < pw; ++s) {
            for (int t = 0; t < ph; ++t) {
                RGB acc(0.f);
                for (int j = 0; j < 4; ++j) {
                    int offset = ModI(tW[t].firstTexel + j, h);
                    if (offset >= 0 && offset < h) acc = acc + tmp[(size_t)offset * pw + s] * tW[t].weight[j];
                }
                work[t] = acc;
            }
            for (int t = 0; t < ph; ++t) {   // clamp(v) = v.Clamp(0, Infinity)
                RGB v = work[t];
                for (int c = 0; c < 3; ++c) v.c[c] = v.c[c] < 0 ? 0 : v.c[c];
                tmp[(size_t)t * pw + s] = v;
            }
        }
        l0.w = pw; l0.h = ph; l0.px.swap(tmp);
        w = pw; h = ph;
    }
    // ---- pyramid (mipmap.h:184-204)
    std::vector<Level> pyr;
    pyr.push_back(l0);
    int nLevels = 1 + Log2IntU((uint32_t)std::max(w, h));
    for (int i = 1; i < nLevels; ++i) {
        const Level &f = pyr[i - 1];
        Level c;
        c.w = std::max(1, f.w / 2); c.h = std::max(1, f.h / 2);
        c.px.resize((size_t)c.w * c.h);
        for (int t = 0; t < c.h; ++t)
            for (int s = 0; s < c.w; ++s)
                c.px[(size_t)t * c.w + s] = (f.at(2 * s, 2 * t) + f.at(2 * s + 1, 2 * t) + f.at(2 * s, 2 * t + 1) + f.at(2 * s + 1, 2 * t + 1)) * .25f;
        pyr.push_back(std::move(c));
    }
    auto env = std::make_shared<EnvMap>();
    env->width = w; env->height = h;
    env->rgb.resize((size_t)w * h * 3);
    for (size_t i = 0; i < pyr[0].px.size(); ++i) for (int c = 0; c < 3; ++c) env->rgb[3 * i + c] = pyr[0].px[i].c[c];
    // ---- scalar image and Distribution2D (infinite.cpp:66-84, sampling.cpp:159-171)
    int width = 2 * w, height = 2 * h;
    std::vector<Float> img((size_t)width * height);
    float fwidth = 0.5f / std::min(width, height);
    for (int v = 0; v < height; ++v) {
        Float vp = (v + .5f) / (Float)height;
        Float sinTheta = std::sin(kPi * (v + .5f) / height);
        for (int u = 0; u < width; ++u) {
            Float up = (u + .5f) / (Float)width;
            img[u + (size_t)v * width] = Lookup(pyr, up, vp, fwidth).y();
            img[u + (size_t)v * width] *= sinTheta;
        }
    }
    env->condFunc.resize((size_t)width * height); env->condCdf.resize((size_t)(width + 1) * height); env->condFuncInt.resize(height);
    env->margFunc.resize(height); env->margCdf.resize(height + 1);
    for (int v = 0; v < height; ++v)
        Distribution1D(&img[(size_t)v * width], width, &env->condFunc[(size_t)v * width], &env->condCdf[(size_t)v * (width + 1)], &env->condFuncInt[v]);
    Distribution1D(env->condFuncInt.data(), height, env->margFunc.data(), env->margCdf.data(), &env->margFuncInt);
    // Power(): Pi r^2 * Lmap->Lookup((.5,.5), .5)  -- the radius is the integrator's business
    env->powerLookup = Lookup(pyr, .5f, .5f, .5f);
    return env;
}

}  // namespace pbrt_amd
