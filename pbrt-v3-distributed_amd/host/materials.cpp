// Material factories (host).  Each Create*Material reads the same parameters with the same
// defaults as the reference factory, then performs the reference's
// ComputeScatteringFunctions(si, arena, TransportMode::Radiance, allowMultipleLobes=true)
// once -- legal here because every texture on this path is constant -- and records the BxDFs
// it would Add(), in order, as mi_bxdf PODs.  File:line of each reference routine is cited.
#include "scene.h"

namespace pbrt_amd {

static Float RoughnessToAlpha(Float roughness) {   // microfacet.h:123-128
    roughness = std::max(roughness, (Float)1e-3);
    Float x = std::log(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}

static mi_bxdf blank(int type) {
    mi_bxdf b;
    std::memset(&b, 0, sizeof(b));
    b.type = type;
    for (int i = 0; i < 3; ++i) b.scale[i] = 1;
    return b;
}
static void set3(float d[3], const RGB &c) { d[0] = c.c[0]; d[1] = c.c[1]; d[2] = c.c[2]; }
static void add(mi_material &m, const mi_bxdf &b) {
    if (m.n_bxdfs >= MI_MAX_BXDFS) { Error("BSDF has more than %d BxDFs (reflection.h:199)", MI_MAX_BXDFS); return; }
    m.bxdfs[m.n_bxdfs++] = b;
}
static mi_bxdf lambertR(const RGB &r) { mi_bxdf b = blank(MI_BXDF_LAMBERT_R); set3(b.R, r); return b; }
static mi_bxdf lambertT(const RGB &t) { mi_bxdf b = blank(MI_BXDF_LAMBERT_T); set3(b.T, t); return b; }
static mi_bxdf orenNayar(const RGB &r, Float sigma) {   // reflection.h:414-420
    mi_bxdf b = blank(MI_BXDF_OREN_NAYAR);
    set3(b.R, r);
    sigma = Radians(sigma);
    Float sigma2 = sigma * sigma;
    b.A = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
    b.B = 0.45f * sigma2 / (sigma2 + 0.09f);
    return b;
}
static mi_bxdf microR(const RGB &r, Float ax, Float ay, int fresnel, Float etaI, Float etaT) {
    mi_bxdf b = blank(MI_BXDF_MICROFACET_R);
    set3(b.R, r); b.alphax = ax; b.alphay = ay; b.fresnel = fresnel; b.etaA = etaI; b.etaB = etaT;
    return b;
}
static mi_bxdf microT(const RGB &t, Float ax, Float ay, Float etaA, Float etaB) {
    mi_bxdf b = blank(MI_BXDF_MICROFACET_T);
    set3(b.T, t); b.alphax = ax; b.alphay = ay; b.etaA = etaA; b.etaB = etaB; b.fresnel = MI_FRESNEL_DIELECTRIC;
    return b;
}
static mi_bxdf specR(const RGB &r, int fresnel, Float etaI, Float etaT) {
    mi_bxdf b = blank(MI_BXDF_SPECULAR_R);
    set3(b.R, r); b.fresnel = fresnel; b.etaA = etaI; b.etaB = etaT;
    return b;
}
static mi_bxdf specT(const RGB &t, Float etaA, Float etaB) {
    mi_bxdf b = blank(MI_BXDF_SPECULAR_T);
    set3(b.T, t); b.etaA = etaA; b.etaB = etaB; b.fresnel = MI_FRESNEL_DIELECTRIC;
    return b;
}

static std::shared_ptr<Material> newMat(const std::string &type, Float eta = 1) {
    auto m = std::make_shared<Material>();
    m->type = type;
    std::memset(&m->bsdf, 0, sizeof(m->bsdf));
    m->bsdf.eta = eta;
    return m;
}
static void warnBump(const TextureParams &mp) {
    Float dummy;
    if (mp.geom().FindTexture("bumpmap") != "" || mp.mat().FindTexture("bumpmap") != "" || mp.GetFloatOrNull("bumpmap", &dummy))
        Warning("bump mapping is not supported by this path (SURVEY.md s.8 row f2); ignored");
}

static std::shared_ptr<Material> CreateMatte(const TextureParams &mp) {   // matte.cpp:45-72
    RGB Kd = mp.GetSpectrum("Kd", RGB(0.5f));
    Float sigma = mp.GetFloat("sigma", 0.f);
    warnBump(mp);
    auto m = newMat("matte");
    RGB r = Kd.Clamp();
    Float sig = Clamp(sigma, 0, 90);
    if (!r.IsBlack()) add(m->bsdf, sig == 0 ? lambertR(r) : orenNayar(r, sig));
    return m;
}

static std::shared_ptr<Material> CreatePlastic(const TextureParams &mp) {   // plastic.cpp:45-85
    RGB Kd = mp.GetSpectrum("Kd", RGB(0.25f)), Ks = mp.GetSpectrum("Ks", RGB(0.25f));
    Float rough = mp.GetFloat("roughness", .1f);
    warnBump(mp);
    bool remap = mp.FindBool("remaproughness", true);
    auto m = newMat("plastic");
    RGB kd = Kd.Clamp();
    if (!kd.IsBlack()) add(m->bsdf, lambertR(kd));
    RGB ks = Ks.Clamp();
    if (!ks.IsBlack()) {
        if (remap) rough = RoughnessToAlpha(rough);
        add(m->bsdf, microR(ks, rough, rough, MI_FRESNEL_DIELECTRIC, 1.5f, 1.f));   // FresnelDielectric(1.5, 1): plastic.cpp:59
    }
    return m;
}

static std::shared_ptr<Material> CreateGlass(const TextureParams &mp) {   // glass.cpp:45-110
    RGB Kr = mp.GetSpectrum("Kr", RGB(1.f)), Kt = mp.GetSpectrum("Kt", RGB(1.f));
    Float eta;
    if (!mp.GetFloatOrNull("eta", &eta)) eta = mp.GetFloat("index", 1.5f);
    Float urough = mp.GetFloat("uroughness", 0.f), vrough = mp.GetFloat("vroughness", 0.f);
    warnBump(mp);
    bool remap = mp.FindBool("remaproughness", true);
    auto m = newMat("glass", eta);
    RGB R = Kr.Clamp(), T = Kt.Clamp();
    if (R.IsBlack() && T.IsBlack()) return m;
    bool isSpecular = urough == 0 && vrough == 0;
    if (isSpecular) {   // allowMultipleLobes is always true from PathIntegrator::Li (path.cpp:107)
        mi_bxdf b = blank(MI_BXDF_FRESNEL_SPEC);
        set3(b.R, R); set3(b.T, T); b.etaA = 1.f; b.etaB = eta;
        add(m->bsdf, b);
    } else {
        if (remap) { urough = RoughnessToAlpha(urough); vrough = RoughnessToAlpha(vrough); }
        if (!R.IsBlack()) add(m->bsdf, microR(R, urough, vrough, MI_FRESNEL_DIELECTRIC, 1.f, eta));
        if (!T.IsBlack()) add(m->bsdf, microT(T, urough, vrough, 1.f, eta));
    }
    return m;
}

static std::shared_ptr<Material> CreateMirror(const TextureParams &mp) {   // mirror.cpp:45-65
    RGB Kr = mp.GetSpectrum("Kr", RGB(0.9f));
    warnBump(mp);
    auto m = newMat("mirror");
    RGB R = Kr.Clamp();
    if (!R.IsBlack()) add(m->bsdf, specR(R, MI_FRESNEL_NOOP, 1, 1));
    return m;
}

static std::shared_ptr<Material> CreateMetal(const TextureParams &mp) {   // metal.cpp:59-134
    // Default eta/k: copper SPD -> RGB via Spectrum::FromSampled (metal.cpp:81-118).  The two RGB
    // triples below are those values as produced by the reference build (oracle/ref_build probe).
    RGB copperN(0.19999069f, 0.92208463f, 1.09987593f), copperK(3.90463543f, 2.44763327f, 2.13765264f);
    RGB eta = mp.GetSpectrum("eta", copperN), k = mp.GetSpectrum("k", copperK);
    Float rough = mp.GetFloat("roughness", .01f), u, v;
    bool hasU = mp.GetFloatOrNull("uroughness", &u), hasV = mp.GetFloatOrNull("vroughness", &v);
    warnBump(mp);
    bool remap = mp.FindBool("remaproughness", true);
    auto m = newMat("metal");
    Float uRough = hasU ? u : rough, vRough = hasV ? v : rough;
    if (remap) { uRough = RoughnessToAlpha(uRough); vRough = RoughnessToAlpha(vRough); }
    mi_bxdf b = microR(RGB(1.f), uRough, vRough, MI_FRESNEL_CONDUCTOR, 1.f, 1.f);
    set3(b.eta_c, eta); set3(b.k_c, k);
    add(m->bsdf, b);
    return m;
}

static std::shared_ptr<Material> CreateUber(const TextureParams &mp) {   // uber.cpp:45-135
    RGB Kd = mp.GetSpectrum("Kd", RGB(0.25f)), Ks = mp.GetSpectrum("Ks", RGB(0.25f));
    RGB Kr = mp.GetSpectrum("Kr", RGB(0.f)), Kt = mp.GetSpectrum("Kt", RGB(0.f));
    Float rough = mp.GetFloat("roughness", .1f), ur, vr;
    bool hasU = mp.GetFloatOrNull("uroughness", &ur), hasV = mp.GetFloatOrNull("vroughness", &vr);
    Float e;
    if (!mp.GetFloatOrNull("eta", &e)) e = mp.GetFloat("index", 1.5f);
    RGB opacity = mp.GetSpectrum("opacity", RGB(1.f));
    warnBump(mp);
    bool remap = mp.FindBool("remaproughness", true);
    RGB op = opacity.Clamp();
    RGB t = (-op + RGB(1.f)).Clamp();
    std::shared_ptr<Material> m;
    if (!t.IsBlack()) {
        m = newMat("uber", 1.f);
        add(m->bsdf, specT(t, 1.f, 1.f));
    } else
        m = newMat("uber", e);
    RGB kd = op * Kd.Clamp();
    if (!kd.IsBlack()) add(m->bsdf, lambertR(kd));
    RGB ks = op * Ks.Clamp();
    if (!ks.IsBlack()) {
        Float roughu = hasU ? ur : rough;
        Float roughv = hasV ? vr : roughu;
        if (remap) { roughu = RoughnessToAlpha(roughu); roughv = RoughnessToAlpha(roughv); }
        add(m->bsdf, microR(ks, roughu, roughv, MI_FRESNEL_DIELECTRIC, 1.f, e));
    }
    RGB kr = op * Kr.Clamp();
    if (!kr.IsBlack()) add(m->bsdf, specR(kr, MI_FRESNEL_DIELECTRIC, 1.f, e));
    RGB kt = op * Kt.Clamp();
    if (!kt.IsBlack()) add(m->bsdf, specT(kt, 1.f, e));
    return m;
}

static std::shared_ptr<Material> CreateSubstrate(const TextureParams &mp) {   // substrate.cpp:45-81
    RGB Kd = mp.GetSpectrum("Kd", RGB(.5f)), Ks = mp.GetSpectrum("Ks", RGB(.5f));
    Float roughu = mp.GetFloat("uroughness", .1f), roughv = mp.GetFloat("vroughness", .1f);
    warnBump(mp);
    bool remap = mp.FindBool("remaproughness", true);
    auto m = newMat("substrate");
    RGB d = Kd.Clamp(), s = Ks.Clamp();
    if (!d.IsBlack() || !s.IsBlack()) {
        if (remap) { roughu = RoughnessToAlpha(roughu); roughv = RoughnessToAlpha(roughv); }
        mi_bxdf b = blank(MI_BXDF_FRESNEL_BLEND);
        set3(b.R, d); set3(b.T, s); b.alphax = roughu; b.alphay = roughv;
        add(m->bsdf, b);
    }
    return m;
}

static std::shared_ptr<Material> CreateTranslucent(const TextureParams &mp) {   // translucent.cpp:45-98
    RGB Kd = mp.GetSpectrum("Kd", RGB(0.25f)), Ks = mp.GetSpectrum("Ks", RGB(0.25f));
    RGB reflect = mp.GetSpectrum("reflect", RGB(0.5f)), transmit = mp.GetSpectrum("transmit", RGB(0.5f));
    Float rough = mp.GetFloat("roughness", .1f);
    warnBump(mp);
    bool remap = mp.FindBool("remaproughness", true);
    Float eta = 1.5f;
    auto m = newMat("translucent", eta);
    RGB r = reflect.Clamp(), t = transmit.Clamp();
    if (r.IsBlack() && t.IsBlack()) return m;
    RGB kd = Kd.Clamp();
    if (!kd.IsBlack()) {
        if (!r.IsBlack()) add(m->bsdf, lambertR(r * kd));
        if (!t.IsBlack()) add(m->bsdf, lambertT(t * kd));
    }
    RGB ks = Ks.Clamp();
    if (!ks.IsBlack() && (!r.IsBlack() || !t.IsBlack())) {
        if (remap) rough = RoughnessToAlpha(rough);
        if (!r.IsBlack()) add(m->bsdf, microR(r * ks, rough, rough, MI_FRESNEL_DIELECTRIC, 1.f, eta));
        if (!t.IsBlack()) add(m->bsdf, microT(t * ks, rough, rough, 1.f, eta));
    }
    return m;
}

static std::shared_ptr<Material> CreateMix(const TextureParams &mp, const std::shared_ptr<Material> &m1,
                                           const std::shared_ptr<Material> &m2) {   // mixmat.cpp:46-77
    RGB amount = mp.GetSpectrum("amount", RGB(0.5f));
    RGB s1 = amount.Clamp();
    RGB s2 = (RGB(1.f) - s1).Clamp();
    // MixMaterial builds on m1's BSDF (its eta) and wraps every lobe in ScaledBxDF
    auto m = newMat("mix", m1 ? m1->bsdf.eta : 1.f);
    if (!m1 || !m2) { Error("mix material needs two non-null materials"); return m; }
    auto wrap = [&](const mi_bxdf &src, const RGB &s) {
        mi_bxdf b = src;
        if (b.scaled) {   // nested mix: ScaledBxDF(ScaledBxDF(x, a), s) == s * (a * f); fold (1-ulp order difference)
            for (int i = 0; i < 3; ++i) b.scale[i] = s.c[i] * b.scale[i];
        } else {
            b.scaled = 1;
            set3(b.scale, s);
        }
        return b;
    };
    for (int i = 0; i < m1->bsdf.n_bxdfs; ++i) add(m->bsdf, wrap(m1->bsdf.bxdfs[i], s1));
    for (int i = 0; i < m2->bsdf.n_bxdfs; ++i) add(m->bsdf, wrap(m2->bsdf.bxdfs[i], s2));
    return m;
}

std::shared_ptr<Material> MakeMaterial(const std::string &name, const TextureParams &mp,
                                       const std::map<std::string, std::shared_ptr<Material>> *named) {   // api.cpp:541-611
    std::shared_ptr<Material> material;
    if (name == "" || name == "none") return nullptr;
    else if (name == "matte") material = CreateMatte(mp);
    else if (name == "plastic") material = CreatePlastic(mp);
    else if (name == "translucent") material = CreateTranslucent(mp);
    else if (name == "glass") material = CreateGlass(mp);
    else if (name == "mirror") material = CreateMirror(mp);
    else if (name == "mix") {
        std::string n1 = mp.FindString("namedmaterial1", ""), n2 = mp.FindString("namedmaterial2", "");
        std::shared_ptr<Material> mat1, mat2;
        auto lookup = [&](const std::string &n) -> std::shared_ptr<Material> {
            if (!named || named->find(n) == named->end()) {
                Error("Named material \"%s\" undefined.  Using \"matte\"", n.c_str());
                return CreateMatte(mp);
            }
            return named->at(n);
        };
        mat1 = lookup(n1); mat2 = lookup(n2);
        material = CreateMix(mp, mat1, mat2);
    } else if (name == "metal") material = CreateMetal(mp);
    else if (name == "substrate") material = CreateSubstrate(mp);
    else if (name == "uber") material = CreateUber(mp);
    else if (name == "hair" || name == "disney" || name == "subsurface" || name == "kdsubsurface" || name == "fourier") {
        Warning("Material \"%s\" is outside the GPU path's scope (SURVEY.md s.2 row 20). Using \"matte\".", name.c_str());
        material = CreateMatte(mp);
    } else {
        Warning("Material \"%s\" unknown. Using \"matte\".", name.c_str());
        material = CreateMatte(mp);
    }
    mp.ReportUnused();
    return material;
}

}  // namespace pbrt_amd
