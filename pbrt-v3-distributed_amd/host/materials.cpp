// Material factories (host).  Each Create*Material reads the same parameters with the same
// defaults as the reference factory and records the parameter textures as nodes (mi_material_desc).
// When every parameter is hit-independent (constants, scale / mix of constants) and there is no bump
// map, it also performs the reference's
// ComputeScatteringFunctions(si, arena, TransportMode::Radiance, allowMultipleLobes=true)
// once and records the BxDFs it would Add(), in order, as mi_bxdf PODs; otherwise the device builds the
// same list per hit from the desc.  File:line of each reference routine is cited.
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

static std::shared_ptr<Material> newMat(const std::string &type, int descType, Float eta = 1) {
    auto m = std::make_shared<Material>();
    m->type = type;
    std::memset(&m->bsdf, 0, sizeof(m->bsdf));
    m->bsdf.eta = eta;
    std::memset(&m->desc, 0xff, sizeof(m->desc));   // every node index -1
    m->desc.type = descType; m->desc.textured = 0; m->desc.remap_roughness = 0; m->desc.pad = 0;
    std::memset(&m->bssrdf, 0, sizeof(m->bssrdf));   // MI_BSSRDF_NONE
    return m;
}

// The parameter textures of one material: node ids for the desc, folded values for the constant path.
// `textured` turns on as soon as one parameter depends on the hit, or a bump map is present (Material::Bump changes
// the shading frame even for a constant displacement when the mesh has shading normals, material.cpp:73-80).
struct ParamNodes {
    const TextureParams &mp;
    std::shared_ptr<TextureStore> st = CurrentTextures();
    bool textured = false;
    explicit ParamNodes(const TextureParams &mp) : mp(mp) {}
    int spec(const char *n, const RGB &def, RGB *v) {
        int t = mp.GetSpectrumTexture(n, def);
        if (!st->Fold(t, v)) { textured = true; *v = def; }
        return t;
    }
    int flt(const char *n, Float def, Float *v) {
        int t = mp.GetFloatTexture(n, def);
        RGB r;
        if (st->Fold(t, &r)) *v = r.c[0]; else { textured = true; *v = def; }
        return t;
    }
    int fltOrNull(const char *n, Float *v, bool *has) {
        int t = mp.GetFloatTextureOrNull(n);
        *has = t >= 0;
        RGB r;
        if (t >= 0) { if (st->Fold(t, &r)) *v = r.c[0]; else textured = true; }
        return t;
    }
    int bump() {
        int t = mp.GetFloatTextureOrNull("bumpmap");
        if (t >= 0) textured = true;
        return t;
    }
};

static std::shared_ptr<Material> CreateMatte(const TextureParams &mp) {   // matte.cpp:45-72
    ParamNodes pn(mp);
    RGB Kd; Float sigma;
    auto m = newMat("matte", MI_MAT_MATTE);
    m->desc.Kd = pn.spec("Kd", RGB(0.5f), &Kd);
    m->desc.sigma = pn.flt("sigma", 0.f, &sigma);
    m->desc.bump = pn.bump();
    if ((m->desc.textured = pn.textured)) return m;
    RGB r = Kd.Clamp();
    Float sig = Clamp(sigma, 0, 90);
    if (!r.IsBlack()) add(m->bsdf, sig == 0 ? lambertR(r) : orenNayar(r, sig));
    return m;
}

static std::shared_ptr<Material> CreatePlastic(const TextureParams &mp) {   // plastic.cpp:45-85
    ParamNodes pn(mp);
    RGB Kd, Ks; Float rough;
    auto m = newMat("plastic", MI_MAT_PLASTIC);
    m->desc.Kd = pn.spec("Kd", RGB(0.25f), &Kd); m->desc.Ks = pn.spec("Ks", RGB(0.25f), &Ks);
    m->desc.roughness = pn.flt("roughness", .1f, &rough);
    m->desc.bump = pn.bump();
    bool remap = mp.FindBool("remaproughness", true);
    m->desc.remap_roughness = remap;
    if ((m->desc.textured = pn.textured)) return m;
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
    ParamNodes pn(mp);
    RGB Kr, Kt; Float eta = 1.5f, urough, vrough;
    bool hasEta;
    auto m = newMat("glass", MI_MAT_GLASS);
    m->desc.Kr = pn.spec("Kr", RGB(1.f), &Kr); m->desc.Kt = pn.spec("Kt", RGB(1.f), &Kt);
    m->desc.eta_f = pn.fltOrNull("eta", &eta, &hasEta);
    if (!hasEta) m->desc.eta_f = pn.flt("index", 1.5f, &eta);
    m->desc.uroughness = pn.flt("uroughness", 0.f, &urough); m->desc.vroughness = pn.flt("vroughness", 0.f, &vrough);
    m->desc.bump = pn.bump();
    bool remap = mp.FindBool("remaproughness", true);
    m->desc.remap_roughness = remap;
    if ((m->desc.textured = pn.textured)) return m;
    m->bsdf.eta = eta;
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

// GetMediumScatteringProperties (core/medium.cpp:174-185) lives in api.cpp next to MakeMedium
bool GetMediumScatteringProperties(const std::string &name, RGB *sigma_a, RGB *sigma_prime_s);

// SubsurfaceMaterial / KdSubsurfaceMaterial (materials/subsurface.cpp:102-135, kdsubsurface.cpp:96-121): GlassMaterial's BSDF with the
// material's constant eta, evaluated per hit (textured = 1), plus the BSSRDF description and the table of the constructor
static std::shared_ptr<Material> CreateSubsurface(const TextureParams &mp, bool kd) {
    ParamNodes pn(mp);
    RGB dummy;
    Float fdummy;
    auto m = newMat(kd ? "kdsubsurface" : "subsurface", MI_MAT_GLASS);
    mi_bssrdf_desc &b = m->bssrdf;
    b.kind = kd ? MI_BSSRDF_KDSUBSURFACE : MI_BSSRDF_SUBSURFACE;
    b.sigma_a = b.sigma_s = b.Kd = b.mfp = -1;
    Float g, scale, eta;
    if (!kd) {
        RGB sig_a(.0011f, .0024f, .014f), sig_s(2.55f, 3.21f, 3.77f);
        std::string name = mp.FindString("name", "");
        bool found = GetMediumScatteringProperties(name, &sig_a, &sig_s);
        g = mp.FindFloat("g", 0.0f);
        if (name != "") {
            if (!found) Warning("Named material \"%s\" not found.  Using defaults.", name.c_str());
            else g = 0;   // the database holds reduced scattering coefficients
        }
        scale = mp.FindFloat("scale", 1.f);
        eta = mp.FindFloat("eta", 1.33f);
        b.sigma_a = pn.spec("sigma_a", sig_a, &dummy); b.sigma_s = pn.spec("sigma_s", sig_s, &dummy);
        m->desc.Kr = pn.spec("Kr", RGB(1.f), &dummy); m->desc.Kt = pn.spec("Kt", RGB(1.f), &dummy);
    } else {
        b.Kd = pn.spec("Kd", RGB(.5f), &dummy); b.mfp = pn.spec("mfp", RGB(1.f), &dummy);
        m->desc.Kr = pn.spec("Kr", RGB(1.f), &dummy); m->desc.Kt = pn.spec("Kt", RGB(1.f), &dummy);
    }
    m->desc.uroughness = pn.flt("uroughness", 0.f, &fdummy); m->desc.vroughness = pn.flt("vroughness", 0.f, &fdummy);
    m->desc.bump = pn.bump();
    if (kd) { eta = mp.FindFloat("eta", 1.33f); scale = mp.FindFloat("scale", 1.0f); g = mp.FindFloat("g", 0.0f); }
    m->desc.remap_roughness = mp.FindBool("remaproughness", true);
    m->desc.eta_f = ConstantTextureNode(false, RGB(eta));
    m->desc.textured = 1;   // always built per hit: the BSSRDF needs the interaction anyway
    b.scale = scale; b.eta = eta; b.g = g;
    m->table = MakeBSSRDFTable(g, eta);
    return m;
}

static std::shared_ptr<Material> CreateMirror(const TextureParams &mp) {   // mirror.cpp:45-65
    ParamNodes pn(mp);
    RGB Kr;
    auto m = newMat("mirror", MI_MAT_MIRROR);
    m->desc.Kr = pn.spec("Kr", RGB(0.9f), &Kr);
    m->desc.bump = pn.bump();
    if ((m->desc.textured = pn.textured)) return m;
    RGB R = Kr.Clamp();
    if (!R.IsBlack()) add(m->bsdf, specR(R, MI_FRESNEL_NOOP, 1, 1));
    return m;
}

static std::shared_ptr<Material> CreateMetal(const TextureParams &mp) {   // metal.cpp:59-134
    // Default eta/k: copper SPD -> RGB via Spectrum::FromSampled (metal.cpp:81-118).  The two RGB
    // triples below are those values as produced by the reference build (oracle/ref_build probe).
    RGB copperN(0.19999069f, 0.92208463f, 1.09987593f), copperK(3.90463543f, 2.44763327f, 2.13765264f);
    ParamNodes pn(mp);
    RGB eta, k; Float rough, u = 0, v = 0;
    bool hasU, hasV;
    auto m = newMat("metal", MI_MAT_METAL);
    m->desc.eta_s = pn.spec("eta", copperN, &eta); m->desc.k_s = pn.spec("k", copperK, &k);
    m->desc.roughness = pn.flt("roughness", .01f, &rough);
    m->desc.uroughness = pn.fltOrNull("uroughness", &u, &hasU); m->desc.vroughness = pn.fltOrNull("vroughness", &v, &hasV);
    m->desc.bump = pn.bump();
    bool remap = mp.FindBool("remaproughness", true);
    m->desc.remap_roughness = remap;
    if ((m->desc.textured = pn.textured)) return m;
    Float uRough = hasU ? u : rough, vRough = hasV ? v : rough;
    if (remap) { uRough = RoughnessToAlpha(uRough); vRough = RoughnessToAlpha(vRough); }
    mi_bxdf b = microR(RGB(1.f), uRough, vRough, MI_FRESNEL_CONDUCTOR, 1.f, 1.f);
    set3(b.eta_c, eta); set3(b.k_c, k);
    add(m->bsdf, b);
    return m;
}

static std::shared_ptr<Material> CreateUber(const TextureParams &mp) {   // uber.cpp:45-135
    ParamNodes pn(mp);
    RGB Kd, Ks, Kr, Kt, opacity; Float rough, ur = 0, vr = 0, e = 1.5f;
    bool hasU, hasV, hasEta;
    auto m = newMat("uber", MI_MAT_UBER);
    m->desc.Kd = pn.spec("Kd", RGB(0.25f), &Kd); m->desc.Ks = pn.spec("Ks", RGB(0.25f), &Ks);
    m->desc.Kr = pn.spec("Kr", RGB(0.f), &Kr); m->desc.Kt = pn.spec("Kt", RGB(0.f), &Kt);
    m->desc.roughness = pn.flt("roughness", .1f, &rough);
    m->desc.uroughness = pn.fltOrNull("uroughness", &ur, &hasU); m->desc.vroughness = pn.fltOrNull("vroughness", &vr, &hasV);
    m->desc.eta_f = pn.fltOrNull("eta", &e, &hasEta);
    if (!hasEta) m->desc.eta_f = pn.flt("index", 1.5f, &e);
    m->desc.opacity = pn.spec("opacity", RGB(1.f), &opacity);
    m->desc.bump = pn.bump();
    bool remap = mp.FindBool("remaproughness", true);
    m->desc.remap_roughness = remap;
    if ((m->desc.textured = pn.textured)) return m;
    RGB op = opacity.Clamp();
    RGB t = (-op + RGB(1.f)).Clamp();
    if (!t.IsBlack()) {
        m->bsdf.eta = 1.f;
        add(m->bsdf, specT(t, 1.f, 1.f));
    } else
        m->bsdf.eta = e;
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
    ParamNodes pn(mp);
    RGB Kd, Ks; Float roughu, roughv;
    auto m = newMat("substrate", MI_MAT_SUBSTRATE);
    m->desc.Kd = pn.spec("Kd", RGB(.5f), &Kd); m->desc.Ks = pn.spec("Ks", RGB(.5f), &Ks);
    m->desc.uroughness = pn.flt("uroughness", .1f, &roughu); m->desc.vroughness = pn.flt("vroughness", .1f, &roughv);
    m->desc.bump = pn.bump();
    bool remap = mp.FindBool("remaproughness", true);
    m->desc.remap_roughness = remap;
    if ((m->desc.textured = pn.textured)) return m;
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
    ParamNodes pn(mp);
    RGB Kd, Ks, reflect, transmit; Float rough;
    Float eta = 1.5f;
    auto m = newMat("translucent", MI_MAT_TRANSLUCENT, eta);
    m->desc.Kd = pn.spec("Kd", RGB(0.25f), &Kd); m->desc.Ks = pn.spec("Ks", RGB(0.25f), &Ks);
    m->desc.reflect = pn.spec("reflect", RGB(0.5f), &reflect); m->desc.transmit = pn.spec("transmit", RGB(0.5f), &transmit);
    m->desc.roughness = pn.flt("roughness", .1f, &rough);
    m->desc.bump = pn.bump();
    bool remap = mp.FindBool("remaproughness", true);
    m->desc.remap_roughness = remap;
    if ((m->desc.textured = pn.textured)) return m;
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
    ParamNodes pn(mp);
    RGB amount;
    // MixMaterial builds on m1's BSDF (its eta) and wraps every lobe in ScaledBxDF
    auto m = newMat("mix", MI_MAT_MIX, m1 ? m1->bsdf.eta : 1.f);
    m->desc.amount = pn.spec("amount", RGB(0.5f), &amount);
    if (!m1 || !m2) { Error("mix material needs two non-null materials"); return m; }
    m->m1 = m1; m->m2 = m2;
    if ((m->desc.textured = (pn.textured || m1->desc.textured || m2->desc.textured))) return m;
    RGB s1 = amount.Clamp();
    RGB s2 = (RGB(1.f) - s1).Clamp();
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
    else if (name == "subsurface") material = CreateSubsurface(mp, false);
    else if (name == "kdsubsurface") material = CreateSubsurface(mp, true);
    else if (name == "hair" || name == "disney" || name == "fourier") {
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
