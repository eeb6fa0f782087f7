// FlattenScene: the reference's own objects -> mi_scene_desc (include/pbrt_amd.h).  The hand-over half of the reference-side binding of
// INTEGRATION.md s.2, shared by integration/wavefrontpath.cpp (the binding a pbrt-v3 maintainer adds: renders the description on the MI355X
// through the mi_* entry points) and by the repository's own test harness, which renders the SAME description with its CPU checker to prove
// the hand-over table of INTEGRATION.md s.1 on a machine without a GPU.  COMPILED AGAINST THE UNMODIFIED REFERENCE (libpbrt_ref.a).
//
// Access to private members: the hand-over needs BVHAccel::nodes / primitives, GeometricPrimitive's members, Triangle::mesh, the cameras'
// matrices, the BxDFs' parameters ...  INTEGRATION.md lists the `friend class WavefrontPathIntegrator;` lines a maintainer adds to the
// reference's headers; a build against the UNMODIFIED headers compiles the including translation unit with g++'s -fno-access-control
// instead (object layout does not depend on access labels).  No reference source is modified or copied.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <functional>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <tuple>
#include <vector>

#include "pbrt.h"
#include "accelerators/bvh.h"
#include "api.h"
#include "camera.h"
#include "cameras/perspective.h"
#include "film.h"
#include "filter.h"
#include "integrator.h"
#include "interaction.h"
#include "light.h"
#include "lights/diffuse.h"
#include "lights/distant.h"
#include "lights/infinite.h"
#include "lights/point.h"
#include "lights/spot.h"
#include "mipmap.h"
#include "sampling.h"
#include "material.h"
#include "memory.h"
#include "microfacet.h"
#include "paramset.h"
#include "primitive.h"
#include "reflection.h"
#include "sampler.h"
#include "samplers/halton.h"
#include "samplers/random.h"
#include "samplers/sobol.h"
#include "samplers/stratified.h"
#include "samplers/zerotwosequence.h"
#include "samplers/maxmin.h"
#include "scene.h"
#include "shapes/sphere.h"
#include "shapes/triangle.h"
#include "texture.h"
#include "textures/bilerp.h"
#include "textures/checkerboard.h"
#include "textures/constant.h"
#include "textures/dots.h"
#include "textures/fbm.h"
#include "textures/imagemap.h"
#include "textures/marble.h"
#include "textures/mix.h"
#include "textures/scale.h"
#include "textures/uv.h"
#include "textures/windy.h"
#include "textures/wrinkled.h"
#include "materials/glass.h"
#include "materials/kdsubsurface.h"
#include "materials/matte.h"
#include "materials/metal.h"
#include "materials/mirror.h"
#include "materials/mixmat.h"
#include "materials/plastic.h"
#include "materials/subsurface.h"
#include "materials/substrate.h"
#include "materials/translucent.h"
#include "materials/uber.h"
#include "bssrdf.h"
#include "integrators/path.h"
#include "integrators/volpath.h"
#include "medium.h"
#include "media/homogeneous.h"
#include "media/grid.h"

#include "pbrt_amd.h"

namespace pbrt {

// LinearBVHNode is private to accelerators/bvh.cpp (:95-104); mi_bvh2_node is the same 32-byte record
static_assert(sizeof(mi_bvh2_node) == 32, "LinearBVHNode layout");

namespace {
struct Flat {
    mi_scene_desc desc;
    std::vector<float> P, N, UV;
    std::vector<uint32_t> triIndices, triMesh;
    std::vector<int32_t> triLight;
    std::vector<mi_mesh> meshes;
    std::vector<mi_material> materials;
    std::vector<mi_light> lights;
    std::vector<mi_sphere> spheres;
    std::vector<float> lightFunc, lightCdf;
    // row f2 / f4: texture nodes, image pyramids, per-material parameter nodes, alpha masks, BSSRDF tables -- only filled when the scene needs them
    std::vector<mi_texture> textures;
    std::vector<mi_image> images;
    // the reference's Texture objects behind the nodes (PBRT_AMD_TEX_PROBE: Texture::Evaluate of the object against the backend's evaluation of the node)
    std::vector<std::pair<const Texture<Float> *, int>> probeFloat;
    std::vector<std::pair<const Texture<Spectrum> *, int>> probeSpectrum;
    std::vector<std::vector<float>> imageKeep;
    std::vector<mi_material_desc> descs;
    std::vector<int32_t> meshAlpha;
    std::vector<mi_bssrdf_table> bssrdfTables;
    std::vector<mi_bssrdf_desc> bssrdfDescs;
    bool needDescs = false, anyAlpha = false, anyBssrdf = false;
    std::vector<mi_instance> instances;     // two-level instancing: TransformedPrimitives of the top-level BVH, the objects' own BVHAccels behind it
    std::vector<mi_object> objects;
    std::vector<mi_bvh2_node> nodes;        // top-level LinearBVHNodes followed by the objects' (only built when the scene has instances)
    std::vector<mi_envmap> envmaps;         // InfiniteAreaLight: level-0 texels of Lmap + its Distribution2D, as the reference built them
    std::vector<std::vector<float>> envKeep;
    std::vector<mi_medium> media;           // row f4: the Medium objects reachable from the primitives' MediumInterfaces and the camera
    std::vector<int32_t> meshMedium;        // 2 per mi_mesh entry: inside, outside (-1: none)
    bool anyInterface = false;
    std::string error;
};

void copyM(float dst[16], const Matrix4x4 &m) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) dst[4 * r + c] = m.m[r][c]; }
void rgb3(float d[3], const Spectrum &s) { Float c[3]; s.ToRGB(c); d[0] = c[0]; d[1] = c[1]; d[2] = c[2]; }

// one BxDF object of the reference -> the POD the library evaluates (include/pbrt_amd.h mi_bxdf)
bool convertBxDF(const BxDF *b, mi_bxdf *out, std::string *err) {
    std::memset(out, 0, sizeof(*out));
    for (int i = 0; i < 3; ++i) out->scale[i] = 1;
    auto distrib = [&](const MicrofacetDistribution *d) -> bool {
        if (auto tr = dynamic_cast<const TrowbridgeReitzDistribution *>(d)) { out->distrib = 0; out->alphax = tr->alphax; out->alphay = tr->alphay; return true; }
        if (auto bk = dynamic_cast<const BeckmannDistribution *>(d)) { out->distrib = 1; out->alphax = bk->alphax; out->alphay = bk->alphay; return true; }
        *err = "unknown MicrofacetDistribution";
        return false;
    };
    auto fresnel = [&](const Fresnel *f) -> bool {
        if (auto fd = dynamic_cast<const FresnelDielectric *>(f)) { out->fresnel = MI_FRESNEL_DIELECTRIC; out->etaA = fd->etaI; out->etaB = fd->etaT; return true; }
        if (auto fc = dynamic_cast<const FresnelConductor *>(f)) {
            out->fresnel = MI_FRESNEL_CONDUCTOR;
            rgb3(out->eta_c, fc->etaT); rgb3(out->k_c, fc->k);
            Float ei[3]; fc->etaI.ToRGB(ei);
            if (ei[0] != 1 || ei[1] != 1 || ei[2] != 1) { *err = "FresnelConductor with etaI != 1"; return false; }
            return true;
        }
        if (dynamic_cast<const FresnelNoOp *>(f)) { out->fresnel = MI_FRESNEL_NOOP; return true; }
        *err = "unknown Fresnel";
        return false;
    };
    if (auto s = dynamic_cast<const ScaledBxDF *>(b)) {   // mixmat.cpp:57-63
        if (!convertBxDF(s->bxdf, out, err)) return false;
        float sc[3];
        rgb3(sc, s->scale);
        if (out->scaled) for (int i = 0; i < 3; ++i) out->scale[i] = sc[i] * out->scale[i];
        else { out->scaled = 1; for (int i = 0; i < 3; ++i) out->scale[i] = sc[i]; }
        return true;
    }
    if (auto l = dynamic_cast<const LambertianReflection *>(b)) { out->type = MI_BXDF_LAMBERT_R; rgb3(out->R, l->R); return true; }
    if (auto l = dynamic_cast<const LambertianTransmission *>(b)) { out->type = MI_BXDF_LAMBERT_T; rgb3(out->T, l->T); return true; }
    if (auto o = dynamic_cast<const OrenNayar *>(b)) { out->type = MI_BXDF_OREN_NAYAR; rgb3(out->R, o->R); out->A = o->A; out->B = o->B; return true; }
    if (auto s = dynamic_cast<const SpecularReflection *>(b)) { out->type = MI_BXDF_SPECULAR_R; rgb3(out->R, s->R); return fresnel(s->fresnel); }
    if (auto s = dynamic_cast<const SpecularTransmission *>(b)) {
        out->type = MI_BXDF_SPECULAR_T; rgb3(out->T, s->T); out->etaA = s->etaA; out->etaB = s->etaB; out->fresnel = MI_FRESNEL_DIELECTRIC;
        return true;
    }
    if (auto s = dynamic_cast<const FresnelSpecular *>(b)) { out->type = MI_BXDF_FRESNEL_SPEC; rgb3(out->R, s->R); rgb3(out->T, s->T); out->etaA = s->etaA; out->etaB = s->etaB; return true; }
    if (auto m = dynamic_cast<const MicrofacetReflection *>(b)) { out->type = MI_BXDF_MICROFACET_R; rgb3(out->R, m->R); return distrib(m->distribution) && fresnel(m->fresnel); }
    if (auto m = dynamic_cast<const MicrofacetTransmission *>(b)) {
        out->type = MI_BXDF_MICROFACET_T; rgb3(out->T, m->T); out->etaA = m->etaA; out->etaB = m->etaB; out->fresnel = MI_FRESNEL_DIELECTRIC;
        return distrib(m->distribution);
    }
    if (auto f = dynamic_cast<const FresnelBlend *>(b)) { out->type = MI_BXDF_FRESNEL_BLEND; rgb3(out->R, f->Rd); rgb3(out->T, f->Rs); return distrib(f->distribution); }
    *err = "BxDF class without a device counterpart: " + b->ToString();
    return false;
}

// the lobe list Material::ComputeScatteringFunctions builds (materials/*.cpp) -- constant textures: independent of the interaction
bool convertMaterial(const Material *m, mi_material *out, std::string *err) {
    std::memset(out, 0, sizeof(*out));
    out->eta = 1;
    MemoryArena arena;
    SurfaceInteraction si(Point3f(0, 0, 0), Vector3f(0, 0, 0), Point2f(0.5f, 0.5f), Vector3f(0, 0, 1), Vector3f(1, 0, 0), Vector3f(0, 1, 0),
                          Normal3f(0, 0, 0), Normal3f(0, 0, 0), 0, nullptr);
    m->ComputeScatteringFunctions(&si, arena, TransportMode::Radiance, true);
    if (!si.bsdf) { *err = "material without a BSDF"; return false; }
    if (si.bssrdf) { *err = "material with a BSSRDF (not handed over by this stub)"; return false; }
    out->eta = si.bsdf->eta;
    out->n_bxdfs = si.bsdf->nBxDFs;
    for (int i = 0; i < si.bsdf->nBxDFs; ++i)
        if (!convertBxDF(si.bsdf->bxdfs[i], &out->bxdfs[i], err)) return false;
    return true;
}

// ---- Texture<Float> / Texture<Spectrum> objects -> mi_texture nodes (children before parents), MIPMaps -> mi_image pyramids
struct TexWalker {
    Flat *fs;
    std::map<const void *, int> node;            // Texture object -> node index
    std::map<const void *, int> image;           // MIPMap object -> image index
    std::string error;
    static mi_texture blank(int type, bool spectrum) {
        mi_texture t;
        std::memset(&t, 0, sizeof(t));
        t.type = type; t.spectrum = spectrum ? 1 : 0;
        t.tex1 = t.tex2 = t.amount = t.image = -1;
        t.su = t.sv = 1;
        return t;
    }
    static void val3(float d[3], Float v) { d[0] = d[1] = d[2] = v; }
    static void val3(float d[3], const Spectrum &v) { rgb3(d, v); }
    bool map2(mi_texture &t, const TextureMapping2D *m) {
        if (auto uv = dynamic_cast<const UVMapping2D *>(m)) { t.mapping = MI_MAP_UV; t.su = uv->su; t.sv = uv->sv; t.du = uv->du; t.dv = uv->dv; }
        else if (auto sp = dynamic_cast<const SphericalMapping2D *>(m)) { t.mapping = MI_MAP_SPHERICAL; copyM(t.w2t, sp->WorldToTexture.GetMatrix()); }
        else if (auto cy = dynamic_cast<const CylindricalMapping2D *>(m)) { t.mapping = MI_MAP_CYLINDRICAL; copyM(t.w2t, cy->WorldToTexture.GetMatrix()); }
        else if (auto pl = dynamic_cast<const PlanarMapping2D *>(m)) {
            t.mapping = MI_MAP_PLANAR;
            for (int c = 0; c < 3; ++c) { t.vs[c] = pl->vs[c]; t.vt[c] = pl->vt[c]; }
            t.du = pl->ds; t.dv = pl->dt;
        } else { error = "unknown TextureMapping2D"; return false; }
        return true;
    }
    bool map3(mi_texture &t, const TextureMapping3D *m) {
        auto id = dynamic_cast<const IdentityMapping3D *>(m);
        if (!id) { error = "unknown TextureMapping3D"; return false; }
        t.mapping = MI_MAP_IDENTITY3D; copyM(t.w2t, id->WorldToTexture.GetMatrix());
        return true;
    }
    template <class TM> int mip(const MIPMap<TM> *mm, int channels) {
        auto it = image.find(mm);
        if (it != image.end()) return it->second;
        mi_image im;
        std::memset(&im, 0, sizeof(im));
        im.width = mm->resolution[0]; im.height = mm->resolution[1]; im.levels = (int)mm->pyramid.size(); im.channels = channels;
        im.trilinear = mm->doTrilinear ? 1 : 0; im.wrap = mm->wrapMode == ImageWrap::Repeat ? 0 : (mm->wrapMode == ImageWrap::Black ? 1 : 2);
        im.max_aniso = mm->maxAnisotropy;
        std::vector<float> tex;
        for (int l = 0; l < im.levels; ++l) {
            const auto &lv = *mm->pyramid[l];
            for (int t = 0; t < lv.vSize(); ++t) for (int sx = 0; sx < lv.uSize(); ++sx) pushTexel(tex, lv(sx, t));
        }
        fs->imageKeep.push_back(std::move(tex));
        im.texels = fs->imageKeep.back().data();
        int idx = (int)fs->images.size();
        fs->images.push_back(im);
        image[mm] = idx;
        return idx;
    }
    static void pushTexel(std::vector<float> &v, Float x) { v.push_back(x); }
    static void pushTexel(std::vector<float> &v, const RGBSpectrum &x) { Float c[3]; x.ToRGB(c); v.push_back(c[0]); v.push_back(c[1]); v.push_back(c[2]); }
    int add(const void *key, const mi_texture &t) { int i = (int)fs->textures.size(); fs->textures.push_back(t); node[key] = i; return i; }
    void remember(const Texture<Float> *t, int i) { fs->probeFloat.emplace_back(t, i); }
    void remember(const Texture<Spectrum> *t, int i) { fs->probeSpectrum.emplace_back(t, i); }
    template <class T> int walk(const Texture<T> *tex) {
        if (!tex) return -1;
        auto it = node.find(tex);
        if (it != node.end()) return it->second;
        int i = walkNew(tex);
        if (i >= 0) remember(tex, i);
        return i;
    }
    template <class T> int walkNew(const Texture<T> *tex) {
        const bool S = std::is_same<T, Spectrum>::value;
        if (auto c = dynamic_cast<const ConstantTexture<T> *>(tex)) { mi_texture t = blank(MI_TEX_CONSTANT, S); val3(t.value, c->value); return add(tex, t); }
        if (auto sc = dynamic_cast<const ScaleTexture<T, T> *>(tex)) { mi_texture t = blank(MI_TEX_SCALE, S); t.tex1 = walk(sc->tex1.get()); t.tex2 = walk(sc->tex2.get()); return add(tex, t); }
        if (auto mx = dynamic_cast<const MixTexture<T> *>(tex)) {
            mi_texture t = blank(MI_TEX_MIX, S);
            t.tex1 = walk(mx->tex1.get()); t.tex2 = walk(mx->tex2.get()); t.amount = walk(mx->amount.get());
            return add(tex, t);
        }
        if (auto bl = dynamic_cast<const BilerpTexture<T> *>(tex)) {
            mi_texture t = blank(MI_TEX_BILERP, S);
            if (!map2(t, bl->mapping.get())) return -1;
            val3(t.v00, bl->v00); val3(t.v01, bl->v01); val3(t.v10, bl->v10); val3(t.v11, bl->v11);
            return add(tex, t);
        }
        if (auto c2 = dynamic_cast<const Checkerboard2DTexture<T> *>(tex)) {
            mi_texture t = blank(MI_TEX_CHECKERBOARD, S);
            t.tex1 = walk(c2->tex1.get()); t.tex2 = walk(c2->tex2.get()); t.dim = 2; t.aa = c2->aaMethod == AAMethod::None ? 0 : 1;
            if (!map2(t, c2->mapping.get())) return -1;
            return add(tex, t);
        }
        if (auto c3 = dynamic_cast<const Checkerboard3DTexture<T> *>(tex)) {
            mi_texture t = blank(MI_TEX_CHECKERBOARD, S);
            t.tex1 = walk(c3->tex1.get()); t.tex2 = walk(c3->tex2.get()); t.dim = 3;
            if (!map3(t, c3->mapping.get())) return -1;
            return add(tex, t);
        }
        if (auto dt = dynamic_cast<const DotsTexture<T> *>(tex)) {
            mi_texture t = blank(MI_TEX_DOTS, S);
            if (!map2(t, dt->mapping.get())) return -1;
            t.tex1 = walk(dt->outsideDot.get()); t.tex2 = walk(dt->insideDot.get());
            return add(tex, t);
        }
        if (auto fb = dynamic_cast<const FBmTexture<T> *>(tex)) { mi_texture t = blank(MI_TEX_FBM, S); if (!map3(t, fb->mapping.get())) return -1; t.octaves = fb->octaves; t.omega = fb->omega; return add(tex, t); }
        if (auto wr = dynamic_cast<const WrinkledTexture<T> *>(tex)) { mi_texture t = blank(MI_TEX_WRINKLED, S); if (!map3(t, wr->mapping.get())) return -1; t.octaves = wr->octaves; t.omega = wr->omega; return add(tex, t); }
        if (auto wd = dynamic_cast<const WindyTexture<T> *>(tex)) { mi_texture t = blank(MI_TEX_WINDY, S); if (!map3(t, wd->mapping.get())) return -1; return add(tex, t); }
        return walkTyped(tex);
    }
    int walkTyped(const Texture<Float> *tex) {
        if (auto im = dynamic_cast<const ImageTexture<Float, Float> *>(tex)) {
            mi_texture t = blank(MI_TEX_IMAGEMAP, false);
            if (!map2(t, im->mapping.get())) return -1;
            t.image = mip(im->mipmap, 1);
            return add(tex, t);
        }
        error = "Texture<Float> class without a device counterpart";
        return -1;
    }
    int walkTyped(const Texture<Spectrum> *tex) {
        if (auto im = dynamic_cast<const ImageTexture<RGBSpectrum, Spectrum> *>(tex)) {
            mi_texture t = blank(MI_TEX_IMAGEMAP, true);
            if (!map2(t, im->mapping.get())) return -1;
            t.image = mip(im->mipmap, 3);
            return add(tex, t);
        }
        if (auto uv = dynamic_cast<const UVTexture *>(tex)) { mi_texture t = blank(MI_TEX_UV, true); if (!map2(t, uv->mapping.get())) return -1; return add(tex, t); }
        if (auto mb = dynamic_cast<const MarbleTexture *>(tex)) {
            mi_texture t = blank(MI_TEX_MARBLE, true);
            if (!map3(t, mb->mapping.get())) return -1;
            t.octaves = mb->octaves; t.omega = mb->omega; t.scale = mb->scale; t.variation = mb->variation;
            return add(tex, t);
        }
        error = "Texture<Spectrum> class without a device counterpart";
        return -1;
    }
};
template <class T> bool isConstant(const std::shared_ptr<Texture<T>> &t) { return !t || dynamic_cast<const ConstantTexture<T> *>(t.get()) != nullptr; }

// Material object -> mi_material_desc (the parameter nodes Material::ComputeScatteringFunctions evaluates per hit, materials/*.cpp).  `textured` = some
// parameter is not a ConstantTexture or there is a bump map; a material with a BSSRDF is always built per hit.  Returns false for unknown classes.
bool describeMaterial(const Material *m, TexWalker &tw, const std::function<int32_t(const Material *)> &slotOf, mi_material_desc *d, mi_bssrdf_desc *b, Flat *fs) {
    std::memset(d, 0xff, sizeof(*d));
    d->textured = 0; d->remap_roughness = 0; d->pad = 0;
    std::memset(b, 0, sizeof(*b));
    bool allConst = true;
    auto S = [&](const std::shared_ptr<Texture<Spectrum>> &t) { allConst &= isConstant(t); return tw.walk(t.get()); };
    auto F = [&](const std::shared_ptr<Texture<Float>> &t) { allConst &= isConstant(t); return tw.walk(t.get()); };
    auto bump = [&](const std::shared_ptr<Texture<Float>> &t) { if (t) allConst = false; return tw.walk(t.get()); };
    if (auto x = dynamic_cast<const MatteMaterial *>(m)) { d->type = MI_MAT_MATTE; d->Kd = S(x->Kd); d->sigma = F(x->sigma); d->bump = bump(x->bumpMap); }
    else if (auto x = dynamic_cast<const PlasticMaterial *>(m)) { d->type = MI_MAT_PLASTIC; d->Kd = S(x->Kd); d->Ks = S(x->Ks); d->roughness = F(x->roughness); d->bump = bump(x->bumpMap); d->remap_roughness = x->remapRoughness; }
    else if (auto x = dynamic_cast<const GlassMaterial *>(m)) {
        d->type = MI_MAT_GLASS; d->Kr = S(x->Kr); d->Kt = S(x->Kt); d->eta_f = F(x->index); d->uroughness = F(x->uRoughness); d->vroughness = F(x->vRoughness);
        d->bump = bump(x->bumpMap); d->remap_roughness = x->remapRoughness;
    } else if (auto x = dynamic_cast<const MirrorMaterial *>(m)) { d->type = MI_MAT_MIRROR; d->Kr = S(x->Kr); d->bump = bump(x->bumpMap); }
    else if (auto x = dynamic_cast<const MetalMaterial *>(m)) {
        d->type = MI_MAT_METAL; d->eta_s = S(x->eta); d->k_s = S(x->k); d->roughness = F(x->roughness); d->uroughness = F(x->uRoughness); d->vroughness = F(x->vRoughness);
        d->bump = bump(x->bumpMap); d->remap_roughness = x->remapRoughness;
    } else if (auto x = dynamic_cast<const UberMaterial *>(m)) {
        d->type = MI_MAT_UBER; d->Kd = S(x->Kd); d->Ks = S(x->Ks); d->Kr = S(x->Kr); d->Kt = S(x->Kt); d->opacity = S(x->opacity);
        d->roughness = F(x->roughness); d->uroughness = F(x->roughnessu); d->vroughness = F(x->roughnessv); d->eta_f = F(x->eta);
        d->bump = bump(x->bumpMap); d->remap_roughness = x->remapRoughness;
    } else if (auto x = dynamic_cast<const SubstrateMaterial *>(m)) {
        d->type = MI_MAT_SUBSTRATE; d->Kd = S(x->Kd); d->Ks = S(x->Ks); d->uroughness = F(x->nu); d->vroughness = F(x->nv); d->bump = bump(x->bumpMap); d->remap_roughness = x->remapRoughness;
    } else if (auto x = dynamic_cast<const TranslucentMaterial *>(m)) {
        d->type = MI_MAT_TRANSLUCENT; d->Kd = S(x->Kd); d->Ks = S(x->Ks); d->roughness = F(x->roughness); d->reflect = S(x->reflect); d->transmit = S(x->transmit);
        d->bump = bump(x->bumpMap); d->remap_roughness = x->remapRoughness;
    } else if (auto x = dynamic_cast<const MixMaterial *>(m)) {
        d->type = MI_MAT_MIX; d->amount = S(x->scale);
        d->m1 = slotOf(x->m1.get()); d->m2 = slotOf(x->m2.get());
        if (d->m1 < 0 || d->m2 < 0) return false;
        allConst = allConst && !fs->descs[d->m1].textured && !fs->descs[d->m2].textured;
    } else if (auto x = dynamic_cast<const SubsurfaceMaterial *>(m)) {   // the BSDF half is GlassMaterial's record; + the TabulatedBSSRDF inputs
        d->type = MI_MAT_GLASS; d->Kr = S(x->Kr); d->Kt = S(x->Kt); d->uroughness = F(x->uRoughness); d->vroughness = F(x->vRoughness); d->bump = bump(x->bumpMap);
        d->remap_roughness = x->remapRoughness;
        mi_texture e = TexWalker::blank(MI_TEX_CONSTANT, false); TexWalker::val3(e.value, x->eta);
        d->eta_f = (int32_t)fs->textures.size(); fs->textures.push_back(e);
        b->kind = MI_BSSRDF_SUBSURFACE; b->sigma_a = S(x->sigma_a); b->sigma_s = S(x->sigma_s); b->Kd = b->mfp = -1; b->scale = x->scale; b->eta = x->eta;
        const BSSRDFTable &tb = x->table;
        mi_bssrdf_table mt = {tb.nRhoSamples, tb.nRadiusSamples, tb.rhoSamples.get(), tb.radiusSamples.get(), tb.profile.get(), tb.rhoEff.get(), tb.profileCDF.get()};
        b->table = (int32_t)fs->bssrdfTables.size(); fs->bssrdfTables.push_back(mt);
        allConst = false; fs->anyBssrdf = true;
    } else if (auto x = dynamic_cast<const KdSubsurfaceMaterial *>(m)) {
        d->type = MI_MAT_GLASS; d->Kr = S(x->Kr); d->Kt = S(x->Kt); d->uroughness = F(x->uRoughness); d->vroughness = F(x->vRoughness); d->bump = bump(x->bumpMap);
        d->remap_roughness = x->remapRoughness;
        mi_texture e = TexWalker::blank(MI_TEX_CONSTANT, false); TexWalker::val3(e.value, x->eta);
        d->eta_f = (int32_t)fs->textures.size(); fs->textures.push_back(e);
        b->kind = MI_BSSRDF_KDSUBSURFACE; b->Kd = S(x->Kd); b->mfp = S(x->mfp); b->sigma_a = b->sigma_s = -1; b->scale = x->scale; b->eta = x->eta;
        const BSSRDFTable &tb = x->table;
        mi_bssrdf_table mt = {tb.nRhoSamples, tb.nRadiusSamples, tb.rhoSamples.get(), tb.radiusSamples.get(), tb.profile.get(), tb.rhoEff.get(), tb.profileCDF.get()};
        b->table = (int32_t)fs->bssrdfTables.size(); fs->bssrdfTables.push_back(mt);
        allConst = false; fs->anyBssrdf = true;
    } else
        return false;
    d->textured = allConst ? 0 : 1;
    return tw.error.empty();
}

uint32_t countNodes(const mi_bvh2_node *n) {   // the flattened array's length is not stored (bvh.cpp:222-229): walk it
    uint32_t maxIdx = 0;
    std::vector<uint32_t> todo{0};
    while (!todo.empty()) {
        uint32_t i = todo.back(); todo.pop_back();
        maxIdx = std::max(maxIdx, i);
        if (n[i].n_prims == 0) { todo.push_back(i + 1); todo.push_back((uint32_t)n[i].offset); }
    }
    return maxIdx + 1;
}

std::unique_ptr<Flat> FlattenScene(const Scene &scene, const Camera &cam, Sampler &sampler, int maxDepth, Float rrThreshold, const Bounds2i &pixelBounds,
                                   const std::string &lightStrategy, bool volpath) {
    std::unique_ptr<Flat> fs(new Flat);
    auto fail = [&](const std::string &m) { fs->error = m; return std::move(fs); };
    const BVHAccel *bvh = dynamic_cast<const BVHAccel *>(scene.aggregate.get());
    if (!bvh) return fail("the aggregate is not a BVHAccel");
    // --- media (row f4): Medium* -> index into mi_scene_desc::media.  HomogeneousMedium (media/homogeneous.h:49-58) and GridDensityMedium
    // (media/grid.h:55-78) hand over their constructor results; the density grid stays where the reference keeps it (host pointer).
    std::map<const Medium *, int32_t> mediumIndex;
    std::string mediumError;
    auto mediumOf = [&](const Medium *m) -> int32_t {
        if (!m) return -1;
        auto it = mediumIndex.find(m);
        if (it != mediumIndex.end()) return it->second;
        mi_medium mm;
        std::memset(&mm, 0, sizeof(mm));
        if (auto h = dynamic_cast<const HomogeneousMedium *>(m)) {
            mm.type = MI_MEDIUM_HOMOGENEOUS;
            rgb3(mm.sigma_a, h->sigma_a); rgb3(mm.sigma_s, h->sigma_s); rgb3(mm.sigma_t, h->sigma_t);
            mm.g = h->g;
        } else if (auto gm = dynamic_cast<const GridDensityMedium *>(m)) {
            mm.type = MI_MEDIUM_GRID;
            rgb3(mm.sigma_a, gm->sigma_a); rgb3(mm.sigma_s, gm->sigma_s);
            mm.sigma_t[0] = mm.sigma_t[1] = mm.sigma_t[2] = gm->sigma_t;
            mm.g = gm->g; mm.nx = gm->nx; mm.ny = gm->ny; mm.nz = gm->nz;
            mm.inv_max_density = gm->invMaxDensity;
            copyM(mm.world_to_medium, gm->WorldToMedium.GetMatrix());
            mm.density = gm->density.get();
        } else { mediumError = "a Medium that is neither homogeneous nor a density grid"; return -1; }
        int32_t idx = (int32_t)fs->media.size();
        fs->media.push_back(mm);
        mediumIndex[m] = idx;
        return idx;
    };
    // --- primitives in BVHAccel::primitives order (bvh.cpp:205)
    std::map<const TriangleMesh *, uint32_t> meshIndex;
    std::map<const Material *, int32_t> materialIndex;
    std::map<const AreaLight *, int32_t> lightOfAreaLight;
    for (size_t i = 0; i < scene.lights.size(); ++i)
        if (auto al = dynamic_cast<const AreaLight *>(scene.lights[i].get())) lightOfAreaLight[al] = (int32_t)i;
    std::vector<uint32_t> meshVertexBase;
    // Every GeometricPrimitive carries its own material pointer while vertices are per mesh: a mi_mesh entry = (TriangleMesh, material)
    std::map<std::tuple<const void *, const Material *, const Medium *, const Medium *>, uint32_t> meshEntry;
    std::map<const TriangleMesh *, uint32_t> vertexBase;
    // Primitive order of the hand-over: the top-level BVHAccel's primitives, then -- two-level instancing, include/pbrt_amd.h mi_instance --
    // the primitives of every instantiated object in ITS accelerator's order (pbrtObjectInstance api.cpp:1555-1591: a BVHAccel over the
    // object's primitives, or the single primitive itself when the object has only one).
    const size_t nTop = bvh->primitives.size();
    std::vector<const Primitive *> order;
    for (size_t k = 0; k < nTop; ++k) order.push_back(bvh->primitives[k].get());
    std::map<const Primitive *, uint32_t> objectIndex;
    std::vector<const Primitive *> objectRoots;
    for (size_t k = 0; k < nTop; ++k)
        if (auto tp = dynamic_cast<const TransformedPrimitive *>(order[k])) {
            if (tp->PrimitiveToWorld.actuallyAnimated) return fail("animated instance transforms are not carried by this path");
            const Primitive *root = tp->primitive.get();
            if (!objectIndex.count(root)) { objectIndex[root] = (uint32_t)objectRoots.size(); objectRoots.push_back(root); }
        }
    if (!objectRoots.empty()) {
        uint32_t nTopNodes = bvh->nodes ? countNodes(reinterpret_cast<const mi_bvh2_node *>(bvh->nodes)) : 0;
        fs->nodes.assign(reinterpret_cast<const mi_bvh2_node *>(bvh->nodes), reinterpret_cast<const mi_bvh2_node *>(bvh->nodes) + nTopNodes);
        for (const Primitive *root : objectRoots) {
            mi_object mo;
            mo.first_prim = (uint32_t)order.size(); mo.first_node = (uint32_t)fs->nodes.size();
            if (auto ob = dynamic_cast<const BVHAccel *>(root)) {
                for (const auto &pp : ob->primitives) order.push_back(pp.get());
                mo.n_prims = (uint32_t)ob->primitives.size();
                mo.n_nodes = ob->nodes ? countNodes(reinterpret_cast<const mi_bvh2_node *>(ob->nodes)) : 0;
                fs->nodes.insert(fs->nodes.end(), reinterpret_cast<const mi_bvh2_node *>(ob->nodes), reinterpret_cast<const mi_bvh2_node *>(ob->nodes) + mo.n_nodes);
            } else {   // a one-primitive object has no accelerator in the reference: one leaf node over its bound (a conservative box test, same hits)
                order.push_back(root);
                mo.n_prims = 1; mo.n_nodes = 1;
                mi_bvh2_node leaf;
                std::memset(&leaf, 0, sizeof(leaf));
                Bounds3f b = root->WorldBound();
                for (int a = 0; a < 3; ++a) { leaf.bmin[a] = b.pMin[a]; leaf.bmax[a] = b.pMax[a]; }
                leaf.offset = 0; leaf.n_prims = 1;
                fs->nodes.push_back(leaf);
            }
            fs->objects.push_back(mo);
        }
    }
    size_t nPrims = order.size();
    fs->triIndices.resize(3 * nPrims); fs->triMesh.resize(nPrims); fs->triLight.assign(nPrims, -1);
    std::vector<int32_t> lightTri(scene.lights.size(), -1), lightSphere(scene.lights.size(), -1);
    // one slot per Material OBJECT (sub-materials of a mix before the mix): the constant lobe list where every parameter is a ConstantTexture
    // (read back from the reference's own ComputeScatteringFunctions), the parameter nodes otherwise (evaluated per hit by the backend)
    TexWalker tw;
    tw.fs = fs.get();
    std::string matError;
    std::function<int32_t(const Material *)> slotOf = [&](const Material *m) -> int32_t {
        auto it = materialIndex.find(m);
        if (it != materialIndex.end()) return it->second;
        mi_material_desc md;
        mi_bssrdf_desc bd;
        if (!describeMaterial(m, tw, slotOf, &md, &bd, fs.get())) { if (matError.empty()) matError = tw.error.empty() ? "material class without a device counterpart" : tw.error; return -1; }
        mi_material mm;
        std::memset(&mm, 0, sizeof(mm));
        mm.eta = 1;
        if (!md.textured) { std::string err; if (!convertMaterial(m, &mm, &err)) { matError = err; return -1; } }
        else fs->needDescs = true;
        int32_t idx = (int32_t)fs->materials.size();
        fs->materials.push_back(mm); fs->descs.push_back(md); fs->bssrdfDescs.push_back(bd);
        materialIndex[m] = idx;
        return idx;
    };
    for (size_t k = 0; k < nPrims; ++k) {
        if (auto tp = dynamic_cast<const TransformedPrimitive *>(order[k])) {   // TransformedPrimitive (core/primitive.h:92-117)
            if (k >= nTop) return fail("nested object instances");
            mi_instance in;
            std::memset(&in, 0, sizeof(in));
            copyM(in.i2w, tp->PrimitiveToWorld.startTransform->GetMatrix());
            copyM(in.w2i, tp->PrimitiveToWorld.startTransform->GetInverseMatrix());
            in.object = objectIndex[tp->primitive.get()];
            fs->triIndices[3 * k] = MI_PRIM_INSTANCE; fs->triIndices[3 * k + 1] = (uint32_t)fs->instances.size(); fs->triIndices[3 * k + 2] = 0;
            fs->instances.push_back(in);
            mi_mesh mm;
            mm.flags = 0; mm.material = -1;
            fs->triMesh[k] = (uint32_t)fs->meshes.size();
            fs->meshes.push_back(mm);
            fs->meshMedium.push_back(-1); fs->meshMedium.push_back(-1);
            fs->meshAlpha.push_back(-1); fs->meshAlpha.push_back(-1);
            continue;
        }
        const GeometricPrimitive *gp = dynamic_cast<const GeometricPrimitive *>(order[k]);
        if (!gp) return fail("a primitive that is neither a GeometricPrimitive nor a TransformedPrimitive");
        int32_t mat = -1;
        if (gp->material) {
            mat = slotOf(gp->material.get());
            if (mat < 0) return fail(matError.empty() ? "material class without a device counterpart" : matError);
        }
        int32_t light = -1;
        if (gp->areaLight) {
            auto it = lightOfAreaLight.find(gp->areaLight.get());
            if (it == lightOfAreaLight.end()) return fail("area light not in scene.lights");
            light = it->second;
        }
        fs->triLight[k] = light;
        if (const Triangle *tri = dynamic_cast<const Triangle *>(gp->shape.get())) {
            const TriangleMesh *mesh = tri->mesh.get();
            if (!vertexBase.count(mesh)) {
                vertexBase[mesh] = (uint32_t)(fs->P.size() / 3);
                for (int v = 0; v < mesh->nVertices; ++v) {
                    fs->P.push_back(mesh->p[v].x); fs->P.push_back(mesh->p[v].y); fs->P.push_back(mesh->p[v].z);
                    if (mesh->n) { fs->N.push_back(mesh->n[v].x); fs->N.push_back(mesh->n[v].y); fs->N.push_back(mesh->n[v].z); }
                    else { fs->N.push_back(0); fs->N.push_back(0); fs->N.push_back(0); }
                    if (mesh->uv) { fs->UV.push_back(mesh->uv[v].x); fs->UV.push_back(mesh->uv[v].y); }
                    else { fs->UV.push_back(0); fs->UV.push_back(0); }
                }
            }
            auto key = std::make_tuple((const void *)mesh, (const Material *)gp->material.get(), gp->mediumInterface.inside, gp->mediumInterface.outside);
            auto me = meshEntry.find(key);
            if (me == meshEntry.end()) {
                mi_mesh mm;
                mm.flags = (mesh->n ? MI_MESH_HAS_N : 0u) | (mesh->uv ? MI_MESH_HAS_UV : 0u) | (mesh->s ? MI_MESH_HAS_S : 0u) |
                           ((tri->reverseOrientation ^ tri->transformSwapsHandedness) ? MI_MESH_FLIP : 0u);
                mm.material = mat;
                me = meshEntry.emplace(key, (uint32_t)fs->meshes.size()).first;
                fs->meshes.push_back(mm);
                fs->meshMedium.push_back(mediumOf(gp->mediumInterface.inside)); fs->meshMedium.push_back(mediumOf(gp->mediumInterface.outside));
                fs->anyInterface |= gp->mediumInterface.inside != nullptr || gp->mediumInterface.outside != nullptr;
                fs->meshAlpha.push_back(tw.walk(mesh->alphaMask.get())); fs->meshAlpha.push_back(tw.walk(mesh->shadowAlphaMask.get()));   // TriangleMesh::alphaMask / shadowAlphaMask
                fs->anyAlpha |= mesh->alphaMask != nullptr || mesh->shadowAlphaMask != nullptr;
                if (!tw.error.empty()) return fail(tw.error);
            }
            fs->triMesh[k] = me->second;
            uint32_t vb = vertexBase[mesh];
            for (int c = 0; c < 3; ++c) fs->triIndices[3 * k + c] = vb + (uint32_t)tri->v[c];
            if (light >= 0) lightTri[light] = (int32_t)k;
        } else if (const Sphere *sp = dynamic_cast<const Sphere *>(gp->shape.get())) {
            mi_sphere ms;
            std::memset(&ms, 0, sizeof(ms));
            copyM(ms.o2w, sp->ObjectToWorld->GetMatrix()); copyM(ms.w2o, sp->WorldToObject->GetMatrix());
            ms.radius = sp->radius; ms.zmin = sp->zMin; ms.zmax = sp->zMax; ms.theta_min = sp->thetaMin; ms.theta_max = sp->thetaMax; ms.phi_max = sp->phiMax;
            ms.flags = (sp->reverseOrientation ? 1u : 0u) | (sp->transformSwapsHandedness ? 2u : 0u);
            ms.area = sp->Area();
            mi_mesh mm;
            mm.flags = (sp->reverseOrientation ^ sp->transformSwapsHandedness) ? MI_MESH_FLIP : 0u;
            mm.material = mat;
            fs->triMesh[k] = (uint32_t)fs->meshes.size();
            fs->meshes.push_back(mm);
            fs->meshMedium.push_back(mediumOf(gp->mediumInterface.inside)); fs->meshMedium.push_back(mediumOf(gp->mediumInterface.outside));
            fs->anyInterface |= gp->mediumInterface.inside != nullptr || gp->mediumInterface.outside != nullptr;
            fs->meshAlpha.push_back(-1); fs->meshAlpha.push_back(-1);
            fs->triIndices[3 * k] = MI_PRIM_SPHERE; fs->triIndices[3 * k + 1] = (uint32_t)fs->spheres.size(); fs->triIndices[3 * k + 2] = 0;
            if (light >= 0) { lightTri[light] = (int32_t)k; lightSphere[light] = (int32_t)fs->spheres.size(); }
            fs->spheres.push_back(ms);
        } else
            return fail("a shape that is neither a Triangle nor a Sphere has no device counterpart");
    }
    // --- lights in scene.lights order
    Point3f worldCenter;
    Float worldRadius;
    scene.WorldBound().BoundingSphere(&worldCenter, &worldRadius);
    std::vector<Float> power;
    for (size_t i = 0; i < scene.lights.size(); ++i) {
        const Light *L = scene.lights[i].get();
        mi_light l;
        std::memset(&l, 0, sizeof(l));
        l.world_radius = worldRadius;
        l.world_center[0] = worldCenter.x; l.world_center[1] = worldCenter.y; l.world_center[2] = worldCenter.z;
        if (auto dl = dynamic_cast<const DiffuseAreaLight *>(L)) {
            if (lightTri[i] < 0) return fail("area light without a primitive in the BVH");
            l.type = lightSphere[i] >= 0 ? MI_LIGHT_AREA_SPHERE : MI_LIGHT_AREA_TRI;
            l.tri = lightTri[i]; l.sphere = std::max(0, lightSphere[i]);
            l.two_sided = dl->twoSided ? 1 : 0;
            rgb3(l.L, dl->Lemit);
            l.area = dl->area;
        } else if (auto pl = dynamic_cast<const PointLight *>(L)) {
            l.type = MI_LIGHT_POINT; rgb3(l.L, pl->I); l.pos[0] = pl->pLight.x; l.pos[1] = pl->pLight.y; l.pos[2] = pl->pLight.z;
        } else if (auto dd = dynamic_cast<const DistantLight *>(L)) {
            l.type = MI_LIGHT_DISTANT; rgb3(l.L, dd->L); l.pos[0] = dd->wLight.x; l.pos[1] = dd->wLight.y; l.pos[2] = dd->wLight.z;
        } else if (auto sl = dynamic_cast<const SpotLight *>(L)) {   // lights/spot.h:48-60
            l.type = MI_LIGHT_SPOT; rgb3(l.L, sl->I); l.pos[0] = sl->pLight.x; l.pos[1] = sl->pLight.y; l.pos[2] = sl->pLight.z;
            const Matrix4x4 &w2l = sl->WorldToLight.GetMatrix();
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) l.frame[3 * r + c] = w2l.m[r][c];
            l.cos_total_width = sl->cosTotalWidth; l.cos_falloff_start = sl->cosFalloffStart;
        } else if (auto il = dynamic_cast<const InfiniteAreaLight *>(L)) {   // lights/infinite.h:51-76: always the real Lmap (1 x 1 for a constant light) + distribution
            l.type = MI_LIGHT_INFINITE;
            const Matrix4x4 &w2l = il->WorldToLight.GetMatrix(), &l2w = il->LightToWorld.GetMatrix();
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { l.frame[3 * r + c] = w2l.m[r][c]; l.l2w[3 * r + c] = l2w.m[r][c]; }
            const MIPMap<RGBSpectrum> &mm = *il->Lmap;
            const int w = mm.Width(), h = mm.Height();
            mi_envmap me;
            std::memset(&me, 0, sizeof(me));
            me.width = w; me.height = h;
            std::vector<float> rgb(3 * (size_t)w * h);
            for (int t = 0; t < h; ++t) for (int sx = 0; sx < w; ++sx) { Float c[3]; mm.Texel(0, sx, t).ToRGB(c); for (int k = 0; k < 3; ++k) rgb[3 * ((size_t)t * w + sx) + k] = c[k]; }
            const Distribution2D &d2 = *il->distribution;
            const int nu = 2 * w, nv = 2 * h;
            std::vector<float> cf((size_t)nu * nv), cc((size_t)(nu + 1) * nv), cfi(nv), mf(nv), mc(nv + 1);
            for (int v = 0; v < nv; ++v) {
                const Distribution1D &d1 = *d2.pConditionalV[v];
                for (int u = 0; u < nu; ++u) cf[(size_t)v * nu + u] = d1.func[u];
                for (int u = 0; u <= nu; ++u) cc[(size_t)v * (nu + 1) + u] = d1.cdf[u];
                cfi[v] = d1.funcInt;
            }
            for (int v = 0; v < nv; ++v) mf[v] = d2.pMarginal->func[v];
            for (int v = 0; v <= nv; ++v) mc[v] = d2.pMarginal->cdf[v];
            me.marg_func_int = d2.pMarginal->funcInt;
            for (auto *vec : {&rgb, &cf, &cc, &cfi, &mf, &mc}) fs->envKeep.push_back(std::move(*vec));
            const size_t base = fs->envKeep.size() - 6;
            me.rgb = fs->envKeep[base].data(); me.cond_func = fs->envKeep[base + 1].data(); me.cond_cdf = fs->envKeep[base + 2].data();
            me.cond_func_int = fs->envKeep[base + 3].data(); me.marg_func = fs->envKeep[base + 4].data(); me.marg_cdf = fs->envKeep[base + 5].data();
            fs->envmaps.push_back(me);
            l.env_map = (int32_t)fs->envmaps.size();
            Float c0[3]; mm.Texel(0, 0, 0).ToRGB(c0);
            for (int k = 0; k < 3; ++k) l.L[k] = c0[k];
        } else
            return fail("light class without a device counterpart (goniometric / projection lights)");
        power.push_back(L->Power().y());
        fs->lights.push_back(l);
    }
    // --- Distribution1D of CreateLightSampleDistribution (lightdistrib.cpp:48-84; sampling.h:55-70)
    size_t nl = fs->lights.size();
    bool spatial = lightStrategy == "spatial" && nl > 1, uniform = lightStrategy == "uniform" || nl == 1;
    fs->lightFunc.resize(nl); fs->lightCdf.resize(nl + 1);
    for (size_t i = 0; i < nl; ++i) fs->lightFunc[i] = uniform ? Float(1) : power[i];
    Float funcInt = 0;
    if (nl) {
        int n = (int)nl;
        fs->lightCdf[0] = 0;
        for (int i = 1; i < n + 1; ++i) fs->lightCdf[i] = fs->lightCdf[i - 1] + fs->lightFunc[i - 1] / n;
        funcInt = fs->lightCdf[n];
        if (funcInt == 0) for (int i = 1; i < n + 1; ++i) fs->lightCdf[i] = Float(i) / Float(n);
        else for (int i = 1; i < n + 1; ++i) fs->lightCdf[i] /= funcInt;
    }
    // --- desc
    mi_scene_desc &d = fs->desc;
    std::memset(&d, 0, sizeof(d));
    d.abi_version = MI_ABI_VERSION;
    d.n_verts = (uint32_t)(fs->P.size() / 3); d.P = fs->P.data(); d.N = fs->N.data(); d.UV = fs->UV.data();
    d.n_tris = (uint32_t)nPrims; d.tri_indices = fs->triIndices.data(); d.tri_mesh = fs->triMesh.data(); d.tri_light = fs->triLight.data();
    d.n_meshes = (uint32_t)fs->meshes.size(); d.meshes = fs->meshes.data();
    if (fs->objects.empty()) {
        d.bvh_nodes = reinterpret_cast<const mi_bvh2_node *>(bvh->nodes);   // BVHAccel::nodes as it is
        d.n_bvh_nodes = bvh->nodes ? countNodes(d.bvh_nodes) : 0;
    } else {   // top-level nodes + the objects' nodes behind them (mi_object::first_node); n_bvh_nodes = the top-level count
        d.bvh_nodes = fs->nodes.data();
        d.n_bvh_nodes = fs->objects[0].first_node;
        d.n_instances = (uint32_t)fs->instances.size(); d.instances = fs->instances.data();
        d.n_objects = (uint32_t)fs->objects.size(); d.objects = fs->objects.data();
    }
    d.n_top_prims = (uint32_t)nTop;
    d.n_materials = (uint32_t)fs->materials.size(); d.materials = fs->materials.data();
    d.n_lights = (uint32_t)nl; d.lights = fs->lights.data();
    d.light_func = fs->lightFunc.data(); d.light_cdf = fs->lightCdf.data(); d.light_func_int = funcInt;
    d.n_spheres = (uint32_t)fs->spheres.size(); d.spheres = fs->spheres.empty() ? nullptr : fs->spheres.data();
    d.n_envmaps = (uint32_t)fs->envmaps.size(); d.envmaps = fs->envmaps.empty() ? nullptr : fs->envmaps.data();
    if (fs->needDescs || fs->anyAlpha || fs->anyBssrdf) {   // textured materials / alpha masks / BSSRDFs: the node table, pyramids and per-material parameter nodes
        d.n_textures = (uint32_t)fs->textures.size(); d.textures = fs->textures.data();
        d.n_images = (uint32_t)fs->images.size(); d.images = fs->images.empty() ? nullptr : fs->images.data();
        d.material_descs = fs->descs.data();
        d.mesh_alpha = fs->anyAlpha ? fs->meshAlpha.data() : nullptr;
        if (fs->anyBssrdf) { d.n_bssrdf_tables = (uint32_t)fs->bssrdfTables.size(); d.bssrdf_tables = fs->bssrdfTables.data(); d.material_bssrdf = fs->bssrdfDescs.data(); }
    }
    // GeometricPrimitive::mediumInterface per mesh entry, Camera::medium, and which Li runs (row f4)
    d.camera_medium = mediumOf(cam.medium);
    if (!mediumError.empty()) return fail(mediumError);
    d.n_media = (uint32_t)fs->media.size(); d.media = fs->media.empty() ? nullptr : fs->media.data();
    d.mesh_medium = fs->anyInterface ? fs->meshMedium.data() : nullptr;
    d.integrator_type = volpath ? MI_INTEGRATOR_VOLPATH : MI_INTEGRATOR_PATH;
    d.integrator.light_strategy = spatial ? MI_LIGHT_STRATEGY_SPATIAL : MI_LIGHT_STRATEGY_TABLE;
    d.integrator.spatial_max_voxels = 64;
    const PerspectiveCamera *pc = dynamic_cast<const PerspectiveCamera *>(&cam);
    if (!pc) return fail("camera is not a PerspectiveCamera");
    if (pc->CameraToWorld.actuallyAnimated) return fail("animated camera transforms are not carried by this path");
    copyM(d.camera.raster_to_camera, pc->RasterToCamera.GetMatrix());
    copyM(d.camera.camera_to_world, pc->CameraToWorld.startTransform->GetMatrix());
    for (int i = 0; i < 3; ++i) { d.camera.dx_camera[i] = pc->dxCamera[i]; d.camera.dy_camera[i] = pc->dyCamera[i]; }
    d.camera.lens_radius = pc->lensRadius; d.camera.focal_distance = pc->focalDistance;
    d.camera.shutter_open = pc->shutterOpen; d.camera.shutter_close = pc->shutterClose;
    const Film &film = *cam.film;
    Bounds2i sb = film.GetSampleBounds();
    for (int i = 0; i < 2; ++i) {
        d.film.full_res[i] = film.fullResolution[i];
        d.film.crop_min[i] = film.croppedPixelBounds.pMin[i]; d.film.crop_max[i] = film.croppedPixelBounds.pMax[i];
        d.film.sample_min[i] = sb.pMin[i]; d.film.sample_max[i] = sb.pMax[i];
        d.integrator.pixel_min[i] = pixelBounds.pMin[i]; d.integrator.pixel_max[i] = pixelBounds.pMax[i];
    }
    d.film.filter_radius[0] = film.filter->radius.x; d.film.filter_radius[1] = film.filter->radius.y;
    static_assert(sizeof(film.filterTable) == sizeof(d.film.filter_table), "filter table");
    std::memcpy(d.film.filter_table, film.filterTable, sizeof(d.film.filter_table));
    d.film.max_sample_luminance = film.maxSampleLuminance;
    d.film.scale = film.scale;
    d.integrator.max_depth = maxDepth; d.integrator.rr_threshold = rrThreshold;
    // PBRT_AMD_FAST_SAMPLERS=1 (the CLI's --fast-samplers): the samplers with one random stream per tile render with the reference's SobolSampler at the same
    // sample count -- wavefront speed instead of the reference's pixel values (the library would walk each tile serially)
    Sampler *smp = &sampler;
    std::unique_ptr<SobolSampler> fastSobol;
    {
        const char *fast = std::getenv("PBRT_AMD_FAST_SAMPLERS");
        const bool tileSerial = dynamic_cast<RandomSampler *>(smp) || dynamic_cast<StratifiedSampler *>(smp) || dynamic_cast<ZeroTwoSequenceSampler *>(smp) || dynamic_cast<MaxMinDistSampler *>(smp);
        if (fast && fast[0] == '1' && tileSerial) {
            fastSobol.reset(new SobolSampler(sampler.samplesPerPixel, sb));
            smp = fastSobol.get();
            Warning("PBRT_AMD_FAST_SAMPLERS: rendering with \"sobol\" at %d spp instead of the scene's sampler.", (int)smp->samplesPerPixel);
        }
    }
    d.integrator.spp = (int32_t)smp->samplesPerPixel;
    if (auto ss = dynamic_cast<SobolSampler *>(smp)) {
        d.integrator.sampler = MI_SAMPLER_SOBOL;
        d.integrator.sobol_resolution = ss->resolution; d.integrator.sobol_log2_resolution = ss->log2Resolution;
    } else if (auto hs = dynamic_cast<HaltonSampler *>(smp)) {
        d.integrator.sampler = MI_SAMPLER_HALTON;
        for (int i = 0; i < 2; ++i) {
            d.integrator.halton_base_scales[i] = hs->baseScales[i]; d.integrator.halton_base_exponents[i] = hs->baseExponents[i];
            d.integrator.halton_mult_inverse[i] = hs->multInverse[i];
        }
        d.integrator.halton_sample_stride = hs->sampleStride;
        d.integrator.halton_sample_at_center = hs->sampleAtPixelCenter ? 1 : 0;
    } else if (dynamic_cast<RandomSampler *>(smp)) {   // ABI v11: the samplers with one PCG32 stream per tile (the library walks each tile's samples in this loop's order)
        d.integrator.sampler = MI_SAMPLER_RANDOM;
    } else if (auto st = dynamic_cast<StratifiedSampler *>(smp)) {
        d.integrator.sampler = MI_SAMPLER_STRATIFIED;
        d.integrator.pixel_sampler_dims = (int32_t)st->samples1D.size();
        d.integrator.strat_samples[0] = st->xPixelSamples; d.integrator.strat_samples[1] = st->yPixelSamples;
        d.integrator.strat_jitter = st->jitterSamples ? 1 : 0;
    } else if (auto zt = dynamic_cast<ZeroTwoSequenceSampler *>(smp)) {
        d.integrator.sampler = MI_SAMPLER_ZEROTWO;
        d.integrator.pixel_sampler_dims = (int32_t)zt->samples1D.size();
    } else if (auto mm = dynamic_cast<MaxMinDistSampler *>(smp)) {   // ABI v12: the generator matrix the reference's sampler selected (CMaxMinDist[Log2Int(spp)], maxmin.h:74-77)
        d.integrator.sampler = MI_SAMPLER_MAXMIN;
        d.integrator.pixel_sampler_dims = (int32_t)mm->samples1D.size();
        for (int i = 0; i < 32; ++i) d.integrator.maxmin_matrix[i] = mm->CPixel[i];
    } else
        return fail("sampler is none of sobol, halton, random, stratified, 02sequence, maxmindist");
    return fs;
}
}  // namespace

// The parameters integrators/path.cpp:190-213 and integrators/volpath.cpp:192-215 read, read exactly as there (+ "integer gpus").
struct WavefrontParams {
    int maxDepth;
    Bounds2i pixelBounds;
    Float rrThreshold;
    std::string lightStrategy;
    int gpus;
};
inline WavefrontParams ReadWavefrontParams(const ParamSet &params, const std::shared_ptr<const Camera> &camera) {
    WavefrontParams w;
    w.maxDepth = params.FindOneInt("maxdepth", 5);
    int np;
    const int *pb = params.FindInt("pixelbounds", &np);
    w.pixelBounds = camera->film->GetSampleBounds();
    if (pb) {
        if (np != 4) Error("Expected four values for \"pixelbounds\" parameter. Got %d.", np);
        else {
            w.pixelBounds = Intersect(w.pixelBounds, Bounds2i{{pb[0], pb[2]}, {pb[1], pb[3]}});
            if (w.pixelBounds.Area() == 0) Error("Degenerate \"pixelbounds\" specified.");
        }
    }
    w.rrThreshold = params.FindOneFloat("rrthreshold", 1.);
    w.lightStrategy = params.FindOneString("lightsamplestrategy", "spatial");
    const char *g = std::getenv("PBRT_AMD_GPUS");   // GPUs of this node to shard the image tiles over (default 1); the scene file may say "integer gpus" instead
    w.gpus = std::max(1, params.FindOneInt("gpus", g ? std::atoi(g) : 1));
    return w;
}

// FilmTilePixel{contribSum rgb, filterWeightSum} per cropped pixel (core/film.h:52-55), row-major as the device film -> the reference's own Film
inline void MergeIntoReferenceFilm(Film *film, const std::vector<float> &rgbw) {
    Bounds2i crop = film->croppedPixelBounds;
    std::unique_ptr<FilmTile> tile = film->GetFilmTile(film->GetSampleBounds());   // one tile spanning the film
    size_t k = 0;
    for (Point2i p : crop) {
        FilmTilePixel &px = tile->GetPixel(p);
        Float rgb[3] = {rgbw[4 * k], rgbw[4 * k + 1], rgbw[4 * k + 2]};
        px.contribSum = Spectrum::FromRGB(rgb);
        px.filterWeightSum = rgbw[4 * k + 3];
        ++k;
    }
    film->MergeFilmTile(std::move(tile));   // RGB -> XYZ, core/film.cpp:117-130
    film->WriteImage();                     // core/film.cpp:168-210
}

}  // namespace pbrt
