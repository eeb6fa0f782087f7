// ref_probe: links the UNMODIFIED reference objects (oracle/_ref/libpbrt_ref.a) and dumps known-answer
// vectors from the reference's own classes for tests/golden/ (tools/gen_golden.py runs it).  The ray /
// triangle sets replay the constructions of the reference's unit tests (src/tests/shapes.cpp
// Triangle.Watertight :28-129, Triangle.Reintersect :154-205, Triangle.BadCases :544-559; src/tests/sampling.cpp
// LowDiscrepancy.Sobol :120-136, Distribution1D.Discrete :231-280) with the same RNG seeds.
// Build infrastructure: includes reference headers from /root/reference, copies nothing into this repo.
#include <cstdio>
#include <functional>
#include <vector>

#include "lowdiscrepancy.h"
#include "memory.h"
#include "microfacet.h"
#include "../../include/pbrt_amd.h"   // mi_bxdf: the parameter record the vectors are written with
#include "materials/metal.h"
#include "paramset.h"
#include "pbrt.h"
#include "reflection.h"
#include "rng.h"
#include "sampling.h"
#include "samplers/sobol.h"
#include "samplers/halton.h"
#include "accelerators/bvh.h"
#include "medium.h"
#include "bssrdf.h"
#include "cameras/perspective.h"
#include "film.h"
#include "filters/box.h"
#include "api.h"
#include "lights/diffuse.h"
#include "lights/distant.h"
#include "lights/infinite.h"
#include "scene.h"
#include "lights/point.h"
#include "lights/spot.h"
#include "shapes/sphere.h"
#include "shapes/triangle.h"
#include "sobolmatrices.h"

using namespace pbrt;

static void put(FILE *f, const void *p, size_t n) { fwrite(p, 1, n, f); }
template <typename T> static void putv(FILE *f, T v) { put(f, &v, sizeof(T)); }

static Float pExp(RNG &rng, Float exp = 8.) {   // tests/shapes.cpp:18-22
    Float logu = Lerp(rng.UniformFloat(), -exp, exp);
    return std::pow(10, logu);
}

// record: p0 p1 p2 (9f) o d (6f) tmax (f) hit (i32) t b0 b1 b2 (4f) n (3f) = 24 words
static void triRecord(FILE *f, const std::shared_ptr<Shape> &tri, const Point3f v[3], const Ray &r, int32_t *count) {
    Float tHit = 0;
    SurfaceInteraction isect;
    Ray ray(r);
    bool hit = tri->Intersect(ray, &tHit, &isect, false);
    for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) putv<float>(f, v[j][k]);
    for (int k = 0; k < 3; ++k) putv<float>(f, r.o[k]);
    for (int k = 0; k < 3; ++k) putv<float>(f, r.d[k]);
    putv<float>(f, r.tMax);
    putv<int32_t>(f, hit ? 1 : 0);
    putv<float>(f, hit ? tHit : 0.f);
    // barycentrics are not exposed; recover them from uv (default uvs (0,0),(1,0),(1,1): u = b1 + b2, v = b2)
    Float b2 = hit ? isect.uv[1] : 0, b1 = hit ? isect.uv[0] - isect.uv[1] : 0;
    putv<float>(f, hit ? isect.uv[0] : 0.f); putv<float>(f, hit ? isect.uv[1] : 0.f); putv<float>(f, b1); putv<float>(f, b2);
    for (int k = 0; k < 3; ++k) putv<float>(f, hit ? isect.n[k] : 0.f);
    ++*count;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: ref_probe <outdir> [radiance map for the infinite-light records]\n"); return 1; }
    std::string dir = argv[1];
    // ---- Sobol: SobolSampleFloat(i, d) and SobolIntervalToIndex
    {
        FILE *f = fopen((dir + "/sobol_samples.bin").c_str(), "wb");
        const int dims[] = {0, 1, 2, 3, 4, 5, 17, 63, 255, 1023};
        int32_t n = 0;
        for (int d : dims)
            for (int64_t i = 0; i < 2048; i += (d < 2 ? 1 : 7)) { putv<int64_t>(f, i); putv<int32_t>(f, d); putv<float>(f, SobolSampleFloat(i, d, 0)); ++n; }
        RNG rng(3);
        for (int k = 0; k < 4000; ++k) {   // big indices
            int64_t i = (int64_t)(rng.UniformUInt32() & 0x7fffffff) << (rng.UniformUInt32() % 4);
            int d = rng.UniformUInt32() % 1024;
            putv<int64_t>(f, i); putv<int32_t>(f, d); putv<float>(f, SobolSampleFloat(i, d, 0)); ++n;
        }
        fclose(f);
        f = fopen((dir + "/sobol_index.bin").c_str(), "wb");
        for (int k = 0; k < 6000; ++k) {
            uint32_t m = 1 + rng.UniformUInt32() % 12;
            uint64_t frame = rng.UniformUInt32() % 4096;
            int px = rng.UniformUInt32() % (1u << m), py = rng.UniformUInt32() % (1u << m);
            putv<uint32_t>(f, m); putv<uint64_t>(f, frame); putv<int32_t>(f, px); putv<int32_t>(f, py);
            putv<uint64_t>(f, SobolIntervalToIndex(m, frame, Point2i(px, py)));
        }
        fclose(f);
        // the sampler class itself: Get1D/Get2D stream for a few pixels (dims 0/1 remapped)
        f = fopen((dir + "/sobol_sampler.bin").c_str(), "wb");
        Bounds2i sb(Point2i(0, 0), Point2i(400, 300));
        SobolSampler s(16, sb);
        const int px[][2] = {{0, 0}, {1, 0}, {399, 299}, {123, 45}, {256, 256 % 300}};
        for (auto &p : px) {
            s.StartPixel(Point2i(p[0], p[1]));
            int k = 0;
            do {
                putv<int32_t>(f, p[0]); putv<int32_t>(f, p[1]); putv<int32_t>(f, k++);
                for (int d = 0; d < 24; ++d) putv<float>(f, s.Get1D());
            } while (s.StartNextSample());
        }
        fclose(f);
        // HaltonSampler (pbrt's default sampler): same record layout, 6 samples per pixel (not a power of two),
        // pixels beyond the 128 x 128 repeat of the pixel-offset computation
        f = fopen((dir + "/halton_sampler.bin").c_str(), "wb");
        HaltonSampler hs(6, sb);
        const int hpx[][2] = {{0, 0}, {1, 0}, {399, 299}, {123, 45}, {256, 130}, {127, 128}};
        for (auto &p : hpx) {
            hs.StartPixel(Point2i(p[0], p[1]));
            int k = 0;
            do {
                putv<int32_t>(f, p[0]); putv<int32_t>(f, p[1]); putv<int32_t>(f, k++);
                for (int d = 0; d < 24; ++d) putv<float>(f, hs.Get1D());
            } while (hs.StartNextSample());
        }
        fclose(f);
    }
    // ---- triangles
    {
        FILE *f = fopen((dir + "/triangles.bin").c_str(), "wb");
        int32_t count = 0;
        static Transform identity;
        int indices[3] = {0, 1, 2};
        // Triangle.BadCases (known answer: miss)
        {
            Point3f p[3] = {Point3f(-1113.45459, -79.049614, -56.2431908), Point3f(-1113.45459, -87.0922699, -56.2431908),
                            Point3f(-1113.45459, -79.2090149, -56.2431908)};
            auto mesh = CreateTriangleMesh(&identity, &identity, false, 1, indices, 3, p, nullptr, nullptr, nullptr, nullptr, nullptr);
            Ray ray(Point3f(-1081.47925, 99.9999542, 87.7701111), Vector3f(-32.1072998, -183.355865, -144.607635), 0.9999);
            triRecord(f, mesh[0], p, ray, &count);
        }
        // Triangle.Reintersect-style: random triangles with coordinates 10^+-8, rays toward a sampled point,
        // then rays spawned from the hit point (expected: no re-intersection)
        for (int i = 0; i < 1000; ++i) {
            RNG rng(i);
            Point3f v[3];
            for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) v[j][k] = pExp(rng);
            if ((Cross(v[1] - v[0], v[2] - v[0]).LengthSquared()) < 1e-20) continue;
            auto triVec = CreateTriangleMesh(&identity, &identity, false, 1, indices, 3, v, nullptr, nullptr, nullptr, nullptr, nullptr);
            Point2f u(rng.UniformFloat(), rng.UniformFloat());
            Float pdf;
            Interaction pTri = triVec[0]->Sample(u, &pdf);
            Point3f o;
            for (int j = 0; j < 3; ++j) o[j] = pExp(rng);
            Ray r(o, pTri.p - o);
            triRecord(f, triVec[0], v, r, &count);
            Float tHit;
            SurfaceInteraction isect;
            if (!triVec[0]->Intersect(r, &tHit, &isect, false)) continue;
            for (int j = 0; j < 6; ++j) {
                Point2f u2(rng.UniformFloat(), rng.UniformFloat());
                Vector3f w = UniformSampleSphere(u2);
                Ray rOut = isect.SpawnRay(w);
                triRecord(f, triVec[0], v, rOut, &count);
                Point3f p2;
                for (int k = 0; k < 3; ++k) p2[k] = pExp(rng);
                rOut = isect.SpawnRayTo(p2);
                triRecord(f, triVec[0], v, rOut, &count);
            }
        }
        // Watertight-style: rays aimed exactly at vertices / edge midpoints of moderate-size triangles
        for (int i = 0; i < 1500; ++i) {
            RNG rng(5000 + i);
            Point3f v[3];
            for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) v[j][k] = Lerp(rng.UniformFloat(), -10, 10);
            if ((Cross(v[1] - v[0], v[2] - v[0]).LengthSquared()) < 1e-20) continue;
            auto triVec = CreateTriangleMesh(&identity, &identity, false, 1, indices, 3, v, nullptr, nullptr, nullptr, nullptr, nullptr);
            Point3f o;
            for (int k = 0; k < 3; ++k) o[k] = Lerp(rng.UniformFloat(), -10, 10);
            int which = rng.UniformUInt32() % 6;
            Point3f target = which < 3 ? v[which] : Point3f((v[which - 3] + v[(which - 2) % 3]) * 0.5f);
            Ray r(o, target - o);
            triRecord(f, triVec[0], v, r, &count);
            Ray r2(o, Normalize(target - o));
            triRecord(f, triVec[0], v, r2, &count);
        }
        fclose(f);
        fprintf(stderr, "ref_probe: %d triangle records\n", count);
    }
    // ---- Sphere::Intersect on the FullSphere / PartialSphere constructions of tests/shapes.cpp:376-497 (+ a transformed half):
    // the first ray toward the bounding box, then rays spawned from the hit (expected: no re-intersection)
    {
        FILE *f = fopen((dir + "/spheres.bin").c_str(), "wb");
        int32_t count = 0;
        auto rec = [&](const Transform &o2w, const Transform &w2o, Float radius, Float zMinIn, Float zMaxIn, Float phiMaxDeg, const Sphere &sp, const Ray &r) {
            Float tHit = 0;
            SurfaceInteraction isect;
            Ray ray(r);
            bool hit = sp.Intersect(ray, &tHit, &isect, false);
            for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) putv<float>(f, o2w.GetMatrix().m[a][b]);
            for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) putv<float>(f, w2o.GetMatrix().m[a][b]);
            // constructor results, recomputed with the constructor's expressions (shapes/sphere.h:50-58; the members are private)
            putv<float>(f, radius);
            putv<float>(f, Clamp(std::min(zMinIn, zMaxIn), -radius, radius));
            putv<float>(f, Clamp(std::max(zMinIn, zMaxIn), -radius, radius));
            putv<float>(f, std::acos(Clamp(std::min(zMinIn, zMaxIn) / radius, -1, 1)));
            putv<float>(f, std::acos(Clamp(std::max(zMinIn, zMaxIn) / radius, -1, 1)));
            putv<float>(f, Radians(Clamp(phiMaxDeg, 0, 360)));
            putv<float>(f, sp.Area());
            putv<int32_t>(f, (sp.reverseOrientation ? 1 : 0) | (sp.transformSwapsHandedness ? 2 : 0));
            for (int k = 0; k < 3; ++k) putv<float>(f, r.o[k]);
            for (int k = 0; k < 3; ++k) putv<float>(f, r.d[k]);
            putv<float>(f, r.tMax);
            putv<int32_t>(f, hit ? 1 : 0);
            putv<float>(f, hit ? tHit : 0.f);
            for (int k = 0; k < 3; ++k) putv<float>(f, hit ? isect.p[k] : 0.f);
            for (int k = 0; k < 3; ++k) putv<float>(f, hit ? isect.pError[k] : 0.f);
            for (int k = 0; k < 3; ++k) putv<float>(f, hit ? isect.n[k] : 0.f);
            ++count;
            return hit ? std::make_pair(true, isect) : std::make_pair(false, isect);
        };
        for (int i = 0; i < 300; ++i) {
            RNG rng(i);
            Float radius = pExp(rng, 4);
            bool partial = i >= 100;
            Float zMin = !partial || rng.UniformFloat() < 0.5 ? -radius : Lerp(rng.UniformFloat(), -radius, radius);
            Float zMax = !partial || rng.UniformFloat() < 0.5 ? radius : Lerp(rng.UniformFloat(), -radius, radius);
            Float phiMax = !partial || rng.UniformFloat() < 0.5 ? 360. : rng.UniformFloat() * 360.;
            Transform o2w;
            if (i >= 200)
                o2w = Translate(Vector3f(pExp(rng, 2), -pExp(rng, 2), pExp(rng, 1))) * Rotate(360 * rng.UniformFloat(), Vector3f(.3f, 1, -.4f)) *
                      Scale(1 + rng.UniformFloat(), (i & 1) ? -1.5f : 1.5f, 0.5f + rng.UniformFloat());
            Transform w2o = Inverse(o2w);
            Sphere sphere(&o2w, &w2o, (i % 7) == 0, radius, zMin, zMax, phiMax);
            Point3f o;
            for (int c = 0; c < 3; ++c) o[c] = pExp(rng, i >= 200 ? 3 : 8);
            Bounds3f bbox = sphere.WorldBound();
            Point3f t;
            for (int c = 0; c < 3; ++c) t[c] = rng.UniformFloat();
            Point3f p2 = bbox.Lerp(t);
            Ray r(o, p2 - o);
            if (rng.UniformFloat() < .5) r.d = Normalize(r.d);
            auto h = rec(o2w, w2o, radius, zMin, zMax, phiMax, sphere, r);
            if (!h.first) continue;
            for (int j = 0; j < 8; ++j) {
                Point2f u(rng.UniformFloat(), rng.UniformFloat());
                Vector3f w = UniformSampleSphere(u);
                if (j < 6) w = Faceforward(w, h.second.n);   // two of them may point back into the sphere
                Ray rOut = h.second.SpawnRay(w);
                rec(o2w, w2o, radius, zMin, zMax, phiMax, sphere, rOut);
            }
        }
        fclose(f);
        printf("ref_probe: %d sphere records\n", count);
    }
    // ---- BxDFs of core/reflection.{h,cpp}: f, Pdf and Sample_f of the reference classes on random directions.
    // record = mi_bxdf (the parameters, as this repository's ABI carries them) + wo wi u | f pdf | Sample_f: wi pdf f sampledType
    {
        FILE *f = fopen((dir + "/bxdfs.bin").c_str(), "wb");
        int32_t count = 0;
        RNG rng(11);
        auto rnd = [&]() { return rng.UniformFloat(); };
        auto rndDir = [&]() { return UniformSampleSphere(Point2f(rnd(), rnd())); };
        MemoryArena arena;
        for (int cfg = 0; cfg < 60; ++cfg) {
            mi_bxdf mb;
            memset(&mb, 0, sizeof(mb));
            Float rgbR[3] = {rnd(), rnd(), rnd()}, rgbT[3] = {rnd(), rnd(), rnd()};
            Spectrum R = Spectrum::FromRGB(rgbR), T = Spectrum::FromRGB(rgbT);
            for (int k = 0; k < 3; ++k) { mb.R[k] = rgbR[k]; mb.T[k] = rgbT[k]; mb.scale[k] = 1; }
            Float etaA = 1, etaB = 1.2f + rnd();
            mb.etaA = etaA; mb.etaB = etaB;
            Float ax = 0.01f + rnd() * (cfg % 3 == 0 ? 0.05f : 0.8f), ay = (cfg & 1) ? ax : 0.01f + rnd() * 0.8f;
            mb.alphax = ax; mb.alphay = ay;
            Float ce[3] = {0.2f + rnd(), 0.9f + rnd(), 1.1f + rnd()}, ck[3] = {3.9f, 2.4f + rnd(), 2.1f};
            for (int k = 0; k < 3; ++k) { mb.eta_c[k] = ce[k]; mb.k_c[k] = ck[k]; }
            Fresnel *fr = nullptr;
            int ftype = cfg % 3;   // 0 noop, 1 dielectric, 2 conductor
            if (ftype == 0) fr = ARENA_ALLOC(arena, FresnelNoOp)();
            else if (ftype == 1) fr = ARENA_ALLOC(arena, FresnelDielectric)(etaA, etaB);
            else fr = ARENA_ALLOC(arena, FresnelConductor)(Spectrum(1.f), Spectrum::FromRGB(ce), Spectrum::FromRGB(ck));
            MicrofacetDistribution *dist = ARENA_ALLOC(arena, TrowbridgeReitzDistribution)(ax, ay);
            BxDF *bx = nullptr;
            int kind = cfg % 9;
            mb.type = kind;
            switch (kind) {
            case 0: bx = ARENA_ALLOC(arena, LambertianReflection)(R); break;
            case 1: bx = ARENA_ALLOC(arena, LambertianTransmission)(T); break;
            case 2: {
                Float sigmaDeg = 5 + 60 * rnd();
                bx = ARENA_ALLOC(arena, OrenNayar)(R, sigmaDeg);
                Float sigma = Radians(sigmaDeg), sigma2 = sigma * sigma;   // the constructor's expressions (reflection.h:414-420; members are private)
                mb.A = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
                mb.B = 0.45f * sigma2 / (sigma2 + 0.09f);
                break;
            }
            case 3: bx = ARENA_ALLOC(arena, SpecularReflection)(R, fr); mb.fresnel = ftype; break;
            case 4: bx = ARENA_ALLOC(arena, SpecularTransmission)(T, etaA, etaB, TransportMode::Radiance); break;
            case 5: bx = ARENA_ALLOC(arena, FresnelSpecular)(R, T, etaA, etaB, TransportMode::Radiance); break;
            case 6: bx = ARENA_ALLOC(arena, MicrofacetReflection)(R, dist, fr); mb.fresnel = ftype; break;
            case 7: bx = ARENA_ALLOC(arena, MicrofacetTransmission)(T, dist, etaA, etaB, TransportMode::Radiance); break;
            case 8: bx = ARENA_ALLOC(arena, FresnelBlend)(R, T, dist); break;   // Rd = R, Rs = T
            }
            for (int k = 0; k < 40; ++k) {
                Vector3f wo = rndDir(), wi = rndDir();
                if (k % 11 == 3) wi.z = 0;   // grazing wi for f / Pdf (wo.z == 0 trips the reference's CHECKs in the specular Sample_f)
                if (k % 5 == 0 && kind >= 6) wi = Reflect(wo, Normalize(Vector3f(0.1f * rnd(), 0.1f * rnd(), 1)));   // near-specular pairs for the microfacet lobes
                Point2f u(rnd(), rnd());
                Spectrum fv = bx->f(wo, wi);
                Float pdf = bx->Pdf(wo, wi);
                Vector3f wis(0, 0, 0);
                Float pdfs = 0;
                BxDFType st = bx->type;
                Spectrum fs = bx->Sample_f(wo, &wis, u, &pdfs, &st);
                put(f, &mb, sizeof(mb));
                for (int c = 0; c < 3; ++c) putv<float>(f, wo[c]);
                for (int c = 0; c < 3; ++c) putv<float>(f, wi[c]);
                putv<float>(f, u[0]); putv<float>(f, u[1]);
                Float rgb[3];
                fv.ToRGB(rgb);
                for (int c = 0; c < 3; ++c) putv<float>(f, rgb[c]);
                putv<float>(f, pdf);
                for (int c = 0; c < 3; ++c) putv<float>(f, wis[c]);
                putv<float>(f, pdfs);
                fs.ToRGB(rgb);
                for (int c = 0; c < 3; ++c) putv<float>(f, rgb[c]);
                putv<int32_t>(f, (int32_t)st);
                ++count;
            }
        }
        fclose(f);
        printf("ref_probe: %d bxdf records\n", count);
    }
    // ---- Distribution1D::SampleDiscrete (sampling.h:90-100)
    {
        FILE *f = fopen((dir + "/distribution1d.bin").c_str(), "wb");
        RNG rng(9);
        for (int trial = 0; trial < 8; ++trial) {
            int n = 1 + (int)(rng.UniformUInt32() % 40);
            std::vector<Float> func(n);
            for (auto &x : func) x = rng.UniformFloat() < 0.2 ? 0.f : rng.UniformFloat() * 10;
            Distribution1D d(&func[0], n);
            putv<int32_t>(f, n);
            for (auto x : func) putv<float>(f, x);
            for (auto x : d.cdf) putv<float>(f, x);
            putv<float>(f, d.funcInt);
            putv<int32_t>(f, 64);
            for (int k = 0; k < 64; ++k) {
                Float u = k < 60 ? rng.UniformFloat() : (k - 60) / 4.f;
                Float pdf;
                int idx = d.SampleDiscrete(u, &pdf);
                putv<float>(f, u); putv<int32_t>(f, idx); putv<float>(f, pdf);
            }
        }
        fclose(f);
    }
    // ---- default copper eta / k of MetalMaterial as RGB (materials/metal.cpp:81-118)
    {
        ParamSet empty;
        std::map<std::string, std::shared_ptr<Texture<Float>>> ft;
        std::map<std::string, std::shared_ptr<Texture<Spectrum>>> st;
        TextureParams tp(empty, empty, ft, st);
        std::unique_ptr<MetalMaterial> m(CreateMetalMaterial(tp));
        SurfaceInteraction si(Point3f(0, 0, 0), Vector3f(), Point2f(), Vector3f(0, 0, 1), Vector3f(1, 0, 0), Vector3f(0, 1, 0), Normal3f(),
                              Normal3f(), 0, nullptr);
        MemoryArena arena;
        m->ComputeScatteringFunctions(&si, arena, TransportMode::Radiance, true);
        FILE *f = fopen((dir + "/copper.txt").c_str(), "w");
        fprintf(f, "%s\n", si.bsdf->ToString().c_str());
        fclose(f);
    }
    // ---- "blackbody" / "spectrum" parameters as the parser converts them (paramset.cpp:134-169 -> RGBSpectrum::FromSampled)
    {
        FILE *f = fopen((dir + "/spectra.bin").c_str(), "wb");
        auto rec = [&](int kind, const std::vector<Float> &vals) {   // kind 0: blackbody (T, scale); 1: sampled (lambda, v) pairs
            ParamSet ps;
            std::unique_ptr<Float[]> fl(new Float[vals.size()]);
            for (size_t i = 0; i < vals.size(); ++i) fl[i] = vals[i];
            if (kind == 0) ps.AddBlackbodySpectrum("s", std::move(fl), (int)vals.size());
            else ps.AddSampledSpectrum("s", std::move(fl), (int)vals.size());
            Float rgb[3];
            ps.FindOneSpectrum("s", Spectrum(0.f)).ToRGB(rgb);
            putv<int32_t>(f, kind); putv<int32_t>(f, (int32_t)vals.size());
            for (int i = 0; i < 80; ++i) putv<float>(f, i < (int)vals.size() ? vals[i] : 0.f);
            for (int k = 0; k < 3; ++k) putv<float>(f, rgb[k]);
        };
        for (Float T : {800.f, 1500.f, 2700.f, 3200.f, 4100.5f, 5000.f, 6500.f, 9000.f, 20000.f})
            for (Float sc : {1.f, 0.37f, 25.f}) rec(0, {T, sc});
        RNG rng(11);
        for (int k = 0; k < 240; ++k) {
            int n = 1 + rng.UniformUInt32() % 40;
            std::vector<Float> lam(n);
            for (int i = 0; i < n; ++i) lam[i] = 300 + 600 * (i + rng.UniformFloat() * .9f) / n;   // strictly increasing
            if (k % 3 == 1) for (int i = n - 1; i > 0; --i) std::swap(lam[i], lam[rng.UniformUInt32() % (i + 1)]);   // shuffled: FromSampled sorts
            if (k % 5 == 4) for (int i = 0; i < n; ++i) lam[i] = 450 + 100 * (lam[i] - 300) / 600;   // inside the table's range only
            std::vector<Float> v;
            for (int i = 0; i < n; ++i) { v.push_back(lam[i]); v.push_back(2 * rng.UniformFloat()); }
            rec(1, v);
        }
        fclose(f);
    }
    // ---- Light::Sample_Li / Pdf_Li (core/light.h:63-75): DiffuseAreaLight over a Triangle and over a Sphere (Shape::Sample(ref, u) / Pdf(ref, wi),
    // shapes/triangle.cpp:583-647, core/shape.cpp:57-108, shapes/sphere.cpp:325-400), PointLight, SpotLight; the shadow ray is
    // VisibilityTester's SpawnRayTo (interaction.h:72-78 -> OffsetRayOrigin, geometry.h:1440-1460).  Geometry as in tests/shapes.cpp
    // Triangle.Sampling (:210-269): random triangles in [-10, 10]^3, reference points outside that box -- plus nearby ones.
    {
        FILE *f = fopen((dir + "/light_samples.bin").c_str(), "wb");
        RNG rng(29);
        auto U = [&](Float lo, Float hi) { return lo + (hi - lo) * rng.UniformFloat(); };
        MediumInterface mi;
        int count = 0;
        for (int k = 0; k < 160; ++k) {
            int kind = k % 4;   // 0 triangle area light, 1 sphere area light, 2 point, 3 spot
            Float geom[12] = {0};
            Float Lrgb[3] = {U(.2f, 8), U(.2f, 8), U(.2f, 8)};
            int twoSided = (k / 4) % 2;
            std::shared_ptr<Light> light;
            static std::vector<std::unique_ptr<Transform>> keep;
            if (kind == 0) {
                Float range = (k % 8 < 4) ? 10.f : 1.f;
                for (int i = 0; i < 9; ++i) geom[i] = U(-range, range);
                keep.emplace_back(new Transform); Transform *id = keep.back().get();
                int idx[3] = {0, 1, 2};
                Point3f P[3] = {Point3f(geom[0], geom[1], geom[2]), Point3f(geom[3], geom[4], geom[5]), Point3f(geom[6], geom[7], geom[8])};
                auto tris = CreateTriangleMesh(id, id, false, 1, idx, 3, P, nullptr, nullptr, nullptr, nullptr, nullptr);
                light = std::make_shared<DiffuseAreaLight>(Transform(), mi, Spectrum::FromRGB(Lrgb), 1, tris[0], twoSided != 0);
            } else if (kind == 1) {
                geom[0] = U(-5, 5); geom[1] = U(-5, 5); geom[2] = U(-5, 5); geom[3] = U(.2f, 3);
                keep.emplace_back(new Transform(Translate(Vector3f(geom[0], geom[1], geom[2])))); Transform *o2w = keep.back().get();
                keep.emplace_back(new Transform(Inverse(*o2w))); Transform *w2o = keep.back().get();
                auto sph = std::make_shared<Sphere>(o2w, w2o, false, geom[3], -geom[3], geom[3], 360.f);
                light = std::make_shared<DiffuseAreaLight>(Transform(), mi, Spectrum::FromRGB(Lrgb), 1, sph, twoSided != 0);
            } else {
                for (int i = 0; i < 6; ++i) geom[i] = U(-6, 6);
                ParamSet ps;
                std::unique_ptr<Spectrum[]> I(new Spectrum[1]); I[0] = Spectrum::FromRGB(Lrgb);
                ps.AddRGBSpectrum("I", std::unique_ptr<Float[]>(new Float[3]{Lrgb[0], Lrgb[1], Lrgb[2]}), 3);
                std::unique_ptr<Point3f[]> from(new Point3f[1]); from[0] = Point3f(geom[0], geom[1], geom[2]);
                ps.AddPoint3f("from", std::move(from), 1);
                if (kind == 2) light = CreatePointLight(Transform(), nullptr, ps);
                else {
                    std::unique_ptr<Point3f[]> to(new Point3f[1]); to[0] = Point3f(geom[3], geom[4], geom[5]);
                    ps.AddPoint3f("to", std::move(to), 1);
                    geom[6] = U(10, 60); geom[7] = U(1, 9);
                    ps.AddFloat("coneangle", std::unique_ptr<Float[]>(new Float[1]{geom[6]}), 1);
                    ps.AddFloat("conedeltaangle", std::unique_ptr<Float[]>(new Float[1]{geom[7]}), 1);
                    light = CreateSpotLight(Transform(), nullptr, ps);
                }
            }
            for (int j = 0; j < 24; ++j) {
                Point3f pc(U(-10, 10), U(-10, 10), U(-10, 10));
                if (j % 3) pc[rng.UniformUInt32() % 3] = rng.UniformFloat() > .5f ? -13.f : 13.f;
                Normal3f n(0, 0, 0);
                if (j % 4) { Vector3f v = UniformSampleSphere(Point2f(rng.UniformFloat(), rng.UniformFloat())); n = Normal3f(v.x, v.y, v.z); }
                Interaction ref(pc, n, Vector3f(), Vector3f(0, 0, 1), 0, mi);
                Point2f u(rng.UniformFloat(), rng.UniformFloat());
                Vector3f wi; Float pdf = 0; VisibilityTester vis;
                Spectrum Li = light->Sample_Li(ref, u, &wi, &pdf, &vis);
                bool ok = pdf > 0 && !Li.IsBlack();
                Float rgb[3] = {0, 0, 0};
                Ray sr;
                if (pdf > 0) { Li.ToRGB(rgb); sr = vis.P0().SpawnRayTo(vis.P1()); }
                // Pdf_Li for the sampled direction and for a direction next to it (mostly still hitting the shape)
                Vector3f wi2 = pdf > 0 ? Normalize(wi + Vector3f(U(-.02f, .02f), U(-.02f, .02f), U(-.02f, .02f))) : Vector3f(0, 0, 1);
                Float pdfA = pdf > 0 ? light->Pdf_Li(ref, wi) : 0, pdfB = light->Pdf_Li(ref, wi2);
                putv<int32_t>(f, kind); putv<int32_t>(f, twoSided);
                for (int i = 0; i < 12; ++i) putv<float>(f, geom[i]);
                for (int i = 0; i < 3; ++i) putv<float>(f, Lrgb[i]);
                for (int i = 0; i < 3; ++i) putv<float>(f, pc[i]);
                for (int i = 0; i < 3; ++i) putv<float>(f, n[i]);
                putv<float>(f, u[0]); putv<float>(f, u[1]);
                for (int i = 0; i < 3; ++i) putv<float>(f, wi[i]);
                putv<float>(f, pdf);
                for (int i = 0; i < 3; ++i) putv<float>(f, rgb[i]);
                for (int i = 0; i < 3; ++i) putv<float>(f, pdf > 0 ? sr.o[i] : 0.f);
                for (int i = 0; i < 3; ++i) putv<float>(f, pdf > 0 ? sr.d[i] : 0.f);
                putv<float>(f, pdf > 0 ? sr.tMax : 0.f);
                for (int i = 0; i < 3; ++i) putv<float>(f, wi2[i]);
                putv<float>(f, pdfA); putv<float>(f, pdfB);
                putv<int32_t>(f, ok ? 1 : 0);
                ++count;
            }
        }
        fclose(f);
        printf("ref_probe: %d light-sample records\n", count);
    }
    // ---- Sampler::GetCameraSample + PerspectiveCamera::GenerateRayDifferential (core/sampler.cpp:46-52, cameras/perspective.cpp:95-144) for a few
    // camera configurations (pinhole and thin lens, non-square films, a crop window, a frame aspect ratio): pFilm, the main ray and its weight
    {
        FILE *f = fopen((dir + "/camera_rays.bin").c_str(), "wb");
        struct Cfg { Float eye[3], look[3], up[3], fov, lensr, focald, aspect; int xres, yres; Float crop[4]; int spp; };
        const Cfg cfgs[] = {
            {{0, 2.2f, -6}, {0, .8f, 0}, {0, 1, 0}, 40, 0, 1e6f, 0, 72, 48, {0, 1, 0, 1}, 4},
            {{3, 1.5f, 4}, {-.5f, .2f, 0}, {0, 1, 0}, 28.5f, .12f, 5.5f, 0, 96, 96, {0, 1, 0, 1}, 8},
            {{-2, 4, 1}, {0, 0, 0}, {0, 0, 1}, 65, .03f, 4.2f, 0, 50, 120, {.2f, .75f, .1f, .9f}, 2},
            {{0, 0, 5}, {0, 0, 0}, {.1f, 1, 0}, 90, 0, 1e6f, 2.2f, 160, 40, {0, 1, 0, 1}, 16},
        };
        int count = 0, ci = 0;
        for (const Cfg &c : cfgs) {
            ParamSet fp;
            fp.AddInt("xresolution", std::unique_ptr<int[]>(new int[1]{c.xres}), 1);
            fp.AddInt("yresolution", std::unique_ptr<int[]>(new int[1]{c.yres}), 1);
            fp.AddFloat("cropwindow", std::unique_ptr<Float[]>(new Float[4]{c.crop[0], c.crop[1], c.crop[2], c.crop[3]}), 4);
            std::unique_ptr<std::string[]> fn(new std::string[1]); fn[0] = "x.pfm";
            fp.AddString("filename", std::move(fn), 1);
            ParamSet bp;
            Film *film = CreateFilm(fp, std::unique_ptr<Filter>(CreateBoxFilter(bp)));
            ParamSet cp;
            cp.AddFloat("fov", std::unique_ptr<Float[]>(new Float[1]{c.fov}), 1);
            cp.AddFloat("lensradius", std::unique_ptr<Float[]>(new Float[1]{c.lensr}), 1);
            cp.AddFloat("focaldistance", std::unique_ptr<Float[]>(new Float[1]{c.focald}), 1);
            if (c.aspect > 0) cp.AddFloat("frameaspectratio", std::unique_ptr<Float[]>(new Float[1]{c.aspect}), 1);
            // pbrtCamera: CameraToWorld = Inverse(CTM), CTM = LookAt(...) (api.cpp:806-816)
            Transform *c2w = new Transform(Inverse(LookAt(Point3f(c.eye[0], c.eye[1], c.eye[2]), Point3f(c.look[0], c.look[1], c.look[2]), Vector3f(c.up[0], c.up[1], c.up[2]))));
            AnimatedTransform anim(c2w, 0, c2w, 1);
            PerspectiveCamera *cam = CreatePerspectiveCamera(cp, anim, film, nullptr);
            ParamSet sp;
            sp.AddInt("pixelsamples", std::unique_ptr<int[]>(new int[1]{c.spp}), 1);
            std::unique_ptr<Sampler> sampler(CreateSobolSampler(sp, film->GetSampleBounds()));
            Bounds2i sb = film->GetSampleBounds();
            RNG rng(41 + ci);
            for (int k = 0; k < 400; ++k) {
                Point2i px(sb.pMin.x + (int)(rng.UniformUInt32() % (uint32_t)(sb.pMax.x - sb.pMin.x)), sb.pMin.y + (int)(rng.UniformUInt32() % (uint32_t)(sb.pMax.y - sb.pMin.y)));
                int sn = (int)(rng.UniformUInt32() % (uint32_t)sampler->samplesPerPixel);
                sampler->StartPixel(px);
                sampler->SetSampleNumber(sn);
                CameraSample cs = sampler->GetCameraSample(px);
                RayDifferential ray;
                Float w = cam->GenerateRayDifferential(cs, &ray);
                putv<int32_t>(f, ci); putv<int32_t>(f, px.x); putv<int32_t>(f, px.y); putv<int32_t>(f, sn);
                for (int i = 0; i < 3; ++i) putv<float>(f, c.eye[i]);
                for (int i = 0; i < 3; ++i) putv<float>(f, c.look[i]);
                for (int i = 0; i < 3; ++i) putv<float>(f, c.up[i]);
                putv<float>(f, c.fov); putv<float>(f, c.lensr); putv<float>(f, c.focald); putv<float>(f, c.aspect);
                putv<int32_t>(f, c.xres); putv<int32_t>(f, c.yres);
                for (int i = 0; i < 4; ++i) putv<float>(f, c.crop[i]);
                putv<int32_t>(f, c.spp);
                putv<float>(f, cs.pFilm.x); putv<float>(f, cs.pFilm.y); putv<float>(f, cs.pLens.x); putv<float>(f, cs.pLens.y); putv<float>(f, cs.time);
                for (int i = 0; i < 3; ++i) putv<float>(f, ray.o[i]);
                for (int i = 0; i < 3; ++i) putv<float>(f, ray.d[i]);
                putv<float>(f, w);
                for (int i = 0; i < 3; ++i) putv<float>(f, ray.rxOrigin[i]);
                for (int i = 0; i < 3; ++i) putv<float>(f, ray.rxDirection[i]);
                for (int i = 0; i < 3; ++i) putv<float>(f, ray.ryOrigin[i]);
                for (int i = 0; i < 3; ++i) putv<float>(f, ray.ryDirection[i]);
                ++count;
            }
            ++ci;
        }
        fclose(f);
        printf("ref_probe: %d camera-ray records\n", count);
    }
    // ---- ComputeBeamDiffusionBSSRDF (core/bssrdf.cpp:113-160): the table the Subsurface / KdSubsurface material constructors compute, for a few (g, eta)
    {
        PbrtOptions.nThreads = 1;   // its ParallelFor
        FILE *f = fopen((dir + "/bssrdf_tables.bin").c_str(), "wb");
        const Float ge[][2] = {{0.f, 1.33f}, {.6f, 1.5f}, {-.3f, 1.1f}};
        for (auto &p : ge) {
            BSSRDFTable t(100, 64);   // materials/subsurface.h:57, kdsubsurface.h:66
            ComputeBeamDiffusionBSSRDF(p[0], p[1], &t);
            putv<float>(f, p[0]); putv<float>(f, p[1]);
            for (int i = 0; i < 100; ++i) putv<float>(f, t.rhoSamples[i]);
            for (int i = 0; i < 64; ++i) putv<float>(f, t.radiusSamples[i]);
            for (int i = 0; i < 6400; ++i) putv<float>(f, t.profile[i]);
            for (int i = 0; i < 100; ++i) putv<float>(f, t.rhoEff[i]);
            for (int i = 0; i < 6400; ++i) putv<float>(f, t.profileCDF[i]);
        }
        fclose(f);
    }
    // ---- TabulatedBSSRDF::Sr / Sample_Sr / Pdf_Sr (core/bssrdf.cpp:199-233, 353-390) and SubsurfaceFromDiffuse (:178-188) on the (g = 0, eta = 1.33)
    // table, HenyeyGreenstein::p / Sample_p (core/medium.cpp:210-246)
    {
        PbrtOptions.nThreads = 1;
        FILE *f = fopen((dir + "/bssrdf_radial.bin").c_str(), "wb");
        BSSRDFTable t(100, 64);
        ComputeBeamDiffusionBSSRDF(0.f, 1.33f, &t);
        RNG rng(53);
        auto U = [&](Float lo, Float hi) { return lo + (hi - lo) * rng.UniformFloat(); };
        SurfaceInteraction po(Point3f(0, 0, 0), Vector3f(), Point2f(), Vector3f(0, 0, 1), Vector3f(1, 0, 0), Vector3f(0, 1, 0), Normal3f(), Normal3f(), 0, nullptr);
        for (int k = 0; k < 2000; ++k) {
            Float sa[3], ss[3];
            for (int c = 0; c < 3; ++c) { sa[c] = std::pow(10.f, U(-3, 1)); ss[c] = std::pow(10.f, U(-2, 1.5f)); }
            if (k % 50 == 0) { sa[1] = 0; ss[1] = 0; }   // sigma_t == 0 in one channel
            TabulatedBSSRDF b(po, nullptr, TransportMode::Radiance, 1.33f, Spectrum::FromRGB(sa), Spectrum::FromRGB(ss), t);
            int ch = (int)(rng.UniformUInt32() % 3);
            Float r = std::pow(10.f, U(-4, 1.2f)) / std::max((Float)1e-3, sa[ch] + ss[ch]);
            Float u = rng.UniformFloat();
            Float sr[3];
            b.Sr(r).ToRGB(sr);
            Float rs = b.Sample_Sr(ch, u), pdf = b.Pdf_Sr(ch, r);
            // SubsurfaceFromDiffuse: reflectance + mean free path -> coefficients
            Float kd[3] = {U(.01f, .95f), U(.01f, .95f), U(.01f, .95f)}, mfp[3] = {U(.05f, 3), U(.05f, 3), U(.05f, 3)};
            Spectrum oa, os;
            SubsurfaceFromDiffuse(t, Spectrum::FromRGB(kd), Spectrum::FromRGB(mfp), &oa, &os);
            Float oar[3], osr[3];
            oa.ToRGB(oar); os.ToRGB(osr);
            for (int c = 0; c < 3; ++c) putv<float>(f, sa[c]);
            for (int c = 0; c < 3; ++c) putv<float>(f, ss[c]);
            putv<int32_t>(f, ch); putv<float>(f, r); putv<float>(f, u);
            for (int c = 0; c < 3; ++c) putv<float>(f, sr[c]);
            putv<float>(f, rs); putv<float>(f, pdf);
            for (int c = 0; c < 3; ++c) putv<float>(f, kd[c]);
            for (int c = 0; c < 3; ++c) putv<float>(f, mfp[c]);
            for (int c = 0; c < 3; ++c) putv<float>(f, oar[c]);
            for (int c = 0; c < 3; ++c) putv<float>(f, osr[c]);
        }
        fclose(f);
        f = fopen((dir + "/hg.bin").c_str(), "wb");
        for (int k = 0; k < 4000; ++k) {
            Float g = k % 10 == 0 ? 0.f : (k % 10 == 1 ? U(-1e-4f, 1e-4f) : U(-.97f, .97f));
            HenyeyGreenstein hg(g);
            Vector3f wo = UniformSampleSphere(Point2f(rng.UniformFloat(), rng.UniformFloat())), wi = UniformSampleSphere(Point2f(rng.UniformFloat(), rng.UniformFloat()));
            Point2f u(rng.UniformFloat(), rng.UniformFloat());
            Vector3f ws;
            Float p = hg.p(wo, wi), ps = hg.Sample_p(wo, &ws, u);
            putv<float>(f, g);
            for (int c = 0; c < 3; ++c) putv<float>(f, wo[c]);
            for (int c = 0; c < 3; ++c) putv<float>(f, wi[c]);
            putv<float>(f, u[0]); putv<float>(f, u[1]);
            putv<float>(f, p);
            for (int c = 0; c < 3; ++c) putv<float>(f, ws[c]);
            putv<float>(f, ps);
        }
        fclose(f);
    }
    // ---- the lights that depend on the scene (Light::Preprocess: world bound): DistantLight and InfiniteAreaLight -- constant, and with a
    // radiance map (argv[2]: MIPMap::Lookup + Distribution2D, lights/infinite.cpp:43-137) under a rotation.  The scene is one triangle.
    if (argc > 2) {
        PbrtOptions.nThreads = 1;   // the InfiniteAreaLight constructor runs a ParallelFor
        FILE *f = fopen((dir + "/light_samples_scene.bin").c_str(), "wb");
        RNG rng(31);
        auto U = [&](Float lo, Float hi) { return lo + (hi - lo) * rng.UniformFloat(); };
        static Transform id;
        int idx[3] = {0, 1, 2};
        Point3f P[3] = {Point3f(0, 0, 0), Point3f(1, 0, 0), Point3f(0, 1, 0)};
        auto tris = CreateTriangleMesh(&id, &id, false, 1, idx, 3, P, nullptr, nullptr, nullptr, nullptr, nullptr);
        std::vector<std::shared_ptr<Primitive>> prims;
        MediumInterface mi;
        prims.push_back(std::make_shared<GeometricPrimitive>(tris[0], nullptr, nullptr, mi));
        std::vector<std::shared_ptr<Light>> lights;
        std::vector<std::vector<Float>> desc;   // per light: kind, params
        for (int k = 0; k < 12; ++k) {
            Float Lrgb[3] = {U(.2f, 4), U(.2f, 4), U(.2f, 4)};
            ParamSet ps;
            ps.AddRGBSpectrum("L", std::unique_ptr<Float[]>(new Float[3]{Lrgb[0], Lrgb[1], Lrgb[2]}), 3);
            if (k < 6) {   // distant: from / to
                Float g[6]; for (int i = 0; i < 6; ++i) g[i] = U(-4, 4);
                std::unique_ptr<Point3f[]> from(new Point3f[1]), to(new Point3f[1]);
                from[0] = Point3f(g[0], g[1], g[2]); to[0] = Point3f(g[3], g[4], g[5]);
                ps.AddPoint3f("from", std::move(from), 1); ps.AddPoint3f("to", std::move(to), 1);
                lights.push_back(CreateDistantLight(Transform(), ps));
                desc.push_back({4, g[0], g[1], g[2], g[3], g[4], g[5], Lrgb[0], Lrgb[1], Lrgb[2]});
            } else {       // infinite: Rotate a1 about x, then a2 about z; k >= 9: with the radiance map
                Float a1 = U(-120, 120), a2 = U(-180, 180);
                if (k >= 9) { std::unique_ptr<std::string[]> m(new std::string[1]); m[0] = argv[2]; ps.AddString("mapname", std::move(m), 1); }
                lights.push_back(CreateInfiniteLight(Rotate(a1, Vector3f(1, 0, 0)) * Rotate(a2, Vector3f(0, 0, 1)), ps));
                desc.push_back({Float(k >= 9 ? 6 : 5), a1, a2, 0, 0, 0, 0, Lrgb[0], Lrgb[1], Lrgb[2]});
            }
        }
        Scene scene(std::make_shared<BVHAccel>(prims), lights);
        int count = 0;
        for (size_t k = 0; k < lights.size(); ++k)
            for (int j = 0; j < 64; ++j) {
                Point3f pc(U(-3, 3), U(-3, 3), U(-3, 3));
                Normal3f n(0, 0, 0);
                if (j % 4) { Vector3f v = UniformSampleSphere(Point2f(rng.UniformFloat(), rng.UniformFloat())); n = Normal3f(v.x, v.y, v.z); }
                Interaction ref(pc, n, Vector3f(), Vector3f(0, 0, 1), 0, mi);
                Point2f u(rng.UniformFloat(), rng.UniformFloat());
                Vector3f wi; Float pdf = 0; VisibilityTester vis;
                Spectrum Li = lights[k]->Sample_Li(ref, u, &wi, &pdf, &vis);
                Float rgb[3] = {0, 0, 0};
                Ray sr;
                if (pdf > 0) { Li.ToRGB(rgb); sr = vis.P0().SpawnRayTo(vis.P1()); }
                Vector3f wi2 = UniformSampleSphere(Point2f(rng.UniformFloat(), rng.UniformFloat()));
                Float pdfB = lights[k]->Pdf_Li(ref, wi2);
                // Le of the escaped ray along wi2 (infinite lights; 0 otherwise): path.cpp:97-98
                Float le[3] = {0, 0, 0};
                RayDifferential esc(pc, wi2);
                lights[k]->Le(esc).ToRGB(le);
                for (Float v : desc[k]) putv<float>(f, v);
                for (int i = 0; i < 3; ++i) putv<float>(f, pc[i]);
                for (int i = 0; i < 3; ++i) putv<float>(f, n[i]);
                putv<float>(f, u[0]); putv<float>(f, u[1]);
                for (int i = 0; i < 3; ++i) putv<float>(f, wi[i]);
                putv<float>(f, pdf);
                for (int i = 0; i < 3; ++i) putv<float>(f, rgb[i]);
                for (int i = 0; i < 3; ++i) putv<float>(f, pdf > 0 ? sr.o[i] : 0.f);
                for (int i = 0; i < 3; ++i) putv<float>(f, pdf > 0 ? sr.d[i] : 0.f);
                putv<float>(f, pdf > 0 ? sr.tMax : 0.f);
                for (int i = 0; i < 3; ++i) putv<float>(f, wi2[i]);
                putv<float>(f, pdfB);
                for (int i = 0; i < 3; ++i) putv<float>(f, le[i]);
                ++count;
            }
        fclose(f);
        printf("ref_probe: %d scene-light records\n", count);
    }
    return 0;
}
