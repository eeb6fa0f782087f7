// TEST INFRASTRUCTURE (not the deliverable binding -- that is integration/wavefrontpath.cpp, which binds the mi_* entry points only).
// The same FlattenScene (integration/flatten.h) behind the reference's own main / parser / API / BVH build, with the flattened description
// rendered by the CPU checker liboracle.so (oracle_render) instead of the device: on a machine without a GPU this proves the hand-over table of
// INTEGRATION.md s.1 -- the image must equal what pbrt_ref renders for the same file (tests/test_reference_binding.py) -- and hosts three probes
// that compare the reference's OWN classes with the checker on the flattened description:
//   PBRT_AMD_TEX_PROBE / PBRT_AMD_HIT_PROBE / PBRT_AMD_BSDF_PROBE = <report file>
// and PBRT_AMD_BVH_DUMP = <file>: the reference's BVHAccel (LinearBVHNode array + ordered primitives) as flattened, for the node-for-node comparison with the host
// library's own build of every split method (tests/test_reference_binding.py)
// Built by oracle/ref_build/Makefile into oracle/_ref/pbrt_ref_flatcheck; PBRT_AMD_BACKEND_LIB names liboracle.so.
#include "../../integration/flatten.h"

namespace pbrt {

// PBRT_AMD_TEX_PROBE=<report file>: every Texture object the scene's materials reach is evaluated BY THE REFERENCE'S OWN CLASS (Texture<T>::Evaluate,
// core/texture.h:139-144) at 2048 random interactions, and the backend evaluates the node it was flattened to (oracle_texture_eval) at the same
// interactions.  One report line per node: node, type, spectrum, evaluations, bit-identical ones, largest absolute difference.
static void TextureProbe(const Flat &flat, void *lib, const char *reportFile) {
    auto tex_eval = (void (*)(const mi_scene_desc *, int32_t, const mi_tex_query *, int64_t, float *))dlsym(lib, "oracle_texture_eval");
    if (!tex_eval) { Error("PBRT_AMD_TEX_PROBE: the backend has no oracle_texture_eval"); return; }
    if (flat.desc.n_textures == 0) {   // no textured material or mask: the description carries no node table (constant parameters were folded into mi_material)
        if (FILE *e = std::fopen(reportFile, "w")) std::fclose(e);
        return;
    }
    const int N = 2048;
    std::vector<mi_tex_query> q(N);
    RNG rng(7);
    auto U = [&](Float lo, Float hi) { return lo + (hi - lo) * rng.UniformFloat(); };
    for (auto &x : q) {
        for (int i = 0; i < 3; ++i) x.p[i] = U(-3, 3);
        x.uv[0] = U(-.5f, 1.5f); x.uv[1] = U(-.5f, 1.5f);
        Float s = std::pow(10.f, U(-3.5f, -.5f));   // footprints from sub-texel to many texels
        for (int i = 0; i < 3; ++i) { x.dpdx[i] = U(-s, s); x.dpdy[i] = U(-s, s); }
        x.dudx = U(-s, s); x.dvdx = U(-s, s); x.dudy = U(-s, s); x.dvdy = U(-s, s);
    }
    auto interaction = [](const mi_tex_query &x) {
        SurfaceInteraction si;
        si.p = Point3f(x.p[0], x.p[1], x.p[2]); si.uv = Point2f(x.uv[0], x.uv[1]);
        si.dpdx = Vector3f(x.dpdx[0], x.dpdx[1], x.dpdx[2]); si.dpdy = Vector3f(x.dpdy[0], x.dpdy[1], x.dpdy[2]);
        si.dudx = x.dudx; si.dvdx = x.dvdx; si.dudy = x.dudy; si.dvdy = x.dvdy;
        return si;
    };
    FILE *f = std::fopen(reportFile, "w");
    if (!f) { Error("PBRT_AMD_TEX_PROBE: cannot write %s", reportFile); return; }
    std::vector<float> got(3 * (size_t)N);
    auto report = [&](int node, const std::vector<float> &ref, int comps) {
        tex_eval(&flat.desc, node, q.data(), N, got.data());
        int same = 0; double worst = 0;
        for (int i = 0; i < N; ++i) {
            bool eq = true;
            for (int c = 0; c < comps; ++c) {
                float a = got[3 * i + c], b = ref[(size_t)comps * i + c];
                if (std::memcmp(&a, &b, 4) != 0 && !(a == b)) { eq = false; worst = std::max(worst, (double)std::abs(a - b)); }
            }
            same += eq;
        }
        std::fprintf(f, "%d %d %d %d %d %.9g\n", node, flat.textures[node].type, comps == 3, N, same, worst);
    };
    for (auto &pr : flat.probeFloat) {
        std::vector<float> ref(N);
        for (int i = 0; i < N; ++i) ref[i] = pr.first->Evaluate(interaction(q[i]));
        report(pr.second, ref, 1);
    }
    for (auto &pr : flat.probeSpectrum) {
        std::vector<float> ref(3 * (size_t)N);
        for (int i = 0; i < N; ++i) { Float c[3]; pr.first->Evaluate(interaction(q[i])).ToRGB(c); ref[3 * i] = c[0]; ref[3 * i + 1] = c[1]; ref[3 * i + 2] = c[2]; }
        report(pr.second, ref, 3);
    }
    std::fclose(f);
}

// PBRT_AMD_HIT_PROBE=<report file>: 20 000 random rays through the scene's bounds (half from outside aimed into them, half from inside; finite and
// infinite tMax) go through the REFERENCE's own Scene::Intersect / IntersectP (BVHAccel + TransformedPrimitive + Triangle / Sphere, alpha masks
// included) and through the backend's traversal of the flattened description (oracle_intersect / oracle_intersect_p): hit or miss, the hit
// distance and the geometric normal must be the same bit for bit.  One report line: rays, reference hits, agreeing hit flags, agreeing t, agreeing n,
// agreeing occlusion flags.
static void HitProbe(const Scene &scene, const Flat &flat, void *lib, const char *reportFile) {
    auto isect_fn = (void (*)(const mi_scene_desc *, const mi_ray *, int64_t, mi_hit *, uint64_t *))dlsym(lib, "oracle_intersect");
    auto occl_fn = (void (*)(const mi_scene_desc *, const mi_ray *, int64_t, uint8_t *, uint64_t *))dlsym(lib, "oracle_intersect_p");
    if (!isect_fn || !occl_fn) { Error("PBRT_AMD_HIT_PROBE: the backend has no oracle_intersect / oracle_intersect_p"); return; }
    const int N = 20000;
    Bounds3f wb = scene.WorldBound();
    Point3f c; Float rad;
    wb.BoundingSphere(&c, &rad);
    if (!(rad > 0)) rad = 1;
    RNG rng(13);
    auto U = [&](Float lo, Float hi) { return lo + (hi - lo) * rng.UniformFloat(); };
    std::vector<mi_ray> rays(N);
    for (int i = 0; i < N; ++i) {
        Point3f target(U(wb.pMin.x, wb.pMax.x), U(wb.pMin.y, wb.pMax.y), U(wb.pMin.z, wb.pMax.z));
        Point3f o;
        Vector3f d;
        if (i % 2) { o = c + (2.5f * rad) * UniformSampleSphere(Point2f(rng.UniformFloat(), rng.UniformFloat())); d = target - o; if (i % 4 == 1) d = Normalize(d); }
        else { o = target; d = UniformSampleSphere(Point2f(rng.UniformFloat(), rng.UniformFloat())); }
        mi_ray &r = rays[i];
        for (int k = 0; k < 3; ++k) { r.o[k] = o[k]; r.d[k] = d[k]; }
        r.tmax = (i % 5 == 0) ? U(.1f, 3.f) : Infinity;
        r.time = 0;
    }
    std::vector<mi_hit> got(N);
    std::vector<uint8_t> occ(N);
    isect_fn(&flat.desc, rays.data(), N, got.data(), nullptr);
    occl_fn(&flat.desc, rays.data(), N, occ.data(), nullptr);
    int hits = 0, sameFlag = 0, sameT = 0, sameN = 0, sameOcc = 0;
    for (int i = 0; i < N; ++i) {
        const mi_ray &r = rays[i];
        Ray ray(Point3f(r.o[0], r.o[1], r.o[2]), Vector3f(r.d[0], r.d[1], r.d[2]), r.tmax, 0.f);
        SurfaceInteraction si;
        bool hit = scene.Intersect(ray, &si);
        Ray shadow(Point3f(r.o[0], r.o[1], r.o[2]), Vector3f(r.d[0], r.d[1], r.d[2]), r.tmax, 0.f);
        bool blocked = scene.IntersectP(shadow);
        hits += hit;
        sameFlag += hit == (got[i].prim >= 0);
        sameOcc += blocked == (occ[i] != 0);
        if (hit && got[i].prim >= 0) {
            float t = ray.tMax, n[3] = {si.n.x, si.n.y, si.n.z};
            sameT += std::memcmp(&t, &got[i].t, 4) == 0;
            sameN += std::memcmp(n, got[i].n, 12) == 0;
        }
    }
    FILE *f = std::fopen(reportFile, "w");
    if (!f) { Error("PBRT_AMD_HIT_PROBE: cannot write %s", reportFile); return; }
    std::fprintf(f, "%d %d %d %d %d %d\n", N, hits, sameFlag, sameT, sameN, sameOcc);
    std::fclose(f);
}

// PBRT_AMD_BSDF_PROBE=<report file>: at the first hit of 20 000 random rays the REFERENCE builds the BSDF with its own Material class
// (SurfaceInteraction::ComputeScatteringFunctions: texture evaluation, bump mapping, lobe list, shading frame) and evaluates BSDF::f, Pdf and
// Sample_f for a random direction / sample; the backend does the same on the flattened description (oracle_bsdf_at_hit).  One report line: rays,
// hits with a BSDF, agreeing states, agreeing component counts, agreeing f, agreeing Pdf, agreeing Sample_f results (direction, pdf, value, type).
static void BsdfProbe(const Scene &scene, const Flat &flat, void *lib, const char *reportFile) {
    auto fn = (void (*)(const mi_scene_desc *, const mi_ray *, const float *, const float *, int64_t, float *))dlsym(lib, "oracle_bsdf_at_hit");
    if (!fn) { Error("PBRT_AMD_BSDF_PROBE: the backend has no oracle_bsdf_at_hit"); return; }
    const int N = 20000;
    Bounds3f wb = scene.WorldBound();
    Point3f c; Float rad;
    wb.BoundingSphere(&c, &rad);
    if (!(rad > 0)) rad = 1;
    RNG rng(17);
    auto U = [&](Float lo, Float hi) { return lo + (hi - lo) * rng.UniformFloat(); };
    std::vector<mi_ray> rays(N);
    std::vector<float> wi(3 * (size_t)N), u(2 * (size_t)N), got(14 * (size_t)N);
    for (int i = 0; i < N; ++i) {
        Point3f target(U(wb.pMin.x, wb.pMax.x), U(wb.pMin.y, wb.pMax.y), U(wb.pMin.z, wb.pMax.z));
        Point3f o = (i % 2) ? c + (2.5f * rad) * UniformSampleSphere(Point2f(rng.UniformFloat(), rng.UniformFloat())) : target;
        Vector3f d = (i % 2) ? Normalize(target - o) : UniformSampleSphere(Point2f(rng.UniformFloat(), rng.UniformFloat()));
        Vector3f w = UniformSampleSphere(Point2f(rng.UniformFloat(), rng.UniformFloat()));
        for (int k = 0; k < 3; ++k) { rays[i].o[k] = o[k]; rays[i].d[k] = d[k]; wi[3 * i + k] = w[k]; }
        rays[i].tmax = Infinity; rays[i].time = 0;
        u[2 * i] = rng.UniformFloat(); u[2 * i + 1] = rng.UniformFloat();
    }
    fn(&flat.desc, rays.data(), wi.data(), u.data(), N, got.data());
    MemoryArena arena;
    int withBsdf = 0, sameState = 0, sameCount = 0, sameF = 0, samePdf = 0, sameSample = 0;
    for (int i = 0; i < N; ++i) {
        const mi_ray &r = rays[i];
        RayDifferential ray(Point3f(r.o[0], r.o[1], r.o[2]), Vector3f(r.d[0], r.d[1], r.d[2]));
        SurfaceInteraction si;
        const float *g = &got[14 * (size_t)i];
        int state = 0;
        if (scene.Intersect(ray, &si)) {
            si.ComputeScatteringFunctions(ray, arena, true, TransportMode::Radiance);
            state = si.bsdf ? 1 : 2;
        }
        sameState += state == (int)g[0];
        if (state == 1 && (int)g[0] == 1) {
            ++withBsdf;
            Vector3f w(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]);
            sameCount += si.bsdf->NumComponents() == (int)g[1];
            Float f[3];
            si.bsdf->f(si.wo, w).ToRGB(f);
            float ff[3] = {f[0], f[1], f[2]};
            sameF += std::memcmp(ff, g + 2, 12) == 0;
            float pdf = si.bsdf->Pdf(si.wo, w);
            samePdf += std::memcmp(&pdf, g + 5, 4) == 0;
            Vector3f ws; Float ps = 0; BxDFType st = BxDFType(0);
            Spectrum fsv = si.bsdf->Sample_f(si.wo, &ws, Point2f(u[2 * i], u[2 * i + 1]), &ps, BSDF_ALL, &st);
            float rec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ps > 0) { Float c3[3]; fsv.ToRGB(c3); rec[0] = ws.x; rec[1] = ws.y; rec[2] = ws.z; rec[3] = ps; rec[4] = c3[0]; rec[5] = c3[1]; rec[6] = c3[2]; rec[7] = (float)(int)st; }
            sameSample += std::memcmp(rec, g + 6, 32) == 0;
            if (std::getenv("PBRT_AMD_BSDF_PROBE_VERBOSE") && (std::memcmp(rec, g + 6, 32) != 0 || std::memcmp(ff, g + 2, 12) != 0 || std::memcmp(&pdf, g + 5, 4) != 0))
                std::fprintf(stderr, "ray %d comps %d: f ref %.9g %.9g %.9g got %.9g %.9g %.9g | pdf %.9g / %.9g | sample ref wi %.9g %.9g %.9g pdf %.9g f %.9g %.9g %.9g type %g got wi %.9g %.9g %.9g pdf %.9g f %.9g %.9g %.9g type %g\n",
                             i, (int)g[1], ff[0], ff[1], ff[2], g[2], g[3], g[4], pdf, g[5], rec[0], rec[1], rec[2], rec[3], rec[4], rec[5], rec[6], rec[7], g[6], g[7], g[8], g[9], g[10], g[11], g[12], g[13]);
        }
        arena.Reset();
    }
    FILE *f = std::fopen(reportFile, "w");
    if (!f) { Error("PBRT_AMD_BSDF_PROBE: cannot write %s", reportFile); return; }
    std::fprintf(f, "%d %d %d %d %d %d %d\n", N, withBsdf, sameState, sameCount, sameF, samePdf, sameSample);
    std::fclose(f);
}

class FlatCheckIntegrator : public Integrator {
  public:
    FlatCheckIntegrator(const WavefrontParams &w, std::shared_ptr<const Camera> camera, std::shared_ptr<Sampler> sampler, bool volpath)
        : w(w), camera(camera), sampler(sampler), volpath(volpath) {}
    void Render(const Scene &scene) {
        std::unique_ptr<Flat> flat = FlattenScene(scene, *camera, *sampler, w.maxDepth, w.rrThreshold, w.pixelBounds, w.lightStrategy, volpath);
        if (!flat->error.empty()) { Error("FlatCheckIntegrator: %s", flat->error.c_str()); return; }
        Film *film = camera->film;
        std::vector<float> rgbw(4 * (size_t)film->croppedPixelBounds.Area());
        const char *libPath = std::getenv("PBRT_AMD_BACKEND_LIB");
        if (!libPath) { Error("FlatCheckIntegrator: PBRT_AMD_BACKEND_LIB (liboracle.so) is not set"); return; }
        void *lib = dlopen(libPath, RTLD_NOW);
        if (!lib) { Error("FlatCheckIntegrator: %s", dlerror()); return; }
        auto oracle_render = (double (*)(const mi_scene_desc *, float *, int, int, int, uint64_t *, const int32_t *))dlsym(lib, "oracle_render");
        if (!oracle_render) { Error("FlatCheckIntegrator: oracle_render not found in %s", libPath); return; }
        if (const char *dump = std::getenv("PBRT_AMD_BVH_DUMP")) {   // the REFERENCE's own BVHAccel as it crosses the boundary: the node array and the ordered triangles (vertex positions)
            if (FILE *f = std::fopen(dump, "wb")) {
                const mi_scene_desc &d = flat->desc;
                uint32_t hdr[2] = {d.n_bvh_nodes, d.n_tris};
                std::fwrite(hdr, 4, 2, f);
                std::fwrite(d.bvh_nodes, sizeof(mi_bvh2_node), d.n_bvh_nodes, f);
                for (uint32_t t = 0; t < d.n_tris; ++t)
                    for (int k = 0; k < 3; ++k) std::fwrite(d.P + 3 * (size_t)d.tri_indices[3 * t + k], 4, 3, f);
                std::fclose(f);
            }
            if (std::getenv("PBRT_AMD_BVH_DUMP_ONLY")) return;
        }
        if (const char *probe = std::getenv("PBRT_AMD_TEX_PROBE")) TextureProbe(*flat, lib, probe);
        if (const char *probe = std::getenv("PBRT_AMD_HIT_PROBE")) HitProbe(scene, *flat, lib, probe);
        if (const char *probe = std::getenv("PBRT_AMD_BSDF_PROBE")) BsdfProbe(scene, *flat, lib, probe);
        uint64_t counters[8] = {0};
        oracle_render(&flat->desc, rgbw.data(), 0, -1, NumSystemCores(), counters, nullptr);
        MergeIntoReferenceFilm(film, rgbw);
    }

  private:
    const WavefrontParams w;
    std::shared_ptr<const Camera> camera;
    std::shared_ptr<Sampler> sampler;
    const bool volpath;
};

// linked in front of libpbrt_ref.a like the binding's factories (integration/wavefrontpath.cpp)
PathIntegrator *CreatePathIntegrator(const ParamSet &params, std::shared_ptr<Sampler> sampler, std::shared_ptr<const Camera> camera) {
    return reinterpret_cast<PathIntegrator *>(static_cast<Integrator *>(new FlatCheckIntegrator(ReadWavefrontParams(params, camera), camera, sampler, false)));
}
VolPathIntegrator *CreateVolPathIntegrator(const ParamSet &params, std::shared_ptr<Sampler> sampler, std::shared_ptr<const Camera> camera) {
    return reinterpret_cast<VolPathIntegrator *>(static_cast<Integrator *>(new FlatCheckIntegrator(ReadWavefrontParams(params, camera), camera, sampler, true)));
}

}  // namespace pbrt
