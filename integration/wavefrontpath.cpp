// Reference-side binding of the MI355X path tracer, COMPILED AGAINST THE UNMODIFIED REFERENCE (libpbrt_ref.a): the
// `WavefrontPathIntegrator : public Integrator` of INTEGRATION.md s.2 -- the code a pbrt-v3 maintainer adds to hand an already built
// `Scene` to include/pbrt_amd.h (SURVEY.md s.8 row b).  It binds the mi_* entry points of libpbrt_amd.so and nothing else: no device,
// no library or a library without mi_ctx_create is an error, never a CPU render.
//
//   PBRT_AMD_DEVICE_LIB   path of libpbrt_amd.so (default: "libpbrt_amd.so" through the loader's search path)
//   PBRT_AMD_GPU_MAP      "2,3": the device ordinal of each rank (default 0, 1, ...)
//   PBRT_AMD_GPUS=N       (or `Integrator "path" "integer gpus" [N]`) shard the 16x16 image tiles over N GPUs of this node: one mi_ctx per GPU,
//                         the scene replicated, mi_render(rank r, world N) on each, one mi_film_gather (RCCL over xGMI) onto GPU 0 --
//                         the reference's tile loop core/integrator.cpp:228-339 with the merge of core/film.cpp:117-130 across devices
//
// How it gets called without touching the reference: RenderOptions::MakeIntegrator (core/api.cpp:1666-1718) instantiates integrators by
// name through `Create<X>Integrator(params, sampler, camera)`.  This file DEFINES pbrt::CreatePathIntegrator / CreateVolPathIntegrator and is
// linked in front of libpbrt_ref.a, so `Integrator "path"` resolves to it (integrators/path.o, whose only symbol api.o needs is that factory,
// is simply not pulled from the archive).  The factories have to return the reference's declared types, hence the casts at the end of this
// file; a maintainer would instead add one `else if (IntegratorName == "wavefrontpath")` line to api.cpp:1681-1701 and return an Integrator *.
#include "flatten.h"

namespace pbrt {

class WavefrontPathIntegrator : public Integrator {   // core/integrator.h:53-58
  public:
    WavefrontPathIntegrator(const WavefrontParams &w, std::shared_ptr<const Camera> camera, std::shared_ptr<Sampler> sampler, bool volpath)
        : w(w), camera(camera), sampler(sampler), volpath(volpath) {}
    void Render(const Scene &scene);   // the one call site: core/api.cpp:1623

  private:
    const WavefrontParams w;
    std::shared_ptr<const Camera> camera;
    std::shared_ptr<Sampler> sampler;
    const bool volpath;   // stands in for VolPathIntegrator (media attenuate and scatter) instead of PathIntegrator
};

namespace {
// the entry points of include/pbrt_amd.h this binding uses, resolved once
struct DeviceApi {
    void *handle = nullptr;
    const char *(*last_error)() = nullptr;
    int (*abi_version)() = nullptr;
    int (*ctx_create)(int, void *, mi_ctx **) = nullptr;
    void (*ctx_destroy)(mi_ctx *) = nullptr;
    int (*scene_upload)(mi_ctx *, const mi_scene_desc *) = nullptr;
    int (*render)(mi_ctx *, const mi_render_params *) = nullptr;
    int (*sync)(mi_ctx *) = nullptr;
    int (*film_download)(mi_ctx *, float *) = nullptr;
    int (*film_gather)(mi_ctx **, int, int) = nullptr;
    int (*counters)(mi_ctx *, uint64_t *) = nullptr;
    bool load() {
        if (handle) return true;
        const char *path = std::getenv("PBRT_AMD_DEVICE_LIB");
        if (!path) path = "libpbrt_amd.so";
        handle = dlopen(path, RTLD_NOW);
        if (!handle) { Error("WavefrontPathIntegrator: cannot load %s (the HIP path tracer; set PBRT_AMD_DEVICE_LIB): %s", path, dlerror()); return false; }
#define BIND(field, sym) field = (decltype(field))dlsym(handle, sym); if (!field) { Error("WavefrontPathIntegrator: %s lacks %s -- not libpbrt_amd.so", path, sym); handle = nullptr; return false; }
        BIND(ctx_create, "mi_ctx_create") BIND(last_error, "mi_last_error") BIND(abi_version, "mi_abi_version") BIND(ctx_destroy, "mi_ctx_destroy")
        BIND(scene_upload, "mi_scene_upload") BIND(render, "mi_render") BIND(sync, "mi_sync") BIND(film_download, "mi_film_download")
        BIND(film_gather, "mi_film_gather") BIND(counters, "mi_counters")
#undef BIND
        if (abi_version() != MI_ABI_VERSION) { Error("WavefrontPathIntegrator: %s has ABI %d, this binding was compiled for %d", path, abi_version(), MI_ABI_VERSION); handle = nullptr; return false; }
        return true;
    }
};
DeviceApi g_dev;
}  // namespace

void WavefrontPathIntegrator::Render(const Scene &scene) {
    std::unique_ptr<Flat> flat = FlattenScene(scene, *camera, *sampler, w.maxDepth, w.rrThreshold, w.pixelBounds, w.lightStrategy, volpath);
    if (!flat->error.empty()) { Error("WavefrontPathIntegrator: %s", flat->error.c_str()); return; }   // pbrt convention: report and return
    if (!g_dev.load()) return;
    Film *film = camera->film;
    std::vector<float> rgbw(4 * (size_t)film->croppedPixelBounds.Area());
    // image tiles shard across the GPUs (scene replicated); each device renders only its tiles.  mi_render is asynchronous: the loop queues every
    // device's frame and the GPUs render concurrently; mi_film_gather waits for all of them and combines the films on GPU 0.
    const int world = w.gpus;
    std::vector<mi_ctx *> ctxs(world, nullptr);
    auto destroyAll = [&]() { for (mi_ctx *c : ctxs) if (c) g_dev.ctx_destroy(c); };
    std::vector<int> device(world);
    for (int r = 0; r < world; ++r) device[r] = r;
    if (const char *m = std::getenv("PBRT_AMD_GPU_MAP")) {   // "2,3,6,7": which device ordinal each rank uses (contexts may share a device: mi_film_gather handles that)
        std::stringstream ss(m);
        std::string tok;
        for (int r = 0; r < world && std::getline(ss, tok, ','); ++r) device[r] = std::atoi(tok.c_str());
    }
    for (int r = 0; r < world; ++r)
        if (g_dev.ctx_create(device[r], nullptr, &ctxs[r]) != 0 || g_dev.scene_upload(ctxs[r], &flat->desc) != 0) {
            Error("WavefrontPathIntegrator: rank %d on GPU %d: %s", r, device[r], g_dev.last_error());
            destroyAll();
            return;
        }
    bool ok = true;
    for (int r = 0; r < world && ok; ++r) {
        mi_render_params rp;
        std::memset(&rp, 0, sizeof(rp));
        rp.rank = r; rp.world = world; rp.spp_begin = 0; rp.spp_end = -1;
        if (g_dev.render(ctxs[r], &rp) != 0) { Error("WavefrontPathIntegrator: GPU %d render: %s", r, g_dev.last_error()); ok = false; }
    }
    if (ok && world > 1 && g_dev.film_gather(ctxs.data(), world, 0) != 0) { Error("WavefrontPathIntegrator: film gather: %s", g_dev.last_error()); ok = false; }
    if (ok && (g_dev.sync(ctxs[0]) != 0 || g_dev.film_download(ctxs[0], rgbw.data()) != 0)) { Error("WavefrontPathIntegrator: GPU 0 film: %s", g_dev.last_error()); ok = false; }
    uint64_t guardTrips = 0;
    for (int r = 0; r < world && ok; ++r) {
        uint64_t c[MI_CNT_COUNT];
        if (g_dev.counters(ctxs[r], c) == 0) guardTrips += c[MI_CNT_TRACE_GUARD_TRIPS];
    }
    destroyAll();
    if (ok && guardTrips) { Error("WavefrontPathIntegrator: %llu traversal wave(s) hit the non-termination guard: the frame is invalid", (unsigned long long)guardTrips); ok = false; }
    if (!ok) { Error("WavefrontPathIntegrator: rendering failed, no image written"); return; }   // never an image with missing tiles
    MergeIntoReferenceFilm(film, rgbw);
}

// Same signature as integrators/path.cpp:190-213, which this definition stands in for (see the header comment): parameters read exactly as there.
// api.cpp keeps the result as an Integrator* and only ever calls the virtual Render on it.
PathIntegrator *CreatePathIntegrator(const ParamSet &params, std::shared_ptr<Sampler> sampler, std::shared_ptr<const Camera> camera) {
    return reinterpret_cast<PathIntegrator *>(static_cast<Integrator *>(new WavefrontPathIntegrator(ReadWavefrontParams(params, camera), camera, sampler, false)));
}
// Same signature as integrators/volpath.cpp:192-215: `Integrator "volpath"` resolves here as well (integrators/volpath.o is not pulled from the archive)
VolPathIntegrator *CreateVolPathIntegrator(const ParamSet &params, std::shared_ptr<Sampler> sampler, std::shared_ptr<const Camera> camera) {
    return reinterpret_cast<VolPathIntegrator *>(static_cast<Integrator *>(new WavefrontPathIntegrator(ReadWavefrontParams(params, camera), camera, sampler, true)));
}

}  // namespace pbrt
