// Build-infrastructure stubs: glog flag globals and the Ptex factory symbols that
// textures/ptex.cpp (skipped: needs the absent Ptex library) would have defined.
#include "pbrt.h"
#include "paramset.h"
#include "textures/ptex.h"
int FLAGS_stderrthreshold = 1, FLAGS_minloglevel = 0, FLAGS_v = 0;
bool FLAGS_logtostderr = false;
std::string FLAGS_log_dir;
// render timing for bench.py's cpu_baseline (see shim/glog/logging.h)
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
namespace shimlog {
bool g_render_times = std::getenv("PBRT_REF_RENDER_TIMES") != nullptr;
static std::atomic<long long> g_t0(0);
static long long NowNs() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void RenderMark(const std::string &msg) {
    if (msg.compare(0, 19, "Starting image tile") == 0) {
        long long expect = 0;
        g_t0.compare_exchange_strong(expect, NowNs());
    } else if (msg.compare(0, 18, "Rendering finished") == 0) {
        long long t0 = g_t0.exchange(0);
        if (t0) fprintf(stderr, "[pbrt_ref] Integrator::Render seconds %.6f\n", (NowNs() - t0) * 1e-9);
    }
}
}
namespace pbrt {
PtexTexture<Float> *CreatePtexFloatTexture(const Transform &, const TextureParams &) {
    Error("ptex textures unavailable in the oracle build");
    return nullptr;
}
PtexTexture<Spectrum> *CreatePtexSpectrumTexture(const Transform &, const TextureParams &) {
    Error("ptex textures unavailable in the oracle build");
    return nullptr;
}
}
