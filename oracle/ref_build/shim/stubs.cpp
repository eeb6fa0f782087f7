// Build-infrastructure stubs: glog flag globals and the Ptex factory symbols that
// textures/ptex.cpp (skipped: needs the absent Ptex library) would have defined.
#include "pbrt.h"
#include "paramset.h"
#include "textures/ptex.h"
int FLAGS_stderrthreshold = 1, FLAGS_minloglevel = 0, FLAGS_v = 0;
bool FLAGS_logtostderr = false;
std::string FLAGS_log_dir;
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
