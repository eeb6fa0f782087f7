// Loop subdivision surfaces (reference shapes/loopsubdiv.cpp:149-400) -- host-side mesh producer.
#include "scene.h"
namespace pbrt_amd {
std::shared_ptr<TriangleMesh> CreateLoopSubdiv(const Transform &, bool, const ParamSet &) {
    Warning("Shape \"loopsubdiv\" not implemented yet; skipped.");
    return nullptr;
}
}
