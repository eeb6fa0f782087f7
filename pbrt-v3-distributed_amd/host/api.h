// The pbrt scene-description API: same entry points, argument meaning and error behaviour as
// the reference's core/api.h:47-92 (RenderMan-style state machine; Error()/Warning() and carry on).
#pragma once
#include <string>

#include "scene.h"

namespace pbrt_amd {

struct Options {   // core/pbrt.h:167-181 (the flags that make sense for this path) + GPU options
    int nThreads = 0;
    bool quickRender = false, quiet = false;
    std::string imageFile;
    Float cropWindow[2][2] = {{0, 1}, {0, 1}};
    int nGpus = 1;               // --gpus N : tile-shard Render over N devices (in-process)
    bool fastSamplers = false;   // --fast-samplers / PBRT_AMD_FAST_SAMPLERS=1: Sampler "random" / "stratified" / "02sequence" / "lowdiscrepancy" render with "sobol" at the same
                                 // sample count (a warning says so) -- the wavefront pipeline at full speed instead of the reference's image through tile-serial rounds
    bool deferRender = false;    // WorldEnd keeps the built Scene/Integrator instead of rendering
};

void pbrtInit(const Options &opt);
int NumHostThreads();   // host threads for scene construction (bvh.cpp)
void pbrtCleanup();
void pbrtIdentity();
void pbrtTranslate(Float dx, Float dy, Float dz);
void pbrtRotate(Float angle, Float ax, Float ay, Float az);
void pbrtScale(Float sx, Float sy, Float sz);
void pbrtLookAt(Float ex, Float ey, Float ez, Float lx, Float ly, Float lz, Float ux, Float uy, Float uz);
void pbrtConcatTransform(Float transform[16]);
void pbrtTransform(Float transform[16]);
void pbrtCoordinateSystem(const std::string &);
void pbrtCoordSysTransform(const std::string &);
void pbrtActiveTransformAll();
void pbrtActiveTransformEndTime();
void pbrtActiveTransformStartTime();
void pbrtTransformTimes(Float start, Float end);
void pbrtPixelFilter(const std::string &name, const ParamSet &params);
void pbrtFilm(const std::string &type, const ParamSet &params);
void pbrtSampler(const std::string &name, const ParamSet &params);
void pbrtAccelerator(const std::string &name, const ParamSet &params);
void pbrtIntegrator(const std::string &name, const ParamSet &params);
void pbrtCamera(const std::string &, const ParamSet &cameraParams);
void pbrtMakeNamedMedium(const std::string &name, const ParamSet &params);
void pbrtMediumInterface(const std::string &insideName, const std::string &outsideName);
void pbrtWorldBegin();
void pbrtAttributeBegin();
void pbrtAttributeEnd();
void pbrtTransformBegin();
void pbrtTransformEnd();
void pbrtTexture(const std::string &name, const std::string &type, const std::string &texname, const ParamSet &params);
void pbrtMaterial(const std::string &name, const ParamSet &params);
void pbrtMakeNamedMaterial(const std::string &name, const ParamSet &params);
void pbrtNamedMaterial(const std::string &name);
void pbrtLightSource(const std::string &name, const ParamSet &params);
void pbrtAreaLightSource(const std::string &name, const ParamSet &params);
void pbrtShape(const std::string &name, const ParamSet &params);
void pbrtReverseOrientation();
void pbrtObjectBegin(const std::string &name);
void pbrtObjectEnd();
void pbrtObjectInstance(const std::string &name);
void pbrtWorldEnd();

void pbrtParseFile(std::string filename);      // core/parser.cpp:1089
void pbrtParseString(std::string str);

// deferRender mode: what WorldEnd built (api.cpp:1611-1612 holds them in unique_ptrs and renders)
struct BuiltScene {
    std::unique_ptr<Scene> scene;
    std::unique_ptr<WavefrontPathIntegrator> integrator;
};
std::unique_ptr<BuiltScene> pbrtTakeBuiltScene();

}  // namespace pbrt_amd
