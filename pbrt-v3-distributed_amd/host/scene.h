// Host-side object model: the slice of pbrt-v3's plugin surface the hot path touches.
// Class / factory names, constructor arguments and defaults follow the reference headers
// cited at each declaration, so a pbrt-v3 user finds the same handles.  Geometry is kept as
// whole meshes + index ranges (no per-triangle heap objects: the reference spends ~233 B per
// triangle on Triangle/GeometricPrimitive/shared_ptr blocks) and the whole
// Scene flattens to the POD mi_scene_desc of include/pbrt_amd.h.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pbrt_amd.h"
#include "geom.h"
#include "paramset.h"

namespace pbrt_amd {

// ---- Shape: TriangleMesh (shapes/triangle.h:51-69; vertices are world space, triangle.cpp:72-74)
struct TriangleMesh {
    std::vector<Vec3> p;        // world space
    std::vector<Vec3> n;        // world space (ObjectToWorld(Normal)), empty if absent
    std::vector<Vec3> s;        // tangents, empty if absent
    std::vector<Float> uv;      // 2 per vertex, empty if absent
    std::vector<int> indices;   // 3 per triangle
    bool reverseOrientation = false, transformSwapsHandedness = false;
    int alphaTex = -1, shadowAlphaTex = -1;   // TriangleMesh::alphaMask / shadowAlphaMask as float texture nodes (triangle.h:66-67)
    int nTriangles() const { return (int)indices.size() / 3; }
};
// CreateTriangleMesh (triangle.cpp:94-110) / CreateTriangleMeshShape (:648-744) / CreatePLYMesh
// (shapes/plymesh.cpp:157-290) / CreateLoopSubdiv (shapes/loopsubdiv.cpp:149-400)
std::shared_ptr<TriangleMesh> CreateTriangleMesh(const Transform &o2w, bool reverseOrientation, int nTris,
                                                 const int *indices, int nVerts, const Vec3 *P, const Vec3 *S,
                                                 const Vec3 *N, const Float *UV);
std::shared_ptr<TriangleMesh> CreateTriangleMeshShape(const Transform &o2w, bool ro, const ParamSet &ps);
std::shared_ptr<TriangleMesh> CreatePLYMesh(const Transform &o2w, bool ro, const ParamSet &ps);
std::shared_ptr<TriangleMesh> CreateLoopSubdiv(const Transform &o2w, bool ro, const ParamSet &ps);
std::shared_ptr<TriangleMesh> CreateNURBS(const Transform &o2w, bool ro, const ParamSet &ps);   // shapes/nurbs.cpp: diced to a TriangleMesh by the reference itself
std::shared_ptr<TriangleMesh> CreateSphereMesh(const Transform &o2w, bool ro, const ParamSet &ps);
// MakeShapes (api.cpp:430-539)
std::shared_ptr<TriangleMesh> MakeShapes(const std::string &name, const Transform &o2w, bool ro, const ParamSet &ps);

// ---- Textures (core/texture.h, core/mipmap.h, textures/*): POD nodes + image pyramids, owned per parse
struct ImagePyramid {   // MIPMap<T> after its constructor
    int width = 0, height = 0, levels = 0, channels = 0, wrap = 0;
    bool trilinear = false;
    Float maxAniso = 8;
    std::vector<Float> texels;   // all levels, finest first
};
struct TextureStore {
    std::vector<mi_texture> nodes;
    std::vector<std::shared_ptr<ImagePyramid>> images;
    std::map<std::string, int> imageCache;   // ImageTexture::textures keyed by TexInfo (imagemap.h:51-75)
    bool Fold(int node, RGB *out) const;     // value of a hit-independent node (constant / scale / mix of such)
};
std::shared_ptr<TextureStore> CurrentTextures();   // of the parse in progress
void ResetTextures();
int ConstantTextureNode(bool spectrum, const RGB &v);
int MakeFloatTexture(const std::string &name, const Transform &tex2world, const TextureParams &tp);      // api.cpp:613-647
int MakeSpectrumTexture(const std::string &name, const Transform &tex2world, const TextureParams &tp);   // api.cpp:649-683
std::shared_ptr<ImagePyramid> BuildMIPMap(int w, int h, int channels, std::vector<Float> img, bool doTrilinear, Float maxAniso, int wrap);
bool ReadImage(const std::string &name, std::vector<Float> *rgb, int *w, int *h);   // imageio.cpp:60-79 (.pfm .png .tga)
bool HasExtension(const std::string &name, const char *ext);

// ---- Material (core/material.h:51-61).  `desc` = the parameter textures; when all of them are hit-independent
// (desc.textured == 0) `bsdf` holds what ComputeScatteringFunctions(allowMultipleLobes=true, Radiance) builds.
// BSSRDFTable (core/bssrdf.h:142-156) as ComputeBeamDiffusionBSSRDF fills it (host/bssrdf.cpp)
struct BSSRDFTableData {
    int nRho = 0, nRadius = 0;
    std::vector<float> rhoSamples, radiusSamples, profile, rhoEff, profileCDF;
};
std::shared_ptr<BSSRDFTableData> MakeBSSRDFTable(Float g, Float eta);
struct Material {
    std::string type;
    mi_material bsdf;
    mi_material_desc desc;
    std::shared_ptr<Material> m1, m2;   // MixMaterial
    mi_bssrdf_desc bssrdf;              // kind = MI_BSSRDF_NONE unless "subsurface" / "kdsubsurface"
    std::shared_ptr<BSSRDFTableData> table;
};
// MakeMaterial (api.cpp:541-611); returns nullptr for "" / "none"
std::shared_ptr<Material> MakeMaterial(const std::string &name, const TextureParams &mp,
                                       const std::map<std::string, std::shared_ptr<Material>> *named);

// ---- Lights (core/light.h:62-112)
struct AreaLightSpec { RGB Lemit; bool twoSided; };          // CreateDiffuseAreaLight diffuse.cpp:135-146
// radiance map of an InfiniteAreaLight after the reference constructor (host/envmap.cpp)
struct EnvMap {
    int width = 0, height = 0;
    std::vector<Float> rgb, condFunc, condCdf, condFuncInt, margFunc, margCdf;
    Float margFuncInt = 0;
    RGB powerLookup;   // Lmap->Lookup((.5,.5), .5) of Power() (infinite.cpp:86-90)
};
std::shared_ptr<EnvMap> CreateEnvMap(const std::string &filename, const RGB &L);
struct Light { mi_light l; std::shared_ptr<EnvMap> env; };   // point / spot / distant / infinite
// one entry of Scene::lights in file order (api.cpp:1418-1424): a LightSource light, or
// "every triangle of primitive `prim` is a DiffuseAreaLight" (api.cpp:1357-1366)
struct LightEntry { std::shared_ptr<Light> light; int prim; };
std::shared_ptr<Light> MakeLight(const std::string &name, const ParamSet &ps, const Transform &light2world);

// ---- Primitive (core/primitive.h:51-64,119-127)
// Sphere (shapes/sphere.h:46-77): the one quadric carried by this path; constructor results only
struct SphereShape {
    Transform o2w, w2o;
    bool reverseOrientation = false, transformSwapsHandedness = false;
    Float radius = 1, zMin = -1, zMax = 1, thetaMin = 0, thetaMax = 0, phiMax = 0;
    Float Area() const { return phiMax * radius * (zMax - zMin); }   // sphere.cpp:219
    Bounds3 WorldBound() const;                                        // ObjectToWorld(ObjectBound()) shape.cpp:52
};
std::shared_ptr<SphereShape> CreateSphereShape(const Transform &o2w, bool reverseOrientation, const ParamSet &ps);

// TransformedPrimitive (primitive.h:92-117) over the object `object` of Scene::objects: in the two-level mode (the default; PBRT_AMD_INSTANCING=0: flattened), otherwise
// instances are flattened into world-space copies when they are instantiated
struct InstanceRef {
    int object = 0;
    Transform i2w;           // PrimitiveToWorld (start transform)
    Bounds3 worldBound;      // PrimitiveToWorld.MotionBounds(primitive->WorldBound()), primitive.h:104-106
};
struct GeometricPrimitive {   // primitive.h:66-90: one per Shape *mesh* here, or one Sphere / one instance (then `shape` is an empty mesh)
    std::shared_ptr<TriangleMesh> shape;
    std::shared_ptr<SphereShape> sphere;
    std::shared_ptr<InstanceRef> instance;
    std::shared_ptr<Material> material;
    std::shared_ptr<AreaLightSpec> areaLight;
    int mediumInside = -1, mediumOutside = -1;   // MediumInterface (medium.h:100-110) as indices into Scene::media; -1 = none
};

// A participating medium as MakeMedium builds it (api.cpp:685-731): the POD record + the density grid it points into
struct MediumSpec {
    mi_medium m;
    std::vector<float> density;
};

// ---- Aggregate: BVHAccel (accelerators/bvh.h:52-99)
class BVHAccel {
  public:
    enum class SplitMethod { SAH, HLBVH, Middle, EqualCounts };
    struct PrimRef { uint32_t prim, tri; };   // (GeometricPrimitive index, triangle within its mesh)
    BVHAccel(const std::vector<GeometricPrimitive> &prims, int maxPrimsInNode, SplitMethod m);
    Bounds3 WorldBound() const;
    std::vector<mi_bvh2_node> nodes;      // LinearBVHNode[] (bvh.cpp:95-104)
    std::vector<PrimRef> primitives;      // ordered primitives (bvh.cpp:205)
    int maxPrimsInNode;
    SplitMethod splitMethod;
};
std::shared_ptr<BVHAccel> CreateBVHAccelerator(const std::vector<GeometricPrimitive> &prims, const ParamSet &ps);

// ---- Filter (core/filter.h:48-58, filters/*.cpp)
struct Filter {
    std::string name;
    Float rx, ry;
    Float p0 = 0, p1 = 0;   // gaussian: alpha,(expX) | mitchell: B,C | sinc: tau
    Float expX = 0, expY = 0;
    Float Evaluate(Float x, Float y) const;
};
std::unique_ptr<Filter> MakeFilter(const std::string &name, const ParamSet &ps);

// ---- Film (core/film.h:58-187)
class Film {
  public:
    Film(int xres, int yres, const Float crop[4], std::unique_ptr<Filter> filt, Float diagonal,
         const std::string &filename, Float scale, Float maxSampleLuminance);
    void GetSampleBounds(int mn[2], int mx[2]) const;         // film.cpp:80-86
    // MergeFilmTile over the whole cropped film from {rgb contribSum, filterWeightSum} records
    // (film.cpp:117-130): RGB->XYZ then accumulate.
    void MergeFilm(const float *rgbw);
    void Clear();
    void WriteImage(Float splatScale = 1);                    // film.cpp:168-210
    std::vector<Float> FinalRGB() const;                      // the array WriteImage hands to imageio
    int fullResolution[2];
    int cropMin[2], cropMax[2];
    std::unique_ptr<Filter> filter;
    std::string filename;
    Float diagonal, scale, maxSampleLuminance;
    Float filterTable[MI_FILTER_TABLE_WIDTH * MI_FILTER_TABLE_WIDTH];
    struct Pixel { Float xyz[3]; Float filterWeightSum; };
    std::vector<Pixel> pixels;
};
Film *CreateFilm(const ParamSet &ps, std::unique_ptr<Filter> filter);
bool WriteImage(const std::string &name, const Float *rgb, const int cropMin[2], const int cropMax[2],
                const int fullRes[2]);   // imageio.cpp:81-122: .pfm (float32), .exr (half, uncompressed)
bool ReadImagePFM(const std::string &name, std::vector<Float> *rgb, int *w, int *h);

// ---- Camera (core/camera.h:50-115, cameras/perspective.cpp)
struct PerspectiveCamera {
    PerspectiveCamera(const Transform &cameraToWorld, const Float screenWindow[4], Float shutterOpen,
                      Float shutterClose, Float lensRadius, Float focalDistance, Float fov, Film *film);
    Transform CameraToWorld, CameraToScreen, RasterToCamera, ScreenToRaster, RasterToScreen;
    Float shutterOpen, shutterClose, lensRadius, focalDistance;
    Vec3 dxCamera, dyCamera;
    std::unique_ptr<Film> film;   // camera owns the film (camera.cpp:42)
};
PerspectiveCamera *CreatePerspectiveCamera(const ParamSet &ps, const Transform &cam2world, Film *film);

// ---- Samplers.  Only the constructor results are kept (the per-sample work is the device's): the GlobalSamplers SobolSampler
// (samplers/sobol.h:48-69) and HaltonSampler (samplers/halton.{h,cpp}, pbrt's default), and -- ABI v11 -- the samplers that draw from one
// PCG32 stream per tile: RandomSampler (samplers/random.cpp), StratifiedSampler (samplers/stratified.cpp), ZeroTwoSequenceSampler
// (samplers/zerotwosequence.cpp; "02sequence" / "lowdiscrepancy")
struct Sampler {
    enum Kind { Sobol, Halton, Random, Stratified, ZeroTwo } kind = Sobol;
    // the tile-serial ones
    int nSampledDimensions = 0, xPixelSamples = 1, yPixelSamples = 1;
    bool jitterSamples = true;
    int64_t samplesPerPixel = 1;
    int sampleMin[2] = {0, 0}, sampleMax[2] = {0, 0};
    // Sobol
    int resolution = 1, log2Resolution = 0;
    // Halton
    int baseScales[2] = {1, 1}, baseExponents[2] = {0, 0}, sampleStride = 1, multInverse[2] = {0, 0};
    bool sampleAtPixelCenter = false;
};
struct SobolSampler : Sampler {
    SobolSampler(int64_t spp, const int sampleMin[2], const int sampleMax[2]);
};
struct HaltonSampler : Sampler {
    HaltonSampler(int64_t spp, const int sampleMin[2], const int sampleMax[2], bool sampleAtPixelCenter);
};
struct TileSerialSampler : Sampler {   // Random / Stratified / ZeroTwo: what their Create* functions read (random.cpp:75-78, stratified.cpp:78-86, zerotwosequence.cpp:76-81)
    TileSerialSampler(Kind kind, const ParamSet &ps, const int sampleMin[2], const int sampleMax[2]);
};

// ---- Scene (core/scene.h:50-80)
class Scene {
  public:
    Scene(std::shared_ptr<BVHAccel> aggregate, std::vector<GeometricPrimitive> prims, std::vector<LightEntry> lights);
    const Bounds3 &WorldBound() const { return worldBound; }
    std::shared_ptr<BVHAccel> aggregate;
    std::vector<GeometricPrimitive> primitives;
    std::vector<LightEntry> lights;   // scene.lights order; area lights expand to one light per triangle
    Bounds3 worldBound;
    struct ObjectDef { std::vector<GeometricPrimitive> prims; std::shared_ptr<BVHAccel> accel; };
    std::vector<ObjectDef> objects;   // instanced objects (two-level mode only): each with the BVHAccel the reference builds for it (api.cpp:1563-1572)
    std::shared_ptr<TextureStore> textures;   // nodes / images the materials and alpha masks refer to
    std::vector<std::shared_ptr<MediumSpec>> media;   // RenderOptions::namedMedia in definition order
    int cameraMedium = -1;                             // Camera::medium (camera.h:70)
};

// ---- Integrator (core/integrator.h:53-58) and the GPU path integrator
class Integrator {
  public:
    virtual ~Integrator() {}
    virtual void Render(const Scene &scene) = 0;
};

// Owns the flattened arrays an mi_scene_desc points into.
struct FlatScene {
    mi_scene_desc desc;
    std::vector<float> P, N, UV, lightFunc, lightCdf;
    std::vector<uint32_t> triIndices, triMesh;
    std::vector<int32_t> triLight;
    std::vector<mi_mesh> meshes;
    std::vector<mi_material> materials;
    std::vector<mi_light> lights;
    std::vector<mi_envmap> envmaps;
    std::vector<mi_sphere> spheres;
    std::vector<std::shared_ptr<EnvMap>> envKeep;
    std::vector<mi_texture> textures;
    std::vector<mi_image> images;
    std::vector<mi_material_desc> materialDescs;
    std::vector<int32_t> meshAlpha;
    std::shared_ptr<TextureStore> texKeep;
    std::vector<mi_bvh2_node> nodes;       // two-level mode: top-level nodes followed by every object's
    std::vector<mi_instance> instances;
    std::vector<mi_object> objects;
    std::vector<mi_medium> media;
    std::vector<int32_t> meshMedium;
    std::vector<mi_bssrdf_desc> materialBssrdf;
    std::vector<mi_bssrdf_table> bssrdfTables;
    std::vector<std::shared_ptr<BSSRDFTableData>> tableKeep;
    std::vector<std::shared_ptr<MediumSpec>> mediaKeep;
};

class WavefrontPathIntegrator : public Integrator {   // stands where PathIntegrator does (path.h:49-71)
  public:
    WavefrontPathIntegrator(int maxDepth, std::shared_ptr<PerspectiveCamera> camera,
                            std::shared_ptr<Sampler> sampler, const int pixelMin[2], const int pixelMax[2],
                            Float rrThreshold, const std::string &lightSampleStrategy);
    void Render(const Scene &scene) override;   // flatten -> mi_scene_upload -> mi_render -> Film
    // Flatten(scene): everything Render hands to the device, as POD
    std::unique_ptr<FlatScene> Flatten(const Scene &scene) const;
    int maxDepth;
    std::shared_ptr<PerspectiveCamera> camera;
    std::shared_ptr<Sampler> sampler;
    int pixelMin[2], pixelMax[2];
    Float rrThreshold;
    std::string lightSampleStrategy;
    int nGpus = 1;   // --gpus: tile-sharded over this many devices in-process
    bool volPath = false;   // created as Integrator "volpath" (integrators/volpath.cpp:192-214): same parameters, handleMedia = true
};
WavefrontPathIntegrator *CreatePathIntegrator(const ParamSet &ps, std::shared_ptr<Sampler> sampler,
                                              std::shared_ptr<PerspectiveCamera> camera);

}  // namespace pbrt_amd
