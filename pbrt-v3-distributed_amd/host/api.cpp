// Scene-description state machine.  Follows the reference's core/api.cpp: graphics-state and
// CTM stacks (:353-365, :1124-1192), named coordinate systems (:988-1006), options block vs
// world block checks (:375-403), defaults (:166-176), pbrtShape creating one area light per
// emissive triangle (:1333-1424), object instancing (:1524-1592; static instances are flattened
// here), and pbrtWorldEnd building Scene + Integrator then calling Render once (:1594-1653).
#include "api.h"

#include <map>

namespace pbrt_amd {

std::string g_imageFileOverride;
bool g_twoLevelInstancing = true;   // keep ObjectInstance as TransformedPrimitive + per-object BVH, the reference's own structure (default since round 2: the device's two-level kernels are validated); PBRT_AMD_INSTANCING=0 flattens instances into world-space copies
Float g_cropWindow[4] = {0, 1, 0, 1};
bool g_quickRender = false;

namespace {

enum class APIState { Uninitialized, OptionsBlock, WorldBlock };
const int MaxTransforms = 2;
const uint32_t StartTransformBits = 1, EndTransformBits = 2, AllTransformsBits = 3;

struct TransformSet {
    Transform t[MaxTransforms];
    Transform &operator[](int i) { return t[i]; }
    const Transform &operator[](int i) const { return t[i]; }
    bool IsAnimated() const { return !(t[0] == t[1]); }
};
TransformSet InverseSet(const TransformSet &ts) { TransformSet r; for (int i = 0; i < MaxTransforms; ++i) r.t[i] = Inverse(ts.t[i]); return r; }

struct MaterialInstance { std::string name; std::shared_ptr<Material> material; ParamSet params; };

struct GraphicsState {
    TextureMaps textures;
    std::map<std::string, std::shared_ptr<MaterialInstance>> namedMaterials;
    std::shared_ptr<MaterialInstance> currentMaterial;
    ParamSet areaLightParams;
    std::string areaLight;
    bool reverseOrientation = false;
    std::string currentInsideMedium, currentOutsideMedium;   // api.cpp:216
    GraphicsState() {
        ParamSet empty;
        TextureParams tp(empty, empty, textures);
        currentMaterial = std::make_shared<MaterialInstance>(MaterialInstance{"matte", MakeMaterial("matte", tp, nullptr), ParamSet()});
    }
};

struct RenderOptions {   // api.cpp:150-186
    Float transformStartTime = 0, transformEndTime = 1;
    std::string FilterName = "box", FilmName = "image", SamplerName = "halton", AcceleratorName = "bvh",
                IntegratorName = "path", CameraName = "perspective";
    ParamSet FilterParams, FilmParams, SamplerParams, AcceleratorParams, IntegratorParams, CameraParams;
    TransformSet CameraToWorld;
    std::vector<LightEntry> lights;
    std::vector<GeometricPrimitive> primitives;
    std::map<std::string, std::vector<GeometricPrimitive>> instances;
    std::vector<GeometricPrimitive> *currentInstance = nullptr;
    std::map<std::string, int> namedMedia;                 // api.cpp:179, as indices into `media`
    std::vector<std::shared_ptr<MediumSpec>> media;
    // two-level mode (the default; PBRT_AMD_INSTANCING=0 flattens): the objects that were instantiated, each with its own BVHAccel
    std::vector<Scene::ObjectDef> objectDefs;
    std::map<std::string, int> objectIndex;
    // flattened mode: the bounds the reference's TransformedPrimitives would have, and which primitives are flattened copies
    Bounds3 instanceBound;
    bool haveFlattenedInstances = false;
    std::vector<uint8_t> flattenedPrim;
};

APIState currentApiState = APIState::Uninitialized;
TransformSet curTransform;
uint32_t activeTransformBits = AllTransformsBits;
std::map<std::string, TransformSet> namedCoordinateSystems;
std::unique_ptr<RenderOptions> renderOptions;
GraphicsState graphicsState;
std::vector<GraphicsState> pushedGraphicsStates;
std::vector<TransformSet> pushedTransforms;
std::vector<uint32_t> pushedActiveTransformBits;
Options PbrtOptions;
std::unique_ptr<BuiltScene> builtScene;

#define VERIFY_INITIALIZED(func)                                                        \
    if (currentApiState == APIState::Uninitialized) {                                   \
        Error("pbrtInit() must be before calling \"%s()\". Ignoring.", func);           \
        return;                                                                         \
    } else
#define VERIFY_OPTIONS(func)                                                            \
    VERIFY_INITIALIZED(func);                                                           \
    if (currentApiState == APIState::WorldBlock) {                                      \
        Error("Options cannot be set inside world block; \"%s\" not allowed.  Ignoring.", func); \
        return;                                                                         \
    } else
#define VERIFY_WORLD(func)                                                              \
    VERIFY_INITIALIZED(func);                                                           \
    if (currentApiState == APIState::OptionsBlock) {                                    \
        Error("Scene description must be inside world block; \"%s\" not allowed. Ignoring.", func); \
        return;                                                                         \
    } else
#define FOR_ACTIVE_TRANSFORMS(expr) \
    for (int i = 0; i < MaxTransforms; ++i) if (activeTransformBits & (1 << i)) { expr }
#define WARN_IF_ANIMATED_TRANSFORM(func)                                                                         \
    do { if (curTransform.IsAnimated())                                                                          \
        Warning("Animated transformations set; ignoring for \"%s\" and using the start transform only", func);   \
    } while (false)

Matrix4x4 fromColumnMajor(const Float tr[16]) {   // api.cpp:922-925
    return Matrix4x4(tr[0], tr[4], tr[8], tr[12], tr[1], tr[5], tr[9], tr[13], tr[2], tr[6], tr[10], tr[14], tr[3],
                     tr[7], tr[11], tr[15]);
}

}  // namespace

int g_optionNThreads = 0;
void pbrtInit(const Options &opt) {
    g_optionNThreads = opt.nThreads;
    PbrtOptions = opt;
    g_imageFileOverride = opt.imageFile;
    g_quickRender = opt.quickRender;
    g_quiet = opt.quiet;
    g_unsupportedCount = 0;
    g_cropWindow[0] = opt.cropWindow[0][0]; g_cropWindow[1] = opt.cropWindow[0][1];
    g_cropWindow[2] = opt.cropWindow[1][0]; g_cropWindow[3] = opt.cropWindow[1][1];
    if (currentApiState != APIState::Uninitialized) Error("pbrtInit() has already been called.");
    currentApiState = APIState::OptionsBlock;
    renderOptions.reset(new RenderOptions);
    { const char *e = std::getenv("PBRT_AMD_INSTANCING"); g_twoLevelInstancing = !(e && e[0] == '0'); }
    ResetTextures();
    graphicsState = GraphicsState();
    curTransform = TransformSet();
    activeTransformBits = AllTransformsBits;
    namedCoordinateSystems.clear();
    pushedGraphicsStates.clear(); pushedTransforms.clear(); pushedActiveTransformBits.clear();
}

void pbrtCleanup() {
    if (currentApiState == APIState::Uninitialized) Error("pbrtCleanup() called without pbrtInit().");
    else if (currentApiState == APIState::WorldBlock) Error("pbrtCleanup() called while inside world block.");
    currentApiState = APIState::Uninitialized;
    renderOptions.reset(nullptr);
}

void pbrtIdentity() { VERIFY_INITIALIZED("Identity"); FOR_ACTIVE_TRANSFORMS(curTransform[i] = Transform();) }
void pbrtTranslate(Float dx, Float dy, Float dz) {
    VERIFY_INITIALIZED("Translate");
    FOR_ACTIVE_TRANSFORMS(curTransform[i] = curTransform[i] * Translate(Vec3(dx, dy, dz));)
}
void pbrtTransform(Float tr[16]) {
    VERIFY_INITIALIZED("Transform");
    FOR_ACTIVE_TRANSFORMS(curTransform[i] = Transform(fromColumnMajor(tr));)
}
void pbrtConcatTransform(Float tr[16]) {
    VERIFY_INITIALIZED("ConcatTransform");
    FOR_ACTIVE_TRANSFORMS(curTransform[i] = curTransform[i] * Transform(fromColumnMajor(tr));)
}
void pbrtRotate(Float angle, Float dx, Float dy, Float dz) {
    VERIFY_INITIALIZED("Rotate");
    FOR_ACTIVE_TRANSFORMS(curTransform[i] = curTransform[i] * Rotate(angle, Vec3(dx, dy, dz));)
}
void pbrtScale(Float sx, Float sy, Float sz) {
    VERIFY_INITIALIZED("Scale");
    FOR_ACTIVE_TRANSFORMS(curTransform[i] = curTransform[i] * Scale(sx, sy, sz);)
}
void pbrtLookAt(Float ex, Float ey, Float ez, Float lx, Float ly, Float lz, Float ux, Float uy, Float uz) {
    VERIFY_INITIALIZED("LookAt");
    bool ok;
    Transform lookAt = LookAt(Vec3(ex, ey, ez), Vec3(lx, ly, lz), Vec3(ux, uy, uz), &ok);
    if (!ok)
        Error("\"up\" vector (%f, %f, %f) and viewing direction passed to LookAt are pointing in the same direction.  "
              "Using the identity transformation.", ux, uy, uz);
    FOR_ACTIVE_TRANSFORMS(curTransform[i] = curTransform[i] * lookAt;)
}
void pbrtCoordinateSystem(const std::string &name) { VERIFY_INITIALIZED("CoordinateSystem"); namedCoordinateSystems[name] = curTransform; }
void pbrtCoordSysTransform(const std::string &name) {
    VERIFY_INITIALIZED("CoordSysTransform");
    if (namedCoordinateSystems.find(name) != namedCoordinateSystems.end()) curTransform = namedCoordinateSystems[name];
    else Warning("Couldn't find named coordinate system \"%s\"", name.c_str());
}
void pbrtActiveTransformAll() { activeTransformBits = AllTransformsBits; }
void pbrtActiveTransformEndTime() { activeTransformBits = EndTransformBits; }
void pbrtActiveTransformStartTime() { activeTransformBits = StartTransformBits; }
void pbrtTransformTimes(Float start, Float end) {
    VERIFY_OPTIONS("TransformTimes");
    renderOptions->transformStartTime = start; renderOptions->transformEndTime = end;
}
void pbrtPixelFilter(const std::string &name, const ParamSet &params) {
    VERIFY_OPTIONS("PixelFilter"); renderOptions->FilterName = name; renderOptions->FilterParams = params;
}
void pbrtFilm(const std::string &type, const ParamSet &params) {
    VERIFY_OPTIONS("Film"); renderOptions->FilmParams = params; renderOptions->FilmName = type;
}
void pbrtSampler(const std::string &name, const ParamSet &params) {
    VERIFY_OPTIONS("Sampler"); renderOptions->SamplerName = name; renderOptions->SamplerParams = params;
}
void pbrtAccelerator(const std::string &name, const ParamSet &params) {
    VERIFY_OPTIONS("Accelerator"); renderOptions->AcceleratorName = name; renderOptions->AcceleratorParams = params;
}
void pbrtIntegrator(const std::string &name, const ParamSet &params) {
    VERIFY_OPTIONS("Integrator"); renderOptions->IntegratorName = name; renderOptions->IntegratorParams = params;
}
void pbrtCamera(const std::string &name, const ParamSet &params) {
    VERIFY_OPTIONS("Camera");
    renderOptions->CameraName = name;
    renderOptions->CameraParams = params;
    renderOptions->CameraToWorld = InverseSet(curTransform);
    namedCoordinateSystems["camera"] = renderOptions->CameraToWorld;
}
// GetMediumScatteringProperties core/medium.cpp:174-185 over the measured-coefficient table (:48-172)
bool GetMediumScatteringProperties(const std::string &name, RGB *sigma_a, RGB *sigma_prime_s) {
    struct MeasuredSS { const char *name; Float sps[3], sa[3]; };
    static const MeasuredSS table[] = {
#define P(n, s0, s1, s2, a0, a1, a2) {n, {(Float)s0, (Float)s1, (Float)s2}, {(Float)a0, (Float)a1, (Float)a2}},
#include "medium_presets.inc"
#undef P
    };
    for (const MeasuredSS &mss : table)
        if (name == mss.name) {
            *sigma_a = RGB(mss.sa[0], mss.sa[1], mss.sa[2]);
            *sigma_prime_s = RGB(mss.sps[0], mss.sps[1], mss.sps[2]);
            return true;
        }
    return false;
}
// MakeMedium core/api.cpp:685-731
static std::shared_ptr<MediumSpec> MakeMedium(const std::string &name, const ParamSet &ps, const Transform &medium2world) {
    RGB sig_a(.0011f, .0024f, .014f), sig_s(2.55f, 3.21f, 3.77f);
    std::string preset = ps.FindOneString("preset", "");
    bool found = GetMediumScatteringProperties(preset, &sig_a, &sig_s);
    if (preset != "" && !found) Warning("Material preset \"%s\" not found.  Using defaults.", preset.c_str());
    Float scale = ps.FindOneFloat("scale", 1.f);
    Float g = ps.FindOneFloat("g", 0.0f);
    sig_a = ps.FindOneSpectrum("sigma_a", sig_a) * scale;
    sig_s = ps.FindOneSpectrum("sigma_s", sig_s) * scale;
    auto spec = std::make_shared<MediumSpec>();
    mi_medium &m = spec->m;
    std::memset(&m, 0, sizeof(m));
    RGB sig_t = sig_a + sig_s;
    for (int c = 0; c < 3; ++c) { m.sigma_a[c] = sig_a.c[c]; m.sigma_s[c] = sig_s.c[c]; m.sigma_t[c] = sig_t.c[c]; }
    m.g = g;
    if (name == "homogeneous") {
        m.type = MI_MEDIUM_HOMOGENEOUS;
    } else if (name == "heterogeneous") {
        int nitems;
        const Float *data = ps.FindFloat("density", &nitems);
        if (!data) { Error("No \"density\" values provided for heterogeneous medium?"); return nullptr; }
        int nx = ps.FindOneInt("nx", 1), ny = ps.FindOneInt("ny", 1), nz = ps.FindOneInt("nz", 1);
        Vec3 p0 = ps.FindOnePoint3("p0", Vec3(0.f, 0.f, 0.f)), p1 = ps.FindOnePoint3("p1", Vec3(1.f, 1.f, 1.f));
        if (nitems != nx * ny * nz) {
            Error("GridDensityMedium has %d density values; expected nx*ny*nz = %d", nitems, nx * ny * nz);
            return nullptr;
        }
        Transform data2Medium = Translate(p0) * Scale(p1.x - p0.x, p1.y - p0.y, p1.z - p0.z);
        Transform worldToMedium = Inverse(medium2world * data2Medium);   // grid.h:61
        m.type = MI_MEDIUM_GRID;
        m.nx = nx; m.ny = ny; m.nz = nz;
        std::memcpy(m.world_to_medium, worldToMedium.m.m, sizeof(m.world_to_medium));
        spec->density.assign(data, data + nitems);
        m.sigma_t[0] = m.sigma_t[1] = m.sigma_t[2] = sig_t.c[0];   // grid.h:69-73
        if (!(sig_t.c[1] == sig_t.c[0] && sig_t.c[2] == sig_t.c[0])) Error("GridDensityMedium requires a spectrally uniform attenuation coefficient!");
        Float maxDensity = 0;
        for (int i = 0; i < nitems; ++i) maxDensity = std::max(maxDensity, data[i]);
        m.inv_max_density = 1 / maxDensity;
    } else {
        Warning("Medium \"%s\" unknown.", name.c_str());
        ps.ReportUnused();
        return nullptr;
    }
    ps.ReportUnused();
    return spec;
}
void pbrtMakeNamedMedium(const std::string &name, const ParamSet &params) {   // api.cpp:1093-1109
    VERIFY_INITIALIZED("MakeNamedMedium");
    std::string type = params.FindOneString("type", "");
    if (type == "") { Error("No parameter string \"type\" found in MakeNamedMedium"); return; }
    std::shared_ptr<MediumSpec> medium = MakeMedium(type, params, curTransform[0]);
    if (medium) {
        renderOptions->namedMedia[name] = (int)renderOptions->media.size();
        renderOptions->media.push_back(medium);
    }
}
void pbrtMediumInterface(const std::string &insideName, const std::string &outsideName) {   // api.cpp:1111-1121
    VERIFY_INITIALIZED("MediumInterface");
    graphicsState.currentInsideMedium = insideName;
    graphicsState.currentOutsideMedium = outsideName;
}
// GraphicsState::CreateMediumInterface api.cpp:1496-1516
static void CreateMediumInterface(int *inside, int *outside) {
    *inside = *outside = -1;
    auto lookup = [](const std::string &n, int *out) {
        if (n == "") return;
        auto it = renderOptions->namedMedia.find(n);
        if (it != renderOptions->namedMedia.end()) *out = it->second;
        else Error("Named medium \"%s\" undefined.", n.c_str());
    };
    lookup(graphicsState.currentInsideMedium, inside);
    lookup(graphicsState.currentOutsideMedium, outside);
}

void pbrtWorldBegin() {
    VERIFY_OPTIONS("WorldBegin");
    currentApiState = APIState::WorldBlock;
    for (int i = 0; i < MaxTransforms; ++i) curTransform[i] = Transform();
    activeTransformBits = AllTransformsBits;
    namedCoordinateSystems["world"] = curTransform;
}
void pbrtAttributeBegin() {
    VERIFY_WORLD("AttributeBegin");
    pushedGraphicsStates.push_back(graphicsState);
    pushedTransforms.push_back(curTransform);
    pushedActiveTransformBits.push_back(activeTransformBits);
}
void pbrtAttributeEnd() {
    VERIFY_WORLD("AttributeEnd");
    if (!pushedGraphicsStates.size()) { Error("Unmatched pbrtAttributeEnd() encountered. Ignoring it."); return; }
    graphicsState = std::move(pushedGraphicsStates.back()); pushedGraphicsStates.pop_back();
    curTransform = pushedTransforms.back(); pushedTransforms.pop_back();
    activeTransformBits = pushedActiveTransformBits.back(); pushedActiveTransformBits.pop_back();
}
void pbrtTransformBegin() {
    VERIFY_WORLD("TransformBegin");
    pushedTransforms.push_back(curTransform);
    pushedActiveTransformBits.push_back(activeTransformBits);
}
void pbrtTransformEnd() {
    VERIFY_WORLD("TransformEnd");
    if (!pushedTransforms.size()) { Error("Unmatched pbrtTransformEnd() encountered. Ignoring it."); return; }
    curTransform = pushedTransforms.back(); pushedTransforms.pop_back();
    activeTransformBits = pushedActiveTransformBits.back(); pushedActiveTransformBits.pop_back();
}

// pbrtTexture (api.cpp:1193-1243): the node goes into the graphics state's name map (copy-on-write there, a plain copy here)
void pbrtTexture(const std::string &name, const std::string &type, const std::string &texname, const ParamSet &params) {
    VERIFY_WORLD("Texture");
    TextureParams tp(params, params, graphicsState.textures);
    bool isFloat = type == "float", isSpec = type == "color" || type == "spectrum";
    if (!isFloat && !isSpec) { Error("Texture type \"%s\" unknown.", type.c_str()); return; }
    if ((isFloat && graphicsState.textures.floats.count(name)) || (isSpec && graphicsState.textures.spectra.count(name)))
        Warning("Texture \"%s\" being redefined", name.c_str());
    WARN_IF_ANIMATED_TRANSFORM("Texture");
    int node = isFloat ? MakeFloatTexture(texname, curTransform[0], tp) : MakeSpectrumTexture(texname, curTransform[0], tp);
    if (node < 0) return;
    if (isFloat) graphicsState.textures.floats[name] = node;
    else graphicsState.textures.spectra[name] = node;
}

static std::map<std::string, std::shared_ptr<Material>> namedMaterialMap() {
    std::map<std::string, std::shared_ptr<Material>> m;
    for (auto &kv : graphicsState.namedMaterials) m[kv.first] = kv.second->material;
    return m;
}

void pbrtMaterial(const std::string &name, const ParamSet &params) {
    VERIFY_WORLD("Material");
    ParamSet empty;
    TextureParams mp(params, empty, graphicsState.textures);
    auto named = namedMaterialMap();
    std::shared_ptr<Material> mtl = MakeMaterial(name, mp, &named);
    graphicsState.currentMaterial = std::make_shared<MaterialInstance>(MaterialInstance{name, mtl, params});
}
void pbrtMakeNamedMaterial(const std::string &name, const ParamSet &params) {
    VERIFY_WORLD("MakeNamedMaterial");
    ParamSet empty;
    TextureParams mp(params, empty, graphicsState.textures);
    std::string matName = mp.FindString("type");
    WARN_IF_ANIMATED_TRANSFORM("MakeNamedMaterial");
    if (matName == "") Error("No parameter string \"type\" found in MakeNamedMaterial");
    auto named = namedMaterialMap();
    std::shared_ptr<Material> mtl = MakeMaterial(matName, mp, &named);
    if (graphicsState.namedMaterials.count(name)) Warning("Named material \"%s\" redefined.", name.c_str());
    graphicsState.namedMaterials[name] = std::make_shared<MaterialInstance>(MaterialInstance{matName, mtl, params});
}
void pbrtNamedMaterial(const std::string &name) {
    VERIFY_WORLD("NamedMaterial");
    auto it = graphicsState.namedMaterials.find(name);
    if (it == graphicsState.namedMaterials.end()) { Error("NamedMaterial \"%s\" unknown.", name.c_str()); return; }
    graphicsState.currentMaterial = it->second;
}

// MakeLight (api.cpp:733-757) for the light types this path carries
std::shared_ptr<Light> MakeLight(const std::string &name, const ParamSet &ps, const Transform &l2w) {
    auto light = std::make_shared<Light>();
    std::memset(&light->l, 0, sizeof(light->l));
    light->l.tri = -1;
    if (name == "point") {   // point.cpp:80-88: pLight = (Translate(from) * light2world)(0,0,0)
        RGB I = ps.FindOneSpectrum("I", RGB(1.0)), sc = ps.FindOneSpectrum("scale", RGB(1.0));
        Vec3 P = ps.FindOnePoint3("from", Vec3(0, 0, 0));
        Transform t = Translate(Vec3(P.x, P.y, P.z)) * l2w;
        Vec3 p = t.Point(Vec3(0, 0, 0));
        RGB Is = I * sc;
        light->l.type = MI_LIGHT_POINT;
        for (int i = 0; i < 3; ++i) { light->l.L[i] = Is.c[i]; light->l.pos[i] = p[i]; }
    } else if (name == "distant") {   // distant.cpp:44-49,94-102
        RGB L = ps.FindOneSpectrum("L", RGB(1.0)), sc = ps.FindOneSpectrum("scale", RGB(1.0));
        Vec3 from = ps.FindOnePoint3("from", Vec3(0, 0, 0)), to = ps.FindOnePoint3("to", Vec3(0, 0, 1));
        Vec3 w = Normalize(l2w.Vector(from - to));
        RGB Ls = L * sc;
        light->l.type = MI_LIGHT_DISTANT;
        for (int i = 0; i < 3; ++i) { light->l.L[i] = Ls.c[i]; light->l.pos[i] = w[i]; }
    } else if (name == "spot") {   // CreateSpotLight lights/spot.cpp:104-125, constructor :40-52
        RGB I = ps.FindOneSpectrum("I", RGB(1.0)), sc = ps.FindOneSpectrum("scale", RGB(1.0));
        Float coneangle = ps.FindOneFloat("coneangle", 30.), conedelta = ps.FindOneFloat("conedeltaangle", 5.);
        Vec3 from = ps.FindOnePoint3("from", Vec3(0, 0, 0)), to = ps.FindOnePoint3("to", Vec3(0, 0, 1));
        Vec3 dir = Normalize(to - from), du, dv;
        CoordinateSystem(dir, &du, &dv);
        Matrix4x4 dz;
        dz.m[0][0] = du.x; dz.m[0][1] = du.y; dz.m[0][2] = du.z; dz.m[0][3] = 0;
        dz.m[1][0] = dv.x; dz.m[1][1] = dv.y; dz.m[1][2] = dv.z; dz.m[1][3] = 0;
        dz.m[2][0] = dir.x; dz.m[2][1] = dir.y; dz.m[2][2] = dir.z; dz.m[2][3] = 0;
        dz.m[3][0] = 0; dz.m[3][1] = 0; dz.m[3][2] = 0; dz.m[3][3] = 1;
        Transform dirToZ(dz);
        Transform light2world = l2w * Translate(Vec3(from.x, from.y, from.z)) * Transform(dirToZ.mInv, dirToZ.m);
        Vec3 p = light2world.Point(Vec3(0, 0, 0));
        RGB Is = I * sc;
        Float totalWidth = coneangle, falloffStart = coneangle - conedelta;
        light->l.type = MI_LIGHT_SPOT;
        for (int i = 0; i < 3; ++i) { light->l.L[i] = Is.c[i]; light->l.pos[i] = p[i]; }
        for (int r = 0; r < 3; ++r) for (int c2 = 0; c2 < 3; ++c2) light->l.frame[3 * r + c2] = light2world.mInv.m[r][c2];   // WorldToLight = Inverse(LightToWorld)
        light->l.cos_total_width = std::cos(Radians(totalWidth));
        light->l.cos_falloff_start = std::cos(Radians(falloffStart));
    } else if (name == "infinite" || name == "exinfinite") {   // CreateInfiniteLight lights/infinite.cpp:204-215
        RGB L = ps.FindOneSpectrum("L", RGB(1.0)), sc = ps.FindOneSpectrum("scale", RGB(1.0));
        std::string texmap = ps.FindOneFilename("mapname", "");
        ps.FindOneInt("samples", 1); ps.FindOneInt("nsamples", 1);
        RGB Ls = L * sc;
        light->l.type = MI_LIGHT_INFINITE;
        for (int i = 0; i < 3; ++i) light->l.L[i] = Ls.c[i];
        for (int r = 0; r < 3; ++r) for (int c2 = 0; c2 < 3; ++c2) { light->l.frame[3 * r + c2] = l2w.mInv.m[r][c2]; light->l.l2w[3 * r + c2] = l2w.m.m[r][c2]; }
        if (texmap != "") {
            light->env = CreateEnvMap(texmap, Ls);   // texels * L (infinite.cpp:52-56); unreadable -> constant L, as the reference
        } else if (!l2w.IsIdentity()) {
            // A constant light is a 1x1 radiance map to the reference (infinite.cpp:58-62), and its LightToWorld decides which direction a
            // sample (u, v) becomes (:116-118).  The map-less fast path of the device assumes the identity there, so a transformed constant
            // light is handed over as that 1x1 map.
            light->env = CreateEnvMap("", Ls);
        }
    } else {
        // the reference's remaining light types would light the scene; rendering without them gives a plausible but wrong image -> refuse
        if (name == "goniometric" || name == "projection")
            Unsupported("LightSource \"%s\" has no counterpart on this path (lights/goniometric.cpp, lights/projection.cpp: SURVEY.md s.2 row 25)", name.c_str());
        else Warning("Light \"%s\" unknown.", name.c_str());   // api.cpp:724 in the reference
        ps.ReportUnused();
        return nullptr;
    }
    ps.ReportUnused();
    return light;
}

void pbrtLightSource(const std::string &name, const ParamSet &params) {
    VERIFY_WORLD("LightSource");
    WARN_IF_ANIMATED_TRANSFORM("LightSource");
    std::shared_ptr<Light> lt = MakeLight(name, params, curTransform[0]);
    if (!lt) Error("LightSource: light type \"%s\" unknown.", name.c_str());
    else renderOptions->lights.push_back(LightEntry{lt, -1});
}
void pbrtAreaLightSource(const std::string &name, const ParamSet &params) {
    VERIFY_WORLD("AreaLightSource");
    graphicsState.areaLight = name;
    graphicsState.areaLightParams = params;
}

void pbrtShape(const std::string &name, const ParamSet &params) {
    VERIFY_WORLD("Shape");
    if (curTransform.IsAnimated())
        Unsupported("Shape \"%s\" under an animated transformation (TransformedPrimitive with an AnimatedTransform, primitive.cpp:76-96, transform.h:412) has no counterpart on this path", name.c_str());
    std::shared_ptr<TriangleMesh> shape;
    std::shared_ptr<SphereShape> sphere;
    if (name == "sphere") {
        sphere = CreateSphereShape(curTransform[0], graphicsState.reverseOrientation, params);
        shape = std::make_shared<TriangleMesh>();   // empty: the primitive is the sphere
    } else {
        shape = MakeShapes(name, curTransform[0], graphicsState.reverseOrientation, params);
        if (!shape || shape->nTriangles() == 0) return;
        if (name == "trianglemesh" || name == "plymesh") {   // alpha masks: triangle.cpp:717-741, plymesh.cpp:259-286
            auto alphaNode = [&](const char *pname) -> int {
                std::string tn = params.FindTexture(pname);
                if (tn != "") {
                    auto it = graphicsState.textures.floats.find(tn);
                    if (it != graphicsState.textures.floats.end()) return it->second;
                    Error("Couldn't find float texture \"%s\" for \"%s\" parameter", tn.c_str(), pname);
                    return -1;
                }
                if (params.FindOneFloat(pname, 1.f) == 0.f) return ConstantTextureNode(false, RGB(0.f));
                return -1;
            };
            shape->alphaTex = alphaNode("alpha");
            shape->shadowAlphaTex = alphaNode("shadowalpha");
        }
    }
    // GraphicsState::GetMaterialForShape (api.cpp:1771-1800): shape parameters may override material ones
    std::shared_ptr<Material> mtl;
    {
        bool shapeHasMaterialParams = false;   // shapeMaySetMaterialParameters heuristics, api.cpp:1431-1480
        for (auto &it : params.items()) {
            size_t nv = it.type == ParamType::String || it.type == ParamType::Texture ? it.s.size()
                        : (it.type == ParamType::Int || it.type == ParamType::Bool) ? it.i.size()
                        : (it.type == ParamType::Float) ? it.f.size()
                        : (it.type == ParamType::Point2 || it.type == ParamType::Vector2) ? it.f.size() / 2 : it.f.size() / 3;
            if (it.type == ParamType::Texture) { if (it.name != "alpha" && it.name != "shadowalpha") shapeHasMaterialParams = true; }
            else if (it.type == ParamType::Float) { if (nv == 1 && it.name != "radius") shapeHasMaterialParams = true; }
            else if (it.type == ParamType::String) { if (nv == 1 && it.name != "filename" && it.name != "type" && it.name != "scheme") shapeHasMaterialParams = true; }
            else if (nv == 1) shapeHasMaterialParams = true;
        }
        if (shapeHasMaterialParams && graphicsState.currentMaterial->material) {
            TextureParams mp(params, graphicsState.currentMaterial->params, graphicsState.textures);
            auto named = namedMaterialMap();
            mtl = MakeMaterial(graphicsState.currentMaterial->name, mp, &named);
        } else
            mtl = graphicsState.currentMaterial->material;
    }
    params.ReportUnused();
    GeometricPrimitive gp;
    gp.shape = shape;
    gp.sphere = sphere;
    gp.material = mtl;
    CreateMediumInterface(&gp.mediumInside, &gp.mediumOutside);   // api.cpp:1355
    if (graphicsState.areaLight != "") {   // MakeAreaLight api.cpp:759-772, CreateDiffuseAreaLight diffuse.cpp:135-146
        if (graphicsState.areaLight == "area" || graphicsState.areaLight == "diffuse") {
            const ParamSet &ap = graphicsState.areaLightParams;
            RGB L = ap.FindOneSpectrum("L", RGB(1.0)), sc = ap.FindOneSpectrum("scale", RGB(1.0));
            ap.FindOneInt("samples", ap.FindOneInt("nsamples", 1));
            bool twoSided = ap.FindOneBool("twosided", false);
            gp.areaLight = std::make_shared<AreaLightSpec>(AreaLightSpec{L * sc, twoSided});
            ap.ReportUnused();
        } else
            Warning("Area light \"%s\" unknown.", graphicsState.areaLight.c_str());
    }
    if (renderOptions->currentInstance) {
        if (gp.areaLight) Warning("Area lights not supported with object instancing");
        gp.areaLight = nullptr;
        renderOptions->currentInstance->push_back(gp);
    } else {
        renderOptions->primitives.push_back(gp);
        if (gp.areaLight) renderOptions->lights.push_back(LightEntry{nullptr, (int)renderOptions->primitives.size() - 1});
    }
}

void pbrtReverseOrientation() { VERIFY_WORLD("ReverseOrientation"); graphicsState.reverseOrientation = !graphicsState.reverseOrientation; }

void pbrtObjectBegin(const std::string &name) {
    VERIFY_WORLD("ObjectBegin");
    pbrtAttributeBegin();
    if (renderOptions->currentInstance) Error("ObjectBegin called inside of instance definition");
    renderOptions->instances[name] = std::vector<GeometricPrimitive>();
    renderOptions->currentInstance = &renderOptions->instances[name];
}
void pbrtObjectEnd() {
    VERIFY_WORLD("ObjectEnd");
    if (!renderOptions->currentInstance) Error("ObjectEnd called outside of instance definition");
    renderOptions->currentInstance = nullptr;
    pbrtAttributeEnd();
}
// Transform::operator()(const Bounds3f &) core/transform.cpp:141-153: the box of the eight transformed corners
static Bounds3 TransformBounds(const Transform &t, const Bounds3 &b) {
    Bounds3 wb(t.Point(Vec3(b.pMin.x, b.pMin.y, b.pMin.z)));
    wb = Union(wb, t.Point(Vec3(b.pMax.x, b.pMin.y, b.pMin.z)));
    wb = Union(wb, t.Point(Vec3(b.pMin.x, b.pMax.y, b.pMin.z)));
    wb = Union(wb, t.Point(Vec3(b.pMin.x, b.pMin.y, b.pMax.z)));
    wb = Union(wb, t.Point(Vec3(b.pMin.x, b.pMax.y, b.pMax.z)));
    wb = Union(wb, t.Point(Vec3(b.pMax.x, b.pMax.y, b.pMin.z)));
    wb = Union(wb, t.Point(Vec3(b.pMax.x, b.pMin.y, b.pMax.z)));
    wb = Union(wb, t.Point(Vec3(b.pMax.x, b.pMax.y, b.pMax.z)));
    return wb;
}
// the WorldBound() of a list of primitives = what their BVHAccel's root node holds: the union of the triangles' / spheres' bounds
static Bounds3 PrimitiveListBound(const std::vector<GeometricPrimitive> &prims, size_t begin, size_t end) {
    Bounds3 b;
    for (size_t i = begin; i < end; ++i) {
        const TriangleMesh &m = *prims[i].shape;
        for (int idx : m.indices) b = Union(b, m.p[idx]);
        if (prims[i].sphere) b = Union(b, prims[i].sphere->WorldBound());
        if (prims[i].instance) b = Union(b, prims[i].instance->worldBound);
    }
    return b;
}

void pbrtObjectInstance(const std::string &name) {
    VERIFY_WORLD("ObjectInstance");
    if (renderOptions->currentInstance) { Error("ObjectInstance can't be called inside instance definition"); return; }
    auto it = renderOptions->instances.find(name);
    if (it == renderOptions->instances.end()) { Error("Unable to find instance named \"%s\"", name.c_str()); return; }
    if (it->second.empty()) return;
    const Transform &i2w = curTransform[0];
    if (curTransform.IsAnimated()) Unsupported("ObjectInstance \"%s\" under an animated transformation (primitive.cpp:76-96) has no counterpart on this path", name.c_str());
    if (g_twoLevelInstancing) {
        // The reference's structure (api.cpp:1555-1591): one BVHAccel per object (built at its first instantiation), one
        // TransformedPrimitive per ObjectInstance holding the CTM; bounds = PrimitiveToWorld(primitive->WorldBound()) (transform.cpp:141-153)
        int oi;
        auto known = renderOptions->objectIndex.find(name);
        if (known == renderOptions->objectIndex.end()) {
            Scene::ObjectDef od;
            od.prims = it->second;
            od.accel = CreateBVHAccelerator(od.prims, renderOptions->AcceleratorParams);
            oi = (int)renderOptions->objectDefs.size();
            renderOptions->objectDefs.push_back(std::move(od));
            renderOptions->objectIndex[name] = oi;
        } else oi = known->second;
        const Bounds3 b = renderOptions->objectDefs[oi].accel->WorldBound();
        auto inst = std::make_shared<InstanceRef>();
        inst->object = oi;
        inst->i2w = i2w;
        Bounds3 wb = TransformBounds(i2w, b);
        inst->worldBound = wb;
        GeometricPrimitive gp;
        gp.shape = std::make_shared<TriangleMesh>();   // empty: the primitive is the instance
        gp.instance = inst;
        renderOptions->primitives.push_back(gp);
        return;
    }
    // Default: static instances are flattened -- a transformed copy of each mesh (documented deviation):
    // same surfaces, hits equal within float tolerance, no second BVH level on the device.  What the reference derives from
    // scene.WorldBound() (the voxel grid of the spatial light distribution, the scene radius of distant / infinite lights) sees the
    // TransformedPrimitive's bound there -- the box of the object's transformed box, looser than the geometry -- so that bound is kept too.
    renderOptions->instanceBound = Union(renderOptions->instanceBound, TransformBounds(i2w, PrimitiveListBound(it->second, 0, it->second.size())));
    renderOptions->haveFlattenedInstances = true;
    const size_t firstFlattened = renderOptions->primitives.size();
    for (const GeometricPrimitive &src : it->second) {
        GeometricPrimitive gp = src;
        auto mesh = std::make_shared<TriangleMesh>(*src.shape);
        for (auto &p : mesh->p) p = i2w.Point(p);
        for (auto &n : mesh->n) n = i2w.Normal(n);
        for (auto &s : mesh->s) s = i2w.Vector(s);
        // A mirroring instance transform: the reference computes the interaction in the object's space and carries the normals over with
        // the inverse transpose, which does not flip them.  Here the geometric normal comes from the cross product of WORLD-space
        // edges, which does flip: compensate through the orientation flag -- but only for meshes without shading normals; with
        // them the flag only enters SetShadingGeometry's flip of the (correctly transformed) shading normal, n follows by Faceforward.
        if (i2w.SwapsHandedness() && mesh->n.empty()) mesh->transformSwapsHandedness = !mesh->transformSwapsHandedness;
        gp.shape = mesh;
        if (src.sphere) {   // the instance transform goes on top of the sphere's own
            auto sp = std::make_shared<SphereShape>(*src.sphere);
            sp->o2w = i2w * src.sphere->o2w;
            sp->w2o = Transform(sp->o2w.mInv, sp->o2w.m);
            // transformSwapsHandedness stays the one of the sphere's OWN ObjectToWorld: its normal is built in the sphere's object space
            // and transformed by inverse transposes (sphere.cpp:149, transform.cpp:262-297), an instance transform never flips it
            gp.sphere = sp;
        }
        renderOptions->primitives.push_back(gp);
    }
    for (size_t i = firstFlattened; i < renderOptions->primitives.size(); ++i) renderOptions->flattenedPrim.resize(i + 1, 0), renderOptions->flattenedPrim[i] = 1;
}

// ------------------------------------------------------------------ WorldEnd
PerspectiveCamera::PerspectiveCamera(const Transform &c2w, const Float sw[4], Float so, Float sc, Float lensr,
                                     Float focald, Float fov, Film *f)
    : CameraToWorld(c2w), shutterOpen(so), shutterClose(sc), lensRadius(lensr), focalDistance(focald), film(f) {
    CameraToScreen = Perspective(fov, 1e-2f, 1000.f);   // perspective.cpp:52
    // camera.h:101-107 ; sw = {pMin.x, pMax.x, pMin.y, pMax.y}
    ScreenToRaster = Scale(film->fullResolution[0], film->fullResolution[1], 1) *
                     Scale(1 / (sw[1] - sw[0]), 1 / (sw[2] - sw[3]), 1) * Translate(Vec3(-sw[0], -sw[3], 0));
    RasterToScreen = Inverse(ScreenToRaster);
    RasterToCamera = Inverse(CameraToScreen) * RasterToScreen;
    dxCamera = RasterToCamera.Point(Vec3(1, 0, 0)) - RasterToCamera.Point(Vec3(0, 0, 0));   // perspective.cpp:56-59
    dyCamera = RasterToCamera.Point(Vec3(0, 1, 0)) - RasterToCamera.Point(Vec3(0, 0, 0));
}

PerspectiveCamera *CreatePerspectiveCamera(const ParamSet &ps, const Transform &cam2world, Film *film) {   // perspective.cpp:228-277
    Float shutteropen = ps.FindOneFloat("shutteropen", 0.f), shutterclose = ps.FindOneFloat("shutterclose", 1.f);
    if (shutterclose < shutteropen) {
        Warning("Shutter close time [%f] < shutter open [%f].  Swapping them.", shutterclose, shutteropen);
        std::swap(shutterclose, shutteropen);
    }
    Float lensradius = ps.FindOneFloat("lensradius", 0.f), focaldistance = ps.FindOneFloat("focaldistance", 1e6);
    Float frame = ps.FindOneFloat("frameaspectratio", Float(film->fullResolution[0]) / Float(film->fullResolution[1]));
    Float screen[4];
    if (frame > 1.f) { screen[0] = -frame; screen[1] = frame; screen[2] = -1.f; screen[3] = 1.f; }
    else { screen[0] = -1.f; screen[1] = 1.f; screen[2] = -1.f / frame; screen[3] = 1.f / frame; }
    int swi;
    const Float *sw = ps.FindFloat("screenwindow", &swi);
    if (sw) {
        if (swi == 4) { screen[0] = sw[0]; screen[1] = sw[1]; screen[2] = sw[2]; screen[3] = sw[3]; }
        else Error("\"screenwindow\" should have four values");
    }
    Float fov = ps.FindOneFloat("fov", 90.);
    Float halffov = ps.FindOneFloat("halffov", -1.f);
    if (halffov > 0.f) fov = 2.f * halffov;
    return new PerspectiveCamera(cam2world, screen, shutteropen, shutterclose, lensradius, focaldistance, fov, film);
}

SobolSampler::SobolSampler(int64_t spp, const int smin[2], const int smax[2]) {   // sobol.h:51-62
    kind = Sobol;
    samplesPerPixel = spp <= 1 ? 1 : (int64_t)RoundUpPow2((int32_t)spp);
    if (!IsPowerOf2(spp))
        Warning("Non power-of-two sample count rounded up to %lld for SobolSampler.", (long long)samplesPerPixel);
    for (int i = 0; i < 2; ++i) { sampleMin[i] = smin[i]; sampleMax[i] = smax[i]; }
    resolution = RoundUpPow2(std::max(smax[0] - smin[0], smax[1] - smin[1]));
    log2Resolution = Log2Int((uint32_t)resolution);
}

// HaltonSampler::HaltonSampler samplers/halton.cpp:71-97 (the digit permutations are the device library's business)
static void extendedGCD(uint64_t a, uint64_t b, int64_t *x, int64_t *y) {   // halton.cpp:55-66
    if (b == 0) { *x = 1; *y = 0; return; }
    int64_t d = (int64_t)(a / b), xp, yp;
    extendedGCD(b, a % b, &xp, &yp);
    *x = yp;
    *y = xp - (d * yp);
}
static uint64_t multiplicativeInverse(int64_t a, int64_t n) {   // halton.cpp:49-53
    int64_t x, y;
    extendedGCD((uint64_t)a, (uint64_t)n, &x, &y);
    int64_t r = x - (x / n) * n;   // Mod(x, n) core/pbrt.h:310-313
    return (uint64_t)(r < 0 ? r + n : r);
}
TileSerialSampler::TileSerialSampler(Kind k, const ParamSet &ps, const int smin[2], const int smax[2]) {
    kind = k;
    for (int i = 0; i < 2; ++i) { sampleMin[i] = smin[i]; sampleMax[i] = smax[i]; }
    if (k == Random) {   // CreateRandomSampler random.cpp:75-78
        samplesPerPixel = ps.FindOneInt("pixelsamples", 4);
        nSampledDimensions = 0;
    } else if (k == Stratified) {   // CreateStratifiedSampler stratified.cpp:78-86
        jitterSamples = ps.FindOneBool("jitter", true);
        xPixelSamples = ps.FindOneInt("xsamples", 4);
        yPixelSamples = ps.FindOneInt("ysamples", 4);
        nSampledDimensions = ps.FindOneInt("dimensions", 4);
        if (PbrtOptions.quickRender) xPixelSamples = yPixelSamples = 1;
        samplesPerPixel = (int64_t)xPixelSamples * yPixelSamples;
    } else {   // CreateZeroTwoSequenceSampler zerotwosequence.cpp:76-81, constructor :43-51
        int nsamp = ps.FindOneInt("pixelsamples", 16);
        nSampledDimensions = ps.FindOneInt("dimensions", 4);
        if (PbrtOptions.quickRender) nsamp = 1;
        samplesPerPixel = nsamp <= 1 ? 1 : (int64_t)RoundUpPow2((int32_t)nsamp);
        if (!IsPowerOf2(nsamp))
            Warning("Pixel samples being rounded up to power of 2 (from %d to %lld).", nsamp, (long long)samplesPerPixel);
    }
    if (samplesPerPixel < 1 || nSampledDimensions < 0) Error("Sampler: pixel samples / dimensions out of range");
}

HaltonSampler::HaltonSampler(int64_t spp, const int smin[2], const int smax[2], bool atCenter) {
    kind = Halton;
    samplesPerPixel = spp;
    sampleAtPixelCenter = atCenter;
    const int kMaxResolution = 128;
    for (int i = 0; i < 2; ++i) {
        sampleMin[i] = smin[i]; sampleMax[i] = smax[i];
        int res = smax[i] - smin[i];
        int base = (i == 0) ? 2 : 3;
        int scale = 1, exp = 0;
        while (scale < std::min(res, kMaxResolution)) { scale *= base; ++exp; }
        baseScales[i] = scale;
        baseExponents[i] = exp;
    }
    sampleStride = baseScales[0] * baseScales[1];
    multInverse[0] = (int)multiplicativeInverse(baseScales[1], baseScales[0]);
    multInverse[1] = (int)multiplicativeInverse(baseScales[0], baseScales[1]);
}

Scene::Scene(std::shared_ptr<BVHAccel> agg, std::vector<GeometricPrimitive> prims, std::vector<LightEntry> l)
    : aggregate(std::move(agg)), primitives(std::move(prims)), lights(std::move(l)) {
    worldBound = aggregate->WorldBound();   // scene.h:56
}

void pbrtWorldEnd() {
    VERIFY_WORLD("WorldEnd");
    while (pushedGraphicsStates.size()) {
        Warning("Missing end to pbrtAttributeBegin()");
        pushedGraphicsStates.pop_back();
        pushedTransforms.pop_back();
    }
    while (pushedTransforms.size()) { Warning("Missing end to pbrtTransformBegin()"); pushedTransforms.pop_back(); }

    // RenderOptions::MakeIntegrator (api.cpp:1666-1718) / MakeCamera (:1720-1731)
    std::unique_ptr<WavefrontPathIntegrator> integrator;
    {
        std::unique_ptr<Filter> filter = MakeFilter(renderOptions->FilterName, renderOptions->FilterParams);
        Film *film = nullptr;
        if (filter) {
            if (renderOptions->FilmName == "image") {
                film = CreateFilm(renderOptions->FilmParams, std::move(filter));
                renderOptions->FilmParams.ReportUnused();
            } else
                Warning("Film \"%s\" unknown.", renderOptions->FilmName.c_str());
        }
        std::shared_ptr<PerspectiveCamera> camera;
        if (!film) Error("Unable to create film.");
        else if (renderOptions->CameraName == "perspective") {
            if (renderOptions->CameraToWorld.IsAnimated())
                Unsupported("an animated camera transformation (camera.h:68, AnimatedTransform CameraToWorld) has no counterpart on this path");
            camera.reset(CreatePerspectiveCamera(renderOptions->CameraParams, renderOptions->CameraToWorld[0], film));
            renderOptions->CameraParams.ReportUnused();
        } else {
            if (renderOptions->CameraName == "orthographic" || renderOptions->CameraName == "realistic" || renderOptions->CameraName == "environment")
                Unsupported("Camera \"%s\" has no counterpart on this path (perspective only: SURVEY.md s.2 row 26)", renderOptions->CameraName.c_str());
            else Warning("Camera \"%s\" unknown.", renderOptions->CameraName.c_str());   // api.cpp:862 in the reference
            delete film;
        }
        if (!camera) Error("Unable to create camera");
        else {
            int smin[2], smax[2];
            camera->film->GetSampleBounds(smin, smax);
            int nsamp = renderOptions->SamplerParams.FindOneInt("pixelsamples", 16);
            if (PbrtOptions.quickRender) nsamp = 1;
            // MakeSampler (api.cpp:815-840): "halton" (pbrt's default) and "sobol" run on the GPU path
            std::shared_ptr<Sampler> sampler;
            const char *fastEnv = std::getenv("PBRT_AMD_FAST_SAMPLERS");
            const bool tileSerialName = renderOptions->SamplerName == "random" || renderOptions->SamplerName == "stratified" || renderOptions->SamplerName == "02sequence" ||
                                        renderOptions->SamplerName == "lowdiscrepancy";
            if (renderOptions->SamplerName == "halton") {
                bool atCenter = renderOptions->SamplerParams.FindOneBool("samplepixelcenter", false);
                sampler = std::make_shared<HaltonSampler>(nsamp, smin, smax, atCenter);
            } else if (tileSerialName && (PbrtOptions.fastSamplers || (fastEnv && fastEnv[0] == '1'))) {
                // the user's choice (--fast-samplers): an unbiased image of the same scene at wavefront speed, not the reference's pixel values
                // the sample count the reference's sampler of that name would use (random.cpp:60-63: 4; stratified.cpp:72-80: xsamples * ysamples, 4 x 4;
                // zerotwosequence.cpp:41-50, 116-121: "pixelsamples" 16 rounded up to a power of two)
                if (renderOptions->SamplerName == "stratified")
                    nsamp = renderOptions->SamplerParams.FindOneInt("xsamples", 4) * renderOptions->SamplerParams.FindOneInt("ysamples", 4);
                else if (renderOptions->SamplerName == "random")
                    nsamp = renderOptions->SamplerParams.FindOneInt("pixelsamples", 4);
                else {
                    int64_t p2 = 1;
                    while (p2 < (int64_t)std::max(1, nsamp)) p2 <<= 1;
                    nsamp = (int)p2;
                }
                if (PbrtOptions.quickRender) nsamp = 1;
                Warning("--fast-samplers: Sampler \"%s\" renders with \"sobol\" at %d spp (not the reference's image; without the flag the tile-serial rounds reproduce it).",
                        renderOptions->SamplerName.c_str(), nsamp);
                sampler = std::make_shared<SobolSampler>(nsamp, smin, smax);
            } else if (tileSerialName) {
                // one PCG32 stream per tile: the device walks the tiles' samples in the reference's order (ABI v11, "tile-serial" -- the reference's image, slowly)
                Sampler::Kind k = renderOptions->SamplerName == "random" ? Sampler::Random : (renderOptions->SamplerName == "stratified" ? Sampler::Stratified : Sampler::ZeroTwo);
                sampler = std::make_shared<TileSerialSampler>(k, renderOptions->SamplerParams, smin, smax);
            } else {
                if (renderOptions->SamplerName != "sobol")   // "maxmindist" (ABI v12 takes its generator matrix from the caller: the reference-side binding has the reference's CMaxMinDist table, this host has not) and unknown names
                    Warning("Sampler \"%s\" is not implemented on the GPU path (\"sobol\", \"halton\", \"random\", \"stratified\" and \"02sequence\" are); rendering with \"sobol\" at %d spp.",
                            renderOptions->SamplerName.c_str(), nsamp);
                sampler = std::make_shared<SobolSampler>(nsamp, smin, smax);
            }
            renderOptions->SamplerParams.ReportUnused();
            if (renderOptions->IntegratorName == "path" || renderOptions->IntegratorName == "volpath") {
                // CreateVolPathIntegrator (volpath.cpp:192-214) reads the parameters CreatePathIntegrator does
                integrator.reset(CreatePathIntegrator(renderOptions->IntegratorParams, sampler, camera));
                integrator->volPath = renderOptions->IntegratorName == "volpath";
                integrator->nGpus = PbrtOptions.nGpus;
                renderOptions->IntegratorParams.ReportUnused();
            } else
            {
                static const char *const known[] = {"whitted", "directlighting", "bdpt", "mlt", "ambientocclusion", "sppm"};
                bool isKnown = false;
                for (const char *k : known) isKnown |= renderOptions->IntegratorName == k;
                if (isKnown) Unsupported("Integrator \"%s\" has no counterpart on this path (\"path\" and \"volpath\": SURVEY.md s.2 row 7)", renderOptions->IntegratorName.c_str());
                else Error("Integrator \"%s\" unknown.", renderOptions->IntegratorName.c_str());   // api.cpp:1710 in the reference: no render either
            }
            if (renderOptions->lights.empty()) Warning("No light sources defined in scene; rendering a black image.");
        }
    }
    // RenderOptions::MakeScene (api.cpp:1655-1664)
    std::unique_ptr<Scene> scene;
    {
        std::shared_ptr<BVHAccel> accel;
        if (renderOptions->AcceleratorName == "bvh") accel = CreateBVHAccelerator(renderOptions->primitives, renderOptions->AcceleratorParams);
        else {
            // (the reference: "kdtree" builds a KdTreeAccel, any other name warns and falls back to "bvh", api.cpp:781-793)
            if (renderOptions->AcceleratorName == "kdtree")
                Unsupported("Accelerator \"kdtree\" has no counterpart on this path (another traversal order: other hits at equal distances); name \"bvh\"");
            else Warning("Accelerator \"%s\" unknown.", renderOptions->AcceleratorName.c_str());
            accel = std::make_shared<BVHAccel>(renderOptions->primitives, 4, BVHAccel::SplitMethod::SAH);
        }
        renderOptions->AcceleratorParams.ReportUnused();
        scene.reset(new Scene(accel, std::move(renderOptions->primitives), std::move(renderOptions->lights)));
        if (renderOptions->haveFlattenedInstances) {   // scene.WorldBound() as the reference's top-level BVH has it (see pbrtObjectInstance)
            Bounds3 wb = renderOptions->instanceBound;
            renderOptions->flattenedPrim.resize(scene->primitives.size(), 0);
            for (size_t i = 0; i < scene->primitives.size(); ++i)
                if (!renderOptions->flattenedPrim[i]) wb = Union(wb, PrimitiveListBound(scene->primitives, i, i + 1));
            scene->worldBound = wb;
        }
        scene->textures = CurrentTextures();
        scene->media = renderOptions->media;
        { int in; CreateMediumInterface(&in, &scene->cameraMedium); }   // MakeCamera at WorldEnd: mediumInterface.outside (api.cpp:793-813)
        scene->objects = std::move(renderOptions->objectDefs);
    }
    if (g_unsupportedCount > 0) {
        Error("%d unsupported parameter(s) above would change the image: not rendering", g_unsupportedCount);
        scene.reset(); integrator.reset();
        g_renderFailed = true;    // sticky: the process exit code (main.cpp)
        g_unsupportedCount = 0;   // per world block: a later, fully supported WorldBegin ... WorldEnd (or scene file) still renders
    }
    if (scene && integrator) {
        if (PbrtOptions.deferRender) {
            builtScene.reset(new BuiltScene{std::move(scene), std::move(integrator)});
        } else
            integrator->Render(*scene);   // api.cpp:1623 -- the drop-in call site
    }
    graphicsState = GraphicsState();
    currentApiState = APIState::OptionsBlock;
    renderOptions.reset(new RenderOptions);
    for (int i = 0; i < MaxTransforms; ++i) curTransform[i] = Transform();
    activeTransformBits = AllTransformsBits;
    namedCoordinateSystems.clear();
}

std::unique_ptr<BuiltScene> pbrtTakeBuiltScene() { return std::move(builtScene); }

}  // namespace pbrt_amd
