// WavefrontPathIntegrator: the Integrator subclass that stands where PathIntegrator does
// (reference integrators/path.{h,cpp}; created by CreatePathIntegrator path.cpp:190-213).
// Render(scene) = flatten the Scene to the POD mi_scene_desc, hand it to the HIP library through
// the C ABI of include/pbrt_amd.h, pull the FilmTilePixel array back and let the host Film do
// MergeFilmTile / WriteImage (film.cpp:117-130,168-210).  The HIP library is loaded at run time;
// there is NO CPU fallback: if it (or a GPU) is missing, Render reports an Error and returns.
#include <dlfcn.h>

#include <chrono>
#include <functional>

#include "api.h"

namespace pbrt_amd {

WavefrontPathIntegrator::WavefrontPathIntegrator(int maxDepth, std::shared_ptr<PerspectiveCamera> camera,
                                                 std::shared_ptr<Sampler> sampler, const int pmin[2],
                                                 const int pmax[2], Float rrThreshold, const std::string &strategy)
    : maxDepth(maxDepth), camera(std::move(camera)), sampler(std::move(sampler)), rrThreshold(rrThreshold),
      lightSampleStrategy(strategy) {
    for (int i = 0; i < 2; ++i) { pixelMin[i] = pmin[i]; pixelMax[i] = pmax[i]; }
}

WavefrontPathIntegrator *CreatePathIntegrator(const ParamSet &ps, std::shared_ptr<Sampler> sampler,
                                              std::shared_ptr<PerspectiveCamera> camera) {   // path.cpp:190-213
    int maxDepth = ps.FindOneInt("maxdepth", 5);
    int np;
    const int *pb = ps.FindInt("pixelbounds", &np);
    int pmin[2], pmax[2];
    camera->film->GetSampleBounds(pmin, pmax);
    if (pb) {
        if (np != 4) Error("Expected four values for \"pixelbounds\" parameter. Got %d.", np);
        else {
            pmin[0] = std::max(pmin[0], pb[0]); pmax[0] = std::min(pmax[0], pb[1]);
            pmin[1] = std::max(pmin[1], pb[2]); pmax[1] = std::min(pmax[1], pb[3]);
            if ((pmax[0] - pmin[0]) * (pmax[1] - pmin[1]) == 0) Error("Degenerate \"pixelbounds\" specified.");
        }
    }
    Float rrThreshold = ps.FindOneFloat("rrthreshold", 1.);
    std::string lightStrategy = ps.FindOneString("lightsamplestrategy", "spatial");
    return new WavefrontPathIntegrator(maxDepth, camera, sampler, pmin, pmax, rrThreshold, lightStrategy);
}

static void copyMatrix(float dst[16], const Matrix4x4 &m) { std::memcpy(dst, m.m, 16 * sizeof(float)); }

std::unique_ptr<FlatScene> WavefrontPathIntegrator::Flatten(const Scene &scene) const {
    std::unique_ptr<FlatScene> fs(new FlatScene);
    const BVHAccel &bvh = *scene.aggregate;
    // every GeometricPrimitive that owns geometry: the scene's own, then (two-level instancing only) those of each instanced object
    std::vector<GeometricPrimitive> prims = scene.primitives;
    std::vector<uint32_t> objectPrimBase(scene.objects.size());
    for (size_t o = 0; o < scene.objects.size(); ++o) {
        objectPrimBase[o] = (uint32_t)prims.size();
        prims.insert(prims.end(), scene.objects[o].prims.begin(), scene.objects[o].prims.end());
    }
    // --- vertices + meshes
    std::vector<uint32_t> vertexOffset(prims.size()), triOffset(prims.size());
    size_t nv = 0, nt = 0;
    bool anyN = false, anyUV = false;
    for (size_t i = 0; i < prims.size(); ++i) {
        vertexOffset[i] = (uint32_t)nv; triOffset[i] = (uint32_t)nt;
        nv += prims[i].shape->p.size(); nt += prims[i].shape->nTriangles() + (prims[i].sphere ? 1 : 0) + (prims[i].instance ? 1 : 0);
        anyN |= !prims[i].shape->n.empty(); anyUV |= !prims[i].shape->uv.empty();
    }
    fs->P.resize(3 * nv);
    if (anyN) fs->N.assign(3 * nv, 0.f);
    if (anyUV) fs->UV.assign(2 * nv, 0.f);
    // materials: identical BxDF lists (constant materials) / identical parameter nodes (textured ones) share one slot,
    // the sort key of the shading kernels; a textured mix refers to its two sub-materials by slot
    std::map<const Material *, int> bssrdfSlot;
    std::function<int(const std::shared_ptr<Material> &)> materialIndex = [&](const std::shared_ptr<Material> &m) -> int {
        if (!m) return -1;
        mi_material_desc md = m->desc;
        if (md.type == MI_MAT_MIX && md.textured) { md.m1 = materialIndex(m->m1); md.m2 = materialIndex(m->m2); }
        if (m->bssrdf.kind != MI_BSSRDF_NONE) {   // one slot per Material OBJECT: Sample_Sp compares material pointers (bssrdf.cpp:302)
            auto it = bssrdfSlot.find(m.get());
            if (it != bssrdfSlot.end()) return it->second;
            mi_bssrdf_desc b = m->bssrdf;
            size_t t = 0;
            while (t < fs->tableKeep.size() && fs->tableKeep[t] != m->table) ++t;
            if (t == fs->tableKeep.size()) fs->tableKeep.push_back(m->table);
            b.table = (int32_t)t;
            fs->materials.push_back(m->bsdf);
            fs->materialDescs.push_back(md);
            fs->materialBssrdf.resize(fs->materials.size());
            fs->materialBssrdf.back() = b;
            return bssrdfSlot[m.get()] = (int)fs->materials.size() - 1;
        }
        for (size_t k = 0; k < fs->materials.size(); ++k) {
            if (k < fs->materialBssrdf.size() && fs->materialBssrdf[k].kind != MI_BSSRDF_NONE) continue;
            if (md.textured != fs->materialDescs[k].textured) continue;
            if (md.textured ? std::memcmp(&fs->materialDescs[k], &md, sizeof(md)) == 0
                            : std::memcmp(&fs->materials[k], &m->bsdf, sizeof(mi_material)) == 0) return (int)k;
        }
        fs->materials.push_back(m->bsdf);
        fs->materialDescs.push_back(md);
        return (int)fs->materials.size() - 1;
    };
    fs->meshes.resize(prims.size());
    for (size_t i = 0; i < prims.size(); ++i) {
        const TriangleMesh &mesh = *prims[i].shape;
        for (size_t v = 0; v < mesh.p.size(); ++v) {
            size_t g = vertexOffset[i] + v;
            fs->P[3 * g] = mesh.p[v].x; fs->P[3 * g + 1] = mesh.p[v].y; fs->P[3 * g + 2] = mesh.p[v].z;
            if (!mesh.n.empty()) { fs->N[3 * g] = mesh.n[v].x; fs->N[3 * g + 1] = mesh.n[v].y; fs->N[3 * g + 2] = mesh.n[v].z; }
            if (!mesh.uv.empty()) { fs->UV[2 * g] = mesh.uv[2 * v]; fs->UV[2 * g + 1] = mesh.uv[2 * v + 1]; }
        }
        uint32_t flags = 0;
        if (!mesh.n.empty()) flags |= MI_MESH_HAS_N;
        if (!mesh.uv.empty()) flags |= MI_MESH_HAS_UV;
        if (!mesh.s.empty()) Warning("per-vertex tangents \"S\" are ignored by this path");
        if (mesh.reverseOrientation ^ mesh.transformSwapsHandedness) flags |= MI_MESH_FLIP;
        fs->meshes[i].flags = flags;
        fs->meshes[i].material = materialIndex(prims[i].material);
        fs->meshAlpha.push_back(mesh.alphaTex); fs->meshAlpha.push_back(mesh.shadowAlphaTex);
        fs->meshMedium.push_back(prims[i].mediumInside); fs->meshMedium.push_back(prims[i].mediumOutside);
    }
    // --- triangles in BVH primitive order: the top-level BVH's, then (two-level instancing) each object's own BVH order
    size_t nTop = bvh.primitives.size(), nTris = nTop;
    for (const auto &od : scene.objects) nTris += od.accel->primitives.size();
    fs->triIndices.resize(3 * nTris);
    fs->triMesh.resize(nTris);
    fs->triLight.assign(nTris, -1);
    std::vector<uint32_t> orderOf(nt);   // (prim, tri) -> position in the ordered primitive arrays
    auto emit = [&](const BVHAccel &acc, uint32_t primBase, size_t outBase) {
        for (size_t j = 0; j < acc.primitives.size(); ++j) {
            const size_t k = outBase + j;
            BVHAccel::PrimRef r = acc.primitives[j];
            r.prim += primBase;
            const TriangleMesh &mesh = *prims[r.prim].shape;
            if (prims[r.prim].instance) {   // MI_PRIM_INSTANCE, index into instances[]
                const InstanceRef &in = *prims[r.prim].instance;
                mi_instance mi;
                std::memset(&mi, 0, sizeof(mi));
                copyMatrix(mi.i2w, in.i2w.m); copyMatrix(mi.w2i, in.i2w.mInv);
                mi.object = (uint32_t)in.object;
                fs->triIndices[3 * k] = MI_PRIM_INSTANCE; fs->triIndices[3 * k + 1] = (uint32_t)fs->instances.size(); fs->triIndices[3 * k + 2] = 0;
                fs->instances.push_back(mi);
            } else if (prims[r.prim].sphere) {   // MI_PRIM_SPHERE, index into spheres[]
                const SphereShape &sp = *prims[r.prim].sphere;
                mi_sphere ms;
                std::memset(&ms, 0, sizeof(ms));
                copyMatrix(ms.o2w, sp.o2w.m); copyMatrix(ms.w2o, sp.w2o.m);
                ms.radius = sp.radius; ms.zmin = sp.zMin; ms.zmax = sp.zMax; ms.theta_min = sp.thetaMin; ms.theta_max = sp.thetaMax; ms.phi_max = sp.phiMax;
                ms.flags = (sp.reverseOrientation ? 1u : 0u) | (sp.transformSwapsHandedness ? 2u : 0u);
                ms.area = sp.Area();
                fs->triIndices[3 * k] = MI_PRIM_SPHERE; fs->triIndices[3 * k + 1] = (uint32_t)fs->spheres.size(); fs->triIndices[3 * k + 2] = 0;
                fs->spheres.push_back(ms);
            } else
                for (int c = 0; c < 3; ++c) fs->triIndices[3 * k + c] = vertexOffset[r.prim] + (uint32_t)mesh.indices[3 * r.tri + c];
            fs->triMesh[k] = r.prim;
            orderOf[triOffset[r.prim] + r.tri] = (uint32_t)k;
        }
    };
    emit(bvh, 0, 0);
    fs->nodes = bvh.nodes;
    if (!fs->nodes.empty()) {
        // scene.WorldBound() is read off the root node by the consumers of the desc (voxel grid of the spatial light distribution).  With
        // flattened instances the reference's bound is looser than the geometry (api.cpp pbrtObjectInstance): widen the root box to it --
        // a looser root only lets a few more rays start the traversal, the hits are the same.
        const Bounds3 &wb = scene.WorldBound();
        for (int a = 0; a < 3; ++a) { fs->nodes[0].bmin[a] = std::min(fs->nodes[0].bmin[a], wb.pMin[a]); fs->nodes[0].bmax[a] = std::max(fs->nodes[0].bmax[a], wb.pMax[a]); }
    }
    {
        size_t outBase = nTop;
        for (size_t o = 0; o < scene.objects.size(); ++o) {
            const BVHAccel &acc = *scene.objects[o].accel;
            emit(acc, objectPrimBase[o], outBase);
            mi_object mo = {(uint32_t)outBase, (uint32_t)acc.primitives.size(), (uint32_t)fs->nodes.size(), (uint32_t)acc.nodes.size()};
            fs->objects.push_back(mo);
            fs->nodes.insert(fs->nodes.end(), acc.nodes.begin(), acc.nodes.end());
            outBase += acc.primitives.size();
        }
    }
    // --- lights, in scene.lights order (api.cpp:1418-1424: appended in file order, one per emissive triangle)
    Vec3 worldCenter = (scene.WorldBound().pMin + scene.WorldBound().pMax) / 2;   // Bounds3::BoundingSphere geometry.h:808-811
    Float worldRadius = 0;
    {
        const Bounds3 &b = scene.WorldBound();
        bool inside = worldCenter.x >= b.pMin.x && worldCenter.x <= b.pMax.x && worldCenter.y >= b.pMin.y &&
                      worldCenter.y <= b.pMax.y && worldCenter.z >= b.pMin.z && worldCenter.z <= b.pMax.z;
        worldRadius = inside ? (worldCenter - b.pMax).Length() : 0;
    }
    std::vector<Float> power;
    for (const LightEntry &e : scene.lights) {
        if (e.light) {
            mi_light l = e.light->l;
            l.world_radius = worldRadius;
            for (int i = 0; i < 3; ++i) l.world_center[i] = worldCenter[i];
            RGB L(l.L[0], l.L[1], l.L[2]);
            if (l.type == MI_LIGHT_POINT) power.push_back((L * (4 * kPi)).y());                       // point.cpp:56
            else if (l.type == MI_LIGHT_SPOT) power.push_back((L * 2 * kPi * (1 - .5f * (l.cos_falloff_start + l.cos_total_width))).y());   // spot.cpp:75-77
            else if (l.type == MI_LIGHT_INFINITE) {   // infinite.cpp:86-90: Pi r^2 * Lmap->Lookup((.5,.5), .5)
                RGB Lc = e.light->env ? e.light->env->powerLookup : L;
                power.push_back((Lc * (kPi * worldRadius * worldRadius)).y());
                if (e.light->env) {
                    const EnvMap &em = *e.light->env;
                    mi_envmap me;
                    std::memset(&me, 0, sizeof(me));
                    me.width = em.width; me.height = em.height; me.rgb = em.rgb.data();
                    me.cond_func = em.condFunc.data(); me.cond_cdf = em.condCdf.data(); me.cond_func_int = em.condFuncInt.data();
                    me.marg_func = em.margFunc.data(); me.marg_cdf = em.margCdf.data(); me.marg_func_int = em.margFuncInt;
                    fs->envmaps.push_back(me);
                    fs->envKeep.push_back(e.light->env);
                    l.env_map = (int32_t)fs->envmaps.size();
                }
            } else power.push_back((L * kPi * worldRadius * worldRadius).y());                        // distant.cpp:64-66
            fs->lights.push_back(l);
        } else {
            const GeometricPrimitive &gp = prims[e.prim];
            const TriangleMesh &mesh = *gp.shape;
            if (gp.sphere) {   // DiffuseAreaLight on the sphere (api.cpp:1357-1366, one light per Shape)
                mi_light l;
                std::memset(&l, 0, sizeof(l));
                l.type = MI_LIGHT_AREA_SPHERE;
                l.tri = (int32_t)orderOf[triOffset[e.prim]];
                l.sphere = (int32_t)fs->triIndices[3 * (size_t)l.tri + 1];
                l.two_sided = gp.areaLight->twoSided;
                for (int i = 0; i < 3; ++i) l.L[i] = gp.areaLight->Lemit.c[i];
                l.area = gp.sphere->Area();
                fs->triLight[l.tri] = (int32_t)fs->lights.size();
                RGB pw = gp.areaLight->Lemit * (Float)(l.two_sided ? 2 : 1) * l.area * kPi;   // diffuse.cpp:64-66
                power.push_back(pw.y());
                fs->lights.push_back(l);
            }
            for (int t = 0; t < mesh.nTriangles(); ++t) {
                mi_light l;
                std::memset(&l, 0, sizeof(l));
                l.type = MI_LIGHT_AREA_TRI;
                l.tri = (int32_t)orderOf[triOffset[e.prim] + t];
                l.two_sided = gp.areaLight->twoSided;
                for (int i = 0; i < 3; ++i) l.L[i] = gp.areaLight->Lemit.c[i];
                const Vec3 &p0 = mesh.p[mesh.indices[3 * t]], &p1 = mesh.p[mesh.indices[3 * t + 1]], &p2 = mesh.p[mesh.indices[3 * t + 2]];
                l.area = 0.5 * Cross(p1 - p0, p2 - p0).Length();   // Triangle::Area triangle.cpp:575-581
                fs->triLight[l.tri] = (int32_t)fs->lights.size();
                RGB pw = gp.areaLight->Lemit * (Float)(l.two_sided ? 2 : 1) * l.area * kPi;   // diffuse.cpp:64-66
                power.push_back(pw.y());
                fs->lights.push_back(l);
            }
        }
    }
    // --- light-selection distribution (lightdistrib.cpp:48-84, sampling.h:55-70)
    size_t nl = fs->lights.size();
    std::string strategy = lightSampleStrategy;
    if (strategy != "uniform" && strategy != "power" && strategy != "spatial") {
        Error("Light sample distribution type \"%s\" unknown. Using \"spatial\".", strategy.c_str());
        strategy = "spatial";
    }
    // "spatial" (the reference's default) with more than one light: the device library evaluates
    // SpatialLightDistribution::ComputeDistribution per voxel itself; light_func/light_cdf then only carry the power table.
    bool spatial = strategy == "spatial" && nl > 1;
    bool uniform = strategy == "uniform" || nl == 1;
    fs->lightFunc.resize(nl);
    fs->lightCdf.resize(nl + 1);
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
    d.n_verts = (uint32_t)nv; d.P = fs->P.data(); d.N = anyN ? fs->N.data() : nullptr; d.UV = anyUV ? fs->UV.data() : nullptr;
    d.n_tris = (uint32_t)nTris; d.tri_indices = fs->triIndices.data(); d.tri_mesh = fs->triMesh.data(); d.tri_light = fs->triLight.data();
    d.n_meshes = (uint32_t)fs->meshes.size(); d.meshes = fs->meshes.data();
    d.n_bvh_nodes = (uint32_t)bvh.nodes.size(); d.bvh_nodes = fs->nodes.data();   // top-level nodes first, objects' nodes behind them
    d.n_top_prims = (uint32_t)nTop;
    d.n_instances = (uint32_t)fs->instances.size(); d.instances = fs->instances.empty() ? nullptr : fs->instances.data();
    d.n_objects = (uint32_t)fs->objects.size(); d.objects = fs->objects.empty() ? nullptr : fs->objects.data();
    d.n_materials = (uint32_t)fs->materials.size(); d.materials = fs->materials.data();
    d.n_lights = (uint32_t)nl; d.lights = fs->lights.data();
    d.light_func = fs->lightFunc.data(); d.light_cdf = fs->lightCdf.data(); d.light_func_int = funcInt;
    d.n_spheres = (uint32_t)fs->spheres.size(); d.spheres = fs->spheres.empty() ? nullptr : fs->spheres.data();
    d.n_envmaps = (uint32_t)fs->envmaps.size(); d.envmaps = fs->envmaps.empty() ? nullptr : fs->envmaps.data();
    // textures: the parse's node table and image pyramids (kept alive by texKeep)
    bool anyAlpha = false, anyTextured = false;
    for (int32_t a : fs->meshAlpha) anyAlpha |= a >= 0;
    for (const mi_material_desc &md : fs->materialDescs) anyTextured |= md.textured != 0;
    if (scene.textures && (anyAlpha || anyTextured)) {
        fs->texKeep = scene.textures;
        fs->textures = scene.textures->nodes;
        for (const auto &im : scene.textures->images) {
            mi_image mi;
            std::memset(&mi, 0, sizeof(mi));
            mi.width = im->width; mi.height = im->height; mi.levels = im->levels; mi.channels = im->channels;
            mi.trilinear = im->trilinear; mi.wrap = im->wrap; mi.max_aniso = im->maxAniso; mi.texels = im->texels.data();
            fs->images.push_back(mi);
        }
        d.n_textures = (uint32_t)fs->textures.size(); d.textures = fs->textures.data();
        d.n_images = (uint32_t)fs->images.size(); d.images = fs->images.empty() ? nullptr : fs->images.data();
    }
    // participating media: used by "volpath" only (PathIntegrator never looks at ray.medium), handed over whenever the scene names any
    bool anyMedium = scene.cameraMedium >= 0;
    for (int32_t m : fs->meshMedium) anyMedium |= m >= 0;
    if (anyMedium) {
        fs->mediaKeep = scene.media;
        for (const auto &ms : scene.media) {
            mi_medium m = ms->m;
            m.density = ms->density.empty() ? nullptr : ms->density.data();
            fs->media.push_back(m);
        }
        d.n_media = (uint32_t)fs->media.size(); d.media = fs->media.data();
        d.mesh_medium = fs->meshMedium.data();
    }
    if (!fs->tableKeep.empty()) {   // subsurface materials
        fs->materialBssrdf.resize(fs->materials.size());   // zero-filled tail = MI_BSSRDF_NONE
        for (const auto &t : fs->tableKeep) {
            mi_bssrdf_table bt;
            bt.n_rho = t->nRho; bt.n_radius = t->nRadius;
            bt.rho_samples = t->rhoSamples.data(); bt.radius_samples = t->radiusSamples.data(); bt.profile = t->profile.data();
            bt.rho_eff = t->rhoEff.data(); bt.profile_cdf = t->profileCDF.data();
            fs->bssrdfTables.push_back(bt);
        }
        d.n_bssrdf_tables = (uint32_t)fs->bssrdfTables.size(); d.bssrdf_tables = fs->bssrdfTables.data();
        d.material_bssrdf = fs->materialBssrdf.data();
    }
    d.camera_medium = anyMedium ? scene.cameraMedium : -1;
    d.integrator_type = volPath ? MI_INTEGRATOR_VOLPATH : MI_INTEGRATOR_PATH;
    d.material_descs = anyTextured ? fs->materialDescs.data() : nullptr;
    d.mesh_alpha = anyAlpha ? fs->meshAlpha.data() : nullptr;
    d.integrator.light_strategy = spatial ? MI_LIGHT_STRATEGY_SPATIAL : MI_LIGHT_STRATEGY_TABLE;
    d.integrator.spatial_max_voxels = 64;
    copyMatrix(d.camera.raster_to_camera, camera->RasterToCamera.m);
    copyMatrix(d.camera.camera_to_world, camera->CameraToWorld.m);
    for (int i = 0; i < 3; ++i) { d.camera.dx_camera[i] = camera->dxCamera[i]; d.camera.dy_camera[i] = camera->dyCamera[i]; }
    d.camera.lens_radius = camera->lensRadius; d.camera.focal_distance = camera->focalDistance;
    d.camera.shutter_open = camera->shutterOpen; d.camera.shutter_close = camera->shutterClose;
    const Film &film = *camera->film;
    for (int i = 0; i < 2; ++i) {
        d.film.full_res[i] = film.fullResolution[i];
        d.film.crop_min[i] = film.cropMin[i]; d.film.crop_max[i] = film.cropMax[i];
        d.film.sample_min[i] = sampler->sampleMin[i]; d.film.sample_max[i] = sampler->sampleMax[i];
        d.integrator.pixel_min[i] = pixelMin[i]; d.integrator.pixel_max[i] = pixelMax[i];
    }
    d.film.filter_radius[0] = film.filter->rx; d.film.filter_radius[1] = film.filter->ry;
    std::memcpy(d.film.filter_table, film.filterTable, sizeof(film.filterTable));
    d.film.max_sample_luminance = film.maxSampleLuminance;
    d.film.scale = film.scale;
    d.integrator.max_depth = maxDepth; d.integrator.rr_threshold = rrThreshold;
    d.integrator.spp = (int32_t)sampler->samplesPerPixel;
    d.integrator.sobol_resolution = sampler->resolution; d.integrator.sobol_log2_resolution = sampler->log2Resolution;
    switch (sampler->kind) {
    case Sampler::Halton: d.integrator.sampler = MI_SAMPLER_HALTON; break;
    case Sampler::Random: d.integrator.sampler = MI_SAMPLER_RANDOM; break;
    case Sampler::Stratified: d.integrator.sampler = MI_SAMPLER_STRATIFIED; break;
    case Sampler::ZeroTwo: d.integrator.sampler = MI_SAMPLER_ZEROTWO; break;
    default: d.integrator.sampler = MI_SAMPLER_SOBOL;
    }
    d.integrator.pixel_sampler_dims = sampler->nSampledDimensions;
    d.integrator.strat_samples[0] = sampler->xPixelSamples; d.integrator.strat_samples[1] = sampler->yPixelSamples;
    d.integrator.strat_jitter = sampler->jitterSamples ? 1 : 0;
    for (int i = 0; i < 2; ++i) {
        d.integrator.halton_base_scales[i] = sampler->baseScales[i]; d.integrator.halton_base_exponents[i] = sampler->baseExponents[i];
        d.integrator.halton_mult_inverse[i] = sampler->multInverse[i];
    }
    d.integrator.halton_sample_stride = sampler->sampleStride;
    d.integrator.halton_sample_at_center = sampler->sampleAtPixelCenter ? 1 : 0;
    return fs;
}

// ---- run-time binding to the HIP library -------------------------------------------------
namespace {
struct DeviceApi {
    void *handle = nullptr;
    const char *(*last_error)() = nullptr;
    int (*ctx_create)(int, void *, mi_ctx **) = nullptr;
    void (*ctx_destroy)(mi_ctx *) = nullptr;
    int (*scene_upload)(mi_ctx *, const mi_scene_desc *) = nullptr;
    int (*render)(mi_ctx *, const mi_render_params *) = nullptr;
    int (*sync)(mi_ctx *) = nullptr;
    int (*film_download)(mi_ctx *, float *) = nullptr;
    int (*counters)(mi_ctx *, uint64_t *) = nullptr;
    int (*film_gather)(mi_ctx **, int, int) = nullptr;
    bool load() {
        if (handle) return true;
        const char *names[] = {"libpbrt_amd.so", "./libpbrt_amd.so"};
        std::string tried;
        const char *env = std::getenv("PBRT_AMD_LIB");
        if (env) handle = dlopen(env, RTLD_NOW);
        for (const char *n : names) { if (handle) break; handle = dlopen(n, RTLD_NOW); }
        if (!handle) {
            Dl_info info;   // next to this host library
            if (dladdr((void *)&CreatePathIntegrator, &info) && info.dli_fname) {
                std::string p(info.dli_fname);
                size_t s = p.find_last_of('/');
                p = (s == std::string::npos ? std::string(".") : p.substr(0, s)) + "/libpbrt_amd.so";
                handle = dlopen(p.c_str(), RTLD_NOW);
            }
        }
        if (!handle) { Error("cannot load libpbrt_amd.so (the HIP path tracer): %s", dlerror()); return false; }
#define BIND(field, sym) field = (decltype(field))dlsym(handle, sym); if (!field) { Error("libpbrt_amd.so lacks %s", sym); return false; }
        BIND(last_error, "mi_last_error") BIND(ctx_create, "mi_ctx_create") BIND(ctx_destroy, "mi_ctx_destroy")
        BIND(scene_upload, "mi_scene_upload") BIND(render, "mi_render") BIND(sync, "mi_sync")
        BIND(film_download, "mi_film_download") BIND(counters, "mi_counters") BIND(film_gather, "mi_film_gather")
#undef BIND
        return true;
    }
};
DeviceApi g_dev;
}  // namespace

void WavefrontPathIntegrator::Render(const Scene &scene) {
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    std::unique_ptr<FlatScene> fs = Flatten(scene);
    if (!g_dev.load()) { g_renderFailed = true; return; }
    int world = std::max(1, nGpus);
    std::vector<mi_ctx *> ctxs(world, nullptr);
    for (int r = 0; r < world; ++r) {
        if (g_dev.ctx_create(r, nullptr, &ctxs[r]) != 0 || g_dev.scene_upload(ctxs[r], &fs->desc) != 0) {
            Error("GPU %d: %s", r, g_dev.last_error());
            for (mi_ctx *c : ctxs) if (c) g_dev.ctx_destroy(c);
            g_renderFailed = true;
            return;
        }
    }
    auto t1 = clk::now();
    // image tiles shard across GPUs (scene replicated); each device renders only its tiles.  mi_render is asynchronous: the loop
    // queues every device's frame and the GPUs render concurrently; mi_film_gather then waits for all of them and sums the films
    // into GPU 0's with one grouped ncclReduce over xGMI (replaces Film::MergeFilmTile across devices, film.cpp:117-130).
    bool ok = true;
    for (int r = 0; r < world && ok; ++r) {
        mi_render_params rp;
        std::memset(&rp, 0, sizeof(rp));
        rp.rank = r; rp.world = world; rp.spp_begin = 0; rp.spp_end = -1;
        if (g_dev.render(ctxs[r], &rp) != 0) { Error("GPU %d render: %s", r, g_dev.last_error()); ok = false; }
    }
    if (ok && g_dev.film_gather(ctxs.data(), world, 0) != 0) { Error("film gather: %s", g_dev.last_error()); ok = false; }
    auto t2 = clk::now();
    Film &film = *camera->film;
    std::vector<float> rgbw(4 * film.pixels.size());
    uint64_t total[MI_CNT_COUNT] = {0};
    if (ok && g_dev.film_download(ctxs[0], rgbw.data()) != 0) { Error("GPU 0 film: %s", g_dev.last_error()); ok = false; }
    for (int r = 0; r < world; ++r) {
        uint64_t c[MI_CNT_COUNT];
        if (ok && g_dev.counters(ctxs[r], c) == 0) for (int i = 0; i < MI_CNT_COUNT; ++i) total[i] += c[i];
        g_dev.ctx_destroy(ctxs[r]);
    }
    if (ok && total[MI_CNT_TRACE_GUARD_TRIPS] != 0) {   // traversal waves dropped their rays (non-termination guard): those pixels are wrong
        Error("%llu traversal wave(s) hit the non-termination guard: the frame is invalid", (unsigned long long)total[MI_CNT_TRACE_GUARD_TRIPS]);
        ok = false;
    }
    if (!ok) { Error("rendering failed: no image written"); g_renderFailed = true; return; }   // never an image with missing tiles
    film.MergeFilm(rgbw.data());
    film.WriteImage();
    if (!g_quiet) {
        double setup = std::chrono::duration<double>(t1 - t0).count(), render = std::chrono::duration<double>(t2 - t1).count();
        double rays = double(total[MI_CNT_CLOSEST_RAYS] + total[MI_CNT_SHADOW_RAYS]);
        std::printf("Integrator::Render(): flatten+upload %.3f s, render %.3f s on %d GPU(s)\n", setup, render, world);
        std::printf("  Camera rays traced %llu (%.2f Msamples/s); Regular + Shadow ray intersection tests %llu + %llu (%.2f Mrays/s)\n",
                    (unsigned long long)total[MI_CNT_CAMERA_RAYS], total[MI_CNT_CAMERA_RAYS] / render * 1e-6,
                    (unsigned long long)total[MI_CNT_CLOSEST_RAYS], (unsigned long long)total[MI_CNT_SHADOW_RAYS], rays / render * 1e-6);
    }
}

}  // namespace pbrt_amd
