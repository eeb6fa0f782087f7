/*
 * pbrt_amd.h -- the drop-in boundary of the MI355X wavefront path tracer.
 *
 * C ABI of libpbrt_amd.so (hand-written HIP for gfx950 behind extern "C").  It replaces ONE
 * call of the reference: `integrator->Render(*scene)` (reference src/core/api.cpp:1623), i.e.
 * SamplerIntegrator::Render -> PathIntegrator::Li (src/core/integrator.cpp:228-339,
 * src/integrators/path.cpp:64-188) with everything under it.  Everything a reference-side
 * Integrator subclass can see of a built Scene is passed in as plain-old-data:
 * pointers + sizes, no C++ types, no torch types.  INTEGRATION.md shows the ~100-line
 * `WavefrontPathIntegrator : public Integrator` a pbrt-v3 maintainer would add on top of this.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; mi_last_error() gives the message
 *     (pbrt's own convention is Error()/Warning() + continue, src/core/error.cpp:89-102;
 *      no exception crosses this boundary).
 *   - all pointers in mi_scene_desc are HOST pointers; the library owns every device
 *     allocation.  mi_scene_upload copies; the caller may free its arrays afterwards.
 *   - one mi_ctx per GPU (one process per GPU in multi-GPU runs); calls on one ctx are
 *     serialised by the caller, work is enqueued on the ctx stream.
 *   - Float == IEEE binary32 (src/core/pbrt.h:156), Spectrum == RGB, 3 floats
 *     (src/core/spectrum.h:429).
 */
#ifndef PBRT_AMD_H
#define PBRT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_ABI_VERSION 14   /* 14: MI_CNT_FILM_GATHER_BUILDS, film contents tracked for mi_film_gather, mi_rccl_probe; 13: mi_owned_tiles / mi_tile_owner (2-D lattice tile map), mi_trace_clock */

/* ---------------------------------------------------------------- geometry ---------- */

/* LinearBVHNode exactly as BVHAccel flattens it (src/accelerators/bvh.cpp:95-104, :640-658):
 * DFS order, first child = this+1, second child = `offset`; leaf iff n_prims > 0. The shim
 * collapses this tree to its 128-byte BVH4 device layout; the reference hands its own array. */
typedef struct mi_bvh2_node {
    float bmin[3], bmax[3];
    int32_t offset;   /* primitivesOffset (leaf) | secondChildOffset (interior) */
    uint16_t n_prims; /* 0 -> interior */
    uint8_t axis;     /* interior: split axis */
    uint8_t pad;
} mi_bvh2_node;

/* per-TriangleMesh flags (src/shapes/triangle.h:51-69) */
#define MI_MESH_HAS_N 1u  /* per-vertex normals present */
#define MI_MESH_HAS_UV 2u /* per-vertex uv present (else (0,0),(1,0),(1,1): triangle.h:98-108) */
#define MI_MESH_HAS_S 4u  /* per-vertex tangents present */
#define MI_MESH_FLIP 8u   /* reverseOrientation ^ transformSwapsHandedness (triangle.cpp:417-421) */

typedef struct mi_mesh {
    uint32_t flags;
    int32_t material; /* index into materials[]; -1 = no material (null BSDF, path.cpp:108) */
} mi_mesh;

/* ---------------------------------------------------------------- materials --------- */

/* One BxDF of a BSDF (src/core/reflection.h).  For a material whose textures are all constant the lobe
 * list Material::ComputeScatteringFunctions would build (src/materials/ *.cpp) is a function of
 * the material alone and is passed pre-evaluated (clamped, black lobes dropped, roughness
 * remapped: matte.cpp:54-61, plastic.cpp:52-69, microfacet.h:123-128).  Materials with image / procedural
 * textures or a bump map carry an mi_material_desc (below) and the same list is built per hit. */
enum mi_bxdf_type {
    MI_BXDF_LAMBERT_R = 0,    /* LambertianReflection     reflection.cpp:178  */
    MI_BXDF_LAMBERT_T = 1,    /* LambertianTransmission   reflection.cpp:187,391-403 */
    MI_BXDF_OREN_NAYAR = 2,   /* OrenNayar                reflection.cpp:197-219 */
    MI_BXDF_SPECULAR_R = 3,   /* SpecularReflection       reflection.cpp:136-143 */
    MI_BXDF_SPECULAR_T = 4,   /* SpecularTransmission     reflection.cpp:150-166 */
    MI_BXDF_FRESNEL_SPEC = 5, /* FresnelSpecular          reflection.cpp:477-511 */
    MI_BXDF_MICROFACET_R = 6, /* MicrofacetReflection     reflection.cpp:226-236,405-423 */
    MI_BXDF_MICROFACET_T = 7, /* MicrofacetTransmission   reflection.cpp:244-266,425-448 */
    MI_BXDF_FRESNEL_BLEND = 8, /* FresnelBlend             reflection.cpp:279-298,450-475 */
    MI_BXDF_BSSRDF_ADAPTER = 9 /* SeparableBSSRDFAdapter   bssrdf.h:158-174: f = Sw(wi) * eta^2, etaB = the BSSRDF's eta (built by k_shade_vol / the oracle at a BSSRDF exit point) */
};
enum mi_fresnel_type { MI_FRESNEL_NOOP = 0, MI_FRESNEL_DIELECTRIC = 1, MI_FRESNEL_CONDUCTOR = 2 };

typedef struct mi_bxdf {
    int32_t type;       /* mi_bxdf_type */
    int32_t fresnel;    /* mi_fresnel_type (SPECULAR_R / MICROFACET_R only) */
    int32_t scaled;     /* 1 -> wrapped in ScaledBxDF(scale) (mixmat.cpp:57-63) */
    int32_t distrib;    /* 0 = TrowbridgeReitz (all stock materials), 1 = Beckmann */
    float R[3];         /* R | Rd (FresnelBlend) | T for *_T lobes lives in T */
    float T[3];         /* T | Rs (FresnelBlend) */
    float scale[3];     /* ScaledBxDF factor */
    float alphax, alphay;
    float etaA, etaB;   /* dielectric: FresnelDielectric(etaI=etaA, etaT=etaB) / etaA,etaB of *_T */
    float eta_c[3];     /* conductor: etaT (etaI = 1) */
    float k_c[3];       /* conductor: k */
    float A, B;         /* OrenNayar precomputed (reflection.h:  A = 1 - s2/(2(s2+.33)), B = .45 s2/(s2+.09)) */
} mi_bxdf;

#define MI_MAX_BXDFS 8 /* reflection.h:199 */
typedef struct mi_material {
    int32_t n_bxdfs;
    float eta; /* BSDF::eta (reflection.h:156; glass.cpp:58, uber.cpp:56-60) */
    mi_bxdf bxdfs[MI_MAX_BXDFS];
} mi_material;


/* ---------------------------------------------------------------- textures ---------- */

/* Texture<Float> / Texture<Spectrum> (core/texture.h:139-144) as a node table: every "Texture" directive and every
 * inline parameter value becomes one mi_texture; parameters of type texture refer to nodes by index (children always
 * precede their parents).  The device (and the oracle) evaluate nodes per hit exactly as the reference's
 * Texture::Evaluate(const SurfaceInteraction&) does; Ptex is not carried. */
enum mi_tex_type {
    MI_TEX_CONSTANT = 0,     /* textures/constant.h */
    MI_TEX_SCALE = 1,        /* textures/scale.h:  tex1 * tex2 */
    MI_TEX_MIX = 2,          /* textures/mix.h:    (1-amount) tex1 + amount tex2 */
    MI_TEX_BILERP = 3,       /* textures/bilerp.h */
    MI_TEX_IMAGEMAP = 4,     /* textures/imagemap.h + core/mipmap.h */
    MI_TEX_UV = 5,           /* textures/uv.h */
    MI_TEX_CHECKERBOARD = 6, /* textures/checkerboard.h (dim 2 with aa none|closedform, dim 3) */
    MI_TEX_DOTS = 7,         /* textures/dots.h */
    MI_TEX_FBM = 8,          /* textures/fbm.h */
    MI_TEX_WRINKLED = 9,     /* textures/wrinkled.h */
    MI_TEX_MARBLE = 10,      /* textures/marble.h */
    MI_TEX_WINDY = 11        /* textures/windy.h */
};
enum mi_tex_mapping { /* core/texture.h:50-137, core/texture.cpp:84-163 */
    MI_MAP_UV = 0, MI_MAP_SPHERICAL = 1, MI_MAP_CYLINDRICAL = 2, MI_MAP_PLANAR = 3, MI_MAP_IDENTITY3D = 4
};
typedef struct mi_texture {
    int32_t type;     /* mi_tex_type */
    int32_t spectrum; /* 0: Texture<Float> (value[0] etc.), 1: Texture<Spectrum> */
    int32_t tex1, tex2, amount; /* child nodes (SCALE, MIX, CHECKERBOARD tex1/tex2, DOTS tex1 = outsideDot, tex2 = insideDot) */
    int32_t mapping;  /* mi_tex_mapping */
    int32_t image;    /* IMAGEMAP: index into images[] */
    int32_t dim, aa;  /* CHECKERBOARD: "dimension" 2|3; "aamode" 0 none, 1 closedform */
    int32_t octaves;  /* FBM / WRINKLED / MARBLE */
    float omega, scale, variation;
    float value[3];   /* CONSTANT */
    float v00[3], v01[3], v10[3], v11[3]; /* BILERP */
    float su, sv, du, dv; /* UVMapping2D; PlanarMapping2D keeps ds, dt in du, dv */
    float vs[3], vt[3];   /* PlanarMapping2D */
    float w2t[16];        /* the Transform the mapping was constructed with (SPHERICAL / CYLINDRICAL / IDENTITY3D), row major */
} mi_texture;

/* MIPMap<T> after its constructor (core/mipmap.h:101-199): image resampled to power-of-two size, box-filtered pyramid.
 * Texel (s,t) of level l is texels[level_offset(l) + (t * w_l + s) * channels], w_l = max(1, width >> l); levels are
 * stored finest first, back to back. */
typedef struct mi_image {
    int32_t width, height; /* level 0 */
    int32_t levels;
    int32_t channels;      /* 1: MIPMap<Float>, 3: MIPMap<RGBSpectrum> */
    int32_t trilinear;     /* doTrilinear */
    int32_t wrap;          /* ImageWrap: 0 Repeat, 1 Black, 2 Clamp (mipmap.h:55) */
    float max_aniso;
    float pad;
    const float *texels;
} mi_image;

/* Material parameters as textures: what Material::ComputeScatteringFunctions evaluates per hit (materials/ *.cpp,
 * Material::Bump core/material.cpp:46-83).  `textured` = 0 means every parameter is a constant and there is no bump map:
 * the pre-evaluated mi_material of the same index is then exact and the per-hit evaluation is skipped.  Node indices are
 * -1 where the material has no such parameter (or the reference's "...OrNull" lookup returned null). */
enum mi_material_type {
    MI_MAT_MATTE = 0, MI_MAT_PLASTIC = 1, MI_MAT_GLASS = 2, MI_MAT_MIRROR = 3, MI_MAT_METAL = 4, MI_MAT_UBER = 5,
    MI_MAT_SUBSTRATE = 6, MI_MAT_TRANSLUCENT = 7, MI_MAT_MIX = 8
};
typedef struct mi_material_desc {
    int32_t type;            /* mi_material_type */
    int32_t textured;
    int32_t remap_roughness;
    int32_t bump;            /* float node */
    int32_t Kd, Ks, Kr, Kt;  /* spectrum nodes */
    int32_t opacity;         /* uber */
    int32_t eta_s, k_s;      /* metal: eta, k (spectrum) */
    int32_t amount;          /* mix: spectrum node */
    int32_t sigma, roughness, uroughness, vroughness; /* float nodes */
    int32_t eta_f;           /* glass / uber: "eta" | "index" (float node) */
    int32_t reflect, transmit; /* translucent (spectrum nodes) */
    int32_t m1, m2;          /* mix: indices into materials[] / material_descs[] */
    int32_t pad;
} mi_material_desc;

/* ---------------------------------------------------------------- lights ------------ */

enum mi_light_type {
    MI_LIGHT_AREA_TRI = 0, /* DiffuseAreaLight on one Triangle (api.cpp:1357-1366, lights/diffuse.cpp) */
    MI_LIGHT_POINT = 1,    /* PointLight   lights/point.cpp:44-53  */
    MI_LIGHT_DISTANT = 2,  /* DistantLight lights/distant.cpp:49-67 */
    MI_LIGHT_SPOT = 4,     /* SpotLight    lights/spot.cpp:54-72   */
    MI_LIGHT_AREA_SPHERE = 5, /* DiffuseAreaLight on a Sphere (shapes/sphere.cpp:221-309): tri = primitive index, sphere = index into spheres[] */
    MI_LIGHT_INFINITE = 3  /* InfiniteAreaLight, constant L only (lights/infinite.cpp:92-132) */
};
typedef struct mi_light {
    int32_t type;
    int32_t tri;       /* AREA_TRI: triangle index in BVH primitive order */
    int32_t two_sided; /* AREA_TRI / AREA_SPHERE: diffuse.h:56-58 */
    int32_t sphere;    /* AREA_SPHERE: index into mi_scene_desc::spheres */
    float L[3];        /* Lemit | I | L */
    float area;        /* AREA_TRI: Triangle::Area() (triangle.cpp:575-581) */
    float pos[3];      /* POINT: pLight; DISTANT: wLight (normalised, world) */
    float world_radius;/* DISTANT/INFINITE: Scene bound radius (Preprocess) */
    float world_center[3];
    float pad2;
    /* SPOT (lights/spot.cpp:40-72): pos = pLight, L = I * scale; frame = the upper-left 3x3 of WorldToLight (row major:
     * Falloff() applies it to a direction), cos_total_width / cos_falloff_start as the constructor computes them */
    float frame[9];
    float cos_total_width, cos_falloff_start;
    float pad3;
    /* INFINITE with a radiance map (lights/infinite.cpp:43-132): env_map = 1 + index into mi_scene_desc::envmaps (0: constant L);
     * frame = WorldToLight 3x3 (Le, Pdf_Li), l2w = LightToWorld 3x3 (Sample_Li) */
    int32_t env_map;
    float l2w[9];
    float pad4[2];
} mi_light;

/* Sphere (shapes/sphere.{h,cpp}), the one quadric this path carries (the reference's example scene lights itself with one).
 * Primitive i (BVH order) is a sphere iff tri_indices[3*i] == MI_PRIM_SPHERE; tri_indices[3*i+1] is then the index here;
 * material and area light come through tri_mesh[i] / tri_light[i] as for triangles. */
#define MI_PRIM_SPHERE 0xFFFFFFFFu
typedef struct mi_sphere {
    float o2w[16], w2o[16];  /* ObjectToWorld / WorldToObject: Transform::m, row major (core/transform.h) */
    float radius, zmin, zmax, theta_min, theta_max, phi_max;   /* as the constructor leaves them (sphere.h:50-58) */
    uint32_t flags;          /* bit 0: reverseOrientation, bit 1: transformSwapsHandedness */
    float area;              /* Sphere::Area() sphere.cpp:219 */
} mi_sphere;

/* Radiance map of an InfiniteAreaLight as its constructor leaves it (infinite.cpp:43-84): MIPMap level 0 (texels * L, resampled to
 * power-of-two sizes by MIPMap's constructor, mipmap.h:120-182) and the Distribution2D over the (2*width) x (2*height) image of
 * luminance * sin(theta).  Lookups on the path use level 0 only (Lookup(st) with zero width -> triangle(0, st), mipmap.h:247-276). */
typedef struct mi_envmap {
    int32_t width, height;
    const float *rgb;           /* 3 * width * height, row major, row 0 = top (theta = 0) */
    const float *cond_func;     /* (2*width) * (2*height): pConditionalV[v]->func */
    const float *cond_cdf;      /* (2*width + 1) * (2*height) */
    const float *cond_func_int; /* 2*height */
    const float *marg_func;     /* 2*height: pMarginal->func */
    const float *marg_cdf;      /* 2*height + 1 */
    float marg_func_int;
    float pad;
} mi_envmap;

/* Object instancing (ObjectBegin / ObjectInstance; api.cpp:1555-1591, TransformedPrimitive core/primitive.cpp:76-111) as a two-level
 * hierarchy.  The host hands the reference's own structure over (the default since round 2); with PBRT_AMD_INSTANCING=0 it FLATTENS instances
 * into world-space copies instead (n_instances = 0: same hits within float tolerance):
 *   - primitive i of the top-level BVH order is a TransformedPrimitive iff tri_indices[3*i] == MI_PRIM_INSTANCE; tri_indices[3*i+1]
 *     indexes instances[];
 *   - an object's primitives (its own BVHAccel's order; vertices in the space the object was defined in) occupy
 *     [first_prim, first_prim + n_prims) of the tri_* arrays, AFTER the n_top_prims top-level primitives; its LinearBVHNode array
 *     occupies [first_node, first_node + n_nodes) of bvh_nodes, AFTER the n_bvh_nodes top-level nodes, with child / primitive offsets
 *     relative to the object's own arrays.
 * Instanced primitives cannot be area lights (api.cpp:1351-1353).  The default since round 2 (PBRT_AMD_INSTANCING=0 flattens): the device
 * traverses the second level in k_trace / k_shade / k_shade_vol <..., INST> (fixtures edge_instances*.pfm, edge_vol_inst.pfm, edge_sss_inst.pfm). */
#define MI_PRIM_INSTANCE 0xFFFFFFFEu
typedef struct mi_instance {
    float i2w[16], w2i[16]; /* InstanceToWorld (the CTM at ObjectInstance) and its stored inverse, row major */
    uint32_t object;        /* index into objects[] */
    uint32_t pad[3];
} mi_instance;
typedef struct mi_object {
    uint32_t first_prim, n_prims;
    uint32_t first_node, n_nodes;
} mi_object;

/* ---------------------------------------------------------------- camera/film ------- */

typedef struct mi_camera {     /* PerspectiveCamera, cameras/perspective.cpp:45-67 */
    float raster_to_camera[16]; /* Transform::m, row major */
    float camera_to_world[16];  /* CameraToWorld (static; AnimatedTransform start) */
    float dx_camera[3], dy_camera[3];
    float lens_radius, focal_distance;
    float shutter_open, shutter_close;
} mi_camera;

#define MI_FILTER_TABLE_WIDTH 16 /* film.h:96 */
typedef struct mi_film {     /* Film, core/film.cpp:45-86 */
    int32_t full_res[2];
    int32_t crop_min[2], crop_max[2];     /* croppedPixelBounds */
    int32_t sample_min[2], sample_max[2]; /* Film::GetSampleBounds() */
    float filter_radius[2];
    float filter_table[MI_FILTER_TABLE_WIDTH * MI_FILTER_TABLE_WIDTH];
    float max_sample_luminance;
    float scale;
} mi_film;

/* Light-selection strategy.  TABLE: one Distribution1D over all lights, passed as light_func / light_cdf
 * (UniformLightDistribution, PowerLightDistribution, or any scene with a single light).
 * SPATIAL: SpatialLightDistribution (lightdistrib.cpp:96-300); the library evaluates ComputeDistribution for
 * every voxel of the grid at upload time (the reference fills its hash table lazily with the same values). */
#define MI_LIGHT_STRATEGY_TABLE 0
#define MI_LIGHT_STRATEGY_SPATIAL 1

#define MI_SAMPLER_SOBOL 0
#define MI_SAMPLER_HALTON 1
/* ABI v11: the samplers whose values come from ONE PCG32 stream per 16x16 tile (Sampler::Clone(seed = tile index), integrator.cpp:246-248): a sample's
 * values depend on how many numbers the tile's earlier samples drew, so the library walks each tile's pixels and samples in the reference's order, one
 * path per tile in flight ("tile-serial" rounds: correct, slow -- DESIGN.md s.7).  RANDOM: samplers/random.cpp (every value from the stream).
 * STRATIFIED / ZEROTWO: PixelSamplers (core/sampler.cpp:100-135) -- the first pixel_sampler_dims 1D and 2D dimensions of a pixel's samples are
 * generated by StartPixel (samplers/stratified.cpp:43-70, zerotwosequence.cpp:53-68), later ones come from the stream. */
#define MI_SAMPLER_RANDOM 2
#define MI_SAMPLER_STRATIFIED 3
#define MI_SAMPLER_ZEROTWO 4
/* ABI v12: MaxMinDistSampler (samplers/maxmin.{h,cpp}) -- a PixelSampler like ZEROTWO whose FIRST 2D dimension is (i / spp, C i) with the generator matrix
 * C = CMaxMinDist[log2 spp] (core/lowdiscrepancy.cpp: a table of matrices found by search, Gruenschloss et al.; it cannot be re-derived, so the caller
 * passes the matrix it selected: the reference-side binding reads it from the reference's own sampler object).  spp: a power of two <= 2^16. */
#define MI_SAMPLER_MAXMIN 5
#define MI_SAMPLER_IS_TILE_SERIAL(s) ((s) >= MI_SAMPLER_RANDOM)

typedef struct mi_integrator { /* PathIntegrator + SobolSampler parameters */
    int32_t max_depth;          /* path.cpp:193, default 5 */
    float rr_threshold;         /* path.cpp:208, default 1 */
    int32_t pixel_min[2], pixel_max[2]; /* pixelBounds (path.cpp:195-207) */
    int32_t spp;                /* RoundUpPow2(pixelsamples) (sobol.h:52) */
    int32_t sobol_resolution;   /* sobol.h:58-59 */
    int32_t sobol_log2_resolution;
    int32_t light_strategy;     /* MI_LIGHT_STRATEGY_*: CreateLightSampleDistribution (lightdistrib.cpp:48-66, path.cpp:72) */
    int32_t spatial_max_voxels; /* SpatialLightDistribution maxVoxels (lightdistrib.h:92), 64; 0 -> 64 */
    /* the GlobalSampler driving the path: SobolSampler (samplers/sobol.cpp) or HaltonSampler (samplers/halton.cpp, pbrt's default).
     * Halton: spp is NOT rounded to a power of two; the fields below are the constructor's results (halton.cpp:71-97);
     * the digit permutations (ComputeRadicalInversePermutations with the default RNG, lowdiscrepancy.cpp:2490-2504) are
     * regenerated by the library. */
    int32_t sampler;                    /* MI_SAMPLER_* */
    int32_t halton_base_scales[2];      /* 2^e0 >= min(res.x,128), 3^e1 >= min(res.y,128) */
    int32_t halton_base_exponents[2];
    int32_t halton_sample_stride;       /* baseScales[0] * baseScales[1] */
    int32_t halton_mult_inverse[2];
    int32_t halton_sample_at_center;    /* "samplepixelcenter" */
    /* tile-serial samplers (ABI v11).  spp: RANDOM "pixelsamples" (default 4); STRATIFIED xsamples * ysamples; ZEROTWO RoundUpPow2("pixelsamples") */
    int32_t pixel_sampler_dims;         /* "dimensions" (nSampledDimensions, default 4); 0 for RANDOM */
    int32_t strat_samples[2];           /* "xsamples", "ysamples" (default 4 x 4) */
    int32_t strat_jitter;               /* "jitter" (default true) */
    uint32_t maxmin_matrix[32];         /* MI_SAMPLER_MAXMIN: MaxMinDistSampler::CPixel[0..31] (the 32 columns of CMaxMinDist[Log2Int(spp)]) */
} mi_integrator;

/* ---------------------------------------------------------------- the scene --------- */

/* Participating media (VolPathIntegrator, integrators/volpath.cpp:55-190).  A medium is what MakeMedium builds
 * (core/api.cpp:685-731): HomogeneousMedium (media/homogeneous.{h,cpp}) or GridDensityMedium (media/grid.{h,cpp}); the phase function of
 * both is HenyeyGreenstein(g) (core/medium.cpp:189-215).  The device renders them in k_shade_vol (csrc/pt_volpath.h) when the integrator is
 * "volpath"; PathIntegrator ignores media (path.cpp passes handleMedia = false), so "path" scenes with media declarations render without them. */
enum mi_medium_type { MI_MEDIUM_HOMOGENEOUS = 0, MI_MEDIUM_GRID = 1 };
typedef struct mi_medium {
    int32_t type;
    float sigma_a[3], sigma_s[3];
    float sigma_t[3];          /* homogeneous: sigma_a + sigma_s (homogeneous.h:52); grid: the scalar (sigma_a + sigma_s)[0] in all three (grid.h:69) */
    float g;
    int32_t nx, ny, nz;        /* grid only */
    float inv_max_density;     /* grid.h:77 */
    float world_to_medium[16]; /* Inverse(medium2world * data2Medium) (api.cpp:723-726, grid.h:61), row major */
    float pad;
    const float *density;      /* nx*ny*nz, index (z*ny + y)*nx + x (grid.h:83); NULL for homogeneous media */
} mi_medium;
enum mi_integrator_type { MI_INTEGRATOR_PATH = 0, MI_INTEGRATOR_VOLPATH = 1 };

/* Subsurface scattering (the BSSRDF branch of path.cpp:153-174 / volpath.cpp:153-180).  SubsurfaceMaterial and
 * KdSubsurfaceMaterial (materials/subsurface.cpp, kdsubsurface.cpp) build the BSDF GlassMaterial builds -- their entry in material_descs[]
 * is an MI_MAT_GLASS record with textured = 1 -- and additionally a TabulatedBSSRDF (core/bssrdf.{h,cpp}) from per-hit coefficients and
 * the BSSRDFTable their constructor computed (ComputeBeamDiffusionBSSRDF, bssrdf.cpp:145-176).  Each Material OBJECT with a BSSRDF keeps
 * its own slot in materials[] (Sample_Sp accepts probe hits on primitives of the SAME material object only, bssrdf.cpp:302).
 * The device evaluates them in k_shade_vol (probe-segment hit chains traced by the shading lane). */
typedef struct mi_bssrdf_table {
    int32_t n_rho, n_radius;      /* 100, 64 */
    const float *rho_samples;     /* n_rho */
    const float *radius_samples;  /* n_radius */
    const float *profile;         /* n_rho * n_radius */
    const float *rho_eff;         /* n_rho */
    const float *profile_cdf;     /* n_rho * n_radius */
} mi_bssrdf_table;
enum mi_bssrdf_kind { MI_BSSRDF_NONE = 0, MI_BSSRDF_SUBSURFACE = 1, MI_BSSRDF_KDSUBSURFACE = 2 };
typedef struct mi_bssrdf_desc {
    int32_t kind;             /* mi_bssrdf_kind */
    int32_t table;            /* index into bssrdf_tables[] */
    int32_t sigma_a, sigma_s; /* SUBSURFACE: spectrum nodes (mm^-1 before `scale`) */
    int32_t Kd, mfp;          /* KDSUBSURFACE: spectrum nodes (SubsurfaceFromDiffuse, bssrdf.cpp:178-188) */
    float scale, eta, g;
    int32_t pad;
} mi_bssrdf_desc;

typedef struct mi_scene_desc {
    uint32_t abi_version; /* = MI_ABI_VERSION */
    /* vertices (world space, triangle.cpp:72-74) */
    uint32_t n_verts;
    const float *P;  /* 3*n_verts */
    const float *N;  /* 3*n_verts, zeros where the mesh has none; may be NULL if no mesh has normals */
    const float *UV; /* 2*n_verts; may be NULL */
    /* triangles in BVHAccel::primitives order (bvh.cpp:205 `primitives.swap(orderedPrims)`) */
    uint32_t n_tris;
    const uint32_t *tri_indices; /* 3*n_tris global vertex indices */
    const uint32_t *tri_mesh;    /* n_tris -> meshes[] */
    const int32_t *tri_light;    /* n_tris -> lights[] index or -1 */
    uint32_t n_meshes;
    const mi_mesh *meshes;
    /* acceleration structure */
    uint32_t n_bvh_nodes;
    const mi_bvh2_node *bvh_nodes;
    /* shading */
    uint32_t n_materials;
    const mi_material *materials;
    uint32_t n_lights;
    const mi_light *lights;
    /* light-selection distribution (Distribution1D over all lights: uniform/power,
     * lightdistrib.cpp:56-75); func has n_lights entries, cdf n_lights+1 */
    const float *light_func;
    const float *light_cdf;
    float light_func_int;
    float pad0;
    mi_camera camera;
    mi_film film;
    mi_integrator integrator;
    uint32_t n_envmaps;
    uint32_t n_spheres;
    const mi_envmap *envmaps;
    const mi_sphere *spheres;
    /* textures (row f2): node table, image pyramids, per-material parameter nodes (NULL or n_materials entries), and
     * per-mesh alpha masks: mesh_alpha[2*m] = TriangleMesh::alphaMask, [2*m+1] = shadowAlphaMask as float nodes, -1 = none
     * (triangle.cpp:333-338,532-570); NULL when no mesh has one */
    uint32_t n_textures;
    uint32_t n_images;
    const mi_texture *textures;
    const mi_image *images;
    const mi_material_desc *material_descs;
    const int32_t *mesh_alpha;
    /* two-level instancing (see mi_instance): 0 / NULL unless the host was asked not to flatten */
    uint32_t n_instances;
    uint32_t n_objects;
    uint32_t n_top_prims;   /* primitives of the top-level BVH (= n_tris when there are no instances) */
    uint32_t pad1;
    const mi_instance *instances;
    const mi_object *objects;
    /* participating media (see mi_medium): mesh_medium[2*m] = the GeometricPrimitive's MediumInterface::inside, [2*m+1] = outside as
     * indices into media[], -1 = no medium (primitive.h:79, api.cpp:1355,1496-1516); NULL when no primitive names one.
     * camera_medium: Camera::medium (camera.h:70; the outside medium of the graphics state at WorldEnd, api.cpp:793), -1 = vacuum */
    uint32_t n_media;
    int32_t camera_medium;
    int32_t integrator_type; /* MI_INTEGRATOR_* */
    uint32_t pad2;
    const mi_medium *media;
    const int32_t *mesh_medium;
    /* subsurface materials (see mi_bssrdf_desc): material_bssrdf has n_materials entries or is NULL */
    uint32_t n_bssrdf_tables;
    uint32_t pad3;
    const mi_bssrdf_table *bssrdf_tables;
    const mi_bssrdf_desc *material_bssrdf;
} mi_scene_desc;

/* ---------------------------------------------------------------- ABI ---------------- */

typedef struct mi_ctx mi_ctx;

const char *mi_last_error(void);
int mi_abi_version(void);

/* context: one per GPU. `stream` = a hipStream_t to run on (0 -> library creates its own).
 * Several contexts may share a device and be driven from different host threads (one thread per context).  Scenes with textures,
 * alpha masks or instances keep their tables in per-device __constant__ symbols, so the passes of such contexts take turns on a device
 * (each drains its stream before the next one starts); a context that is alone on its device stays fully asynchronous. */
int mi_ctx_create(int device_ordinal, void *stream, mi_ctx **out);
void mi_ctx_destroy(mi_ctx *ctx);

/* Scene hand-over: what `Integrator::Render(const Scene&)` receives (integrator.h:53-58).
 * Builds the BVH4, triangle records and tables in HBM. */
int mi_scene_upload(mi_ctx *ctx, const mi_scene_desc *scene);

/* Tile -> rank map of an N-way tile-sharded frame (the ONE definition: mi_render, the reference-side bindings, parallel.py and the test
 * checker all call it).  Tile (tx, ty) of the reference's 16x16 grid (integrator.cpp:233-240) belongs to rank (tx + skew(world) * ty) mod world:
 * a skewed 2-D lattice, so that a rank's tiles are spread over the whole image in BOTH directions (the row-major t mod world of rounds 1-3 gave
 * every rank full-height tile columns whenever world divides the tiles per row, e.g. 120 tiles per row at 1080p and world = 8).  skew(world) is
 * the s in [0, world) whose lattice {(dx, dy): dx + s dy = 0 mod world} has the longest shortest vector (ties: the smallest s): 1 for world 2,
 * 2 for 4 and 5, 3 for 8 (nearest same-rank tiles sqrt(8) tiles apart).  Any exact cover of the tiles renders the same image: the Sobol' /
 * Halton sample of (pixel, k) is a function of the pixel and k alone. */
static inline int mi_tile_skew(int world) {
    int best = 0, bestLen = -1;
    for (int s = 0; s < world; ++s) {
        int len = 0x7fffffff;
        for (int dy = 0; dy <= world; ++dy)
            for (int dx = -world; dx <= world; ++dx) {
                if ((dx == 0 && dy == 0) || ((dx + s * dy) % world + world) % world != 0) continue;
                if (dx * dx + dy * dy < len) len = dx * dx + dy * dy;
            }
        if (len > bestLen) { bestLen = len; best = s; }
    }
    return best;
}
static inline int mi_tile_owner(int tx, int ty, int world, int skew) { return world <= 1 ? 0 : (int)(((int64_t)tx + (int64_t)skew * ty) % world); }
/* the row-major tile ids (ty * n_tiles_x + tx, ascending) of `rank`; returns their number; `out` may be NULL (count only).  Exported by
 * libpbrt_amd.so for callers that cannot include this header (pbrt-v3-distributed_amd/parallel.py). */
int64_t mi_owned_tiles(int n_tiles_x, int n_tiles_y, int rank, int world, uint32_t *out);

/* SamplerIntegrator::Render (integrator.cpp:228-339) over the 16x16 tiles owned by `rank` of
 * `world` (mi_tile_owner above; integrator.cpp:235-240 gives the tile grid), samples
 * [spp_begin, spp_end) of every owned pixel, accumulated into the device film
 * (FilmTile::AddSample semantics, film.h:121-161).  Asynchronous on the ctx stream: the call returns when the launches are
 * queued (contexts on different GPUs render concurrently when driven from one host thread); mi_sync / mi_film_download wait.
 * Two exceptions block inside the call: the first frame after mi_scene_upload or a change of (rank, world) allocates the
 * path state / tile list, and scenes with null-material surfaces (medium interfaces, which do not count as bounces,
 * path.cpp:107-111) read one queue counter back per pass to know when the last path has ended. */
typedef struct mi_render_params {
    int32_t rank, world;
    int32_t spp_begin, spp_end; /* spp_end = -1 -> integrator.spp */
    int32_t count_work;         /* 1 -> maintain nodes-visited / tris-tested counters (slower) */
    int32_t max_paths_in_flight;/* 0 -> default */
} mi_render_params;
int mi_render(mi_ctx *ctx, const mi_render_params *params);
int mi_sync(mi_ctx *ctx);

/* Film: FilmTilePixel{contribSum rgb, filterWeightSum} = 16 B per cropped pixel
 * (film.h:52-55) -- the multi-GPU gather payload.  The host Film performs
 * MergeFilmTile/WriteImage (film.cpp:117-130,168-210) on the downloaded array. */
int mi_film_clear(mi_ctx *ctx);
int mi_film_download(mi_ctx *ctx, float *rgbw /* 4 * cropped pixel count */);
void *mi_film_device_ptr(mi_ctx *ctx); /* float4 per cropped pixel, for RCCL by the caller */
/* Render into a caller-owned device buffer (e.g. a torch tensor that torch.distributed reduces over RCCL);
 * NULL switches back to the library's own film. */
int mi_film_bind(mi_ctx *ctx, void *device_float4_buffer);
int64_t mi_film_pixel_count(mi_ctx *ctx);
/* The exchange step of the tile-sharded render (replaces Film::MergeFilmTile, film.cpp:117-130, across GPUs):
 * adds the films of the n contexts (one per GPU, same scene, rendered with rank = i, world = n) into ctxs[root]'s film.  SPARSE
 * since round 5: context i packs the FilmTilePixels its samples can reach -- its tiles (mi_tile_owner) grown by the filter's ring,
 * what Film::GetFilmTile computes per tile (film.cpp:95-106) -- and the root adds the packed lists in context order: exact for
 * every filter (a context's film is zero elsewhere) and deterministic.  One context per GPU: the lists travel in ONE group of
 * ncclSend / ncclRecv over xGMI (RCCL, loaded at first use from librccl.so; one communicator per device set, cached).  Contexts
 * that share a device (testing on a one-GPU box; RCCL refuses duplicate devices) hand their lists over directly.  The library tracks
 * what a context's film holds: ONE sharding since the last mi_film_clear -> sparse; no shard (world 1), a second sharding accumulated
 * into the same film, or a buffer bound with mi_film_bind and not cleared since -> the whole film is added (nothing is ever dropped).
 * The reach lists, index and exchange buffers are kept in the sender's context and rebuilt only when the sharding, the film size or
 * the root's device changes (MI_CNT_FILM_GATHER_BUILDS): repeated frames allocate and upload nothing.  Waits for the renders of all
 * contexts first and returns when the root film is complete; n == 1 is a no-op. */
int mi_film_gather(mi_ctx **ctxs, int n, int root);
/* Stage-level BSDF lobes (core/reflection.cpp:703-785 building blocks): BxDF::f, Pdf and Sample_f(wo, u) of bxdfs[i] for record i
 * (wo, wi in the shading frame) -- replays the vectors dumped from the reference's own BxDF classes on the device. */
int mi_bxdf_eval(int device_ordinal, const mi_bxdf *bxdfs, const float *wo, const float *wi, const float *u, int64_t n, float *f, float *pdf,
                 float *wi_s, float *pdf_s, float *f_s, int32_t *type_s);
/* Stage-level light sampling: Light::Sample_Li(ref, u) and Light::Pdf_Li(ref, wi) (core/light.h:63-75) of light `light` of the uploaded scene
 * at a reference point (p, n with pError = 0; n = 0: a point in a medium) -- DiffuseAreaLight over a Triangle / a Sphere (Shape::Sample(ref, u),
 * Shape::Pdf(ref, wi): shapes/triangle.cpp:583-647, core/shape.cpp:57-108, shapes/sphere.cpp:325-400), point, spot, distant and infinite lights.
 * ray_*: the shadow ray of the VisibilityTester (Interaction::SpawnRayTo, interaction.h:72-78).  pdf_wi = Pdf_Li(ref, query.wi);
 * le_wi = Light::Le of a ray leaving along query.wi (what an escaped ray collects, path.cpp:97-98): infinite lights, 0 for the others. */
typedef struct mi_light_query { int32_t light; float p[3], n[3], u[2], wi[3]; } mi_light_query;
typedef struct mi_light_result { float wi[3], pdf, Li[3], ray_o[3], ray_d[3], ray_tmax, pdf_wi; int32_t delta; float le_wi[3]; } mi_light_result;
int mi_light_sample(mi_ctx *ctx, const mi_light_query *queries, int64_t n, mi_light_result *out);
/* Stage-level row f4: TabulatedBSSRDF::Sr / Sample_Sr / Pdf_Sr (core/bssrdf.cpp:199-233, 353-390) on an explicit table for coefficient triples, and
 * SubsurfaceFromDiffuse (:178-188: reflectance kd + mean free path -> sigma_a, sigma_s); HenyeyGreenstein::p / Sample_p (core/medium.cpp:189-213). */
typedef struct mi_bssrdf_query { float sigma_a[3], sigma_s[3]; int32_t ch; float r, u, kd[3], mfp[3]; } mi_bssrdf_query;
typedef struct mi_bssrdf_result { float sr[3], sample_sr, pdf_sr, sigma_a[3], sigma_s[3]; } mi_bssrdf_result;
int mi_bssrdf_eval(int device_ordinal, const mi_bssrdf_table *table, float eta, const mi_bssrdf_query *queries, int64_t n, mi_bssrdf_result *out);
typedef struct mi_hg_query { float g, wo[3], wi[3], u[2]; } mi_hg_query;
typedef struct mi_hg_result { float p, wi_s[3], p_s; } mi_hg_result;
int mi_phase_hg(int device_ordinal, const mi_hg_query *queries, int64_t n, mi_hg_result *out);
/* Stage-level libm: the float routines the path calls (std::sin / cos / acos / atan / atan2 / exp / log on Float: core/sampling.cpp:93-150,
 * core/geometry.h:1463-1486, core/microfacet.cpp:146-163, media/homogeneous.cpp:56, ...) as the device evaluates them (csrc/pt_libm.h: the
 * operation sequences of the glibc the reference links against).  out[i] = fn(a[i]) (atan2f: fn(a[i], b[i]); sincosf: out = sin, out2 = cos). */
enum { MI_LIBM_SINF = 0, MI_LIBM_COSF = 1, MI_LIBM_SINCOSF = 2, MI_LIBM_EXPF = 3, MI_LIBM_LOGF = 4, MI_LIBM_ACOSF = 5, MI_LIBM_ATANF = 6, MI_LIBM_ATAN2F = 7 };
int mi_libm_eval(int device_ordinal, int fn, const float *a, const float *b, int64_t n, float *out, float *out2);
/* which traversal kernels the uploaded scene runs: out[0] = 0 general steps over the full-precision 128-byte BVH4 (PBRT_AMD_TRACE=general),
 * 4 two-level (instanced) scene over the same nodes, 5 general steps over the 64-byte quantised BVH4 (the default for single-level scenes;
 * values 1-3 named layouts measured slower in round 2 and removed from the library);
 * out[1] = bytes per node, out[2] = nodes, out[3] = stack entries held in LDS per lane, out[4] = hot nodes each traversal block keeps in LDS
 * (nodesq[0 .. n_hot): the most visited nodes of the hot-node probe at upload; 0: none), out[5] = share of the probe paths' node visits that fell
 * on them, in 1e-6, out[6] = threads per block and out[7] = blocks per CU of the closest-hit / any-hit launches */
int mi_trace_info(mi_ctx *ctx, int64_t out[8]);

/* Work counters (names follow the reference's STAT_COUNTERs: integrator.cpp:48,
 * scene.cpp:40-42, triangle.cpp:45) */
enum mi_counter {
    MI_CNT_CAMERA_RAYS = 0,
    MI_CNT_CLOSEST_RAYS = 1, /* Scene::Intersect calls  */
    MI_CNT_SHADOW_RAYS = 2,  /* Scene::IntersectP calls */
    MI_CNT_NODES_CLOSEST = 3,/* BVH4 nodes fetched by closest-hit kernel launches */
    MI_CNT_TRIS_CLOSEST = 4,
    MI_CNT_NODES_ANY = 5,
    MI_CNT_TRIS_ANY = 6,
    MI_CNT_PATH_SEGMENTS = 7,
    MI_CNT_MIS_RAYS = 8,     /* the part of CLOSEST_RAYS traced by the MIS launches (integrator.cpp:202) */
    MI_CNT_NODES_MIS = 9,
    MI_CNT_TRIS_MIS = 10,
    MI_CNT_NODES_HOT_CLOSEST = 11, /* the part of NODES_CLOSEST / NODES_ANY / NODES_MIS served from the traversal blocks' LDS copy of the scene's hot nodes */
    MI_CNT_NODES_HOT_ANY = 12,
    MI_CNT_NODES_HOT_MIS = 13,
    MI_CNT_FILM_GATHER_BUILDS = 14, /* times mi_film_gather had to (re)build this context's reach list and exchange buffers (host-side, cumulative, not reset by
                                     * mi_counters_reset): 1 after any number of frames of one sharding */
    MI_CNT_TRACE_GUARD_TRIPS = 15, /* waves that hit the non-termination guard of the traversal kernels: must stay 0 */
    MI_CNT_COUNT = 16
};
int mi_counters(mi_ctx *ctx, uint64_t out[MI_CNT_COUNT]);
int mi_counters_reset(mi_ctx *ctx);

/* Per-kernel device time (HIP events recorded on the ctx stream around every launch while
 * enabled).  Kernel ids: */
enum mi_kernel_id {
    MI_K_RAYGEN = 0, MI_K_CLOSEST = 1, MI_K_SORT = 2, MI_K_SHADE = 3, MI_K_ANYHIT = 4,
    MI_K_MIS_CLOSEST = 5, MI_K_FILM = 6, MI_K_COUNT = 8
};
int mi_timing_enable(mi_ctx *ctx, int on);
int mi_timing_get(mi_ctx *ctx, double ms_total[MI_K_COUNT], uint64_t launches[MI_K_COUNT]);
/* Host-only self check of the BVH2 -> BVH4 collapse mi_scene_upload performs (no device needed): builds the device tree from
 * scene->bvh_nodes and verifies that every primitive lies in exactly one leaf range, that leaf references hold 1..16 primitives, that
 * every child box equals the box of the reference node it stands for and contains the primitives below it, and that the stack bound
 * covers the depth.  stats: [0] BVH4 nodes, [1] leaf references, [2] depth, [3] stack entries needed, [4] primitives covered.
 * Returns 0 if all invariants hold (else -1 with mi_last_error()). */
int mi_bvh4_validate(const mi_scene_desc *scene, int64_t stats[8]);

/* Stage-level check of the texture path (row f2): Texture<T>::Evaluate(const SurfaceInteraction&) (core/texture.h:139-144) of
 * node `node` of the uploaded scene's texture table at n recorded interactions -> 3 floats each (Float textures: the value in
 * all three).  Only what textures read of the interaction is passed (core/texture.cpp:84-153). */
typedef struct mi_tex_query {
    float p[3];
    float uv[2];
    float dpdx[3], dpdy[3];
    float dudx, dvdx, dudy, dvdy;
} mi_tex_query;
int mi_texture_eval(mi_ctx *ctx, int32_t node, const mi_tex_query *queries, int64_t n, float *rgb_out);
/* Measurement aid: achievable HBM read rate on this device -- a streaming 16-byte-per-lane read of
 * `bytes` (>= 1 GiB recommended, beyond the 256 MiB Infinity Cache), best of 3 timed launches -- reported beside the
 * 8 TB/s specification peak. */
int mi_stream_read_gbps(mi_ctx *ctx, uint64_t bytes, double *gbps);
/* Measurement aid (bench.py's `request_rate` object): the rate at which this device serves chains of DEPENDENT random 64-byte record fetches
 * (loads_per_record x 16-byte loads per record, per lane, traversal launch shape, no arithmetic) from a buffer of `bytes` bytes, in 1e9 lane
 * requests per second -- the memory-side ceiling of a BVH interior step (the traversal kernels are bound by the request rate of incoherent
 * 16-byte loads, not by HBM bandwidth: DESIGN.md s.5). */
int mi_gather_rate(mi_ctx *ctx, uint64_t bytes, int loads_per_record, double *grequests_per_s);
/* Stage-level check of mi_film_gather's RCCL step on a ONE-GPU box (RCCL refuses two ranks on one device, so the gather of two contexts sharing a
 * GPU never reaches it): loads librccl.so the way the gather does, makes the communicator of {ctx's device} and moves n_pixels packed FilmTilePixels
 * (float4) through ONE group of ncclSend / ncclRecv -- the same helper the gather calls, rank 0 to itself -- on the context's stream, then adds them into
 * a zeroed film with the gather's own kernel.  Returns 0 when every pixel arrived bit for bit; -1 with mi_last_error otherwise. */
int mi_rccl_probe(mi_ctx *ctx, int64_t n_pixels);
/* Measurement aid (bench.py's `roofline.valu_issue`): the shader clock the traversal kernels really ran at during the COUNTING passes since
 * the last mi_counters_reset -- every wave stamps s_memtime (shader-clock ticks) and s_memrealtime (constant-rate ticks) when it starts and when it
 * ends.  out[0] = GHz inside the closest-hit launches, out[1] = GHz inside the any-hit launches (0: no counting pass ran), out[2], out[3] = the
 * summed wave lifetimes in shader cycles (closest, any). */
int mi_trace_clock(mi_ctx *ctx, double out[4]);

/* ---- stage-level entry points (the same kernels, exposed for ray-by-ray parity tests) ---- */

typedef struct mi_ray { float o[3]; float tmax; float d[3]; float time; } mi_ray;  /* geometry.h:869-890 */
typedef struct mi_hit { int32_t prim; float t; float b0, b1, b2; float n[3]; } mi_hit; /* prim -1 = miss */

/* BVHAccel::Intersect (bvh.cpp:662-700) + Triangle::Intersect (triangle.cpp:188-425) */
int mi_intersect(mi_ctx *ctx, const mi_ray *rays, int64_t n, mi_hit *hits);
/* Triangle::Intersect alone (triangle.cpp:188-425) for n independent (triangle, ray) pairs, no scene needed:
 * tri9 = 9 floats per triangle (p0 p1 p2); replays the reference's Triangle.* unit-test vectors on the device. */
int mi_triangle_intersect(int device_ordinal, const float *tri9, const mi_ray *rays, int64_t n, mi_hit *hits);
/* BVHAccel::IntersectP (bvh.cpp:702-738) */
int mi_intersect_p(mi_ctx *ctx, const mi_ray *rays, int64_t n, uint8_t *occluded);

/* Host-only check of the 64-byte quantised BVH4 the default traversal kernels walk (csrc/pt_bvh4q.h): collapses the reference's BVH2, quantises
 * the child boxes, checks every quantised box against its reference box in exact arithmetic, and runs the kernel's per-ray state machine ON THE
 * HOST for `rays` (closest hit, or first hit found when any_hit != 0).  hits (may be NULL): prim / t / barycentrics as mi_intersect reports them.
 * stats: [0] nodes, [1] leaf references, [2] depth, [3] deepest stack seen, [4] primitives covered, [5] nodes visited, [6] primitives tested,
 * [7] rays that hit.  No GPU needed. */
int mi_bvh4q_validate(const mi_scene_desc *scene, const mi_ray *rays, int64_t n, int any_hit, mi_hit *hits, int64_t stats[8]);
/* Sphere::Intersect (shapes/sphere.cpp:48-162) of ray i against spheres[i] (explicit records, no scene): hit flag, tHit and the
 * world-space interaction's p, pError, n -- for the FullSphere / PartialSphere reintersection vectors of the reference's tests */
typedef struct mi_sphere_hit { int32_t hit; float t; float p[3], p_error[3], n[3]; } mi_sphere_hit;
int mi_sphere_intersect(int device_ordinal, const mi_sphere *spheres, const mi_ray *rays, int64_t n, mi_sphere_hit *hits);
/* SobolSampler: GetIndexForSample + SampleDimension (sobol.cpp:42-59) for pixel (px,py),
 * sample numbers [0,n_samples), dimensions [0,n_dims); out[s*n_dims+d]; index_out[s] */
int mi_sobol(mi_ctx *ctx, int px, int py, int n_samples, int n_dims, float *out, uint64_t *index_out);
/* Sampler::GetCameraSample + PerspectiveCamera::GenerateRayDifferential (sampler.cpp:46-52,
 * perspective.cpp:95-144) for n (pixel, sample) pairs */
int mi_camera_rays(mi_ctx *ctx, const int32_t *pixels_xy, const int32_t *sample_num, int64_t n,
                   mi_ray *rays, float *p_film /* 2*n */);
/* The offset rays of the same camera samples as the shading kernels rebuild them at the first hit of a textured scene:
 * PerspectiveCamera::GenerateRayDifferential's rx / ry (perspective.cpp:118-141) after RayDifferential::ScaleDifferentials(1 / sqrt(spp))
 * (integrator.cpp:262-263).  diffs[12 i ..] = rxOrigin, rxDirection, ryOrigin, ryDirection of pair i */
int mi_camera_differentials(mi_ctx *ctx, const int32_t *pixels_xy, const int32_t *sample_num, int64_t n, float *diffs /* 12*n */);
/* PathIntegrator::Li per camera sample, before the film: radiance rgb for n (pixel,sample) pairs */
int mi_li(mi_ctx *ctx, const int32_t *pixels_xy, const int32_t *sample_num, int64_t n, float *L_rgb);

#ifdef __cplusplus
}
#endif
#endif /* PBRT_AMD_H */
