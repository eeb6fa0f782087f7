// C entry points of libpbrt_amd_host.so: scene loading for Python tests / bench (ctypes) and
// for any other FFI user.  Pure host code: no HIP dependency, usable without a GPU.
#include "api.h"

using namespace pbrt_amd;

#include <sys/mman.h>

#include "capi.h"

extern "C" {

// Parse `filename` (or, if is_text != 0, the scene text itself) up to WorldEnd and flatten the
// result; returns NULL on failure.  quick/spp/res overrides <= 0 are ignored.
static pbrt_amd_scene *LoadScene(const char *filename_or_text, int is_text, int quiet, const char *outfile, const float *crop);
pbrt_amd_scene *pbrt_amd_scene_load(const char *filename_or_text, int is_text, int quiet, const char *outfile) {
    return LoadScene(filename_or_text, is_text, quiet, outfile, nullptr);
}
// the same with the command line's --cropwindow x0 x1 y0 y1 (main/pbrt.cpp:94-100: it overrides the Film's own "cropwindow")
pbrt_amd_scene *pbrt_amd_scene_load_crop(const char *filename_or_text, int is_text, int quiet, const char *outfile, const float crop[4]) {
    return LoadScene(filename_or_text, is_text, quiet, outfile, crop);
}
static pbrt_amd_scene *LoadScene(const char *filename_or_text, int is_text, int quiet, const char *outfile, const float *crop) {
    Options opt;
    opt.quiet = quiet != 0;
    opt.deferRender = true;
    if (outfile) opt.imageFile = outfile;
    if (crop) { opt.cropWindow[0][0] = crop[0]; opt.cropWindow[0][1] = crop[1]; opt.cropWindow[1][0] = crop[2]; opt.cropWindow[1][1] = crop[3]; }
    pbrtInit(opt);
    if (is_text) pbrtParseString(filename_or_text); else pbrtParseFile(filename_or_text);
    pbrtCleanup();
    std::unique_ptr<BuiltScene> built = pbrtTakeBuiltScene();
    if (!built) return nullptr;
    pbrt_amd_scene *s = new pbrt_amd_scene;
    s->flat = built->integrator->Flatten(*built->scene);
    s->built = std::move(built);
    return s;
}
void pbrt_amd_scene_free(pbrt_amd_scene *s) {
    if (s && s->map) { s->flat.reset(); ::munmap(s->map, s->mapBytes); }   // a mapped blob (host/blob.cpp)
    delete s;
}
const mi_scene_desc *pbrt_amd_scene_desc(pbrt_amd_scene *s) { return &s->flat->desc; }
int pbrt_amd_error_count() { return g_errorCount; }
// n_verts n_tris n_meshes n_bvh_nodes n_materials n_lights xres yres crop(x0 y0 x1 y1) spp max_depth sobol_res log2res
void pbrt_amd_scene_info(pbrt_amd_scene *s, int64_t *out) {
    const mi_scene_desc &d = s->flat->desc;
    int64_t v[16] = {d.n_verts, d.n_tris, d.n_meshes, d.n_bvh_nodes, d.n_materials, d.n_lights, d.film.full_res[0],
                     d.film.full_res[1], d.film.crop_min[0], d.film.crop_min[1], d.film.crop_max[0], d.film.crop_max[1],
                     d.integrator.spp, d.integrator.max_depth, d.integrator.sobol_resolution, d.integrator.sobol_log2_resolution};
    for (int i = 0; i < 16; ++i) out[i] = v[i];
}

// the film's filter radius (pixels) and sample bounds -- what a tile-sharded frame needs to know which pixels OUTSIDE a rank's tiles its samples reach
void pbrt_amd_scene_film_info(pbrt_amd_scene *s, double *out) {
    const mi_film &f = s->flat->desc.film;
    const double v[6] = {f.filter_radius[0], f.filter_radius[1], (double)f.sample_min[0], (double)f.sample_min[1], (double)f.sample_max[0], (double)f.sample_max[1]};
    for (int i = 0; i < 6; ++i) out[i] = v[i];
}

// ComputeBeamDiffusionBSSRDF as the host restates it (host/bssrdf.cpp): the 100 x 64 table of a Subsurface / KdSubsurface material for (g, eta).
// out: rho samples [100], radius samples [64], profile [6400], rhoEff [100], profileCDF [6400], in that order
int pbrt_amd_bssrdf_table(float g, float eta, float *out) {
    std::shared_ptr<BSSRDFTableData> t = MakeBSSRDFTable(g, eta);
    if (!t || t->nRho != 100 || t->nRadius != 64) return -1;
    float *p = out;
    for (const std::vector<float> *v : {&t->rhoSamples, &t->radiusSamples, &t->profile, &t->rhoEff, &t->profileCDF}) { std::copy(v->begin(), v->end(), p); p += v->size(); }
    return 0;
}

// light i of the flattened scene: type and the emitted quantity (Lemit | I | L) as it crosses the boundary; returns 0, -1 when out of range
int pbrt_amd_scene_light(pbrt_amd_scene *s, int i, int *type, float rgb[3]) {
    const mi_scene_desc &d = s->flat->desc;
    if (i < 0 || (uint32_t)i >= d.n_lights) return -1;
    *type = d.lights[i].type;
    for (int k = 0; k < 3; ++k) rgb[k] = d.lights[i].L[k];
    return 0;
}

// Film: merge a downloaded FilmTilePixel array (4 floats per cropped pixel) and produce the
// final RGB image exactly as Film::WriteImage would; optionally write it.
// (a scene mapped from a blob has no Film: the rank that built it owns the image -- these return -1 there)
int pbrt_amd_film_merge(pbrt_amd_scene *s, const float *rgbw) { if (!s->built) return -1; s->built->integrator->camera->film->MergeFilm(rgbw); return 0; }
int pbrt_amd_film_clear(pbrt_amd_scene *s) { if (!s->built) return -1; s->built->integrator->camera->film->Clear(); return 0; }
int pbrt_amd_film_rgb(pbrt_amd_scene *s, float *rgb_out) {
    if (!s->built) return -1;
    std::vector<Float> rgb = s->built->integrator->camera->film->FinalRGB();
    std::memcpy(rgb_out, rgb.data(), rgb.size() * sizeof(float));
    return 0;
}
int pbrt_amd_film_write(pbrt_amd_scene *s, const char *filename) {
    if (!s->built) return -1;
    Film &f = *s->built->integrator->camera->film;
    if (filename && filename[0]) f.filename = filename;
    f.WriteImage();
    return 0;
}
// number of GeometricPrimitives (meshes) of the scene, and a binary-little-endian PLY dump of one of them
// (world-space positions [+ normals]); used by tools/make_killeroo.py to ship subdivided geometry as plymesh
// texture nodes / images / textured materials / masked meshes of the flattened scene (row f2)
void pbrt_amd_scene_texture_info(pbrt_amd_scene *s, int64_t *out) {
    const mi_scene_desc &d = s->flat->desc;
    out[0] = d.n_textures; out[1] = d.n_images; out[2] = 0; out[3] = 0;
    if (d.material_descs) for (uint32_t m = 0; m < d.n_materials; ++m) out[2] += d.material_descs[m].textured != 0;
    if (d.mesh_alpha) for (uint32_t m = 0; m < d.n_meshes; ++m) out[3] += d.mesh_alpha[2 * m] >= 0 || d.mesh_alpha[2 * m + 1] >= 0;
}
// participating media of the flattened scene (row f4): media, meshes whose two sides differ, camera medium, integrator type
void pbrt_amd_scene_media_info(pbrt_amd_scene *s, int64_t *out) {
    const mi_scene_desc &d = s->flat->desc;
    out[0] = d.n_media; out[1] = 0; out[2] = d.camera_medium; out[3] = d.integrator_type;
    if (d.mesh_medium) for (uint32_t m = 0; m < d.n_meshes; ++m) out[1] += d.mesh_medium[2 * m] != d.mesh_medium[2 * m + 1];
}
int pbrt_amd_scene_num_prims(pbrt_amd_scene *s) { return s->built ? (int)s->built->scene->primitives.size() : 0; }
int pbrt_amd_scene_write_ply(pbrt_amd_scene *s, int prim, const char *filename) {
    if (!s->built || prim < 0 || prim >= (int)s->built->scene->primitives.size()) return -1;
    const TriangleMesh &m = *s->built->scene->primitives[prim].shape;
    FILE *f = std::fopen(filename, "wb");
    if (!f) return -2;
    bool hasN = !m.n.empty();
    std::fprintf(f, "ply\nformat binary_little_endian 1.0\nelement vertex %zu\nproperty float x\nproperty float y\nproperty float z\n", m.p.size());
    if (hasN) std::fprintf(f, "property float nx\nproperty float ny\nproperty float nz\n");
    std::fprintf(f, "element face %d\nproperty list uchar int vertex_indices\nend_header\n", m.nTriangles());
    for (size_t i = 0; i < m.p.size(); ++i) {
        float v[6] = {m.p[i].x, m.p[i].y, m.p[i].z, 0, 0, 0};
        if (hasN) { v[3] = m.n[i].x; v[4] = m.n[i].y; v[5] = m.n[i].z; }
        std::fwrite(v, 4, hasN ? 6 : 3, f);
    }
    for (int t = 0; t < m.nTriangles(); ++t) {
        unsigned char n = 3;
        std::fwrite(&n, 1, 1, f);
        std::fwrite(&m.indices[3 * t], 4, 3, f);
    }
    std::fclose(f);
    return 0;
}
// ReadImage (core/imageio.cpp:60-79) by extension: .pfm, .png, .tga -> RGB floats, row 0 = top
int pbrt_amd_read_image(const char *filename, float *rgb, int capacity_floats, int *w, int *h) {
    std::vector<Float> v;
    if (!ReadImage(filename, &v, w, h)) return -1;
    if ((int)v.size() > capacity_floats) return -2;
    std::memcpy(rgb, v.data(), v.size() * sizeof(float));
    return 0;
}
int pbrt_amd_read_pfm(const char *filename, float *rgb, int capacity_floats, int *w, int *h) {
    std::vector<Float> v;
    if (!ReadImagePFM(filename, &v, w, h)) return -1;
    if ((int)v.size() > capacity_floats) return -2;
    std::memcpy(rgb, v.data(), v.size() * sizeof(float));
    return 0;
}
}
