// One scene build per NODE instead of one per rank (SURVEY.md s.8 row e).
//
// The tile-sharded render replicates the scene on every GPU: each rank needs the same mi_scene_desc.  Parsing the scene and building the
// reference's BVH once per rank costs N times the memory traffic of one build on the shared host (measured: eight concurrent loads of the
// 10 M-triangle frame take 4.6 x as long as one).  Instead the rank that built the scene writes its mi_scene_desc -- every array it points
// to, nested ones included -- into ONE file (pointers stored as offsets), and the other ranks of the node map that file: the big arrays
// are page-cache pages shared by all of them, only the few pages that hold pointers are touched (private copy-on-write mapping) when the
// offsets are turned back into addresses.  The mapped description is byte for byte what mi_scene_upload receives from the builder.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "capi.h"

using namespace pbrt_amd;

namespace {
const char kMagic[8] = {'P', 'B', 'R', 'T', 'A', 'M', 'D', 'B'};
struct BlobHeader {
    char magic[8];
    uint32_t abi, headerBytes;
    uint64_t totalBytes, descOffset;
    uint64_t nFixups;       // offsets (from the file start) of every pointer slot that holds an offset
    uint64_t fixupOffset;
};

// collects the arrays; a pointer slot is registered by its position relative to the struct array it lives in
struct Writer {
    std::vector<char> head;                                  // header + desc + nested struct arrays + fix-up table: small
    struct Chunk { const void *src; uint64_t bytes, at; };
    std::vector<Chunk> chunks;                               // the bulk arrays, written straight from the scene's memory
    std::vector<uint64_t> fixups;
    uint64_t cursor = 0;                                     // next free offset of the bulk region (relative to bulkBase, resolved at the end)
    static uint64_t align(uint64_t v) { return (v + 63) & ~uint64_t(63); }
    uint64_t addHead(const void *p, size_t n) { uint64_t at = align(head.size()); head.resize(at + n); if (n) std::memcpy(head.data() + at, p, n); return at; }
    // bulk array: returns a provisional offset tagged with the top bit (rebased once the head size is known)
    uint64_t addBulk(const void *p, uint64_t n) {
        if (!p || !n) return 0;
        uint64_t at = align(cursor);
        chunks.push_back({p, n, at});
        cursor = at + n;
        return at | (uint64_t(1) << 63);
    }
};
template <class T> void setSlot(Writer &w, uint64_t structAt, size_t memberOff, uint64_t value) {
    std::memcpy(w.head.data() + structAt + memberOff, &value, 8);
    if (value) w.fixups.push_back(structAt + memberOff);
}
#define SLOT(base, type, member, value) setSlot<type>(w, base, offsetof(type, member), value)
}  // namespace

extern "C" {

int pbrt_amd_scene_save_blob(pbrt_amd_scene *s, const char *path) {
    if (!s || !s->flat || !path) return -1;
    const mi_scene_desc &d = s->flat->desc;
    Writer w;
    BlobHeader hdr;
    std::memset(&hdr, 0, sizeof(hdr));
    w.addHead(&hdr, sizeof(hdr));
    const uint64_t descAt = w.addHead(&d, sizeof(d));
    // total lengths of the per-primitive / node arrays (two-level scenes keep the objects' primitives and nodes behind the top level's)
    uint64_t nNodes = d.n_bvh_nodes;
    for (uint32_t o = 0; o < d.n_objects; ++o) nNodes = std::max<uint64_t>(nNodes, (uint64_t)d.objects[o].first_node + d.objects[o].n_nodes);
    SLOT(descAt, mi_scene_desc, P, w.addBulk(d.P, (uint64_t)d.n_verts * 12));
    SLOT(descAt, mi_scene_desc, N, w.addBulk(d.N, (uint64_t)d.n_verts * 12));
    SLOT(descAt, mi_scene_desc, UV, w.addBulk(d.UV, (uint64_t)d.n_verts * 8));
    SLOT(descAt, mi_scene_desc, tri_indices, w.addBulk(d.tri_indices, (uint64_t)d.n_tris * 12));
    SLOT(descAt, mi_scene_desc, tri_mesh, w.addBulk(d.tri_mesh, (uint64_t)d.n_tris * 4));
    SLOT(descAt, mi_scene_desc, tri_light, w.addBulk(d.tri_light, (uint64_t)d.n_tris * 4));
    SLOT(descAt, mi_scene_desc, meshes, w.addBulk(d.meshes, (uint64_t)d.n_meshes * sizeof(mi_mesh)));
    SLOT(descAt, mi_scene_desc, bvh_nodes, w.addBulk(d.bvh_nodes, nNodes * sizeof(mi_bvh2_node)));
    SLOT(descAt, mi_scene_desc, materials, w.addBulk(d.materials, (uint64_t)d.n_materials * sizeof(mi_material)));
    SLOT(descAt, mi_scene_desc, lights, w.addBulk(d.lights, (uint64_t)d.n_lights * sizeof(mi_light)));
    SLOT(descAt, mi_scene_desc, light_func, w.addBulk(d.light_func, (uint64_t)d.n_lights * 4));
    SLOT(descAt, mi_scene_desc, light_cdf, w.addBulk(d.light_cdf, d.light_cdf ? ((uint64_t)d.n_lights + 1) * 4 : 0));
    SLOT(descAt, mi_scene_desc, spheres, w.addBulk(d.spheres, (uint64_t)d.n_spheres * sizeof(mi_sphere)));
    SLOT(descAt, mi_scene_desc, textures, w.addBulk(d.textures, (uint64_t)d.n_textures * sizeof(mi_texture)));
    SLOT(descAt, mi_scene_desc, material_descs, w.addBulk(d.material_descs, d.material_descs ? (uint64_t)d.n_materials * sizeof(mi_material_desc) : 0));
    SLOT(descAt, mi_scene_desc, mesh_alpha, w.addBulk(d.mesh_alpha, d.mesh_alpha ? (uint64_t)d.n_meshes * 8 : 0));
    SLOT(descAt, mi_scene_desc, instances, w.addBulk(d.instances, (uint64_t)d.n_instances * sizeof(mi_instance)));
    SLOT(descAt, mi_scene_desc, objects, w.addBulk(d.objects, (uint64_t)d.n_objects * sizeof(mi_object)));
    SLOT(descAt, mi_scene_desc, mesh_medium, w.addBulk(d.mesh_medium, d.mesh_medium ? (uint64_t)d.n_meshes * 8 : 0));
    SLOT(descAt, mi_scene_desc, material_bssrdf, w.addBulk(d.material_bssrdf, d.material_bssrdf ? (uint64_t)d.n_materials * sizeof(mi_bssrdf_desc) : 0));
    // arrays of structs that hold pointers themselves: the structs go to the head (their slots are fixed up), their payload to the bulk
    if (d.n_envmaps && d.envmaps) {
        const uint64_t at = w.addHead(d.envmaps, (size_t)d.n_envmaps * sizeof(mi_envmap));
        setSlot<mi_scene_desc>(w, descAt, offsetof(mi_scene_desc, envmaps), at);
        for (uint32_t i = 0; i < d.n_envmaps; ++i) {
            const mi_envmap &e = d.envmaps[i];
            const uint64_t b = at + i * sizeof(mi_envmap), nu = 2 * (uint64_t)e.width, nv = 2 * (uint64_t)e.height;
            SLOT(b, mi_envmap, rgb, w.addBulk(e.rgb, 3 * (uint64_t)e.width * e.height * 4));
            SLOT(b, mi_envmap, cond_func, w.addBulk(e.cond_func, nu * nv * 4));
            SLOT(b, mi_envmap, cond_cdf, w.addBulk(e.cond_cdf, (nu + 1) * nv * 4));
            SLOT(b, mi_envmap, cond_func_int, w.addBulk(e.cond_func_int, nv * 4));
            SLOT(b, mi_envmap, marg_func, w.addBulk(e.marg_func, nv * 4));
            SLOT(b, mi_envmap, marg_cdf, w.addBulk(e.marg_cdf, (nv + 1) * 4));
        }
    }
    if (d.n_images && d.images) {
        const uint64_t at = w.addHead(d.images, (size_t)d.n_images * sizeof(mi_image));
        setSlot<mi_scene_desc>(w, descAt, offsetof(mi_scene_desc, images), at);
        for (uint32_t i = 0; i < d.n_images; ++i) {
            const mi_image &im = d.images[i];
            uint64_t texels = 0;
            for (int l = 0; l < im.levels; ++l) texels += (uint64_t)std::max(1, im.width >> l) * std::max(1, im.height >> l) * im.channels;
            SLOT(at + i * sizeof(mi_image), mi_image, texels, w.addBulk(im.texels, texels * 4));
        }
    }
    if (d.n_media && d.media) {
        const uint64_t at = w.addHead(d.media, (size_t)d.n_media * sizeof(mi_medium));
        setSlot<mi_scene_desc>(w, descAt, offsetof(mi_scene_desc, media), at);
        for (uint32_t i = 0; i < d.n_media; ++i) {
            const mi_medium &m = d.media[i];
            SLOT(at + i * sizeof(mi_medium), mi_medium, density, w.addBulk(m.density, m.type == MI_MEDIUM_GRID ? (uint64_t)m.nx * m.ny * m.nz * 4 : 0));
        }
    }
    if (d.n_bssrdf_tables && d.bssrdf_tables) {
        const uint64_t at = w.addHead(d.bssrdf_tables, (size_t)d.n_bssrdf_tables * sizeof(mi_bssrdf_table));
        setSlot<mi_scene_desc>(w, descAt, offsetof(mi_scene_desc, bssrdf_tables), at);
        for (uint32_t i = 0; i < d.n_bssrdf_tables; ++i) {
            const mi_bssrdf_table &t = d.bssrdf_tables[i];
            const uint64_t b = at + i * sizeof(mi_bssrdf_table);
            SLOT(b, mi_bssrdf_table, rho_samples, w.addBulk(t.rho_samples, (uint64_t)t.n_rho * 4));
            SLOT(b, mi_bssrdf_table, radius_samples, w.addBulk(t.radius_samples, (uint64_t)t.n_radius * 4));
            SLOT(b, mi_bssrdf_table, profile, w.addBulk(t.profile, (uint64_t)t.n_rho * t.n_radius * 4));
            SLOT(b, mi_bssrdf_table, rho_eff, w.addBulk(t.rho_eff, (uint64_t)t.n_rho * 4));
            SLOT(b, mi_bssrdf_table, profile_cdf, w.addBulk(t.profile_cdf, (uint64_t)t.n_rho * t.n_radius * 4));
        }
    }
    // fix-up table, then the bulk region; bulk offsets are rebased now that the head is complete
    const uint64_t fixAt = Writer::align(w.head.size());
    const uint64_t bulkBase = Writer::align(fixAt + w.fixups.size() * 8 + 8);
    for (uint64_t f : w.fixups) {
        uint64_t v;
        std::memcpy(&v, w.head.data() + f, 8);
        if (v >> 63) { v = (v & ~(uint64_t(1) << 63)) + bulkBase; std::memcpy(w.head.data() + f, &v, 8); }
    }
    w.head.resize(bulkBase, 0);
    if (!w.fixups.empty()) std::memcpy(w.head.data() + fixAt, w.fixups.data(), w.fixups.size() * 8);
    std::memcpy(hdr.magic, kMagic, 8);
    hdr.abi = MI_ABI_VERSION; hdr.headerBytes = sizeof(hdr);
    hdr.totalBytes = bulkBase + Writer::align(w.cursor); hdr.descOffset = descAt;
    hdr.nFixups = w.fixups.size(); hdr.fixupOffset = fixAt;
    std::memcpy(w.head.data(), &hdr, sizeof(hdr));
    // write to a temporary name and publish with rename(): a reader never sees a partial blob
    const std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return -2;
    bool ok = std::fwrite(w.head.data(), 1, w.head.size(), f) == w.head.size();
    uint64_t pos = bulkBase;
    static const char zeros[64] = {0};
    for (const Writer::Chunk &c : w.chunks) {
        const uint64_t at = bulkBase + c.at;
        while (ok && pos < at) { size_t n = (size_t)std::min<uint64_t>(64, at - pos); ok = std::fwrite(zeros, 1, n, f) == n; pos += n; }
        ok = ok && std::fwrite(c.src, 1, c.bytes, f) == c.bytes;
        pos += c.bytes;
    }
    while (ok && pos < hdr.totalBytes) { size_t n = (size_t)std::min<uint64_t>(64, hdr.totalBytes - pos); ok = std::fwrite(zeros, 1, n, f) == n; pos += n; }
    ok = (std::fclose(f) == 0) && ok;
    if (!ok || std::rename(tmp.c_str(), path) != 0) { std::remove(tmp.c_str()); return -3; }
    return 0;
}

// Maps a blob written by pbrt_amd_scene_save_blob; the handle serves pbrt_amd_scene_desc / _info / _texture_info / _media_info / _light / _free
// (not the Film entry points: the rank that built the scene owns the Film).  NULL: missing file, foreign magic, other ABI version, short file.
pbrt_amd_scene *pbrt_amd_scene_map_blob(const char *path) {
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) return nullptr;
    struct stat st;
    BlobHeader hdr;
    if (::fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(hdr) || ::pread(fd, &hdr, sizeof(hdr), 0) != (ssize_t)sizeof(hdr) || std::memcmp(hdr.magic, kMagic, 8) != 0 ||
        hdr.abi != MI_ABI_VERSION || hdr.headerBytes != sizeof(hdr) || hdr.totalBytes > (uint64_t)st.st_size || hdr.descOffset > hdr.totalBytes || sizeof(mi_scene_desc) > hdr.totalBytes - hdr.descOffset ||
        hdr.fixupOffset > hdr.totalBytes || hdr.nFixups > (hdr.totalBytes - hdr.fixupOffset) / 8) {   // (no sum or product that could wrap)
        ::close(fd);
        return nullptr;
    }
    void *m = ::mmap(nullptr, (size_t)hdr.totalBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE, fd, 0);   // private: the fix-ups below stay in this process
    ::close(fd);
    if (m == MAP_FAILED) return nullptr;
    char *base = (char *)m;
    const uint64_t *fix = (const uint64_t *)(base + hdr.fixupOffset);
    for (uint64_t i = 0; i < hdr.nFixups; ++i) {
        if (fix[i] + 8 > hdr.totalBytes) { ::munmap(m, (size_t)hdr.totalBytes); return nullptr; }
        uint64_t off;
        std::memcpy(&off, base + fix[i], 8);
        if (off >= hdr.totalBytes) { ::munmap(m, (size_t)hdr.totalBytes); return nullptr; }
        const char *p = base + off;
        std::memcpy(base + fix[i], &p, 8);
    }
    pbrt_amd_scene *s = new pbrt_amd_scene;
    s->flat.reset(new FlatScene);
    std::memcpy(&s->flat->desc, base + hdr.descOffset, sizeof(mi_scene_desc));
    s->map = m; s->mapBytes = (size_t)hdr.totalBytes;
    return s;
}

}  // extern "C"
