// Typed name->array parameter bag of the .pbrt scene format.
// Interface mirrors the reference's ParamSet / TextureParams (core/paramset.h:57-118,173-217):
// same Find*/FindOne* names, (type,name) keyed lookup, "looked up" tracking and ReportUnused().
// Spectrum == RGB (core/spectrum.h:429): "rgb"/"color" map through unchanged, "xyz" through
// XYZToRGB (spectrum.h:56-60), "blackbody" / "spectrum" (inline samples or SPD files) through the
// CIE integration of RGBSpectrum::FromSampled (host/spectrum.cpp).
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "geom.h"

namespace pbrt_amd {

void Warning(const char *fmt, ...);
void Error(const char *fmt, ...);
// Something the reference renders and this host does not restate: reported like an
// Error AND counted, so that a render which would come out plausible but wrong is refused -- pbrtWorldEnd skips Render,
// pbrt_amd_scene_load returns NULL and the command-line renderer exits non-zero (the reference itself would go on).
void Unsupported(const char *fmt, ...);
extern int g_errorCount, g_unsupportedCount;
extern bool g_renderFailed;   // the device refused / failed a render: the command-line renderer exits non-zero
extern bool g_quiet;

struct RGB {
    Float c[3];
    RGB(Float v = 0) { c[0] = c[1] = c[2] = v; }
    RGB(Float r, Float g, Float b) { c[0] = r; c[1] = g; c[2] = b; }
    bool IsBlack() const { return c[0] == 0 && c[1] == 0 && c[2] == 0; }
    RGB Clamp(Float lo = 0, Float hi = kInfinity) const {   // spectrum.h CoefficientSpectrum::Clamp
        return RGB(pbrt_amd::Clamp(c[0], lo, hi), pbrt_amd::Clamp(c[1], lo, hi), pbrt_amd::Clamp(c[2], lo, hi));
    }
    RGB operator*(const RGB &o) const { return RGB(c[0] * o.c[0], c[1] * o.c[1], c[2] * o.c[2]); }
    RGB operator*(Float s) const { return RGB(c[0] * s, c[1] * s, c[2] * s); }
    RGB operator+(const RGB &o) const { return RGB(c[0] + o.c[0], c[1] + o.c[1], c[2] + o.c[2]); }
    RGB operator-(const RGB &o) const { return RGB(c[0] - o.c[0], c[1] - o.c[1], c[2] - o.c[2]); }
    RGB operator-() const { return RGB(-c[0], -c[1], -c[2]); }
    Float y() const { return 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2]; }   // spectrum.h:462-465
};

// host/spectrum.cpp: sampled spectra -> RGB as the reference's parser does it (paramset.cpp:134-205, spectrum.h:466-489)
RGB SpectrumFromSampled(const Float *lambda, const Float *v, int n);
RGB BlackbodyRGB(Float T, Float scale);
RGB SpectrumFromFile(const std::string &absoluteFilename);
bool ReadFloatFile(const char *filename, std::vector<Float> *values);

enum class ParamType { Int, Bool, Float, Point2, Vector2, Point3, Vector3, Normal, Spectrum, String, Texture };

class ParamSet {
  public:
    struct Item {
        ParamType type;
        std::string name;
        std::vector<Float> f;   // numeric payload (ints stored exactly: |v| < 2^24 checked at parse)
        std::vector<int> i;
        std::vector<std::string> s;
        mutable bool lookedUp = false;
    };
    void Add(Item item);
    bool Empty() const { return items_.empty(); }

    Float FindOneFloat(const std::string &n, Float d) const;
    int FindOneInt(const std::string &n, int d) const;
    bool FindOneBool(const std::string &n, bool d) const;
    std::string FindOneString(const std::string &n, const std::string &d) const;
    std::string FindOneFilename(const std::string &n, const std::string &d) const;
    Vec3 FindOnePoint3(const std::string &n, const Vec3 &d) const;
    Vec3 FindOneVector3(const std::string &n, const Vec3 &d) const;
    RGB FindOneSpectrum(const std::string &n, const RGB &d) const;
    std::string FindTexture(const std::string &n) const;
    const Float *FindFloat(const std::string &n, int *count) const;
    const int *FindInt(const std::string &n, int *count) const;
    const Float *FindPoint2(const std::string &n, int *count) const;    // count = #points
    const Float *FindPoint3(const std::string &n, int *count) const;
    const Float *FindVector3(const std::string &n, int *count) const;
    const Float *FindNormal3(const std::string &n, int *count) const;
    const Float *FindSpectrum(const std::string &n, int *count) const;
    bool Has(ParamType t, const std::string &n) const { return find(t, n, false) != nullptr; }
    void ReportUnused() const;
    const std::vector<Item> &items() const { return items_; }

  private:
    const Item *find(ParamType t, const std::string &n, bool mark = true) const;
    std::vector<Item> items_;
};

// Named textures of the graphics state (api.cpp:224-230): name -> node of the parse's TextureStore (host/texture.cpp).
struct TextureMaps {
    std::map<std::string, int> floats;
    std::map<std::string, int> spectra;
};

class TextureParams {   // core/paramset.h:173-217: geometry params shadow material params
  public:
    TextureParams(const ParamSet &geom, const ParamSet &mat, const TextureMaps &tex)
        : geom_(geom), mat_(mat), tex_(tex) {}
    // texture nodes (paramset.cpp:720-836): a named texture, or a constant node made from the inline value / default;
    // the ...OrNull form returns -1 where the reference returns nullptr
    int GetSpectrumTexture(const std::string &n, const RGB &def) const;
    int GetSpectrumTextureOrNull(const std::string &n) const;
    int GetFloatTexture(const std::string &n, Float def) const;
    int GetFloatTextureOrNull(const std::string &n) const;
    Float FindFloat(const std::string &n, Float d) const { return geom_.FindOneFloat(n, mat_.FindOneFloat(n, d)); }
    int FindInt(const std::string &n, int d) const { return geom_.FindOneInt(n, mat_.FindOneInt(n, d)); }
    bool FindBool(const std::string &n, bool d) const { return geom_.FindOneBool(n, mat_.FindOneBool(n, d)); }
    std::string FindString(const std::string &n, const std::string &d = "") const {
        return geom_.FindOneString(n, mat_.FindOneString(n, d));
    }
    std::string FindFilename(const std::string &n, const std::string &d = "") const {
        return geom_.FindOneFilename(n, mat_.FindOneFilename(n, d));
    }
    Vec3 FindVector3f(const std::string &n, const Vec3 &d) const { return geom_.FindOneVector3(n, mat_.FindOneVector3(n, d)); }
    RGB FindSpectrum(const std::string &n, const RGB &d) const { return geom_.FindOneSpectrum(n, mat_.FindOneSpectrum(n, d)); }
    void ReportUnused() const { geom_.ReportUnused(); mat_.ReportUnused(); }
    const ParamSet &geom() const { return geom_; }
    const ParamSet &mat() const { return mat_; }
    const TextureMaps &tex() const { return tex_; }

  private:
    const ParamSet &geom_, &mat_;
    const TextureMaps &tex_;
};

}  // namespace pbrt_amd
