#include "paramset.h"

#include <cstdarg>

namespace pbrt_amd {

int g_errorCount = 0, g_unsupportedCount = 0;
bool g_renderFailed = false;
bool g_quiet = false;
extern std::string CurrentParserLocation();   // parser.cpp

static void report(const char *tag, const char *fmt, va_list args) {   // error.cpp:62-102 format
    char buf[2048];
    vsnprintf(buf, sizeof(buf), fmt, args);
    std::string loc = CurrentParserLocation();
    std::fprintf(stderr, "%s%s: %s\n", loc.c_str(), tag, buf);
}
void Warning(const char *fmt, ...) {
    if (g_quiet) return;
    va_list a; va_start(a, fmt); report("Warning", fmt, a); va_end(a);
}
void Error(const char *fmt, ...) {
    ++g_errorCount;
    va_list a; va_start(a, fmt); report("Error", fmt, a); va_end(a);
}

void Unsupported(const char *fmt, ...) {
    ++g_errorCount; ++g_unsupportedCount;
    va_list a; va_start(a, fmt); report("Error (unsupported, the scene will not be rendered)", fmt, a); va_end(a);
}

void ParamSet::Add(Item item) {
    for (auto &it : items_)
        if (it.type == item.type && it.name == item.name) {   // paramset.cpp Add*: replace + warn
            Warning("%s redefined", item.name.c_str());
            it = std::move(item);
            return;
        }
    items_.push_back(std::move(item));
}

const ParamSet::Item *ParamSet::find(ParamType t, const std::string &n, bool mark) const {
    for (auto &it : items_)
        if (it.type == t && it.name == n) {
            if (mark) it.lookedUp = true;
            return &it;
        }
    return nullptr;
}

// FindOne*: value only if exactly one item was given (paramset.cpp LOOKUP_ONE)
Float ParamSet::FindOneFloat(const std::string &n, Float d) const {
    const Item *it = find(ParamType::Float, n);
    return (it && it->f.size() == 1) ? it->f[0] : d;
}
int ParamSet::FindOneInt(const std::string &n, int d) const {
    const Item *it = find(ParamType::Int, n);
    return (it && it->i.size() == 1) ? it->i[0] : d;
}
bool ParamSet::FindOneBool(const std::string &n, bool d) const {
    const Item *it = find(ParamType::Bool, n);
    return (it && it->i.size() == 1) ? it->i[0] != 0 : d;
}
std::string ParamSet::FindOneString(const std::string &n, const std::string &d) const {
    const Item *it = find(ParamType::String, n);
    return (it && it->s.size() == 1) ? it->s[0] : d;
}
extern std::string AbsolutePathFromScene(const std::string &f);   // parser.cpp (fileutil.cpp:AbsolutePath/ResolveFilename)
std::string ParamSet::FindOneFilename(const std::string &n, const std::string &d) const {
    std::string f = FindOneString(n, "");
    if (f == "") return d;
    return AbsolutePathFromScene(f);
}
Vec3 ParamSet::FindOnePoint3(const std::string &n, const Vec3 &d) const {
    const Item *it = find(ParamType::Point3, n);
    return (it && it->f.size() == 3) ? Vec3(it->f[0], it->f[1], it->f[2]) : d;
}
Vec3 ParamSet::FindOneVector3(const std::string &n, const Vec3 &d) const {
    const Item *it = find(ParamType::Vector3, n);
    return (it && it->f.size() == 3) ? Vec3(it->f[0], it->f[1], it->f[2]) : d;
}
RGB ParamSet::FindOneSpectrum(const std::string &n, const RGB &d) const {
    const Item *it = find(ParamType::Spectrum, n);
    return (it && it->f.size() == 3) ? RGB(it->f[0], it->f[1], it->f[2]) : d;
}
std::string ParamSet::FindTexture(const std::string &n) const {
    const Item *it = find(ParamType::Texture, n);
    return (it && it->s.size() == 1) ? it->s[0] : "";
}
#define FIND_ARRAY(fn, T, member, div)                                        \
    const T *ParamSet::fn(const std::string &n, int *count) const {           \
        const Item *it = find(ParamType::div##_t, n);                         \
        if (!it) { if (count) *count = 0; return nullptr; }                   \
        if (count) *count = (int)(it->member.size() / div##_n);               \
        return it->member.data();                                             \
    }
static const int Float_n = 1, Int_n = 1, Point2_n = 2, Point3_n = 3, Vector3_n = 3, Normal_n = 3, Spectrum_n = 3;
#define Float_t Float
#define Int_t Int
#define Point2_t Point2
#define Point3_t Point3
#define Vector3_t Vector3
#define Normal_t Normal
#define Spectrum_t Spectrum
FIND_ARRAY(FindFloat, Float, f, Float)
FIND_ARRAY(FindInt, int, i, Int)
FIND_ARRAY(FindPoint2, Float, f, Point2)
FIND_ARRAY(FindPoint3, Float, f, Point3)
FIND_ARRAY(FindVector3, Float, f, Vector3)
FIND_ARRAY(FindNormal3, Float, f, Normal)
FIND_ARRAY(FindSpectrum, Float, f, Spectrum)

void ParamSet::ReportUnused() const {   // paramset.cpp:443-457
    for (auto &it : items_)
        if (!it.lookedUp) Warning("Parameter \"%s\" not used", it.name.c_str());
}

// paramset.cpp:729-775 / :779-836 lookup order: shape texture, shape value, material texture, material value
int ConstantTextureNode(bool spectrum, const RGB &v);   // host/texture.cpp
int TextureParams::GetSpectrumTextureOrNull(const std::string &n) const {
    std::string name = geom_.FindTexture(n);
    if (name.empty()) {
        int cnt;
        const Float *s = geom_.FindSpectrum(n, &cnt);
        if (s) {
            if (cnt > 1) Warning("Ignoring excess values provided with parameter \"%s\"", n.c_str());
            return ConstantTextureNode(true, RGB(s[0], s[1], s[2]));
        }
        name = mat_.FindTexture(n);
        if (name.empty()) {
            s = mat_.FindSpectrum(n, &cnt);
            if (s) {
                if (cnt > 1) Warning("Ignoring excess values provided with parameter \"%s\"", n.c_str());
                return ConstantTextureNode(true, RGB(s[0], s[1], s[2]));
            }
            return -1;
        }
    }
    auto it = tex_.spectra.find(name);
    if (it != tex_.spectra.end()) return it->second;
    Error("Couldn't find spectrum texture named \"%s\" for parameter \"%s\"", name.c_str(), n.c_str());
    return -1;
}
int TextureParams::GetSpectrumTexture(const std::string &n, const RGB &def) const {
    int t = GetSpectrumTextureOrNull(n);
    return t >= 0 ? t : ConstantTextureNode(true, def);
}
int TextureParams::GetFloatTextureOrNull(const std::string &n) const {
    std::string name = geom_.FindTexture(n);
    if (name.empty()) {
        int cnt;
        const Float *s = geom_.FindFloat(n, &cnt);
        if (s) {
            if (cnt > 1) Warning("Ignoring excess values provided with parameter \"%s\"", n.c_str());
            return ConstantTextureNode(false, RGB(*s));
        }
        name = mat_.FindTexture(n);
        if (name.empty()) {
            s = mat_.FindFloat(n, &cnt);
            if (s) {
                if (cnt > 1) Warning("Ignoring excess values provided with parameter \"%s\"", n.c_str());
                return ConstantTextureNode(false, RGB(*s));
            }
            return -1;
        }
    }
    auto it = tex_.floats.find(name);
    if (it != tex_.floats.end()) return it->second;
    Error("Couldn't find float texture named \"%s\" for parameter \"%s\"", name.c_str(), n.c_str());
    return -1;
}
int TextureParams::GetFloatTexture(const std::string &n, Float def) const {
    int t = GetFloatTextureOrNull(n);
    return t >= 0 ? t : ConstantTextureNode(false, RGB(def));
}

}  // namespace pbrt_amd
