// .pbrt tokenizer and directive parser.  Grammar and behaviour follow the reference's
// hand-written parser (core/parser.cpp): '#' comments, quoted strings with \-escapes
// (:98-318), numbers through strtol/strtof (:320-366), "type name" parameter declarations with
// bracketed or bare values (:709-779), the directive switch (:868-1086) and the Include file
// stack with paths relative to the main scene file's directory (:785-840, fileutil.cpp).
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "api.h"

namespace pbrt_amd {
std::string AbsolutePathFromScene(const std::string &f);

namespace {
struct Loc { std::string filename; int line = 1; };
Loc *g_loc = nullptr;
std::string g_searchDirectory;

struct Tokenizer {
    std::string text;
    size_t pos = 0;
    Loc loc;
    bool hasUnget = false;
    std::string ungot;

    static bool fromFile(const std::string &fn, Tokenizer *t) {
        std::ifstream in(fn, std::ios::binary);
        if (!in) return false;
        std::stringstream ss;
        ss << in.rdbuf();
        t->text = ss.str();
        t->loc.filename = fn;
        return true;
    }
    // returns false at EOF
    bool next(std::string *tok) {
        if (hasUnget) { hasUnget = false; *tok = ungot; return true; }
        while (pos < text.size()) {
            char c = text[pos];
            if (c == '\n') { ++loc.line; ++pos; }
            else if (c == ' ' || c == '\t' || c == '\r') ++pos;
            else if (c == '#') { while (pos < text.size() && text[pos] != '\n') ++pos; }
            else break;
        }
        if (pos >= text.size()) return false;
        char c = text[pos];
        if (c == '"') {
            size_t start = pos++;
            std::string out = "\"";
            bool closed = false;
            while (pos < text.size()) {
                char ch = text[pos++];
                if (ch == '\n') { Error("Unterminated string"); ++loc.line; break; }
                if (ch == '\\' && pos < text.size()) {
                    char e = text[pos++];
                    switch (e) {
                    case 'b': out += '\b'; break; case 'f': out += '\f'; break; case 'n': out += '\n'; break;
                    case 'r': out += '\r'; break; case 't': out += '\t'; break; case '\\': out += '\\'; break;
                    case '\'': out += '\''; break; case '"': out += '"'; break;
                    default: Error("Unexpected escaped character \"%c\"", e); break;
                    }
                    continue;
                }
                if (ch == '"') { closed = true; break; }
                out += ch;
            }
            (void)start;
            if (!closed) Error("premature EOF in string");
            out += '"';
            *tok = out;
            return true;
        }
        if (c == '[' || c == ']') { *tok = std::string(1, c); ++pos; return true; }
        size_t start = pos;
        while (pos < text.size()) {
            char ch = text[pos];
            if (ch == ' ' || ch == '\n' || ch == '\t' || ch == '\r' || ch == '"' || ch == '[' || ch == ']') break;
            ++pos;
        }
        *tok = text.substr(start, pos - start);
        return true;
    }
    void unget(const std::string &t) { hasUnget = true; ungot = t; }
};

bool isQuoted(const std::string &s) { return s.size() >= 2 && s.front() == '"' && s.back() == '"'; }
std::string dequote(const std::string &s) { return s.substr(1, s.size() - 2); }

double parseNumber(const std::string &str) {   // parser.cpp:320-366
    if (str.size() == 1) {
        if (!(str[0] >= '0' && str[0] <= '9')) { Error("\"%c\": expected a number", str[0]); std::exit(1); }
        return str[0] - '0';
    }
    bool isInteger = true;
    for (char ch : str) if (!(ch >= '0' && ch <= '9')) isInteger = false;
    char *endptr = nullptr;
    double val;
    if (isInteger) val = double(std::strtol(str.c_str(), &endptr, 10));
    else val = std::strtof(str.c_str(), &endptr);
    if (val == 0 && endptr == str.c_str()) { Error("%s: expected a number", str.c_str()); std::exit(1); }
    return val;
}

struct DeclType { const char *name; ParamType type; int xyz; };
bool lookupType(const std::string &decl, ParamType *type, int *kind, std::string *name) {   // parser.cpp:391-470
    std::istringstream ss(decl);
    std::string t, n;
    ss >> t >> n;
    if (t.empty() || n.empty()) { Error("Parameter \"%s\" doesn't have a type declaration?!", decl.c_str()); return false; }
    *name = n;
    *kind = 0;
    if (t == "float") *type = ParamType::Float;
    else if (t == "integer") *type = ParamType::Int;
    else if (t == "bool") *type = ParamType::Bool;
    else if (t == "point2") *type = ParamType::Point2;
    else if (t == "vector2") *type = ParamType::Vector2;
    else if (t == "point3" || t == "point") *type = ParamType::Point3;
    else if (t == "vector3" || t == "vector") *type = ParamType::Vector3;
    else if (t == "normal3" || t == "normal") *type = ParamType::Normal;
    else if (t == "string") *type = ParamType::String;
    else if (t == "texture") *type = ParamType::Texture;
    else if (t == "color" || t == "rgb") *type = ParamType::Spectrum;
    else if (t == "xyz") { *type = ParamType::Spectrum; *kind = 1; }
    else if (t == "blackbody") { *type = ParamType::Spectrum; *kind = 2; }
    else if (t == "spectrum") { *type = ParamType::Spectrum; *kind = 3; }
    else { Error("Unable to decode type from \"%s\"", decl.c_str()); return false; }
    return true;
}

ParamSet parseParams(Tokenizer &tk) {
    ParamSet ps;
    std::string decl;
    while (tk.next(&decl)) {
        if (!isQuoted(decl)) { tk.unget(decl); return ps; }
        std::vector<double> nums;
        std::vector<std::string> strs;
        auto addVal = [&](const std::string &v) {
            if (isQuoted(v)) {
                if (!nums.empty()) { Error("mixed string and numeric parameters"); std::exit(1); }
                strs.push_back(dequote(v));
            } else {
                if (!strs.empty()) { Error("mixed string and numeric parameters"); std::exit(1); }
                nums.push_back(parseNumber(v));
            }
        };
        std::string val;
        if (!tk.next(&val)) { Error("premature EOF"); std::exit(1); }
        if (val == "[") {
            while (true) {
                if (!tk.next(&val)) { Error("premature EOF"); std::exit(1); }
                if (val == "]") break;
                addVal(val);
            }
        } else
            addVal(val);

        ParamSet::Item item;
        int kind;
        if (!lookupType(dequote(decl), &item.type, &kind, &item.name)) continue;
        bool wantStrings = item.type == ParamType::String || item.type == ParamType::Texture || item.type == ParamType::Bool;
        if (wantStrings && strs.empty() && !nums.empty()) { Error("Expected string parameter value for parameter \"%s\"", item.name.c_str()); continue; }
        if (!wantStrings && !strs.empty() && !(item.type == ParamType::Spectrum && kind == 3)) {
            Error("Expected numeric parameter value for parameter \"%s\"", item.name.c_str());
            continue;
        }
        switch (item.type) {
        case ParamType::Int:
            for (double d : nums) item.i.push_back((int)d);
            break;
        case ParamType::Bool:
            for (auto &s : strs) {
                if (s == "true") item.i.push_back(1);
                else if (s == "false") item.i.push_back(0);
                else { Warning("Value \"%s\" unknown for Boolean parameter \"%s\".Using \"false\".", s.c_str(), item.name.c_str()); item.i.push_back(0); }
            }
            break;
        case ParamType::String: item.s = strs; break;
        case ParamType::Texture:
            if (strs.size() == 1) item.s = strs;
            else { Error("Only one string allowed for \"texture\" parameter \"%s\"", item.name.c_str()); continue; }
            break;
        case ParamType::Spectrum:
            if (kind == 2) {   // temperature (K), scale pairs: parser.cpp:663-672 -> ParamSet::AddBlackbodySpectrum
                if (nums.size() % 2) { Warning("Excess value given with blackbody parameter \"%s\". Ignoring extra one.", item.name.c_str()); nums.resize(nums.size() - 1); }
                for (size_t j = 0; j + 1 < nums.size(); j += 2) {
                    RGB s = BlackbodyRGB((Float)nums[j], (Float)nums[j + 1]);
                    item.f.push_back(s.c[0]); item.f.push_back(s.c[1]); item.f.push_back(s.c[2]);
                }
                break;
            }
            if (kind == 3) {   // SPD file names (one spectrum each) or inline (wavelength nm, value) pairs (one spectrum): parser.cpp:673-689
                if (!strs.empty()) {
                    for (auto &fn : strs) {
                        RGB s = SpectrumFromFile(AbsolutePathFromScene(fn));
                        item.f.push_back(s.c[0]); item.f.push_back(s.c[1]); item.f.push_back(s.c[2]);
                    }
                } else {
                    if (nums.size() % 2) { Warning("Non-even number of values given with sampled spectrum parameter \"%s\". Ignoring extra.", item.name.c_str()); nums.resize(nums.size() - 1); }
                    std::vector<Float> wl, v;
                    for (size_t j = 0; j + 1 < nums.size(); j += 2) { wl.push_back((Float)nums[j]); v.push_back((Float)nums[j + 1]); }
                    RGB s = SpectrumFromSampled(wl.data(), v.data(), (int)wl.size());
                    item.f.push_back(s.c[0]); item.f.push_back(s.c[1]); item.f.push_back(s.c[2]);
                }
                break;
            }
            if (nums.size() % 3) { Warning("Excess RGB values given with parameter \"%s\". Ignoring last %d of them", item.name.c_str(), (int)(nums.size() % 3)); nums.resize(nums.size() - nums.size() % 3); }
            for (size_t j = 0; j + 2 < nums.size() + 0; j += 3) {
                Float a = (Float)nums[j], b = (Float)nums[j + 1], c = (Float)nums[j + 2];
                if (kind == 1) {   // XYZToRGB, spectrum.h:56-60
                    Float r = 3.240479f * a - 1.537150f * b - 0.498535f * c;
                    Float g = -0.969256f * a + 1.875991f * b + 0.041556f * c;
                    Float bl = 0.055648f * a - 0.204043f * b + 1.057311f * c;
                    a = r; b = g; c = bl;
                }
                item.f.push_back(a); item.f.push_back(b); item.f.push_back(c);
            }
            break;
        default: {
            size_t per = (item.type == ParamType::Float) ? 1 : (item.type == ParamType::Point2 || item.type == ParamType::Vector2) ? 2 : 3;
            if (nums.size() % per) { Warning("Excess values given with parameter \"%s\". Ignoring last %d of them.", item.name.c_str(), (int)(nums.size() % per)); nums.resize(nums.size() - nums.size() % per); }
            for (double d : nums) item.f.push_back((Float)d);
        }
        }
        ps.Add(std::move(item));
    }
    return ps;
}

void parse(std::unique_ptr<Tokenizer> first) {
    std::vector<std::unique_ptr<Tokenizer>> stack;
    stack.push_back(std::move(first));
    g_loc = &stack.back()->loc;
    auto nextToken = [&](std::string *tok, bool required) -> bool {
        while (true) {
            if (stack.empty()) { if (required) { Error("premature EOF"); std::exit(1); } return false; }
            if (stack.back()->next(tok)) return true;
            stack.pop_back();
            g_loc = stack.empty() ? nullptr : &stack.back()->loc;
        }
    };
    auto needNum = [&]() { std::string t; nextToken(&t, true); return (Float)parseNumber(t); };
    auto needStr = [&]() { std::string t; nextToken(&t, true); if (!isQuoted(t)) { Error("Expected quoted string, found \"%s\"", t.c_str()); std::exit(1); } return dequote(t); };
    auto basicParam = [&](void (*fn)(const std::string &, const ParamSet &)) {
        std::string n = needStr();
        ParamSet ps = parseParams(*stack.back());
        fn(n, ps);
    };
    auto syntaxError = [&](const std::string &t) { Error("Unknown directive: %s", t.c_str()); std::exit(1); };

    std::string tok;
    while (nextToken(&tok, false)) {
        switch (tok[0]) {
        case 'A':
            if (tok == "AttributeBegin") pbrtAttributeBegin();
            else if (tok == "AttributeEnd") pbrtAttributeEnd();
            else if (tok == "ActiveTransform") {
                std::string a; nextToken(&a, true);
                if (a == "All") pbrtActiveTransformAll();
                else if (a == "EndTime") pbrtActiveTransformEndTime();
                else if (a == "StartTime") pbrtActiveTransformStartTime();
                else syntaxError(tok);
            } else if (tok == "AreaLightSource") basicParam(pbrtAreaLightSource);
            else if (tok == "Accelerator") basicParam(pbrtAccelerator);
            else syntaxError(tok);
            break;
        case 'C':
            if (tok == "ConcatTransform") {
                std::string b; nextToken(&b, true);
                if (b != "[") syntaxError(tok);
                Float m[16];
                for (int i = 0; i < 16; ++i) m[i] = needNum();
                nextToken(&b, true);
                if (b != "]") syntaxError(tok);
                pbrtConcatTransform(m);
            } else if (tok == "CoordinateSystem") pbrtCoordinateSystem(needStr());
            else if (tok == "CoordSysTransform") pbrtCoordSysTransform(needStr());
            else if (tok == "Camera") basicParam(pbrtCamera);
            else syntaxError(tok);
            break;
        case 'F':
            if (tok == "Film") basicParam(pbrtFilm); else syntaxError(tok);
            break;
        case 'I':
            if (tok == "Integrator") basicParam(pbrtIntegrator);
            else if (tok == "Include") {
                std::string fn = needStr();
                std::string path = (fn.size() && fn[0] == '/') ? fn : g_searchDirectory + fn;
                std::unique_ptr<Tokenizer> t(new Tokenizer);
                if (Tokenizer::fromFile(path, t.get())) { stack.push_back(std::move(t)); g_loc = &stack.back()->loc; }
                else Error("Couldn't open included file \"%s\"", path.c_str());
            } else if (tok == "Identity") pbrtIdentity();
            else syntaxError(tok);
            break;
        case 'L':
            if (tok == "LightSource") basicParam(pbrtLightSource);
            else if (tok == "LookAt") { Float v[9]; for (int i = 0; i < 9; ++i) v[i] = needNum(); pbrtLookAt(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]); }
            else syntaxError(tok);
            break;
        case 'M':
            if (tok == "MakeNamedMaterial") basicParam(pbrtMakeNamedMaterial);
            else if (tok == "MakeNamedMedium") basicParam(pbrtMakeNamedMedium);
            else if (tok == "Material") basicParam(pbrtMaterial);
            else if (tok == "MediumInterface") {
                std::string a = needStr(), b = a, t;
                if (nextToken(&t, false)) { if (isQuoted(t)) b = dequote(t); else stack.back()->unget(t); }
                pbrtMediumInterface(a, b);
            } else syntaxError(tok);
            break;
        case 'N':
            if (tok == "NamedMaterial") pbrtNamedMaterial(needStr()); else syntaxError(tok);
            break;
        case 'O':
            if (tok == "ObjectBegin") pbrtObjectBegin(needStr());
            else if (tok == "ObjectEnd") pbrtObjectEnd();
            else if (tok == "ObjectInstance") pbrtObjectInstance(needStr());
            else syntaxError(tok);
            break;
        case 'P':
            if (tok == "PixelFilter") basicParam(pbrtPixelFilter); else syntaxError(tok);
            break;
        case 'R':
            if (tok == "ReverseOrientation") pbrtReverseOrientation();
            else if (tok == "Rotate") { Float v[4]; for (int i = 0; i < 4; ++i) v[i] = needNum(); pbrtRotate(v[0], v[1], v[2], v[3]); }
            else syntaxError(tok);
            break;
        case 'S':
            if (tok == "Shape") basicParam(pbrtShape);
            else if (tok == "Sampler") basicParam(pbrtSampler);
            else if (tok == "Scale") { Float v[3]; for (int i = 0; i < 3; ++i) v[i] = needNum(); pbrtScale(v[0], v[1], v[2]); }
            else syntaxError(tok);
            break;
        case 'T':
            if (tok == "TransformBegin") pbrtTransformBegin();
            else if (tok == "TransformEnd") pbrtTransformEnd();
            else if (tok == "Transform") {
                std::string b; nextToken(&b, true);
                if (b != "[") syntaxError(tok);
                Float m[16];
                for (int i = 0; i < 16; ++i) m[i] = needNum();
                nextToken(&b, true);
                if (b != "]") syntaxError(tok);
                pbrtTransform(m);
            } else if (tok == "Translate") { Float v[3]; for (int i = 0; i < 3; ++i) v[i] = needNum(); pbrtTranslate(v[0], v[1], v[2]); }
            else if (tok == "TransformTimes") { Float a = needNum(), b = needNum(); pbrtTransformTimes(a, b); }
            else if (tok == "Texture") {
                std::string n = needStr(), type = needStr(), texName = needStr();
                ParamSet ps = parseParams(*stack.back());
                pbrtTexture(n, type, texName, ps);
            } else syntaxError(tok);
            break;
        case 'W':
            if (tok == "WorldBegin") pbrtWorldBegin();
            else if (tok == "WorldEnd") pbrtWorldEnd();
            else syntaxError(tok);
            break;
        default: syntaxError(tok);
        }
    }
    g_loc = nullptr;
}
}  // namespace

std::string CurrentParserLocation() {
    if (!g_loc) return "";
    return g_loc->filename + ":" + std::to_string(g_loc->line) + ": ";
}

std::string AbsolutePathFromScene(const std::string &f) {   // fileutil.cpp ResolveFilename
    if (f.empty() || f[0] == '/') return f;
    return g_searchDirectory + f;
}

void pbrtParseFile(std::string filename) {
    size_t slash = filename.find_last_of('/');
    g_searchDirectory = slash == std::string::npos ? "" : filename.substr(0, slash + 1);
    std::unique_ptr<Tokenizer> t(new Tokenizer);
    if (!Tokenizer::fromFile(filename, t.get())) { Error("Couldn't open scene file \"%s\"", filename.c_str()); return; }
    parse(std::move(t));
}

void pbrtParseString(std::string str) {
    std::unique_ptr<Tokenizer> t(new Tokenizer);
    t->text = std::move(str);
    t->loc.filename = "<string>";
    parse(std::move(t));
}

}  // namespace pbrt_amd
