// Build-infrastructure stub for the absent OpenEXR submodule (EXR I/O disabled in the
// oracle build; parity uses .pfm, reference core/imageio.cpp:437-482).
#ifndef ORACLE_SHIM_IMFRGBA_H
#define ORACLE_SHIM_IMFRGBA_H
#include <stdexcept>
namespace Imath {
struct V2i { int x, y; V2i(int x = 0, int y = 0) : x(x), y(y) {} };
struct Box2i { V2i min, max; Box2i() {} Box2i(V2i a, V2i b) : min(a), max(b) {} };
}
namespace Imf {
struct Rgba {
    float r, g, b, a;
    Rgba() : r(0), g(0), b(0), a(1) {}
    Rgba(float r, float g, float b, float a = 1.f) : r(r), g(g), b(b), a(a) {}
};
enum RgbaChannels { WRITE_RGBA = 0xf };
}
#endif
