// see ImfRgba.h: stub, every EXR operation throws (caught by the reference's own handlers)
#ifndef ORACLE_SHIM_IMFRGBAFILE_H
#define ORACLE_SHIM_IMFRGBAFILE_H
#include "ImfRgba.h"
namespace Imf {
struct RgbaInputFile {
    RgbaInputFile(const char *) { throw std::runtime_error("EXR disabled in oracle build"); }
    Imath::Box2i dataWindow() const { return Imath::Box2i(); }
    Imath::Box2i displayWindow() const { return Imath::Box2i(); }
    void setFrameBuffer(Rgba *, int, int) {}
    void readPixels(int, int) {}
};
struct RgbaOutputFile {
    RgbaOutputFile(const char *, const Imath::Box2i &, const Imath::Box2i &, RgbaChannels) {
        throw std::runtime_error("EXR disabled in oracle build; use a .pfm output name");
    }
    void setFrameBuffer(Rgba *, int, int) {}
    void writePixels(int) {}
};
}
#endif
