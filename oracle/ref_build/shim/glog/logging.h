// Build-infrastructure shim, NOT part of the product and not a copy of glog.
// Lets the unmodified pbrt-v3 sources under /root/reference compile without the
// (empty) src/ext/glog submodule. Semantics needed by the reference (SURVEY.md
// Appendix B): LOG(sev) evaluates its stream operands only for FATAL;
// VLOG never evaluates; CHECK* abort with a message; DCHECK* compile out.
#ifndef ORACLE_SHIM_GLOG_LOGGING_H
#define ORACLE_SHIM_GLOG_LOGGING_H
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>

namespace google {
inline void InitGoogleLogging(const char *) {}
}
extern int FLAGS_stderrthreshold, FLAGS_minloglevel, FLAGS_v;
extern bool FLAGS_logtostderr;
extern std::string FLAGS_log_dir;

namespace shimlog {
struct FatalSink {
    std::ostringstream os;
    FatalSink(const char *file, int line, const char *what) {
        os << "FATAL " << file << ":" << line << " " << what << " ";
    }
    template <typename T> FatalSink &operator<<(const T &v) { os << v; return *this; }
    FatalSink &operator<<(std::ostream &(*f)(std::ostream &)) { os << f; return *this; }
    [[noreturn]] ~FatalSink() {
        std::cerr << os.str() << std::endl;
        std::abort();
    }
};
struct NullSink {
    template <typename T> NullSink &operator<<(const T &) { return *this; }
    NullSink &operator<<(std::ostream &(*)(std::ostream &)) { return *this; }
};
// Timing of the reference's own hot path (bench.py's cpu_baseline): with PBRT_REF_RENDER_TIMES=1 in the
// environment the LOG(INFO) lines of SamplerIntegrator::Render ("Starting image tile ..." / "Rendering finished",
// core/integrator.cpp:256,335) are evaluated and stamped; the wall time between the first tile start and the end of
// the tile loop goes to stderr as "[pbrt_ref] Integrator::Render seconds <t>".  Off (the default): LOG(INFO) costs one
// load of a global flag and evaluates nothing, as before.  The reference's sources are not touched.
extern bool g_render_times;
void RenderMark(const std::string &msg);   // shim/stubs.cpp
struct InfoSink {
    std::ostringstream os;
    template <typename T> InfoSink &operator<<(const T &v) { os << v; return *this; }
    InfoSink &operator<<(std::ostream &(*f)(std::ostream &)) { os << f; return *this; }
    ~InfoSink() { RenderMark(os.str()); }
};
struct Voidify {
    void operator&(const FatalSink &) {}
    void operator&(const NullSink &) {}
    void operator&(const InfoSink &) {}
};
enum { INFO = 0, WARNING = 1, ERROR = 2, FATAL = 3 };
}  // namespace shimlog

// operands are never evaluated unless the severity is FATAL (or INFO while PBRT_REF_RENDER_TIMES=1 asks for the render timing)
#define LOG(sev) SHIM_LOG_##sev
#define SHIM_LOG_FATAL shimlog::Voidify() & shimlog::FatalSink(__FILE__, __LINE__, "LOG(FATAL)")
#define SHIM_LOG_ERROR true ? (void)0 : shimlog::Voidify() & shimlog::NullSink()
#define SHIM_LOG_WARNING true ? (void)0 : shimlog::Voidify() & shimlog::NullSink()
#define SHIM_LOG_INFO !shimlog::g_render_times ? (void)0 : shimlog::Voidify() & shimlog::InfoSink()
#define VLOG(n) true ? (void)0 : shimlog::Voidify() & shimlog::NullSink()
#define CHECK(c) \
    (c) ? (void)0 : shimlog::Voidify() & shimlog::FatalSink(__FILE__, __LINE__, "Check failed: " #c)
#define SHIM_CHECK_OP(a, op, b)                                                       \
    ((a)op(b)) ? (void)0                                                              \
               : shimlog::Voidify() &                                                 \
                     shimlog::FatalSink(__FILE__, __LINE__, "Check failed: " #a " " #op " " #b)
#define CHECK_EQ(a, b) SHIM_CHECK_OP(a, ==, b)
#define CHECK_NE(a, b) SHIM_CHECK_OP(a, !=, b)
#define CHECK_LT(a, b) SHIM_CHECK_OP(a, <, b)
#define CHECK_LE(a, b) SHIM_CHECK_OP(a, <=, b)
#define CHECK_GT(a, b) SHIM_CHECK_OP(a, >, b)
#define CHECK_GE(a, b) SHIM_CHECK_OP(a, >=, b)
#define CHECK_NOTNULL(p) (p)
// NDEBUG build: DCHECKs vanish but must still accept trailing <<
#define DCHECK(c) true ? (void)0 : shimlog::Voidify() & shimlog::NullSink()
#define DCHECK_EQ(a, b) DCHECK(0)
#define DCHECK_NE(a, b) DCHECK(0)
#define DCHECK_LT(a, b) DCHECK(0)
#define DCHECK_LE(a, b) DCHECK(0)
#define DCHECK_GT(a, b) DCHECK(0)
#define DCHECK_GE(a, b) DCHECK(0)
#endif
