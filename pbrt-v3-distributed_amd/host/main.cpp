// pbrt_amd: command-line front end, same shape as the reference's src/main/pbrt.cpp:76-173
// (flags --nthreads --outfile --cropwindow --quick --quiet), plus --gpus N.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "api.h"

using namespace pbrt_amd;

static void usage(const char *msg = nullptr) {
    if (msg) std::fprintf(stderr, "pbrt_amd: %s\n\n", msg);
    std::fprintf(stderr, "usage: pbrt_amd [<options>] <filename.pbrt...>\n"
                         "  --cropwindow <x0> <x1> <y0> <y1>  Specify an image crop window.\n"
                         "  --gpus <num>         Shard image tiles over this many MI355X devices.\n"
                         "  --help               Print this help text.\n"
                         "  --nthreads <num>     Host threads for scene construction (default: all cores).\n"
                         "  --outfile <filename> Write the final image to the given filename (.pfm, .exr).\n"
                         "  --fast-samplers      Render scenes that name the random / stratified / 02sequence samplers with sobol (same sample count):\n"
                         "                       full wavefront speed instead of the reference's exact image (one random stream per tile, walked serially).\n"
                         "  --quick              Automatically reduce a number of quality settings to render more quickly.\n"
                         "  --quiet              Suppress all text output other than error messages.\n");
    std::exit(msg ? 1 : 0);
}

int main(int argc, char *argv[]) {
    Options options;
    std::vector<std::string> filenames;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--nthreads") || !std::strcmp(argv[i], "-nthreads")) { if (i + 1 == argc) usage("missing value after --nthreads argument"); options.nThreads = std::atoi(argv[++i]); }
        else if (!std::strncmp(argv[i], "--nthreads=", 11)) options.nThreads = std::atoi(&argv[i][11]);
        else if (!std::strcmp(argv[i], "--gpus")) { if (i + 1 == argc) usage("missing value after --gpus argument"); options.nGpus = std::atoi(argv[++i]); }
        else if (!std::strcmp(argv[i], "--outfile") || !std::strcmp(argv[i], "-outfile")) { if (i + 1 == argc) usage("missing value after --outfile argument"); options.imageFile = argv[++i]; }
        else if (!std::strncmp(argv[i], "--outfile=", 10)) options.imageFile = &argv[i][10];
        else if (!std::strcmp(argv[i], "--cropwindow") || !std::strcmp(argv[i], "-cropwindow")) {
            if (i + 4 >= argc) usage("missing value after --cropwindow argument");
            options.cropWindow[0][0] = std::atof(argv[++i]); options.cropWindow[0][1] = std::atof(argv[++i]);
            options.cropWindow[1][0] = std::atof(argv[++i]); options.cropWindow[1][1] = std::atof(argv[++i]);
        } else if (!std::strcmp(argv[i], "--quick") || !std::strcmp(argv[i], "-quick")) options.quickRender = true;
        else if (!std::strcmp(argv[i], "--fast-samplers")) options.fastSamplers = true;
        else if (!std::strcmp(argv[i], "--quiet") || !std::strcmp(argv[i], "-quiet")) options.quiet = true;
        else if (!std::strcmp(argv[i], "--help") || !std::strcmp(argv[i], "-help") || !std::strcmp(argv[i], "-h")) usage();
        else if (argv[i][0] == '-') usage((std::string("unknown option ") + argv[i]).c_str());
        else filenames.push_back(argv[i]);
    }
    if (!options.quiet) std::printf("pbrt_amd: MI355X wavefront path tracer behind pbrt-v3's scene API\n");
    if (filenames.empty()) usage("no scene file given");
    pbrtInit(options);
    for (const std::string &f : filenames) pbrtParseFile(f);
    pbrtCleanup();
    return (g_unsupportedCount > 0 || g_renderFailed) ? 1 : 0;
}
