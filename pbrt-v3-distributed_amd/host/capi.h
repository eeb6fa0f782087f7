// The handle behind libpbrt_amd_host.so's C entry points (host/capi.cpp, host/blob.cpp)
#pragma once
#include <memory>

#include "api.h"

struct pbrt_amd_scene {
    std::unique_ptr<pbrt_amd::BuiltScene> built;   // null for a scene mapped from a blob: no parser state, no Film
    std::unique_ptr<pbrt_amd::FlatScene> flat;     // the mi_scene_desc (+ the arrays it points to, or pointers into `map`)
    void *map = nullptr;                           // pbrt_amd_scene_map_blob: the private mapping of the blob file
    size_t mapBytes = 0;
};
