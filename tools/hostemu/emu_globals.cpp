// see shim/hip/hip_runtime.h -- developer tool, not part of the product
#include <hip/hip_runtime.h>
thread_local emu_idx threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;
thread_local void *emu_dyn_lds = nullptr;
