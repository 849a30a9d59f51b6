// capi.hip -- version / diagnostics entry points of the C ABI (include/vision3d_hip.h).
// Replaces the probes of vision3d/ops/csrc/vision.cpp:14-58 and cuda_version.cu:6-8.
#include "v3d_common.h"

#define V3D_STR_(x) #x
#define V3D_STR(x) V3D_STR_(x)

extern "C" const char* v3d_version(void) { return "vision3d_hip 0.1 (gfx950)"; }

extern "C" int v3d_hip_runtime_version(void) {
  int v = 0;
  if (hipRuntimeGetVersion(&v) != hipSuccess) return -1;
  return v;
}

extern "C" const char* v3d_compiler_version(void) {
  return "clang " V3D_STR(__clang_major__) "." V3D_STR(__clang_minor__) "." V3D_STR(__clang_patchlevel__);
}

extern "C" const char* v3d_error_string(int code) {
  switch (code) {
    case V3D_OK: return "ok";
    case V3D_EINVAL: return "invalid argument (size, null pointer or unsupported shape)";
    case V3D_EWORKSPACE: return "workspace too small";
    case V3D_EUNSUPPORTED: return "unsupported configuration";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}
