// Host-side compile of vision3d_amd/csrc/rotated_iou.h (the DEVICE geometry core) so that the CPU
// test-suite can check its logic against oracle/ without a GPU.  Test infrastructure only.
#include <cstddef>
#include "../../vision3d_amd/csrc/rotated_iou.h"

extern "C" void host_box_iou_rotated(const float* b1, int M, const float* b2, int N, float* out) {
  for (int i = 0; i < M; i++)
    for (int j = 0; j < N; j++) out[(size_t)i * N + j] = v3d::single_box_iou_rotated(b1 + 5 * i, b2 + 5 * j);
}

// The same pairs through the STRIDED work-array form the kernels use (element e of "lane" L at [e * 64 + L], the per-wave
// LDS slab of the device): 64 pairs share one slab, exactly as 64 lanes do.  Must equal the plain form bit for bit.
extern "C" void host_box_iou_rotated_strided(const float* b1, int M, const float* b2, int N, float* out) {
  static v3d::P2 pts[24 * 64];
  static float dist[24 * 64];
  long long pair = 0;
  for (int i = 0; i < M; i++)
    for (int j = 0; j < N; j++, pair++) {
      const int lane = (int)(pair & 63);
      const v3d::BoxPrep a = v3d::prep_box(b1 + 5 * i), b = v3d::prep_box(b2 + 5 * j);
      out[(size_t)i * N + j] = v3d::iou_prepped_lds(a, b, pts + lane, dist + lane);
    }
}
