// Host-side compile of vision3d_amd/csrc/rotated_iou.h (the DEVICE geometry core) so that the CPU
// test-suite can check its logic against oracle/ without a GPU.  Test infrastructure only.
#include <cstddef>
#include "../../vision3d_amd/csrc/rotated_iou.h"

extern "C" void host_box_iou_rotated(const float* b1, int M, const float* b2, int N, float* out) {
  for (int i = 0; i < M; i++)
    for (int j = 0; j < N; j++) out[(size_t)i * N + j] = v3d::single_box_iou_rotated(b1 + 5 * i, b2 + 5 * j);
}
