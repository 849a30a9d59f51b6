// ref_driver.cpp -- C-ABI driver around the REFERENCE's own rotated-IoU core.
//
// TEST INFRASTRUCTURE ONLY (see the header of oracle/oracle.py and DESIGN.md section 3).  This file contains no geometry: it #includes
// the reference header in place (-I/root/reference/vision3d/ops/csrc/box_iou_rotated, see
// oracle/Makefile) and exposes detectron2::single_box_iou_rotated<float>
// (box_iou_rotated_utils.h:313-340) through plain C symbols, so tests can pin oracle/v3d_oracle.c
// and the HIP kernels against the reference arithmetic itself.  Built only when /root/reference
// exists; output goes to oracle/_ref/ (git-ignored and gpurun-ignored: a CPU-container artefact).
//
// The pairwise loop and the greedy NMS loop below restate box_iou_rotated_cpu.cpp:23-28 and
// nms_rotated_cpu.cpp:36-58 (those files need ATen and are not compiled here); every IoU value
// they consume comes from the reference header.
#include <cstdint>
#include <vector>

#include "box_iou_rotated_utils.h"

extern "C" {

float ref_single_box_iou_rotated(const float* b1, const float* b2) {
  return detectron2::single_box_iou_rotated<float>(b1, b2);
}

void ref_box_iou_rotated(const float* b1, int M, const float* b2, int N, float* out) {
  for (int i = 0; i < M; i++)
    for (int j = 0; j < N; j++)
      out[(size_t)i * N + j] = detectron2::single_box_iou_rotated<float>(b1 + 5 * i, b2 + 5 * j);
}

// order = indices by descending score (what scores.sort(0, true) returns, nms_rotated_cpu.cpp:25)
int ref_nms_rotated(const float* boxes, const int64_t* order, int N, float thr, int64_t* keep) {
  std::vector<uint8_t> sup((size_t)(N > 0 ? N : 1), 0);
  int nk = 0;
  for (int _i = 0; _i < N; _i++) {
    int64_t i = order[_i];
    if (sup[i]) continue;
    keep[nk++] = i;
    for (int _j = _i + 1; _j < N; _j++) {
      int64_t j = order[_j];
      if (sup[j]) continue;
      float ovr = detectron2::single_box_iou_rotated<float>(boxes + 5 * i, boxes + 5 * j);
      if (ovr >= thr) sup[j] = 1;
    }
  }
  return nk;
}

}  // extern "C"
