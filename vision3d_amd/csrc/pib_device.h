// pib_device.h -- the point-in-rotated-box test shared by points_in_boxes_kernel (iou_nms.hip) and the fused augmentation
// (augment.hip): vision3d/core/geometry.py:4-65.  Corner arithmetic in fp64 exactly where numpy promotes (geometry.py:21);
// cos / sin evaluated on the float32 yaw.  Compiled with -ffp-contract=off like every geometry kernel.
#pragma once
#include <math.h>

struct PibBox {
  double cx[4], cy[4];
  float zlo, zhi;
};

// bx = (x, y, z, w, l, h, yaw), float32
__device__ __forceinline__ PibBox pib_prep(const float* bx) {
  const float cf = cosf(bx[6]), sf = sinf(bx[6]);
  const double ux[4] = {-0.5, 0.5, 0.5, -0.5}, uy[4] = {-0.5, -0.5, 0.5, 0.5};
  PibBox pb;
#pragma unroll
  for (int v = 0; v < 4; v++) {
    const double lx = (double)bx[3] * ux[v], ly = (double)bx[4] * uy[v];
    pb.cx[v] = ((double)cf * lx + (double)(-sf) * ly) + (double)bx[0];
    pb.cy[v] = ((double)sf * lx + (double)cf * ly) + (double)bx[1];
  }
  pb.zlo = bx[2] - bx[5] / 2;
  pb.zhi = bx[2] + bx[5] / 2;
  return pb;
}

__device__ __forceinline__ bool pib_inside(const PibBox& pb, float px, float py, float pz, bool use_z) {
  bool in = true;
  if (use_z) in = (pz > pb.zlo) && (pz < pb.zhi);
#pragma unroll
  for (int v = 0; v < 4; v++) {
    const int pv = (v + 3) & 3;
    const double sx = -(pb.cx[v] - pb.cx[pv]), sy = -(pb.cy[v] - pb.cy[pv]);
    const double vx = pb.cx[v] - (double)px, vy = pb.cy[v] - (double)py;
    in = in && (sx * vy - sy * vx > 0);
  }
  return in;
}
