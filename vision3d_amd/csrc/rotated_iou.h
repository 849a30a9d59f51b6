// rotated_iou.h -- IoU of two rotated BEV rectangles (xc, yc, w, h, angle_degrees), fp32.
//
// Arithmetic contract: value-for-value the HOST branch of the reference's
// vision3d/ops/csrc/box_iou_rotated/box_iou_rotated_utils.h (the CPU path is the parity target named
// by BASELINE.json): same operation order, the same promotion points to double (:61, :318-319, :283),
// the same tolerances (1e-14 parallel test :97, 1e-6 angular tie :220, 1e-8 duplicate test :233),
// the same hull-sort visiting order as libstdc++'s std::sort for the <= 23 points that can occur,
// and the reference's quirk of consulting the UNSORTED dist[] after the sort (:216-236).
// Must be compiled with -ffp-contract=off (the x86-64 reference build has no FMA contraction).
//
// Written for the device (flat register arrays, no recursion, early exits that cannot change the
// value); the V3D_HD macro also lets tests/ compile it for the host with g++ to check the logic
// against oracle/ where no GPU exists.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define V3D_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define V3D_HD inline
#endif

namespace v3d {

struct P2 {
  float x, y;
};

V3D_HD float cross2(const P2 a, const P2 b) { return a.x * b.y - b.x * a.y; }
V3D_HD float dot2(const P2 a, const P2 b) { return a.x * b.x + a.y * b.y; }
V3D_HD P2 sub2(const P2 a, const P2 b) { return P2{a.x - b.x, a.y - b.y}; }

// Half-extent direction cosines of a box: (cos(theta)/2, sin(theta)/2), theta evaluated in double
// (utils.h:61-63).  They depend on the angle only, so kernels compute them ONCE per box.
V3D_HD void half_trig(float angle_deg, float& c2, float& s2) {
  const double theta = angle_deg * 0.01745329251;
  c2 = (float)cos(theta) * 0.5f;
  s2 = (float)sin(theta) * 0.5f;
}

// utils.h:56-74 with the trig factored out
V3D_HD void vertices(float xc, float yc, float w, float h, float c2, float s2, P2 (&p)[4]) {
  p[0].x = xc - s2 * h - c2 * w;
  p[0].y = yc + c2 * h - s2 * w;
  p[1].x = xc + s2 * h - c2 * w;
  p[1].y = yc - c2 * h - s2 * w;
  p[2].x = 2 * xc - p[0].x;
  p[2].y = 2 * yc - p[0].y;
  p[3].x = 2 * xc - p[1].x;
  p[3].y = 2 * yc - p[1].y;
}

// hull comparator, utils.h:217-225
V3D_HD bool hull_less(const P2 A, const P2 B) {
  const float t = cross2(A, B);
  if (fabs((double)t) < 1e-6) return dot2(A, A) < dot2(B, B);
  return t > 0;
}

// ---- point / distance work arrays --------------------------------------------------------------
// The clipper indexes its <= 24-point arrays dynamically, so as plain locals they live in SCRATCH (400 bytes per
// lane: every access is a global-memory round trip -- the 19 us of one NMS mask evaluation were mostly that).  The
// algorithms below are written against a strided view instead: stride 1 over locals (host, and kernels that keep the
// scratch form), stride 64 over a per-wave LDS slab in which element e of lane L sits at [e * 64 + L] (conflict-free).
template <int S>
struct P2View {
  P2* p;
  V3D_HD P2& operator[](int i) const { return p[i * S]; }
};
template <int S>
struct F1View {
  float* p;
  V3D_HD float& operator[](int i) const { return p[i * S]; }
};

// ---- std::sort (libstdc++ introsort) visiting order, for n <= 23 elements -------------------------
template <class A>
V3D_HD void ins_unguarded(A a, int last) {
  const P2 val = a[last];
  int next = last - 1;
  while (hull_less(val, a[next])) {
    a[last] = a[next];
    last = next;
    --next;
  }
  a[last] = val;
}
template <class A>
V3D_HD void ins_sort(A a, int first, int last) {  // [first, last)
  for (int i = first + 1; i < last; ++i) {
    if (hull_less(a[i], a[first])) {
      const P2 val = a[i];
      for (int j = i; j > first; --j) a[j] = a[j - 1];
      a[first] = val;
    } else {
      ins_unguarded(a, i);
    }
  }
}
template <class A>
V3D_HD void swap2(A a, int i, int j) {
  const P2 t = a[i];
  a[i] = a[j];
  a[j] = t;
}
template <class A>
V3D_HD void median_to_first(A q, int result, int a, int b, int c) {
  if (hull_less(q[a], q[b])) {
    if (hull_less(q[b], q[c])) swap2(q, result, b);
    else if (hull_less(q[a], q[c])) swap2(q, result, c);
    else swap2(q, result, a);
  } else if (hull_less(q[a], q[c])) swap2(q, result, a);
  else if (hull_less(q[b], q[c])) swap2(q, result, c);
  else swap2(q, result, b);
}
template <class A>
V3D_HD int unguarded_partition(A q, int first, int last, int pivot) {
  for (;;) {
    while (hull_less(q[first], q[pivot])) ++first;
    --last;
    while (hull_less(q[pivot], q[last])) --last;
    if (!(first < last)) return first;
    swap2(q, first, last);
    ++first;
  }
}
// sort q[first, last) exactly as std::sort would (depth limit is never reached for n <= 23)
template <class A>
V3D_HD void std_sort(A q, int first, int last) {
  if (last - first <= 1) return;
  // introsort loop, recursion on the right part unrolled into a small explicit stack
  int stack_first[6], stack_last[6], sp = 0;
  stack_first[0] = first;
  stack_last[0] = last;
  sp = 1;
  while (sp > 0) {
    --sp;
    int f = stack_first[sp], l = stack_last[sp];
    while (l - f > 16) {
      const int mid = f + (l - f) / 2;
      median_to_first(q, f, f + 1, mid, l - 1);
      const int cut = unguarded_partition(q, f + 1, l, f);
      if (sp < 6) {  // right part handled later (order of the two parts does not matter: disjoint)
        stack_first[sp] = cut;
        stack_last[sp] = l;
        ++sp;
      }
      l = cut;
    }
  }
  if (last - first > 16) {
    ins_sort(q, first, first + 16);
    for (int i = first + 16; i < last; ++i) ins_unguarded(q, i);
  } else {
    ins_sort(q, first, last);
  }
}

// Intersection area of two rectangles given their vertices (utils.h:76-309).  q: 24 points, dist: 24 floats of work
// space (the intersection points are shifted in place: the reference's separate `ip` and `q` hold ip[i] and
// ip[i] - start, and ip is dead once q is built).
template <class A, class F>
V3D_HD float intersection_area_ws(const P2 (&p1)[4], const P2 (&p2)[4], A q, F dist) {
  int num = 0;
  P2 v1[4], v2[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    v1[i] = sub2(p1[(i + 1) & 3], p1[i]);
    v2[i] = sub2(p2[(i + 1) & 3], p2[i]);
  }
  for (int i = 0; i < 4; i++) {
    for (int j = 0; j < 4; j++) {
      const float det = cross2(v2[j], v1[i]);
      if (fabs((double)det) <= 1e-14) continue;
      const P2 v12 = sub2(p2[j], p1[i]);
      const float t1 = cross2(v2[j], v12) / det;
      const float t2 = cross2(v1[i], v12) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) {
        q[num] = P2{p1[i].x + v1[i].x * t1, p1[i].y + v1[i].y * t1};
        num++;
      }
    }
  }
  {
    const P2 AB = v2[0], DA = v2[3];
    const float ABdotAB = dot2(AB, AB), ADdotAD = dot2(DA, DA);
    for (int i = 0; i < 4; i++) {
      const P2 AP = sub2(p1[i], p2[0]);
      const float APdotAB = dot2(AP, AB);
      const float APdotAD = -dot2(AP, DA);
      if (APdotAB >= 0 && APdotAD >= 0 && APdotAB <= ABdotAB && APdotAD <= ADdotAD) q[num++] = p1[i];
    }
  }
  {
    const P2 AB = v1[0], DA = v1[3];
    const float ABdotAB = dot2(AB, AB), ADdotAD = dot2(DA, DA);
    for (int i = 0; i < 4; i++) {
      const P2 AP = sub2(p2[i], p1[0]);
      const float APdotAB = dot2(AP, AB);
      const float APdotAD = -dot2(AP, DA);
      if (APdotAB >= 0 && APdotAD >= 0 && APdotAB <= ABdotAB && APdotAD <= ADdotAD) q[num++] = p2[i];
    }
  }
  if (num <= 2) return 0.0f;

  // Graham scan (utils.h:157-270, shift_to_zero = true)
  int t = 0;
  for (int i = 1; i < num; i++) {
    const P2 pi = q[i], pt = q[t];
    if (pi.y < pt.y || (pi.y == pt.y && pi.x < pt.x)) t = i;
  }
  const P2 start = q[t];
  for (int i = 0; i < num; i++) q[i] = sub2(q[i], start);
  swap2(q, 0, t);
  for (int i = 0; i < num; i++) {
    const P2 qi = q[i];
    dist[i] = dot2(qi, qi);
  }
  std_sort(q, 1, num);  // dist[] deliberately NOT permuted (reference host-branch quirk)
  int k;
  for (k = 1; k < num; k++)
    if ((double)dist[k] > 1e-8) break;
  if (k == num) return 0.0f;  // hull is a single point -> polygon_area(m=1) = 0
  q[1] = q[k];
  int m = 2;
  for (int i = k + 1; i < num; i++) {
    const P2 qi = q[i];
    while (m > 1 && cross2(sub2(qi, q[m - 2]), sub2(q[m - 1], q[m - 2])) >= 0) m--;
    q[m++] = qi;
  }
  if (m <= 2) return 0.0f;
  float area = 0;
  const P2 q0 = q[0];
  for (int i = 1; i < m - 1; i++) area += (float)fabs((double)cross2(sub2(q[i], q0), sub2(q[i + 1], q0)));
  return (float)(area / 2.0);
}

V3D_HD float intersection_area(const P2 (&p1)[4], const P2 (&p2)[4]) {
  P2 q[24];
  float dist[24];
  return intersection_area_ws(p1, p2, P2View<1>{q}, F1View<1>{dist});
}

// A box prepared once: raw centre, size, half-trig, area.
struct BoxPrep {
  float x, y, w, h, c2, s2, area;
};

V3D_HD BoxPrep prep_box(const float* b) {
  BoxPrep r;
  r.x = b[0];
  r.y = b[1];
  r.w = b[2];
  r.h = b[3];
  half_trig(b[4], r.c2, r.s2);
  r.area = b[2] * b[3];
  return r;
}

// utils.h:313-340.  Early exit: when the centre distance exceeds the sum of the circumradii (with a
// 1% + 1e-3 margin) the rectangles are disjoint, the reference finds no intersection point and
// returns inter = 0, i.e. exactly 0/(a1+a2) = +0.0f -- the same value without running the clipper.
V3D_HD bool iou_needs_clip(const BoxPrep& a, const BoxPrep& b) {  // false <=> iou_prepped returns +0.0f without the clipper
  if (a.area < 1e-14 || b.area < 1e-14) return false;
  const float dx = a.x - b.x, dy = a.y - b.y;
  const float ra = 0.5f * sqrtf(a.w * a.w + a.h * a.h), rb = 0.5f * sqrtf(b.w * b.w + b.h * b.h);
  const float reach = (ra + rb) * 1.01f + 1e-3f;
  if (dx * dx + dy * dy > reach * reach && (a.area + b.area) > 0.f && (a.area + b.area) < 3.0e38f) return false;
  return true;
}

V3D_HD float iou_prepped(const BoxPrep& a, const BoxPrep& b) {
  if (!iou_needs_clip(a, b)) return 0.f;
  const double csx = (a.x + b.x) / 2.0;
  const double csy = (a.y + b.y) / 2.0;
  P2 p1[4], p2[4];
  vertices((float)(a.x - csx), (float)(a.y - csy), a.w, a.h, a.c2, a.s2, p1);
  vertices((float)(b.x - csx), (float)(b.y - csy), b.w, b.h, b.c2, b.s2, p2);
  const float inter = intersection_area(p1, p2);
  return inter / (a.area + b.area - inter);
}

// iou_prepped with the clipper's work arrays in a per-wave LDS slab: pts = slab of 24 * 64 P2 + this lane's index,
// dist = slab of 24 * 64 floats + this lane's index.  Same value, bit for bit.
V3D_HD float inter_prepped_lds(const BoxPrep& a, const BoxPrep& b, P2* pts, float* dist) {  // intersection AREA
  if (!iou_needs_clip(a, b)) return 0.f;
  const double csx = (a.x + b.x) / 2.0;
  const double csy = (a.y + b.y) / 2.0;
  P2 p1[4], p2[4];
  vertices((float)(a.x - csx), (float)(a.y - csy), a.w, a.h, a.c2, a.s2, p1);
  vertices((float)(b.x - csx), (float)(b.y - csy), b.w, b.h, b.c2, b.s2, p2);
  return intersection_area_ws(p1, p2, P2View<64>{pts}, F1View<64>{dist});
}

V3D_HD float iou_prepped_lds(const BoxPrep& a, const BoxPrep& b, P2* pts, float* dist) {
  if (!iou_needs_clip(a, b)) return 0.f;
  const float inter = inter_prepped_lds(a, b, pts, dist);
  return inter / (a.area + b.area - inter);
}

V3D_HD float single_box_iou_rotated(const float* b1, const float* b2) {
  const BoxPrep a = prep_box(b1), b = prep_box(b2);
  return iou_prepped(a, b);
}

}  // namespace v3d
