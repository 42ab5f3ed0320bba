// Evaluation-side consumer of the layout branch (include/omnipq_eval.h): quad proposals -> oriented thin boxes, their
// axis-aligned extents, the four F1 corners, the quad probability; greedy 3D NMS per scene.
// Reference: models/ap_helper_pq.py:323-460, utils/box_util.py:185-233, utils/nms.py:77-113.
#include "common.h"
#include "omnipq_eval.h"

namespace omnipq {

__global__ __launch_bounds__(256) void parse_quads_kernel(long long rows, const float *__restrict__ center,
                                                          const float *__restrict__ normal, const float *__restrict__ size,
                                                          const float *__restrict__ scores, float length,
                                                          double *__restrict__ corners8, double *__restrict__ aabb,
                                                          float *__restrict__ verts4, float *__restrict__ prob) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const float nx = normal[r * 3], ny = normal[r * 3 + 1], nz = normal[r * 3 + 2];
  const float cx = center[r * 3], cy = center[r * 3 + 1], cz = center[r * 3 + 2];
  const float width = size[r * 2], height = size[r * 2 + 1];
  const float norm = sqrtf(nx * nx + ny * ny + nz * nz);
  if (corners8 || aabb) {
    // torch.cosine_similarity(n, e) = sum (n / max(|n|, eps)) (e / max(|e|, eps)), eps = 1e-8, in f32 (:364-368)
    const float den = fmaxf(norm, 1e-8f);
    const float cos_y = ny / den, cos_x = nx / den;
    float heading = acosf(cos_y);
    if (cos_x > 0.f) heading = 6.283185307179586f - heading;                  // np.pi * 2 - f32 tensor -> f32
    const double c = (double)cosf(heading), s = (double)sinf(heading);        // roty on an f32 angle, then f64 arithmetic
    const double l = (double)width, w = (double)length, h = (double)height;   // box_size = [width, LENGTH, height]
    const double ccx = (double)cx, ccy = (double)(-cz), ccz = (double)cy;     // flip_axis_to_camera: (x, -z, y)
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const double x = ((i & 3) < 2 ? l : -l) / 2;                            // l/2, l/2, -l/2, -l/2, ...
      const double y = (i < 4 ? h : -h) / 2;
      const double z = ((i & 3) == 0 || (i & 3) == 3 ? w : -w) / 2;           // w/2, -w/2, -w/2, w/2, ...
      const double p[3] = {c * x + s * z + ccx, y + ccy, -s * x + c * z + ccz};
      for (int a = 0; a < 3; ++a) {
        if (corners8) corners8[(r * 8 + i) * 3 + a] = p[a];
        lo[a] = fmin(lo[a], p[a]);
        hi[a] = fmax(hi[a], p[a]);
      }
    }
    if (aabb)
      for (int a = 0; a < 3; ++a) {
        aabb[r * 6 + a] = lo[a];
        aabb[r * 6 + 3 + a] = hi[a];
      }
  }
  if (verts4) {
    // get_verts (:270-296): the normal scaled by 1 / max(|n|, 1e-6), all in f32
    const float den = fmaxf(norm, 1e-6f);
    const float ux = nx / den, uy = ny / den;
    const float x1 = cx + width * uy / 2.f, x2 = cx - width * uy / 2.f;
    const float y1 = cy - width * ux / 2.f, y2 = cy + width * ux / 2.f;
    const float h1 = cz + height / 2.f, h2 = cz - height / 2.f;
    float *o = verts4 + r * 12;
    o[0] = x1, o[1] = y1, o[2] = h1;
    o[3] = x2, o[4] = y2, o[5] = h1;
    o[6] = x1, o[7] = y1, o[8] = h2;
    o[9] = x2, o[10] = y2, o[11] = h2;
  }
  if (prob && scores) {
    const float a = scores[r * 2], b = scores[r * 2 + 1];
    const float mx = fmaxf(a, b);
    const float ea = expf(a - mx), eb = expf(b - mx);
    prob[r] = eb / (ea + eb);
  }
}


// corners (and extents) of n oriented boxes from their parameters: centre f32 (depth frame, flipped to upright camera here),
// size f64 (l, w, h) -- the class mean is kept in f64 by the reference's class2size -- heading f32
__global__ __launch_bounds__(256) void box_corners_kernel(long long n, const float *__restrict__ center,
                                                          const double *__restrict__ size,
                                                          const float *__restrict__ heading,
                                                          double *__restrict__ corners8, double *__restrict__ aabb) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const float ang = heading ? heading[r] : 0.f;
  const double c = (double)cosf(ang), s = (double)sinf(ang);
  const double l = size[r * 3], w = size[r * 3 + 1], h = size[r * 3 + 2];
  const double ccx = (double)center[r * 3], ccy = (double)(-center[r * 3 + 2]), ccz = (double)center[r * 3 + 1];
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const double x = ((i & 3) < 2 ? l : -l) / 2;
    const double y = (i < 4 ? h : -h) / 2;
    const double z = ((i & 3) == 0 || (i & 3) == 3 ? w : -w) / 2;
    const double p[3] = {c * x + s * z + ccx, y + ccy, -s * x + c * z + ccz};
    for (int a = 0; a < 3; ++a) {
      if (corners8) corners8[(r * 8 + i) * 3 + a] = p[a];
      lo[a] = fmin(lo[a], p[a]);
      hi[a] = fmax(hi[a], p[a]);
    }
  }
  if (aabb)
    for (int a = 0; a < 3; ++a) {
      aabb[r * 6 + a] = lo[a];
      aabb[r * 6 + 3 + a] = hi[a];
    }
}

// nonempty[b][k] = the box holds at least `min_points` of the scene's points.  One workgroup per box; a point is inside when
// its offset from the centre, rotated into the box frame, lies within half the size on every axis (what the reference asks
// of a Delaunay triangulation of the eight corners per box, models/utils/ap_util.py:4-13).
__global__ __launch_bounds__(256) void points_in_boxes_kernel(int n, int k, const float *__restrict__ xyz,
                                                              const float *__restrict__ center,
                                                              const double *__restrict__ size,
                                                              const float *__restrict__ heading, int min_points,
                                                              unsigned char *__restrict__ nonempty) {
  __shared__ int count;
  const int box = (int)blockIdx.x, b = (int)blockIdx.y;
  const size_t r = (size_t)b * k + box;
  if (threadIdx.x == 0) count = 0;
  __syncthreads();
  const float ang = heading ? heading[r] : 0.f;
  // depth frame: the box is rotated about z by -heading (camera y axis = -depth z, box_util.py:185-191)
  const float c = cosf(ang), s = sinf(ang);
  const float cx = center[r * 3], cy = center[r * 3 + 1], cz = center[r * 3 + 2];
  const float hl = (float)(size[r * 3] / 2), hw = (float)(size[r * 3 + 1] / 2), hh = (float)(size[r * 3 + 2] / 2);
  const float *pts = xyz + (size_t)b * n * 3;
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + (int)threadIdx.x;
    bool in = false;
    if (i < n) {
      const float dx = pts[i * 3] - cx, dy = pts[i * 3 + 1] - cy, dz = pts[i * 3 + 2] - cz;
      // camera frame: (x, y_c, z_c) = (dx, -dz, dy); box axes there: x' = c x - s z_c, z' = s x + c z_c
      const float u = c * dx - s * dy, v = s * dx + c * dy;
      in = fabsf(u) <= hl && fabsf(v) <= hw && fabsf(dz) <= hh;
    }
    if (in) atomicAdd(&count, 1);
    __syncthreads();
    if (count >= min_points) break;                          // uniform: count is read after the barrier
    __syncthreads();
  }
  __syncthreads();
  if (threadIdx.x == 0) nonempty[r] = count >= min_points ? 1 : 0;
}

constexpr int kNmsMax = 4096;
constexpr int kNmsThreads = 256;

// one workgroup per scene: rank the boxes by counting, then walk them in rank order; every lane owns k / 256 boxes and
// tests them against the box being kept
__global__ __launch_bounds__(kNmsThreads) void nms3d_kernel(int k, const double *__restrict__ aabb,
                                                            const float *__restrict__ score,
                                                            const unsigned char *__restrict__ valid, double thr,
                                                            int old_type, unsigned char *__restrict__ keep,
                                                            const int *__restrict__ cls) {
  __shared__ int order[kNmsMax];                 // order[rank] = box index, rank 0 = visited first
  __shared__ unsigned char dead[kNmsMax];
  __shared__ float sc[kNmsMax];
  __shared__ int n_live;
  const int b = (int)blockIdx.x;
  aabb += (size_t)b * k * 6;
  score += (size_t)b * k;
  keep += (size_t)b * k;
  if (valid) valid += (size_t)b * k;
  if (cls) cls += (size_t)b * k;
  if (threadIdx.x == 0) n_live = 0;
  for (int j = (int)threadIdx.x; j < k; j += kNmsThreads) {
    // a NaN score (a diverged checkpoint) must not break the ranking below, which needs a total order: it sorts last, like
    // the reference's np.argsort puts it, by standing for -inf (ties among them fall back to the index rule)
    const float sv = score[j];
    sc[j] = (sv != sv) ? -INFINITY : sv;
    dead[j] = valid ? (valid[j] ? 0 : 1) : 0;
    keep[j] = 0;
  }
  __syncthreads();
  for (int j = (int)threadIdx.x; j < k; j += kNmsThreads) {
    if (dead[j]) continue;
    int rank = 0;
    for (int i = 0; i < k; ++i)
      if (!dead[i] && (sc[i] > sc[j] || (sc[i] == sc[j] && i > j))) ++rank;
    order[rank] = j;
    atomicAdd(&n_live, 1);
  }
  __syncthreads();
  const int live = n_live;
  for (int t = 0; t < live; ++t) {
    const int i = order[t];
    if (dead[i]) continue;                       // uniform: dead[] only changes between barriers
    const double ax1 = aabb[i * 6], ay1 = aabb[i * 6 + 1], az1 = aabb[i * 6 + 2];
    const double ax2 = aabb[i * 6 + 3], ay2 = aabb[i * 6 + 4], az2 = aabb[i * 6 + 5];
    const double area_i = (ax2 - ax1) * (ay2 - ay1) * (az2 - az1);
    __syncthreads();                             // everyone has read dead[i] before it can change
    for (int j = (int)threadIdx.x; j < k; j += kNmsThreads) {
      if (dead[j]) continue;
      if (j == i) {
        keep[j] = 1;
        dead[j] = 1;
        continue;
      }
      const double l = fmax(0.0, fmin(ax2, aabb[j * 6 + 3]) - fmax(ax1, aabb[j * 6]));
      const double w = fmax(0.0, fmin(ay2, aabb[j * 6 + 4]) - fmax(ay1, aabb[j * 6 + 1]));
      const double h = fmax(0.0, fmin(az2, aabb[j * 6 + 5]) - fmax(az1, aabb[j * 6 + 2]));
      const double area_j = (aabb[j * 6 + 3] - aabb[j * 6]) * (aabb[j * 6 + 4] - aabb[j * 6 + 1]) *
                            (aabb[j * 6 + 5] - aabb[j * 6 + 2]);
      const double inter = l * w * h;
      const double o = old_type ? inter / area_j : inter / (area_i + area_j - inter);
      if (o > thr && (!cls || cls[j] == cls[i])) dead[j] = 1;      // same-class suppression only (nms_3d_faster_samecls)
    }
    __syncthreads();
  }
}

}  // namespace omnipq

using namespace omnipq;

extern "C" int omnipq_parse_quads(int b, int k, const float *quad_center, const float *normal_vector,
                                  const float *quad_size, const float *quad_scores, float length, double *corners8,
                                  double *aabb, float *verts4, float *prob, void *stream) {
  if (b < 0 || k < 0) return OMNIPQ_EINVAL;
  const long long rows = (long long)b * k;
  if (rows == 0) return OMNIPQ_OK;
  if (!quad_center || !normal_vector || !quad_size) return OMNIPQ_EINVAL;
  if (rows >= (1ll << 31)) return OMNIPQ_ETOOLARGE;
  parse_quads_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      rows, quad_center, normal_vector, quad_size, quad_scores, length, corners8, aabb, verts4, prob);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

static int nms3d_impl(int b, int k, const double *aabb, const float *score, const unsigned char *valid, const int *cls,
                      double overlap_threshold, int old_type, unsigned char *keep, void *stream) {
  if (b < 0 || k < 0) return OMNIPQ_EINVAL;
  if (k > kNmsMax) return OMNIPQ_ETOOLARGE;
  if (b == 0 || k == 0) return OMNIPQ_OK;
  if (!aabb || !score || !keep) return OMNIPQ_EINVAL;
  nms3d_kernel<<<b, kNmsThreads, 0, (hipStream_t)stream>>>(k, aabb, score, valid, overlap_threshold, old_type, keep, cls);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_nms3d(int b, int k, const double *aabb, const float *score, const unsigned char *valid,
                            double overlap_threshold, int old_type, unsigned char *keep, void *stream) {
  return nms3d_impl(b, k, aabb, score, valid, nullptr, overlap_threshold, old_type, keep, stream);
}

extern "C" int omnipq_nms3d_samecls(int b, int k, const double *aabb, const float *score, const unsigned char *valid,
                                    const int *cls, double overlap_threshold, int old_type, unsigned char *keep,
                                    void *stream) {
  if (b > 0 && k > 0 && !cls) return OMNIPQ_EINVAL;
  return nms3d_impl(b, k, aabb, score, valid, cls, overlap_threshold, old_type, keep, stream);
}

extern "C" int omnipq_box_corners(long long n, const float *center, const double *size, const float *heading,
                                  double *corners8, double *aabb, void *stream) {
  if (n < 0) return OMNIPQ_EINVAL;
  if (n == 0) return OMNIPQ_OK;
  if (!center || !size) return OMNIPQ_EINVAL;
  if (n >= (1ll << 31)) return OMNIPQ_ETOOLARGE;
  box_corners_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(n, center, size, heading, corners8, aabb);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_points_in_boxes(int b, int n, int k, const float *xyz, const float *center, const double *size,
                                      const float *heading, int min_points, unsigned char *nonempty, void *stream) {
  if (b < 0 || n < 0 || k < 0 || min_points < 1) return OMNIPQ_EINVAL;
  if (b == 0 || k == 0) return OMNIPQ_OK;
  if (!center || !size || !nonempty || (n > 0 && !xyz)) return OMNIPQ_EINVAL;
  if (b > 65535) return OMNIPQ_ETOOLARGE;
  points_in_boxes_kernel<<<dim3(k, b), 256, 0, (hipStream_t)stream>>>(n, k, xyz, center, size, heading, min_points, nonempty);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}
