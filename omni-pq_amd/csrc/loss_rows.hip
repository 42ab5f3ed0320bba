// The supervised loss `get_loss` of the reference (models/loss_helper_pq.py:412-486) as a handful of row kernels
// (C ABI and the reference lines each one replaces: include/omnipq_loss.h).
//
// The work is tiny (7 heads x B x 256 proposals x ~100 floats) and the reference spends it on ~2000 launches and thousands
// of host reads; what matters here is the launch count and that nothing goes through the host: one lane owns one
// (head, scene, proposal) row, reads its scores straight from the head outputs (no stacking copy: the descriptor carries
// one pointer per head), block-reduces in f64 and adds one f64 atomic per block and term.  Every backward twin recomputes
// the row from the same inputs and writes ALL gradient entries (zeros included), so the host side allocates with
// torch.empty and nothing is accumulated across launches.
#include "common.h"
#include "omnipq_loss.h"

namespace omnipq {

constexpr int kLossThreads = 256;

__device__ __forceinline__ double block_sum(double v, double *sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__device__ __forceinline__ float smoothl1(float e) {            // models/utils/losses.py:5-13, delta = 1
  const float a = fabsf(e);
  return a < 1.f ? 0.5f * a * a : a - 0.5f;
}
__device__ __forceinline__ float smoothl1_slope(float e) {      // autograd of the torch.where above
  const float a = fabsf(e);
  return a < 1.f ? e : (e > 0.f ? 1.f : -1.f);
}
__device__ __forceinline__ int clampi(long long v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : (int)v); }

// log-sum-exp of n contiguous scores (what log_softmax subtracts)
__device__ __forceinline__ float row_lse(const float *s, int n) {
  float mx = s[0];
  for (int c = 1; c < n; ++c) mx = fmaxf(mx, s[c]);
  float z = 0.f;
  for (int c = 0; c < n; ++c) z += expf(s[c] - mx);
  return mx + logf(z);
}
// gradient of coef * CE(s, y) into g (coef == 0: zeros)
__device__ __forceinline__ void ce_grad(const float *s, int n, int y, float lse, float coef, float *g) {
  for (int c = 0; c < n; ++c) g[c] = coef == 0.f ? 0.f : coef * (expf(s[c] - lse) - (c == y ? 1.f : 0.f));
}

// ------------------------------------------------------------------------------------------------ assignment
__global__ __launch_bounds__(kLossThreads) void loss_assign_kernel(int k, int k2, const float *__restrict__ query,
                                                                   const float *__restrict__ gt,
                                                                   const long long *__restrict__ num_gt, float near_thr,
                                                                   float far_thr, long long *__restrict__ label,
                                                                   float *__restrict__ mask, long long *__restrict__ assignment,
                                                                   float *__restrict__ counts) {
  constexpr int kTile = 512;
  __shared__ float tile[kTile * 3];
  __shared__ double sh[4];
  const int b = (int)blockIdx.y;
  const int i = (int)(blockIdx.x * kLossThreads + threadIdx.x);
  const bool in = i < k;
  const float *q = query + ((size_t)b * k + (in ? i : 0)) * 3;
  const float x = q[0], y = q[1], z = q[2];
  float best = INFINITY;
  int bi = 0;
  for (int j0 = 0; j0 < k2; j0 += kTile) {
    const int cnt = k2 - j0 < kTile ? k2 - j0 : kTile;
    __syncthreads();
    for (int t = (int)threadIdx.x; t < cnt * 3; t += kLossThreads) tile[t] = gt[((size_t)b * k2 + j0) * 3 + t];
    __syncthreads();
    for (int j = 0; j < cnt; ++j) {
      const float dx = x - tile[j * 3], dy = y - tile[j * 3 + 1], dz = z - tile[j * 3 + 2];
      float d = 0.f;                                       // the operation order of nn_distance (squares, then a running sum)
      d += dx * dx;
      d += dy * dy;
      d += dz * dz;
      if (d < best) {
        best = d;
        bi = j0 + j;
      }
    }
  }
  const float e = sqrtf(best + 1e-6f);
  const bool lab = in && e < near_thr && (long long)bi < num_gt[(size_t)b * k + (in ? i : 0)];
  const bool m = in && (e < near_thr || e > far_thr);
  if (in) {
    label[(size_t)b * k + i] = lab ? 1 : 0;
    mask[(size_t)b * k + i] = m ? 1.f : 0.f;
    assignment[(size_t)b * k + i] = lab ? bi : k2 - 1;
  }
  const double nl = block_sum(lab ? 1.0 : 0.0, sh);
  const double nm = block_sum(m ? 1.0 : 0.0, sh);
  if (threadIdx.x == 0) {                                  // integer-valued partial sums: exact in f32 whatever the order
    if (nl != 0.0) atomicAdd(counts, (float)nl);
    if (nm != 0.0) atomicAdd(counts + 1, (float)nm);
  }
}

// ------------------------------------------------------------------------------------------------ box rows
template <bool GRAD>
__global__ __launch_bounds__(kLossThreads) void box_rows_kernel(omnipq_box_rows_desc d, double *__restrict__ sums,
                                                                const float *__restrict__ g_terms,
                                                                omnipq_box_rows_grads g) {
  __shared__ double sh[4];
  const int h = (int)blockIdx.y;
  const long long rows = (long long)d.b * d.k;
  const long long r = (long long)blockIdx.x * kLossThreads + threadIdx.x;
  const bool in = r < rows;
  float t[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (in) {
    const bool lab = d.label[r] != 0;
    const float m = d.mask[r];
    const long long gi = d.only_objectness ? 0 : (r / d.k) * d.k2 + clampi(d.assignment[r], d.k2);
    const float inv_mask = GRAD ? 1.f / (d.counts[1] + 1e-6f) : 0.f;
    const float inv_pos = GRAD ? (lab ? 1.f / (d.counts[0] + 1e-6f) : 0.f) : 0.f;
    const float *gt = GRAD ? g_terms + h * 8 : nullptr;
    {  // objectness (:73-80): weighted two-class cross entropy under the NEAR / FAR mask
      const float *s = d.objectness_scores[h] + r * 2;
      const float lse = row_lse(s, 2);
      const float w = lab ? d.w_object : d.w_background;
      t[0] = w * (lse - s[lab ? 1 : 0]) * m;
      if (GRAD && g.objectness_scores[h]) ce_grad(s, 2, lab ? 1 : 0, lse, gt[0] * inv_mask * w * m, g.objectness_scores[h] + r * 2);
    }
    if (!d.only_objectness) {
      {  // centre (:112-117)
        const float *c = d.center[h] + r * 3;
        for (int a = 0; a < 3; ++a) {
          const float e = d.gt_center[gi * 3 + a] - c[a];
          if (lab) t[1] += smoothl1(e);
          if (GRAD && g.center[h]) g.center[h][r * 3 + a] = lab ? -gt[1] * inv_pos * smoothl1_slope(e) : 0.f;
        }
      }
      {  // heading class and residual (:119-143)
        const int hc = clampi(d.gt_heading_class[gi], d.nh);
        const float *s = d.heading_scores[h] + r * d.nh;
        const float lse = row_lse(s, d.nh);
        if (lab) t[2] = lse - s[hc];
        if (GRAD && g.heading_scores[h]) ce_grad(s, d.nh, hc, lse, gt[2] * inv_pos, g.heading_scores[h] + r * d.nh);
        const float want = d.gt_heading_residual[gi] / (3.14159265358979323846f / (float)d.nh);
        const float e = d.heading_residuals_normalized[h][r * d.nh + hc] - want;
        if (lab) t[3] = smoothl1(e);
        if (GRAD && g.heading_residuals_normalized[h])
          for (int c = 0; c < d.nh; ++c)
            g.heading_residuals_normalized[h][r * d.nh + c] = (lab && c == hc) ? gt[3] * inv_pos * smoothl1_slope(e) : 0.f;
      }
      {  // size class and residual (:145-172)
        const int sc = clampi(d.gt_size_class[gi], d.ns);
        const float *s = d.size_scores[h] + r * d.ns;
        const float lse = row_lse(s, d.ns);
        if (lab) t[4] = lse - s[sc];
        if (GRAD && g.size_scores[h]) ce_grad(s, d.ns, sc, lse, gt[4] * inv_pos, g.size_scores[h] + r * d.ns);
        const float *p = d.size_residuals_normalized[h] + (r * d.ns + sc) * 3;
        float slope[3];
        for (int a = 0; a < 3; ++a) {
          const float e = p[a] - d.gt_size_residual[gi * 3 + a] / d.mean_size[sc * 3 + a];
          if (lab) t[5] += smoothl1(e);
          slope[a] = smoothl1_slope(e);
        }
        if (GRAD && g.size_residuals_normalized[h]) {
          float *o = g.size_residuals_normalized[h] + r * d.ns * 3;
          for (int c = 0; c < d.ns * 3; ++c) o[c] = 0.f;
          if (lab)
            for (int a = 0; a < 3; ++a) o[sc * 3 + a] = gt[5] * inv_pos * slope[a];
        }
      }
      {  // semantic class (:174-178)
        const int y = clampi(d.gt_sem_cls[gi], d.nc);
        const float *s = d.sem_cls_scores[h] + r * d.nc;
        const float lse = row_lse(s, d.nc);
        if (lab) t[6] = lse - s[y];
        if (GRAD && g.sem_cls_scores[h]) ce_grad(s, d.nc, y, lse, gt[6] * inv_pos, g.sem_cls_scores[h] + r * d.nc);
      }
    }
  }
  if (!GRAD) {
    for (int i = 0; i < 7; ++i) {
      const double v = block_sum((double)t[i], sh);
      if (threadIdx.x == 0 && v != 0.0) atomicAdd(sums + h * 8 + i, v);
    }
  }
}

// terms[h][i] = f32(sum) / (count + 1e-6): column 0 over the mask count, the others over the label count
__global__ void rows_finalize_kernel(int n, const double *__restrict__ sums, const float *__restrict__ counts,
                                     float *__restrict__ terms) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  const float den = ((i & 7) == 0 ? counts[1] : counts[0]) + 1e-6f;
  terms[i] = (float)sums[i] / den;
}

// ------------------------------------------------------------------------------------------------ quad rows
template <bool GRAD>
__global__ __launch_bounds__(kLossThreads) void quad_rows_kernel(omnipq_quad_rows_desc d, double *__restrict__ sums,
                                                                 const float *__restrict__ g_terms,
                                                                 omnipq_quad_rows_grads g) {
  __shared__ double sh[4];
  const int h = (int)blockIdx.y;
  const long long rows = (long long)d.b * d.k;
  const long long r = (long long)blockIdx.x * kLossThreads + threadIdx.x;
  const bool in = r < rows;
  float t[4] = {0.f, 0.f, 0.f, 0.f};
  if (in) {
    const bool lab = d.label[r] != 0;
    const float m = d.mask[r];
    const long long gi = (r / d.k) * d.k2 + clampi(d.assignment[r], d.k2);
    const float inv_mask = GRAD ? 1.f / (d.counts[1] + 1e-6f) : 0.f;
    const float inv_pos = GRAD ? (lab ? 1.f / (d.counts[0] + 1e-6f) : 0.f) : 0.f;
    const float *gt = GRAD ? g_terms + h * 8 : nullptr;
    {  // is-a-quad score (:236-241)
      const float *s = d.quad_scores[h] + r * 2;
      const float lse = row_lse(s, 2);
      const float w = lab ? d.w_quad : d.w_background;
      t[0] = w * (lse - s[lab ? 1 : 0]) * m;
      if (GRAD && g.quad_scores[h]) ce_grad(s, 2, lab ? 1 : 0, lse, gt[0] * inv_mask * w * m, g.quad_scores[h] + r * 2);
    }
    {  // centre (:265-272)
      const float *c = d.quad_center[h] + r * 3;
      for (int a = 0; a < 3; ++a) {
        const float e = d.gt_center[gi * 3 + a] - c[a];
        if (lab) t[1] += smoothl1(e);
        if (GRAD && g.quad_center[h]) g.quad_center[h][r * 3 + a] = lab ? -gt[1] * inv_pos * smoothl1_slope(e) : 0.f;
      }
    }
    {  // normal (:274-283): 1 - cos, torch.cosine_similarity = sum (x / max(|x|, eps)) (y / max(|y|, eps))
      const float *x = d.normal_vector[h] + r * 3;
      const float *y = d.gt_normal + gi * 3;
      const float nx = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
      const float ny = sqrtf(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
      const float dx = fmaxf(nx, 1e-8f), dy = fmaxf(ny, 1e-8f);
      const float cs = (x[0] / dx) * (y[0] / dy) + (x[1] / dx) * (y[1] / dy) + (x[2] / dx) * (y[2] / dy);
      if (lab) t[2] = 1.f - cs;
      if (GRAD && g.normal_vector[h]) {
        for (int a = 0; a < 3; ++a) {
          // d cos / d x_a: through the numerator, and (when |x| > eps) through the norm in the denominator
          float dc = (y[a] / dy) / dx;
          if (nx > 1e-8f) dc -= cs * x[a] / (nx * nx);
          g.normal_vector[h][r * 3 + a] = lab ? -gt[2] * inv_pos * dc : 0.f;
        }
      }
    }
    {  // size (:285-293)
      const float *s = d.quad_size[h] + r * 2;
      for (int a = 0; a < 2; ++a) {
        const float e = s[a] - d.gt_size[gi * 2 + a];
        if (lab) t[3] += smoothl1(e);
        if (GRAD && g.quad_size[h]) g.quad_size[h][r * 2 + a] = lab ? gt[3] * inv_pos * smoothl1_slope(e) : 0.f;
      }
    }
  }
  if (!GRAD) {
    for (int i = 0; i < 4; ++i) {
      const double v = block_sum((double)t[i], sh);
      if (threadIdx.x == 0 && v != 0.0) atomicAdd(sums + h * 8 + i, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------ votes
constexpr int kMaxGtVotes = 8;

// one lane per seed; returns the seed's distance and (for the gradient) the vote / coordinate signs that attain it
template <bool GRAD>
__global__ __launch_bounds__(kLossThreads) void votes_kernel(long long seeds, int s, int n, int vf, int gv,
                                                             const float *__restrict__ seed_xyz,
                                                             const float *__restrict__ vote_xyz,
                                                             const int *__restrict__ seed_inds,
                                                             const float *__restrict__ vote_label,
                                                             const long long *__restrict__ vote_label_mask,
                                                             double *__restrict__ sums, const float *__restrict__ g_loss,
                                                             float *__restrict__ g_vote) {
  __shared__ double sh[4];
  const long long t = (long long)blockIdx.x * kLossThreads + threadIdx.x;
  const bool in = t < seeds;
  float dist = 0.f, m = 0.f;
  if (in) {
    const long long b = t / s;
    long long src = seed_inds[t];
    src = src < 0 ? 0 : (src >= n ? n - 1 : src);
    m = vote_label_mask[b * n + src] != 0 ? 1.f : 0.f;
    const float sx = seed_xyz[t * 3], sy = seed_xyz[t * 3 + 1], sz = seed_xyz[t * 3 + 2];
    float gx[kMaxGtVotes], gy[kMaxGtVotes], gz[kMaxGtVotes];
    const float *vl = vote_label + (b * n + src) * 3 * gv;
    for (int j = 0; j < kMaxGtVotes; ++j)
      if (j < gv) {
        gx[j] = vl[j * 3] + sx;
        gy[j] = vl[j * 3 + 1] + sy;
        gz[j] = vl[j * 3 + 2] + sz;
      }
    // dist2[j] = min_i |vote_i - gt_j|_1 (first i on ties), then min_j (first j on ties): torch.min twice (:39-40)
    float best = INFINITY;
    int bi = 0, bj = 0;
    for (int j = 0; j < gv; ++j) {
      float dj = INFINITY;
      int ij = 0;
      for (int i = 0; i < vf; ++i) {
        const float *v = vote_xyz + (t * vf + i) * 3;
        float dd = 0.f;
        dd += fabsf(v[0] - gx[j]);
        dd += fabsf(v[1] - gy[j]);
        dd += fabsf(v[2] - gz[j]);
        if (dd < dj) {
          dj = dd;
          ij = i;
        }
      }
      if (dj < best) {
        best = dj;
        bi = ij;
        bj = j;
      }
    }
    dist = best;
    if (GRAD) {
      const float coef = g_loss[0] * m / ((float)sums[1] + 1e-6f);
      for (int i = 0; i < vf; ++i) {
        const float *v = vote_xyz + (t * vf + i) * 3;
        float *o = g_vote + (t * vf + i) * 3;
        const float ex = v[0] - gx[bj], ey = v[1] - gy[bj], ez = v[2] - gz[bj];
        const bool hit = i == bi;
        o[0] = hit ? coef * (ex > 0.f ? 1.f : (ex < 0.f ? -1.f : 0.f)) : 0.f;
        o[1] = hit ? coef * (ey > 0.f ? 1.f : (ey < 0.f ? -1.f : 0.f)) : 0.f;
        o[2] = hit ? coef * (ez > 0.f ? 1.f : (ez < 0.f ? -1.f : 0.f)) : 0.f;
      }
    }
  }
  if (!GRAD) {
    const double a = block_sum((double)(dist * m), sh);
    const double c = block_sum((double)m, sh);
    if (threadIdx.x == 0) {
      if (a != 0.0) atomicAdd(sums, a);
      if (c != 0.0) atomicAdd(sums + 1, c);
    }
  }
}

__global__ void votes_finalize_kernel(const double *__restrict__ sums, float *__restrict__ loss) {
  loss[0] = (float)sums[0] / ((float)sums[1] + 1e-6f);
}

// ------------------------------------------------------------------------------------------------ physical constraints
// pre-pass, one lane per box: footprint (centre x, y, half length, half width) + whether it takes part; n_box per scene
__global__ __launch_bounds__(kLossThreads) void pc_boxes_kernel(omnipq_pc_desc d, float *__restrict__ boxes,
                                                                int *__restrict__ cls_out, float *__restrict__ n_box) {
  __shared__ double sh[4];
  const int b = (int)blockIdx.y;
  const int i = (int)(blockIdx.x * kLossThreads + threadIdx.x);
  bool use = false;
  if (i < d.k) {
    const size_t r = (size_t)b * d.k + i;
    const float *s = d.size_scores + r * d.ns;
    int cls = 0;
    float mx = s[0];
    for (int c = 1; c < d.ns; ++c)
      if (s[c] > mx) {                                           // torch.argmax: first maximum
        mx = s[c];
        cls = c;
      }
    const float *res = d.size_residuals + (r * d.ns + cls) * 3;
    const double l = d.mean_size64[cls * 3] + (double)res[0];    // box_size is f64 in the reference (:383)
    const double w = d.mean_size64[cls * 3 + 1] + (double)res[1];
    const long long sem = d.sem_cls_label[(size_t)b * d.k2 + clampi(d.object_assignment[r], d.k2)];
    const bool solid = !(sem >= 0 && sem < 64 && ((d.not_solid >> sem) & 1ull));
    use = d.object_label[r] != 0 && solid;
    float *o = boxes + r * 4;
    o[0] = d.center[r * 3];
    o[1] = d.center[r * 3 + 1];
    o[2] = (float)(l / 2);                                       // rounded to f32 when stored in the corner tensor (:313-316)
    o[3] = (float)(w / 2);
    cls_out[r] = use ? cls : -1 - cls;
  }
  const double cnt = block_sum(use ? 1.0 : 0.0, sh);
  if (threadIdx.x == 0 && cnt != 0.0) atomicAdd(n_box + b, (float)cnt);
}

// depth of a footprint corner behind the quad's line, gated by the projection test (:323-353); returns -delta (> 0) or 0
__device__ __forceinline__ float pc_pair(float px, float py, float a, float bq, float dq, float cx, float cy, float half) {
  const float delta = (px * a + py * bq) + dq;
  if (!(delta < 0.f)) return 0.f;
  const float kk = -delta;
  const float tx = px + a * kk, ty = py + bq * kk;
  const float ex = tx - cx, ey = ty - cy;
  const float w = sqrtf(ex * ex + ey * ey);
  return w < half ? kk : 0.f;
}

// forward: block per (quad, scene), lanes over the boxes
__global__ __launch_bounds__(kLossThreads) void pc_fwd_kernel(omnipq_pc_desc d, const float *__restrict__ boxes,
                                                              const int *__restrict__ cls, const float *__restrict__ n_box,
                                                              double *__restrict__ sums) {
  __shared__ double sh[4];
  const int q = (int)blockIdx.x, b = (int)blockIdx.y;
  const size_t qr = (size_t)b * d.q + q;
  if (d.quad_label[qr] == 0 || n_box[b] == 0.f) return;          // uniform over the block
  const float a = d.normal_vector[qr * 3], bq = d.normal_vector[qr * 3 + 1];
  const float cx = d.quad_center[qr * 3], cy = d.quad_center[qr * 3 + 1];
  const float dq = -(a * cx + bq * cy);
  const float half = d.quad_size[qr * 2];
  float acc = 0.f;
  int hits = 0;
  for (int i = (int)threadIdx.x; i < d.k; i += kLossThreads) {
    const size_t r = (size_t)b * d.k + i;
    if (cls[r] < 0) continue;
    const float bx = boxes[r * 4], by = boxes[r * 4 + 1], hl = boxes[r * 4 + 2], hw = boxes[r * 4 + 3];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float px = (c < 2 ? hl : -hl) + bx, py = ((c & 1) ? -hw : hw) + by;
      const float pen = pc_pair(px, py, a, bq, dq, cx, cy, half);
      acc += pen;
      hits += pen > 1e-4f ? 1 : 0;
    }
  }
  const double tot = block_sum((double)acc, sh);
  const double cnt = block_sum((double)hits, sh);
  if (threadIdx.x == 0) {
    if (tot != 0.0) atomicAdd(sums, (double)((float)tot / n_box[b]));        // loss / num_box per quad (:405)
    if (cnt != 0.0) atomicAdd(sums + 1, cnt);
  }
}

__global__ void pc_finalize_kernel(const double *__restrict__ sums, float *__restrict__ out) {
  out[0] = (float)sums[0];
  out[1] = (float)sums[1];
}

// backward, quad side: block per (quad, scene) reduces over the boxes; pen = -delta on live pairs, so
//   d pen / d a = -(px - cx),  d pen / d b = -(py - cy),  d pen / d cx = a,  d pen / d cy = b
__global__ __launch_bounds__(kLossThreads) void pc_bwd_quads_kernel(omnipq_pc_desc d, const float *__restrict__ boxes,
                                                                    const int *__restrict__ cls,
                                                                    const float *__restrict__ n_box,
                                                                    const float *__restrict__ g_out,
                                                                    float *__restrict__ g_quad_center,
                                                                    float *__restrict__ g_normal) {
  __shared__ double sh[4];
  const int q = (int)blockIdx.x, b = (int)blockIdx.y;
  const size_t qr = (size_t)b * d.q + q;
  const bool live = d.quad_label[qr] != 0 && n_box[b] != 0.f;
  float ga = 0.f, gb = 0.f, cnt = 0.f;
  float a = 0.f, bq = 0.f;
  if (live) {
    a = d.normal_vector[qr * 3];
    bq = d.normal_vector[qr * 3 + 1];
    const float cx = d.quad_center[qr * 3], cy = d.quad_center[qr * 3 + 1];
    const float dq = -(a * cx + bq * cy);
    const float half = d.quad_size[qr * 2];
    for (int i = (int)threadIdx.x; i < d.k; i += kLossThreads) {
      const size_t r = (size_t)b * d.k + i;
      if (cls[r] < 0) continue;
      const float bx = boxes[r * 4], by = boxes[r * 4 + 1], hl = boxes[r * 4 + 2], hw = boxes[r * 4 + 3];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float px = (c < 2 ? hl : -hl) + bx, py = ((c & 1) ? -hw : hw) + by;
        if (pc_pair(px, py, a, bq, dq, cx, cy, half) > 0.f) {
          ga -= px - cx;
          gb -= py - cy;
          cnt += 1.f;
        }
      }
    }
  }
  const double sa = block_sum((double)ga, sh);
  const double sb = block_sum((double)gb, sh);
  const double sc = block_sum((double)cnt, sh);
  if (threadIdx.x == 0) {
    const float scale = live ? g_out[0] / n_box[b] : 0.f;
    g_normal[qr * 3] = scale * (float)sa;
    g_normal[qr * 3 + 1] = scale * (float)sb;
    g_normal[qr * 3 + 2] = 0.f;
    g_quad_center[qr * 3] = scale * (float)sc * a;
    g_quad_center[qr * 3 + 1] = scale * (float)sc * bq;
    g_quad_center[qr * 3 + 2] = 0.f;
  }
}

// backward, box side: 64 boxes per block, the four waves split the scene's quads (staged in LDS) between them and fold
// their partial sums through LDS: d pen / d px = -a, d pen / d py = -b; px = cx +- l / 2 so the size gradient is +-1/2 of
// the corner's
constexpr int kPcBoxes = 64;

__global__ __launch_bounds__(kLossThreads) void pc_bwd_boxes_kernel(omnipq_pc_desc d, const float *__restrict__ boxes,
                                                                    const int *__restrict__ cls,
                                                                    const float *__restrict__ n_box,
                                                                    const float *__restrict__ g_out,
                                                                    float *__restrict__ g_center,
                                                                    float *__restrict__ g_size_residuals) {
  constexpr int kTile = 256;
  __shared__ float qs[kTile * 6];                                // a, b, d, cx, cy, half (half < 0: not a quad)
  __shared__ float part[4][kPcBoxes][4];
  const int b = (int)blockIdx.y;
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  const int i = (int)blockIdx.x * kPcBoxes + lane;
  const bool in = i < d.k;
  const size_t r = (size_t)b * d.k + (in ? i : 0);
  const int c0 = cls[r];
  const bool use = in && c0 >= 0 && n_box[b] != 0.f;
  const float bx = boxes[r * 4], by = boxes[r * 4 + 1], hl = boxes[r * 4 + 2], hw = boxes[r * 4 + 3];
  float gx = 0.f, gy = 0.f, gl = 0.f, gw = 0.f;
  for (int q0 = 0; q0 < d.q; q0 += kTile) {
    const int cnt = d.q - q0 < kTile ? d.q - q0 : kTile;
    __syncthreads();
    for (int t = (int)threadIdx.x; t < cnt; t += kLossThreads) {
      const size_t qr = (size_t)b * d.q + q0 + t;
      const float a = d.normal_vector[qr * 3], bq = d.normal_vector[qr * 3 + 1];
      const float cx = d.quad_center[qr * 3], cy = d.quad_center[qr * 3 + 1];
      qs[t * 6] = a;
      qs[t * 6 + 1] = bq;
      qs[t * 6 + 2] = -(a * cx + bq * cy);
      qs[t * 6 + 3] = cx;
      qs[t * 6 + 4] = cy;
      qs[t * 6 + 5] = d.quad_label[qr] != 0 ? d.quad_size[qr * 2] : -1.f;
    }
    __syncthreads();
    if (use)
      for (int t = wave; t < cnt; t += 4) {
        const float a = qs[t * 6], bq = qs[t * 6 + 1], dq = qs[t * 6 + 2], cx = qs[t * 6 + 3], cy = qs[t * 6 + 4];
        const float half = qs[t * 6 + 5];
        if (half < 0.f) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float sx = c < 2 ? 1.f : -1.f, sy = (c & 1) ? -1.f : 1.f;
          if (pc_pair(sx * hl + bx, sy * hw + by, a, bq, dq, cx, cy, half) > 0.f) {
            gx -= a;
            gy -= bq;
            gl -= 0.5f * sx * a;
            gw -= 0.5f * sy * bq;
          }
        }
      }
  }
  part[wave][lane][0] = gx;
  part[wave][lane][1] = gy;
  part[wave][lane][2] = gl;
  part[wave][lane][3] = gw;
  __syncthreads();
  if (wave != 0 || !in) return;
  float tot[4];
  for (int c = 0; c < 4; ++c) tot[c] = ((part[0][lane][c] + part[1][lane][c]) + part[2][lane][c]) + part[3][lane][c];
  const float scale = use ? g_out[0] / n_box[b] : 0.f;
  g_center[r * 3] = scale * tot[0];
  g_center[r * 3 + 1] = scale * tot[1];
  g_center[r * 3 + 2] = 0.f;
  float *o = g_size_residuals + r * d.ns * 3;
  for (int c = 0; c < d.ns * 3; ++c) o[c] = 0.f;
  const int cl = c0 >= 0 ? c0 : -1 - c0;
  o[cl * 3] = scale * tot[2];
  o[cl * 3 + 1] = scale * tot[3];
}

static inline unsigned blocks_for(long long n) { return (unsigned)((n + kLossThreads - 1) / kLossThreads); }

static int check_box_desc(const omnipq_box_rows_desc *d) {
  if (!d || d->heads < 1 || d->heads > OMNIPQ_LOSS_MAX_HEADS || d->b < 0 || d->k < 0 || d->k2 < 1) return OMNIPQ_EINVAL;
  if (d->nh < 1 || d->nh > 64 || d->ns < 1 || d->ns > 64 || d->nc < 1 || d->nc > 64) return OMNIPQ_EINVAL;
  if ((long long)d->b * d->k >= (1ll << 31) || d->b > 65535) return OMNIPQ_ETOOLARGE;
  for (int h = 0; h < d->heads; ++h) {
    if (!d->objectness_scores[h]) return OMNIPQ_EINVAL;
    if (!d->only_objectness && (!d->center[h] || !d->heading_scores[h] || !d->heading_residuals_normalized[h] ||
                                !d->size_scores[h] || !d->size_residuals_normalized[h] || !d->sem_cls_scores[h]))
      return OMNIPQ_EINVAL;
  }
  if (!d->label || !d->mask || !d->counts) return OMNIPQ_EINVAL;
  if (!d->only_objectness && (!d->assignment || !d->gt_center || !d->gt_heading_class || !d->gt_heading_residual ||
                              !d->gt_size_class || !d->gt_size_residual || !d->gt_sem_cls || !d->mean_size))
    return OMNIPQ_EINVAL;
  return OMNIPQ_OK;
}

static int check_quad_desc(const omnipq_quad_rows_desc *d) {
  if (!d || d->heads < 1 || d->heads > OMNIPQ_LOSS_MAX_HEADS || d->b < 0 || d->k < 0 || d->k2 < 1) return OMNIPQ_EINVAL;
  if ((long long)d->b * d->k >= (1ll << 31) || d->b > 65535) return OMNIPQ_ETOOLARGE;
  for (int h = 0; h < d->heads; ++h)
    if (!d->quad_scores[h] || !d->quad_center[h] || !d->normal_vector[h] || !d->quad_size[h]) return OMNIPQ_EINVAL;
  if (!d->label || !d->mask || !d->assignment || !d->counts || !d->gt_center || !d->gt_normal || !d->gt_size)
    return OMNIPQ_EINVAL;
  return OMNIPQ_OK;
}

static int check_pc_desc(const omnipq_pc_desc *d) {
  if (!d || d->b < 0 || d->k < 0 || d->q < 0 || d->k2 < 1 || d->ns < 1 || d->ns > 64) return OMNIPQ_EINVAL;
  if (d->b > 65535 || d->q > 65535 * 32) return OMNIPQ_ETOOLARGE;
  if (!d->center || !d->size_scores || !d->size_residuals || !d->object_label || !d->object_assignment ||
      !d->sem_cls_label || !d->mean_size64 || !d->quad_center || !d->normal_vector || !d->quad_size || !d->quad_label)
    return OMNIPQ_EINVAL;
  return OMNIPQ_OK;
}

}  // namespace omnipq

using namespace omnipq;

extern "C" int omnipq_loss_assign(int b, int k, int k2, const float *query, const float *gt, const long long *num_gt,
                                  float near_thr, float far_thr, long long *label, float *mask, long long *assignment,
                                  float *counts, void *stream) {
  if (b < 0 || k < 0 || k2 < 1 || b > 65535) return OMNIPQ_EINVAL;
  if (!counts) return OMNIPQ_EINVAL;
  OMNIPQ_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(float), (hipStream_t)stream));
  if (b == 0 || k == 0) return OMNIPQ_OK;
  if (!query || !gt || !num_gt || !label || !mask || !assignment) return OMNIPQ_EINVAL;
  loss_assign_kernel<<<dim3(blocks_for(k), b), kLossThreads, 0, (hipStream_t)stream>>>(k, k2, query, gt, num_gt, near_thr,
                                                                                      far_thr, label, mask, assignment, counts);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_loss_box_rows(const omnipq_box_rows_desc *d, double *sums, float *terms, void *stream) {
  const int rc = check_box_desc(d);
  if (rc != OMNIPQ_OK) return rc;
  if (!sums || !terms) return OMNIPQ_EINVAL;
  OMNIPQ_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 8 * d->heads, (hipStream_t)stream));
  const long long rows = (long long)d->b * d->k;
  if (rows > 0) {
    box_rows_kernel<false><<<dim3(blocks_for(rows), d->heads), kLossThreads, 0, (hipStream_t)stream>>>(
        *d, sums, nullptr, omnipq_box_rows_grads{});
    OMNIPQ_LAUNCH_CHECK();
  }
  rows_finalize_kernel<<<1, 64, 0, (hipStream_t)stream>>>(8 * d->heads, sums, d->counts, terms);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_loss_box_rows_grad(const omnipq_box_rows_desc *d, const float *g_terms,
                                         const omnipq_box_rows_grads *g, void *stream) {
  const int rc = check_box_desc(d);
  if (rc != OMNIPQ_OK) return rc;
  if (!g_terms || !g) return OMNIPQ_EINVAL;
  const long long rows = (long long)d->b * d->k;
  if (rows == 0) return OMNIPQ_OK;
  box_rows_kernel<true><<<dim3(blocks_for(rows), d->heads), kLossThreads, 0, (hipStream_t)stream>>>(*d, nullptr, g_terms, *g);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_loss_quad_rows(const omnipq_quad_rows_desc *d, double *sums, float *terms, void *stream) {
  const int rc = check_quad_desc(d);
  if (rc != OMNIPQ_OK) return rc;
  if (!sums || !terms) return OMNIPQ_EINVAL;
  OMNIPQ_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 8 * d->heads, (hipStream_t)stream));
  const long long rows = (long long)d->b * d->k;
  if (rows > 0) {
    quad_rows_kernel<false><<<dim3(blocks_for(rows), d->heads), kLossThreads, 0, (hipStream_t)stream>>>(
        *d, sums, nullptr, omnipq_quad_rows_grads{});
    OMNIPQ_LAUNCH_CHECK();
  }
  rows_finalize_kernel<<<1, 64, 0, (hipStream_t)stream>>>(8 * d->heads, sums, d->counts, terms);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_loss_quad_rows_grad(const omnipq_quad_rows_desc *d, const float *g_terms,
                                          const omnipq_quad_rows_grads *g, void *stream) {
  const int rc = check_quad_desc(d);
  if (rc != OMNIPQ_OK) return rc;
  if (!g_terms || !g) return OMNIPQ_EINVAL;
  const long long rows = (long long)d->b * d->k;
  if (rows == 0) return OMNIPQ_OK;
  quad_rows_kernel<true><<<dim3(blocks_for(rows), d->heads), kLossThreads, 0, (hipStream_t)stream>>>(*d, nullptr, g_terms, *g);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

static int check_votes(int b, int s, int n, int vf, int gv) {
  if (b < 0 || s < 0 || n < 1 || vf < 1 || gv < 1 || gv > kMaxGtVotes) return OMNIPQ_EINVAL;
  if ((long long)b * s * vf >= (1ll << 31)) return OMNIPQ_ETOOLARGE;
  return OMNIPQ_OK;
}

extern "C" int omnipq_loss_votes(int b, int s, int n, int vote_factor, int gt_votes, const float *seed_xyz,
                                 const float *vote_xyz, const int *seed_inds, const float *vote_label,
                                 const long long *vote_label_mask, double *sums, float *loss, void *stream) {
  const int rc = check_votes(b, s, n, vote_factor, gt_votes);
  if (rc != OMNIPQ_OK) return rc;
  if (!sums || !loss) return OMNIPQ_EINVAL;
  OMNIPQ_HIP(hipMemsetAsync(sums, 0, 2 * sizeof(double), (hipStream_t)stream));
  const long long seeds = (long long)b * s;
  if (seeds > 0) {
    if (!seed_xyz || !vote_xyz || !seed_inds || !vote_label || !vote_label_mask) return OMNIPQ_EINVAL;
    votes_kernel<false><<<blocks_for(seeds), kLossThreads, 0, (hipStream_t)stream>>>(
        seeds, s, n, vote_factor, gt_votes, seed_xyz, vote_xyz, seed_inds, vote_label, vote_label_mask, sums, nullptr, nullptr);
    OMNIPQ_LAUNCH_CHECK();
  }
  votes_finalize_kernel<<<1, 1, 0, (hipStream_t)stream>>>(sums, loss);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_loss_votes_grad(int b, int s, int n, int vote_factor, int gt_votes, const float *seed_xyz,
                                      const float *vote_xyz, const int *seed_inds, const float *vote_label,
                                      const long long *vote_label_mask, const double *sums, const float *g_loss,
                                      float *g_vote_xyz, void *stream) {
  const int rc = check_votes(b, s, n, vote_factor, gt_votes);
  if (rc != OMNIPQ_OK) return rc;
  const long long seeds = (long long)b * s;
  if (seeds == 0) return OMNIPQ_OK;
  if (!seed_xyz || !vote_xyz || !seed_inds || !vote_label || !vote_label_mask || !sums || !g_loss || !g_vote_xyz)
    return OMNIPQ_EINVAL;
  votes_kernel<true><<<blocks_for(seeds), kLossThreads, 0, (hipStream_t)stream>>>(
      seeds, s, n, vote_factor, gt_votes, seed_xyz, vote_xyz, seed_inds, vote_label, vote_label_mask,
      const_cast<double *>(sums), g_loss, g_vote_xyz);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

// workspace layout (floats): boxes [b][k][4] | class / use [b][k] (int) | n_box [b]
extern "C" long long omnipq_loss_physical_workspace_floats(int b, int k) {
  if (b < 0 || k < 0) return 0;
  return (long long)b * k * 5 + b;
}

static int pc_prepass(const omnipq_pc_desc *d, float *ws, float **boxes, int **cls, float **n_box, hipStream_t stream) {
  *boxes = ws;
  *cls = reinterpret_cast<int *>(ws + (size_t)d->b * d->k * 4);
  *n_box = ws + (size_t)d->b * d->k * 5;
  OMNIPQ_HIP(hipMemsetAsync(*n_box, 0, sizeof(float) * d->b, stream));
  pc_boxes_kernel<<<dim3(blocks_for(d->k), d->b), kLossThreads, 0, stream>>>(*d, *boxes, *cls, *n_box);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_loss_physical(const omnipq_pc_desc *d, float *workspace, double *sums, float *out, void *stream) {
  const int rc = check_pc_desc(d);
  if (rc != OMNIPQ_OK) return rc;
  if (!sums || !out) return OMNIPQ_EINVAL;
  OMNIPQ_HIP(hipMemsetAsync(sums, 0, 2 * sizeof(double), (hipStream_t)stream));
  if (d->b > 0 && d->k > 0 && d->q > 0) {
    if (!workspace) return OMNIPQ_EINVAL;
    float *boxes, *n_box;
    int *cls;
    const int rc2 = pc_prepass(d, workspace, &boxes, &cls, &n_box, (hipStream_t)stream);
    if (rc2 != OMNIPQ_OK) return rc2;
    pc_fwd_kernel<<<dim3(d->q, d->b), kLossThreads, 0, (hipStream_t)stream>>>(*d, boxes, cls, n_box, sums);
    OMNIPQ_LAUNCH_CHECK();
  }
  pc_finalize_kernel<<<1, 1, 0, (hipStream_t)stream>>>(sums, out);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}

extern "C" int omnipq_loss_physical_grad(const omnipq_pc_desc *d, float *workspace, const float *g_out, float *g_center,
                                         float *g_size_residuals, float *g_quad_center, float *g_normal, void *stream) {
  const int rc = check_pc_desc(d);
  if (rc != OMNIPQ_OK) return rc;
  if (d->b == 0) return OMNIPQ_OK;
  if (!g_out || !g_center || !g_size_residuals || !g_quad_center || !g_normal) return OMNIPQ_EINVAL;
  if (d->k == 0 || d->q == 0) {
    if (d->k > 0) {
      OMNIPQ_HIP(hipMemsetAsync(g_center, 0, sizeof(float) * (size_t)d->b * d->k * 3, (hipStream_t)stream));
      OMNIPQ_HIP(hipMemsetAsync(g_size_residuals, 0, sizeof(float) * (size_t)d->b * d->k * d->ns * 3, (hipStream_t)stream));
    }
    if (d->q > 0) {
      OMNIPQ_HIP(hipMemsetAsync(g_quad_center, 0, sizeof(float) * (size_t)d->b * d->q * 3, (hipStream_t)stream));
      OMNIPQ_HIP(hipMemsetAsync(g_normal, 0, sizeof(float) * (size_t)d->b * d->q * 3, (hipStream_t)stream));
    }
    return OMNIPQ_OK;
  }
  if (!workspace) return OMNIPQ_EINVAL;
  float *boxes, *n_box;
  int *cls;
  const int rc2 = pc_prepass(d, workspace, &boxes, &cls, &n_box, (hipStream_t)stream);
  if (rc2 != OMNIPQ_OK) return rc2;
  pc_bwd_quads_kernel<<<dim3(d->q, d->b), kLossThreads, 0, (hipStream_t)stream>>>(*d, boxes, cls, n_box, g_out,
                                                                                 g_quad_center, g_normal);
  OMNIPQ_LAUNCH_CHECK();
  pc_bwd_boxes_kernel<<<dim3((d->k + kPcBoxes - 1) / kPcBoxes, d->b), kLossThreads, 0, (hipStream_t)stream>>>(*d, boxes, cls, n_box, g_out,
                                                                                             g_center, g_size_residuals);
  OMNIPQ_LAUNCH_CHECK();
  return OMNIPQ_OK;
}
