// Hungarian matcher on the device: per-image cost matrix + batched rectangular linear-sum-assignment.
// Replaces training/hungarian_matcher.py:34-85 (softmax / cdist / GIoU cost on the GPU, then a
// D2H copy and scipy.optimize.linear_sum_assignment per image on the host).
//
// LSAP: one 256-thread workgroup per image runs the shortest-augmenting-path algorithm of
// Crouse 2016 (what scipy 1.15.3 implements) in fp64, with every piece of per-image state in
// LDS.  The scan over the remaining columns is parallel; the arg-min reduction reproduces the
// sequential tie rule exactly (strictly smaller wins; among equal values the LAST unassigned
// column in scan order, else the FIRST column), and the swap-remove on the `remaining` list is the
// same, so the indices are bit-exact with scipy / oracle/lsap.c, ties included.  No MFMA here:
// this is latency-bound integer/compare work, not a contraction.
#include "am_common.h"

namespace {

// D = 4: boxes are cxcywh, 2-D GIoU.  D = 7: [cx, cy, cz, w, l, h, yaw] -- the L1 term runs over all seven numbers, the
// GIoU term is the reference's axis-aligned BEV approximation on (cx -+ w/2, cy -+ l/2) (hungarian_matcher.py:52-66).
// Any other D: L1 + class only (GIoU term zero, hungarian_matcher.py:67-68).
template <int D>
__global__ __launch_bounds__(256) void match_cost_k(const float* __restrict__ logits, const float* __restrict__ boxes,
                                                    const long long* __restrict__ tgt_labels, const float* __restrict__ tgt_boxes,
                                                    const int* __restrict__ n_tgt, int Q, int C, int Nmax, int Dr, float w_class,
                                                    float w_bbox, float w_giou, float* __restrict__ cost) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= Q) return;
  const int dd = D > 0 ? D : Dr;
  const int ni = n_tgt[b];
  const float* lg = logits + ((size_t)b * Q + q) * C;
  float mx = -INFINITY;
  for (int c = 0; c < C; ++c) mx = fmaxf(mx, lg[c]);
  float se = 0.f;
  for (int c = 0; c < C; ++c) se += expf(lg[c] - mx);
  const float* pb = boxes + ((size_t)b * Q + q) * dd;
  constexpr bool HAS_GIOU = D == 4 || D == 7;
  // the two box extents the GIoU term uses: (w, h) for D = 4, (w, l) for D = 7
  const float pcx = pb[0], pcy = pb[1], pw = HAS_GIOU ? pb[D == 4 ? 2 : 3] : 0.f, ph = HAS_GIOU ? pb[D == 4 ? 3 : 4] : 0.f;
  const float px1 = pcx - 0.5f * pw, py1 = pcy - 0.5f * ph, px2 = pcx + 0.5f * pw, py2 = pcy + 0.5f * ph;
  const float parea = (px2 - px1) * (py2 - py1);
  for (int j = 0; j < ni; ++j) {
    const long long lab = tgt_labels[(size_t)b * Nmax + j];
    const float prob = (lab >= 0 && lab < C) ? expf(lg[lab] - mx) / se : 0.f;
    const float* tb = tgt_boxes + ((size_t)b * Nmax + j) * dd;
    float l1 = 0.f;
    for (int k = 0; k < dd; ++k) l1 += fabsf(pb[k] - tb[k]);  // torch.cdist(p=1): sum over the box dimension in order
    float giou = 0.f;
    if (HAS_GIOU && w_giou > 0.f) {
      const float tcx = tb[0], tcy = tb[1], tw = tb[D == 4 ? 2 : 3], th = tb[D == 4 ? 3 : 4];
      const float tx1 = tcx - 0.5f * tw, ty1 = tcy - 0.5f * th, tx2 = tcx + 0.5f * tw, ty2 = tcy + 0.5f * th;
      const float tarea = (tx2 - tx1) * (ty2 - ty1);
      const float iw = fmaxf(fminf(px2, tx2) - fmaxf(px1, tx1), 0.f), ih = fmaxf(fminf(py2, ty2) - fmaxf(py1, ty1), 0.f);
      const float inter = iw * ih;
      const float uni = parea + tarea - inter;
      const float cw = fmaxf(fmaxf(px2, tx2) - fminf(px1, tx1), 0.f), chh = fmaxf(fmaxf(py2, ty2) - fminf(py1, ty1), 0.f);
      const float carea = cw * chh;
      giou = inter / uni - (carea - uni) / carea;
    }
    // reference summation order (hungarian_matcher.py:73-75): bbox, class, giou
    const float cval = (w_bbox * l1 + w_class * (-prob)) + w_giou * (-giou);
    cost[((size_t)b * Nmax + j) * Q + q] = cval;
  }
}

struct Cand {
  double v;
  int it;       // position in `remaining`
  int unas;     // column unassigned?
};

__device__ __forceinline__ Cand better(const Cand& a, const Cand& b) {
  // result of scanning both in position order with: take if v < lowest || (v == lowest && unassigned)
  if (a.it < 0) return b;
  if (b.it < 0) return a;
  if (a.v < b.v) return a;
  if (b.v < a.v) return b;
  if (a.unas != b.unas) return a.unas ? a : b;
  if (a.unas) return a.it > b.it ? a : b;  // last unassigned wins
  return a.it < b.it ? a : b;              // first assigned stays
}

__device__ __forceinline__ Cand shfl_cand(const Cand& c, int off) {
  Cand r;
  r.v = __shfl_xor(c.v, off, 64);
  r.it = __shfl_xor(c.it, off, 64);
  r.unas = __shfl_xor(c.unas, off, 64);
  return r;
}

// cost(i,j) of image b at cost[b*bs + i*rs + j*cs], i < nr (same for all images), j < nc[b]
__global__ __launch_bounds__(256) void lsap_k(const float* __restrict__ cost, long long bs, long long rs, long long cs, int nr_in,
                                              const int* __restrict__ nc_per, int nc_max, long long* __restrict__ row_idx,
                                              long long* __restrict__ col_idx, int kmax, int* __restrict__ count,
                                              int* __restrict__ status, int R_cap, int C_cap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* Cb = cost + (size_t)b * bs;
  const int nc_in = nc_per ? nc_per[b] : nc_max;
  // internal orientation: rows <= cols ("wide"); a tall matrix is solved transposed
  const bool transpose = nc_in < nr_in;
  const int nr = transpose ? nc_in : nr_in;
  const int nc = transpose ? nr_in : nc_in;
  const long long irs = transpose ? cs : rs, ics = transpose ? rs : cs;

  double* u = reinterpret_cast<double*>(smem);          // [R_cap]
  double* v = u + R_cap;                                 // [C_cap]
  double* spc = v + C_cap;                               // [C_cap]
  int* path = reinterpret_cast<int*>(spc + C_cap);       // [C_cap]
  int* row4col = path + C_cap;                           // [C_cap]
  int* remaining = row4col + C_cap;                      // [C_cap]
  int* col4row = remaining + C_cap;                      // [R_cap]
  unsigned char* SR = reinterpret_cast<unsigned char*>(col4row + R_cap);  // [R_cap]
  unsigned char* SC = SR + R_cap;                        // [C_cap]
  __shared__ Cand wbest[4];
  __shared__ int s_ctl[4];  // 0: sink, 1: cur row i, 2: n_rem, 3: error
  __shared__ double s_min;

  if (nr == 0 || nc == 0) {
    if (tid == 0) { count[b] = 0; status[b] = 0; }
    return;
  }
  // invalid entries: NaN or -inf (scipy raises ValueError)
  int bad = 0;
  for (long long e = tid; e < (long long)nr * nc; e += 256) {
    const int i = (int)(e / nc), j = (int)(e - (long long)i * nc);
    const float c = Cb[i * irs + j * ics];
    if (c != c || c == -INFINITY) bad = 1;
  }
  bad = __syncthreads_or(bad);
  if (bad) {
    if (tid == 0) { count[b] = 0; status[b] = -2; }
    return;
  }
  for (int i = tid; i < nr; i += 256) { u[i] = 0.0; col4row[i] = -1; }
  for (int j = tid; j < nc; j += 256) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
  if (tid == 0) s_ctl[3] = 0;
  __syncthreads();

  for (int cur = 0; cur < nr; ++cur) {
    for (int j = tid; j < nc; j += 256) { remaining[j] = nc - j - 1; spc[j] = INFINITY; SC[j] = 0; }
    for (int i = tid; i < nr; i += 256) SR[i] = 0;
    if (tid == 0) { s_ctl[0] = -1; s_ctl[1] = cur; s_ctl[2] = nc; s_min = 0.0; }
    __syncthreads();
    while (true) {
      const int i = s_ctl[1], n_rem = s_ctl[2];
      const double min_val = s_min, ui = u[i];
      const float* Ci = Cb + i * irs;
      Cand best;
      best.v = INFINITY; best.it = -1; best.unas = 0;
      for (int it = tid; it < n_rem; it += 256) {
        const int j = remaining[it];
        const double r = ((min_val + (double)Ci[j * ics]) - ui) - v[j];
        double sj = spc[j];
        if (r < sj) { path[j] = i; spc[j] = r; sj = r; }
        Cand c;
        c.v = sj; c.it = it; c.unas = row4col[j] == -1;
        best = better(best, c);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) best = better(best, shfl_cand(best, off));
      if (lane == 0) wbest[wid] = best;
      __syncthreads();
      if (tid == 0) {
        SR[i] = 1;
        Cand bb = better(better(wbest[0], wbest[1]), better(wbest[2], wbest[3]));
        s_min = bb.v;
        if (bb.it < 0 || bb.v == INFINITY) {
          s_ctl[3] = 1;  // infeasible
          s_ctl[0] = 0;
        } else {
          const int j = remaining[bb.it];
          if (row4col[j] == -1) s_ctl[0] = j; else s_ctl[1] = row4col[j];
          SC[j] = 1;
          remaining[bb.it] = remaining[n_rem - 1];
          s_ctl[2] = n_rem - 1;
        }
      }
      __syncthreads();
      if (s_ctl[0] != -1) break;
    }
    if (s_ctl[3]) break;
    const double min_val = s_min;
    const int sink = s_ctl[0];
    // dual update (reads spc / col4row of the pre-augmentation state)
    for (int i = tid; i < nr; i += 256) {
      if (i == cur) u[i] += min_val;
      else if (SR[i]) u[i] += min_val - spc[col4row[i]];
    }
    for (int j = tid; j < nc; j += 256)
      if (SC[j]) v[j] -= min_val - spc[j];
    __syncthreads();
    if (tid == 0) {
      int j = sink;
      while (true) {
        const int i = path[j];
        row4col[j] = i;
        const int t = col4row[i];
        col4row[i] = j;
        j = t;
        if (i == cur) break;
      }
    }
    __syncthreads();
  }

  if (s_ctl[3]) {
    if (tid == 0) { count[b] = 0; status[b] = -1; }
    return;
  }
  long long* ro = row_idx + (size_t)b * kmax;
  long long* co = col_idx + (size_t)b * kmax;
  if (!transpose) {
    for (int i = tid; i < nr && i < kmax; i += 256) { ro[i] = i; co[i] = col4row[i]; }
  } else if (tid == 0) {
    int n = 0;
    for (int j = 0; j < nc && n < kmax; ++j)
      if (row4col[j] != -1) { ro[n] = j; co[n] = row4col[j]; ++n; }
  }
  if (tid == 0) { count[b] = nr < kmax ? nr : kmax; status[b] = 0; }
}

}  // namespace

extern "C" int am_match_cost_d(const float* logits, const float* boxes, int D, const int64_t* tgt_labels, const float* tgt_boxes,
                               const int32_t* n_tgt, int B, int Q, int C, int Nmax, float w_class, float w_bbox, float w_giou,
                               float* cost, am_stream_t stream) {
  if (!logits || !boxes || !n_tgt || !cost || B < 0 || Q <= 0 || C <= 0 || Nmax < 0 || D <= 0) return AM_ERR_ARG;
  if (Nmax > 0 && (!tgt_labels || !tgt_boxes)) return AM_ERR_ARG;
  if (B == 0 || Nmax == 0) return AM_OK;
  const dim3 grid(am_cdiv(Q, 256), B);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define AM_MC_ARGS logits, boxes, (const long long*)tgt_labels, tgt_boxes, (const int*)n_tgt, Q, C, Nmax, D, w_class, w_bbox, w_giou, cost
  if (D == 4) hipLaunchKernelGGL(match_cost_k<4>, grid, dim3(256), 0, s, AM_MC_ARGS);
  else if (D == 7) hipLaunchKernelGGL(match_cost_k<7>, grid, dim3(256), 0, s, AM_MC_ARGS);
  else hipLaunchKernelGGL(match_cost_k<0>, grid, dim3(256), 0, s, AM_MC_ARGS);
#undef AM_MC_ARGS
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_match_cost(const float* logits, const float* boxes, const int64_t* tgt_labels, const float* tgt_boxes,
                             const int32_t* n_tgt, int B, int Q, int C, int Nmax, float w_class, float w_bbox, float w_giou,
                             float* cost, am_stream_t stream) {
  return am_match_cost_d(logits, boxes, 4, tgt_labels, tgt_boxes, n_tgt, B, Q, C, Nmax, w_class, w_bbox, w_giou, cost, stream);
}

extern "C" int am_lsap_batched(const float* cost, int B, int nr, const int32_t* nc_per, int nc_max, long long batch_stride,
                               long long row_stride, long long col_stride, int64_t* row_idx, int64_t* col_idx, int kmax,
                               int32_t* count, int32_t* status, am_stream_t stream) {
  if (!row_idx || !col_idx || !count || !status || B < 0 || nr < 0 || nc_max < 0 || kmax < 0) return AM_ERR_ARG;
  if (B == 0) return AM_OK;
  if (!cost && nr > 0 && nc_max > 0) return AM_ERR_ARG;
  const int R_cap = ((nr < nc_max ? nr : nc_max) + 7) & ~7;  // rows after orienting wide, worst case
  const int C_cap = ((nr > nc_max ? nr : nc_max) + 7) & ~7;
  const size_t lds = (size_t)R_cap * (8 + 4 + 1) + (size_t)C_cap * (8 + 8 + 4 + 4 + 4 + 1) + 64;
  if (lds > 150 * 1024) return AM_ERR_UNSUPPORTED;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lsap_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return AM_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(lsap_k, dim3(B), dim3(256), lds, static_cast<hipStream_t>(stream), cost, batch_stride, row_stride, col_stride,
                     nr, (const int*)nc_per, nc_max, (long long*)row_idx, (long long*)col_idx, kmax, (int*)count, (int*)status,
                     R_cap, C_cap);
  AM_CHECK_LAUNCH();
  return AM_OK;
}
