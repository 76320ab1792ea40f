// The "MoE tail": extractor / context / gating / policy-head MLP arithmetic.  All fp32: the batch is
// the row count (M <= 64 per GPU), so these layers are weight-bandwidth / latency bound, not MFMA
// work.  Wave-per-column GEMV batches, wave-per-row LayerNorm and gate softmax-combine, with wave64
// shuffle reductions.
#include "am_common.h"

namespace {

constexpr int MB = 8;  // rows per pass

// y[m][n] = act(sum_k x[m][k] * W[n][k] + b[n]).  One wave per (output column n, chunk of MB rows): grid
// (ceil(N/4), ceil(M/MB)).  Lanes stride over K with 16-byte loads when the rows allow it, so a wave has
// (1 + MB) independent float4 loads in flight per iteration; wave64 shuffle reduction per row.
template <bool VEC>
__global__ __launch_bounds__(256) void linear_fwd_k(const float* __restrict__ x, int ldx, const float* __restrict__ W,
                                                    const float* __restrict__ bias, float* __restrict__ y, int ldy, int M, int N,
                                                    int K, int relu) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int m0 = blockIdx.y * MB;
  const float* w = W + (size_t)n * K;
  const float* xr[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) xr[i] = x + (size_t)min(m0 + i, M - 1) * ldx;  // rows past M alias the last row (discarded)
  float acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) acc[i] = 0.f;
  if constexpr (VEC) {
#pragma unroll 2
    for (int k = lane * 4; k < K; k += 256) {
      const float4 wv = *reinterpret_cast<const float4*>(w + k);
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const float4 xv = *reinterpret_cast<const float4*>(xr[i] + k);
        acc[i] += (wv.x * xv.x + wv.y * xv.y) + (wv.z * xv.z + wv.w * xv.w);
      }
    }
  } else {
    for (int k = lane; k < K; k += 64) {
      const float wv = w[k];
#pragma unroll
      for (int i = 0; i < MB; ++i) acc[i] += wv * xr[i][k];
    }
  }
  const float b = bias ? bias[n] : 0.f;
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const float s = wave_sum(acc[i]);
    if (lane == 0 && m0 + i < M) {
      float v = s + b;
      if (relu) v = fmaxf(v, 0.f);
      y[(size_t)(m0 + i) * ldy + n] = v;
    }
  }
}

// dx[m][k] (+)= sum_n dz[m][n] * W[n][k], dz = dy * (yact > 0) when yact given.
// block = 64 k-columns x 16 n-slices (1024 threads): dz of the block's MB rows is staged once in LDS (ReLU mask applied),
// each thread streams its slice of W rows (coalesced across k) against LDS-broadcast dz; LDS reduce over slices.
constexpr int BI_SLICES = 16;
__global__ __launch_bounds__(1024) void linear_bwd_input_k(const float* __restrict__ dy, int lddy, const float* __restrict__ yact,
                                                           int ldya, const float* __restrict__ W, float* __restrict__ dx, int lddx,
                                                           int M, int N, int K, int accumulate) {
  extern __shared__ float sm[];            // dz [MB][N] then red [BI_SLICES][MB][64]
  float* dz = sm;
  float* red = sm + (size_t)MB * N;
  const int kx = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + kx;
  const int m0 = blockIdx.y * MB;
  for (int e = threadIdx.x; e < MB * N; e += 1024) {
    const int i = e / N, n = e - i * N;
    float g = 0.f;
    if (m0 + i < M) {
      g = dy[(size_t)(m0 + i) * lddy + n];
      if (yact && !(yact[(size_t)(m0 + i) * ldya + n] > 0.f)) g = 0.f;
    }
    dz[e] = g;
  }
  __syncthreads();
  float acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) acc[i] = 0.f;
  const int nper = (N + BI_SLICES - 1) / BI_SLICES;
  const int nb = slice * nper, ne = min(N, nb + nper);
  if (k < K) {
#pragma unroll 4
    for (int n = nb; n < ne; ++n) {
      const float wv = W[(size_t)n * K + k];
#pragma unroll
      for (int i = 0; i < MB; ++i) acc[i] += dz[i * N + n] * wv;
    }
  }
#pragma unroll
  for (int i = 0; i < MB; ++i) red[(slice * MB + i) * 64 + kx] = acc[i];
  __syncthreads();
  for (int e = threadIdx.x; e < MB * 64; e += 1024) {
    const int i = e >> 6, kk = e & 63;
    const int ko = blockIdx.x * 64 + kk;
    if (m0 + i < M && ko < K) {
      float v = 0.f;
#pragma unroll
      for (int sl = 0; sl < BI_SLICES; ++sl) v += red[(sl * MB + i) * 64 + kk];
      float* d = dx + (size_t)(m0 + i) * lddx + ko;
      *d = accumulate ? *d + v : v;
    }
  }
}

// dW[n][k] += sum_m dz[m][n] * x[m][k]; dbias[n] += sum_m dz[m][n].  thread per (n,k)
__global__ __launch_bounds__(256) void linear_bwd_weight_k(const float* __restrict__ dy, int lddy, const float* __restrict__ yact,
                                                           int ldya, const float* __restrict__ x, int ldx, float* __restrict__ dW,
                                                           float* __restrict__ dbias, int M, int N, int K) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (k >= K) return;
  float acc = 0.f, bacc = 0.f;
  for (int m = 0; m < M; ++m) {
    float g = dy[(size_t)m * lddy + n];
    if (yact && !(yact[(size_t)m * ldya + n] > 0.f)) g = 0.f;
    acc += g * x[(size_t)m * ldx + k];
    bacc += g;
  }
  dW[(size_t)n * K + k] += acc;
  if (dbias && k == 0) dbias[n] += bacc;
}

// ---- LayerNorm over the last dim (D <= 4096), one wave per row -----------------------------
__global__ __launch_bounds__(256) void layernorm_fwd_k(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, float* __restrict__ y, int ldy,
                                                       float* __restrict__ mean, float* __restrict__ rstd, int M, int D) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* xr = x + (size_t)m * ldx;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += xr[d];
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int d = lane; d < D; d += 64) { const float t = xr[d] - mu; q += t * t; }
  const float rs = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  for (int d = lane; d < D; d += 64) y[(size_t)m * ldy + d] = (xr[d] - mu) * rs * gamma[d] + beta[d];
  if (lane == 0) { mean[m] = mu; rstd[m] = rs; }
}

__global__ __launch_bounds__(256) void layernorm_bwd_x_k(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* __restrict__ dx, int lddx, int M,
                                                         int D) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float mu = mean[m], rs = rstd[m];
  float a = 0.f, b = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float g = dy[(size_t)m * lddy + d] * gamma[d];
    a += g;
    b += g * (x[(size_t)m * ldx + d] - mu) * rs;
  }
  a = wave_sum(a) / (float)D;
  b = wave_sum(b) / (float)D;
  for (int d = lane; d < D; d += 64) {
    const float g = dy[(size_t)m * lddy + d] * gamma[d];
    const float xh = (x[(size_t)m * ldx + d] - mu) * rs;
    dx[(size_t)m * lddx + d] = rs * (g - a - xh * b);
  }
}

__global__ __launch_bounds__(256) void layernorm_bwd_p_k(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int D) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  float a = 0.f, b = 0.f;
  for (int m = 0; m < M; ++m) {
    const float g = dy[(size_t)m * lddy + d];
    a += g * (x[(size_t)m * ldx + d] - mean[m]) * rstd[m];
    b += g;
  }
  dgamma[d] += a;
  dbeta[d] += b;
}

// ---- gate: weights from logits (softmax/temperature, or sigmoid-normalise; optional top-k mask)
//      and combined = sum_e w[:,e] * P_e        (models/gating/gating_network.py:149-166) ----------
struct GatePtrs {
  const float* p[AM_MAX_EXPERTS];
  float* dp[AM_MAX_EXPERTS];
};

__device__ __forceinline__ void gate_weights_row(const float* lg, int E, float temp, int use_softmax, int topk, float* w,
                                                 float* sig) {
  float l[AM_MAX_EXPERTS];
  bool keep[AM_MAX_EXPERTS];
  for (int e = 0; e < E; ++e) { l[e] = lg[e]; keep[e] = true; }
  if (topk > 0 && topk < E) {
    // torch.topk: k largest, ties -> lower index first
    for (int e = 0; e < E; ++e) {
      int rank = 0;
      for (int j = 0; j < E; ++j)
        if (l[j] > l[e] || (l[j] == l[e] && j < e)) ++rank;
      keep[e] = rank < topk;
    }
  }
  if (use_softmax) {
    float mx = -INFINITY;
    for (int e = 0; e < E; ++e) if (keep[e]) mx = fmaxf(mx, l[e] / temp);
    float se = 0.f;
    for (int e = 0; e < E; ++e) { w[e] = keep[e] ? expf(l[e] / temp - mx) : 0.f; se += w[e]; }
    for (int e = 0; e < E; ++e) w[e] /= se;
  } else {
    float ss = 0.f;
    for (int e = 0; e < E; ++e) { sig[e] = keep[e] ? 1.f / (1.f + expf(-l[e])) : 0.f; ss += sig[e]; }
    for (int e = 0; e < E; ++e) w[e] = sig[e] / (ss + 1e-8f);
  }
}

__global__ __launch_bounds__(256) void gate_combine_fwd_k(const float* __restrict__ logits, GatePtrs ptrs, int ldp, float temp,
                                                          int use_softmax, int topk, float* __restrict__ weights,
                                                          float* __restrict__ combined, int B, int E, int D) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float w[AM_MAX_EXPERTS], sig[AM_MAX_EXPERTS];
  gate_weights_row(logits + (size_t)b * E, E, temp, use_softmax, topk, w, sig);
  if (lane < E) weights[(size_t)b * E + lane] = w[lane];
  for (int d = lane; d < D; d += 64) {
    float acc = 0.f;  // accumulated in expert order into an fp32 zero, as the reference does (:163-166)
    for (int e = 0; e < E; ++e) acc += w[e] * ptrs.p[e][(size_t)b * ldp + d];
    combined[(size_t)b * D + d] = acc;
  }
}

__global__ __launch_bounds__(256) void gate_combine_bwd_k(const float* __restrict__ logits, GatePtrs ptrs, int ldp, float temp,
                                                          int use_softmax, int topk, const float* __restrict__ dcombined,
                                                          const float* __restrict__ dweights_ext, float* __restrict__ dlogits,
                                                          int B, int E, int D) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float w[AM_MAX_EXPERTS], sig[AM_MAX_EXPERTS], dw[AM_MAX_EXPERTS];
  gate_weights_row(logits + (size_t)b * E, E, temp, use_softmax, topk, w, sig);
  for (int e = 0; e < E; ++e) dw[e] = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float g = dcombined[(size_t)b * D + d];
    for (int e = 0; e < E; ++e) {
      dw[e] += g * ptrs.p[e][(size_t)b * ldp + d];
      ptrs.dp[e][(size_t)b * ldp + d] = w[e] * g;
    }
  }
  for (int e = 0; e < E; ++e) {
    dw[e] = wave_sum(dw[e]);
    if (dweights_ext) dw[e] += dweights_ext[(size_t)b * E + e];
  }
  if (lane == 0) {
    if (use_softmax) {
      float dot = 0.f;
      for (int e = 0; e < E; ++e) dot += w[e] * dw[e];
      for (int e = 0; e < E; ++e) dlogits[(size_t)b * E + e] = w[e] * (dw[e] - dot) / temp;
    } else {
      float ss = 1e-8f, dot = 0.f;
      for (int e = 0; e < E; ++e) ss += sig[e];
      for (int e = 0; e < E; ++e) dot += dw[e] * sig[e];
      // w_j = s_j / S  ->  dL/ds_e = dw_e / S - dot / S^2 ; ds/dl = s (1 - s); masked (top-k) entries have s = 0
      for (int e = 0; e < E; ++e) dlogits[(size_t)b * E + e] = (dw[e] / ss - dot / (ss * ss)) * sig[e] * (1.f - sig[e]);
    }
  }
}

// ---- dropout: counter-based hash RNG (statistically equivalent to torch's, not the same stream) ----
__device__ __forceinline__ unsigned hash_u32(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z = z ^ (z >> 31);
  return (unsigned)(z >> 32);
}

__global__ __launch_bounds__(256) void dropout_fwd_k(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ mask,
                                                     long long n, float p, unsigned long long seed,
                                                     const long long* __restrict__ dev_step) {
  if (dev_step) seed += (unsigned long long)dev_step[0] * 0xD1B54A32D192ED03ULL;  // per-step stream under graph replay
  const float inv = 1.f / (1.f - p);
  const unsigned thr = (unsigned)((double)p * 4294967296.0);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const bool keep = hash_u32(seed, (unsigned long long)i) >= thr;
    mask[i] = keep;
    y[i] = keep ? x[i] * inv : 0.f;
  }
}

__global__ __launch_bounds__(256) void dropout_bwd_k(const float* __restrict__ dy, const uint8_t* __restrict__ mask,
                                                     float* __restrict__ dx, long long n, float p) {
  const float inv = 1.f / (1.f - p);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dx[i] = mask[i] ? dy[i] * inv : 0.f;
}

inline int ew_grid(long long total) {
  long long b = (total + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}


// Gating-stage objective and its gradient in one single-workgroup launch (B*T*2 ~ 500 elements; ~70 tiny torch launches
// otherwise).  L1 means use sign(0) = 0 like torch's abs backward.
//   ade = mean|wp - twp|, fde = mean|wp[:,T-1] - twp[:,T-1]|, speed = mean|spd - tspd|, smooth = mean|d[t+1] - d[t]| with
//   d[t] = wp[t+1] - wp[t], lb = mean_e (mean_b w[b,e] - 1/E)^2, ent = mean_b sum_e w log(w + 1e-8);
//   total = cw[0] ade + cw[1] fde + cw[2] speed + cw[3] smooth + cw[4] lb + cw[5] ent.
struct GatingLossArgs {
  const float* wp; const float* twp; int B, T;
  const float* spd; const float* tspd; int ld_spd, ld_tspd, S;  // S == 0: no speed term
  const float* w; int E;
  float cw[6];
  int use_lb, use_ent;
  float* total;   // scalar
  float* parts;   // [6]: ade, fde, speed, smooth, lb, ent
  float* g_wp; float* g_spd; float* g_w;  // d total / d input (dense [B,T,2], [B,S], [B,E])
};

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__device__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
  return t;
}

__global__ __launch_bounds__(256) void gating_losses_k(const GatingLossArgs a) {
  __shared__ float red[4];
  __shared__ float usage[64];
  const int tid = threadIdx.x;
  const int B = a.B, T = a.T, E = a.E, S = a.S;
  const int nwp = B * T * 2;
  const float c_ade = a.cw[0] / (float)nwp, c_fde = a.cw[1] / (float)(B * 2);
  const int nsm = T > 2 ? B * (T - 2) * 2 : 0;
  const float c_sm = nsm > 0 ? a.cw[3] / (float)nsm : 0.f;
  float s_ade = 0.f, s_fde = 0.f, s_sm = 0.f;
  for (int i = tid; i < nwp; i += blockDim.x) {
    const int c = i & 1, t = (i >> 1) % T, b = i / (2 * T);
    const float* row = a.wp + (size_t)b * T * 2;
    const float v = row[t * 2 + c], d = v - a.twp[i];
    s_ade += fabsf(d);
    float g = c_ade * sgn(d);
    if (t == T - 1) { s_fde += fabsf(d); g += c_fde * sgn(d); }
    // second differences q(u) = wp[u+2] - 2 wp[u+1] + wp[u], u = 0..T-3; element t appears in q(t), q(t-1), q(t-2)
    if (t + 2 < T) {
      const float q = row[(t + 2) * 2 + c] - 2.f * row[(t + 1) * 2 + c] + v;
      s_sm += fabsf(q);
      g += c_sm * sgn(q);
    }
    if (t >= 1 && t + 1 < T) {
      const float q = row[(t + 1) * 2 + c] - 2.f * v + row[(t - 1) * 2 + c];
      g -= 2.f * c_sm * sgn(q);
    }
    if (t >= 2) {
      const float q = v - 2.f * row[(t - 1) * 2 + c] + row[(t - 2) * 2 + c];
      g += c_sm * sgn(q);
    }
    a.g_wp[i] = g;
  }
  float s_spd = 0.f;
  if (S > 0) {
    const float c_spd = a.cw[2] / (float)(B * S);
    for (int i = tid; i < B * S; i += blockDim.x) {
      const int b = i / S, k = i - b * S;
      const float d = a.spd[(size_t)b * a.ld_spd + k] - a.tspd[(size_t)b * a.ld_tspd + k];
      s_spd += fabsf(d);
      a.g_spd[i] = c_spd * sgn(d);
    }
  }
  // expert usage (E <= 64)
  if (tid < E) {
    float u = 0.f;
    for (int b = 0; b < B; ++b) u += a.w[(size_t)b * E + tid];
    usage[tid] = u / (float)B;
  }
  __syncthreads();
  float s_ent = 0.f, s_lb = 0.f;
  if (tid < E && a.use_lb) { const float d = usage[tid] - 1.f / (float)E; s_lb = d * d; }
  for (int i = tid; i < B * E; i += blockDim.x) {
    const int e = i % E;
    const float wv = a.w[i];
    float g = 0.f;
    if (a.use_ent) {
      const float lg = logf(wv + 1e-8f);
      s_ent += wv * lg;
      g += a.cw[5] / (float)B * (lg + wv / (wv + 1e-8f));
    }
    if (a.use_lb) g += a.cw[4] * 2.f / (float)E * (usage[e] - 1.f / (float)E) / (float)B;
    a.g_w[i] = g;
  }
  const float ade = block_sum(s_ade, red) / (float)nwp;
  const float fde = block_sum(s_fde, red) / (float)(B * 2);
  const float sm = nsm > 0 ? block_sum(s_sm, red) / (float)nsm : __builtin_nanf("");  // torch: mean of an empty tensor (T <= 2)
  const float spd = S > 0 ? block_sum(s_spd, red) / (float)(B * S) : 0.f;
  const float lb = a.use_lb ? block_sum(s_lb, red) / (float)E : 0.f;
  const float ent = a.use_ent ? block_sum(s_ent, red) / (float)B : 0.f;
  if (tid == 0) {
    a.total[0] = a.cw[0] * ade + a.cw[1] * fde + a.cw[2] * spd + a.cw[3] * sm + a.cw[4] * lb + a.cw[5] * ent;
    a.parts[0] = ade; a.parts[1] = fde; a.parts[2] = spd; a.parts[3] = sm; a.parts[4] = lb; a.parts[5] = ent;
  }
}


// ---- grouped forms: several INDEPENDENT layers of the MoE tail in one launch (blockIdx.z picks the layer) --------------------
// The tail is ~13 dependent stages, but every stage has 2-5 parallel branches (one extractor / processor MLP per expert, the
// context encoder, the two policy heads); as separate launches each branch costs a 5-17 us kernel of its own.
struct LinGroup { am_tail_linear p[AM_TAIL_MAX_GROUP]; };
struct LnGroup { am_tail_layernorm p[AM_TAIL_MAX_GROUP]; };

// y = dropout(relu(x W^T + b)): linear_fwd_k's wave-per-(column, 8-row chunk) scheme; the dropout of the reference's
// Linear -> ReLU -> Dropout triples is applied in the epilogue (same counter-based hash as dropout_fwd_k, element index m*N + n),
// so the mask is recoverable from the stored output (zero <=> dropped or rectified): backward needs no mask tensor.
__global__ __launch_bounds__(256) void linear_group_fwd_k(const LinGroup grp, int M, const long long* __restrict__ dev_step) {
  const am_tail_linear& d = grp.p[blockIdx.z];
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int N = d.N, K = d.K;
  if (n >= N) return;
  const int m0 = blockIdx.y * MB;
  if (m0 >= M) return;
  const float* w = d.W + (size_t)n * K;
  const float* xr[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) xr[i] = d.x + (size_t)min(m0 + i, M - 1) * d.ldx;
  float acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) acc[i] = 0.f;
  const bool vec = (K % 4 == 0) && (d.ldx % 4 == 0) && (((uintptr_t)d.x | (uintptr_t)d.W) % 16 == 0);
  if (vec) {
#pragma unroll 2
    for (int k = lane * 4; k < K; k += 256) {
      const float4 wv = *reinterpret_cast<const float4*>(w + k);
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const float4 xv = *reinterpret_cast<const float4*>(xr[i] + k);
        acc[i] += (wv.x * xv.x + wv.y * xv.y) + (wv.z * xv.z + wv.w * xv.w);
      }
    }
  } else {
    for (int k = lane; k < K; k += 64) {
      const float wv = w[k];
#pragma unroll
      for (int i = 0; i < MB; ++i) acc[i] += wv * xr[i][k];
    }
  }
  const float b = d.bias ? d.bias[n] : 0.f;
  unsigned long long seed = d.seed;
  if (dev_step) seed += (unsigned long long)dev_step[0] * 0xD1B54A32D192ED03ULL;
  const float inv = d.drop_p > 0.f ? 1.f / (1.f - d.drop_p) : 1.f;
  const unsigned thr = (unsigned)((double)d.drop_p * 4294967296.0);
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const float sum = wave_sum(acc[i]);
    if (lane == 0 && m0 + i < M) {
      float v = sum + b;
      if (d.relu) v = fmaxf(v, 0.f);
      if (d.drop_p > 0.f) v = hash_u32(seed, (unsigned long long)(m0 + i) * N + n) >= thr ? v * inv : 0.f;
      d.y[(size_t)(m0 + i) * d.ldy + n] = v;
    }
  }
}

// dx (+)= dz W, dz = dy * gscale * (yact > 0): linear_bwd_input_k per group member (dynamic LDS sized for the widest N)
__global__ __launch_bounds__(1024) void linear_group_bwd_input_k(const LinGroup grp, int M) {
  const am_tail_linear& d = grp.p[blockIdx.z];
  if (d.dx == nullptr) return;
  extern __shared__ float sm[];
  const int N = d.N, K = d.K;
  if ((int)blockIdx.x * 64 >= K) return;
  float* dz = sm;
  float* red = sm + (size_t)MB * N;
  const int kx = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + kx;
  const int m0 = blockIdx.y * MB;
  if (m0 >= M) return;
  // both loads of an element unconditional (clamped row) and four elements per thread in flight: behind the row / mask branches
  // each was a memory round trip of its own
#pragma unroll 4
  for (int e = threadIdx.x; e < MB * N; e += 1024) {
    const int i = e / N, n = e - i * N;
    const int m = min(m0 + i, M - 1);
    const float gy = d.dy[(size_t)m * d.lddy + n];
    const float ya = d.yact ? d.yact[(size_t)m * d.ldya + n] : 1.f;
    dz[e] = (m0 + i < M && ya > 0.f) ? gy * d.gscale : 0.f;
  }
  __syncthreads();
  float acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) acc[i] = 0.f;
  const int nper = (N + BI_SLICES - 1) / BI_SLICES;
  const int nb = slice * nper, ne = min(N, nb + nper);
  if (k < K) {
    // eight weight loads in flight per batch (the four of `#pragma unroll 4` left the loop one memory round trip per 4 channels)
    for (int n = nb; n < ne; n += 8) {
      float wv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) wv[j] = d.W[(size_t)min(n + j, ne - 1) * K + k];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (n + j < ne) {
#pragma unroll
          for (int i = 0; i < MB; ++i) acc[i] += dz[i * N + n + j] * wv[j];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MB; ++i) red[(slice * MB + i) * 64 + kx] = acc[i];
  __syncthreads();
  for (int e = threadIdx.x; e < MB * 64; e += 1024) {
    const int i = e >> 6, kk = e & 63;
    const int ko = blockIdx.x * 64 + kk;
    if (m0 + i < M && ko < K) {
      float v = 0.f;
#pragma unroll
      for (int sl = 0; sl < BI_SLICES; ++sl) v += red[(sl * MB + i) * 64 + kk];
      float* o = d.dx + (size_t)(m0 + i) * d.lddx + ko;
      *o = d.dx_accumulate ? *o + v : v;
    }
  }
}

// dW[n][k] += sum_m dz[m][n] x[m][k], dbias[n] += sum_m dz[m][n].  A thread owns one k-column and a block of GW_NB output
// channels: its x column (up to GW_MR rows per pass) sits in registers and is reused for every channel of the block.  The
// [GW_MR][GW_NB] tile of dz (gscale and ReLU / dropout mask applied) is staged in LDS by one load per thread, all in flight at
// once; read per element inside the channel / row loops (wave-uniform loads behind `break`s) every one of the 256 products waited
// for its own memory round trip: 62 us per launch at M = 32 for 25 MFLOP.  Same products in the same order as before.
// grid (ceil(Kmax/256), ceil(Nmax/GW_NB), group).
constexpr int GW_NB = 8, GW_MR = 32;
static_assert(GW_NB * GW_MR == 256, "one dz element per thread");
__global__ __launch_bounds__(256) void linear_group_bwd_weight_k(const LinGroup grp, int M) {
  const am_tail_linear& d = grp.p[blockIdx.z];
  if (d.dW == nullptr) return;
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int nb = blockIdx.y * GW_NB;
  if (nb >= d.N || (int)blockIdx.x * 256 >= d.K) return;
  __shared__ __attribute__((aligned(16))) float dzs[GW_MR][GW_NB];
  const bool kok = k < d.K;
  const int si = threadIdx.x >> 3, sj = threadIdx.x & 7;
  float acc[GW_NB], bacc[GW_NB];
#pragma unroll
  for (int j = 0; j < GW_NB; ++j) { acc[j] = 0.f; bacc[j] = 0.f; }
  for (int m0 = 0; m0 < M; m0 += GW_MR) {
    const bool sok = m0 + si < M && nb + sj < d.N;
    const int sm = min(m0 + si, M - 1), sn = min(nb + sj, d.N - 1);
    const float gy = d.dy[(size_t)sm * d.lddy + sn];
    const float ya = d.yact ? d.yact[(size_t)sm * d.ldya + sn] : 1.f;
    float xv[GW_MR];
#pragma unroll
    for (int i = 0; i < GW_MR; ++i) xv[i] = (kok && m0 + i < M) ? d.x[(size_t)(m0 + i) * d.ldx + k] : 0.f;
    float g = gy * d.gscale;
    if (!(ya > 0.f) || !sok) g = 0.f;
    __syncthreads();  // the previous pass has read its tile
    dzs[si][sj] = g;
    __syncthreads();
    const int rows = min(GW_MR, M - m0);
#pragma unroll
    for (int i = 0; i < GW_MR; ++i) {
      if (i < rows) {  // (a row past M would add +0 products: skipped so that sums are those of the per-row loop)
        const float4 g0 = *reinterpret_cast<const float4*>(&dzs[i][0]), g1 = *reinterpret_cast<const float4*>(&dzs[i][4]);
        const float gv[GW_NB] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
        for (int j = 0; j < GW_NB; ++j) {
          acc[j] += gv[j] * xv[i];
          bacc[j] += gv[j];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < GW_NB; ++j) {
    const int n = nb + j;
    if (n >= d.N) break;
    if (kok) d.dW[(size_t)n * d.K + k] += acc[j];
    if (d.dbias && k == 0) d.dbias[n] += bacc[j];
  }
}

__global__ __launch_bounds__(256) void layernorm_group_fwd_k(const LnGroup grp, int M) {
  const am_tail_layernorm& d = grp.p[blockIdx.z];
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int D = d.D;
  const float* xr = d.x + (size_t)m * d.ldx;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += xr[i];
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int i = lane; i < D; i += 64) { const float t = xr[i] - mu; q += t * t; }
  const float rs = 1.0f / sqrtf(wave_sum(q) / (float)D + d.eps);
  for (int i = lane; i < D; i += 64) d.y[(size_t)m * d.ldy + i] = (xr[i] - mu) * rs * d.gamma[i] + d.beta[i];
  if (lane == 0) { d.mean[m] = mu; d.rstd[m] = rs; }
}

// blockIdx.y == 0: input gradient (wave per row); blockIdx.y == 1: parameter gradients (thread per column), one launch for both
__global__ __launch_bounds__(256) void layernorm_group_bwd_k(const LnGroup grp, int M) {
  const am_tail_layernorm& d = grp.p[blockIdx.z];
  const int D = d.D;
  if (blockIdx.y == 0) {
    if (d.dx == nullptr) return;
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float mu = d.mean[m], rs = d.rstd[m];
    float a = 0.f, b = 0.f;
    for (int i = lane; i < D; i += 64) {
      const float g = d.dy[(size_t)m * d.lddy + i] * d.gamma[i];
      a += g;
      b += g * (d.x[(size_t)m * d.ldx + i] - mu) * rs;
    }
    a = wave_sum(a) / (float)D;
    b = wave_sum(b) / (float)D;
    for (int i = lane; i < D; i += 64) {
      const float g = d.dy[(size_t)m * d.lddy + i] * d.gamma[i];
      const float xh = (d.x[(size_t)m * d.ldx + i] - mu) * rs;
      d.dx[(size_t)m * d.lddx + i] = rs * (g - a - xh * b);
    }
  } else {
    if (d.dgamma == nullptr || d.dbeta == nullptr) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= D) return;
    float a = 0.f, b = 0.f;
    for (int m = 0; m < M; ++m) {
      const float g = d.dy[(size_t)m * d.lddy + i];
      a += g * (d.x[(size_t)m * d.ldx + i] - d.mean[m]) * d.rstd[m];
      b += g;
    }
    d.dgamma[i] += a;
    d.dbeta[i] += b;
  }
}

}  // namespace

#define ST(s) static_cast<hipStream_t>(s)

extern "C" int am_linear_fwd(const float* x, int ldx, const float* W, const float* bias, float* y, int ldy, int M, int N,
                             int K, int relu, am_stream_t stream) {
  if (!x || !W || !y || M < 0 || N <= 0 || K <= 0) return AM_ERR_ARG;
  if (M == 0) return AM_OK;
  const bool vec = (K % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W)) % 16 == 0);
  if (vec) hipLaunchKernelGGL(linear_fwd_k<true>, dim3(am_cdiv(N, 4), am_cdiv(M, MB)), dim3(256), 0, ST(stream), x, ldx, W, bias, y, ldy, M, N, K, relu);
  else hipLaunchKernelGGL(linear_fwd_k<false>, dim3(am_cdiv(N, 4), am_cdiv(M, MB)), dim3(256), 0, ST(stream), x, ldx, W, bias, y, ldy, M, N, K, relu);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_linear_bwd_input(const float* dy, int lddy, const float* yact, int ldya, const float* W, float* dx, int lddx,
                                   int M, int N, int K, int accumulate, am_stream_t stream) {
  if (!dy || !W || !dx || M < 0 || N <= 0 || K <= 0) return AM_ERR_ARG;
  if (M == 0) return AM_OK;
  const size_t lds = sizeof(float) * ((size_t)MB * N + (size_t)BI_SLICES * MB * 64);
  if (lds > 150 * 1024) return AM_ERR_UNSUPPORTED;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bwd_input_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return AM_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(linear_bwd_input_k, dim3(am_cdiv(K, 64), am_cdiv(M, MB)), dim3(1024), lds, ST(stream), dy, lddy, yact, ldya, W, dx, lddx, M, N, K, accumulate);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_linear_bwd_weight(const float* dy, int lddy, const float* yact, int ldya, const float* x, int ldx, float* dW,
                                    float* dbias, int M, int N, int K, am_stream_t stream) {
  if (!dy || !x || !dW || M < 0 || N <= 0 || K <= 0 || N > 65535) return AM_ERR_ARG;
  hipLaunchKernelGGL(linear_bwd_weight_k, dim3(am_cdiv(K, 256), N), dim3(256), 0, ST(stream), dy, lddy, yact, ldya, x, ldx, dW, dbias, M, N, K);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float eps, float* y, int ldy,
                                float* mean, float* rstd, int M, int D, am_stream_t stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || M < 0 || D <= 0) return AM_ERR_ARG;
  if (M == 0) return AM_OK;
  hipLaunchKernelGGL(layernorm_fwd_k, dim3(am_cdiv(M, 4)), dim3(256), 0, ST(stream), x, ldx, gamma, beta, eps, y, ldy, mean, rstd, M, D);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_layernorm_bwd(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* mean,
                                const float* rstd, float* dx, int lddx, float* dgamma, float* dbeta, int M, int D,
                                am_stream_t stream) {
  if (!dy || !x || !gamma || !mean || !rstd || M < 0 || D <= 0) return AM_ERR_ARG;
  if (M == 0) return AM_OK;
  if (dx) hipLaunchKernelGGL(layernorm_bwd_x_k, dim3(am_cdiv(M, 4)), dim3(256), 0, ST(stream), dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx, M, D);
  if (dgamma && dbeta) hipLaunchKernelGGL(layernorm_bwd_p_k, dim3(am_cdiv(D, 256)), dim3(256), 0, ST(stream), dy, lddy, x, ldx, mean, rstd, dgamma, dbeta, M, D);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_gate_combine_fwd(const float* logits, const float* const* processed, int E, int ldp, float temperature,
                                   int use_softmax, int top_k, float* weights, float* combined, int B, int D,
                                   am_stream_t stream) {
  if (!logits || !processed || !weights || !combined || E <= 0 || E > AM_MAX_EXPERTS || D <= 0 || temperature == 0.f) return AM_ERR_ARG;
  if (B == 0) return AM_OK;
  GatePtrs ptrs;
  for (int e = 0; e < AM_MAX_EXPERTS; ++e) { ptrs.p[e] = e < E ? processed[e] : nullptr; ptrs.dp[e] = nullptr; }
  hipLaunchKernelGGL(gate_combine_fwd_k, dim3(am_cdiv(B, 4)), dim3(256), 0, ST(stream), logits, ptrs, ldp, temperature, use_softmax, top_k, weights, combined, B, E, D);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_gate_combine_bwd(const float* logits, const float* const* processed, int E, int ldp, float temperature,
                                   int use_softmax, int top_k, const float* dcombined, const float* dweights_ext,
                                   float* dlogits, float* const* dprocessed, int B, int D, am_stream_t stream) {
  if (!logits || !processed || !dcombined || !dlogits || !dprocessed || E <= 0 || E > AM_MAX_EXPERTS || D <= 0 || temperature == 0.f) return AM_ERR_ARG;
  if (B == 0) return AM_OK;
  GatePtrs ptrs;
  for (int e = 0; e < AM_MAX_EXPERTS; ++e) { ptrs.p[e] = e < E ? processed[e] : nullptr; ptrs.dp[e] = e < E ? dprocessed[e] : nullptr; }
  hipLaunchKernelGGL(gate_combine_bwd_k, dim3(am_cdiv(B, 4)), dim3(256), 0, ST(stream), logits, ptrs, ldp, temperature, use_softmax, top_k, dcombined, dweights_ext, dlogits, B, E, D);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_dropout_fwd(const float* x, float* y, uint8_t* mask, long long n, float p, unsigned long long seed,
                              const long long* dev_step, am_stream_t stream) {
  if (!x || !y || !mask || n < 0 || p < 0.f || p >= 1.f) return AM_ERR_ARG;
  if (n == 0) return AM_OK;
  hipLaunchKernelGGL(dropout_fwd_k, dim3(ew_grid(n)), dim3(256), 0, ST(stream), x, y, mask, n, p, seed, dev_step);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, long long n, float p, am_stream_t stream) {
  if (!dy || !dx || !mask || n < 0 || p < 0.f || p >= 1.f) return AM_ERR_ARG;
  if (n == 0) return AM_OK;
  hipLaunchKernelGGL(dropout_bwd_k, dim3(ew_grid(n)), dim3(256), 0, ST(stream), dy, mask, dx, n, p);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_gating_losses(const float* wp, const float* twp, int B, int T, const float* spd, const float* tspd, int ld_spd,
                                int ld_tspd, int S, const float* w, int E, const float* coef6, int use_lb, int use_ent,
                                float* total, float* parts6, float* g_wp, float* g_spd, float* g_w, am_stream_t stream) {
  if (!wp || !twp || !w || !coef6 || !total || !parts6 || !g_wp || !g_w || B <= 0 || T <= 0 || E <= 0 || E > 64 || S < 0) return AM_ERR_ARG;
  if (S > 0 && (!spd || !tspd || !g_spd)) return AM_ERR_ARG;
  GatingLossArgs a;
  a.wp = wp; a.twp = twp; a.B = B; a.T = T;
  a.spd = spd; a.tspd = tspd; a.ld_spd = ld_spd; a.ld_tspd = ld_tspd; a.S = S;
  a.w = w; a.E = E;
  for (int i = 0; i < 6; ++i) a.cw[i] = coef6[i];
  a.use_lb = use_lb; a.use_ent = use_ent;
  a.total = total; a.parts = parts6; a.g_wp = g_wp; a.g_spd = g_spd; a.g_w = g_w;
  hipLaunchKernelGGL(gating_losses_k, dim3(1), dim3(256), 0, ST(stream), a);
  AM_CHECK_LAUNCH();
  return AM_OK;
}


// ---- grouped MoE-tail entries (include/automoe_hip.h am_moe_tail_*) ----------------------------------------------------------
extern "C" int am_moe_tail_linear_fwd(const am_tail_linear* group, int count, int M, const long long* dev_step, am_stream_t stream) {
  if (!group || count < 1 || count > AM_TAIL_MAX_GROUP || M < 0) return AM_ERR_ARG;
  if (M == 0) return AM_OK;
  LinGroup g;
  int nmax = 0;
  for (int i = 0; i < count; ++i) {
    const am_tail_linear& d = group[i];
    if (!d.x || !d.W || !d.y || d.N <= 0 || d.K <= 0 || d.drop_p < 0.f || d.drop_p >= 1.f) return AM_ERR_ARG;
    g.p[i] = d;
    nmax = d.N > nmax ? d.N : nmax;
  }
  hipLaunchKernelGGL(linear_group_fwd_k, dim3(am_cdiv(nmax, 4), am_cdiv(M, MB), count), dim3(256), 0, ST(stream), g, M, dev_step);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_moe_tail_linear_bwd(const am_tail_linear* group, int count, int M, am_stream_t stream) {
  if (!group || count < 1 || count > AM_TAIL_MAX_GROUP || M < 0) return AM_ERR_ARG;
  if (M == 0) return AM_OK;
  LinGroup g;
  int nmax = 0, kmax_in = 0, kmax_w = 0, nmax_w = 0;
  for (int i = 0; i < count; ++i) {
    const am_tail_linear& d = group[i];
    if (!d.dy || !d.W || !d.x || d.N <= 0 || d.K <= 0 || d.N > 65535) return AM_ERR_ARG;
    g.p[i] = d;
    if (d.dx) { nmax = d.N > nmax ? d.N : nmax; kmax_in = d.K > kmax_in ? d.K : kmax_in; }
    if (d.dW) { kmax_w = d.K > kmax_w ? d.K : kmax_w; nmax_w = d.N > nmax_w ? d.N : nmax_w; }
  }
  if (kmax_in > 0) {
    const size_t lds = sizeof(float) * ((size_t)MB * nmax + (size_t)BI_SLICES * MB * 64);
    if (lds > 150 * 1024) return AM_ERR_UNSUPPORTED;
    static bool attr_done_dev[AM_MAX_DEVICES] = {};
    bool& attr_done = attr_done_dev[am_current_device()];
    if (lds > 64 * 1024 && !attr_done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(linear_group_bwd_input_k), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return AM_ERR_LAUNCH;
      attr_done = true;
    }
    hipLaunchKernelGGL(linear_group_bwd_input_k, dim3(am_cdiv(kmax_in, 64), am_cdiv(M, MB), count), dim3(1024), lds, ST(stream), g, M);
    AM_CHECK_LAUNCH();
  }
  if (kmax_w > 0) {
    hipLaunchKernelGGL(linear_group_bwd_weight_k, dim3(am_cdiv(kmax_w, 256), am_cdiv(nmax_w, GW_NB), count), dim3(256), 0, ST(stream), g, M);
    AM_CHECK_LAUNCH();
  }
  return AM_OK;
}

extern "C" int am_moe_tail_layernorm_fwd(const am_tail_layernorm* group, int count, int M, am_stream_t stream) {
  if (!group || count < 1 || count > AM_TAIL_MAX_GROUP || M < 0) return AM_ERR_ARG;
  if (M == 0) return AM_OK;
  LnGroup g;
  for (int i = 0; i < count; ++i) {
    const am_tail_layernorm& d = group[i];
    if (!d.x || !d.gamma || !d.beta || !d.y || !d.mean || !d.rstd || d.D <= 0) return AM_ERR_ARG;
    g.p[i] = d;
  }
  hipLaunchKernelGGL(layernorm_group_fwd_k, dim3(am_cdiv(M, 4), 1, count), dim3(256), 0, ST(stream), g, M);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_moe_tail_layernorm_bwd(const am_tail_layernorm* group, int count, int M, am_stream_t stream) {
  if (!group || count < 1 || count > AM_TAIL_MAX_GROUP || M < 0) return AM_ERR_ARG;
  if (M == 0) return AM_OK;
  LnGroup g;
  int dmax = 0;
  for (int i = 0; i < count; ++i) {
    const am_tail_layernorm& d = group[i];
    if (!d.dy || !d.x || !d.gamma || !d.mean || !d.rstd || d.D <= 0) return AM_ERR_ARG;
    g.p[i] = d;
    dmax = d.D > dmax ? d.D : dmax;
  }
  const int gx = am_cdiv(M, 4) > am_cdiv(dmax, 256) ? am_cdiv(M, 4) : am_cdiv(dmax, 256);
  hipLaunchKernelGGL(layernorm_group_bwd_k, dim3(gx, 2, count), dim3(256), 0, ST(stream), g, M);
  AM_CHECK_LAUNCH();
  return AM_OK;
}
