// BatchNorm2d (train and eval) around the conv gather-GEMM, NHWC, HBM-bound elementwise kernels.
// Forward: conv epilogue -> fp64 per-channel sums (conv_gemm.hip) -> am_bn_finalize -> am_bn_apply
// (normalise + optional residual add + ReLU in one pass, 16 B per lane).
// Backward: am_bn_bwd_reduce (sum dz, sum dz*xhat; ReLU mask from the saved output) ->
// am_bn_bwd_finalize -> am_bn_bwd_apply (dx, and dz for the residual branch).
#include "am_common.h"

namespace {

__global__ void bn_finalize_k(const double* __restrict__ stats, int nrep, double count, const float* __restrict__ conv_bias,
                              const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
                              float* running_var, float momentum, float eps, int training, float* scale, float* shift,
                              float* save_mean, float* save_rstd, int C, int sgn_stats) {
  // sgn_stats (am_bn_finalize_signed): the sums were taken on x' = sgn(gamma) * x (am_conv_first_fused mode 4).  mean(x) =
  // sgn * mean(x'), var(x) = var(x'); the emitted scale / shift apply to x':  gamma * (x - mean) * rstd + beta =
  // |gamma| * (x' - mean') * rstd + beta, so scale = |gamma| * rstd >= 0 and shift = beta - mean' * scale.
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, rstd;
  const float gsign = (sgn_stats && gamma && gamma[c] < 0.f) ? -1.f : 1.f;
  if (training) {
    // all replica loads in flight at once: issued one per iteration (runtime trip count, dependent adds) the 2*nrep L2 round
    // trips were most of this tiny kernel's 8 us
    double sv[AM_STATS_REPLICAS], qv[AM_STATS_REPLICAS];
#pragma unroll
    for (int r = 0; r < AM_STATS_REPLICAS; ++r) {
      sv[r] = r < nrep ? stats[(size_t)r * 2 * C + c] : 0.0;
      qv[r] = r < nrep ? stats[(size_t)r * 2 * C + C + c] : 0.0;
    }
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int r = 0; r < AM_STATS_REPLICAS; ++r) { s += sv[r]; q += qv[r]; }
    const double m0 = s / count;
    double var = q / count - m0 * m0;
    if (var < 0.0) var = 0.0;
    const double m = m0 + (conv_bias ? (double)conv_bias[c] : 0.0);
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (gsign * mean);
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  } else {
    mean = running_mean[c];
    rstd = 1.0f / sqrtf(running_var[c] + eps);
  }
  const float g = (gamma ? gamma[c] : 1.f) * gsign, b = beta ? beta[c] : 0.f;
  scale[c] = g * rstd;
  shift[c] = b - mean * g * rstd;
  if (save_mean) save_mean[c] = gsign * mean;
  if (save_rstd) save_rstd[c] = rstd;
}

template <typename T>
struct Vec16 {
  static constexpr int N = 16 / (int)sizeof(T);
  T v[N];
};

template <typename T, bool RAFF>
__global__ __launch_bounds__(256) void bn_apply_k(const T* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, const T* __restrict__ res, int ldr,
                                                  const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                  int relu, T* __restrict__ y, int ldy, long long P, int C) {
  constexpr int E = 16 / (int)sizeof(T);
  const int cpr = C / E;  // chunks per pixel row
  const long long total = P * cpr;
  const long long stride = (long long)gridDim.x * blockDim.x;
  // The launch has a multiple of 256 threads: when cpr divides 256 (every power-of-two channel count of the trunks) a thread
  // meets the SAME channel chunk in every iteration, so its per-channel constants are loaded once, the pixel advances by a
  // constant and the loop body is the data loads, the arithmetic and the store (per iteration the constants used to be four
  // 16-byte loads next to two or three of data).  Other channel counts re-derive chunk and constants per iteration.
  const bool fixed = (256 % cpr) == 0;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long pix = i / cpr;
  int c0 = (int)(i - pix * cpr) * E;
  const long long pix_step = fixed ? stride / cpr : 0;
  float sc[E], sh[E], rs[E], rh[E];
  auto load_consts = [&]() {
#pragma unroll
    for (int e = 0; e < E; e += 4) {
      *reinterpret_cast<f32x4*>(sc + e) = *reinterpret_cast<const f32x4*>(scale + c0 + e);
      *reinterpret_cast<f32x4*>(sh + e) = *reinterpret_cast<const f32x4*>(shift + c0 + e);
      if (RAFF) {
        *reinterpret_cast<f32x4*>(rs + e) = *reinterpret_cast<const f32x4*>(rscale + c0 + e);
        *reinterpret_cast<f32x4*>(rh + e) = *reinterpret_cast<const f32x4*>(rshift + c0 + e);
      }
    }
  };
  if (i < total) load_consts();
  for (; i < total; i += stride) {
    if (!fixed) {
      pix = i / cpr;
      c0 = (int)(i - pix * cpr) * E;
      load_consts();
    }
    Vec16<T> xv = *reinterpret_cast<const Vec16<T>*>(x + pix * ldx + c0);
    Vec16<T> rv;
    if (res) rv = *reinterpret_cast<const Vec16<T>*>(res + pix * ldr + c0);
    Vec16<T> out;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float v = am_to_f32(xv.v[e]) * sc[e] + sh[e];
      if (res) {
        // a raw residual (the downsample conv's output) is normalised here, rounded to T first like the tensor a separate
        // am_bn_apply pass would have written
        const float r = am_to_f32(rv.v[e]);
        // (relu & 2: the residual is itself a BatchNorm + ReLU output that was never written -- the ResNet stem behind the
        // one-pass pool, am_conv_first_fused mode 4)
        float ra = r * rs[e] + rh[e];
        if (RAFF && (relu & 2)) ra = fmaxf(ra, 0.f);
        v += RAFF ? am_to_f32(am_from_f32<T>(ra)) : r;
      }
      if (relu & 1) v = fmaxf(v, 0.f);
      out.v[e] = am_from_f32<T>(v);
    }
    *reinterpret_cast<Vec16<T>*>(y + pix * ldy + c0) = out;
    pix += pix_step;
  }
}

// One workgroup strides over pixel rows; thread t owns channel chunk (t % cpr) for rows t / cpr + k*rpb.
// Partial sums stay in registers; block-level LDS reduce; fp64 atomics into replicas.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_k(const T* __restrict__ dy, int lddy, const T* __restrict__ yout, int ldyo,
                                                       const T* __restrict__ x, int ldx, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, int relu, double* __restrict__ sums,
                                                       long long P, int C, const float* __restrict__ sg_scale,
                                                       const float* __restrict__ sg_shift) {
  // sg_scale / sg_shift (the *_sign entries): the layer has no residual, so its ReLU mask is the sign of its own normalised
  // output x * scale + shift (am_bn_apply's fp32 arithmetic) -- recomputed from the conv output that is read anyway instead of
  // reading the activation tensor
  constexpr int E = 16 / (int)sizeof(T);
  using S = typename am_stat_acc<T>::type;  // double in fp32 parity mode (torch-CPU's accumulation type), float for f16
  extern __shared__ char red_raw[];
  S* red = reinterpret_cast<S*>(red_raw);  // [256][2*E]
  const int cpr = C / E;
  const int tid = threadIdx.x;
  // threads are laid out [rows_per_pass][cpr]; threads beyond rows_per_pass*cpr idle
  const int rpp = 256 / cpr > 0 ? 256 / cpr : 1;
  const int chunk = tid % cpr, rloc = tid / cpr;
  S s[E], q[E];
#pragma unroll
  for (int e = 0; e < E; ++e) s[e] = q[e] = (S)0;
  if (cpr <= 256 && rloc < rpp) {
    const int c0 = chunk * E;
    float mu[E], rs[E], sc[E], sh[E];
    const bool sign = sg_scale != nullptr;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      mu[e] = mean[c0 + e]; rs[e] = rstd[c0 + e];
      sc[e] = sign ? sg_scale[c0 + e] : 0.f; sh[e] = sign ? sg_shift[c0 + e] : 0.f;
    }
    const bool mask_y = relu && !sign;
    auto accumulate = [&](const Vec16<T>& g, const Vec16<T>& xv, const Vec16<T>& yo) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        float dz = am_to_f32(g.v[e]);
        const float xf = am_to_f32(xv.v[e]);
        if (mask_y && !(am_to_f32(yo.v[e]) > 0.f)) dz = 0.f;
        if (sign && !(xf * sc[e] + sh[e] > 0.f)) dz = 0.f;
        s[e] += (S)dz;
        if constexpr (sizeof(S) == 8) q[e] += (double)dz * ((double)xf - (double)mu[e]) * (double)rs[e];
        else q[e] += dz * (xf - mu[e]) * rs[e];
      }
    };
    // two rows per iteration, all their loads issued first: with one row (2-3 loads of 16 B per thread, four waves per SIMD) the
    // pass ran at the latency bound, 4.3 TB/s on the 118 MB layer1 tensors; the sums are taken in the same order as before
    const long long step = (long long)gridDim.x * rpp;
    long long pix = (long long)blockIdx.x * rpp + rloc;
    for (; pix + step < P; pix += 2 * step) {
      const long long pix2 = pix + step;
      const Vec16<T> g0 = *reinterpret_cast<const Vec16<T>*>(dy + pix * lddy + c0), g1 = *reinterpret_cast<const Vec16<T>*>(dy + pix2 * lddy + c0);
      const Vec16<T> x0 = *reinterpret_cast<const Vec16<T>*>(x + pix * ldx + c0), x1 = *reinterpret_cast<const Vec16<T>*>(x + pix2 * ldx + c0);
      Vec16<T> y0, y1;
      if (mask_y) {
        y0 = *reinterpret_cast<const Vec16<T>*>(yout + pix * ldyo + c0);
        y1 = *reinterpret_cast<const Vec16<T>*>(yout + pix2 * ldyo + c0);
      }
      accumulate(g0, x0, y0);
      accumulate(g1, x1, y1);
    }
    if (pix < P) {
      const Vec16<T> g0 = *reinterpret_cast<const Vec16<T>*>(dy + pix * lddy + c0);
      const Vec16<T> x0 = *reinterpret_cast<const Vec16<T>*>(x + pix * ldx + c0);
      Vec16<T> y0;
      if (mask_y) y0 = *reinterpret_cast<const Vec16<T>*>(yout + pix * ldyo + c0);
      accumulate(g0, x0, y0);
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) { red[tid * 2 * E + e] = s[e]; red[tid * 2 * E + E + e] = q[e]; }
  __syncthreads();
  // thread (chunk, e) pairs: C*2 outputs; reduce over rloc
  for (int o = tid; o < cpr * E * 2; o += 256) {
    const int which = o / (cpr * E), ce = o % (cpr * E);
    const int ch = ce / E, e = ce % E;
    double a = 0.0;
    for (int r = 0; r < rpp; ++r) a += (double)red[(r * cpr + ch) * 2 * E + which * E + e];
    atomicAdd(sums + (size_t)(blockIdx.x % AM_STATS_REPLICAS) * 2 * C + (size_t)which * C + ce, a);
  }
}

__global__ void bn_bwd_finalize_k(const double* __restrict__ sums, int nrep, double count, const float* __restrict__ gamma,
                                  const float* __restrict__ rstd, float gscale, float* __restrict__ dgamma,
                                  float* __restrict__ dbeta, float* __restrict__ coef, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double sv[AM_STATS_REPLICAS], qv[AM_STATS_REPLICAS];
#pragma unroll
  for (int r = 0; r < AM_STATS_REPLICAS; ++r) {
    sv[r] = r < nrep ? sums[(size_t)r * 2 * C + c] : 0.0;
    qv[r] = r < nrep ? sums[(size_t)r * 2 * C + C + c] : 0.0;
  }
  double s = 0.0, q = 0.0;
#pragma unroll
  for (int r = 0; r < AM_STATS_REPLICAS; ++r) { s += sv[r]; q += qv[r]; }
  if (dgamma) dgamma[c] += (float)(q * gscale);
  if (dbeta) dbeta[c] += (float)(s * gscale);
  const float g = gamma ? gamma[c] : 1.f;
  coef[c] = g * rstd[c];                 // c1
  coef[C + c] = (float)(s / count);      // mean(dz)
  coef[2 * C + c] = (float)(q / count);  // mean(dz*xhat)
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_k(const T* __restrict__ dy, int lddy, const T* __restrict__ yout, int ldyo,
                                                      const T* __restrict__ x, int ldx, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ coef, int relu,
                                                      T* __restrict__ dx, int lddx, T* __restrict__ dz_out, int lddz,
                                                      long long P, int C, const float* __restrict__ sg_scale,
                                                      const float* __restrict__ sg_shift) {
  constexpr int E = 16 / (int)sizeof(T);
  const int cpr = C / E;
  const long long total = P * cpr;
  const bool sign = sg_scale != nullptr, mask_y = relu && !sign;  // (see bn_bwd_reduce_k)
  const long long stride = (long long)gridDim.x * blockDim.x;
  const bool fixed = (256 % cpr) == 0;  // the thread's channel chunk is loop-invariant: constants in registers (see bn_apply_k)
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long pix = i / cpr;
  int c0 = (int)(i - pix * cpr) * E;
  const long long pix_step = fixed ? stride / cpr : 0;
  float mu[E], rs[E], k0[E], k1[E], k2[E], ssc[E], ssh[E];
  auto load_consts = [&]() {
#pragma unroll
    for (int e = 0; e < E; e += 4) {
      *reinterpret_cast<f32x4*>(mu + e) = *reinterpret_cast<const f32x4*>(mean + c0 + e);
      *reinterpret_cast<f32x4*>(rs + e) = *reinterpret_cast<const f32x4*>(rstd + c0 + e);
      *reinterpret_cast<f32x4*>(k0 + e) = *reinterpret_cast<const f32x4*>(coef + c0 + e);
      *reinterpret_cast<f32x4*>(k1 + e) = *reinterpret_cast<const f32x4*>(coef + C + c0 + e);
      *reinterpret_cast<f32x4*>(k2 + e) = *reinterpret_cast<const f32x4*>(coef + 2 * C + c0 + e);
      if (sign) {
        *reinterpret_cast<f32x4*>(ssc + e) = *reinterpret_cast<const f32x4*>(sg_scale + c0 + e);
        *reinterpret_cast<f32x4*>(ssh + e) = *reinterpret_cast<const f32x4*>(sg_shift + c0 + e);
      }
    }
  };
  if (i < total) load_consts();
  auto transform = [&](const Vec16<T>& g, const Vec16<T>& xv, const Vec16<T>& yo, long long at) {
    Vec16<T> o, z;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float dz = am_to_f32(g.v[e]);
      const float xf = am_to_f32(xv.v[e]);
      if (mask_y && !(am_to_f32(yo.v[e]) > 0.f)) dz = 0.f;
      if (sign && !(xf * ssc[e] + ssh[e] > 0.f)) dz = 0.f;
      const float xhat = (xf - mu[e]) * rs[e];
      o.v[e] = am_from_f32<T>(k0[e] * (dz - k1[e] - xhat * k2[e]));
      z.v[e] = am_from_f32<T>(dz);
    }
    *reinterpret_cast<Vec16<T>*>(dx + at * lddx + c0) = o;
    if (dz_out) *reinterpret_cast<Vec16<T>*>(dz_out + at * lddz + c0) = z;
  };
  if (fixed) {
    // two rows per iteration with all their loads issued first (the one-row loop below is latency-bound on the large tensors)
    for (; i + stride < total; i += 2 * stride) {
      const long long p2 = pix + pix_step;
      const Vec16<T> g0 = *reinterpret_cast<const Vec16<T>*>(dy + pix * lddy + c0), g1 = *reinterpret_cast<const Vec16<T>*>(dy + p2 * lddy + c0);
      const Vec16<T> x0 = *reinterpret_cast<const Vec16<T>*>(x + pix * ldx + c0), x1 = *reinterpret_cast<const Vec16<T>*>(x + p2 * ldx + c0);
      Vec16<T> y0, y1;
      if (mask_y) {
        y0 = *reinterpret_cast<const Vec16<T>*>(yout + pix * ldyo + c0);
        y1 = *reinterpret_cast<const Vec16<T>*>(yout + p2 * ldyo + c0);
      }
      transform(g0, x0, y0, pix);
      transform(g1, x1, y1, p2);
      pix += 2 * pix_step;
    }
  }
  for (; i < total; i += stride) {
    if (!fixed) {
      pix = i / cpr;
      c0 = (int)(i - pix * cpr) * E;
      load_consts();
    }
    const Vec16<T> g = *reinterpret_cast<const Vec16<T>*>(dy + pix * lddy + c0);
    const Vec16<T> xv = *reinterpret_cast<const Vec16<T>*>(x + pix * ldx + c0);
    Vec16<T> yo;
    if (mask_y) yo = *reinterpret_cast<const Vec16<T>*>(yout + pix * ldyo + c0);
    transform(g, xv, yo, pix);
    pix += pix_step;
  }
}

// dz = dy * (y > 0) (optional), written to dz_out (optional), column sums -> dbias (fp32, += scale*sum)
template <typename T>
__global__ __launch_bounds__(256) void bias_relu_bwd_k(const T* __restrict__ dy, int lddy, const T* __restrict__ yout, int ldyo,
                                                       int relu, T* __restrict__ dz_out, int lddz, float* __restrict__ dbias,
                                                       float gscale, long long P, int C, int Cvalid) {
  constexpr int E = 16 / (int)sizeof(T);
  extern __shared__ float red[];  // [256][E]
  const int cpr = C / E;
  const int tid = threadIdx.x;
  const int rpp = 256 / cpr > 0 ? 256 / cpr : 1;
  const int chunk = tid % cpr, rloc = tid / cpr;
  float s[E];
#pragma unroll
  for (int e = 0; e < E; ++e) s[e] = 0.f;
  if (cpr <= 256 && rloc < rpp) {
    const int c0 = chunk * E;
    for (long long pix = (long long)blockIdx.x * rpp + rloc; pix < P; pix += (long long)gridDim.x * rpp) {
      Vec16<T> g = *reinterpret_cast<const Vec16<T>*>(dy + pix * lddy + c0);
      Vec16<T> yo;
      if (relu) yo = *reinterpret_cast<const Vec16<T>*>(yout + pix * ldyo + c0);
      Vec16<T> z;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        float dz = am_to_f32(g.v[e]);
        if (relu && !(am_to_f32(yo.v[e]) > 0.f)) dz = 0.f;
        s[e] += dz;
        z.v[e] = am_from_f32<T>(dz);
      }
      if (dz_out) *reinterpret_cast<Vec16<T>*>(dz_out + pix * lddz + c0) = z;
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) red[tid * E + e] = s[e];
  __syncthreads();
  if (dbias) {
    for (int ce = tid; ce < cpr * E; ce += 256) {
      const int ch = ce / E, e = ce % E;
      float a = 0.f;
      for (int r = 0; r < rpp; ++r) a += red[(r * cpr + ch) * E + e];
      if (ce < Cvalid) atomicAdd(dbias + ce, a * gscale);
    }
  }
}

inline int ew_grid(long long total_threads) {
  long long b = (total_threads + 255) / 256;
  if (b > 2048) b = 2048;  // 256 CUs x 8 blocks, grid-stride the rest
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int am_bn_finalize(const double* stats, int nrep, double count, const float* conv_bias, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                              int training, float* scale, float* shift, float* save_mean, float* save_rstd, int C,
                              am_stream_t stream) {
  if (C <= 0 || !scale || !shift) return AM_ERR_ARG;
  if (training && (!stats || count <= 0.0 || nrep > AM_STATS_REPLICAS)) return AM_ERR_ARG;
  if (!training && (!running_mean || !running_var)) return AM_ERR_ARG;
  hipLaunchKernelGGL(bn_finalize_k, dim3(am_cdiv(C, 128)), dim3(128), 0, static_cast<hipStream_t>(stream), stats, nrep, count,
                     conv_bias, gamma, beta, running_mean, running_var, momentum, eps, training, scale, shift, save_mean,
                     save_rstd, C, 0);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_bn_finalize_signed(const double* stats, int nrep, double count, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, float momentum, float eps, float* scale,
                                     float* shift, int C, am_stream_t stream) {
  if (C <= 0 || !scale || !shift || !gamma || !stats || count <= 0.0 || nrep > AM_STATS_REPLICAS) return AM_ERR_ARG;
  hipLaunchKernelGGL(bn_finalize_k, dim3(am_cdiv(C, 128)), dim3(128), 0, static_cast<hipStream_t>(stream), stats, nrep, count,
                     (const float*)nullptr, gamma, beta, running_mean, running_var, momentum, eps, 1, scale, shift,
                     (float*)nullptr, (float*)nullptr, C, 1);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

#define AM_EW_CHECK(C, ld, es) (((C) * (es)) % 16 != 0 || ((ld) * (es)) % 16 != 0)

extern "C" int am_bn_apply2(int dtype, const void* x, int ldx, const float* scale, const float* shift, const void* res, int ldr,
                            const float* res_scale, const float* res_shift, int relu, void* y, int ldy, long long P, int C,
                            am_stream_t stream);

extern "C" int am_bn_apply(int dtype, const void* x, int ldx, const float* scale, const float* shift, const void* res, int ldr,
                           int relu, void* y, int ldy, long long P, int C, am_stream_t stream) {
  return am_bn_apply2(dtype, x, ldx, scale, shift, res, ldr, nullptr, nullptr, relu, y, ldy, P, C, stream);
}

extern "C" int am_bn_apply2(int dtype, const void* x, int ldx, const float* scale, const float* shift, const void* res, int ldr,
                            const float* res_scale, const float* res_shift, int relu, void* y, int ldy, long long P, int C,
                            am_stream_t stream) {
  if ((res_scale == nullptr) != (res_shift == nullptr) || (res_scale && !res)) return AM_ERR_ARG;
  const int es = dtype == AM_F16 ? 2 : 4;
  if ((dtype != AM_F16 && dtype != AM_F32) || !x || !y || !scale || !shift || P < 0 || C <= 0) return AM_ERR_ARG;
  if (AM_EW_CHECK(C, ldx, es) || (ldy * es) % 16 != 0 || (res && (ldr * es) % 16 != 0)) return AM_ERR_ARG;
  if (P == 0) return AM_OK;
  const int grid = ew_grid(P * (C * es / 16));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == AM_F16)
    {
    if (res_scale) hipLaunchKernelGGL((bn_apply_k<half_t, true>), dim3(grid), dim3(256), 0, s, (const half_t*)x, ldx, scale, shift, (const half_t*)res, ldr, res_scale, res_shift, relu, (half_t*)y, ldy, P, C);
    else hipLaunchKernelGGL((bn_apply_k<half_t, false>), dim3(grid), dim3(256), 0, s, (const half_t*)x, ldx, scale, shift, (const half_t*)res, ldr, res_scale, res_shift, relu, (half_t*)y, ldy, P, C);
  }
  else
    {
    if (res_scale) hipLaunchKernelGGL((bn_apply_k<float, true>), dim3(grid), dim3(256), 0, s, (const float*)x, ldx, scale, shift, (const float*)res, ldr, res_scale, res_shift, relu, (float*)y, ldy, P, C);
    else hipLaunchKernelGGL((bn_apply_k<float, false>), dim3(grid), dim3(256), 0, s, (const float*)x, ldx, scale, shift, (const float*)res, ldr, res_scale, res_shift, relu, (float*)y, ldy, P, C);
  }
  AM_CHECK_LAUNCH();
  return AM_OK;
}

static int bn_bwd_reduce_impl(int dtype, const void* dy, int lddy, const void* yout, int ldyo, const void* x, int ldx,
                              const float* mean, const float* rstd, int relu, double* sums, long long P, int C,
                              const float* sg_scale, const float* sg_shift, am_stream_t stream) {
  const int es = dtype == AM_F16 ? 2 : 4;
  if ((dtype != AM_F16 && dtype != AM_F32) || !dy || !x || !mean || !rstd || !sums || (relu && !yout) || C <= 0) return AM_ERR_ARG;
  if (AM_EW_CHECK(C, ldx, es) || (lddy * es) % 16 != 0 || (relu && (ldyo * es) % 16 != 0)) return AM_ERR_ARG;
  const int E = 16 / es, cpr = C / E;
  if (cpr > 256) return AM_ERR_UNSUPPORTED;
  if (P == 0) return AM_OK;
  const int rpp = 256 / cpr;
  int grid = (int)((P + rpp * 8 - 1) / (rpp * 8));
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  const size_t lds = 256 * 2 * E * (dtype == AM_F16 ? sizeof(float) : sizeof(double));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == AM_F16)
    hipLaunchKernelGGL(bn_bwd_reduce_k<half_t>, dim3(grid), dim3(256), lds, s, (const half_t*)dy, lddy, (const half_t*)yout, ldyo, (const half_t*)x, ldx, mean, rstd, relu, sums, P, C, sg_scale, sg_shift);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_k<float>, dim3(grid), dim3(256), lds, s, (const float*)dy, lddy, (const float*)yout, ldyo, (const float*)x, ldx, mean, rstd, relu, sums, P, C, sg_scale, sg_shift);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_bn_bwd_reduce(int dtype, const void* dy, int lddy, const void* yout, int ldyo, const void* x, int ldx,
                                const float* mean, const float* rstd, int relu, double* sums, long long P, int C,
                                am_stream_t stream) {
  return bn_bwd_reduce_impl(dtype, dy, lddy, yout, ldyo, x, ldx, mean, rstd, relu, sums, P, C, nullptr, nullptr, stream);
}

extern "C" int am_bn_bwd_reduce_sign(int dtype, const void* dy, int lddy, const void* x, int ldx, const float* mean, const float* rstd,
                                     const float* scale, const float* shift, double* sums, long long P, int C, am_stream_t stream) {
  if (!scale || !shift) return AM_ERR_ARG;
  return bn_bwd_reduce_impl(dtype, dy, lddy, nullptr, 0, x, ldx, mean, rstd, 0, sums, P, C, scale, shift, stream);
}

extern "C" int am_bn_bwd_finalize(const double* sums, int nrep, double count, const float* gamma, const float* rstd,
                                  float gscale, float* dgamma, float* dbeta, float* coef, int C, am_stream_t stream) {
  if (!sums || !rstd || !coef || C <= 0 || count <= 0.0 || nrep > AM_STATS_REPLICAS) return AM_ERR_ARG;
  hipLaunchKernelGGL(bn_bwd_finalize_k, dim3(am_cdiv(C, 128)), dim3(128), 0, static_cast<hipStream_t>(stream), sums, nrep, count, gamma, rstd, gscale, dgamma, dbeta, coef, C);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

static int bn_bwd_apply_impl(int dtype, const void* dy, int lddy, const void* yout, int ldyo, const void* x, int ldx,
                             const float* mean, const float* rstd, const float* coef, int relu, void* dx, int lddx,
                             void* dz_out, int lddz, long long P, int C, const float* sg_scale, const float* sg_shift,
                             am_stream_t stream) {
  const int es = dtype == AM_F16 ? 2 : 4;
  if ((dtype != AM_F16 && dtype != AM_F32) || !dy || !x || !mean || !rstd || !coef || !dx || (relu && !yout) || C <= 0) return AM_ERR_ARG;
  if (AM_EW_CHECK(C, ldx, es) || (lddy * es) % 16 != 0 || (lddx * es) % 16 != 0 || (relu && (ldyo * es) % 16 != 0) || (dz_out && (lddz * es) % 16 != 0)) return AM_ERR_ARG;
  if (P == 0) return AM_OK;
  const int grid = ew_grid(P * (C * es / 16));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == AM_F16)
    hipLaunchKernelGGL(bn_bwd_apply_k<half_t>, dim3(grid), dim3(256), 0, s, (const half_t*)dy, lddy, (const half_t*)yout, ldyo, (const half_t*)x, ldx, mean, rstd, coef, relu, (half_t*)dx, lddx, (half_t*)dz_out, lddz, P, C, sg_scale, sg_shift);
  else
    hipLaunchKernelGGL(bn_bwd_apply_k<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, lddy, (const float*)yout, ldyo, (const float*)x, ldx, mean, rstd, coef, relu, (float*)dx, lddx, (float*)dz_out, lddz, P, C, sg_scale, sg_shift);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_bn_bwd_apply(int dtype, const void* dy, int lddy, const void* yout, int ldyo, const void* x, int ldx,
                               const float* mean, const float* rstd, const float* coef, int relu, void* dx, int lddx,
                               void* dz_out, int lddz, long long P, int C, am_stream_t stream) {
  return bn_bwd_apply_impl(dtype, dy, lddy, yout, ldyo, x, ldx, mean, rstd, coef, relu, dx, lddx, dz_out, lddz, P, C, nullptr, nullptr, stream);
}

extern "C" int am_bn_bwd_apply_sign(int dtype, const void* dy, int lddy, const void* x, int ldx, const float* mean, const float* rstd,
                                    const float* coef, const float* scale, const float* shift, void* dx, int lddx, long long P, int C,
                                    am_stream_t stream) {
  if (!scale || !shift) return AM_ERR_ARG;
  return bn_bwd_apply_impl(dtype, dy, lddy, nullptr, 0, x, ldx, mean, rstd, coef, 0, dx, lddx, nullptr, 0, P, C, scale, shift, stream);
}

extern "C" int am_bias_relu_bwd(int dtype, const void* dy, int lddy, const void* yout, int ldyo, int relu, void* dz_out,
                                int lddz, float* dbias, float gscale, long long P, int C, int Cvalid, am_stream_t stream) {
  const int es = dtype == AM_F16 ? 2 : 4;
  if ((dtype != AM_F16 && dtype != AM_F32) || !dy || (relu && !yout) || C <= 0 || Cvalid > C) return AM_ERR_ARG;
  if ((C * es) % 16 != 0 || (lddy * es) % 16 != 0 || (relu && (ldyo * es) % 16 != 0) || (dz_out && (lddz * es) % 16 != 0)) return AM_ERR_ARG;
  const int E = 16 / es, cpr = C / E;
  if (cpr > 256) return AM_ERR_UNSUPPORTED;
  if (P == 0) return AM_OK;
  const int rpp = 256 / cpr;
  int grid = (int)((P + rpp * 8 - 1) / (rpp * 8));
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  const size_t lds = 256 * E * sizeof(float);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == AM_F16)
    hipLaunchKernelGGL(bias_relu_bwd_k<half_t>, dim3(grid), dim3(256), lds, s, (const half_t*)dy, lddy, (const half_t*)yout, ldyo, relu, (half_t*)dz_out, lddz, dbias, gscale, P, C, Cvalid);
  else
    hipLaunchKernelGGL(bias_relu_bwd_k<float>, dim3(grid), dim3(256), lds, s, (const float*)dy, lddy, (const float*)yout, ldyo, relu, (float*)dz_out, lddz, dbias, gscale, P, C, Cvalid);
  AM_CHECK_LAUNCH();
  return AM_OK;
}
