// Step glue on flat fp32 buffers: global grad-norm, clip + AdamW in one pass.
// Replaces clip_grad_norm_(max_norm) + AdamW.step() (training/train_bdd100k_ddp.py:98-99,
// training/train_gating_network.py:103-105) without any host synchronisation: the norm stays on
// the device and the update kernel reads it from there.
#include "am_common.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_k(const float* __restrict__ x, long long n, double* __restrict__ acc) {
  __shared__ double red[4];
  double s = 0.0;
  const long long n4 = n / 4;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  // four 16-byte loads in flight per thread and at most 512 workgroups (one fp64 atomic each onto the same address): one load per
  // iteration and 2048 atomics ran 50 MB of gradients at 1.4 TB/s
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
    s += ((double)(v0.x * v0.x + v0.y * v0.y) + (double)(v0.z * v0.z + v0.w * v0.w)) +
         ((double)(v1.x * v1.x + v1.y * v1.y) + (double)(v1.z * v1.z + v1.w * v1.w));
    s += ((double)(v2.x * v2.x + v2.y * v2.y) + (double)(v2.z * v2.z + v2.w * v2.w)) +
         ((double)(v3.x * v3.x + v3.y * v3.y) + (double)(v3.z * v3.z + v3.w * v3.w));
  }
  for (; i < n4; i += stride) {
    const float4 v = x4[i];
    s += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0)
    for (long long i = n4 * 4 + threadIdx.x; i < n; i += 256) s += (double)x[i] * x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, (red[0] + red[1]) + (red[2] + red[3]));
}

// p, g, m, v: flat fp32 [n].  norm_sq: device scalar (sum of squares of ALL grads of the step, may
// span several flat buffers).  clip_coef = min(1, max_norm / (sqrt(norm_sq) + 1e-6)) as torch's
// clip_grad_norm_.  A non-finite norm skips the update (fp16 loss-scale overflow) and counts it.
// Scalars arrive as torch applies them: computed in double on the host, rounded to fp32 once (decay = 1 - lr*wd, omb1 = 1 - beta1,
// omb2 = 1 - beta2, step_size = lr / (1 - beta1^t), bias_c2_sqrt = sqrt(1 - beta2^t)); the moment updates are torch's own
// expressions (exp_avg.lerp_(g, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(g, g, value = 1 - beta2)), so optimizer state
// moves between this kernel and torch.optim.AdamW to the last bits.
__global__ __launch_bounds__(256) void adamw_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, long long n, float decay, float omb1, float beta2, float omb2,
                                               float eps, float step_size, float bias_c2_sqrt, float max_norm,
                                               const double* __restrict__ norm_sq, int* __restrict__ skipped) {
  float clip = 1.f;
  if (norm_sq) {
    const double ns = norm_sq[0];
    if (!(ns == ns) || ns > 1.7e308 * 0.5) {  // NaN or inf
      if (skipped && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(skipped, 1);
      return;
    }
    if (max_norm > 0.f) {
      const float coef = max_norm / ((float)sqrt(ns) + 1e-6f);
      clip = coef < 1.f ? coef : 1.f;
    }
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * clip;
    float pi = p[i] * decay;
    const float mi = m[i] + omb1 * (gi - m[i]);
    const float vi = beta2 * v[i] + omb2 * gi * gi;
    const float denom = sqrtf(vi) / bias_c2_sqrt + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

__global__ __launch_bounds__(256) void scale_k(float* __restrict__ x, long long n, float mul, const double* __restrict__ denom) {
  const float f = denom ? mul / (float)denom[0] : mul;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] *= f;
}

inline int ew_grid(long long total) {
  long long b = (total + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}


// dst[i] = idx[i] < 0 ? 0 : (T)src[idx[i]]: every packed conv operand (forward and dgrad layouts) of a layer, rebuilt
// from the fp32 master weight in one launch after an optimizer step.
// Eight outputs per thread: two 16-byte index loads, eight independent gathers in flight (issued unconditionally from a clamped
// index: a branch per element would serialise them), one 16-byte store (f16).  One element per thread ran at 84 G elements/s
// (131 us for a ResNet-18 expert's forward + input-gradient operands): the 2-byte stores and one gather in flight per lane.
template <typename T>
__global__ void gather_cast_k(const float* __restrict__ src, const int* __restrict__ idx, T* __restrict__ dst, long long n) {
  const long long n8 = n >> 3, stride = (long long)gridDim.x * blockDim.x, t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long long i = t0; i < n8; i += stride) {
    const int4 j0 = reinterpret_cast<const int4*>(idx)[2 * i], j1 = reinterpret_cast<const int4*>(idx)[2 * i + 1];
    const int j[8] = {j0.x, j0.y, j0.z, j0.w, j1.x, j1.y, j1.z, j1.w};
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[max(j[e], 0)];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = j[e] < 0 ? 0.f : v[e];
    if constexpr (sizeof(T) == 2) {
      const half8_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3], (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
      reinterpret_cast<half8_t*>(dst)[i] = o;
    } else {
      reinterpret_cast<float4*>(dst)[2 * i] = make_float4(v[0], v[1], v[2], v[3]);
      reinterpret_cast<float4*>(dst)[2 * i + 1] = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
  for (long long i = n8 * 8 + t0; i < n; i += stride) {
    const int j = idx[i];
    dst[i] = j < 0 ? (T)0.f : (T)src[j];
  }
}

}  // namespace

#define ST(s) static_cast<hipStream_t>(s)

extern "C" int am_sumsq_accumulate(const float* x, long long n, double* acc, am_stream_t stream) {
  if (!x || !acc || n < 0) return AM_ERR_ARG;
  if (n == 0) return AM_OK;
  const int grid = ew_grid(n / 4 + 1);
  hipLaunchKernelGGL(sumsq_k, dim3(grid < 512 ? grid : 512), dim3(256), 0, ST(stream), x, n, acc);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_adamw_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2,
                             double eps, double weight_decay, int step, float max_norm, const double* norm_sq, int* skipped,
                             am_stream_t stream) {
  if (!p || !g || !m || !v || n < 0 || step < 1) return AM_ERR_ARG;
  if (n == 0) return AM_OK;
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adamw_k, dim3(ew_grid(n)), dim3(256), 0, ST(stream), p, g, m, v, n, (float)(1.0 - lr * weight_decay),
                     (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)(lr / bc1), (float)sqrt(bc2), max_norm,
                     norm_sq, skipped);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_scale_inplace(float* x, long long n, float mul, const double* denom, am_stream_t stream) {
  if (!x || n < 0) return AM_ERR_ARG;
  if (n == 0) return AM_OK;
  hipLaunchKernelGGL(scale_k, dim3(ew_grid(n)), dim3(256), 0, ST(stream), x, n, mul, denom);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_gather_cast(int dtype, const float* src, const int* idx, void* dst, long long n, am_stream_t stream) {
  if ((dtype != AM_F16 && dtype != AM_F32) || !src || !idx || !dst || n < 0) return AM_ERR_ARG;
  if (n == 0) return AM_OK;
  if ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(dst)) & 15) return AM_ERR_ARG;  // 16-byte vector accesses
  if (dtype == AM_F16) hipLaunchKernelGGL(gather_cast_k<half_t>, dim3(ew_grid(n / 8 + 1)), dim3(256), 0, ST(stream), src, idx, (half_t*)dst, n);
  else hipLaunchKernelGGL(gather_cast_k<float>, dim3(ew_grid(n / 8 + 1)), dim3(256), 0, ST(stream), src, idx, (float*)dst, n);
  AM_CHECK_LAUNCH();
  return AM_OK;
}
