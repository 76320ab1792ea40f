// What the MI355X sustains on v_mfma_f32_32x32x16_f16 alone: 256 workgroups x 8 waves (two per SIMD, as conv_ring_k<256,256>),
// 8 independent accumulators per wave (the ring kernel's half K-step), operands in registers (random f16 or zeros), no LDS, no
// memory traffic inside the loop.  Prints TFLOP/s, shader cycles per MFMA per SIMD and the clock held (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void mfma_k(const half8* __restrict__ in, float* __restrict__ out, long long* __restrict__ clk, int iters) {
  const int tid = threadIdx.x;
  half8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = in[(blockIdx.x * 512 + tid) * 6 + i];
  for (int i = 0; i < 2; ++i) b[i] = in[(blockIdx.x * 512 + tid) * 6 + 4 + i];
  f32x16 acc[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  __syncthreads();
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

// the same FLOPs per iteration from v_mfma_f32_16x16x32_f16: 16 MFMAs (4 x 4 fragments, 16 accumulators of 4 registers)
__global__ __launch_bounds__(512) void mfma16_k(const half8* __restrict__ in, float* __restrict__ out, long long* __restrict__ clk, int iters) {
  const int tid = threadIdx.x;
  half8 a[4], b[4];
  for (int i = 0; i < 4; ++i) a[i] = in[(blockIdx.x * 512 + tid) * 6 + i];
  for (int i = 0; i < 4; ++i) b[i] = in[(blockIdx.x * 512 + tid) * 6 + (i + 2) % 6];
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  __syncthreads();
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

int main(int argc, char** argv) {
  const int nwg = argc > 2 ? atoi(argv[2]) : 256, iters = argc > 1 ? atoi(argv[1]) : 20000;
  const size_t n = (size_t)nwg * 512 * 6;
  for (int shape = 0; shape < 2; ++shape)
  for (int zeros = 0; zeros < 2; ++zeros) {
    std::vector<half8> h(n);
    srand(1);
    for (auto& v : h) for (int k = 0; k < 8; ++k) v[k] = zeros ? (_Float16)0.f : (_Float16)((rand() % 2001 - 1000) / 1000.f);
    half8* din; float* dout; long long* dclk;
    hipMalloc(&din, n * sizeof(half8)); hipMalloc(&dout, nwg * 512 * sizeof(float)); hipMalloc(&dclk, nwg * 2 * sizeof(long long));
    hipMemcpy(din, h.data(), n * sizeof(half8), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (shape == 0) hipLaunchKernelGGL(mfma_k, dim3(nwg), dim3(512), 0, 0, din, dout, dclk, iters);
      else hipLaunchKernelGGL(mfma16_k, dim3(nwg), dim3(512), 0, 0, din, dout, dclk, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> c(nwg * 2);
      hipMemcpy(c.data(), dclk, c.size() * sizeof(long long), hipMemcpyDeviceToHost);
      double cyc = 0, wall = 0;
      for (int i = 0; i < nwg; ++i) { cyc += c[2 * i]; wall += c[2 * i + 1]; }
      cyc /= nwg; wall /= nwg;
      const double mfma_per_simd = (double)iters * 8 * 2;  // two waves per SIMD
      const double flops = (double)nwg * 8 * iters * 8 * 2.0 * 32 * 32 * 16;
      printf("%s %s rep %d: %.1f TFLOP/s (event), %.2f cycles per MFMA per SIMD, clock %.3f GHz, kernel %.2f ms\n", shape ? "16x16x32" : "32x32x16", zeros ? "zeros " : "random", rep,
             flops / (ms * 1e-3) / 1e12, cyc / mfma_per_simd, cyc / (wall * 10.0), ms);
    }
    hipFree(din); hipFree(dout); hipFree(dclk);
  }
  return 0;
}
