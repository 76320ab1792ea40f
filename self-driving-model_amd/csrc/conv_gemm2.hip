// Second-generation forward / dgrad gather-GEMM for f16: tiles are filled by LDS-DMA
// (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per wave-instruction, no VGPR round trip, no ds_write),
// K-step 128 bytes (64 halves: 16 MFMAs per wave between barriers), per-row address state hoisted out of the
// K-loop (one 64-bit add + one select per 16-byte chunk per K-step).
//
// LDS image: unpadded rows of BKB bytes.  A wave-instruction writes 1 KiB linearly (LDS-DMA destination =
// uniform base + lane*16), i.e. 1024/BKB consecutive rows; bank conflicts of the ds_read_b128 fragment reads
// are removed by an XOR swizzle of the 16-byte chunk index applied on the SOURCE address (which chunk of the
// row a lane fetches) and again on the read: chunk' = chunk ^ swz(row), swz(row) = (row >> log2(256/BKB)) & (BKB/16-1).
// Zero padding: lanes whose pixel lies outside the image fetch from a 64-byte zero line in device memory.
#include "am_common.h"
#include <cstdlib>

namespace am2 {

__device__ __attribute__((aligned(64))) unsigned char g_zero_line[64];  // never written: conv zero padding source

struct Conv2Params {
  am_conv_geom g;
  const void* x;
  const void* w;
  void* y;
  const float* bias;
  double* stats;
  int M, nk, ksteps_per_tap, Ktot, relu, mtiles, ntiles;
  long long tap_off[AM_MAX_TAPS];  // byte offset of tap t relative to the row's base pixel
};

template <int BM, int BN, int WM, int WN, int BKB>
__global__ __launch_bounds__(256) void conv_gemm2_k(const Conv2Params p) {
  typedef half_t T;
  constexpr int CPR = BKB / 16;            // 16-byte chunks per row
  constexpr int RPI = 1024 / BKB;          // rows per wave-instruction
  constexpr int SWZ_SHIFT = (BKB == 128) ? 1 : 2;
  constexpr int AI = BM / RPI / 4;         // A instructions per wave per K-step
  constexpr int BTOT = BN / RPI;           // B instructions per K-step in the whole workgroup
  constexpr int BI = (BTOT + 3) / 4;       // per wave (waves past BTOT idle on B)
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int STAGE = (BM + BN) * BKB;
  constexpr int KSUB = BKB / 32;           // MFMA k16 sub-steps per K-step
  static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && AI >= 1 && BTOT >= 1, "tile");

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  int* opix_s = reinterpret_cast<int*>(smem + 4096);  // filled after the K-loop, inside the then-free stage buffers

  const am_conv_geom& g = p.g;
  const char* __restrict__ x = static_cast<const char*>(p.x);
  const char* __restrict__ w = static_cast<const char*>(p.w);
  T* __restrict__ y = static_cast<T*>(p.y);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int lb = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
  const int mt = lb / p.ntiles, nt = lb - mt * p.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread loader state: instruction j of this wave covers tile rows (wid*AI + j)*RPI + lane/CPR ----
  const int lrow = lane / CPR, cpos = lane % CPR;
  const char* a_base[AI];
  unsigned a_mask[AI];
  const int hw = g.MH * g.MW;
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    const int r = (wid * AI + j) * RPI + lrow;
    const int c = cpos ^ ((r >> SWZ_SHIFT) & (CPR - 1));  // source chunk for this LDS position
    const int m = m0 + r;
    unsigned mask = 0;
    const char* base = reinterpret_cast<const char*>(g_zero_line);
    if (m < p.M) {
      const int img = m / hw;
      const int rem = m - img * hw;
      const int my = rem / g.MW, mx = rem - my * g.MW;
      const int iy0 = my * g.iys, ix0 = mx * g.ixs;
      base = x + (((long long)(img * g.IH + iy0) * g.IW + ix0) * g.ldi + g.x_coff) * 2 + c * 16;
      if (g.pix_shift < 31) {
        // multi-pixel run (first layers): bits 0..15 = row validity per tap, bits 16..23 = pixel validity inside the run
        for (int t = 0; t < g.ntaps; ++t) mask |= ((unsigned)(iy0 + g.dy[t]) < (unsigned)g.IH ? 1u : 0u) << t;
        const int ppc = (16 / 2) >> g.pix_shift;  // pixels per 16-byte chunk (1 for 8-halves pixels)
        for (int q = 0; q < 8; ++q) {
          const int px = ix0 + g.dx[0] + q;
          mask |= ((unsigned)px < (unsigned)g.IW ? 1u : 0u) << (16 + q);
        }
        (void)ppc;
      } else {
        for (int t = 0; t < g.ntaps; ++t)
          mask |= (((unsigned)(iy0 + g.dy[t]) < (unsigned)g.IH && (unsigned)(ix0 + g.dx[t]) < (unsigned)g.IW) ? 1u : 0u) << t;
        mask |= 0xff0000u;
      }
    }
    a_base[j] = base;
    a_mask[j] = mask;
  }
  const char* b_base[BI];
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int r = min((wid * BI + j) * RPI + lrow, BN - 1);
    const int c = cpos ^ ((r >> SWZ_SHIFT) & (CPR - 1));
    b_base[j] = w + ((long long)(n0 + r) * p.Ktot) * 2 + c * 16;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const bool multi = g.pix_shift < 31;

  auto issue_tile = [&](int kk, int stage) {
    const int tap = kk / p.ksteps_per_tap;
    const int kin = kk - tap * p.ksteps_per_tap;
    const long long uoff = p.tap_off[tap] + (long long)kin * BKB;  // wave-uniform
    char* As = smem + stage * STAGE;
    char* Bs = As + BM * BKB;
#pragma unroll
    for (int j = 0; j < AI; ++j) {
      bool ok = (a_mask[j] >> tap) & 1u;
      if (multi) {
        // the chunk this lane fetches covers pixel (kin*BKB/16 + source chunk) of the run; source chunk = cpos ^ swz
        const int r = (wid * AI + j) * RPI + lrow;
        const int c = cpos ^ ((r >> SWZ_SHIFT) & (CPR - 1));
        ok = ok && ((a_mask[j] >> (16 + kin * CPR + c)) & 1u);
      }
      const char* src = ok ? a_base[j] + uoff : reinterpret_cast<const char*>(g_zero_line);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(As + (wid * AI + j) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
      if (BTOT % 4 != 0 && wid * BI + j >= BTOT) continue;  // wave-uniform
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_base[j] + (long long)kk * BKB),
                                       (__attribute__((address_space(3))) void*)(Bs + (wid * BI + j) * 1024), 16, 0, 0);
    }
  };

  if (p.nk > 0) issue_tile(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // fragment read addressing: lane reads row (lane&31) of its 32-row sub-tile, k-group (lane>>5) of each k16 sub-step
  const int frow_a = wm * TM * 32 + (lane & 31);
  const int frow_b = wn * TN * 32 + (lane & 31);

  for (int kk = 0; kk < p.nk; ++kk) {
    const int stage = kk & 1;
    if (kk + 1 < p.nk) issue_tile(kk + 1, stage ^ 1);
    const char* As = smem + stage * STAGE;
    const char* Bs = As + BM * BKB;
#pragma unroll
    for (int ks = 0; ks < KSUB; ++ks) {
      half8_t a[TM], b[TN];
      const int cidx = ks * 2 + (lane >> 5);
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int r = frow_a + t * 32;
        a[t] = *reinterpret_cast<const half8_t*>(As + r * BKB + ((cidx ^ ((r >> SWZ_SHIFT) & (CPR - 1))) << 4));
      }
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        const int r = frow_b + t * 32;
        b[t] = *reinterpret_cast<const half8_t*>(Bs + r * BKB + ((cidx ^ ((r >> SWZ_SHIFT) & (CPR - 1))) << 4));
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue (as v1): BN statistics from the accumulators, bias, ReLU, store ----
  for (int r = tid; r < BM; r += 256) {  // output pixel of every tile row
    const int m = m0 + r;
    int op = -1;
    if (m < p.M) {
      const int img = m / hw;
      const int rem = m - img * hw;
      const int my = rem / g.MW, mx = rem - my * g.MW;
      op = (img * g.OH + my * g.oys + g.oy0) * g.OW + mx * g.oxs + g.ox0;
    }
    opix_s[r] = op;
  }

  if (p.stats != nullptr) {
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[tm][tn][r];
          s += v;
          q += v * v;
        }
      s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 32, 64);
      if (lane < 32) {
        const int col = wn * TN * 32 + tn * 32 + lane;
        red[(wm * BN + col) * 2 + 0] = s;
        red[(wm * BN + col) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < g.N) {
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        s += (double)red[(a * BN + tid) * 2 + 0];
        q += (double)red[(a * BN + tid) * 2 + 1];
      }
      double* st = p.stats + (size_t)(lb % AM_STATS_REPLICAS) * 2 * g.N;
      atomicAdd(st + n0 + tid, s);
      atomicAdd(st + g.N + n0 + tid, q);
    }
  }
  __syncthreads();
  {
    // staged epilogue (see conv_gemm.hip): whole 16-byte chunks of pixel rows instead of 2-byte scattered stores
    constexpr int SP = TN * 64 + 16;
    char* stg = smem + 8192 + wid * (TM * 32) * SP;  // past the stats scratch and the output-pixel table
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = n0 + wn * TN * 32 + tn * 32 + (lane & 31);
      const float bv = (p.bias != nullptr && col < g.N) ? p.bias[col] : 0.f;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          float v = acc[tm][tn][r] + bv;
          if (p.relu) v = fmaxf(v, 0.f);
          *reinterpret_cast<half_t*>(stg + row * SP + (tn * 32 + (lane & 31)) * 2) = (half_t)v;
        }
    }
    // the wave reads back what its own lanes wrote: LDS executes a wave's accesses in order, so draining the
    // writes is enough; the asm also stops the compiler from moving the (differently typed) reads above the writes
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    constexpr int CPRW = TN * 4;
    const int ncols = (g.N + 7) & ~7;
#pragma unroll
    for (int it = 0; it < TM * TN * 2; ++it) {
      const int q = it * 64 + lane;
      const int row = q / CPRW, cc = q - row * CPRW;
      const int op = opix_s[wm * TM * 32 + row];
      const int col0 = n0 + wn * TN * 32 + cc * 8;
      if (op >= 0 && col0 < ncols)
        *reinterpret_cast<uint4*>(y + (size_t)op * g.ldo + g.y_coff + col0) = *reinterpret_cast<const uint4*>(stg + row * SP + cc * 16);
    }
  }
}


template <int BM, int BN, int WM, int WN, int BKB>
int launch2(const Conv2Params& p0, hipStream_t s) {
  constexpr int STAGE = (BM + BN) * BKB;
  Conv2Params p = p0;
  p.mtiles = am_cdiv(p.M, BM);
  p.ntiles = am_cdiv(p.g.N, BN);
  p.ksteps_per_tap = p.g.krun * 2 / BKB;
  p.nk = p.g.ntaps * p.ksteps_per_tap;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr size_t EPI = 8192 + 4 * (size_t)(TM * 32) * (TN * 64 + 16);  // staged epilogue (reuses the stage buffers)
  const size_t lds = 2 * STAGE > EPI ? 2 * STAGE : EPI;
  static bool attr_done_dev[AM_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[am_current_device()];
  if (lds > 64 * 1024 && !attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm2_k<BM, BN, WM, WN, BKB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return AM_ERR_LAUNCH;
    attr_done = true;
  }
  g_am_conv_variant = AM_CV_LDSDMA_V2;
  hipLaunchKernelGGL((conv_gemm2_k<BM, BN, WM, WN, BKB>), dim3(p.mtiles * p.ntiles), dim3(256), lds, s, p);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

}  // namespace am2

// Called by am_conv_gemm (conv_gemm.hip) for f16 problems; returns AM_ERR_UNSUPPORTED when the shape is not covered
// so the caller falls back to the register-staged kernel.
int am_conv_ring_f16(const am_conv_geom* g, const void* x, const void* w, const float* bias, int relu, const void* res, void* y, double* stats,
                     hipStream_t s);  // conv_ring.hip

int am_conv_gemm2_f16(const am_conv_geom* g, const void* x, const void* w, const float* bias, int relu, void* y, double* stats,
                      hipStream_t s) {
  using namespace am2;
  if (g->ntaps <= 0) return AM_ERR_UNSUPPORTED;
  const int run_bytes = g->krun * 2;
  Conv2Params p;
  p.g = *g;
  p.x = x; p.w = w; p.y = y; p.bias = bias; p.stats = stats;
  p.M = g->B * g->MH * g->MW;
  p.Ktot = g->ntaps * g->krun;
  p.relu = relu;
  p.mtiles = p.ntiles = p.nk = p.ksteps_per_tap = 0;
  for (int t = 0; t < AM_MAX_TAPS; ++t)
    p.tap_off[t] = t < g->ntaps ? ((long long)g->dy[t] * g->IW + g->dx[t]) * (long long)g->ldi * 2 : 0;
  if (g->pix_shift < 31) {
    // multi-pixel run: every tap must start at the same dx and a chunk must be exactly one pixel (8 halves)
    if (g->pix_shift != 3 || g->krun != 64) return AM_ERR_UNSUPPORTED;
    for (int t = 1; t < g->ntaps; ++t) if (g->dx[t] != g->dx[0]) return AM_ERR_UNSUPPORTED;
  }
  {
    // Ring kernels (3-stage LDS-DMA pipeline).  The layers are LDS-bandwidth bound at one fragment read per MFMA (64x64 per
    // wave), so prefer 128x64 per wave (0.75 reads per MFMA) whenever the grid still fills the 256 CUs.
    const int rc = am_conv_ring_f16(g, x, w, bias, relu, nullptr, y, stats, s);
    if (rc != AM_ERR_UNSUPPORTED) return rc;
  }
  if (run_bytes % 128 == 0) {
    if (g->N > 64) return launch2<128, 128, 2, 2, 128>(p, s);
    if (g->N > 32) return launch2<256, 64, 4, 1, 128>(p, s);
    return launch2<256, 32, 4, 1, 128>(p, s);
  }
  if (run_bytes % 64 == 0) {
    if (g->N > 64) return launch2<128, 128, 2, 2, 64>(p, s);
    if (g->N > 32) return launch2<256, 64, 4, 1, 64>(p, s);
    return launch2<256, 32, 4, 1, 64>(p, s);
  }
  return AM_ERR_UNSUPPORTED;
}
