// Convolution as implicit GEMM on gfx950 matrix cores: forward / dgrad ("gather-GEMM") and wgrad.
//
// Data layout in HBM: activations NHWC (pixel-major, channels contiguous), packed weights
// [N][taps][run] (k contiguous).  A "tap" gathers one contiguous run of `krun` elements per output
// row, so every A-operand row of a K-step is one 64-byte slab of one input pixel (or of 8 adjacent
// pixels for the 3-channel first layers).  Zero padding = slabs whose pixel is outside the image.
//
// Forward/dgrad kernel: BM x BN output tile per 256-thread workgroup (4 waves), K-step 64 bytes,
// register-staged global->LDS double buffer (one barrier per K-step), LDS rows padded to 80 B so
// ds_read_b128 fragment reads are conflict-free, MFMA 32x32x16 f16 (or exact-f32 32x32x2) with
// fp32 accumulators.  Epilogue: BatchNorm batch statistics (per-channel sum / sum of squares of
// the accumulator, fp64 atomics into 16 replicas), bias, ReLU, store.
//
// Wgrad kernel: both operands are pixel-major, the reduction runs over pixels, so fragments are
// read from LDS with the hardware transpose read ds_read_b64_tr_b16 (f16) or plain b32 (f32).
// Split over pixel chunks, fp32 atomic accumulation into dW.
#include "am_common.h"
#include <cstdlib>

namespace {

struct ConvKParams {
  am_conv_geom g;
  const void* x;
  const void* w;
  void* y;
  const float* bias;
  double* stats;
  int M, nk, ksteps_per_tap, Ktot, relu, mtiles, ntiles;
  long long tap_off[AM_MAX_TAPS];  // element offset of tap t relative to a row's base pixel: (dy*IW + dx)*ldi
};

__device__ __attribute__((aligned(64))) unsigned char g_zero_line_v1[64];  // zero-padding source of the fast loader

struct RowInfo {
  int img;  // image index, -1 when the row is past M
  int iy0, ix0;
};

__device__ __forceinline__ void decode_row(const am_conv_geom& g, int m, int M, RowInfo& r, int& opix) {
  if (m < M) {
    const int hw = g.MH * g.MW;
    const int img = m / hw;
    const int rem = m - img * hw;
    const int my = rem / g.MW;
    const int mx = rem - my * g.MW;
    r.img = img;
    r.iy0 = my * g.iys;
    r.ix0 = mx * g.ixs;
    opix = (img * g.OH + my * g.oys + g.oy0) * g.OW + mx * g.oxs + g.ox0;
  } else {
    r.img = -1;
    r.iy0 = r.ix0 = 0;
    opix = -1;
  }
}

// 16-byte slab chunk of the gathered operand: row `r`, tap offsets (dy,dx), element offset `roff`
// inside the run.  Out-of-image pixels read as zero.
template <typename T>
__device__ __forceinline__ uint4 gather_chunk(const am_conv_geom& g, const T* __restrict__ x, const RowInfo& r,
                                              int dy, int dx, int roff) {
  const int iy = r.iy0 + dy;
  const int ixb = r.ix0 + dx;
  const int ix = ixb + (roff >> g.pix_shift);
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (r.img >= 0 && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW) {
    const long long pix = (long long)(r.img * g.IH + iy) * g.IW + ixb;
    v = *reinterpret_cast<const uint4*>(x + pix * g.ldi + g.x_coff + roff);
  }
  return v;
}

template <typename T, int BM, int BN, int WM, int WN, bool MULTI>
__global__ __launch_bounds__(WM * WN * 64) void conv_gemm_k(const ConvKParams p) {
  constexpr int NTH = WM * WN * 64;        // 4 or 8 waves
  constexpr int RPP = NTH / 4;             // tile rows covered per loader pass (4 threads per 64-byte row)
  constexpr int EPC = 16 / (int)sizeof(T);  // elements per 16-byte chunk
  constexpr int BK = 64 / (int)sizeof(T);   // elements per K-step
  constexpr int PITCH = 80;                 // LDS row pitch in bytes (64 + 16 pad)
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AR = BM / RPP;
  constexpr int BCH = (BN * 4 + NTH - 1) / NTH;
  constexpr int STAGE = (BM + BN) * PITCH;
  static_assert((WM * WN == 4 || WM * WN == 8) && TM >= 1 && TN >= 1 && AR >= 1, "4 or 8 waves per workgroup");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int EPI_BYTES = sizeof(T) == 2 ? WM * WN * (TM * 32) * (TN * 64 + 16) : 0;
  int* opix_s = reinterpret_cast<int*>(smem + (2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES));

  const am_conv_geom& g = p.g;
  const T* __restrict__ x = static_cast<const T*>(p.x);
  const T* __restrict__ w = static_cast<const T*>(p.w);
  T* __restrict__ y = static_cast<T*>(p.y);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int lb = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
  const int mt = lb / p.ntiles, nt = lb - mt * p.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int chunk = tid & 3;

  RowInfo rows[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    int dummy;
    decode_row(g, m0 + (tid >> 2) + RPP * i, p.M, rows[i], dummy);
  }
  for (int r = tid; r < BM; r += NTH) {
    RowInfo tmp;
    int op;
    decode_row(g, m0 + r, p.M, tmp, op);
    opix_s[r] = op;
  }
  // Hoisted loader state (single-pixel runs): per row a base pointer and a tap-validity bitmask, so a K-step costs one
  // 64-bit add and one select per 16-byte chunk instead of re-deriving pixel coordinates (the kernel was VALU-bound:
  // 9 VALU instructions per MFMA measured with SQ_INSTS_VALU / SQ_INSTS_MFMA).
  constexpr bool multi = MULTI;  // runs that span several pixels (3-channel first layers) keep the general gather
  const T* a_base[AR];
  unsigned a_mask[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    a_base[i] = reinterpret_cast<const T*>(g_zero_line_v1);
    a_mask[i] = 0u;
    if (!multi && rows[i].img >= 0) {
      a_base[i] = x + ((long long)(rows[i].img * g.IH + rows[i].iy0) * g.IW + rows[i].ix0) * g.ldi + g.x_coff + chunk * EPC;
      for (int t = 0; t < g.ntaps; ++t)
        a_mask[i] |= (((unsigned)(rows[i].iy0 + g.dy[t]) < (unsigned)g.IH && (unsigned)(rows[i].ix0 + g.dx[t]) < (unsigned)g.IW) ? 1u : 0u) << t;
    }
  }
  int ld_tap = 0, ld_kin = 0;  // (tap, K-step inside the tap) of the NEXT tile to load: advanced incrementally

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  double dacc[sizeof(T) == 4 ? TM : 1][sizeof(T) == 4 ? TN : 1][16];  // fp32 parity mode: master accumulator (see the K-loop)
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[a][b][r] = 0.0;
  }

  uint4 ra0, ra1, ra2, ra3, rb0, rb1;  // scalars, not an array: hipcc keeps an indexed array captured by the lambdas in scratch
  static_assert(AR <= 4, "A loader handles at most 4 rows per thread");
  static_assert(BCH <= 2, "B tile loader handles at most 128 rows");

  auto load_tile = [&](int kk) {
    const int tap = ld_tap, kin = ld_kin;
    if (++ld_kin == p.ksteps_per_tap) { ld_kin = 0; ++ld_tap; }
    if constexpr (!multi) {
      const long long uoff = p.tap_off[tap] + (long long)kin * BK;  // wave-uniform
      auto fetch = [&](int i) {
        const T* src = ((a_mask[i] >> tap) & 1u) ? a_base[i] + uoff : reinterpret_cast<const T*>(g_zero_line_v1);
        return *reinterpret_cast<const uint4*>(src);
      };
      ra0 = fetch(0);
      if constexpr (AR > 1) ra1 = fetch(1);
      if constexpr (AR > 2) ra2 = fetch(2);
      if constexpr (AR > 3) ra3 = fetch(3);
    } else {
      const int roff = kin * BK + chunk * EPC;
      const int dy = g.dy[tap], dx = g.dx[tap];
      ra0 = gather_chunk<T>(g, x, rows[0], dy, dx, roff);
      if constexpr (AR > 1) ra1 = gather_chunk<T>(g, x, rows[1], dy, dx, roff);
      if constexpr (AR > 2) ra2 = gather_chunk<T>(g, x, rows[2], dy, dx, roff);
      if constexpr (AR > 3) ra3 = gather_chunk<T>(g, x, rows[3], dy, dx, roff);
    }
    {
      // BN*4 chunks per K-step; when BN*4 < 256 the upper threads re-read a valid row (never stored)
      const T* wk = w + (size_t)kk * BK + (tid & 3) * EPC;
      rb0 = *reinterpret_cast<const uint4*>(wk + (size_t)(n0 + ((tid % (BN * 4)) >> 2)) * p.Ktot);
      if constexpr (BCH > 1) rb1 = *reinterpret_cast<const uint4*>(wk + (size_t)(n0 + RPP + (tid >> 2)) * p.Ktot);
    }
  };
  auto store_tile = [&](int stage) {
    char* As = smem + stage * STAGE;
    char* Bs = As + BM * PITCH;
    char* arow = As + (tid >> 2) * PITCH + chunk * 16;
    *reinterpret_cast<uint4*>(arow) = ra0;
    if constexpr (AR > 1) *reinterpret_cast<uint4*>(arow + RPP * PITCH) = ra1;
    if constexpr (AR > 2) *reinterpret_cast<uint4*>(arow + 2 * RPP * PITCH) = ra2;
    if constexpr (AR > 3) *reinterpret_cast<uint4*>(arow + 3 * RPP * PITCH) = ra3;
    if (BN * 4 >= NTH || tid < BN * 4) *reinterpret_cast<uint4*>(Bs + (tid >> 2) * PITCH + (tid & 3) * 16) = rb0;
    if constexpr (BCH > 1) *reinterpret_cast<uint4*>(Bs + (RPP + (tid >> 2)) * PITCH + (tid & 3) * 16) = rb1;
  };

  if (p.nk > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();

  for (int kk = 0; kk < p.nk; ++kk) {
    const int stage = kk & 1;
    if (kk + 1 < p.nk) load_tile(kk + 1);
    const char* As = smem + stage * STAGE + (wm * TM * 32 + (lane & 31)) * PITCH;
    const char* Bs = smem + stage * STAGE + BM * PITCH + (wn * TN * 32 + (lane & 31)) * PITCH;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        half8_t a[TM], b[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) a[t] = *reinterpret_cast<const half8_t*>(As + t * 32 * PITCH + ks * 32 + (lane >> 5) * 16);
#pragma unroll
        for (int t = 0; t < TN; ++t) b[t] = *reinterpret_cast<const half8_t*>(Bs + t * 32 * PITCH + ks * 32 + (lane >> 5) * 16);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
      }
    } else {
      // fp32 parity mode: the 16 products of a K-step are summed by the exact-fp32 MFMA chain into a partial,
      // the partials into a double master accumulator.  One fp32 chain over the whole contraction (2,304 dependent MFMA steps
      // for a 512-channel 3x3 layer) left every block output ~1.6x farther from an fp64 run than torch-CPU's vectorised FMA
      // sums (scratch/dbg_bn_b16.py, round 3) -- and with train-mode BatchNorm + ReLU that noise decides which near-zero
      // activations flip, i.e. whether a gradient lands 1e-6 or 5e-3 from the fp64 result.
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        float a[TM], b[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) a[t] = *reinterpret_cast<const float*>(As + t * 32 * PITCH + (ks * 2 + (lane >> 5)) * 4);
#pragma unroll
        for (int t = 0; t < TN; ++t) b[t] = *reinterpret_cast<const float*>(Bs + t * 32 * PITCH + (ks * 2 + (lane >> 5)) * 4);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
      }
      {  // fold every K-step (every fourth was tried: ~7 % faster in this mode, but the longer fp32 chain showed up as extra ReLU flips in the parity tests)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              dacc[tm][tn][r] += (double)acc[tm][tn][r];
              acc[tm][tn][r] = 0.f;
            }
      }
    }
    if (kk + 1 < p.nk) store_tile(stage ^ 1);
    __syncthreads();
  }

  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = (float)dacc[a][b][r];  // ONE rounding of the whole contraction
  }

  // ---- epilogue: BN statistics (pre-bias accumulator), bias, ReLU, store ----
  if (p.stats != nullptr) {
    // (per-lane partials in double in fp32 parity mode, am_common.h am_stat_acc: torch-CPU's accumulation type)
    using S = typename am_stat_acc<T>::type;
    S* red = reinterpret_cast<S*>(smem);  // [WM][BN][2]; stage buffers are free after the last barrier
    static_assert(WM * BN * 2 * (int)sizeof(S) <= 2 * STAGE, "statistics scratch must fit the stage buffers");
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      S s = (S)0, q = (S)0;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const S v = (S)acc[tm][tn][r];
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
  if constexpr (sizeof(T) == 2) {
    // f16: stage the wave's output tile in LDS and write it out as whole 16-byte chunks of pixel rows.  (Direct
    // stores from the MFMA layout are 2 bytes per lane, 64 B per row segment: measured ~1 TB/s, the largest single
    // cost of the kernel.)  Pad columns (N..round_up(N,8)) receive exact zeros: zero weights, zero bias.
    constexpr int SP = TN * 64 + 16;  // staging row pitch in bytes
    __syncthreads();                  // stats scratch / last K-step reads are done
    char* stg = smem + wid * (TM * 32) * SP;
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
    constexpr int CPRW = TN * 4;  // 16-byte chunks per staged row
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
  } else {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = n0 + wn * TN * 32 + tn * 32 + (lane & 31);
      const bool colok = col < g.N;
      const float bv = (p.bias != nullptr && colok) ? p.bias[col] : 0.f;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int op = opix_s[row];
          float v = acc[tm][tn][r] + bv;
          if (p.relu) v = fmaxf(v, 0.f);
          if (op >= 0 && colok) y[(size_t)op * g.ldo + g.y_coff + col] = am_from_f32<T>(v);
        }
    }
  }
}

template <typename T, int BM, int BN, int WM, int WN, bool MULTI>
int launch_conv_m(const ConvKParams& p, hipStream_t s) {
  constexpr int STAGE = (BM + BN) * 80;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr size_t EPI = sizeof(T) == 2 ? (size_t)WM * WN * (TM * 32) * (TN * 64 + 16) : 0;  // staged f16 epilogue
  ConvKParams q = p;
  q.mtiles = am_cdiv(p.M, BM);
  q.ntiles = am_cdiv(p.g.N, BN);
  const size_t lds = (2 * STAGE > EPI ? 2 * STAGE : EPI) + BM * sizeof(int);
  static bool attr_done_dev[AM_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[am_current_device()];
  if (lds > 64 * 1024 && !attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_k<T, BM, BN, WM, WN, MULTI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return AM_ERR_LAUNCH;
    attr_done = true;
  }
  g_am_conv_variant = AM_CV_REGSTAGED;
  hipLaunchKernelGGL((conv_gemm_k<T, BM, BN, WM, WN, MULTI>), dim3(q.mtiles * q.ntiles), dim3(WM * WN * 64), lds, s, q);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_conv(const ConvKParams& p, hipStream_t s) {
  return p.g.pix_shift < 31 ? launch_conv_m<T, BM, BN, WM, WN, true>(p, s) : launch_conv_m<T, BM, BN, WM, WN, false>(p, s);
}

static int big_tile_mode() { return 1; }

template <typename T>
int dispatch_conv(const ConvKParams& p, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    // 8-wave 256x128 tile: 87 FLOP per byte through the CU's L2->LDS path instead of 64 (the bound of the 128^2 tile)
    if (p.g.N > 64 && p.M >= 256 * 256 && big_tile_mode() == 1) return launch_conv<T, 256, 128, 4, 2>(p, s);
  }
  if constexpr (sizeof(T) == 4) {
    // fp32 parity mode: eight waves per tile -- with the double master accumulators a 4-wave tile needs 192 accumulator registers
    // per lane (one wave per SIMD: 407 -> 260 img/s for the 4a step in this mode), an 8-wave tile 96
    if (p.g.N > 64) return launch_conv<T, 128, 128, 4, 2>(p, s);
    if (p.g.N > 32) return launch_conv<T, 256, 64, 8, 1>(p, s);
    return launch_conv<T, 256, 32, 4, 1>(p, s);
  }
  if (p.g.N > 64) return launch_conv<T, 128, 128, 2, 2>(p, s);
  if (p.g.N > 32) return launch_conv<T, 256, 64, 4, 1>(p, s);
  return launch_conv<T, 256, 32, 4, 1>(p, s);
}

int check_geom(const am_conv_geom* g, int dtype) {
  if (!g || (dtype != AM_F32 && dtype != AM_F16)) return AM_ERR_ARG;
  const int es = dtype == AM_F16 ? 2 : 4;
  if (g->ntaps < 0 || g->ntaps > AM_MAX_TAPS || g->N <= 0 || g->B < 0) return AM_ERR_ARG;
  if (g->krun <= 0 || (g->krun * es) % 64 != 0) return AM_ERR_ARG;
  if ((g->ldi * es) % 16 != 0 || (g->x_coff * es) % 16 != 0) return AM_ERR_ARG;
  if (g->pix_shift < 0 || g->pix_shift > 31) return AM_ERR_ARG;
  if ((long long)g->B * g->MH * g->MW > 0x7fffffffLL || (long long)g->B * g->OH * g->OW > 0x7fffffffLL) return AM_ERR_ARG;
  if (g->osplit < 0 || (g->osplit > 0 && (g->N != 2 * g->osplit || g->osplit % 8 != 0 || g->osplit_stride % 8 != 0))) return AM_ERR_ARG;
  return AM_OK;
}

// --------------------------------------------------------------------------------------------
// wgrad
// --------------------------------------------------------------------------------------------
struct WgradParams {
  am_conv_geom g;
  const void* x;
  const void* dy;
  float* dw;
  float* ws;            // slab mode (am_conv_wgrad_ws): chunk c stores its unscaled partial tile at ws + c * ws_stride
  long long ws_stride;
  float scale;
  int M, nk, ksteps_per_tap, Ktot, ktiles, ntiles, mchunks, mc;
};

template <typename T, int BNO, int NS>
__global__ __launch_bounds__(256) void conv_wgrad_k(const WgradParams p) {
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int BK = 64 / (int)sizeof(T);  // elements per 64-byte slab
  constexpr int BKC = NS * BK;             // k-columns per tile
  constexpr int PS = sizeof(T) == 2 ? 64 : 32;  // pixels per step
  constexpr int DYB = BNO * (int)sizeof(T);     // dY row bytes
  constexpr int PDY = DYB + 64;                 // pitches == 64 (mod 256): conflict-free transposed reads
  constexpr int PX = NS * 64 + 64;
  constexpr int TMN = BNO / 2 / 32, TNK = BKC / 2 / 32;
  constexpr int DCH = DYB / 16 / 4;  // dY chunks per thread (4 threads per pixel row)
  static_assert(TMN >= 1 && TNK >= 1, "tile too small");
  static_assert(PDY % 128 == 64 && PX % 128 == 64, "pitch");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* dYs = smem;
  char* Xs = smem + PS * PDY;

  const am_conv_geom& g = p.g;
  const T* __restrict__ x = static_cast<const T*>(p.x);
  const T* __restrict__ dy = static_cast<const T*>(p.dy);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wn = wid >> 1, wk = wid & 1;
  const int tiles = p.ktiles * p.ntiles;
  const int lb = xcd_remap(blockIdx.x, tiles * p.mchunks);
  const int mcid = lb / tiles;
  const int tile = lb - mcid * tiles;
  const int nt = tile / p.ktiles, kt = tile - nt * p.ktiles;
  const int n0 = nt * BNO, kk0 = kt * NS;
  const int mbeg = mcid * p.mc;
  const int mend = min(p.M, mbeg + p.mc);

  // slab -> tap geometry (fixed per block)
  int s_dy[NS], s_dx[NS], s_roff[NS];
  bool s_ok[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int kk = kk0 + s;
    s_ok[s] = kk < p.nk;
    const int tap = s_ok[s] ? kk / p.ksteps_per_tap : 0;
    s_dy[s] = g.dy[tap];
    s_dx[s] = g.dx[tap];
    s_roff[s] = (kk - tap * p.ksteps_per_tap) * BK + (tid & 3) * EPC;
  }

  f32x16 acc[TMN][TNK];
#pragma unroll
  for (int a = 0; a < TMN; ++a)
#pragma unroll
    for (int b = 0; b < TNK; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  double dacc[sizeof(T) == 4 ? TMN : 1][sizeof(T) == 4 ? TNK : 1][16];  // fp32 parity mode: master accumulator
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int a = 0; a < TMN; ++a)
#pragma unroll
      for (int b = 0; b < TNK; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[a][b][r] = 0.0;
  }

  // PS*4 loader threads per step: thread -> pixel row (tid>>2), 16-byte lane (tid&3)
  const bool loader = (tid >> 2) < PS;
  uint4 rdy[DCH], rx[NS];

  auto load_step = [&](int mstep) {
    if (!loader) return;
    RowInfo ri;
    int op;
    const int m = mstep + (tid >> 2);
    decode_row(g, m, mend, ri, op);
#pragma unroll
    for (int i = 0; i < DCH; ++i) {
      const int c = (tid & 3) + 4 * i;  // chunk inside the dY row
      const int n = n0 + c * EPC;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (op >= 0 && n < g.N) v = *reinterpret_cast<const uint4*>(dy + (size_t)op * g.ldo + g.y_coff + n);
      rdy[i] = v;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) rx[s] = s_ok[s] ? gather_chunk<T>(g, x, ri, s_dy[s], s_dx[s], s_roff[s]) : make_uint4(0u, 0u, 0u, 0u);
  };
  auto store_step = [&]() {
    if (!loader) return;
    const int r = tid >> 2;
#pragma unroll
    for (int i = 0; i < DCH; ++i) *reinterpret_cast<uint4*>(dYs + r * PDY + ((tid & 3) + 4 * i) * 16) = rdy[i];
#pragma unroll
    for (int s = 0; s < NS; ++s) *reinterpret_cast<uint4*>(Xs + r * PX + s * 64 + (tid & 3) * 16) = rx[s];
  };

  if (mbeg < mend) load_step(mbeg);
  for (int ms = mbeg; ms < mend; ms += PS) {
    __syncthreads();  // previous step's fragment reads are done
    store_step();
    __syncthreads();
    if (ms + PS < mend) load_step(ms + PS);
    if constexpr (sizeof(T) == 2) {
      const int gq = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
#pragma unroll
      for (int ks = 0; ks < PS / 16; ++ks) {
        half8_t a[TMN], b[TNK];
        const int prow = ks * 16 + 8 * (gq >> 1) + q;
#pragma unroll
        for (int t = 0; t < TMN; ++t) {
          const char* base = dYs + prow * PDY + ((wn * TMN + t) * 32 + (gq & 1) * 16 + 4 * pp) * 2;
          s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(base));
          s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(base + 4 * PDY));
          half4_t l4 = __builtin_bit_cast(half4_t, lo), h4 = __builtin_bit_cast(half4_t, hi);
          a[t] = half8_t{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
        }
#pragma unroll
        for (int t = 0; t < TNK; ++t) {
          const char* base = Xs + prow * PX + ((wk * TNK + t) * 32 + (gq & 1) * 16 + 4 * pp) * 2;
          s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(base));
          s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(base + 4 * PX));
          half4_t l4 = __builtin_bit_cast(half4_t, lo), h4 = __builtin_bit_cast(half4_t, hi);
          b[t] = half8_t{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
        }
#pragma unroll
        for (int tm = 0; tm < TMN; ++tm)
#pragma unroll
          for (int tn = 0; tn < TNK; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
      }
    } else {
      // fp32 parity mode: the fp32 chain runs over one 32-pixel step, then folds into a double master (conv_gemm_k's reasoning;
      // the contraction here runs over every pixel of the chunk)
#pragma unroll 4
      for (int ks = 0; ks < PS / 2; ++ks) {
        float a[TMN], b[TNK];
        const int prow = ks * 2 + (lane >> 5);
#pragma unroll
        for (int t = 0; t < TMN; ++t) a[t] = *reinterpret_cast<const float*>(dYs + prow * PDY + ((wn * TMN + t) * 32 + (lane & 31)) * 4);
#pragma unroll
        for (int t = 0; t < TNK; ++t) b[t] = *reinterpret_cast<const float*>(Xs + prow * PX + ((wk * TNK + t) * 32 + (lane & 31)) * 4);
#pragma unroll
        for (int tm = 0; tm < TMN; ++tm)
#pragma unroll
          for (int tn = 0; tn < TNK; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
      }
      {
#pragma unroll
        for (int tm = 0; tm < TMN; ++tm)
#pragma unroll
          for (int tn = 0; tn < TNK; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              dacc[tm][tn][r] += (double)acc[tm][tn][r];
              acc[tm][tn][r] = 0.f;
            }
      }
    }
  }

  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int a = 0; a < TMN; ++a)
#pragma unroll
      for (int b = 0; b < TNK; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = (float)dacc[a][b][r];
  }
  // flush, rows = output channel n, cols = packed k: plain stores into this pixel chunk's slab (slab mode) or fp32 atomics
  float* slab = p.ws ? p.ws + (size_t)mcid * p.ws_stride : nullptr;
#pragma unroll
  for (int tn = 0; tn < TNK; ++tn) {
    const int kc = kk0 * BK + (wk * TNK + tn) * 32 + (lane & 31);
    if (kc >= p.Ktot) continue;
#pragma unroll
    for (int tm = 0; tm < TMN; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + (wn * TMN + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n < g.N) {
          if (slab) slab[(size_t)n * p.Ktot + kc] = acc[tm][tn][r];
          else atomicAdd(p.dw + (size_t)n * p.Ktot + kc, acc[tm][tn][r] * p.scale);
        }
      }
  }
}

// Second pass of am_conv_wgrad_ws: out = (accumulate ? out : 0) + scale * sum over the pixel chunks' slabs, written in the
// nn.Conv2d weight layout out[n][c][t] (OIHW: t = kh * KW + kw) from the packed k = t * krun + c.  The re-layout is a permutation
// INSIDE an output-channel row, so a workgroup owns (a part of) one row: slab reads coalesced along k, the permutation through
// LDS, the read-modify-write of `out` coalesced again.  grid = (parts, N); a part covers whole taps' worth of channels.
// wr_ch = channels per part: a part handles channels [c0, c0 + wr_ch) of every tap -> 4 * wr_ch * ntaps floats of LDS.  32 by default;
// 16 when the slabs are many and the rows few (conv_patch_wgrad_k: 256 slabs x 64 rows -- twice the workgroups to hide the latency
// of 64 dependent slab reads per wave; a part's reads are then 64-byte pieces)

__global__ __launch_bounds__(1024) void wgrad_reduce_k(const float* __restrict__ ws, int nchunks, long long ws_stride, int Ktot, int krun,
                                                      int cin, int ntaps, float scale, float* __restrict__ out, int accumulate, int wr_ch) {
  extern __shared__ float row[];  // [blockDim / 64 chunk groups][cpart * ntaps] in output order
  const int n = blockIdx.y, c0 = blockIdx.x * wr_ch;
  const int cpart = min(wr_ch, cin - c0);
  const int nel = cpart * ntaps;
  const float* src = ws + (size_t)n * Ktot;
  // wave w of nwv sums the chunks s = w, w + nwv, ...: a wave-instruction reads 64 consecutive packed k of one slab (256 B), eight
  // of them in flight per lane (a layer with few output channels has hundreds of short slabs: latency, not bytes, is its bound).
  // nwv = blockDim.x / 64: 4 by default, 16 for the per-workgroup slabs of conv_patch_wgrad_k (80-256 of them: with 4 waves every
  // lane ran 8 dependent rounds of loads, 25 us for 38 MB)
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, nwv = blockDim.x >> 6;
  for (int i = lane; i < nel; i += 64) {
    const int t = i / cpart, c = i - t * cpart;
    const float* col = src + t * krun + c0 + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    int sidx = w;
    for (; sidx + 7 * nwv < nchunks; sidx += 8 * nwv) {
      a0 += col[(size_t)sidx * ws_stride];
      a1 += col[(size_t)(sidx + nwv) * ws_stride];
      a2 += col[(size_t)(sidx + 2 * nwv) * ws_stride];
      a3 += col[(size_t)(sidx + 3 * nwv) * ws_stride];
      a4 += col[(size_t)(sidx + 4 * nwv) * ws_stride];
      a5 += col[(size_t)(sidx + 5 * nwv) * ws_stride];
      a6 += col[(size_t)(sidx + 6 * nwv) * ws_stride];
      a7 += col[(size_t)(sidx + 7 * nwv) * ws_stride];
    }
    for (; sidx < nchunks; sidx += nwv) a0 += col[(size_t)sidx * ws_stride];
    row[w * nel + c * ntaps + t] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  }
  __syncthreads();
  float* o = out + ((size_t)n * cin + c0) * ntaps;
  for (int i = threadIdx.x; i < nel; i += blockDim.x) {
    float v = (row[i] + row[nel + i]) + (row[2 * nel + i] + row[3 * nel + i]);
    for (int g = 4; g < nwv; g += 4) v += (row[g * nel + i] + row[(g + 1) * nel + i]) + (row[(g + 2) * nel + i] + row[(g + 3) * nel + i]);
    v *= scale;
    o[i] = accumulate ? o[i] + v : v;
  }
}

template <typename T, int BNO, int NS>
int launch_wgrad(const WgradParams& p0, hipStream_t s, bool plan_only = false) {
  constexpr int PS = sizeof(T) == 2 ? 64 : 32;
  constexpr int PDY = BNO * (int)sizeof(T) + 64, PX = NS * 64 + 64;
  WgradParams p = p0;
  p.ktiles = am_cdiv(p.nk, NS);
  p.ntiles = am_cdiv(p.g.N, BNO);
  const int tiles = p.ktiles * p.ntiles;
  // aim for ~1024 workgroups (4 per CU); each pixel chunk a multiple of the step
  int mchunks = 1024 / tiles;
  if (mchunks < 1) mchunks = 1;
  int mc = am_cdiv(p.M, mchunks);
  mc = am_cdiv(mc, PS * 4) * PS * 4;  // at least 4 steps per chunk
  p.mc = mc;
  p.mchunks = am_cdiv(p.M, mc);
  if (plan_only) return p.mchunks;
  const size_t lds = (size_t)PS * (PDY + PX);
  g_am_conv_variant = AM_CV_WGRAD_REGSTAGED;
  hipLaunchKernelGGL((conv_wgrad_k<T, BNO, NS>), dim3(tiles * p.mchunks), dim3(256), lds, s, p);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

}  // namespace

int am_conv_wgrad_ring_f16(const am_conv_geom* g, const void* x, const void* dy, float scale, float* dw, float* ws, long long ws_stride,
                           bool plan_only, hipStream_t s);  // conv_wgrad_ring.hip
int am_conv_patch_wgrad_f16(const am_conv_geom* g, const void* x, const void* dy, float scale, float* dw, float* ws, long long ws_stride,
                            bool plan_only, hipStream_t s);  // conv_patch_wgrad.hip
int am_conv_s2d_wgrad_f16(const am_conv_geom* g, const void* x, const void* dy, const void* yout, const void* raw, const float* mean,
                          const float* rstd, const float* coef, int relu, const float* sg_scale, const float* sg_shift, float scale, float* dw,
                          hipStream_t s);  // conv_s2d_wgrad.hip
int am_conv_gemm2_f16(const am_conv_geom* g, const void* x, const void* w, const float* bias, int relu, void* y, double* stats,
                      hipStream_t s);  // conv_gemm2.hip
int am_conv_s2d_f16(const am_conv_geom* g, int mode, const void* x, const void* w, const float* bias, const float* scale,
                    const float* shift, int relu, void* y, double* stats, hipStream_t s);  // conv_s2d.hip
int am_conv3x3_c64n64_duo_f16(const am_conv_geom* g, const void* x, const void* w, const float* bias, int relu, const void* res, void* y,
                              double* stats, hipStream_t s);  // conv_patch3.hip

thread_local int g_am_conv_variant = AM_CV_NONE;

extern "C" int am_conv_last_variant(void) { return g_am_conv_variant; }

static int g_tuning[AM_TUNE_COUNT] = {
    /* AM_TUNE_RING */ 4,
    /* AM_TUNE_RING128_MIN_TILES */ 100,
    /* AM_TUNE_WGRAD_RING */ 1,
    /* AM_TUNE_WGRAD_MAX_SLABS */ 32,
    /* AM_TUNE_RING_SHORT_K */ 8,
    /* AM_TUNE_HALO_MIN_TILES */ 256,
    /* AM_TUNE_PATCH_WGRAD_MIN_TILES */ 512,
    /* AM_TUNE_PATCH_WGRAD_C128 */ 1,
    /* AM_TUNE_DUO_MFMA16 */ 1,
    /* AM_TUNE_BAND_MIN_TILES */ 200,
    /* AM_TUNE_RING_DIAG */ 0,
    /* AM_TUNE_RING16_M128_MIN_TILES */ 200,
};

int am_tuning(int key) { return key >= 0 && key < AM_TUNE_COUNT ? g_tuning[key] : 0; }

extern "C" int am_set_tuning(int key, int value) {
  if (key < 0 || key >= AM_TUNE_COUNT) return AM_ERR_ARG;
  const int old = g_tuning[key];
  g_tuning[key] = value;
  return old;
}

extern "C" int am_get_tuning(int key) { return key >= 0 && key < AM_TUNE_COUNT ? g_tuning[key] : AM_ERR_ARG; }

int am_conv_ring_f16(const am_conv_geom* g, const void* x, const void* w, const float* bias, int relu, const void* res, void* y,
                     double* stats, hipStream_t s);  // conv_ring.hip

int am_conv_halo_f16(const am_conv_geom* g, const void* x, const void* w, const float* bias, int relu, const void* res, void* y,
                     double* stats, hipStream_t s);  // conv_halo.hip
int am_conv_band16_f16(const am_conv_geom* g, const void* x, const void* w, const float* bias, int relu, const void* res, void* y,
                       double* stats, hipStream_t s);  // conv_band16.hip
int am_conv_halo_pre_f16(const am_conv_geom* g, const void* x, const float* pre_scale, const float* pre_shift, const void* w,
                         const float* bias, int relu, const void* res, void* y, double* stats, hipStream_t s);  // conv_halo.hip

extern "C" int am_conv_npad(int N) {
  if (N > 64) return am_cdiv(N, 128) * 128;
  if (N > 32) return 64;
  return 32;
}

extern "C" int am_conv_gemm(const am_conv_geom* g, int dtype, const void* x, const void* w, const float* bias,
                            int relu, void* y, double* stats, am_stream_t stream) {
  int rc = check_geom(g, dtype);
  if (rc != AM_OK) return rc;
  if (!y || (g->ntaps > 0 && (!x || !w))) return AM_ERR_ARG;
  const int es = dtype == AM_F16 ? 2 : 4;
  ConvKParams p;
  p.g = *g;
  p.x = x; p.w = w; p.y = y; p.bias = bias; p.stats = stats;
  p.M = g->B * g->MH * g->MW;
  if (p.M == 0) return AM_OK;
  p.ksteps_per_tap = g->krun * es / 64;
  p.nk = g->ntaps * p.ksteps_per_tap;
  p.Ktot = g->ntaps * g->krun;
  p.relu = relu;
  p.mtiles = p.ntiles = 0;
  for (int t = 0; t < AM_MAX_TAPS; ++t)
    p.tap_off[t] = t < g->ntaps ? ((long long)g->dy[t] * g->IW + g->dx[t]) * (long long)g->ldi : 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (g->osplit > 0)  // split output rows (fused stride-2 dgrad): ring kernels only
    return dtype == AM_F16 && bias == nullptr && !relu && stats == nullptr ? am_conv_ring_f16(g, x, w, bias, relu, nullptr, y, stats, s) : AM_ERR_UNSUPPORTED;
  if (dtype == AM_F16) {
    // 3-channel first layers on the space-to-depth image: weights-stationary patch kernel
    rc = am_conv_s2d_f16(g, 0, x, w, bias, nullptr, nullptr, relu, y, stats, s);
    if (rc != AM_ERR_UNSUPPORTED) return rc;
    // 64->64 3x3 layers: weights-stationary patch kernel (per-CU load bandwidth is the bound of the gather form there)
    rc = am_conv3x3_c64n64_duo_f16(g, x, w, bias, relu, nullptr, y, stats, s);
    if (rc != AM_ERR_UNSUPPORTED) return rc;
    // 3x3 / stride-1 layers with 64 < N <= 128 (layer2 and its dgrad): halo-staged patch instead of nine gathers
    rc = am_conv_halo_f16(g, x, w, bias, relu, nullptr, y, stats, s);
    if (rc != AM_ERR_UNSUPPORTED) return rc;
    // 3x3 / stride-1 layers with N = 256 / 512 on maps up to 128 pixels wide (layers 3-4, the heads, their dgrads): row-band halo
    rc = am_conv_band16_f16(g, x, w, bias, relu, nullptr, y, stats, s);
    if (rc != AM_ERR_UNSUPPORTED) return rc;
    // N > 64: the LDS-DMA ring kernels (conv_ring.hip) win at every M; N <= 64 with a large M (policy layers, dgrads
    // into 64 channels) stays on the register-staged kernel
    if ((long long)p.M <= 65536 || g->N > 64) {
      rc = am_conv_gemm2_f16(g, x, w, bias, relu, y, stats, s);
      if (rc != AM_ERR_UNSUPPORTED) return rc;
    }
  }
  return dtype == AM_F16 ? dispatch_conv<half_t>(p, s) : dispatch_conv<float>(p, s);
}

// Shared dispatcher of am_conv_wgrad (atomic form: ws == nullptr) and am_conv_wgrad_ws (slab form).  plan_only: launches nothing,
// returns the number of pixel chunks (= slabs) of the kernel that would run, 0 when that kernel has no slab form.
// *own_slabs (plan_only): the kernel wants one slab per workgroup whatever their number (it has no pixel chunks to merge).
static int wgrad_dispatch(const am_conv_geom* g, int dtype, const void* x, const void* dy, float scale, float* dw, float* ws,
                          long long ws_stride, bool plan_only, hipStream_t s, bool* own_slabs = nullptr) {
  const int es = dtype == AM_F16 ? 2 : 4;
  WgradParams p;
  p.g = *g;
  p.x = x; p.dy = dy; p.dw = dw; p.scale = scale; p.ws = ws; p.ws_stride = ws_stride;
  p.M = g->B * g->MH * g->MW;
  p.ksteps_per_tap = g->krun * es / 64;
  p.nk = g->ntaps * p.ksteps_per_tap;
  p.Ktot = g->ntaps * g->krun;
  p.ktiles = p.ntiles = p.mchunks = p.mc = 0;
  if (dtype == AM_F16) {
    if (g->pix_shift == 4) {  // first layers on the space-to-depth image: dY read once (conv_s2d_wgrad.hip); atomic form only
      if (plan_only || ws) return plan_only ? 0 : AM_ERR_UNSUPPORTED;
      const int rc = am_conv_s2d_wgrad_f16(g, x, dy, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, scale, dw, s);
      if (rc != AM_ERR_UNSUPPORTED) return rc;
    }
    int rc = am_conv_patch_wgrad_f16(g, x, dy, scale, dw, ws, ws_stride, plan_only, s);  // 64 -> 64 channels 3x3 / s1: operands read once
    if (rc != AM_ERR_UNSUPPORTED) {
      if (own_slabs) *own_slabs = true;
      return rc;
    }
    rc = am_conv_wgrad_ring_f16(g, x, dy, scale, dw, ws, ws_stride, plan_only, s);  // LDS-DMA ring kernel: N % 128 == 0, long contractions
    if (rc != AM_ERR_UNSUPPORTED) return rc;
    if (g->N > 64) return launch_wgrad<half_t, 128, 4>(p, s, plan_only);
    return launch_wgrad<half_t, 64, 4>(p, s, plan_only);
  }
  return launch_wgrad<float, 64, 4>(p, s, plan_only);
}

extern "C" int am_conv_wgrad(const am_conv_geom* g, int dtype, const void* x, const void* dy, float scale,
                             float* dw, am_stream_t stream) {
  int rc = check_geom(g, dtype);
  if (rc != AM_OK) return rc;
  if (!x || !dy || !dw || g->ntaps <= 0) return AM_ERR_ARG;
  const int es = dtype == AM_F16 ? 2 : 4;
  if ((g->ldo * es) % 16 != 0 || (g->y_coff * es) % 16 != 0) return AM_ERR_ARG;
  if ((long long)g->B * g->MH * g->MW == 0) return AM_OK;
  return wgrad_dispatch(g, dtype, x, dy, scale, dw, nullptr, 0, false, static_cast<hipStream_t>(stream));
}

// Slabs the workspace form uses for a geometry: the kernel's pixel chunks while they are few (each slab is written and read
// once: S * |dW| * 8 bytes of traffic), else ONE zero-filled slab that the kernel's atomic form accumulates into -- with
// hundreds of chunks (small dW, long contraction) the slabs' traffic costs more than the atomics it removes (measured on the
// layer1 / layer2 / policy shapes: +20..+50 us per launch; layer4 / head shapes with 7-13 slabs: -10..-18 us).
static int ws_slabs(int chunks) { return chunks <= am_tuning(AM_TUNE_WGRAD_MAX_SLABS) ? chunks : 1; }

extern "C" int am_conv_wgrad_workspace_bytes(const am_conv_geom* g, int dtype, long long* bytes) {
  int rc = check_geom(g, dtype);
  if (rc != AM_OK) return rc;
  if (!bytes || g->ntaps <= 0) return AM_ERR_ARG;
  *bytes = 0;
  if ((long long)g->B * g->MH * g->MW == 0) return AM_OK;
  bool own = false;
  const int chunks = wgrad_dispatch(g, dtype, nullptr, nullptr, 1.f, nullptr, nullptr, 0, true, nullptr, &own);
  if (chunks < 0) return chunks;
  *bytes = (long long)(own ? chunks : ws_slabs(chunks)) * g->N * g->ntaps * g->krun * 4;
  return AM_OK;
}

extern "C" int am_conv_wgrad_ws(const am_conv_geom* g, int dtype, const void* x, const void* dy, float scale, void* workspace,
                                long long workspace_bytes, float* dw_oihw, int cin, int accumulate, am_stream_t stream) {
  int rc = check_geom(g, dtype);
  if (rc != AM_OK) return rc;
  if (!x || !dy || !dw_oihw || g->ntaps <= 0 || cin <= 0 || cin > g->krun) return AM_ERR_ARG;
  const int es = dtype == AM_F16 ? 2 : 4;
  if ((g->ldo * es) % 16 != 0 || (g->y_coff * es) % 16 != 0) return AM_ERR_ARG;
  if ((long long)g->B * g->MH * g->MW == 0) return AM_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  bool own = false;
  const int chunks = wgrad_dispatch(g, dtype, nullptr, nullptr, 1.f, nullptr, nullptr, 0, true, nullptr, &own);
  if (chunks < 0) return chunks;
  if (chunks == 0) return AM_ERR_UNSUPPORTED;  // first-layer kernel: atomic form only (caller: am_conv_wgrad + its own unpack)
  const int slabs = own ? chunks : ws_slabs(chunks);
  const long long Ktot = (long long)g->ntaps * g->krun, stride = (long long)g->N * Ktot;
  if (!workspace || workspace_bytes < slabs * stride * 4 || (reinterpret_cast<uintptr_t>(workspace) & 15)) return AM_ERR_ARG;
  float* ws = static_cast<float*>(workspace);
  if (slabs == chunks) {
    rc = wgrad_dispatch(g, dtype, x, dy, 1.f, nullptr, ws, stride, false, s);
  } else {  // many short chunks: atomics into one zeroed slab (a memset node under capture), re-laid out by the same second pass
    if (hipMemsetAsync(ws, 0, (size_t)stride * 4, s) != hipSuccess) return AM_ERR_LAUNCH;
    rc = wgrad_dispatch(g, dtype, x, dy, 1.f, ws, nullptr, 0, false, s);
  }
  if (rc != AM_OK) return rc;
  const int wr_ch = own ? 16 : 32;
  const int parts = am_cdiv(cin, wr_ch);
  const int nwv = own ? 16 : 4;
  const size_t lds = nwv * (size_t)(cin < wr_ch ? cin : wr_ch) * g->ntaps * sizeof(float);
  hipLaunchKernelGGL(wgrad_reduce_k, dim3(parts, g->N), dim3(nwv * 64), lds, s, ws, slabs, stride, (int)Ktot, g->krun, cin, g->ntaps, scale,
                     dw_oihw, accumulate, wr_ch);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

int am_conv3x3_c64n64_duo_pre_f16(const am_conv_geom* g, const void* x, const float* pre_scale, const float* pre_shift, const void* w,
                                  const float* bias, int relu, const void* res, void* y, double* stats, hipStream_t s);  // conv_patch3.hip

// y = act(conv + bias + res): the block end of an inference ResNet block (eval-mode BatchNorm folded into w / bias by the caller).
extern "C" int am_conv_gemm_res(const am_conv_geom* g, int dtype, const void* x, const void* w, const float* bias, const void* res,
                                int relu, void* y, am_stream_t stream) {
  int rc = check_geom(g, dtype);
  if (rc != AM_OK) return rc;
  if (!x || !w || !y || !res || g->ntaps <= 0) return AM_ERR_ARG;
  if (dtype != AM_F16 || g->osplit > 0) return AM_ERR_UNSUPPORTED;
  if ((long long)g->B * g->MH * g->MW == 0) return AM_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  rc = am_conv3x3_c64n64_duo_f16(g, x, w, bias, relu, res, y, nullptr, s);
  if (rc != AM_ERR_UNSUPPORTED) return rc;
  rc = am_conv_halo_f16(g, x, w, bias, relu, res, y, nullptr, s);
  if (rc != AM_ERR_UNSUPPORTED) return rc;
  rc = am_conv_band16_f16(g, x, w, bias, relu, res, y, nullptr, s);
  if (rc != AM_ERR_UNSUPPORTED) return rc;
  return am_conv_ring_f16(g, x, w, bias, relu, res, y, nullptr, s);
}

extern "C" int am_conv_gemm_prebn(const am_conv_geom* g, int dtype, const void* x, const float* pre_scale, const float* pre_shift,
                                  const void* w, void* y, double* stats, am_stream_t stream) {
  int rc = check_geom(g, dtype);
  if (rc != AM_OK) return rc;
  if (!x || !w || !y || !pre_scale || !pre_shift) return AM_ERR_ARG;
  if (dtype != AM_F16) return AM_ERR_UNSUPPORTED;
  rc = am_conv3x3_c64n64_duo_pre_f16(g, x, pre_scale, pre_shift, w, nullptr, 0, nullptr, y, stats, static_cast<hipStream_t>(stream));
  if (rc != AM_ERR_UNSUPPORTED) return rc;
  return am_conv_halo_pre_f16(g, x, pre_scale, pre_shift, w, nullptr, 0, nullptr, y, stats, static_cast<hipStream_t>(stream));
}

extern "C" int am_conv_wgrad_bn(const am_conv_geom* g, int dtype, const void* x, const void* dy, const void* yout, const void* raw,
                                const float* mean, const float* rstd, const float* coef, int relu, float scale, float* dw,
                                am_stream_t stream) {
  int rc = check_geom(g, dtype);
  if (rc != AM_OK) return rc;
  if (!x || !dy || !raw || !mean || !rstd || !coef || !dw || (relu && !yout) || g->ntaps <= 0) return AM_ERR_ARG;
  if (dtype != AM_F16 || g->pix_shift != 4) return AM_ERR_UNSUPPORTED;
  if ((long long)g->B * g->MH * g->MW == 0) return AM_OK;
  return am_conv_s2d_wgrad_f16(g, x, dy, yout, raw, mean, rstd, coef, relu, nullptr, nullptr, scale, dw, static_cast<hipStream_t>(stream));
}

extern "C" int am_conv_wgrad_bn_sign(const am_conv_geom* g, int dtype, const void* x, const void* dy, const void* raw, const float* mean,
                                     const float* rstd, const float* coef, const float* bn_scale, const float* bn_shift, float scale,
                                     float* dw, am_stream_t stream) {
  int rc = check_geom(g, dtype);
  if (rc != AM_OK) return rc;
  if (!x || !dy || !raw || !mean || !rstd || !coef || !bn_scale || !bn_shift || !dw || g->ntaps <= 0) return AM_ERR_ARG;
  if (dtype != AM_F16 || g->pix_shift != 4) return AM_ERR_UNSUPPORTED;
  if ((long long)g->B * g->MH * g->MW == 0) return AM_OK;
  return am_conv_s2d_wgrad_f16(g, x, dy, nullptr, raw, mean, rstd, coef, 1, bn_scale, bn_shift, scale, dw, static_cast<hipStream_t>(stream));
}

extern "C" int am_conv_first_fused(const am_conv_geom* g, int dtype, int mode, const void* x, const void* w, const float* scale,
                                   const float* shift, void* y, double* stats, am_stream_t stream) {
  int rc = check_geom(g, dtype);
  if (rc != AM_OK) return rc;
  if (dtype != AM_F16) return AM_ERR_UNSUPPORTED;
  if (!x || !w || (mode == 1 && !stats) || ((mode == 2 || mode == 3) && (!y || !scale || !shift)) || (mode == 4 && (!y || !scale || !stats)) || mode < 1 || mode > 4)
    return AM_ERR_ARG;
  return am_conv_s2d_f16(g, mode, x, w, nullptr, scale, shift, 1, y, stats, static_cast<hipStream_t>(stream));
}
