// Third-generation forward / dgrad gather-GEMM for f16 ("ring" kernel): same 3-stage LDS-DMA ring as conv_gemm3_k
// (conv_gemm2.hip) with a main loop that is ONE basic block, so the compiler can overlap its parts:
//   * tiles are fetched with buffer_load_dwordx4 ... lds through raw buffer descriptors: a lane whose tap falls outside
//     the image asks for an out-of-range offset and the hardware writes zeros (no zero line, no 64-bit pointer select,
//     no exec-masked branch); weights need no per-lane arithmetic at all (uniform K offset in soffset);
//   * tap offsets live in the lanes of one VGPR (v_readlane with the uniform tap index) instead of a kernarg load per K-step, whose
//     s_waitcnt lgkmcnt(0) used to drain the fragment reads too;
//   * both k16 sub-steps' fragments are requested before the first MFMA and the loads of tile kk+2 are issued between
//     them, so LDS latency is hidden behind the first MFMAs instead of being exposed four times per K-step.
#include "am_common.h"
#include <cstdlib>

namespace amr {

// Diagnostic: shader-clock cycles and 100 MHz wall-clock ticks that workgroup 0 of the last 256x256 launch spent in its
// K-loop, and its K-step count (am_diag_ring_clock; bench.py reports the clock the chip held under this kernel).
__device__ long long g_ring_clk[3];

constexpr int RING_MAX_TAPS = 9;
constexpr unsigned OOB = 0x80000000u;  // offsets at or above every buffer's num_records

struct RingParams {
  am_conv_geom g;
  const void* x;
  const void* w;
  void* y;
  const float* bias;
  const void* res;  // optional residual with y's geometry: y = act(conv + bias + res) (see conv_ring16.hip)
  double* stats;
  int M, nk, kpt, Ktot, relu, mtiles, ntiles;
  unsigned x_bytes, w_bytes;
  unsigned hw_mul, hw_sh, mw_mul, mw_sh;  // n / (MH*MW) and n / MW as mulhi + shift (am_fastdiv: exact for n < 2^31)
  int tap_off[RING_MAX_TAPS];  // byte offset of tap t relative to the row's base pixel
  int tap_yx[RING_MAX_TAPS];   // (dy << 16) | (dx & 0xffff)
};

typedef __attribute__((address_space(3))) void* lds_ptr;

// 16 bytes per lane from a raw buffer (base, num_records = bytes) straight into LDS at dst + lane*16; offsets at or past
// `bytes` deliver zeros.  Kept in a __device__ helper: the buffer-resource type does not exist in the host pass.
__device__ __forceinline__ void buffer_to_lds16(const void* base, unsigned bytes, char* dst, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000),
                                           (lds_ptr)dst, 16, voff, soff, 0, 0);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void conv_ring_k(const RingParams p) {
  constexpr int BKB = 64;                  // K-step in bytes (32 halves = two k16 MFMA sub-steps)
  constexpr int NW = WM * WN, NTH = NW * 64, NSTG = 3;
  typedef half_t T;
  constexpr int RPI = 1024 / BKB;          // rows per wave-instruction (16)
  constexpr int AI = BM / RPI / NW;        // A instructions per wave per K-step
  constexpr int BI = BN / RPI / NW;
  constexpr int NLOAD = AI + BI;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int STAGE = (BM + BN) * BKB;
  static_assert((NW == 4 || NW == 8) && TM >= 1 && TN >= 1 && AI >= 1 && BI >= 1 && (BN / RPI) % NW == 0 && (BM / RPI) % NW == 0, "tile");

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  int* opix_s = reinterpret_cast<int*>(smem + 4096);  // filled after the K-loop, inside the then-free stage buffers

  const am_conv_geom& g = p.g;
  T* __restrict__ y = static_cast<T*>(p.y);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int lb = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
  const int mt = lb / p.ntiles, nt = lb - mt * p.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-lane loader state: instruction j of this wave covers tile rows (wid*AI + j)*16 + lane/4 ----
  // (prologue and epilogue run once per tile on all waves: divisions are mulhi + shift with host-made reciprocals, the padding
  // test of a tap sits in tap_offsets(), once per tap, instead of a ntaps x AI loop up front -- see conv_ring16.hip)
  const int lrow = lane >> 2, cpos = lane & 3;
  unsigned a_off[AI], b_off[BI];
  int a_iy[AI], a_ix[AI];
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    const int r = (wid * AI + j) * RPI + lrow;
    const int c = cpos ^ ((r >> 2) & 3);  // source chunk for this LDS position
    const unsigned m = (unsigned)(m0 + r);
    const unsigned img = am_fastdiv(m, p.hw_mul, p.hw_sh);
    const unsigned rem = m - img * (unsigned)(g.MH * g.MW);
    const unsigned my = am_fastdiv(rem, p.mw_mul, p.mw_sh), mx = rem - my * (unsigned)g.MW;
    const int iy0 = (int)my * g.iys, ix0 = (int)mx * g.ixs;
    a_off[j] = (unsigned)((((img * g.IH + iy0) * g.IW + ix0) * g.ldi + g.x_coff) * 2 + c * 16);
    a_iy[j] = (int)m < p.M ? iy0 : -(1 << 20);  // rows past M: far outside every image
    a_ix[j] = ix0;
  }
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int r = (wid * BI + j) * RPI + lrow;
    const int c = cpos ^ ((r >> 2) & 3);
    b_off[j] = (unsigned)((n0 + r) * p.Ktot * 2 + c * 16);  // rows past the packed matrix are out of range: zeros
  }
  int tapv = 0, tapyx = 0;  // lane t holds tap t's byte offset / (dy, dx): one v_readlane per tap instead of a kernarg load
#pragma unroll
  for (int t = 0; t < RING_MAX_TAPS; ++t) {
    tapv = (lane == t) ? p.tap_off[t] : tapv;
    tapyx = (lane == t) ? p.tap_yx[t] : tapyx;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // tile `kk` = (tap, kin): A rows gather [tap offset + kin*64 B) of each pixel's run, B rows K bytes [kk*64, +64)
  // per-lane A offsets depend on the tap only (the K-step inside the tap goes into the scalar offset): recomputed when the
  // tap changes, once per kpt K-steps, so a K-step's loader block is scalar arithmetic plus the loads
  unsigned a_vo[AI];
  auto tap_offsets = [&](int tap) {
    const int toff = __builtin_amdgcn_readlane(tapv, tap);
    const int yx = __builtin_amdgcn_readlane(tapyx, tap);
    const int tdy = yx >> 16, tdx = (int)(short)(yx & 0xffff);
#pragma unroll
    for (int j = 0; j < AI; ++j) {
      const bool ok = (unsigned)(a_iy[j] + tdy) < (unsigned)g.IH && (unsigned)(a_ix[j] + tdx) < (unsigned)g.IW;
      a_vo[j] = ok ? a_off[j] + (unsigned)toff : OOB;
    }
  };
  auto issue_tile = [&](int kk, int tap, int kin, int stage) {
    char* As = smem + stage * STAGE + wid * (AI * 1024);
    char* Bs = smem + stage * STAGE + BM * BKB + wid * (BI * 1024);
#pragma unroll
    for (int j = 0; j < AI; ++j) buffer_to_lds16(p.x, p.x_bytes, As + j * 1024, a_vo[j], kin * BKB);
#pragma unroll
    for (int j = 0; j < BI; ++j)
      buffer_to_lds16(p.w, p.w_bytes, Bs + j * 1024, b_off[j], kk * BKB);
  };

  // issue cursor (tile index, tap, K-step inside the tap), advanced without divisions.  Tiles past the last one are
  // still issued (their taps have no validity bit: zeros into a stage nobody reads), so every K-step is the same
  // straight-line code with the same vmcnt bookkeeping.
  const int kpt = __builtin_amdgcn_readfirstlane(p.kpt), nk = __builtin_amdgcn_readfirstlane(p.nk);
  int ikk = 0, itap = 0, ikin = 0;
  auto advance = [&]() {
    ++ikk;
    const bool wrap = ikin + 1 == kpt;
    ikin = wrap ? 0 : ikin + 1;
    itap = wrap ? itap + 1 : itap;
  };
  tap_offsets(0);
  issue_tile(ikk, itap, ikin, 0); advance();
  if (ikin == 0) tap_offsets(itap);
  issue_tile(ikk, itap, ikin, 1); advance();

  // fragment read addressing: lane reads row (lane&31) of its 32-row sub-tile, 16-byte chunk (2*ks + lane>>5) ^ swz(row);
  // swz(row) is the same for the TM (TN) sub-tiles of a lane because they are 32 rows apart
  const int frow_a = wm * TM * 32 + (lane & 31);
  const int frow_b = wn * TN * 32 + (lane & 31);
  int fa[2], fb[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int cidx = ks * 2 + (lane >> 5);
    fa[ks] = frow_a * BKB + ((cidx ^ ((frow_a >> 2) & 3)) << 4);
    fb[ks] = BM * BKB + frow_b * BKB + ((cidx ^ ((frow_b >> 2) & 3)) << 4);
  }

  // Software pipeline, rotated by half a K-step so that no fragment read is waited for right after it is issued:
  //   top:    read k16 sub-step 1 of tile kk, issue the loads of tile kk+2 (into the stage of tile kk-1, which every wave
  //           finished reading before the previous barrier), 8 MFMAs of sub-step 0 (fragments read last iteration)
  //   middle: tile kk+1 has landed (at most tile kk+2's loads outstanding) -> barrier -> read its sub-step 0
  //   bottom: 8 MFMAs of sub-step 1
  half8_t a0[TM], b0[TN], a1[TM], b1[TN];
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int t = 0; t < TM; ++t) a0[t] = *reinterpret_cast<const half8_t*>(smem + fa[0] + t * 32 * BKB);
#pragma unroll
  for (int t = 0; t < TN; ++t) b0[t] = *reinterpret_cast<const half8_t*>(smem + fb[0] + t * 32 * BKB);

  const bool diag = BN >= 256 && blockIdx.x == 0 && tid == 0;
  const long long c0 = diag ? clock64() : 0, w0 = diag ? wall_clock64() : 0;
  int stage = 0;
  for (int kk = 0; kk < nk; ++kk) {
    const char* S = smem + stage * STAGE;
    const int nstage = stage == NSTG - 1 ? 0 : stage + 1;
    // the previous half's fragment reads returned long ago (8 MFMAs back): waiting here, before the branch, keeps hipcc from
    // putting an lgkmcnt(0) in front of this half's MFMAs once the reads below are in flight
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (ikin == 0) tap_offsets(itap);  // the tile requested below starts a new tap: its per-lane offsets
#pragma unroll
    for (int t = 0; t < TM; ++t) a1[t] = *reinterpret_cast<const half8_t*>(S + fa[1] + t * 32 * BKB);
#pragma unroll
    for (int t = 0; t < TN; ++t) b1[t] = *reinterpret_cast<const half8_t*>(S + fb[1] + t * 32 * BKB);
    issue_tile(ikk, itap, ikin, stage == 0 ? 2 : stage - 1);
    advance();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[tm], b0[tn], acc[tm][tn], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // own part of tile kk+1 landed, own reads of tile kk returned (so the next iteration may overwrite ... tile kk-1's
    // stage is the one written above; tile kk's stage is written after the NEXT barrier)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NLOAD) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* Sn = smem + nstage * STAGE;
#pragma unroll
    for (int t = 0; t < TM; ++t) a0[t] = *reinterpret_cast<const half8_t*>(Sn + fa[0] + t * 32 * BKB);
#pragma unroll
    for (int t = 0; t < TN; ++t) b0[t] = *reinterpret_cast<const half8_t*>(Sn + fb[0] + t * 32 * BKB);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[tm], b1[tn], acc[tm][tn], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    stage = nstage;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two tiles issued past the end
  __syncthreads();  // all fragment reads done before the epilogue reuses the stage buffers
  if (diag) {
    g_ring_clk[0] = clock64() - c0;
    g_ring_clk[1] = wall_clock64() - w0;
    g_ring_clk[2] = nk;
  }

  // ---- epilogue: BN statistics from the accumulators, bias, ReLU, LDS-staged 16-byte stores ----
  for (int r = tid; r < BM; r += NTH) {  // element offset of every tile row's output pixel (the launcher checks it fits 31 bits)
    const unsigned m = (unsigned)(m0 + r);
    int off = -1;
    if ((int)m < p.M) {
      const unsigned img = am_fastdiv(m, p.hw_mul, p.hw_sh);
      const unsigned rem = m - img * (unsigned)(g.MH * g.MW);
      const unsigned my = am_fastdiv(rem, p.mw_mul, p.mw_sh), mx = rem - my * (unsigned)g.MW;
      off = (int)(((img * g.OH + my * g.oys + g.oy0) * g.OW + mx * g.oxs + g.ox0) * g.ldo + g.y_coff);
    }
    opix_s[r] = off;
  }

  if (p.stats != nullptr) {
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      // column (channel) sums over this lane's TM*16 rows: whole-vector adds / FMAs (v_pk_add_f32, v_pk_fma_f32 on register
      // pairs), then a horizontal add.  Rows past M were fetched as zeros, so they add nothing.
      f32x16 sv = acc[0][tn], qv = acc[0][tn] * acc[0][tn];
#pragma unroll
      for (int tm = 1; tm < TM; ++tm) {
        sv += acc[tm][tn];
        qv = __builtin_elementwise_fma(acc[tm][tn], acc[tm][tn], qv);
      }
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s += sv[r];
        q += qv[r];
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
  // (LDS reads done; the fp64 atomics stay in flight: __syncthreads() would wait for their round trip)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  {
    constexpr int SP = TN * 64 + 16;
    char* stg = smem + 8192 + wid * (TM * 32) * SP;  // past the stats scratch and the output-pixel table
    const T* __restrict__ res = static_cast<const T*>(p.res);
    const bool relu_early = p.relu && res == nullptr;
    if (p.bias == nullptr && !relu_early) {  // BN layers (almost every launch): convert and stage, nothing else
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            *reinterpret_cast<half_t*>(stg + row * SP + (tn * 32 + (lane & 31)) * 2) = (half_t)acc[tm][tn][r];
          }
    } else {
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
            if (relu_early) v = fmaxf(v, 0.f);
            *reinterpret_cast<half_t*>(stg + row * SP + (tn * 32 + (lane & 31)) * 2) = (half_t)v;
          }
      }
    }
    // the wave reads back what its own lanes wrote: LDS executes a wave's accesses in order, so draining the
    // writes is enough; the asm also stops the compiler from moving the (differently typed) reads above the writes
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    constexpr int CPRW = TN * 4;
    const int ncols = (g.N + 7) & ~7;
    constexpr int NIT = TM * TN * 2;
    // all LDS reads first (the accumulators are dead), then the stores back to back; 64 % CPRW == 0: a lane keeps its column chunk
    int offv[NIT];
    uint4 dat[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * 64 + lane;
      const int row = q / CPRW, cc = q - row * CPRW;
      offv[it] = opix_s[wm * TM * 32 + row];
      dat[it] = *reinterpret_cast<const uint4*>(stg + row * SP + cc * 16);
    }
    const int col0 = n0 + wn * TN * 32 + (lane % CPRW) * 8;
    // split rows (fused stride-2 dgrad): the second half of the columns continues one image row further down
    const int seg = (g.osplit > 0 && col0 >= g.osplit) ? g.osplit_stride - g.osplit : 0;
    const bool col_ok = col0 < ncols;
    if (res != nullptr) {  // (never with split rows: the launcher checks)
      uint4 rv[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) rv[it] = (offv[it] >= 0 && col_ok) ? *reinterpret_cast<const uint4*>(res + (unsigned)(offv[it] + col0)) : uint4{0, 0, 0, 0};
      const bool act = p.relu != 0;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        dat[it].x = am_addh2_act(dat[it].x, rv[it].x, act);
        dat[it].y = am_addh2_act(dat[it].y, rv[it].y, act);
        dat[it].z = am_addh2_act(dat[it].z, rv[it].z, act);
        dat[it].w = am_addh2_act(dat[it].w, rv[it].w, act);
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (offv[it] >= 0 && col_ok) *reinterpret_cast<uint4*>(y + (unsigned)(offv[it] + col0 + seg)) = dat[it];
  }
}

template <int BM, int BN, int WM, int WN>
int launch_ring(const RingParams& p0, hipStream_t s) {
  constexpr int STAGE = (BM + BN) * 64;
  RingParams p = p0;
  p.mtiles = am_cdiv(p.M, BM);
  p.ntiles = am_cdiv(p.g.N, BN);
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr size_t EPI = 8192 + (size_t)WM * WN * (TM * 32) * (TN * 64 + 16);
  const size_t lds = 3 * STAGE > EPI ? 3 * STAGE : EPI;
  static bool attr_done_dev[AM_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[am_current_device()];
  if (lds > 64 * 1024 && !attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ring_k<BM, BN, WM, WN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return AM_ERR_LAUNCH;
    attr_done = true;
  }
  g_am_conv_variant = BN >= 256 ? AM_CV_RING_256x256 : AM_CV_RING_256x128;
  hipLaunchKernelGGL((conv_ring_k<BM, BN, WM, WN>), dim3(p.mtiles * p.ntiles), dim3(WM * WN * 64), lds, s, p);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

}  // namespace amr

// Called by am_conv_gemm2_f16 (conv_gemm2.hip); returns AM_ERR_UNSUPPORTED when the shape is not covered.
// `npad_rows` = rows of the packed weight matrix (am_conv_npad).
int am_conv_ring16_f16(const am_conv_geom* g, const void* x, const void* w, const float* bias, int relu, const void* res, void* y,
                       double* stats, int variant, int* tile_out, hipStream_t s);  // conv_ring16.hip

// res (may be null): residual tensor with y's geometry, y = act(conv + bias + res); not with split output rows.
int am_conv_ring_f16(const am_conv_geom* g, const void* x, const void* w, const float* bias, int relu, const void* res, void* y,
                     double* stats, hipStream_t s) {
  using namespace amr;
  if (res != nullptr && g->osplit > 0) return AM_ERR_UNSUPPORTED;
  if (const int t = am_tuning(AM_TUNE_RING); t > 0) {  // 16x16x32 generation (conv_ring16.hip): the 256x256 tile
    int tile = 0;
    const int rc = am_conv_ring16_f16(g, x, w, bias, relu, res, y, stats, t - 1, &tile, s);
    if (rc == AM_OK) g_am_conv_variant = tile == 1 ? AM_CV_RING16_256x256 : tile == 3 ? AM_CV_RING16_128x256 : AM_CV_RING16_256x128;
    if (rc != AM_ERR_UNSUPPORTED) return rc;
  }
  if (g->ntaps <= 0 || g->ntaps > RING_MAX_TAPS || g->pix_shift != 31 || (g->krun * 2) % 64 != 0 || g->N <= 64) return AM_ERR_UNSUPPORTED;
  const long long x_bytes = (long long)g->B * g->IH * g->IW * g->ldi * 2;
  const long long Ktot = (long long)g->ntaps * g->krun;
  const long long w_bytes = (long long)am_conv_npad(g->N) * Ktot * 2;
  if (x_bytes >= (1ll << 31) || w_bytes >= (1ll << 31)) return AM_ERR_UNSUPPORTED;
  RingParams p;
  p.g = *g;
  p.x = x; p.w = w; p.y = y; p.bias = bias; p.res = res; p.stats = stats;
  p.M = g->B * g->MH * g->MW;
  p.Ktot = (int)Ktot;
  p.relu = relu;
  p.kpt = g->krun * 2 / 64;
  p.nk = g->ntaps * p.kpt;
  p.mtiles = p.ntiles = 0;
  p.x_bytes = (unsigned)x_bytes;
  p.w_bytes = (unsigned)w_bytes;
  for (int t = 0; t < RING_MAX_TAPS; ++t) {
    p.tap_off[t] = t < g->ntaps ? (int)(((long long)g->dy[t] * g->IW + g->dx[t]) * (long long)g->ldi * 2) : 0;
    p.tap_yx[t] = t < g->ntaps ? (int)(((unsigned)(unsigned short)g->dy[t] << 16) | (unsigned short)g->dx[t]) : (int)0x80008000u;
  }
  if (((long long)g->B * g->OH * g->OW + 1) * g->ldo + g->osplit_stride + g->y_coff >= (1ll << 31)) return AM_ERR_UNSUPPORTED;  // 31-bit output offsets
  am_fastdiv_make((unsigned)(g->MH * g->MW), &p.hw_mul, &p.hw_sh);
  am_fastdiv_make((unsigned)g->MW, &p.mw_mul, &p.mw_sh);
  const long long mt256 = (p.M + 255) / 256;
  // (a contraction of a few K-steps -- the 1x1 / stride-2 shortcut convolutions -- is all prologue and epilogue: two 256x128
  // workgroups per CU overlap one's epilogue with the other's loads, one 256x256 workgroup cannot)
  if (g->N >= 256 && mt256 * ((g->N + 255) / 256) >= 200 && p.nk > am_tuning(AM_TUNE_RING_SHORT_K)) return launch_ring<256, 256, 2, 4>(p, s);
  // 256x128 tiles run two workgroups per CU (72 KiB of LDS each); below one workgroup per CU a lone workgroup still has its CU's
  // matrix pipes to itself, so the ring kernel keeps beating the two-stage kernels down to AM_TUNE_RING128_MIN_TILES tiles (100: layer4
  // at B = 8 -- 116 tiles -- runs cfg3 2 % faster on it than on conv_gemm2_k)
  if (mt256 * ((g->N + 127) / 128) >= am_tuning(AM_TUNE_RING128_MIN_TILES)) return launch_ring<256, 128, 4, 2>(p, s);
  return AM_ERR_UNSUPPORTED;
}

// out[0] = shader cycles, out[1] = 100 MHz ticks, out[2] = K-steps of workgroup 0's K-loop in the last 256x256 ring launch
// (synchronises the stream).  MFMA floor of that loop: 1024 cycles per K-step.
int am_diag_ring16_clock(long long* out);  // conv_ring16.hip

extern "C" int am_diag_ring_clock(long long* out, void* stream) {
  if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return AM_ERR_LAUNCH;
  if (am_tuning(AM_TUNE_RING) > 0) return am_diag_ring16_clock(out);
  out[3] = out[4] = 0;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(amr::g_ring_clk), 3 * sizeof(long long)) != hipSuccess) return AM_ERR_LAUNCH;
  return AM_OK;
}
