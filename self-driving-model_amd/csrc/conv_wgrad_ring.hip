// Weight gradient of the gather-GEMM as an LDS-DMA ring kernel (f16):  dW[n][k] += scale * sum_m dY[m][n] * gather(m, k).
//
// The contraction runs over PIXELS, so this is conv_ring_k with the roles turned: the accumulator tile is a block of the
// weight matrix (BNT output channels x 256 packed k-columns, 8 waves of (BNT/2) x 64), a "K-step" is 32 pixels, and both
// operand tiles are pixel-major in LDS -- [32 pixels][BNT channels of dY] and [32 pixels][8 slabs of 64 B of the gathered
// input] -- so MFMA fragments (8 consecutive pixels of one channel per lane) come from ds_read_b64_tr_b16 transposing
// reads.  What the register-staged kernel (conv_wgrad_k, conv_gemm.hip) lacks is all here: buffer_load ... lds fills with
// hardware zero padding (taps outside the image, rows past the last pixel), a 3-stage ring with one barrier per step, a
// 256 x 256 tile (128 FLOP per LDS-DMA byte instead of 64) and four times fewer fp32 atomics per weight.
//
// LDS image of one operand stage: a 1 KiB DMA piece holds 1024 / rowbytes pixel rows back to back; piece j sits at
// j * 1088 B.  Pixel p lives in piece p % NI at row p / NI, so the four pixel rows a 16-lane group of a transposing read
// touches (p .. p+3) are in four consecutive pieces, 64 B apart modulo 256 B: conflict free.
//
// Pixels of a workgroup's chunk are consecutive, so their image coordinates advance by 32 per step with one compare-and-
// wrap (needs an output row of at least 32 pixels); dY needs no coordinates at all (output pixel index = GEMM row).
#include "am_common.h"
#include <cstdlib>

namespace awr {

constexpr unsigned OOB = 0x80000000u;
constexpr int PS = 32;           // pixels per step
constexpr int PIECE = 1024 + 64;  // LDS pitch of a DMA piece
constexpr int XROW = 512;         // 8 slabs of 64 B
constexpr int NI_X = 16;          // DMA pieces per X stage (2 pixel rows each)

struct WrParams {
  am_conv_geom g;
  const void* x;
  const void* dy;
  float* dw;
  float* ws;             // slab mode: chunk c stores its unscaled partial tile at ws + c * ws_stride (plain stores, no atomics)
  long long ws_stride;   // elements between two chunks' slabs (>= rows * Ktot)
  float scale;
  int M, Ktot, kpt, nslab, ktiles, ntiles, mchunks, mc;
  unsigned x_bytes, dy_bytes;
  int tap_off[AM_MAX_TAPS];  // byte offset of tap t relative to a pixel's base input pixel
};

typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ void buffer_to_lds16(const void* base, unsigned bytes, char* dst, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000),
                                           (lds_ptr)dst, 16, voff, soff, 0, 0);
}

__device__ __forceinline__ half8_t read_tr(const char* lo_addr, const char* hi_addr) {
  s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(lo_addr));
  s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(hi_addr));
  half4_t l4 = __builtin_bit_cast(half4_t, lo), h4 = __builtin_bit_cast(half4_t, hi);
  return half8_t{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
}

// M16 (TM = 4 only): v_mfma_f32_16x16x32_f16 -- one MFMA spans the 32 pixels of a step (less power per FLOP on this power-limited part:
// DESIGN.md); a fragment is 16 channels x 32 pixels, lane group gq = pixel octet.  Pieces 8-15 of a stage sit 32 bytes further: the octets
// gq and gq + 1 of one 32-lane service group of the transposing read are in pieces j and j + 8, which would otherwise share their banks.
template <int TM, bool M16>
__global__ __launch_bounds__(512) void wgrad_ring_k(const WrParams p) {
  static_assert(!M16 || TM == 4, "M16: 16 pieces of two pixel rows per operand stage");
  constexpr int WM = 2, WN = 4, TN = 2, NW = 8, NSTG = 3;
  constexpr int BNT = WM * TM * 32;          // output channels per tile
  constexpr int AROW = BNT * 2;              // bytes of a dY pixel row in the tile
  constexpr int RPI_A = 1024 / AROW;         // dY pixel rows per DMA piece (2 or 4)
  constexpr int NI_A = PS / RPI_A;           // dY pieces per stage (16 or 8)
  constexpr int LPR_A = AROW / 16;           // lanes per dY row
  constexpr int AIW = NI_A / NW;             // dY pieces per wave per step (2 or 1)
  constexpr int XIW = NI_X / NW;             // X pieces per wave per step (2)
  constexpr int NLOAD = AIW + XIW;
  constexpr int SKEW = M16 ? 32 : 0;
  constexpr int A_BYTES = NI_A * PIECE + SKEW, X_BYTES = NI_X * PIECE + SKEW;
  constexpr int STAGE = A_BYTES + X_BYTES;

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const am_conv_geom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int tiles = p.ktiles * p.ntiles;
  const int mcid = blockIdx.x / tiles;
  const int tile = blockIdx.x - mcid * tiles;
  const int nt = tile / p.ktiles, kt = tile - nt * p.ktiles;
  const int n0 = nt * BNT;
  const int mbeg = mcid * p.mc;
  const int nsteps = __builtin_amdgcn_readfirstlane((min(p.M, mbeg + p.mc) - mbeg + PS - 1) / PS);

  // ---- dY pieces of this wave: lane -> (row slot, 16-byte chunk); pixel of the slot inside a step = piece + slot * NI_A ----
  unsigned a_off[AIW];
#pragma unroll
  for (int i = 0; i < AIW; ++i) {
    const int j = wid * AIW + i;
    const int slot = lane / LPR_A, chunk = lane - slot * LPR_A;
    const int pix = j + slot * NI_A;
    a_off[i] = (unsigned)(((long long)(mbeg + pix) * g.ldo + g.y_coff + n0) * 2 + chunk * 16);  // rows past M: beyond dy_bytes
  }
  // ---- X pieces: lane -> (row slot, slab, quarter); per lane one tap; pixel coordinates advance by 32 per step ----
  unsigned x_const[XIW];
  int x_img[XIW], x_my[XIW], x_mx[XIW];
  const int chunk = lane & 31, slab = chunk >> 2, quarter = chunk & 3;
  const int gs = kt * 8 + slab;
  const bool slab_ok = gs < p.nslab;
  const int tap = slab_ok ? gs / p.kpt : 0;
  const int within = gs - tap * p.kpt;
  const int tdy = g.dy[tap], tdx = g.dx[tap];
  const unsigned lane_xoff = (unsigned)(p.tap_off[tap] + within * 64 + quarter * 16 + g.x_coff * 2);
  const int hw = g.MH * g.MW;
#pragma unroll
  for (int i = 0; i < XIW; ++i) {
    const int j = wid * XIW + i;
    const int pix = j + (lane >> 5) * NI_X;
    const int m = mbeg + pix;
    const int img = m / hw;
    const int rem = m - img * hw;
    x_img[i] = img;
    x_my[i] = rem / g.MW;
    x_mx[i] = rem - x_my[i] * g.MW;
    x_const[i] = lane_xoff;
  }

  f32x16 acc[M16 ? 1 : TM][M16 ? 1 : TN];
  f32x4 acc4[M16 ? 2 * TM : 1][M16 ? 2 * TN : 1];  // M16: [16-channel block of dY][16-column block of the gradient]
  if constexpr (M16) {
#pragma unroll
    for (int a = 0; a < 2 * TM; ++a)
#pragma unroll
      for (int b = 0; b < 2 * TN; ++b) acc4[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  }

  int istep = 0;  // next step to request
  auto issue_step = [&](int stage) {
    char* As = smem + stage * STAGE;
    char* Xs = As + A_BYTES;
    const unsigned soff = (unsigned)(istep * PS * g.ldo * 2);
#pragma unroll
    for (int i = 0; i < AIW; ++i) buffer_to_lds16(p.dy, p.dy_bytes, As + (wid * AIW + i) * PIECE + ((wid * AIW + i) >= 8 ? SKEW : 0), a_off[i], soff);
#pragma unroll
    for (int i = 0; i < XIW; ++i) {
      const int iy = x_my[i] * g.iys + tdy, ix = x_mx[i] * g.ixs + tdx;
      const bool ok = slab_ok && x_img[i] < g.B && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW;
      const unsigned voff = (unsigned)(((x_img[i] * g.IH + iy - tdy) * g.IW + ix - tdx) * g.ldi * 2) + x_const[i];
      buffer_to_lds16(p.x, p.x_bytes, Xs + (wid * XIW + i) * PIECE + ((wid * XIW + i) >= 8 ? SKEW : 0), ok ? voff : OOB, 0);
      // next step: 32 pixels on (an output row holds at least 32: one wrap at most)
      x_mx[i] += PS;
      if (x_mx[i] >= g.MW) {
        x_mx[i] -= g.MW;
        x_my[i] += 1;
        if (x_my[i] >= g.MH) { x_my[i] = 0; x_img[i] += 1; }
      }
    }
    ++istep;
  };
#pragma unroll
  for (int st = 0; st < NSTG - 1; ++st) issue_step(st);

  if constexpr (M16) {
    // lane -> (pixel octet gq, pixel q (+4 for the second read), channels 4*pp..+3 of a 16-channel block); pixel 8*gq + q + 4*h is
    // row gq >> 1 of piece 8*(gq & 1) + q + 4*h
    const int gq = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
    int fa[2], fb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int piece = 8 * (gq & 1) + q + 4 * h, row = gq >> 1;
      const int base = piece * PIECE + ((gq & 1) ? SKEW : 0);
      fa[h] = base + row * AROW + (wm * TM * 32 + 4 * pp) * 2;
      fb[h] = A_BYTES + base + row * XROW + (wn * TN * 32 + 4 * pp) * 2;
    }
    // per step: dY blocks 0..3 (aL) x the 4 column blocks, then blocks 4..7 (aH); the fragments of the second half are requested
    // before the MFMAs of the first, those of the next step (aL and the OTHER set of column fragments) before the MFMAs of the second
    half8_t aL[TM], aH[TM], bA[2 * TN], bB[2 * TN];
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < TM; ++t) aL[t] = read_tr(smem + fa[0] + t * 32, smem + fa[1] + t * 32);
#pragma unroll
    for (int t = 0; t < 2 * TN; ++t) bA[t] = read_tr(smem + fb[0] + t * 32, smem + fb[1] + t * 32);
    int stage = 0;
    auto step = [&](half8_t (&bc)[2 * TN], half8_t (&bn)[2 * TN]) {
      const char* S = smem + stage * STAGE;
      const int nstage = stage == NSTG - 1 ? 0 : stage + 1;
#pragma unroll
      for (int t = 0; t < TM; ++t) aH[t] = read_tr(S + fa[0] + (TM + t) * 32, S + fa[1] + (TM + t) * 32);
      issue_step(stage == 0 ? NSTG - 1 : stage - 1);  // steps past the chunk: rows the MFMAs below never read again
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ta = 0; ta < TM; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2 * TN; ++tb) acc4[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aL[ta], bc[tb], acc4[ta][tb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NLOAD) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* Sn = smem + nstage * STAGE;
#pragma unroll
      for (int t = 0; t < TM; ++t) aL[t] = read_tr(Sn + fa[0] + t * 32, Sn + fa[1] + t * 32);
#pragma unroll
      for (int t = 0; t < 2 * TN; ++t) bn[t] = read_tr(Sn + fb[0] + t * 32, Sn + fb[1] + t * 32);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ta = 0; ta < TM; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2 * TN; ++tb) acc4[TM + ta][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aH[ta], bc[tb], acc4[TM + ta][tb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      stage = nstage;
    };
    for (int st = 0; st < nsteps; st += 2) {
      step(bA, bB);
      if (st + 1 < nsteps) step(bB, bA);
    }
  } else {
  // fragment addressing (see conv_wgrad_k): a 16-lane group reads pixel rows prow + q, q = 0..3 (lo) and + 4 (hi), 8 bytes
    // (4 channels) per lane; after the transpose a lane holds 8 consecutive pixels of channel / k-column (lane & 31)
    const int gq = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
    int fa[2][2], fb[2][2];  // [sub-step][lo / hi]
  #pragma unroll
    for (int ks = 0; ks < 2; ++ks)
  #pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int prow = ks * 16 + 8 * (gq >> 1) + q + 4 * h;
        fa[ks][h] = (prow % NI_A) * PIECE + (prow / NI_A) * AROW + ((wm * TM * 32) + (gq & 1) * 16 + 4 * pp) * 2;
        fb[ks][h] = A_BYTES + (prow % NI_X) * PIECE + (prow / NI_X) * XROW + ((wn * TN * 32) + (gq & 1) * 16 + 4 * pp) * 2;
      }
  
    half8_t a0[TM], b0[TN], a1[TM], b1[TN];
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  #pragma unroll
    for (int t = 0; t < TM; ++t) a0[t] = read_tr(smem + fa[0][0] + t * 64, smem + fa[0][1] + t * 64);
  #pragma unroll
    for (int t = 0; t < TN; ++t) b0[t] = read_tr(smem + fb[0][0] + t * 64, smem + fb[0][1] + t * 64);
  
    int stage = 0;
    for (int st = 0; st < nsteps; ++st) {
      const char* S = smem + stage * STAGE;
      const int nstage = stage == NSTG - 1 ? 0 : stage + 1;
  #pragma unroll
      for (int t = 0; t < TM; ++t) a1[t] = read_tr(S + fa[1][0] + t * 64, S + fa[1][1] + t * 64);
  #pragma unroll
      for (int t = 0; t < TN; ++t) b1[t] = read_tr(S + fb[1][0] + t * 64, S + fb[1][1] + t * 64);
      issue_step(stage == 0 ? NSTG - 1 : stage - 1);  // steps past the chunk: rows the MFMAs below never read again
      __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
      for (int tm = 0; tm < TM; ++tm)
  #pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[tm], b0[tn], acc[tm][tn], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NLOAD) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* Sn = smem + nstage * STAGE;
  #pragma unroll
      for (int t = 0; t < TM; ++t) a0[t] = read_tr(Sn + fa[0][0] + t * 64, Sn + fa[0][1] + t * 64);
  #pragma unroll
      for (int t = 0; t < TN; ++t) b0[t] = read_tr(Sn + fb[0][0] + t * 64, Sn + fb[0][1] + t * 64);
      __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
      for (int tm = 0; tm < TM; ++tm)
  #pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[tm], b1[tn], acc[tm][tn], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      stage = nstage;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- flush; a lane owns k-column (lane & 31) of its 32-block: 128 contiguous bytes per register and row.  Slab mode: plain
  // stores of the unscaled partial tile into this pixel chunk's slab (am_conv_wgrad_ws sums the slabs in a second pass: no
  // device-scope atomics, which run at ~1.3 TB/s of added bytes chip-wide and cost this kernel ~30 %); else fp32 atomics ----
  float* part = p.ws ? p.ws + (size_t)mcid * p.ws_stride : nullptr;
  if constexpr (M16) {
#pragma unroll
    for (int tb = 0; tb < 2 * TN; ++tb) {
      const int kc = kt * 256 + wn * TN * 32 + tb * 16 + (lane & 15);
      if (kc >= p.Ktot) continue;
#pragma unroll
      for (int ta = 0; ta < 2 * TM; ++ta)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = n0 + wm * TM * 32 + ta * 16 + 4 * (lane >> 4) + j;
          if (n < g.N) {
            if (part) part[(size_t)n * p.Ktot + kc] = acc4[ta][tb][j];
            else atomicAdd(p.dw + (size_t)n * p.Ktot + kc, acc4[ta][tb][j] * p.scale);
          }
        }
    }
  } else {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int kc = kt * 256 + (wn * TN + tn) * 32 + (lane & 31);
      if (kc >= p.Ktot) continue;
  #pragma unroll
      for (int tm = 0; tm < TM; ++tm)
  #pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (n < g.N) {
            if (part) part[(size_t)n * p.Ktot + kc] = acc[tm][tn][r];
            else atomicAdd(p.dw + (size_t)n * p.Ktot + kc, acc[tm][tn][r] * p.scale);
          }
        }
    }
  }
}

template <int TM, bool M16>
int launch(const WrParams& p0, hipStream_t s, bool plan_only) {
  constexpr int BNT = 2 * TM * 32;
  constexpr int NI_A = PS / (1024 / (BNT * 2));
  constexpr int LDS = 3 * ((NI_A + NI_X) * PIECE + (M16 ? 64 : 0));
  WrParams p = p0;
  p.ntiles = am_cdiv(p.g.N, BNT);
  p.ktiles = am_cdiv(p.nslab, 8);
  const int tiles = p.ktiles * p.ntiles;
  // One round of workgroups (256-channel tiles own a CU; two 128-channel ones fit): every extra pixel chunk is another
  // full set of fp32 atomics on the same weights, and those -- device scope, resolved at the memory side -- cost about as
  // much as a third of the MFMA time at two rounds (layer3 shape: 216 us with 512 workgroups, 171 us with 256).
  const int target = TM == 4 ? 256 : 512;
  int mchunks = target / tiles;
  if (mchunks < 1) mchunks = 1;
  int mc = am_cdiv(p.M, mchunks);
  mc = am_cdiv(mc, PS * 8) * PS * 8;  // at least 8 steps per chunk, whole steps
  p.mc = mc;
  p.mchunks = am_cdiv(p.M, mc);
  if (plan_only) return p.mchunks;
  static bool attr_done_dev[AM_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[am_current_device()];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_ring_k<TM, M16>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return AM_ERR_LAUNCH;
    attr_done = true;
  }
  g_am_conv_variant = AM_CV_WGRAD_RING;
  hipLaunchKernelGGL((wgrad_ring_k<TM, M16>), dim3(tiles * p.mchunks), dim3(512), LDS, s, p);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

}  // namespace awr

// Called by am_conv_wgrad / am_conv_wgrad_ws (conv_gemm.hip) for f16 problems; AM_ERR_UNSUPPORTED when the shape is not covered.
// ws != nullptr: slab mode (ws_stride elements per pixel chunk).  plan_only: launches nothing and returns the number of pixel
// chunks (> 0) the launch would use.
int am_conv_wgrad_ring_f16(const am_conv_geom* g, const void* x, const void* dy, float scale, float* dw, float* ws, long long ws_stride,
                           bool plan_only, hipStream_t s) {
  using namespace awr;
  if (!am_tuning(AM_TUNE_WGRAD_RING)) return AM_ERR_UNSUPPORTED;
  if (g->ntaps <= 0 || g->pix_shift != 31 || (g->krun * 2) % 64 != 0) return AM_ERR_UNSUPPORTED;
  if (g->oys != 1 || g->oxs != 1 || g->oy0 != 0 || g->ox0 != 0 || g->MH != g->OH || g->MW != g->OW) return AM_ERR_UNSUPPORTED;  // GEMM row == output pixel
  if (g->MW < PS || (g->N % 128) != 0 || (g->y_coff * 2) % 16 != 0 || (g->ldo * 2) % 16 != 0 || (g->x_coff * 2) % 16 != 0) return AM_ERR_UNSUPPORTED;
  const long long M = (long long)g->B * g->MH * g->MW;
  if (M < 8192) return AM_ERR_UNSUPPORTED;  // short contractions: the register-staged kernel's finer M split
  const long long x_bytes = (long long)g->B * g->IH * g->IW * g->ldi * 2;
  const long long dy_bytes = M * g->ldo * 2;
  if (x_bytes >= (1ll << 31) || dy_bytes >= (1ll << 31)) return AM_ERR_UNSUPPORTED;
  WrParams p;
  p.g = *g;
  p.x = x; p.dy = dy; p.dw = dw; p.scale = scale; p.ws = ws; p.ws_stride = ws_stride;
  p.M = (int)M;
  p.Ktot = g->ntaps * g->krun;
  p.kpt = g->krun * 2 / 64;
  p.nslab = g->ntaps * p.kpt;
  p.x_bytes = (unsigned)x_bytes;
  p.dy_bytes = (unsigned)dy_bytes;
  p.ktiles = p.ntiles = p.mchunks = p.mc = 0;
  for (int t = 0; t < AM_MAX_TAPS; ++t)
    p.tap_off[t] = t < g->ntaps ? (int)(((long long)g->dy[t] * g->IW + g->dx[t]) * (long long)g->ldi * 2) : 0;
  // (16x16x32 on the 256-channel tile: 1-2 % less time per launch in isolation, 0.4 % MORE per cfg2 step -- not the default)
  if (g->N % 256 == 0) return am_tuning(AM_TUNE_WGRAD_RING) == 2 ? launch<4, true>(p, s, plan_only) : launch<4, false>(p, s, plan_only);
  return launch<2, false>(p, s, plan_only);
}
