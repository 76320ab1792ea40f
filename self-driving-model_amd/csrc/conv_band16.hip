// Row-band halo kernel for the dense 3x3 / stride-1 / pad-1 f16 convolutions with N a multiple of 256 (ResNet-18 layers 3-4 and
// the 512 -> 256 heads, forward and stride-1 input gradient): conv_ring16_k's MFMA / weight-ring / epilogue structure
// (v_mfma_f32_16x16x32_f16, transposed product, 256 x 256 tile, 8 waves) with the PIXEL operand staged the way conv_halo_k stages
// it (round 3).
//
// conv_ring16_k gathers the 256 pixel rows of a K-step once per (tap, 32-channel chunk): 16 KB of LDS-DMA per K-step next to
// 16 KB of weights -- 32 wave-instructions per K-step, and its K-step takes ~1,550 cycles against a 1,024-cycle MFMA floor with
// 2.1x the algorithmic HBM / L2 traffic (nine taps re-read every input row).  Here a workgroup owns a BAND of RB = floor(256 / W)
// whole image rows (W = 80: 3 rows = 240 pixels, W = 40: 6 rows = 240 pixels; the remaining tile rows are dead) and stages, per
// 32-channel chunk, the (RB + 2) x (W + 2)-pixel input patch ONCE, double-buffered; the nine taps of the chunk read their pixel
// fragments from it.  Zero padding and image borders are in the patch itself (out-of-image pixels are requested at an
// out-of-range buffer offset: the hardware writes zeros), so the fragment path has no masks.  Per K-step a wave issues its two
// weight pieces and, on the first PS taps of a chunk, one piece of the NEXT chunk's patch: 16 + 40 / 9 = 20.4 wave-instructions
// per K-step instead of 32, and the input is read ~1.1x instead of 9x.
//
// Whole rows, not 16 x 16 squares: the 45 x 80 and 23 x 40 maps of layers 3 / 4 would waste 7 % / 67 % of a square tiling; bands
// waste 6 % / 10 % of the MFMA rows and land on the SAME number of rounds as the exact tiling (layer 3: 480 tiles instead of 450 on
// 256 CUs = two rounds either way; layer 4: 256 instead of 230 = one round).
//
// LDS: patch pixel pitch 96 B (64 B of channels + 32 B pad: 6 sixteen-byte granules, granules 4 and 5 requested out of range):
// with the 16x16x32 operand map (lane l reads pixel l & 15, granule l >> 4) the 16 lanes of every ds_read_b128 service group
// fall on 16 different bank quads at this pitch -- worked out group by group, DESIGN.md section 3 -- for 16 consecutive patch
// pixels; a fragment that wraps from one band row to the next (W = 40: two of five) skips the two border pixels and may
// take a two-way conflict on part of its lanes (the LDS array is ~20 % busy in this kernel: not the bound).
// Weights: conv_ring16_k's three 16 KB stages (64-byte rows, XOR swizzle).
#include "am_common.h"

// Timing ablations for scratch/ablate_band16 (results are garbage with any bit set; the library is built with 0):
//   1 no LDS-DMA in the K-loop, 2 no s_barrier in the K-loop, 4 no fragment reads in the K-loop, 8 no MFMAs in the K-loop
#ifndef AMB_ABL
#define AMB_ABL 0
#endif
// AMB_SCHED > 0: the fragment reads and the LDS-DMA pieces of a half K-step are INTERLEAVED with its MFMAs (sched_group_barrier)
// instead of being issued as a block in front of them: the LDS port is this kernel's second bound (scratch/ablate_band16:
// MFMA alone 86 us, everything but the MFMAs 65 us, together 134 us on the layer3 shape), and a burst of 8 x 12 wave-reads
// behind every barrier queues up against the DMA writes.  0: block form (conv_ring16_k's round-2 schedule); 6 (default):
// first half = four (read, 2 MFMAs) groups, then the DMA pieces one per 2 MFMAs; second half = four (2 reads, 4 MFMAs) groups:
// 137 -> 112 us on the layer3 shape, 112 -> 105 us on layer4 (B = 32), measured back to back on one box.
#ifndef AMB_SCHED
#define AMB_SCHED 6
#endif

namespace amb {

constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NW = 8, NTH = NW * 64;
constexpr int TM = BM / WM / 16, TN = BN / WN / 16, HM = TM / 2;  // 8 x 4 accumulator quads per wave
constexpr int BKB = 64, PP = 96, NSTG = 3, BSTAGE = BN * BKB, BI = BN / 16 / NW;  // two weight pieces per wave and K-step
constexpr int MAX_PIECE = 48, PSMAX = MAX_PIECE / NW;                            // patch pieces (1 KiB each) / slots per wave
constexpr unsigned OOB = 0x80000000u;
static_assert(BI == 2 && TM == 8 && TN == 4, "tile");

struct Params {
  const void* x;
  const void* w;  // packed [npad(N)][9 * Cin] halves, tap-major (forward / dgrad packing)
  void* y;
  const float* bias;
  const void* res;
  double* stats;
  int B, H, W, ldi, x_coff, ldo, y_coff, Cin, N, relu;
  int RB, bands, ntn, ntiles, nchunk, npix, npiece, ps, nvalid;
  unsigned x_bytes, w_bytes;
  unsigned w_mul, w_sh, pw_mul, pw_sh;  // n / W and n / (W + 2) as mulhi + shift
};

typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ void buffer_to_lds16(const void* base, unsigned bytes, char* dst, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000),
                                           (lds_ptr)dst, 16, voff, soff, 0, 0);
}

__device__ __forceinline__ int swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }  // {0,2,3,1}: conv_ring16_k's weight image

template <int PS>
__global__ __launch_bounds__(NTH) void conv_band16_k(const Params p) {
  typedef half_t T;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  T* __restrict__ y = static_cast<T*>(p.y);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int lb = xcd_remap(blockIdx.x, p.ntiles);
  const int mt = lb / p.ntn, nt = lb - mt * p.ntn;
  const int img = mt / p.bands, band = mt - img * p.bands;
  const int y0 = band * p.RB, n0 = nt * BN;
  const int PWp = p.W + 2;
  const int patch_bytes = __builtin_amdgcn_readfirstlane(p.npiece * 1024);
  const int B_BASE = 2 * patch_bytes;

  // ---- loader state ----
  // patch: slot s of this wave is piece s * 8 + wid; a slot past the last piece fetches the wave's OWN slot-0 piece again (same
  // bytes, same place, same wave: nothing races), so every wave issues the same number of pieces per K-step, which the counted
  // vmcnt waits rely on
  unsigned pvo[PS];
  int pdst[PS];
#pragma unroll
  for (int s = 0; s < PS; ++s) {
    int piece = s * NW + wid;
    piece = piece >= p.npiece ? wid : piece;
    const unsigned gidx = (unsigned)(piece * 64 + lane);
    const unsigned pix = __umulhi(gidx, 0x2AAAAAABu);  // gidx / 6 (exact below 2^31)
    const int cc = (int)(gidx - pix * 6u);
    const unsigned prow = am_fastdiv(pix, p.pw_mul, p.pw_sh);
    const int pcol = (int)(pix - prow * (unsigned)PWp);
    const int iy = y0 - 1 + (int)prow, ix = pcol - 1;
    const bool ok = cc < 4 && (int)pix < p.npix && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    pvo[s] = ok ? (unsigned)((((img * p.H + iy) * p.W + ix) * p.ldi + p.x_coff) * 2 + cc * 16) : OOB;
    pdst[s] = piece * 1024;
  }
  // weights: piece j of this wave covers tile rows (wid * BI + j) * 16 + lane / 4 (conv_ring16_k's image)
  const int lrow = lane >> 2, cpos = lane & 3;
  unsigned b_off[BI];
  const int ktot2 = __builtin_amdgcn_readfirstlane(9 * p.Cin * 2);
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int r = (wid * BI + j) * 16 + lrow;
    b_off[j] = (unsigned)((n0 + r) * ktot2 + ((cpos ^ swz(r)) << 4));  // rows past the packed matrix are out of range: zeros
  }
  const int krun2 = __builtin_amdgcn_readfirstlane(p.Cin * 2);
  const int nchunk = __builtin_amdgcn_readfirstlane(p.nchunk);

  // K index kk = chunk * 9 + tap; weight tile kk: K bytes [tap * Cin * 2 + chunk * 64, +64)
  auto issue_b = [&](int tap, int chunk, int stage) {
#pragma unroll
    for (int j = 0; j < BI; ++j)
      buffer_to_lds16(p.w, p.w_bytes, smem + B_BASE + stage * BSTAGE + (wid * BI + j) * 1024, b_off[j], (unsigned)(tap * krun2 + chunk * BKB));
  };
  auto issue_patch = [&](int slot, int chunk, int buf) {
    // past the last chunk the pieces are still issued (out of range: zeros into the idle buffer), so every K-step keeps its count
    buffer_to_lds16(p.x, p.x_bytes, smem + buf * patch_bytes + pdst[slot], chunk < nchunk ? pvo[slot] : OOB, (unsigned)(chunk * BKB));
  };

#pragma unroll
  for (int s = 0; s < PS; ++s) issue_patch(s, 0, 0);
  issue_b(0, 0, 0);
  issue_b(1, 0, 1);

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;

  // ---- fragment addressing ----
  // pixels: lane reads tile pixel t = wm * 128 + tm * 16 + (lane & 15) = band row t / W, column t % W, i.e. patch pixel
  // (row + ky, col + kx) for tap (ky, kx) (the patch origin is image pixel (y0 - 1, -1)), granule lane >> 4.  Dead tile rows
  // (t >= nvalid) read the last valid pixel: their accumulators are never stored and are zeroed before the statistics.
  int pbase[TM];
  unsigned vmask = 0;  // bit tm: this lane's pixel of sub-tile tm is a conv output of the image
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int t = wm * (TM * 16) + tm * 16 + (lane & 15);
    const unsigned tt = (unsigned)min(t, p.nvalid - 1);
    const unsigned r = am_fastdiv(tt, p.w_mul, p.w_sh);
    const int xx = (int)(tt - r * (unsigned)p.W);
    pbase[tm] = ((int)r * PWp + xx) * PP + ((lane >> 4) << 4);
    if (t < p.nvalid && y0 + (int)r < p.H) vmask |= 1u << tm;
  }
  const int frow_w = wn * TN * 16 + (lane & 15);
  const int fw = B_BASE + frow_w * BKB + (((lane >> 4) ^ swz(frow_w)) << 4);
  const int rowoff = __builtin_amdgcn_readfirstlane(PWp * PP);

  half8_t wA[TN], wB[TN], p0[HM], p1[HM];
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BI) : "memory");  // patch 0 and weight tile 0 landed (tile 1 may be in flight)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int t = 0; t < TN; ++t) wA[t] = *reinterpret_cast<const half8_t*>(smem + fw + t * 16 * BKB);
#pragma unroll
  for (int t = 0; t < HM; ++t) p0[t] = *reinterpret_cast<const half8_t*>(smem + pbase[t]);

  // One K-step = tap T of chunk c (stage = T % 3 because 9 % 3 == 0).  conv_ring16_k's schedule: the K-step's two halves are
  // split by pixel sub-tile; the second half's pixel fragments and the next tile's weight + first-half fragments are requested
  // one half ahead, so no read is waited for right after its issue.
  auto kstep = [&](auto tapc, int c, half8_t(&wc)[TN], half8_t(&wnx)[TN]) {
    constexpr int T_ = decltype(tapc)::value;
    constexpr int ky = T_ / 3, kx = T_ % 3, stage = T_ % 3, nstage = (T_ + 1) % 3, istage = (T_ + 2) % 3;
    const char* P = smem + (c & 1) * patch_bytes;
    const int toff = ky * rowoff + kx * PP;
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the fragments read one half-step ago
    if constexpr (!(AMB_ABL & 4)) {
#pragma unroll
      for (int t = 0; t < HM; ++t) p1[t] = *reinterpret_cast<const half8_t*>(P + pbase[HM + t] + toff);
    }
    // weight tile kk+2 into the stage of tile kk-1 (every wave finished reading it before the last barrier); one piece of the
    // next chunk's patch into the other patch buffer (last read during the previous chunk)
    if constexpr (!(AMB_ABL & 1)) {
      if (T_ + 2 < 9) issue_b(T_ + 2, c, istage);
      else issue_b(T_ + 2 - 9, c + 1, istage);
      if (T_ < PS) issue_patch(T_, c + 1, (c + 1) & 1);
    }
#if !AMB_SCHED
    __builtin_amdgcn_sched_barrier(0);
#endif
    if constexpr (!(AMB_ABL & 8)) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < HM; ++tm) acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[tn], p0[tm], acc[tn][tm], 0, 0, 0);
#if AMB_SCHED == 1 || AMB_SCHED == 5
      // four groups of (one pixel-fragment read, two MFMAs, one LDS-DMA piece if any is left, two MFMAs)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read (the LDS-DMA pieces)
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
#elif AMB_SCHED == 2
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
#elif AMB_SCHED == 3 || AMB_SCHED == 6
      // reads first (one per MFMA), the DMA pieces behind the last MFMAs
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
#elif AMB_SCHED == 4
      // DMA pieces first (their landing time is the long pole), reads behind
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
#endif
    } else {
#pragma unroll
      for (int t = 0; t < TN; ++t) asm volatile("" ::"v"(wc[t]));
#pragma unroll
      for (int t = 0; t < HM; ++t) asm volatile("" ::"v"(p0[t]));
    }
    __builtin_amdgcn_sched_barrier(0);
    // everything older than this K-step's own pieces has landed: weight tile kk+1 and (before a chunk's first tap) its patch
    if (T_ < PS) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BI + 1) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BI) : "memory");
    if constexpr (!(AMB_ABL & 2)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (!(AMB_ABL & 4)) {
      constexpr int T1 = (T_ + 1) % 9, ky1 = T1 / 3, kx1 = T1 % 3;
      const char* Q = smem + ((T_ + 1 < 9 ? c : c + 1) & 1) * patch_bytes;
      const int toff1 = ky1 * rowoff + kx1 * PP;
#pragma unroll
      for (int t = 0; t < TN; ++t) wnx[t] = *reinterpret_cast<const half8_t*>(smem + fw + nstage * BSTAGE + t * 16 * BKB);
#pragma unroll
      for (int t = 0; t < HM; ++t) p0[t] = *reinterpret_cast<const half8_t*>(Q + pbase[t] + toff1);
    }
#if !AMB_SCHED
    __builtin_amdgcn_sched_barrier(0);
#endif
    if constexpr (!(AMB_ABL & 8)) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < HM; ++tm) acc[tn][HM + tm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[tn], p1[tm], acc[tn][HM + tm], 0, 0, 0);
#if AMB_SCHED >= 1 && AMB_SCHED <= 4
      // eight groups of (one read of the next tile's fragments, two MFMAs)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
#elif AMB_SCHED == 5 || AMB_SCHED == 6
      // the eight reads in pairs
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
#endif
    } else {
#pragma unroll
      for (int t = 0; t < HM; ++t) asm volatile("" ::"v"(p1[t]));
    }
    __builtin_amdgcn_sched_barrier(0);
    (void)stage;
  };
#ifndef AMB_NOPRIO
  if (wid >= NW / 2) __builtin_amdgcn_s_setprio(1);  // static priority for the second-dispatched half (conv_ring16_k SCHED 1)
#endif
  using std::integral_constant;
  for (int c = 0; c < nchunk; c += 2) {  // two chunks = 18 K-steps per trip: the weight registers alternate, 9 is odd
    kstep(integral_constant<int, 0>{}, c, wA, wB);
    kstep(integral_constant<int, 1>{}, c, wB, wA);
    kstep(integral_constant<int, 2>{}, c, wA, wB);
    kstep(integral_constant<int, 3>{}, c, wB, wA);
    kstep(integral_constant<int, 4>{}, c, wA, wB);
    kstep(integral_constant<int, 5>{}, c, wB, wA);
    kstep(integral_constant<int, 6>{}, c, wA, wB);
    kstep(integral_constant<int, 7>{}, c, wB, wA);
    kstep(integral_constant<int, 8>{}, c, wA, wB);
    kstep(integral_constant<int, 0>{}, c + 1, wB, wA);
    kstep(integral_constant<int, 1>{}, c + 1, wA, wB);
    kstep(integral_constant<int, 2>{}, c + 1, wB, wA);
    kstep(integral_constant<int, 3>{}, c + 1, wA, wB);
    kstep(integral_constant<int, 4>{}, c + 1, wB, wA);
    kstep(integral_constant<int, 5>{}, c + 1, wA, wB);
    kstep(integral_constant<int, 6>{}, c + 1, wB, wA);
    kstep(integral_constant<int, 7>{}, c + 1, wA, wB);
    kstep(integral_constant<int, 8>{}, c + 1, wB, wA);
  }
  __builtin_amdgcn_s_setprio(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the pieces issued past the end, the fragments read past the end
  __syncthreads();  // all fragment reads done before the epilogue reuses the buffers

  // ---- epilogue (conv_ring16_k's) ----
  // LDS map: [0, 4096) output-pixel table (BM ints); from 4096 the BatchNorm partial sums [col][s|q][wm*16 + pixel lane] and,
  // after the barrier that ends their use, one staging area per wave
  int* opix_s = reinterpret_cast<int*>(smem);
  float* red = reinterpret_cast<float*>(smem + 4096);
  for (int r = tid; r < BM; r += NTH) {
    int off = -1;
    if (r < p.nvalid) {
      const unsigned br = am_fastdiv((unsigned)r, p.w_mul, p.w_sh);
      const int xx = r - (int)br * p.W, yy = y0 + (int)br;
      if (yy < p.H) off = ((img * p.H + yy) * p.W + xx) * p.ldo + p.y_coff;
    }
    opix_s[r] = off;
  }
  const int cg = lane >> 4, pl = lane & 15;  // channel group (4 channels each) and pixel lane of the accumulator map
  if (p.stats != nullptr) {
    // dead tile rows / rows below the image hold the convolution of some other pixel: not part of the statistics
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const float m = (vmask >> tm) & 1u ? 1.f : 0.f;
      const f32x4 m4 = {m, m, m, m};
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tn][tm] *= m4;
    }
    constexpr int SL = WM * 16;
    constexpr int CS = 2 * SL + 4;  // floats per column: s[SL] | q[SL] | 16 B pad (conv_ring16_k: bank-conflict-free column reads)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      f32x4 sv = acc[tn][0], qv = acc[tn][0] * acc[tn][0];
#pragma unroll
      for (int tm = 1; tm < TM; ++tm) {
        sv += acc[tn][tm];
        qv = __builtin_elementwise_fma(acc[tn][tm], acc[tn][tm], qv);
      }
      const int slot = (wm * 16 + pl) ^ ((cg & 1) << 4) % SL;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = wn * TN * 16 + tn * 16 + cg * 4 + r;
        red[col * CS + slot] = sv[r];
        red[col * CS + SL + slot] = qv[r];
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      double s = 0.0, q = 0.0;
      const float4* rs = reinterpret_cast<const float4*>(red + tid * CS);
      const float4* rq = reinterpret_cast<const float4*>(red + tid * CS + SL);
#pragma unroll
      for (int a = 0; a < SL / 4; ++a) {
        const float4 u = rs[a], v = rq[a];
        s += (double)((u.x + u.y) + (u.z + u.w));
        q += (double)((v.x + v.y) + (v.z + v.w));
      }
      double* st = p.stats + (size_t)(lb % AM_STATS_REPLICAS) * 2 * p.N;
      atomicAdd(st + n0 + tid, s);
      atomicAdd(st + p.N + n0 + tid, q);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // (raw barrier: the fp64 atomics stay in flight)
  asm volatile("" ::: "memory");
  {
    constexpr int WCOLS = TN * 16;      // channels per wave (64)
    constexpr int SP = WCOLS * 2 + 16;  // staging row pitch in bytes
    char* stg = smem + 4096 + wid * (TM * 16) * SP;
    float bv[TN][4];
    const T* __restrict__ res = static_cast<const T*>(p.res);
    const bool relu_early = p.relu && res == nullptr;
    const bool plain = p.bias == nullptr && !relu_early;  // BN layers (almost every launch): convert and stage, nothing else
    if (!plain) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = n0 + wn * WCOLS + tn * 16 + cg * 4 + r;
          bv[tn][r] = (p.bias != nullptr && col < p.N) ? p.bias[col] : 0.f;
        }
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        f32x4 v = acc[tn][tm];
        if (!plain) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] += bv[tn][r];
            if (relu_early) v[r] = fmaxf(v[r], 0.f);
          }
        }
        half4_t h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = (half_t)v[r];
        *reinterpret_cast<half4_t*>(stg + (tm * 16 + pl) * SP + (tn * 16 + cg * 4) * 2) = h;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    constexpr int CPRW = WCOLS / 8;  // 16-byte chunks per row
    const int ncols = (p.N + 7) & ~7;
    constexpr int NIT = TM * 16 * CPRW / 64;
    int offv[NIT];
    uint4 dat[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * 64 + lane;
      const int row = q / CPRW, cc = q - row * CPRW;
      offv[it] = opix_s[wm * TM * 16 + row];
      dat[it] = *reinterpret_cast<const uint4*>(stg + row * SP + cc * 16);
    }
    const int col0 = n0 + wn * WCOLS + (lane % CPRW) * 8;
    const bool col_ok = col0 < ncols;
    if (res != nullptr) {
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
      if (offv[it] >= 0 && col_ok) *reinterpret_cast<uint4*>(y + (unsigned)(offv[it] + col0)) = dat[it];
  }
}

template <int PS>
int launch(const Params& p, hipStream_t s) {
  constexpr size_t RED = (size_t)BN * (2 * (WM * 16) + 4) * 4, STG = (size_t)NW * (TM * 16) * (TN * 32 + 16);
  constexpr size_t EPI = 4096 + (RED > STG ? RED : STG);
  const size_t ring = (size_t)2 * p.npiece * 1024 + NSTG * BSTAGE;
  const size_t lds = ring > EPI ? ring : EPI;
  if (lds > 160 * 1024) return AM_ERR_UNSUPPORTED;
  static size_t attr_dev[AM_MAX_DEVICES] = {};
  size_t& attr = attr_dev[am_current_device()];
  if (lds > attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_band16_k<PS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return AM_ERR_LAUNCH;
    attr = lds;
  }
  hipLaunchKernelGGL((conv_band16_k<PS>), dim3(p.ntiles), dim3(NTH), lds, s, p);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

}  // namespace amb

// Returns AM_ERR_UNSUPPORTED unless the geometry is a dense 3x3 / stride 1 / pad 1 convolution (canonical tap order, as fwd_geom
// and the stride-1 dgrad plan produce it), f16, Cin a multiple of 64, N a multiple of 256, an image width whose row bands fill at
// least 85 % of the 256-row tiles, and enough tiles to fill the chip.
int am_conv_band16_f16(const am_conv_geom* g, const void* x, const void* w, const float* bias, int relu, const void* res, void* y,
                       double* stats, hipStream_t s) {
  using namespace amb;
  if (g->ntaps != 9 || g->pix_shift != 31 || g->N < 256 || g->N % 256 != 0 || g->krun % 64 != 0 || g->krun < 64 || g->osplit > 0) return AM_ERR_UNSUPPORTED;
  if (g->iys != 1 || g->ixs != 1 || g->oys != 1 || g->oxs != 1 || g->oy0 != 0 || g->ox0 != 0) return AM_ERR_UNSUPPORTED;
  if (g->MH != g->IH || g->MW != g->IW || g->OH != g->IH || g->OW != g->IW) return AM_ERR_UNSUPPORTED;
  for (int t = 0; t < 9; ++t)
    if (g->dy[t] != t / 3 - 1 || g->dx[t] != t % 3 - 1) return AM_ERR_UNSUPPORTED;
  const int W = g->IW, H = g->IH;
  if (W < 16 || W > 128) return AM_ERR_UNSUPPORTED;
  Params p;
  p.RB = BM / W;
  if (p.RB > H) p.RB = H;
  p.bands = am_cdiv(H, p.RB);
  p.nvalid = p.RB * W;
  p.ntn = g->N / BN;
  p.ntiles = g->B * p.bands * p.ntn;
  p.npix = (p.RB + 2) * (W + 2);
  p.npiece = am_cdiv((long long)p.npix * 6, 64);
  if (p.npiece > MAX_PIECE || p.npiece < NW) return AM_ERR_UNSUPPORTED;
  p.ps = am_cdiv(p.npiece, NW);
  // dead tile rows waste their MFMAs: at most 15 % (45 x 80: 6 %, 23 x 40: 10 %); and the chip must be filled (conv_ring16_k's gate)
  if ((long long)H * W * 100 < (long long)p.bands * BM * 85) return AM_ERR_UNSUPPORTED;
  if (p.ntiles < am_tuning(AM_TUNE_BAND_MIN_TILES)) return AM_ERR_UNSUPPORTED;
  const long long x_bytes = (long long)g->B * H * W * g->ldi * 2;
  const long long y_elems = ((long long)g->B * H * W + 1) * g->ldo + g->y_coff;
  const long long w_bytes = (long long)am_conv_npad(g->N) * 9 * g->krun * 2;
  if (x_bytes >= (1ll << 31) || y_elems >= (1ll << 31) || w_bytes >= (1ll << 31)) return AM_ERR_UNSUPPORTED;
  p.x = x; p.w = w; p.y = y; p.bias = bias; p.res = res; p.stats = stats;
  p.B = g->B; p.H = H; p.W = W; p.ldi = g->ldi; p.x_coff = g->x_coff; p.ldo = g->ldo; p.y_coff = g->y_coff;
  p.Cin = g->krun; p.N = g->N; p.relu = relu;
  p.nchunk = g->krun / 32;  // even: Cin % 64 == 0
  p.x_bytes = (unsigned)x_bytes;
  p.w_bytes = (unsigned)w_bytes;
  am_fastdiv_make((unsigned)W, &p.w_mul, &p.w_sh);
  am_fastdiv_make((unsigned)(W + 2), &p.pw_mul, &p.pw_sh);
  g_am_conv_variant = AM_CV_BAND16_256x256;
  switch (p.ps) {
    case 1: return launch<1>(p, s);
    case 2: return launch<2>(p, s);
    case 3: return launch<3>(p, s);
    case 4: return launch<4>(p, s);
    case 5: return launch<5>(p, s);
    default: return launch<6>(p, s);
  }
}
