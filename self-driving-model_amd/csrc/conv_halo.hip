// Halo-staged forward / input-gradient kernel for dense 3x3 / stride-1 / pad-1 convolutions with 64 < N <= 128 output
// channels, f16 (ResNet-18 layer2 and its dgrad): conv_ring_k<256,128>'s MFMA / weight-ring structure with the PIXEL operand
// staged the way conv3x3_c64n64_duo_k stages it.
//
// conv_ring_k gathers the 256 pixel rows of a K-step once per (tap, 32-channel chunk): 16 KB of LDS-DMA per K-step next to
// 8 KB of weights, 24 wave-instructions per K-step -- the N = 128 tile is bound by the CU's vector-memory issue, not by its
// MFMAs (issuing the pixel pieces of one tap only ran the layer2 shape 21 % faster).  Here a workgroup owns an
// 8 x 32-pixel output tile and stages, per 32-channel chunk, the 10 x 34-pixel input patch ONCE (27 KB instead of 9 x 16 KB);
// the nine taps of the chunk read their pixel fragments from it at compile-time offsets.  Zero padding and image borders are
// in the patch itself (out-of-image pixels are fetched at an out-of-range buffer offset: the hardware writes zeros), so the
// fragment path has no masks.  Per K-step a wave issues one weight piece and, on the first four taps of a chunk, one piece of
// the NEXT chunk's patch (double-buffered): 10.7 wave-instructions per K-step instead of 24, 2.3x fewer HBM / L2 bytes.
//
// LDS: patch pixel pitch 80 B (64 B of channels + 16 B pad): 5 is odd, so the 16 pixels of a ds_read_b128 lane group fall on
// 16 different 16-byte bank groups at any base -- no swizzle, every (tap, sub-tile, k16) offset an instruction immediate.
// The LDS-DMA destination is lane-linear, so lane j of piece i owns 16-byte granule 64 i + j = (pixel g / 5, chunk g % 5);
// chunk 4 is the pad (requested out of range).  Weights: three 8 KB stages exactly as conv_ring_k (64-byte rows, XOR swizzle).
// 78 KB of LDS + an 80 KB epilogue alias: two workgroups per CU.
//
// PRE (am_conv_gemm_prebn): the convolution runs on relu(x * pre_scale[c] + pre_shift[c]) -- the BatchNorm + ReLU of the
// producing layer (bn.hip bn_apply_k's fp32 arithmetic) -- applied to the staged patch, once per element instead of once per
// tap: every wave rewrites the 16-byte granules its own DMA pieces delivered, one K-step before the chunk's first use.  Pixels
// outside the image stay zero (the padding applies to the transformed tensor).
#include "am_common.h"

#ifndef AMH_SCHED
#define AMH_SCHED 0  // 1: fragment reads / DMA pieces interleaved with the MFMAs (conv_band16_k gained 15-20 % from it; here +-0 over two A/B runs: 4 MFMAs per half K-step), 0: issued as a block
#endif

namespace amh {

constexpr int TH = 8, TW = 32;            // output tile
constexpr int PH = TH + 2, PW = TW + 2;   // input patch
constexpr int NPIX = PH * PW;             // 340
constexpr int PP = 80;                    // LDS bytes per patch pixel
constexpr int NPIECE = (NPIX * 5 + 63) / 64;  // 27 wave-instructions of 64 x 16 B
constexpr int PATCH_BYTES = NPIECE * 1024;    // 27648
constexpr int BM = TH * TW, BN = 128, WM = 4, WN = 2, NW = 8, NTH = NW * 64;
constexpr int TM = BM / WM / 32, TN = BN / WN / 32;  // 2 x 2 MFMA tiles (32x32) per wave
constexpr int BKB = 64, BSTAGE = BN * BKB, NSTG = 3;
constexpr int PSLOTS = 4;                 // K-steps of a chunk on which a wave issues one patch piece (8 waves x 4 >= 27)
constexpr int B_BASE = 2 * PATCH_BYTES;   // weight ring behind the two patch buffers
constexpr int RING_BYTES = B_BASE + NSTG * BSTAGE;
constexpr int MAX_PRE_C = 256;            // input channels of the PRE form (scale / shift table behind the ring)
constexpr int AFF_BASE = RING_BYTES;
constexpr int SP = TN * 64 + 16;          // epilogue staging pitch per output pixel
constexpr int EPI_BYTES = 8192 + NW * (TM * 32) * SP;
constexpr int LDS_BYTES = (RING_BYTES + 2 * MAX_PRE_C * 4) > EPI_BYTES ? (RING_BYTES + 2 * MAX_PRE_C * 4) : EPI_BYTES;
constexpr unsigned OOB = 0x80000000u;
static_assert(NW * PSLOTS >= NPIECE && 2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");

struct HaloParams {
  const void* x;
  const void* w;   // packed [npad(N)][9 * Cin] halves, tap-major (forward / dgrad packing)
  void* y;
  const float* bias;
  const void* res;
  double* stats;
  int B, H, W, ldi, x_coff, ldo, y_coff, Cin, N, relu;
  int tiles_y, tiles_x, ntiles, nchunk;
  unsigned x_bytes, w_bytes;
  const float* pre_scale;  // PRE: [Cin] each
  const float* pre_shift;
};

typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ void buffer_to_lds16(const void* base, unsigned bytes, char* dst, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000),
                                           (lds_ptr)dst, 16, voff, soff, 0, 0);
}

template <bool PRE>
__global__ __launch_bounds__(NTH, 4) void conv_halo_k(const HaloParams p) {
  typedef half_t T;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  T* __restrict__ y = static_cast<T*>(p.y);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;

  const int lb = xcd_remap(blockIdx.x, p.ntiles);
  const int tpi = p.tiles_y * p.tiles_x;
  const int img = lb / tpi, trem = lb - img * tpi;
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;

  // ---- loader state ----
  // patch: slot s of this wave is piece s * 8 + wid; the last slot has only three pieces left, waves 3..7 fetch their OWN slot-0
  // piece again (same bytes, same place, same wave: nothing races), so every wave issues the same number of pieces per K-step,
  // which the counted vmcnt waits rely on
  unsigned pvo[PSLOTS];
  int pdst[PSLOTS];
#pragma unroll
  for (int s = 0; s < PSLOTS; ++s) {
    int piece = s * 8 + wid;
    piece = piece >= NPIECE ? wid : piece;
    const int gidx = piece * 64 + lane;
    const int pix = (int)__umulhi((unsigned)gidx, 0x33333334u);  // gidx / 5 (exact below 2^30)
    const int cc = gidx - pix * 5;
    const int prow = pix / PW, pcol = pix - prow * PW;
    const int iy = iy0 + prow, ix = ix0 + pcol;
    const bool ok = cc < 4 && pix < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    pvo[s] = ok ? (unsigned)((((img * p.H + iy) * p.W + ix) * p.ldi + p.x_coff) * 2 + cc * 16) : OOB;
    pdst[s] = piece * 1024;
  }
  const int own_slots = wid < NPIECE - 8 * (PSLOTS - 1) ? PSLOTS : PSLOTS - 1;  // slots whose piece this wave transforms (PRE)
  // weights: this wave's 16 rows of the 128-row tile, one piece per K-step
  const int lrow = lane >> 2, cpos = lane & 3;
  const int brow = wid * 16 + lrow;
  const unsigned b_off = (unsigned)(brow * (9 * p.Cin) * 2 + ((cpos ^ ((brow >> 2) & 3)) << 4));  // rows past npad(N): out of range -> zeros
  const int krun2 = __builtin_amdgcn_readfirstlane(p.Cin * 2);
  const int nchunk = __builtin_amdgcn_readfirstlane(p.nchunk);

  // K index kk = chunk * 9 + tap; weight tile kk: K bytes [tap * Cin * 2 + chunk * 64, +64)
  auto issue_b = [&](int tap, int chunk, int stage) {
    buffer_to_lds16(p.w, p.w_bytes, smem + B_BASE + stage * BSTAGE + wid * 1024, b_off, (unsigned)(tap * krun2 + chunk * BKB));
  };
  auto issue_patch = [&](int slot, int chunk, int buf) {
    // past the last chunk the pieces are still issued (out of range: zeros into the idle buffer), so every K-step keeps its count
    buffer_to_lds16(p.x, p.x_bytes, smem + buf * PATCH_BYTES + pdst[slot], chunk < nchunk ? pvo[slot] : OOB, (unsigned)(chunk * BKB));
  };

#pragma unroll
  for (int s = 0; s < PSLOTS; ++s) issue_patch(s, 0, 0);
  issue_b(0, 0, 0);
  issue_b(1, 0, 1);

  // PRE: relu(x * scale + shift) on this wave's granules of the patch of `chunk` in buffer `buf` (after the wave's own vmcnt
  // wait, before the barrier that publishes the patch); out-of-image granules stay zero
  float* aff = reinterpret_cast<float*>(smem + AFF_BASE);  // [2][MAX_PRE_C]
  auto transform_slot = [&](int s, int chunk, int buf) {
    {
      if (s < own_slots && pvo[s] != OOB) {
        // (in two halves of four channels, each finished before the next is loaded: the kernel sits at its 128-register budget --
        // four waves per SIMD -- and a spill reload next to the LDS-DMA costs a vmcnt(0): 21 spills made this form 27 % slower)
        half4_t* slot4 = reinterpret_cast<half4_t*>(smem + buf * PATCH_BYTES + pdst[s] + lane * 16);
        const int c0 = chunk * 32 + (int)((pvo[s] >> 4) & 3u) * 8;  // (PRE: x_coff = 0 and whole pixels of 64-byte multiples: bits 4-5 are the chunk)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const half4_t v = slot4[h];
          const f32x4 sc = *reinterpret_cast<const f32x4*>(aff + c0 + 4 * h);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(aff + MAX_PRE_C + c0 + 4 * h);
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (half_t)fmaxf((float)v[e] * sc[e] + sh[e], 0.f);
          slot4[h] = o;
          asm volatile("" ::: "memory");
        }
      }
    }
  };
  auto transform = [&](int chunk, int buf) {
#pragma unroll
    for (int s = 0; s < PSLOTS; ++s) transform_slot(s, chunk, buf);
  };
  if (PRE) {
    for (int i = tid; i < p.Cin; i += NTH) {
      aff[i] = p.pre_scale[i];
      aff[MAX_PRE_C + i] = p.pre_shift[i];
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment addressing.  Pixels: lane reads patch pixel (2*wm + tm + ky, (lane & 31) + kx), k16 chunk 2*ks + (lane >> 5).
  const int abase = ((2 * wm) * PW + (lane & 31)) * PP + (lane >> 5) * 16;
  const int frow_b = wn * TN * 32 + (lane & 31);
  int fb[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) fb[ks] = B_BASE + frow_b * BKB + (((ks * 2 + (lane >> 5)) ^ ((frow_b >> 2) & 3)) << 4);

  half8_t a0[TM], b0[TN], a1[TM], b1[TN];
  asm volatile("s_waitcnt vmcnt(1)" ::: "memory");  // patch 0 and weight tile 0 landed (tile 1 may be in flight)
  if (PRE) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the scale / shift table is in LDS
    asm volatile("" ::: "memory");
    transform(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int t = 0; t < TM; ++t) a0[t] = *reinterpret_cast<const half8_t*>(smem + abase + (t * PW) * PP);
#pragma unroll
  for (int t = 0; t < TN; ++t) b0[t] = *reinterpret_cast<const half8_t*>(smem + fb[0] + t * 32 * BKB);

  // Software pipeline as conv_ring_k (rotated by half a K-step); the nine taps of a chunk are unrolled: stage = tap % 3,
  // fragment offsets and vmcnt counts are immediates.
  for (int c = 0; c < nchunk; ++c) {
    const char* P = smem + (c & 1) * PATCH_BYTES + abase;
    const char* Pn = smem + ((c + 1) & 1) * PATCH_BYTES + abase;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ky = t / 3, kx = t - ky * 3;
      const int stage = t % 3, nstage = (t + 1) % 3, istage = (t + 2) % 3;
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the fragments read half a K-step ago
      if (PRE && t >= 4 && t < 4 + PSLOTS) {
        // One slot of the next chunk's patch per K-step on taps 4..7 (instead of four on one: the rewrite sits under the other
        // waves' MFMAs), placed where only half of the fragment registers are live (before this K-step's second-half fragments
        // are requested): the kernel sits at its 128-register budget, and a spill reload next to the LDS-DMA costs a vmcnt(0)
        // -- 21 spills made this form 27 % slower than it had to be.  Order: slots 1, 2, 3, then 0 -- waves 3..7 fetch their
        // slot-0 piece a second time on tap 3 (see the loader state), so slot 0 is rewritten only once that copy has landed too.
        // Slot s was issued on tap s behind that tap's weight piece; the counts below are the pieces issued after it.
        // Published by the barrier of tap 7, first read behind the barrier of tap 8.
        if (t == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // slot 1: taps 2, 3 (two pieces each)
        else if (t == 5) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // slot 2: tap 3 (two), tap 4
        else if (t == 6) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // slot 3: taps 4, 5
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");              // slot 0 and its second copy (tap 3): taps 4, 5, 6
        if (c + 1 < nchunk) transform_slot(t == 7 ? 0 : t - 3, c + 1, (c + 1) & 1);
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) a1[tm] = *reinterpret_cast<const half8_t*>(P + ((tm + ky) * PW + kx) * PP + 32);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b1[tn] = *reinterpret_cast<const half8_t*>(smem + fb[1] + stage * BSTAGE + tn * 32 * BKB);
      // weight tile kk+2 into the stage of tile kk-1 (everyone finished reading it before the last barrier); patch piece of the
      // next chunk into the other patch buffer (last read during the previous chunk)
      if (t + 2 < 9) issue_b(t + 2, c, istage);
      else issue_b(t + 2 - 9, c + 1, istage);
      if (t < PSLOTS) issue_patch(t, c + 1, (c + 1) & 1);
#if !AMH_SCHED
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[tm], b0[tn], acc[tm][tn], 0, 0, 0);
#if AMH_SCHED
      // reads and LDS-DMA pieces interleaved with the four MFMAs of the half K-step (conv_band16_k's finding: the LDS port is
      // the second bound, bursts of reads behind the barrier queue up against the DMA writes)
#pragma unroll
      for (int i = 0; i < TM * TN; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        if (i < 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read: an LDS-DMA piece
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      // everything older than this K-step's own pieces has landed: weight tile kk+1, and (before a chunk's first tap) its patch
      if (t < PSLOTS) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      {
        const int t1 = (t + 1) % 9, ky1 = t1 / 3, kx1 = t1 - ky1 * 3;
        const char* Q = t + 1 < 9 ? P : Pn;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) a0[tm] = *reinterpret_cast<const half8_t*>(Q + ((tm + ky1) * PW + kx1) * PP);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b0[tn] = *reinterpret_cast<const half8_t*>(smem + fb[0] + nstage * BSTAGE + tn * 32 * BKB);
      }
#if !AMH_SCHED
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[tm], b1[tn], acc[tm][tn], 0, 0, 0);
#if AMH_SCHED
#pragma unroll
      for (int i = 0; i < TM * TN; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the pieces issued past the end, the fragments read past the end
  __syncthreads();

  // ---- epilogue (conv_ring_k's): statistics, bias / ReLU / residual, LDS-staged 16-byte stores ----
  int* opix_s = reinterpret_cast<int*>(smem + 4096);
  for (int r = tid; r < BM; r += NTH) {
    const int oy = ty * TH + (r >> 5), ox = tx * TW + (r & 31);
    opix_s[r] = (oy < p.H && ox < p.W) ? ((img * p.H + oy) * p.W + ox) * p.ldo + p.y_coff : -1;
  }
  if (p.stats != nullptr) {
    if (ty * TH + TH > p.H || tx * TW + TW > p.W) {
      // edge tile: pixels outside the image are not conv outputs (their patch neighbours inside the image are not zero)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const bool row_ok = ty * TH + 2 * wm + tm < p.H;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool ok = row_ok && tx * TW + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) < p.W;
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tm][tn][r] = ok ? acc[tm][tn][r] : 0.f;
        }
      }
    }
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
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
    if (tid < BN && tid < p.N) {
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        s += (double)red[(a * BN + tid) * 2 + 0];
        q += (double)red[(a * BN + tid) * 2 + 1];
      }
      double* st = p.stats + (size_t)(lb % AM_STATS_REPLICAS) * 2 * p.N;
      atomicAdd(st + tid, s);
      atomicAdd(st + p.N + tid, q);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  {
    char* stg = smem + 8192 + wid * (TM * 32) * SP;
    const T* __restrict__ res = static_cast<const T*>(p.res);
    const bool relu_early = p.relu && res == nullptr;
    if (p.bias == nullptr && !relu_early) {
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
        const int col = wn * TN * 32 + tn * 32 + (lane & 31);
        const float bv = (p.bias != nullptr && col < p.N) ? p.bias[col] : 0.f;
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    constexpr int CPRW = TN * 4;
    const int ncols = (p.N + 7) & ~7;
    constexpr int NIT = TM * TN * 2;
    int offv[NIT];
    uint4 dat[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * 64 + lane;
      const int row = q / CPRW, cc = q - row * CPRW;
      offv[it] = opix_s[wm * TM * 32 + row];
      dat[it] = *reinterpret_cast<const uint4*>(stg + row * SP + cc * 16);
    }
    const int col0 = wn * TN * 32 + (lane % CPRW) * 8;
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

}  // namespace amh

// Returns AM_ERR_UNSUPPORTED unless the geometry is a dense 3x3 / stride 1 / pad 1 convolution (canonical tap order, as
// fwd_geom and the stride-1 dgrad plan produce it) with Cin a multiple of 32, 64 < N <= 128, f16, large enough to fill the chip.
int am_conv_halo_pre_f16(const am_conv_geom* g, const void* x, const float* pre_scale, const float* pre_shift, const void* w,
                         const float* bias, int relu, const void* res, void* y, double* stats, hipStream_t s);

int am_conv_halo_f16(const am_conv_geom* g, const void* x, const void* w, const float* bias, int relu, const void* res, void* y,
                     double* stats, hipStream_t s) {
  return am_conv_halo_pre_f16(g, x, nullptr, nullptr, w, bias, relu, res, y, stats, s);
}

// pre_scale / pre_shift (both or neither, [Cin], Cin <= 256): the convolution runs on relu(x * pre_scale[c] + pre_shift[c]).
int am_conv_halo_pre_f16(const am_conv_geom* g, const void* x, const float* pre_scale, const float* pre_shift, const void* w,
                         const float* bias, int relu, const void* res, void* y, double* stats, hipStream_t s) {
  using namespace amh;
  const bool pre = pre_scale != nullptr;
  if (pre && (g->krun > MAX_PRE_C || g->krun != g->ldi || g->x_coff != 0)) return AM_ERR_UNSUPPORTED;
  if (g->ntaps != 9 || g->pix_shift != 31 || g->N <= 64 || g->N > BN || g->krun % 32 != 0 || g->krun < 32 || g->osplit > 0) return AM_ERR_UNSUPPORTED;
  if (g->iys != 1 || g->ixs != 1 || g->oys != 1 || g->oxs != 1 || g->oy0 != 0 || g->ox0 != 0) return AM_ERR_UNSUPPORTED;
  if (g->MH != g->IH || g->MW != g->IW || g->OH != g->IH || g->OW != g->IW) return AM_ERR_UNSUPPORTED;
  for (int t = 0; t < 9; ++t)
    if (g->dy[t] != t / 3 - 1 || g->dx[t] != t % 3 - 1) return AM_ERR_UNSUPPORTED;
  HaloParams p;
  p.tiles_y = am_cdiv(g->IH, TH);
  p.tiles_x = am_cdiv(g->IW, TW);
  p.ntiles = g->B * p.tiles_y * p.tiles_x;
  // below ~a tile per CU the gather kernels' smaller grids do better; tiles that overhang the image waste their MFMAs on it
  // (45 x 80: 22 % -- the ring kernel is faster there; 90 x 160: 6 %): at most 15 % unless the caller forces the kernel
  const int min_tiles = am_tuning(AM_TUNE_HALO_MIN_TILES);
  if (p.ntiles < min_tiles) return AM_ERR_UNSUPPORTED;
  if (min_tiles > 1 && (long long)g->IH * g->IW * 100 < (long long)p.tiles_y * TH * p.tiles_x * TW * 85) return AM_ERR_UNSUPPORTED;
  const long long x_bytes = (long long)g->B * g->IH * g->IW * g->ldi * 2;
  const long long y_elems = ((long long)g->B * g->OH * g->OW + 1) * g->ldo + g->y_coff;
  const long long w_bytes = (long long)am_conv_npad(g->N) * 9 * g->krun * 2;
  if (x_bytes >= (1ll << 31) || y_elems >= (1ll << 31) || w_bytes >= (1ll << 31)) return AM_ERR_UNSUPPORTED;
  p.x = x; p.w = w; p.y = y; p.bias = bias; p.res = res; p.stats = stats;
  p.B = g->B; p.H = g->IH; p.W = g->IW; p.ldi = g->ldi; p.x_coff = g->x_coff; p.ldo = g->ldo; p.y_coff = g->y_coff;
  p.Cin = g->krun; p.N = g->N; p.relu = relu;
  p.nchunk = g->krun / 32;
  p.x_bytes = (unsigned)x_bytes;
  p.w_bytes = (unsigned)w_bytes;
  p.pre_scale = pre_scale; p.pre_shift = pre_shift;
  static bool attr_done_dev[AM_MAX_DEVICES][2] = {};
  bool& attr_done = attr_done_dev[am_current_device()][pre ? 1 : 0];
  if (!attr_done) {
    const void* fn = pre ? reinterpret_cast<const void*>(conv_halo_k<true>) : reinterpret_cast<const void*>(conv_halo_k<false>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return AM_ERR_LAUNCH;
    attr_done = true;
  }
  g_am_conv_variant = AM_CV_HALO_256x128;
  if (pre) hipLaunchKernelGGL(conv_halo_k<true>, dim3(p.ntiles), dim3(NTH), LDS_BYTES, s, p);
  else hipLaunchKernelGGL(conv_halo_k<false>, dim3(p.ntiles), dim3(NTH), LDS_BYTES, s, p);
  AM_CHECK_LAUNCH();
  return AM_OK;
}
