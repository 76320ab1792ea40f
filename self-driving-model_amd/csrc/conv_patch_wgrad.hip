// Weight gradient of the C -> C channel 3x3 / stride 1 / pad 1 convolutions with C = 64 (ResNet layer1) and C = 128 (layer2) on
// gfx950, f16.  The text below describes the C = 64 form; the C = 128 form follows it.
//
// dW[n][t*64 + c] = sum over output pixels of dY[pixel][n] * X[pixel + tap t][c]: a [64] x [576] result contracted over
// B*H*W pixels (920 k at B = 16, 720p), i.e. tiny output and a very long reduction.  The generic conv_wgrad_k gathers X once
// per tap (nine reads of the activation through the load path) and splits the pixels over ~1000 workgroups that meet in
// fp32 atomics: 122 us at B = 16 (555 TFLOP/s) for 236 MB of operands.  Here, as in conv_s2d_wgrad.hip, a persistent
// workgroup walks 8x32-pixel output tiles, stages each tile's dY block and its 10x34 input patch ONCE in LDS (pixel-major,
// pitch 192 B = 64 mod 128: both MFMA operands come out of the ds_read_b64_tr_b16 transposing read) and keeps the whole
// gradient in registers: twelve waves = 2 output-channel blocks x 3 horizontal taps x 2 input-channel blocks, each with the
// three vertical taps of its column (3 accumulators of 32x32; as v_mfma_f32_16x16x32_f16 over whole 32-pixel rows: see M16).  A wave walks the patch rows once: the row's fragment feeds
// the three taps (output rows r, r-1, r-2), so an MFMA costs 0.75 KB of LDS reads.  The next tile's global loads are issued
// before the MFMA phase and land under it.  One flush per workgroup: plain stores into its slab of the caller's workspace
// (am_conv_wgrad_ws; wgrad_reduce_k sums the slabs) or, without a workspace, fp32 atomics.
// Where a tile period goes at B = 32 (5.8 us, measured by leaving parts out): MFMA phase 3.0 us (floor at the ~2.0 GHz the chip
// holds under MFMA load: 2.3 us), tile store + two barriers 0.6 us, 1.0 us waiting for the next tile's loads (all CUs request
// their 76 KB in the same burst after the barrier: ~4 us to drain at HBM rate, the MFMA phase covers 3); per launch another
// ~33 us of flush + reduce pass.
//
// C = 128 (layer2: [128] x [1152], 230 k pixels at B = 16): the gradient (590 KB of fp32) does not fit one CU's registers, so the
// three horizontal taps go to three workgroups (blockIdx / G; the three of a tile stream share an XCD, i.e. an L2, because G is a
// multiple of 8) that write disjoint column ranges of the same slab.  16 waves = 4 output-channel blocks x 4 input-channel blocks
// with the three vertical taps as accumulators, 8x16-pixel tiles, a 10x16 patch window shifted by the tap (pitch 320 B = 64 mod
// 128), 92 KB of LDS.  Replaces wgrad_ring_k on these layers (its 128x256 tiles over ~100 pixel chunks: 124 us + 32 us of reduce
// pass at B = 16, 548 TFLOP/s).
#include "am_common.h"

namespace apw {

constexpr int TH = 8, PH = TH + 2;

struct Params {
  const void* x;    // [B, H, W, ldi] halves (+ x_coff)
  const void* dy;   // [B, H, W, ldo] halves (+ y_coff)
  float* dw;        // atomic form: packed [C][9 * C] fp32, accumulated (scaled)
  float* ws;        // slab form: tile stream j stores its unscaled partial at ws + j * ws_stride
  long long ws_stride;
  float scale;
  int B, H, W, ldi, x_coff, ldo, y_coff;
  int tiles_y, tiles_x, ntiles;
  int streams;      // tile streams (= slabs); grid = streams * (KXB ? 3 : 1)
};

typedef __attribute__((address_space(3))) s4v* lds_s4v;

// 32 channels x 8 pixels of a pixel-major LDS tile as an MFMA operand (lane: channel lane % 32, pixels 8 * (lane / 32) ..+7)
template <int PITCH>
__device__ __forceinline__ half8_t tr_frag(const char* lo_addr) {
  const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v)(lo_addr));
  const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v)(lo_addr + 4 * PITCH));
  const half4_t l4 = __builtin_bit_cast(half4_t, lo), h4 = __builtin_bit_cast(half4_t, hi);
  return half8_t{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
}

// C: channels (in = out).  TW: tile width.  KXB: the horizontal tap comes from the block index (three workgroups per tile stream,
// patch = the TW columns of that tap) instead of the wave index (patch = TW + 2 columns).
template <int C, int TW, bool KXB>
struct Cfg {
  static constexpr int CB = C / 32;                       // 32-channel blocks
  static constexpr int NW = CB * CB * (KXB ? 1 : 3), NTH = NW * 64;
  static constexpr int PW = KXB ? TW : TW + 2;
  static constexpr int PITCH = C * 2 + 64;                // LDS bytes per pixel: = 64 (mod 128)
  static constexpr int CPP = C / 8;                       // 16-byte chunks per pixel
  static constexpr int PATCH_BYTES = PH * PW * PITCH, DY_BYTES = TH * TW * PITCH, LDS_BYTES = PATCH_BYTES + DY_BYTES;
  static constexpr int PCHUNKS = PH * PW * CPP, DCHUNKS = TH * TW * CPP;
  static constexpr int PCH = (PCHUNKS + NTH - 1) / NTH, DCH = (DCHUNKS + NTH - 1) / NTH;
  static constexpr int KTOT = 9 * C;
  static constexpr int XH = TW / 16;
  static_assert(NTH % CPP == 0 && PITCH % 128 == 64 && TW % 16 == 0 && NTH <= 1024, "");
};

// M16: v_mfma_f32_16x16x32_f16 with a whole 32-pixel tile row as the k of one MFMA (TW = 32 only) instead of 32x32x16 over half rows:
// the same fragment bytes and MFMA cycles, less power per FLOP (scratch/mfma_probe: +13 % sustained under the power limit)
template <int C, int TW, bool KXB, bool M16>
__global__ __launch_bounds__((Cfg<C, TW, KXB>::NTH)) void conv_patch_wgrad_k(const Params p) {
  static_assert(!M16 || TW == 32, "M16: one MFMA spans a 32-pixel tile row");
  using K = Cfg<C, TW, KXB>;
  constexpr int NTH = K::NTH, PW = K::PW, PITCH = K::PITCH, CPP = K::CPP, PCH = K::PCH, DCH = K::DCH, CB = K::CB;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* patch = smem;
  char* dYs = smem + K::PATCH_BYTES;

  const half_t* __restrict__ x = static_cast<const half_t*>(p.x) + p.x_coff;
  const half_t* __restrict__ dy = static_cast<const half_t*>(p.dy) + p.y_coff;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int gq = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  const int nblk = wid % CB, cblk = (wid / CB) % CB;
  const int kx = KXB ? (int)blockIdx.x / p.streams : wid / (CB * CB);
  const int stream = KXB ? (int)blockIdx.x % p.streams : (int)blockIdx.x;

  f32x16 acc[M16 ? 1 : 3];      // 32x32x16: [ky]
  f32x4 acc4[M16 ? 3 : 1][2][2];  // 16x16x32: [ky][16-channel half of the output block][16-channel half of the input block]
  if constexpr (M16) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc4[i][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  }

  // which patch pixel / dY pixel and 16-byte piece a thread fetches never changes: (row << 8 | column) per chunk
  int ppos[PCH];
#pragma unroll
  for (int k = 0; k < PCH; ++k) {
    const int pix = (tid + k * NTH) / CPP;
    const int prow = pix / PW;
    ppos[k] = (prow << 8) | (pix - prow * PW);
  }
  const int sub = (tid % CPP) * 8;  // NTH is a multiple of CPP: the channel piece does not depend on k
  const int xoff = KXB ? kx - 1 : -1;  // image column of patch column 0, relative to the tile

  // (a second register set, so that a tile's loads have two tile periods to land instead of one MFMA phase, does not fit: 168
  // registers at three waves per SIMD, 114 spilled)
  uint4 rpA[PCH], rdA[DCH];
  auto load_tile = [&](int tile, uint4 (&rp)[PCH], uint4 (&rd)[DCH]) {
    const int img = tile / (p.tiles_y * p.tiles_x);
    const int rem = tile - img * (p.tiles_y * p.tiles_x);
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const long long ibase = (long long)img * p.H * p.W;
    // every load is issued unconditionally from a clamped address and zeroed afterwards: a branch around a load would
    // serialise the memory round trips
#pragma unroll
    for (int k = 0; k < PCH; ++k) {
      const int iy = ty * TH + (ppos[k] >> 8) - 1, ix = tx * TW + (ppos[k] & 255) + xoff;
      const bool ok = (tid + k * NTH) < K::PCHUNKS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const long long off = ok ? (ibase + (long long)iy * p.W + ix) * p.ldi + sub : 0;
      const uint4 v = *reinterpret_cast<const uint4*>(x + off);
      rp[k] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int k = 0; k < DCH; ++k) {
      const int pix = (tid + k * NTH) / CPP;
      const int oy = ty * TH + pix / TW, ox = tx * TW + pix % TW;
      const bool ok = (tid + k * NTH) < K::DCHUNKS && oy < p.H && ox < p.W;
      const long long off = ok ? (ibase + (long long)oy * p.W + ox) * p.ldo + sub : 0;
      const uint4 v = *reinterpret_cast<const uint4*>(dy + off);
      rd[k] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto store_tile = [&](const uint4 (&rp)[PCH], const uint4 (&rd)[DCH]) {
#pragma unroll
    for (int k = 0; k < PCH; ++k) {
      const int c = tid + k * NTH;
      if (c < K::PCHUNKS) *reinterpret_cast<uint4*>(patch + (c / CPP) * PITCH + (c % CPP) * 16) = rp[k];
    }
#pragma unroll
    for (int k = 0; k < DCH; ++k) {
      const int c = tid + k * NTH;
      if (c < K::DCHUNKS) *reinterpret_cast<uint4*>(dYs + (c / CPP) * PITCH + (c % CPP) * 16) = rd[k];
    }
  };

  // tiles of one XCD (workgroup b runs on XCD b % 8; the stream count is a multiple of 8 whenever this branch is taken, so all
  // workgroups of a stream share it) are consecutive: the halos its tiles share are hits in that XCD's L2
  int tile, tend, tstep;
  if ((p.streams & 7) == 0) {
    const int per = (p.ntiles + 7) >> 3, xcd = stream & 7;
    tile = xcd * per + (stream >> 3);
    tend = min(p.ntiles, (xcd + 1) * per);
    tstep = p.streams >> 3;
  } else {
    tile = stream; tend = p.ntiles; tstep = p.streams;
  }

  // lane's byte offset inside a pixel-major tile: pixel 8 * (gq >> 1) + q, channels (gq & 1) * 16 + 4 * pp ..+3 of a 32-block
  const int frag_off = (8 * (gq >> 1) + q) * PITCH + ((gq & 1) * 16 + 4 * pp) * 2;
  const char* a_base = dYs + frag_off + nblk * 64;
  const char* b_base = patch + frag_off + (KXB ? 0 : kx * PITCH) + cblk * 64;

  // M16 fragments: lane = 16 channels (i16) x 4 pixel groups (gq): pixel 8 * gq + q, channels 4 * pp ..+3 of a 16-channel half
  const int frag16_off = (8 * gq + q) * PITCH + 4 * pp * 2;
  auto mfma_phase16 = [&]() {
    const char* a16 = dYs + frag16_off + nblk * 64;
    const char* b16 = patch + frag16_off + (KXB ? 0 : kx * PITCH) + cblk * 64;
    half8_t a[TH][2];
    half8_t bn[2] = {tr_frag<PITCH>(b16), tr_frag<PITCH>(b16 + 32)}, an[2] = {tr_frag<PITCH>(a16), tr_frag<PITCH>(a16 + 32)};
#pragma unroll
    for (int pr = 0; pr < PH; ++pr) {
      const half8_t b0 = bn[0], b1 = bn[1];
      if (pr < TH) { a[pr][0] = an[0]; a[pr][1] = an[1]; }
      if (pr + 1 < PH) {
        bn[0] = tr_frag<PITCH>(b16 + (pr + 1) * PW * PITCH);
        bn[1] = tr_frag<PITCH>(b16 + (pr + 1) * PW * PITCH + 32);
        if (pr + 1 < TH) {
          an[0] = tr_frag<PITCH>(a16 + (pr + 1) * TW * PITCH);
          an[1] = tr_frag<PITCH>(a16 + (pr + 1) * TW * PITCH + 32);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int r = pr - ky;
        if (r >= 0 && r < TH) {
#pragma unroll
          for (int ns = 0; ns < 2; ++ns) {
            acc4[ky][ns][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[r][ns], b0, acc4[ky][ns][0], 0, 0, 0);
            acc4[ky][ns][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[r][ns], b1, acc4[ky][ns][1], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto mfma_phase = [&]() {
    // software pipeline over the XH x 10 (pixel half, patch row) steps: the fragments of step s + 1 are requested before the
    // MFMAs of step s are issued (left alone the compiler orders it read, wait for everything, multiply)
    constexpr int NST = K::XH * PH;
    half8_t a[K::XH][TH];
    half8_t bn = tr_frag<PITCH>(b_base), an = tr_frag<PITCH>(a_base);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const int xh = st / PH, pr = st % PH;
      const half8_t b = bn;
      if (pr < TH) a[xh][pr] = an;
      if (st + 1 < NST) {
        const int xh1 = (st + 1) / PH, pr1 = (st + 1) % PH;
        bn = tr_frag<PITCH>(b_base + (pr1 * PW + xh1 * 16) * PITCH);
        if (pr1 < TH) an = tr_frag<PITCH>(a_base + (pr1 * TW + xh1 * 16) * PITCH);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the requests above the MFMAs: the wait before them then counts only the older reads
      // patch row pr feeds output row pr - ky through vertical tap ky
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int r = pr - ky;
        if (r >= 0 && r < TH) acc[ky] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[xh][r], b, acc[ky], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if (tile < tend) load_tile(tile, rpA, rdA);
  for (; tile < tend; tile += tstep) {
    __syncthreads();  // previous tile's fragment reads are done
    store_tile(rpA, rdA);
    __syncthreads();
    if (tile + tstep < tend) load_tile(tile + tstep, rpA, rdA);  // in flight during the MFMA phase
    if constexpr (M16) mfma_phase16(); else mfma_phase();
  }

  // ---- flush: a wave owns its (channel blocks, taps) outright -- no reduction across waves ----
  float* slab = p.ws ? p.ws + (long long)stream * p.ws_stride : nullptr;
  if constexpr (M16) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int ns = 0; ns < 2; ++ns)
#pragma unroll
        for (int cs = 0; cs < 2; ++cs)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int n = nblk * 32 + ns * 16 + 4 * (lane >> 4) + j;
            const int k = (ky * 3 + kx) * C + cblk * 32 + cs * 16 + (lane & 15);
            if (slab) slab[n * K::KTOT + k] = acc4[ky][ns][cs][j];
            else atomicAdd(p.dw + n * K::KTOT + k, acc4[ky][ns][cs][j] * p.scale);
          }
  } else {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nblk * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int k = (ky * 3 + kx) * C + cblk * 32 + (lane & 31);
        if (slab) slab[n * K::KTOT + k] = acc[ky][r];
        else atomicAdd(p.dw + n * K::KTOT + k, acc[ky][r] * p.scale);
      }
  }
}

template <int C, int TW, bool KXB, bool M16>
int launch(Params& p, const am_conv_geom* g, bool plan_only, hipStream_t s) {
  using K = Cfg<C, TW, KXB>;
  p.tiles_y = am_cdiv(g->IH, TH);
  p.tiles_x = am_cdiv(g->IW, TW);
  p.ntiles = g->B * p.tiles_y * p.tiles_x;
  const int min_tiles = am_tuning(AM_TUNE_PATCH_WGRAD_MIN_TILES);
  if (p.ntiles < min_tiles) return AM_ERR_UNSUPPORTED;  // below ~two tiles per CU the flush (one slab per stream) outweighs the single read
  // tiles that overhang the image waste their MFMAs on it: at most 20 % unless the caller forces the kernel
  if (min_tiles > 1 && (long long)g->IH * g->IW * 100 < (long long)p.tiles_y * TH * p.tiles_x * TW * 80) return AM_ERR_UNSUPPORTED;
  // persistent: one workgroup per CU; KXB: three per tile stream, 80 streams (a multiple of 8: see the XCD note in the kernel)
  const int cap = KXB ? 80 : 256;
  p.streams = p.ntiles < cap ? p.ntiles : cap;
  if (plan_only) return p.streams;
  static bool attr_done_dev[AM_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[am_current_device()];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_patch_wgrad_k<C, TW, KXB, M16>), hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES) != hipSuccess)
      return AM_ERR_LAUNCH;
    attr_done = true;
  }
  g_am_conv_variant = AM_CV_WGRAD_PATCH_C64;
  hipLaunchKernelGGL((conv_patch_wgrad_k<C, TW, KXB, M16>), dim3(p.streams * (KXB ? 3 : 1)), dim3(K::NTH), K::LDS_BYTES, s, p);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

}  // namespace apw

// Called by wgrad_dispatch (conv_gemm.hip).  plan_only: nothing is launched, the return value is the number of slabs (one per
// tile stream) the slab form writes.  ws != NULL: slab form (unscaled partials at ws + j * ws_stride); else atomics into dw.
// Returns AM_ERR_UNSUPPORTED unless the geometry is a dense 3x3 / stride 1 / pad 1 convolution with 64 or 128 input and output
// channels (canonical tap order) large enough to fill the chip.
int am_conv_patch_wgrad_f16(const am_conv_geom* g, const void* x, const void* dy, float scale, float* dw, float* ws, long long ws_stride,
                            bool plan_only, hipStream_t s) {
  using namespace apw;
  if (g->ntaps != 9 || g->pix_shift != 31 || g->N != g->krun || (g->N != 64 && g->N != 128) || g->osplit > 0) return AM_ERR_UNSUPPORTED;
  if (g->iys != 1 || g->ixs != 1 || g->oys != 1 || g->oxs != 1 || g->oy0 != 0 || g->ox0 != 0) return AM_ERR_UNSUPPORTED;
  if (g->MH != g->IH || g->MW != g->IW || g->OH != g->IH || g->OW != g->IW) return AM_ERR_UNSUPPORTED;
  if (g->ldi % 8 != 0 || g->x_coff % 8 != 0 || g->ldo % 8 != 0 || g->y_coff % 8 != 0 || g->ldi - g->x_coff < g->krun || g->ldo - g->y_coff < g->N)
    return AM_ERR_UNSUPPORTED;
  for (int t = 0; t < 9; ++t)
    if (g->dy[t] != t / 3 - 1 || g->dx[t] != t % 3 - 1) return AM_ERR_UNSUPPORTED;
  Params p;
  p.x = x; p.dy = dy; p.dw = dw; p.ws = ws; p.ws_stride = ws_stride; p.scale = scale;
  p.B = g->B; p.H = g->IH; p.W = g->IW; p.ldi = g->ldi; p.x_coff = g->x_coff; p.ldo = g->ldo; p.y_coff = g->y_coff;
  if (g->N == 64) return launch<64, 32, false, true>(p, g, plan_only, s);  // (32x32x16 on the same tiles: 145 us against 139 at B = 32)
  if (am_tuning(AM_TUNE_PATCH_WGRAD_C128) == 0) return AM_ERR_UNSUPPORTED;
  return launch<128, 16, true, false>(p, g, plan_only, s);
}
