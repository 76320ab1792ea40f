// Weight gradient of the 3-channel first layers on the space-to-depth image (policy conv 5x5/s2 = 3x3 taps, ResNet stem
// 7x7/s2 = 4x4 taps; forward: conv_s2d.hip).
//
// The generic conv_wgrad_k tiles the packed K (taps x 64) in 128-column slabs and re-reads dY once per slab: with
// M = 7.4 M output pixels and only 32-64 output channels that is 5x 472 MB of dY -- the launch is bound by those re-reads
// (530 us for the policy layer at B = 32, HBM floor ~150 us).  Here a persistent workgroup walks 8x32-pixel output tiles,
// stages each tile's dY block and its input patch ONCE in LDS and accumulates the whole [N][taps*64] gradient in
// registers (pixels are the MFMA reduction dimension: both operands are pixel-major, so fragments come from the
// ds_read_b64_tr_b16 hardware-transpose read, LDS pitches = 64 mod 128 bytes); one flush per workgroup at the end
// (LDS reduction across the waves of a channel block, then fp32 atomics).
// One workgroup of EIGHT waves per CU (two per SIMD, 128 accumulator registers each): the next tile's global loads -- dY and,
// in the fused-BatchNorm form, the conv output and the ReLU mask, 14 x 16 B per thread -- are issued before the MFMA phase and
// consumed after it (the BatchNorm-backward transform runs when the tile is stored to LDS), so ~115 KB per CU are in flight
// under the MFMAs.  (Four waves with the transform at load time ran the fused form at 2.5 TB/s: 688 us for the stem at B = 16.)
// The transform's per-channel constants sit in registers as pairs for the packed fp32 instructions and the mask mode is a template
// parameter (read from LDS per element they made the transform LDS- and VALU-bound: stem at B = 16 448 -> 288 us, 3.7 TB/s).
#include "am_common.h"

namespace amw {

constexpr int TH = 8, TW = 32;
constexpr int PW = TW + 3;       // patch columns (runs of 4 s2d pixels per tap row)
constexpr int PPITCH = 64;       // LDS bytes per patch pixel: 32 B of channels + 32 B pad (= 64 mod 128)

struct S2dWgradParams {
  const void* x;   // s2d image [B, IH, IW, 16] halves
  const void* dy;  // [B, OH, OW, ldo] halves
  float* dw;       // packed [>= N][taps*64] fp32, accumulated
  float scale;
  int B, IH, IW, OH, OW, ldo, y_coff, off0, N;
  int tiles_y, tiles_x, ntiles;
  // BNF (fused BatchNorm backward): dy is the gradient w.r.t. the BN(+ReLU) OUTPUT; the gradient w.r.t. the conv output is
  // formed while the tile is loaded, exactly as am_bn_bwd_apply would have written it (bn.hip bn_bwd_apply_k)
  const void* raw;   // conv output (BN input), same layout as dy
  const void* yout;  // BN+ReLU output (ReLU mask), NULL without ReLU
  const float* mean; const float* rstd; const float* coef;  // coef = [3][N] from am_bn_bwd_finalize
  int relu;
  // ReLU mask recomputed as the sign of raw * sg_scale + sg_shift (the layer's own normalised output: no residual on a first
  // layer) instead of reading yout: one tensor less per launch
  const float* sg_scale; const float* sg_shift;
};

typedef __attribute__((address_space(3))) s4v* lds_s4v;

__device__ __forceinline__ half8_t tr_frag(const char* lo_addr, int hi_delta) {
  const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v)(lo_addr));
  const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v)(lo_addr + hi_delta));
  const half4_t l4 = __builtin_bit_cast(half4_t, lo), h4 = __builtin_bit_cast(half4_t, hi);
  return half8_t{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
}

// NT = number of 32-channel blocks of dY (1: N <= 32, 2: N <= 64).  Wave w of 8: NT == 1 -> tile row w; NT == 2 ->
// channel block w & 1, then tap group, then row group (see TSP below).
// BNF: 0 plain, 1 fused BatchNorm backward with the ReLU mask read from yout (or no ReLU), 2 the same with the sign mask
template <int TAPS, int NT, int BNF>
__global__ __launch_bounds__(512) void conv_s2d_wgrad_k(const S2dWgradParams p) {
  constexpr int NW = 8, NTH = NW * 64;  // (two workgroups of four waves per CU: the fused forms spill, the plain form gains nothing)
  constexpr int PH = TH + TAPS - 1;
  constexpr int PATCH_PIX = PH * PW;
  constexpr int PATCH_BYTES = PATCH_PIX * PPITCH;
  constexpr int DYB = NT * 64;                      // dY bytes per pixel in LDS (32 channels per block)
  constexpr int PDY = NT == 1 ? 64 : 192;           // pitch = 64 (mod 128)
  constexpr int PCH = (PATCH_PIX * 2 + NTH - 1) / NTH;  // patch 16-byte chunks per thread
  constexpr int DCH = TH * TW * (DYB / 16) / NTH;       // dY chunks per thread
  // eight waves = NT channel blocks x TSP tap groups x row groups: the stem (4x4 taps, 64 channels) splits its taps over two
  // waves, which halves the accumulator registers (64 instead of 128: no spills next to 14 loads in flight -- a spill reload
  // costs an s_waitcnt vmcnt(0), i.e. the whole prefetch)
  constexpr int TSP = (NT == 2 && TAPS % 2 == 0) ? 2 : 1;
  constexpr int TPW = TAPS / TSP;                       // taps (patch rows) per wave
  constexpr int ROWS = TH / (NW / (NT * TSP));           // tile rows per wave
  constexpr int KTOT = TAPS * 64;

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* patch = smem;
  char* dYs = smem + ((PATCH_BYTES + 1023) / 1024) * 1024;

  const half_t* __restrict__ x = static_cast<const half_t*>(p.x);
  const half_t* __restrict__ dy = static_cast<const half_t*>(p.dy);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int gq = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  const int nblk = NT == 1 ? 0 : (wid & 1);
  const int tap0 = NT == 1 ? 0 : ((wid >> 1) % TSP) * TPW;
  const int row0 = (NT == 1 ? wid : wid / (2 * TSP)) * ROWS;

  f32x16 acc[TPW][2];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jp][r] = 0.f;

  // BNF: a thread always handles the same 8-channel chunk of a pixel (NTH is a multiple of the chunks per pixel), so its
  // per-channel constants live in registers as pairs for the packed fp32 instructions: mean, rstd, coef0..2, sign scale / shift
  // (from LDS they cost 56 dword reads per chunk, 1.5 us per tile of LDS bandwidth next to the fragment reads)
  static_assert(NTH % (DYB / 16) == 0, "chunk index must not depend on k");
  constexpr bool sign = BNF == 2;
  f32x2 k_mean[4], k_rstd[4], k_c0[4], k_c1[4], k_c2[4], k_ss[sign ? 4 : 1], k_sh[sign ? 4 : 1];
  unsigned keep[4];  // packed-half mask of the channels below N
  if constexpr (BNF) {
    const int cbase = (tid % (DYB / 16)) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = min(cbase + e, p.N - 1);
      k_mean[e >> 1][e & 1] = p.mean[c];
      k_rstd[e >> 1][e & 1] = p.rstd[c];
      k_c0[e >> 1][e & 1] = p.coef[c];
      k_c1[e >> 1][e & 1] = p.coef[p.N + c];
      k_c2[e >> 1][e & 1] = p.coef[2 * p.N + c];
      if constexpr (sign) {
        k_ss[e >> 1][e & 1] = p.sg_scale[c];
        k_sh[e >> 1][e & 1] = p.sg_shift[c];
      }
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) keep[h] = (cbase + 2 * h < p.N ? 0xffffu : 0u) | (cbase + 2 * h + 1 < p.N ? 0xffff0000u : 0u);
  }
  const half_t* __restrict__ raw = static_cast<const half_t*>(p.raw);
  const half_t* __restrict__ yout = static_cast<const half_t*>(p.yout);

  uint4 rp[PCH], rd[DCH], rx[BNF ? DCH : 1], ry[BNF == 1 ? DCH : 1];
  unsigned inside = 0;  // bit k: chunk k of the loaded tile lies inside the output image (outside: dY stays zero, no transform)
  auto load_tile = [&](int tile) {
    inside = 0;
    const int img = tile / (p.tiles_y * p.tiles_x);
    const int rem = tile - img * (p.tiles_y * p.tiles_x);
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const int iy0 = ty * TH + p.off0, ix0 = tx * TW + p.off0;
#pragma unroll
    for (int k = 0; k < PCH; ++k) {
      const int c = tid + k * NTH;
      const int pidx = c >> 1;
      const int prow = pidx / PW, pcol = pidx - prow * PW;
      const int iy = iy0 + prow, ix = ix0 + pcol;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (pidx < PATCH_PIX && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW)
        v = *reinterpret_cast<const uint4*>(x + ((long long)(img * p.IH + iy) * p.IW + ix) * 16 + (c & 1) * 8);
      rp[k] = v;
    }
#pragma unroll
    for (int k = 0; k < DCH; ++k) {
      const int c = tid + k * NTH;
      const int pix = c / (DYB / 16), cc = c - pix * (DYB / 16);
      const int oy = ty * TH + (pix >> 5), ox = tx * TW + (pix & 31);
      uint4 v = make_uint4(0u, 0u, 0u, 0u), xr = v, yr = v;
      if (oy < p.OH && ox < p.OW && cc * 8 < p.ldo - p.y_coff) {
        inside |= 1u << k;
        const long long off = ((long long)(img * p.OH + oy) * p.OW + ox) * p.ldo + p.y_coff + cc * 8;
        v = *reinterpret_cast<const uint4*>(dy + off);
        if constexpr (BNF) {
          xr = *reinterpret_cast<const uint4*>(raw + off);
          if constexpr (BNF == 1) { if (p.relu) yr = *reinterpret_cast<const uint4*>(yout + off); }
        }
      }
      rd[k] = v;
      if constexpr (BNF) rx[k] = xr;
      if constexpr (BNF == 1) ry[k] = yr;
    }
  };
  // the gradient w.r.t. the conv output from (dY, conv output, ReLU mask): bn.hip bn_bwd_apply_k's arithmetic
  auto bn_transform = [&](uint4 v, uint4 xr, uint4 yr) -> uint4 {
    const unsigned gw[4] = {v.x, v.y, v.z, v.w}, xw[4] = {xr.x, xr.y, xr.z, xr.w}, yw[4] = {yr.x, yr.y, yr.z, yr.w};
    unsigned ow[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const half2_t g2 = __builtin_bit_cast(half2_t, gw[h]), x2 = __builtin_bit_cast(half2_t, xw[h]), y2 = __builtin_bit_cast(half2_t, yw[h]);
      f32x2 dz = {(float)g2[0], (float)g2[1]};
      const f32x2 xf = {(float)x2[0], (float)x2[1]};
      if constexpr (sign) {
        const f32x2 z = xf * k_ss[h] + k_sh[h];  // (the build has -ffp-contract=off: mul, add as in bn_bwd_apply_k)
        dz[0] = z[0] > 0.f ? dz[0] : 0.f;
        dz[1] = z[1] > 0.f ? dz[1] : 0.f;
      } else if (p.relu) {
        dz[0] = (float)y2[0] > 0.f ? dz[0] : 0.f;
        dz[1] = (float)y2[1] > 0.f ? dz[1] : 0.f;
      }
      const f32x2 xhat = (xf - k_mean[h]) * k_rstd[h];
      const f32x2 o = k_c0[h] * (dz - k_c1[h] - xhat * k_c2[h]);
      const half2_t oh = {(half_t)o[0], (half_t)o[1]};
      ow[h] = __builtin_bit_cast(unsigned, oh) & keep[h];
    }
    return make_uint4(ow[0], ow[1], ow[2], ow[3]);
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int k = 0; k < PCH; ++k) {
      const int c = tid + k * NTH;
      if ((c >> 1) < PATCH_PIX) *reinterpret_cast<uint4*>(patch + (c >> 1) * PPITCH + (c & 1) * 16) = rp[k];
    }
#pragma unroll
    for (int k = 0; k < DCH; ++k) {
      const int c = tid + k * NTH;
      const int pix = c / (DYB / 16), cc = c - pix * (DYB / 16);
      uint4 v = rd[k];
      if constexpr (BNF) {
        if (inside & (1u << k)) v = bn_transform(v, rx[k], ry[BNF == 1 ? k : 0]);  // (a pixel outside the image has no gradient: the affine part of the
      }                                                                 // BatchNorm backward would otherwise leave -c0 * (c1 + xhat * c2) there)
      *reinterpret_cast<uint4*>(dYs + pix * PDY + cc * 16) = v;
    }
  };

  int tile = blockIdx.x;
  if (tile < p.ntiles) load_tile(tile);
  for (; tile < p.ntiles; tile += gridDim.x) {
    __syncthreads();  // previous tile's fragment reads are done
    store_tile();
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < p.ntiles) load_tile(next);  // in flight during the MFMA phase
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
      const int row = row0 + rr;
#pragma unroll
      for (int xh = 0; xh < 2; ++xh) {
        // 16 consecutive pixels of one tile row = the k16 of one MFMA; this lane's pixel for the transpose read
        const int px = xh * 16 + 8 * (gq >> 1) + q;
        const half8_t a = tr_frag(dYs + (row * TW + px) * PDY + nblk * 64 + ((gq & 1) * 16 + 4 * pp) * 2, 4 * PDY);
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            // 32 gradient columns = the 16 channels of run pixels 2*jp and 2*jp + 1 (the 16-lane group picks the pixel)
            const half8_t b = tr_frag(patch + ((row + tap0 + i) * PW + px + 2 * jp + (gq & 1)) * PPITCH + pp * 8, 4 * PPITCH);
            acc[i][jp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i][jp], 0, 0, 0);
          }
      }
    }
  }

  // ---- flush: waves that share a channel block add up in LDS, then one fp32 atomic per element per workgroup ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);  // [NT][32][KTOT]
  for (int e = tid; e < NT * 32 * KTOT; e += NTH) red[e] = 0.f;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        atomicAdd(red + (nblk * 32 + n) * KTOT + (tap0 + i) * 64 + jp * 32 + (lane & 31), acc[i][jp][r]);
      }
  __syncthreads();
  // every workgroup finishes at about the same time: each starts its pass over dW at a different row, so they do not all hit
  // the same addresses together
  constexpr int TOT = NT * 32 * KTOT;
  const int rot = (int)((blockIdx.x * 37u) % (unsigned)(NT * 32)) * KTOT;
  for (int e0 = tid; e0 < TOT; e0 += NTH) {
    int e = e0 + rot;
    e = e >= TOT ? e - TOT : e;
    const int n = e / KTOT;
    if (n < p.N) atomicAdd(p.dw + (size_t)n * KTOT + (e - n * KTOT), red[e] * p.scale);
  }
}

template <int TAPS, int NT, int BNF>
int launch(const S2dWgradParams& p, hipStream_t s) {
  constexpr int PH = TH + TAPS - 1;
  constexpr int PATCH = ((PH * PW * PPITCH + 1023) / 1024) * 1024;
  constexpr int DYS = TH * TW * (NT == 1 ? 64 : 192);
  constexpr int RED = NT * 32 * TAPS * 64 * 4;
  constexpr int LDS = (PATCH + DYS) > RED ? (PATCH + DYS) : RED;
  static bool attr_done_dev[AM_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[am_current_device()];
  if (LDS > 64 * 1024 && !attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_s2d_wgrad_k<TAPS, NT, BNF>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return AM_ERR_LAUNCH;
    attr_done = true;
  }
  const int grid = p.ntiles < 256 ? p.ntiles : 256;  // persistent: one 8-wave workgroup per CU
  g_am_conv_variant = AM_CV_WGRAD_S2D;
  hipLaunchKernelGGL((conv_s2d_wgrad_k<TAPS, NT, BNF>), dim3(grid), dim3(512), LDS, s, p);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

}  // namespace amw

// Called by am_conv_wgrad / am_conv_wgrad_bn (conv_gemm.hip) for first-layer (space-to-depth) geometries in f16; returns
// AM_ERR_UNSUPPORTED when the shape is not covered so the caller uses the generic kernel(s).  raw != NULL selects the fused
// BatchNorm-backward form; sg_scale / sg_shift != NULL its variant that takes the ReLU mask from the sign of the normalised output.
int am_conv_s2d_wgrad_f16(const am_conv_geom* g, const void* x, const void* dy, const void* yout, const void* raw, const float* mean,
                          const float* rstd, const float* coef, int relu, const float* sg_scale, const float* sg_shift, float scale, float* dw,
                          hipStream_t s) {
  using namespace amw;
  if (g->pix_shift != 4 || g->krun != 64 || g->ldi != 16 || g->x_coff != 0) return AM_ERR_UNSUPPORTED;
  if (g->ntaps < 3 || g->ntaps > 4 || g->N > 64) return AM_ERR_UNSUPPORTED;
  if (g->iys != 1 || g->ixs != 1 || g->oys != 1 || g->oxs != 1 || g->oy0 != 0 || g->ox0 != 0) return AM_ERR_UNSUPPORTED;
  if (g->MH != g->OH || g->MW != g->OW || (g->ldo * 2) % 16 != 0 || (g->y_coff * 2) % 16 != 0) return AM_ERR_UNSUPPORTED;
  for (int t = 0; t < g->ntaps; ++t)
    if (g->dy[t] != g->dy[0] + t || g->dx[t] != g->dy[0]) return AM_ERR_UNSUPPORTED;
  if ((long long)g->B * g->OH * g->OW < 2048) return AM_ERR_UNSUPPORTED;  // tiny problems: generic kernel
  S2dWgradParams p;
  p.x = x; p.dy = dy; p.dw = dw; p.scale = scale;
  p.B = g->B; p.IH = g->IH; p.IW = g->IW; p.OH = g->OH; p.OW = g->OW; p.ldo = g->ldo; p.y_coff = g->y_coff;
  p.off0 = g->dy[0]; p.N = g->N;
  p.tiles_y = am_cdiv(g->OH, TH);
  p.tiles_x = am_cdiv(g->OW, TW);
  p.ntiles = p.B * p.tiles_y * p.tiles_x;
  p.raw = raw; p.yout = yout; p.mean = mean; p.rstd = rstd; p.coef = coef; p.relu = relu;
  p.sg_scale = sg_scale; p.sg_shift = sg_shift;
  if (raw != nullptr) {
    if (!mean || !rstd || !coef || (relu && !yout && !sg_scale) || (sg_scale && !sg_shift)) return AM_ERR_ARG;
    if (sg_scale != nullptr) {
      if (g->N <= 32) return g->ntaps == 3 ? launch<3, 1, 2>(p, s) : launch<4, 1, 2>(p, s);
      return g->ntaps == 3 ? launch<3, 2, 2>(p, s) : launch<4, 2, 2>(p, s);
    }
    if (g->N <= 32) return g->ntaps == 3 ? launch<3, 1, 1>(p, s) : launch<4, 1, 1>(p, s);
    return g->ntaps == 3 ? launch<3, 2, 1>(p, s) : launch<4, 2, 1>(p, s);
  }
  if (g->N <= 32) return g->ntaps == 3 ? launch<3, 1, 0>(p, s) : launch<4, 1, 0>(p, s);
  return g->ntaps == 3 ? launch<3, 2, 0>(p, s) : launch<4, 2, 0>(p, s);
}
