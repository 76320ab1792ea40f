// First-layer convolutions (ResNet stem 7x7/s2 3->64, EasyBackbone 5x5/s2 3->32) on the space-to-depth(2) image,
// weights-stationary, f16.
//
// On the s2d image [B,H/2,W/2,16] the stem is a stride-1 4x4-tap conv with 16 input channels (K = 256), the policy
// conv a 3x3-tap one (K = 144).  The output is huge (64 x 360 x 640 per image) and K tiny, so the layer is bound by
// bytes, not MFMA: a persistent workgroup keeps the whole weight matrix in LDS and stages one input PATCH per 8x32
// output tile ((8+T-1) x (32+T-1) pixels x 32 B = 12 KiB, double-buffered LDS-DMA): the input is read ~1.5x instead of
// T*T times through the CU's load path, and the epilogue writes whole pixel rows.
//
// Modes (MODE template parameter)
//   0  raw conv output (+bias) and BatchNorm statistics           -- drop-in for the gather-GEMM
//   1  BatchNorm statistics only, nothing written                  -- pass 1 of the fused stem
//   2  y = relu(conv * scale[n] + shift[n])                        -- pass 2: BN(train) + ReLU applied in the epilogue
//   3  y = maxpool3x3s2(relu(conv * scale[n] + shift[n]))          -- pass 2 of the ResNet stem: only the pooled map is written
// Pass 1 + pass 2 recompute the (cheap) conv instead of writing the raw output, reading it back for the normalise
// pass and writing it again: 0.47 GB instead of 3.3 GB of HBM traffic per 32 images for the ResNet stem.
#include "am_common.h"
#include <cstdlib>

namespace ams {

__device__ __attribute__((aligned(64))) unsigned char g_zero_line[64];

constexpr int TH = 8, TW = 32;
constexpr int CB = 32;  // bytes per s2d pixel (16 halves)

struct S2dParams {
  const void* x;      // s2d image [B, IH, IW, 16] halves
  const void* w;      // packed [>=N][taps*64] halves: (i, j in 0..3, 16 ch), j >= taps zero
  void* y;            // [B, OH, OW, ldo] halves
  const float* bias;  // mode 0
  const float* scale; // mode 2
  const float* shift; // mode 2
  double* stats;      // modes 0, 1: [16][2][N]
  int B, IH, IW, OH, OW, ldo, y_coff, off0, relu;
  int tiles_y, tiles_x, ntiles, N;
};

template <int TAPS, int NT, int MODE>
__global__ __launch_bounds__(256) void conv_s2d_k(const S2dParams p) {
  constexpr int PH = TH + TAPS - 1, PW = TW + TAPS - 1;
  constexpr int PATCH_BYTES = PH * PW * CB;
  constexpr int PATCH_INST = (PATCH_BYTES + 1023) / 1024;
  constexpr int PATCH_SLOT = PATCH_INST * 1024;
  constexpr int KSTEPS = TAPS * TAPS;              // one k16 step per tap (16 channels)
  constexpr int WROW = KSTEPS * 32;                // bytes per weight row in LDS
  constexpr int WPITCH = WROW + 16;
  constexpr int NCH = NT * 32;
  constexpr int W_BYTES = ((NCH * WPITCH + 1023) / 1024) * 1024;
  constexpr int SP = NCH * 2 + 16;                 // staging row pitch
  constexpr int STG_WAVE = 32 * SP;                // one 32-pixel tile row per wave at a time

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* Wl = smem;
  char* patch0 = smem + W_BYTES;
  char* stg0 = smem + W_BYTES + 2 * PATCH_SLOT;

  const char* __restrict__ x = static_cast<const char*>(p.x);
  const char* __restrict__ w = static_cast<const char*>(p.w);
  half_t* __restrict__ y = static_cast<half_t*>(p.y);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const char* zl = reinterpret_cast<const char*>(g_zero_line);
  const int ktot_bytes = TAPS * 64 * 2;            // packed global row: taps x (4 px x 16 ch) halves

  // ---- resident weights: LDS chunk q -> (n, cc); cc = tap*2 + half, tap = i*TAPS + j; pad chunk -> zero ----
  constexpr int WCH = WPITCH / 16;                 // chunks per padded LDS row
  for (int inst = wid; inst < W_BYTES / 1024; inst += 4) {
    const int q = inst * 64 + lane;
    const int n = q / WCH, cc = q - n * WCH;
    const char* src = zl;
    if (n < NCH && cc < KSTEPS * 2) {
      const int tap = cc >> 1, i = tap / TAPS, j = tap - i * TAPS;
      src = w + (long long)n * ktot_bytes + ((i * 4 + j) * 16) * 2 + (cc & 1) * 16;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(Wl + inst * 1024), 16, 0, 0);
  }

  auto issue_patch = [&](int tile, int buf) {
    const int img = tile / (p.tiles_y * p.tiles_x);
    const int rem = tile - img * (p.tiles_y * p.tiles_x);
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const int iy0 = ty * TH + p.off0, ix0 = tx * TW + p.off0;
    char* dst = patch0 + buf * PATCH_SLOT;
    for (int inst = wid; inst < PATCH_INST; inst += 4) {
      const int q = inst * 64 + lane;
      const int pidx = q >> 1, cpos = q & 1;
      const int c = cpos ^ ((pidx >> 3) & 1);
      const int prow = pidx / PW, pcol = pidx - prow * PW;
      const int iy = iy0 + prow, ix = ix0 + pcol;
      const bool ok = q < PATCH_BYTES / 16 && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      const char* src = ok ? x + ((long long)(img * p.IH + iy) * p.IW + ix) * CB + c * 16 : zl;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + inst * 1024), 16, 0, 0);
    }
  };

  int tile = blockIdx.x;
  if (tile < p.ntiles) issue_patch(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  float st_s[NT], st_q[NT], sc[NT], sh[NT], bv[NT];
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  f32x8 sv[NT], qv[NT];  // MODE 1: per-register-pair partial sums (packed adds / FMAs per tile), folded once at the end
#pragma unroll
  for (int tn = 0; tn < NT; ++tn) {
#pragma unroll
    for (int r = 0; r < 8; ++r) { sv[tn][r] = 0.f; qv[tn][r] = 0.f; }
    st_s[tn] = st_q[tn] = 0.f;
    const int col = tn * 32 + (lane & 31);
    sc[tn] = (MODE == 2 && col < p.N) ? p.scale[col] : 1.f;
    sh[tn] = (MODE == 2 && col < p.N) ? p.shift[col] : 0.f;
    bv[tn] = (MODE == 0 && p.bias && col < p.N) ? p.bias[col] : 0.f;
  }
  const int rx = lane & 31, kg = lane >> 5;
  int buf = 0;
  for (; tile < p.ntiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    if (next < p.ntiles) issue_patch(next, buf ^ 1);
    const char* pt = patch0 + buf * PATCH_SLOT;

    // KSTEPS k16 steps (one per tap), software-pipelined by hand: the fragment reads of step s are issued while the MFMAs
    // of step s-2 run; the counted wait is the builtin so that hipcc's waitcnt pass sees it (see conv_patch.hip).
    f32x16 acc[2][NT];
    half8_t fa[3][2], fb[3][NT];
#pragma unroll
    for (int s = 0; s < KSTEPS + 2; ++s) {
      if (s >= 2) {
        if (s == KSTEPS + 1) __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
        else if (NT == 2) __builtin_amdgcn_s_waitcnt(0xC47F);            // lgkmcnt(4): the reads of step s-1 stay in flight
        else __builtin_amdgcn_s_waitcnt(0xC37F);                         // lgkmcnt(3)
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s < KSTEPS) {
        const int i = s / TAPS, j = s - i * TAPS;
        const int pid0 = (2 * wid + i) * PW + rx + j;
        const int pid1 = pid0 + PW;
        fa[s % 3][0] = *reinterpret_cast<const half8_t*>(pt + pid0 * CB + ((kg ^ ((pid0 >> 3) & 1)) << 4));
        fa[s % 3][1] = *reinterpret_cast<const half8_t*>(pt + pid1 * CB + ((kg ^ ((pid1 >> 3) & 1)) << 4));
        const char* bb = Wl + (lane & 31) * WPITCH + s * 32 + kg * 16;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) fb[s % 3][tn] = *reinterpret_cast<const half8_t*>(bb + tn * 32 * WPITCH);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s >= 2) {
        const int c = (s - 2) % 3;
        if (s == 2) {  // first step starts from the constant zero: no per-tile accumulator clears
          const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int tn = 0; tn < NT; ++tn) {
            acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[c][0], fb[c][tn], z, 0, 0, 0);
            acc[1][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[c][1], fb[c][tn], z, 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int tn = 0; tn < NT; ++tn) {
            acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[c][0], fb[c][tn], acc[0][tn], 0, 0, 0);
            acc[1][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[c][1], fb[c][tn], acc[1][tn], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    // next patch has had the MFMA phase to land; drain the DMA before any global store (see conv_patch.hip)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    buf ^= 1;

    const int img = tile / (p.tiles_y * p.tiles_x);
    const int rem = tile - img * (p.tiles_y * p.tiles_x);
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    char* stg = stg0 + wid * STG_WAVE;
    if (MODE == 1) {
      // statistics only: whole-vector adds / FMAs into per-register partial sums; pixels outside the image (edge tiles
      // only) are zeroed first so they add nothing
      if (ty * TH + TH > p.OH || tx * TW + TW > p.OW) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          const bool rowok = ty * TH + 2 * wid + tm < p.OH;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool ok = rowok && tx * TW + (r & 3) + 8 * (r >> 2) + 4 * kg < p.OW;
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) acc[tm][tn][r] = ok ? acc[tm][tn][r] : 0.f;
          }
        }
      }
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          const f32x8 lo = __builtin_shufflevector(acc[tm][tn], acc[tm][tn], 0, 1, 2, 3, 4, 5, 6, 7);
          const f32x8 hi = __builtin_shufflevector(acc[tm][tn], acc[tm][tn], 8, 9, 10, 11, 12, 13, 14, 15);
          sv[tn] += lo;
          sv[tn] += hi;
          qv[tn] = __builtin_elementwise_fma(lo, lo, qv[tn]);
          qv[tn] = __builtin_elementwise_fma(hi, hi, qv[tn]);
        }
      continue;
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int oy = ty * TH + 2 * wid + tm;
      const bool rowok = oy < p.OH;
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = (r & 3) + 8 * (r >> 2) + 4 * kg;
          const float a = acc[tm][tn][r];
          if (MODE != 2 && rowok && tx * TW + px < p.OW) {
            st_s[tn] += a;
            st_q[tn] += a * a;
          }
          if (MODE != 1) {
            float v = MODE == 2 ? a * sc[tn] + sh[tn] : a + bv[tn];
            if (MODE == 2 || p.relu) v = fmaxf(v, 0.f);
            *reinterpret_cast<half_t*>(stg + px * SP + (tn * 32 + (lane & 31)) * 2) = (half_t)v;
          }
        }
      }
      if (MODE != 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        constexpr int CPR = NCH / 8;  // 16-byte chunks per pixel row
        const int ncols = (p.N + 7) & ~7;
#pragma unroll
        for (int it = 0; it < 32 * CPR / 64; ++it) {
          const int q = it * 64 + lane;
          const int px = q / CPR, cc = q - px * CPR;
          const int ox = tx * TW + px;
          if (rowok && ox < p.OW && cc * 8 < ncols)
            *reinterpret_cast<uint4*>(y + ((long long)(img * p.OH + oy) * p.OW + ox) * p.ldo + p.y_coff + cc * 8) =
                *reinterpret_cast<const uint4*>(stg + px * SP + cc * 16);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }

  if (MODE != 2 && p.stats != nullptr) {
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      if (MODE == 1) {
#pragma unroll
        for (int r = 0; r < 8; ++r) { st_s[tn] += sv[tn][r]; st_q[tn] += qv[tn][r]; }
      }
      st_s[tn] += __shfl_xor(st_s[tn], 32, 64);
      st_q[tn] += __shfl_xor(st_q[tn], 32, 64);
    }
    __syncthreads();
    float* part = reinterpret_cast<float*>(stg0);  // [4 waves][NCH][2]
    if (lane < 32) {
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        part[(wid * NCH + tn * 32 + lane) * 2 + 0] = st_s[tn];
        part[(wid * NCH + tn * 32 + lane) * 2 + 1] = st_q[tn];
      }
    }
    __syncthreads();
    if (tid < NCH && tid < p.N) {
      double s = 0.0, q = 0.0;
      for (int a = 0; a < 4; ++a) {
        s += (double)part[(a * NCH + tid) * 2 + 0];
        q += (double)part[(a * NCH + tid) * 2 + 1];
      }
      double* st = p.stats + (size_t)(blockIdx.x % AM_STATS_REPLICAS) * 2 * p.N;
      atomicAdd(st + tid, s);
      atomicAdd(st + p.N + tid, q);
    }
  }
}

template <int TAPS, int NT, int MODE>
int launch_s2d(const S2dParams& p, hipStream_t s) {
  constexpr int PH = TH + TAPS - 1, PW = TW + TAPS - 1;
  constexpr int PATCH_SLOT = ((PH * PW * CB + 1023) / 1024) * 1024;
  constexpr int WPITCH = TAPS * TAPS * 32 + 16;
  constexpr int NCH = NT * 32;
  constexpr int W_BYTES = ((NCH * WPITCH + 1023) / 1024) * 1024;
  constexpr int STG = 4 * 32 * (NCH * 2 + 16);
  constexpr int LDS = W_BYTES + 2 * PATCH_SLOT + (STG > 4 * NCH * 8 ? STG : 4 * NCH * 8);
  static bool attr_done = false;
  if (LDS > 64 * 1024 && !attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_s2d_k<TAPS, NT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return AM_ERR_LAUNCH;
    attr_done = true;
  }
  const int per_cu = LDS <= 80 * 1024 ? 2 : 1;
  const int grid = p.ntiles < 256 * per_cu ? p.ntiles : 256 * per_cu;
  g_am_conv_variant = AM_CV_S2D;
  hipLaunchKernelGGL((conv_s2d_k<TAPS, NT, MODE>), dim3(grid), dim3(256), LDS, s, p);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

// ---- mode 3: conv -> BN(scale, shift) -> ReLU -> MaxPool2d(3, 2, 1), nothing but the pooled map is written ----------
// 8 waves, conv tile 16 x 32 (wave w: rows 2w, 2w+1) = the 15 x 31 conv outputs behind a 7 x 15 pooled tile (+1 spare
// row / column); neighbouring tiles recompute the one-pixel overlap (27 % extra MFMA work on a layer that is far from
// MFMA-bound) instead of exchanging halos.  Post-ReLU values are >= 0 and every pooling window holds at least one
// in-image conv output, so out-of-image positions are staged as 0 (equivalent to max-pool's -inf padding).
constexpr int PTH = 7, PTW = 15, CTH = 16, CTW = 32;

template <int TAPS>
__global__ __launch_bounds__(512) void conv_s2d_pool_k(const S2dParams p, int POH, int POW, int ptiles_y, int ptiles_x) {
  constexpr int NT = 2, NCH = 64;
  constexpr int PH = CTH + TAPS - 1, PW = CTW + TAPS - 1;
  constexpr int PATCH_BYTES = PH * PW * CB;
  constexpr int PATCH_INST = (PATCH_BYTES + 1023) / 1024;
  constexpr int PATCH_SLOT = PATCH_INST * 1024;
  constexpr int KSTEPS = TAPS * TAPS;
  constexpr int WPITCH = KSTEPS * 32 + 16;
  constexpr int WCH = WPITCH / 16;
  constexpr int W_BYTES = ((NCH * WPITCH + 1023) / 1024) * 1024;
  constexpr int SPX = NCH * 2;  // staging: [16 rows][32 px][64 ch] halves, 128 B per pixel

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* Wl = smem;
  char* patch0 = smem + W_BYTES;
  char* stg = smem + W_BYTES + 2 * PATCH_SLOT;

  const char* __restrict__ x = static_cast<const char*>(p.x);
  const char* __restrict__ w = static_cast<const char*>(p.w);
  half_t* __restrict__ y = static_cast<half_t*>(p.y);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const char* zl = reinterpret_cast<const char*>(g_zero_line);
  const int ktot_bytes = TAPS * 64 * 2;

  for (int inst = wid; inst < W_BYTES / 1024; inst += 8) {
    const int q = inst * 64 + lane;
    const int n = q / WCH, cc = q - n * WCH;
    const char* src = zl;
    if (n < NCH && cc < KSTEPS * 2) {
      const int tap = cc >> 1, i = tap / TAPS, j = tap - i * TAPS;
      src = w + (long long)n * ktot_bytes + ((i * 4 + j) * 16) * 2 + (cc & 1) * 16;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(Wl + inst * 1024), 16, 0, 0);
  }

  auto tile_origin = [&](int tile, int& img, int& pty, int& ptx) {
    img = tile / (ptiles_y * ptiles_x);
    const int rem = tile - img * (ptiles_y * ptiles_x);
    pty = rem / ptiles_x;
    ptx = rem - pty * ptiles_x;
  };
  auto issue_patch = [&](int tile, int buf) {
    int img, pty, ptx;
    tile_origin(tile, img, pty, ptx);
    const int iy0 = 2 * PTH * pty - 1 + p.off0, ix0 = 2 * PTW * ptx - 1 + p.off0;
    char* dst = patch0 + buf * PATCH_SLOT;
    for (int inst = wid; inst < PATCH_INST; inst += 8) {
      const int q = inst * 64 + lane;
      const int pidx = q >> 1, cpos = q & 1;
      const int c = cpos ^ ((pidx >> 3) & 1);
      const int prow = pidx / PW, pcol = pidx - prow * PW;
      const int iy = iy0 + prow, ix = ix0 + pcol;
      const bool ok = q < PATCH_BYTES / 16 && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      const char* src = ok ? x + ((long long)(img * p.IH + iy) * p.IW + ix) * CB + c * 16 : zl;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + inst * 1024), 16, 0, 0);
    }
  };

  int tile = blockIdx.x;
  if (tile < p.ntiles) issue_patch(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  float sc[NT], sh[NT];
#pragma unroll
  for (int tn = 0; tn < NT; ++tn) {
    sc[tn] = p.scale[tn * 32 + (lane & 31)];
    sh[tn] = p.shift[tn * 32 + (lane & 31)];
  }
  const int rx = lane & 31, kg = lane >> 5;
  int buf = 0;
  for (; tile < p.ntiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    if (next < p.ntiles) issue_patch(next, buf ^ 1);
    const char* pt = patch0 + buf * PATCH_SLOT;
    f32x16 acc[2][NT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#pragma unroll
    for (int i = 0; i < TAPS; ++i) {
#pragma unroll
      for (int j = 0; j < TAPS; ++j) {
        const int pid0 = (2 * wid + i) * PW + rx + j;
        const int pid1 = pid0 + PW;
        const half8_t fa0 = *reinterpret_cast<const half8_t*>(pt + pid0 * CB + ((kg ^ ((pid0 >> 3) & 1)) << 4));
        const half8_t fa1 = *reinterpret_cast<const half8_t*>(pt + pid1 * CB + ((kg ^ ((pid1 >> 3) & 1)) << 4));
        const char* bb = Wl + (lane & 31) * WPITCH + (i * TAPS + j) * 32 + kg * 16;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) {
          const half8_t fb = *reinterpret_cast<const half8_t*>(bb + tn * 32 * WPITCH);
          acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0, fb, acc[0][tn], 0, 0, 0);
          acc[1][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1, fb, acc[1][tn], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // next patch landed; every wave is done with this patch AND with the previous tile's staging reads
    buf ^= 1;

    int img, pty, ptx;
    tile_origin(tile, img, pty, ptx);
    const int cy0 = 2 * PTH * pty - 1, cx0 = 2 * PTW * ptx - 1;  // conv-output coordinates of tile row/col 0
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int ry = 2 * wid + tm;
      const bool rowok = (unsigned)(cy0 + ry) < (unsigned)p.OH;
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = (r & 3) + 8 * (r >> 2) + 4 * kg;
          float v = fmaxf(acc[tm][tn][r] * sc[tn] + sh[tn], 0.f);
          if (!(rowok && (unsigned)(cx0 + px) < (unsigned)p.OW)) v = 0.f;
          *reinterpret_cast<half_t*>(stg + (ry * CTW + px) * SPX + (tn * 32 + (lane & 31)) * 2) = (half_t)v;
        }
    }
    __syncthreads();
    // pooled tile: PTH x PTW pixels x 8 chunks of 8 channels; pooled (ppy, ppx) <- conv tile rows 2ppy..2ppy+2, cols 2ppx..2ppx+2
    for (int e = tid; e < PTH * PTW * 8; e += 512) {
      const int c8 = e & 7, pp = e >> 3;
      const int ppy = pp / PTW, ppx = pp - ppy * PTW;
      const int oy = pty * PTH + ppy, ox = ptx * PTW + ppx;
      if (oy >= POH || ox >= POW) continue;
      half8_t m = *reinterpret_cast<const half8_t*>(stg + ((2 * ppy) * CTW + 2 * ppx) * SPX + c8 * 16);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          if (dy == 0 && dx == 0) continue;
          const half8_t t = *reinterpret_cast<const half8_t*>(stg + ((2 * ppy + dy) * CTW + 2 * ppx + dx) * SPX + c8 * 16);
#pragma unroll
          for (int k = 0; k < 8; ++k) m[k] = t[k] > m[k] ? t[k] : m[k];
        }
      *reinterpret_cast<half8_t*>(y + ((long long)(img * POH + oy) * POW + ox) * p.ldo + p.y_coff + c8 * 8) = m;
    }
    // the barrier after the next tile's MFMA phase orders these staging reads before the next staging writes
  }
}

template <int TAPS>
int launch_s2d_pool(const S2dParams& p0, int POH, int POW, hipStream_t s) {
  constexpr int PH = CTH + TAPS - 1, PW = CTW + TAPS - 1;
  constexpr int PATCH_SLOT = ((PH * PW * CB + 1023) / 1024) * 1024;
  constexpr int WPITCH = TAPS * TAPS * 32 + 16;
  constexpr int W_BYTES = ((64 * WPITCH + 1023) / 1024) * 1024;
  constexpr int LDS = W_BYTES + 2 * PATCH_SLOT + CTH * CTW * 128;
  S2dParams p = p0;
  const int pty = am_cdiv(POH, PTH), ptx = am_cdiv(POW, PTW);
  p.ntiles = p.B * pty * ptx;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_s2d_pool_k<TAPS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return AM_ERR_LAUNCH;
    attr_done = true;
  }
  const int grid = p.ntiles < 256 ? p.ntiles : 256;
  g_am_conv_variant = AM_CV_S2D_POOL;
  hipLaunchKernelGGL((conv_s2d_pool_k<TAPS>), dim3(grid), dim3(512), LDS, s, p, POH, POW, pty, ptx);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

template <int MODE>
int dispatch_s2d(const S2dParams& p, int taps, hipStream_t s) {
  if (taps == 4 && p.N > 32 && p.N <= 64) return launch_s2d<4, 2, MODE>(p, s);
  if (taps == 4 && p.N <= 32) return launch_s2d<4, 1, MODE>(p, s);
  if (taps == 3 && p.N > 32 && p.N <= 64) return launch_s2d<3, 2, MODE>(p, s);
  if (taps == 3 && p.N <= 32) return launch_s2d<3, 1, MODE>(p, s);
  return AM_ERR_UNSUPPORTED;
}

}  // namespace ams

// mode 0/1/2 as in the file header.  Geometry must be the first-layer s2d form produced by the host (pix_shift 4,
// krun 64, dy = off0 + i, dx = off0).  Returns AM_ERR_UNSUPPORTED otherwise (caller falls back to the gather-GEMM).
int am_conv_s2d_f16(const am_conv_geom* g, int mode, const void* x, const void* w, const float* bias, const float* scale,
                    const float* shift, int relu, void* y, double* stats, hipStream_t s) {
  using namespace ams;
  if (g->pix_shift != 4 || g->krun != 64 || g->ldi != 16 || g->x_coff != 0) return AM_ERR_UNSUPPORTED;
  if (g->ntaps < 3 || g->ntaps > 4 || g->N > 64) return AM_ERR_UNSUPPORTED;
  if (g->iys != 1 || g->ixs != 1 || g->oys != 1 || g->oxs != 1 || g->oy0 != 0 || g->ox0 != 0) return AM_ERR_UNSUPPORTED;
  if (g->MH != g->OH || g->MW != g->OW) return AM_ERR_UNSUPPORTED;
  for (int t = 0; t < g->ntaps; ++t)
    if (g->dy[t] != g->dy[0] + t || g->dx[t] != g->dy[0]) return AM_ERR_UNSUPPORTED;
  if ((long long)g->B * g->OH * g->OW < 64 * 1024) return AM_ERR_UNSUPPORTED;  // small problems: gather-GEMM
  S2dParams p;
  p.x = x; p.w = w; p.y = y; p.bias = bias; p.scale = scale; p.shift = shift; p.stats = stats;
  p.B = g->B; p.IH = g->IH; p.IW = g->IW; p.OH = g->OH; p.OW = g->OW; p.ldo = g->ldo; p.y_coff = g->y_coff;
  p.off0 = g->dy[0]; p.relu = relu; p.N = g->N;
  p.tiles_y = am_cdiv(g->OH, TH);
  p.tiles_x = am_cdiv(g->OW, TW);
  p.ntiles = p.B * p.tiles_y * p.tiles_x;
  if (mode == 3) {
    // y is the max-pooled map [B, POH, POW, ldo], POH = (OH-1)/2+1
    if (g->ntaps != 4 || g->N != 64) return AM_ERR_UNSUPPORTED;
    return launch_s2d_pool<4>(p, (g->OH - 1) / 2 + 1, (g->OW - 1) / 2 + 1, s);
  }
  if (mode == 0) return dispatch_s2d<0>(p, g->ntaps, s);
  if (mode == 1) return dispatch_s2d<1>(p, g->ntaps, s);
  if (mode == 2) return dispatch_s2d<2>(p, g->ntaps, s);
  return AM_ERR_ARG;
}
