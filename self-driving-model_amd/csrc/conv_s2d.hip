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
//   4  y = maxpool3x3s2(sgn(gamma[n]) * conv) + statistics of sgn * conv  -- the frozen train-mode stem in ONE pass (round 3):
//      max-pool commutes with the monotone map v -> f16(relu(v * scale + shift)), so the normalisation moves behind the pool
//      (into the consumers' input staging); a channel with gamma < 0 is pooled on the negated conv output (min instead of max).
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
  unsigned x_bytes;
};

typedef __attribute__((address_space(3))) void* lds_ptr;
constexpr unsigned OOB = 0x80000000u;  // at or above every image's byte size (checked by the host entry)

// 16 bytes per lane from the raw buffer (base, bytes) into LDS at dst + lane*16; offsets at or past `bytes` deliver zeros.
__device__ __forceinline__ void buffer_to_lds16(const void* base, unsigned bytes, char* dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000),
                                           (lds_ptr)dst, 16, voff, 0, 0, 0);
}

// Input patch loader: PH x PW pixels x 32 B as 1 KiB LDS-DMA pieces dealt round-robin to the NWAVES waves.  Which patch
// pixel and 16-byte half a lane fetches never changes, so its row / column / byte offset inside the patch are computed
// once; per tile a piece costs two adds, two compares and a select (pixels outside the image ask for an out-of-range
// offset: the buffer load writes the zero padding).
template <int PH, int PW, int NWAVES>
struct PatchLoader {
  static constexpr int PATCH_BYTES = PH * PW * CB;
  static constexpr int PATCH_INST = (PATCH_BYTES + 1023) / 1024;
  static constexpr int NI = (PATCH_INST + NWAVES - 1) / NWAVES;
  int prow[NI], pcol[NI];
  unsigned loff[NI];
  __device__ __forceinline__ void init(int wid, int lane, int IW) {
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int q = (wid + k * NWAVES) * 64 + lane;
      const int pidx = q >> 1, cpos = q & 1;
      const int c = cpos ^ ((pidx >> 3) & 1);
      const int pr = pidx / PW, pc = pidx - pr * PW;
      prow[k] = q < PATCH_BYTES / 16 ? pr : (1 << 28);  // past the patch: never inside the image
      pcol[k] = pc;
      loff[k] = (unsigned)((pr * IW + pc) * CB + c * 16);
    }
  }
  // patch whose pixel (0, 0) is image pixel (iy0, ix0) of image img
  __device__ __forceinline__ void issue(const void* x, unsigned x_bytes, int IH, int IW, int img, int iy0, int ix0, char* dst,
                                        int wid) const {
    const int base = ((img * IH + iy0) * IW + ix0) * CB;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int inst = wid + k * NWAVES;
      if (inst < PATCH_INST) {
        const bool ok = (unsigned)(iy0 + prow[k]) < (unsigned)IH && (unsigned)(ix0 + pcol[k]) < (unsigned)IW;
        buffer_to_lds16(x, x_bytes, dst + inst * 1024, ok ? (unsigned)base + loff[k] : OOB);
      }
    }
  }
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
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
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

  PatchLoader<PH, PW, 4> loader;
  loader.init(wid, lane, p.IW);
  auto issue_patch = [&](int tile, int buf) {
    const int img = tile / (p.tiles_y * p.tiles_x);
    const int rem = tile - img * (p.tiles_y * p.tiles_x);
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    loader.issue(p.x, p.x_bytes, p.IH, p.IW, img, ty * TH + p.off0, tx * TW + p.off0, patch0 + buf * PATCH_SLOT, wid);
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
  static bool attr_done_dev[AM_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[am_current_device()];
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
// in-image conv output, so out-of-image positions count as 0 (equivalent to max-pool's -inf padding).
//
// The MFMAs run transposed (A = weights, B = pixels): a lane owns ONE conv pixel (column lane & 31 of its wave's two
// rows) and, per accumulator register quad, four consecutive channels.  The whole epilogue therefore stays in
// registers as packed halves: BN + ReLU (packed f32 mul/add, v_cvt_pk_f16_f32, v_pk_max_f16), the vertical maximum of
// the wave's own two rows, ONE row exchanged with the next wave through LDS (pooled row w needs conv rows 2w, 2w+1 and
// 2w+2, the last one is wave w+1's first), and the horizontal 3-window as two whole-wave DPP shifts.  Only the pooled
// row goes through a small per-wave staging strip to leave as 16-byte stores.  One workgroup barrier per tile (shared
// with the patch ring); the first version staged the whole 16 x 32 x 64 map and re-read it nine times.
constexpr int PTH = 7, PTW = 15, CTH = 16, CTW = 32;
constexpr int XPITCH = 144;                      // bytes per staged pixel (64 channels + 16): conflict-free 8-byte accesses
constexpr int XROW = CTW * XPITCH;               // one exchanged conv row
constexpr int OROW = ((PTW * XPITCH + 255) / 256) * 256;  // one pooled row

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// lane i <- lane i + 1 (whole-wave shift, DPP wave_shl:1; lane 63 gets 0)
__device__ __forceinline__ unsigned wave_next(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned pkmax(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(half2_t, a), __builtin_bit_cast(half2_t, b)));
}

// K-steps [S0, S1) of one conv tile (one k16 step per tap), transposed MFMAs (A = weights, B = pixels), software-pipelined
// like conv_s2d_k: the fragment reads of step s are issued while the MFMAs of step s-2 run.  S0 == 0 starts from zero.
template <int TAPS, int S0, int S1>
__device__ __forceinline__ void pool_mfma(const char* pt, const char* Wl, int wid, int lane, f32x16 (&acc)[2][2]) {
  constexpr int NT = 2;
  constexpr int PW = CTW + TAPS - 1;
  constexpr int WPITCH = TAPS * TAPS * 32 + 16;
  const int rx = lane & 31, kg = lane >> 5;
  half8_t fa[3][2], fb[3][NT];
#pragma unroll
  for (int s = S0; s < S1 + 2; ++s) {
    if (s >= S0 + 2) {
      if (s == S1 + 1) __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
      else __builtin_amdgcn_s_waitcnt(0xC47F);              // lgkmcnt(4): the reads of step s-1 stay in flight
    }
    __builtin_amdgcn_sched_barrier(0);
    if (s < S1) {
      const int i = s / TAPS, j = s - i * TAPS;
      const int pid0 = (2 * wid + i) * PW + rx + j;
      const int pid1 = pid0 + PW;
      fa[s % 3][0] = *reinterpret_cast<const half8_t*>(pt + pid0 * CB + ((kg ^ ((pid0 >> 3) & 1)) << 4));
      fa[s % 3][1] = *reinterpret_cast<const half8_t*>(pt + pid1 * CB + ((kg ^ ((pid1 >> 3) & 1)) << 4));
      const char* bb = Wl + rx * WPITCH + s * 32 + kg * 16;
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) fb[s % 3][tn] = *reinterpret_cast<const half8_t*>(bb + tn * 32 * WPITCH);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (s >= S0 + 2) {
      const int c = (s - 2) % 3;
      if (s == 2) {  // S0 == 0: the first step starts from the constant zero, no accumulator clears
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) {
          acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[c][tn], fa[c][0], z, 0, 0, 0);
          acc[1][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[c][tn], fa[c][1], z, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) {
          acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[c][tn], fa[c][0], acc[0][tn], 0, 0, 0);
          acc[1][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[c][tn], fa[c][1], acc[1][tn], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// RAW (mode 4): no BatchNorm / ReLU in the epilogue -- the f16-rounded conv output is pooled as it is, with the weight rows of
// channels whose gamma is negative NEGATED in LDS once per workgroup (conv' = sgn * conv bit for bit: every product and every
// partial sum just changes sign), and the BatchNorm statistics of conv' are accumulated over the conv pixels this tile OWNS
// (local rows 1..14, columns 1..30: row / column 0 belong to the previous tile, 15 / 31 are the spare ones) from the fp32
// accumulators, as the statistics pass did.  Out-of-image conv positions count as -inf (raw values have both signs).
template <int TAPS, bool RAW>
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

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* Wl = smem;
  char* patch0 = smem + W_BYTES;
  char* xbuf = smem + W_BYTES + 2 * PATCH_SLOT;  // [2 tile parities][waves 1..7][32 px][XPITCH]: first conv row of each wave
  char* obuf = xbuf + 2 * 7 * XROW;              // [waves 0..6][OROW]: pooled row staging, private to the wave
  float* bnc = reinterpret_cast<float*>(obuf + 7 * OROW);  // [scale 64][shift 64]

  const char* __restrict__ x = static_cast<const char*>(p.x);
  const char* __restrict__ w = static_cast<const char*>(p.w);
  half_t* __restrict__ y = static_cast<half_t*>(p.y);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* zl = reinterpret_cast<const char*>(g_zero_line);
  const int ktot_bytes = TAPS * 64 * 2;
  if (!RAW && tid < 2 * NCH) bnc[tid] = tid < NCH ? p.scale[tid] : p.shift[tid - NCH];

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
  PatchLoader<PH, PW, 8> loader;
  loader.init(wid, lane, p.IW);
  auto issue_patch = [&](int tile, int buf) {
    int img, pty, ptx;
    tile_origin(tile, img, pty, ptx);
    loader.issue(p.x, p.x_bytes, p.IH, p.IW, img, 2 * PTH * pty - 1 + p.off0, 2 * PTW * ptx - 1 + p.off0, patch0 + buf * PATCH_SLOT, wid);
  };

  int tile = blockIdx.x;
  if (tile < p.ntiles) issue_patch(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (RAW) {  // negate the resident weight rows of the channels with gamma < 0 (p.scale = gamma here)
    constexpr int DW_ROW = KSTEPS * 8;  // dwords of weights per row (the 16-byte pad is left alone)
    for (int i = tid; i < NCH * DW_ROW; i += 512) {
      const int n = i / DW_ROW, d = i - n * DW_ROW;
      if (p.scale[n] < 0.f) {
        unsigned* wp = reinterpret_cast<unsigned*>(Wl + n * WPITCH) + d;
        *wp ^= 0x80008000u;
      }
    }
    __syncthreads();
  }

  const int rx = lane & 31, kg = lane >> 5;
  f32x4 ssum[NT][4], ssq[NT][4];  // RAW: this lane's statistics partials (2 x 16 channels of its pixel column), all tiles
#pragma unroll
  for (int tn = 0; tn < NT; ++tn)
#pragma unroll
    for (int q = 0; q < 4; ++q) { ssum[tn][q] = f32x4{0.f, 0.f, 0.f, 0.f}; ssq[tn][q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  // Second half of a tile's epilogue (needs the row exchanged at the barrier): vertical window with the next wave's
  // first row, horizontal window by DPP, pooled row out through the wave's staging strip.
  auto pool_out = [&](int t, int tbuf, const half4_t (&v)[NT][4]) {
    if (wid >= 7) return;
    int img, pty, ptx;
    tile_origin(t, img, pty, ptx);
    const char* xn = xbuf + (tbuf * 7 + wid) * XROW + rx * XPITCH + 8 * kg;  // wave wid+1's first row = conv row 2*wid + 2
    char* ow = obuf + wid * OROW;
    u32x2 m[NT][4];
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const half4_t nb = *reinterpret_cast<const half4_t*>(xn + tn * 64 + q * 16);
        m[tn][q] = __builtin_bit_cast(u32x2, __builtin_elementwise_max(v[tn][q], nb));  // vertical 3-window
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          unsigned u = m[tn][q][e];
          u = pkmax(u, wave_next(u));  // columns rx, rx+1
          u = pkmax(u, wave_next(u));  // columns rx .. rx+2 (rx even: pooled column rx/2)
          m[tn][q][e] = u;
        }
      }
    if (!(rx & 1) && rx < 2 * PTW) {
      char* od = ow + (rx >> 1) * XPITCH + 8 * kg;
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x2*>(od + tn * 64 + q * 16) = m[tn][q];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int oy = pty * PTH + wid;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int e = it * 64 + lane;
      const int ppx = e >> 3, c8 = e & 7;
      const int ox = ptx * PTW + ppx;
      if (ppx < PTW && oy < POH && ox < POW)
        *reinterpret_cast<uint4*>(y + ((long long)(img * POH + oy) * POW + ox) * p.ldo + p.y_coff + c8 * 8) =
            *reinterpret_cast<const uint4*>(ow + ppx * XPITCH + c8 * 16);
    }
  };

  // Waves 4..7 share their SIMDs with waves 0..3.  All eight run the same per-tile program between two barriers, so
  // left alone the partners would reach their MFMA phases and their (VALU-only) epilogues together.  The previous tile's
  // pool_out is therefore placed differently: waves 0..3 run it right after the barrier, before their MFMAs; waves 4..7
  // start their MFMAs at once and run it between the two halves of the tap loop.
  const bool late = !RAW && wid >= 4;  // (RAW: the 64 statistics registers leave no room for pool_out's temporaries inside the MFMA phase: 6 spills)
  int buf = 0, ptile = -1;
  half4_t v[NT][4];
  for (; tile < p.ntiles; tile += gridDim.x) {
    if (!late && ptile >= 0) pool_out(ptile, buf ^ 1, v);
    const int next = tile + gridDim.x;
    if (next < p.ntiles) issue_patch(next, buf ^ 1);
    const char* pt = patch0 + buf * PATCH_SLOT;
    f32x16 acc[2][NT];
    pool_mfma<TAPS, 0, KSTEPS / 2>(pt, Wl, wid, lane, acc);
    if (late && ptile >= 0) pool_out(ptile, buf ^ 1, v);
    pool_mfma<TAPS, KSTEPS / 2, KSTEPS>(pt, Wl, wid, lane, acc);

    int img, pty, ptx;
    tile_origin(tile, img, pty, ptx);
    const int cy0 = 2 * PTH * pty - 1, cx0 = 2 * PTW * ptx - 1;  // conv-output coordinates of tile row/col 0
    // BN + ReLU on packed registers; h0 = the wave's first row, v = max of its two rows.  Per (tn, q): channels
    // tn*32 + 8q + 4kg + (0..3) of column rx.
    half4_t h0[NT][4];
    const half_t zv = RAW ? (half_t)(-__builtin_inff()) : (half_t)0.f;
    const half4_t z = {zv, zv, zv, zv};
    if constexpr (RAW) {
      // statistics of the owned, in-image conv pixels (multiplier 0 / 1 per lane and row), then the plain f16 conversion
      const bool colown = rx >= 1 && rx <= CTW - 2 && (unsigned)(cx0 + rx) < (unsigned)p.OW;
      const float m0 = (colown && wid > 0 && (unsigned)(cy0 + 2 * wid) < (unsigned)p.OH) ? 1.f : 0.f;
      const float m1 = (colown && wid < 7 && (unsigned)(cy0 + 2 * wid + 1) < (unsigned)p.OH) ? 1.f : 0.f;
      const f32x4 m04 = {m0, m0, m0, m0}, m14 = {m1, m1, m1, m1};
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 a0 = {acc[0][tn][4 * q], acc[0][tn][4 * q + 1], acc[0][tn][4 * q + 2], acc[0][tn][4 * q + 3]};
          const f32x4 a1 = {acc[1][tn][4 * q], acc[1][tn][4 * q + 1], acc[1][tn][4 * q + 2], acc[1][tn][4 * q + 3]};
          const f32x4 b0 = a0 * m04, b1 = a1 * m14;
          ssum[tn][q] += b0;
          ssum[tn][q] += b1;
          ssq[tn][q] = __builtin_elementwise_fma(b0, a0, ssq[tn][q]);
          ssq[tn][q] = __builtin_elementwise_fma(b1, a1, ssq[tn][q]);
          h0[tn][q] = __builtin_convertvector(a0, half4_t);
          v[tn][q] = __builtin_convertvector(a1, half4_t);
        }
    } else {
    f32x4 scv[NT][4], shv[NT][4];  // BN constants of this lane's 2 x 16 channels: all reads issued before the first use
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        scv[tn][q] = *reinterpret_cast<const f32x4*>(bnc + tn * 32 + 8 * q + 4 * kg);
        shv[tn][q] = *reinterpret_cast<const f32x4*>(bnc + NCH + tn * 32 + 8 * q + 4 * kg);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 sc4 = scv[tn][q], sh4 = shv[tn][q];
        const f32x4 a0 = {acc[0][tn][4 * q], acc[0][tn][4 * q + 1], acc[0][tn][4 * q + 2], acc[0][tn][4 * q + 3]};
        const f32x4 a1 = {acc[1][tn][4 * q], acc[1][tn][4 * q + 1], acc[1][tn][4 * q + 2], acc[1][tn][4 * q + 3]};
        h0[tn][q] = __builtin_elementwise_max(__builtin_convertvector(__builtin_elementwise_fma(a0, sc4, sh4), half4_t), z);
        v[tn][q] = __builtin_elementwise_max(__builtin_convertvector(__builtin_elementwise_fma(a1, sc4, sh4), half4_t), z);
      }
    }
    // conv positions outside the image count as 0 (RAW: -inf): only tiles on the image border have any (wave-uniform test)
    if (cy0 < 0 || cx0 < 0 || cy0 + CTH > p.OH || cx0 + CTW > p.OW) {
      const bool colok = (unsigned)(cx0 + rx) < (unsigned)p.OW;
      const bool ok0 = colok && (unsigned)(cy0 + 2 * wid) < (unsigned)p.OH;
      const bool ok1 = colok && (unsigned)(cy0 + 2 * wid + 1) < (unsigned)p.OH;
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          h0[tn][q] = ok0 ? h0[tn][q] : z;
          v[tn][q] = ok1 ? v[tn][q] : z;
        }
    }
    if (wid > 0) {  // this wave's first row, for the wave above
      char* xw = xbuf + (buf * 7 + (wid - 1)) * XROW + rx * XPITCH + 8 * kg;
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<half4_t*>(xw + tn * 64 + q * 16) = h0[tn][q];
    }
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q) v[tn][q] = __builtin_elementwise_max(v[tn][q], h0[tn][q]);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();  // next patch landed, every wave is done with this one; the exchanged rows are visible
    ptile = tile;
    buf ^= 1;
  }
  if (ptile >= 0) pool_out(ptile, buf ^ 1, v);
  if constexpr (RAW) {
    // fold the per-lane partials: over the 32 pixel columns of a half-wave (same kg = same channels), then over the eight
    // waves through LDS, then one fp64 atomic pair per channel and workgroup (bn.hip's replica layout)
    if (p.stats != nullptr) {
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float a = ssum[tn][q][e], b = ssq[tn][q][e];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
            ssum[tn][q][e] = a; ssq[tn][q][e] = b;
          }
      __syncthreads();  // every wave is past its last use of xbuf
      float* part = reinterpret_cast<float*>(xbuf);  // [8 waves][64 channels][2]
      if (rx == 0) {
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int ch = tn * 32 + 8 * q + 4 * kg + e;
              part[(wid * NCH + ch) * 2 + 0] = ssum[tn][q][e];
              part[(wid * NCH + ch) * 2 + 1] = ssq[tn][q][e];
            }
      }
      __syncthreads();
      if (tid < NCH) {
        double sd = 0.0, qd = 0.0;
        for (int a = 0; a < 8; ++a) {
          sd += (double)part[(a * NCH + tid) * 2 + 0];
          qd += (double)part[(a * NCH + tid) * 2 + 1];
        }
        double* st = p.stats + (size_t)(blockIdx.x % AM_STATS_REPLICAS) * 2 * NCH;
        atomicAdd(st + tid, sd);
        atomicAdd(st + NCH + tid, qd);
      }
    }
  }
}

template <int TAPS, bool RAW>
int launch_s2d_pool(const S2dParams& p0, int POH, int POW, hipStream_t s) {
  constexpr int PH = CTH + TAPS - 1, PW = CTW + TAPS - 1;
  constexpr int PATCH_SLOT = ((PH * PW * CB + 1023) / 1024) * 1024;
  constexpr int WPITCH = TAPS * TAPS * 32 + 16;
  constexpr int W_BYTES = ((64 * WPITCH + 1023) / 1024) * 1024;
  constexpr int LDS = W_BYTES + 2 * PATCH_SLOT + 2 * 7 * XROW + 7 * OROW + 512;
  static_assert(LDS <= 160 * 1024, "LDS");
  S2dParams p = p0;
  const int pty = am_cdiv(POH, PTH), ptx = am_cdiv(POW, PTW);
  p.ntiles = p.B * pty * ptx;
  static bool attr_done_dev[AM_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[am_current_device()];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_s2d_pool_k<TAPS, RAW>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return AM_ERR_LAUNCH;
    attr_done = true;
  }
  const int grid = p.ntiles < 256 ? p.ntiles : 256;
  g_am_conv_variant = AM_CV_S2D_POOL;
  hipLaunchKernelGGL((conv_s2d_pool_k<TAPS, RAW>), dim3(grid), dim3(512), LDS, s, p, POH, POW, pty, ptx);
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
  const long long x_bytes = (long long)g->B * g->IH * g->IW * CB;
  if (x_bytes >= (1ll << 31)) return AM_ERR_UNSUPPORTED;  // 32-bit buffer offsets (B = 291 at 720p)
  S2dParams p;
  p.x_bytes = (unsigned)x_bytes;
  p.x = x; p.w = w; p.y = y; p.bias = bias; p.scale = scale; p.shift = shift; p.stats = stats;
  p.B = g->B; p.IH = g->IH; p.IW = g->IW; p.OH = g->OH; p.OW = g->OW; p.ldo = g->ldo; p.y_coff = g->y_coff;
  p.off0 = g->dy[0]; p.relu = relu; p.N = g->N;
  p.tiles_y = am_cdiv(g->OH, TH);
  p.tiles_x = am_cdiv(g->OW, TW);
  p.ntiles = p.B * p.tiles_y * p.tiles_x;
  if (mode == 3 || mode == 4) {
    // y is the max-pooled map [B, POH, POW, ldo], POH = (OH-1)/2+1
    if (g->ntaps != 4 || g->N != 64) return AM_ERR_UNSUPPORTED;
    return mode == 3 ? launch_s2d_pool<4, false>(p, (g->OH - 1) / 2 + 1, (g->OW - 1) / 2 + 1, s)
                     : launch_s2d_pool<4, true>(p, (g->OH - 1) / 2 + 1, (g->OW - 1) / 2 + 1, s);
  }
  if (mode == 0) return dispatch_s2d<0>(p, g->ntaps, s);
  if (mode == 1) return dispatch_s2d<1>(p, g->ntaps, s);
  if (mode == 2) return dispatch_s2d<2>(p, g->ntaps, s);
  return AM_ERR_ARG;
}
