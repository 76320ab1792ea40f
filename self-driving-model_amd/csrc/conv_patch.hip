// Weights-stationary 3x3 / stride-1 / pad-1 convolution for 64 -> 64 channels, f16 (ResNet-18 layer1: 35 % of the
// trunk's FLOPs, and its dgrad, which is the same shape with flipped taps).
//
// Why a second formulation: in the gather-GEMM every input pixel crosses the CU's L2->LDS path nine times (once per
// tap) and the 64-wide N tile cannot amortise it -- the layer is bound by per-CU load bandwidth (~66 GB/s), not by
// MFMA or HBM.  Here each persistent workgroup (one per CU) keeps the whole weight matrix (64 x 576 halves, 73 KiB) in
// LDS for its lifetime and stages one input PATCH per 8x32-pixel output tile (10 x 34 pixels x 128 B = 42.5 KiB,
// double-buffered, LDS-DMA): a pixel is fetched ~1.33x instead of 9x, and all nine taps read it from LDS.
// Bank conflicts: weight rows are padded to 1168 B (odd multiple of 16); patch pixels are 128 B, chunk index XOR-swizzled
// with (pixel >> 1) & 7 on the DMA source side and on the fragment read.
// BatchNorm statistics are accumulated across the workgroup's tiles in registers and flushed once (fp64 atomics).
#include "am_common.h"
#include <cstdlib>

namespace amp {

__device__ __attribute__((aligned(64))) unsigned char g_zero_line[64];

constexpr int TH = 8, TW = 32;             // output tile
constexpr int PH = TH + 2, PW = TW + 2;    // input patch
constexpr int CB = 128;                    // bytes per pixel (64 halves)
constexpr int PATCH_BYTES = PH * PW * CB;  // 43520
constexpr int PATCH_SLOT = 44032;          // + 512 B slack so the last DMA instruction stays inside its slot
constexpr int WROW = 1152, WPITCH = 1168;  // weight row bytes / padded pitch
constexpr int W_BYTES = 64 * WPITCH;       // 74752
constexpr int LDS_BYTES = W_BYTES + 2 * PATCH_SLOT + 1024;  // 163840 = 160 KiB

struct PatchParams {
  const void* x;
  const void* w;   // packed [>=64][576] halves (gather-GEMM forward packing)
  void* y;
  const float* bias;
  double* stats;
  int B, H, W, ldi, x_coff, ldo, y_coff, relu;
  int tiles_y, tiles_x, ntiles;
  int dbg;  // timing experiments only (AM_PATCH_DEBUG): 1 = no stores, 2 = no MFMA loop, 4 = no patch DMA after the first
};

__global__ __launch_bounds__(256) void conv3x3_c64n64_k(const PatchParams p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* Wl = smem;
  char* patch0 = smem + W_BYTES;

  const char* __restrict__ x = static_cast<const char*>(p.x);
  const char* __restrict__ w = static_cast<const char*>(p.w);
  half_t* __restrict__ y = static_cast<half_t*>(p.y);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const char* zl = reinterpret_cast<const char*>(g_zero_line);

  // ---- resident weights: LDS position q -> (n = q / 73, cc = q % 73); cc == 72 is the pad chunk ----
  for (int inst = wid; inst < W_BYTES / 1024; inst += 4) {
    const int q = inst * 64 + lane;
    const int n = q / 73, cc = q - n * 73;
    const char* src = cc < 72 ? w + (long long)n * WROW + cc * 16 : zl;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(Wl + inst * 1024), 16, 0, 0);
  }

  auto issue_patch = [&](int tile, int buf) {
    const int img = tile / (p.tiles_y * p.tiles_x);
    const int rem = tile - img * (p.tiles_y * p.tiles_x);
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
    char* dst = patch0 + buf * PATCH_SLOT;
    constexpr int NINST = (PATCH_BYTES + 1023) / 1024;  // 43; wave w owns the contiguous KiB range [11w, 11w+11)
    for (int inst = wid * 11; inst < NINST && inst < wid * 11 + 11; ++inst) {
      const int q = inst * 64 + lane;
      const int pidx = q >> 3, cpos = q & 7;
      const int c = cpos ^ ((pidx >> 1) & 7);
      const int prow = pidx / PW, pcol = pidx - prow * PW;
      const int iy = iy0 + prow, ix = ix0 + pcol;
      const bool ok = q < PATCH_BYTES / 16 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const char* src = ok ? x + (((long long)(img * p.H + iy) * p.W + ix) * p.ldi + p.x_coff) * 2 + c * 16 : zl;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + inst * 1024), 16, 0, 0);
    }
  };

  int tile = blockIdx.x;
  if (tile < p.ntiles) issue_patch(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // The MFMAs run "transposed" (A = weights, B = pixels): a lane owns ONE pixel column (rx) and, per 32-channel block tn,
  // the 16 channels tn*32 + 8*(r>>2) + 4*kg + (r&3) -- four consecutive channels per register quad, so the epilogue
  // converts and stages 8 bytes at a time and the BN statistics are plain (packed) vector adds into per-lane partial
  // sums, folded across the 32 pixel lanes once at the end of the kernel.
  f32x16 ssum[2], ssq[2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn)
#pragma unroll
    for (int r = 0; r < 16; ++r) { ssum[tn][r] = 0.f; ssq[tn][r] = 0.f; }
  const int rx = lane & 31, kg = lane >> 5;
  int buf = 0;
  for (; tile < p.ntiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    if (next < p.ntiles && !(p.dbg & 4)) issue_patch(next, buf ^ 1);
    const char* pt = patch0 + buf * PATCH_SLOT;

    f32x16 acc[2][2];  // [tn][tm]
    if (p.dbg & 2) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    }

    if (!(p.dbg & 2)) {
      // 36 k16 steps (tap-major), software-pipelined by hand: the four fragment reads of step s+2 are issued before the
      // four MFMAs of step s (the workgroup runs one wave per SIMD, so nothing else hides the LDS latency; left to
      // itself hipcc issues each step's reads right before an s_waitcnt lgkmcnt(0)).
      half8_t fa[3][2], fb[3][2];
      const char* brow = Wl + (lane & 31) * WPITCH + kg * 16;
      const int prow0 = (2 * wid) * PW + rx;
#pragma unroll
      for (int s = 0; s < 36 + 2; ++s) {
        // counted wait for step s-2's fragments BEFORE this step's reads are issued (LDS returns in order: the reads of
        // step s-1 stay in flight).  Spelled as the builtin so hipcc's own waitcnt pass sees it; by itself it emits
        // lgkmcnt(0) after the new reads.
        if (s >= 2) {
          if (s < 37) __builtin_amdgcn_s_waitcnt(0xC47F);  // lgkmcnt(4)
          else __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0)
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s < 36) {
          const int tap = s >> 2, ks = s & 3, kh = tap / 3, kw = tap - kh * 3;
          const int pid0 = prow0 + kh * PW + kw, pid1 = pid0 + PW;
          const int ch = ks * 2 + kg;
          fa[s % 3][0] = *reinterpret_cast<const half8_t*>(pt + pid0 * CB + ((ch ^ ((pid0 >> 1) & 7)) << 4));
          fa[s % 3][1] = *reinterpret_cast<const half8_t*>(pt + pid1 * CB + ((ch ^ ((pid1 >> 1) & 7)) << 4));
          fb[s % 3][0] = *reinterpret_cast<const half8_t*>(brow + tap * CB + ks * 32);
          fb[s % 3][1] = *reinterpret_cast<const half8_t*>(brow + 32 * WPITCH + tap * CB + ks * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s >= 2) {
          const int c = (s - 2) % 3;
          if (s == 2) {  // first step starts from the constant zero: no per-tile accumulator clears
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
              for (int tm = 0; tm < 2; ++tm) acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[c][tn], fa[c][tm], z, 0, 0, 0);
          } else {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
              for (int tm = 0; tm < 2; ++tm) acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[c][tn], fa[c][tm], acc[tn][tm], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // The next patch had the whole MFMA phase to land.  Drain the DMA and release the current buffer BEFORE the
    // epilogue: with an LDS-DMA in flight hipcc puts s_waitcnt vmcnt(0) in front of every global store (it cannot
    // prove the DMA's source does not alias the store), which serialises the 64 stores of a lane.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    buf ^= 1;

    // ---- epilogue: acc[tn][tm][r] = out(pixel (ry = 2*wid+tm, px = rx), channel tn*32 + 8*(r>>2) + 4*kg + (r&3)) ----
    const int img = tile / (p.tiles_y * p.tiles_x);
    const int rem = tile - img * (p.tiles_y * p.tiles_x);
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    if (ty * TH + TH > p.H || tx * TW + TW > p.W) {
      // edge tile: pixels outside the image are not conv outputs -- zero them so they stay out of the statistics
      // (they are not stored either)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const bool ok = ty * TH + 2 * wid + tm < p.H && tx * TW + rx < p.W;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[tn][tm][r] = ok ? acc[tn][tm][r] : 0.f;
      }
    }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        ssum[tn] += acc[tn][tm];
        ssq[tn] = __builtin_elementwise_fma(acc[tn][tm], acc[tn][tm], ssq[tn]);
      }
    if (p.bias != nullptr || p.relu) {  // not used by the BN trunk; kept for the generic conv contract
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float bv = p.bias ? p.bias[tn * 32 + 8 * (r >> 2) + 4 * kg + (r & 3)] : 0.f;
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) {
            float v = acc[tn][tm][r] + bv;
            acc[tn][tm][r] = p.relu ? fmaxf(v, 0.f) : v;
          }
        }
    }
    // Stage this wave's 64 pixels x 64 channels in its own KiB range of the patch buffer it just finished reading
    // (the same range this wave refills by DMA next iteration, so no other wave touches it): 8-byte writes of four
    // consecutive channels, then whole 128-byte pixel rows go out with 16-byte stores.
    constexpr int SP = 144;
    char* stg = const_cast<char*>(pt) + wid * 11 * 1024;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 v = {acc[tn][tm][4 * j], acc[tn][tm][4 * j + 1], acc[tn][tm][4 * j + 2], acc[tn][tm][4 * j + 3]};
          *reinterpret_cast<half4_t*>(stg + (tm * 32 + rx) * SP + (tn * 32 + 8 * j + 4 * kg) * 2) = __builtin_convertvector(v, half4_t);
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (!(p.dbg & 1)) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int q = it * 64 + lane;
        const int row = q >> 3, cc = q & 7;  // row = tm*32 + px
        const int oy = ty * TH + 2 * wid + (row >> 5), ox = tx * TW + (row & 31);
        if (oy < p.H && ox < p.W)
          *reinterpret_cast<uint4*>(y + ((long long)(img * p.H + oy) * p.W + ox) * p.ldo + p.y_coff + cc * 8) =
              *reinterpret_cast<const uint4*>(stg + row * SP + cc * 16);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging reads done before this wave's next DMA reuses the range
  }

  if (p.stats != nullptr) {
    // fold the 32 pixel lanes of each half-wave (xor < 32 stays inside the half), then 4 waves -> LDS -> one fp64 atomic
    // per channel per workgroup
    float* part = reinterpret_cast<float*>(smem + W_BYTES);  // [4 waves][64 channels][2] in the (now idle) patch area
    __syncthreads();
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sv = ssum[tn][r], qv = ssq[tn][r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
          sv += __shfl_xor(sv, o, 64);
          qv += __shfl_xor(qv, o, 64);
        }
        if (rx == 0) {
          const int ch = tn * 32 + 8 * (r >> 2) + 4 * kg + (r & 3);
          part[(wid * 64 + ch) * 2 + 0] = sv;
          part[(wid * 64 + ch) * 2 + 1] = qv;
        }
      }
    __syncthreads();
    if (tid < 64) {
      double s = 0.0, q = 0.0;
      for (int a = 0; a < 4; ++a) {
        s += (double)part[(a * 64 + tid) * 2 + 0];
        q += (double)part[(a * 64 + tid) * 2 + 1];
      }
      double* st = p.stats + (size_t)(blockIdx.x % AM_STATS_REPLICAS) * 2 * 64;
      atomicAdd(st + tid, s);
      atomicAdd(st + 64 + tid, q);
    }
  }
  (void)Wl;
}

}  // namespace amp

// Returns AM_ERR_UNSUPPORTED unless the geometry is exactly a dense 3x3 / stride 1 / pad 1, 64 -> 64 f16 convolution
// (forward packing, tap order kh-major) -- then runs the weights-stationary kernel.
int am_conv3x3_c64n64_f16(const am_conv_geom* g, const void* x, const void* w, const float* bias, int relu, void* y, double* stats,
                          hipStream_t s) {
  using namespace amp;
  if (g->ntaps != 9 || g->krun != 64 || g->N != 64 || g->pix_shift != 31) return AM_ERR_UNSUPPORTED;
  if (g->iys != 1 || g->ixs != 1 || g->oys != 1 || g->oxs != 1 || g->oy0 != 0 || g->ox0 != 0) return AM_ERR_UNSUPPORTED;
  if (g->MH != g->IH || g->MW != g->IW || g->OH != g->IH || g->OW != g->IW) return AM_ERR_UNSUPPORTED;
  for (int t = 0; t < 9; ++t)
    if (g->dy[t] != t / 3 - 1 || g->dx[t] != t % 3 - 1) return AM_ERR_UNSUPPORTED;
  if (g->IW < TW || (long long)g->B * g->IH * g->IW < 64 * 1024) return AM_ERR_UNSUPPORTED;  // small problems: gather-GEMM
  PatchParams p;
  p.x = x; p.w = w; p.y = y; p.bias = bias; p.stats = stats;
  p.B = g->B; p.H = g->IH; p.W = g->IW; p.ldi = g->ldi; p.x_coff = g->x_coff; p.ldo = g->ldo; p.y_coff = g->y_coff; p.relu = relu;
  p.tiles_y = am_cdiv(g->IH, TH);
  p.tiles_x = am_cdiv(g->IW, TW);
  p.ntiles = p.B * p.tiles_y * p.tiles_x;
  p.dbg = 0;
  static bool attr_done_dev[AM_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[am_current_device()];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64n64_k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
      return AM_ERR_LAUNCH;
    attr_done = true;
  }
  const int grid = p.ntiles < 256 ? p.ntiles : 256;  // one persistent workgroup per CU
  g_am_conv_variant = AM_CV_PATCH_C64;
  hipLaunchKernelGGL(conv3x3_c64n64_k, dim3(grid), dim3(256), LDS_BYTES, s, p);
  AM_CHECK_LAUNCH();
  return AM_OK;
}
