// Weights-in-registers 3x3 / stride-1 / pad-1 convolution for 64 -> 64 channels, f16 (ResNet-18 layer1 and its dgrad).
//
// Successor of conv_patch.hip (weights resident in LDS).  That kernel issues one 1-KiB LDS fragment read per MFMA and
// runs one wave per SIMD (LDS capacity), so its main loop is bound by LDS bandwidth and nothing overlaps its epilogue.
// Here every wave keeps the WHOLE weight matrix as MFMA A-fragments in registers -- 36 k16-steps x 2 channel blocks x
// half8 = 288 VGPRs; with one wave per SIMD a lane may use 512 -- so only the pixel fragments come from LDS (half the
// fragment traffic), and the 73 KiB the weights used to occupy hold a dedicated epilogue staging area instead.
//   * input patch per 8x32-pixel output tile: 10 x 34 pixels at a 144-byte LDS pitch (128 B of channels + 16 B pad:
//     16 consecutive pixels start in 16 different 4-bank groups, so the ds_read_b128 fragment reads are conflict free
//     without a swizzle and every (tap, k16-step) offset is an instruction immediate), double-buffered, filled by
//     buffer_load ... lds (out-of-range offsets deliver the zero padding);
//   * MFMAs run "transposed" (A = weights, B = pixels): a lane owns one pixel and four consecutive channels per register
//     quad, so the epilogue is packed adds/FMAs for the BN statistics, v_cvt_pk_f16_f32 and 8-byte LDS writes.
#include "am_common.h"
#include <cstdlib>

namespace amp2 {

constexpr int TH = 8, TW = 32;             // output tile
constexpr int PH = TH + 2, PW = TW + 2;    // input patch
constexpr int PP = 144;                    // LDS bytes per patch pixel (9 chunks of 16 B, the last one padding)
constexpr int NPIX = PH * PW;              // 340
constexpr int NINST = (NPIX * 9 + 63) / 64;   // 48 wave-instructions of 64 x 16 B
constexpr int PATCH_SLOT = NINST * 1024;      // 49152
constexpr int IPW = NINST / 4;                // instructions per wave (12)
constexpr int WROW = 1152;                    // bytes per packed weight row (576 halves)
constexpr int SP = 144;                       // staging pitch per output pixel
constexpr int STG_WAVE = 64 * SP;             // 9216
constexpr int LDS_BYTES = 2 * PATCH_SLOT + 4 * STG_WAVE + 2048;  // 137216
constexpr unsigned OOB = 0xC0000000u;         // + any tile base stays above num_records (< 2^30)
static_assert(NINST % 4 == 0, "patch instructions split evenly over the four waves");

struct Patch2Params {
  const void* x;
  const void* w;   // packed [>=64][576] halves (gather-GEMM forward packing)
  void* y;
  double* stats;
  int B, H, W, ldi, x_coff, ldo, y_coff;
  int tiles_y, tiles_x, ntiles;
  unsigned x_bytes;
  int dbg;  // timing experiments only (AM_PATCH_DEBUG): 1 = no stores, 2 = no MFMA loop, 4 = no patch DMA after the first, 8 = no epilogue math
};

typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ void buffer_to_lds16(const void* base, unsigned bytes, char* dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000),
                                           (lds_ptr)dst, 16, voff, 0, 0, 0);
}

__global__ __launch_bounds__(256) void conv3x3_c64n64_wreg_k(const Patch2Params p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const char* __restrict__ w = static_cast<const char*>(p.w);
  half_t* __restrict__ y = static_cast<half_t*>(p.y);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rx = lane & 31, kg = lane >> 5;

  // ---- weights -> registers: wf[tap*4 + ks][tn] = A fragment (row = channel tn*32 + rx, k chunk 2*ks + kg) ----
  half8_t wf[36][2];
#pragma unroll
  for (int s = 0; s < 36; ++s)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
      wf[s][tn] = *reinterpret_cast<const half8_t*>(w + (long long)(tn * 32 + rx) * WROW + (s >> 2) * 128 + (s & 3) * 32 + kg * 16);

  // ---- patch loader state (tile independent): LDS position q = (wid*IPW + i)*64 + lane -> pixel q/9, chunk q%9 ----
  const int row_bytes = p.W * p.ldi * 2, pix_bytes = p.ldi * 2;

  auto issue_patch = [&](int tile, int buf) {
    const int img = tile / (p.tiles_y * p.tiles_x);
    const int rem = tile - img * (p.tiles_y * p.tiles_x);
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
    const unsigned tb = (unsigned)((((img * p.H + iy0) * p.W + ix0) * p.ldi + p.x_coff) * 2);  // wraps for the halo row/col: fine
    char* dst = smem + buf * PATCH_SLOT + wid * (IPW * 1024);
    // LDS position q = (wid*IPW + i)*64 + lane -> patch pixel q/9, 16-byte chunk q%9 (chunk 8 = pad).  Recomputed per
    // tile from an opaque copy of the lane id: 512 registers are spoken for, and values hipcc hoists out of the tile loop
    // get spilled -- a spill reload next to the LDS-DMA costs an s_waitcnt vmcnt(0), i.e. the whole DMA/MFMA overlap.
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int q = (wid * IPW + i) * 64 + ln;
      const int pix = q / 9, cc = q - pix * 9;
      const int prow = pix / PW, pcol = pix - prow * PW;
      const bool ok = cc < 8 && pix < NPIX && (unsigned)(iy0 + prow) < (unsigned)p.H && (unsigned)(ix0 + pcol) < (unsigned)p.W;
      const unsigned vo = ok ? tb + (unsigned)(prow * row_bytes + pcol * pix_bytes + cc * 16) : OOB;
      buffer_to_lds16(p.x, p.x_bytes, dst + i * 1024, vo);
      __builtin_amdgcn_sched_barrier(0);  // one offset live at a time (512 registers are spoken for)
    }
  };

  int tile = blockIdx.x;
  if (tile < p.ntiles) issue_patch(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // per-lane partial BN sums of the lane's 32 channels (see conv_patch.hip), folded across pixel lanes at the end
  f32x16 ssum[2], ssq[2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn)
#pragma unroll
    for (int r = 0; r < 16; ++r) { ssum[tn][r] = 0.f; ssq[tn][r] = 0.f; }

  const int fbase = ((2 * wid) * PW + rx) * PP + kg * 16;  // this lane's pixel fragment origin inside a patch
  char* stg = smem + 2 * PATCH_SLOT + wid * STG_WAVE;
  int buf = 0;
  for (; tile < p.ntiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    if (next < p.ntiles && !(p.dbg & 4)) issue_patch(next, buf ^ 1);
    const char* pt = smem + buf * PATCH_SLOT + fbase;

    // 36 k16 steps (tap-major), software-pipelined by hand: the two pixel-fragment reads of step s+1 are issued before
    // the four MFMAs of step s (two register slots).  The wait is the builtin so that hipcc's waitcnt pass sees it (by
    // itself it emits lgkmcnt(0) right AFTER the newest reads).
    f32x16 acc[2][2];  // [tn][tm]
    half8_t fp[2][2];
    if (p.dbg & 2) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    }
    if (!(p.dbg & 2))
#pragma unroll
    for (int s = 0; s < 36 + 1; ++s) {
      if (s >= 1) __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): step s-1's fragments (issued one MFMA group ago)
      __builtin_amdgcn_sched_barrier(0);
      if (s < 36) {
        const int tap = s >> 2, ks = s & 3, kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
          fp[s & 1][tm] = *reinterpret_cast<const half8_t*>(pt + ((kh + tm) * PW + kw) * PP + ks * 32);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s >= 1) {
        const int c = (s - 1) & 1;
        if (s == 1) {  // first step starts from the constant zero: no per-tile accumulator clears
          const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][tn], fp[c][tm], z, 0, 0, 0);
        } else {
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s - 1][tn], fp[c][tm], acc[tn][tm], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    // next patch landed (it had the whole MFMA phase); every wave is done reading the current one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    buf ^= 1;

    // ---- epilogue: acc[tn][tm][r] = out(pixel (ry = 2*wid+tm, px = rx), channel tn*32 + 8*(r>>2) + 4*kg + (r&3)) ----
    const int img = tile / (p.tiles_y * p.tiles_x);
    const int rem = tile - img * (p.tiles_y * p.tiles_x);
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    if (ty * TH + TH > p.H || tx * TW + TW > p.W) {
      // edge tile: pixels outside the image are not conv outputs -- zero them so they stay out of the statistics
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        const bool ok = ty * TH + 2 * wid + tm < p.H && tx * TW + rx < p.W;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[tn][tm][r] = ok ? acc[tn][tm][r] : 0.f;
      }
    }
    if (!(p.dbg & 8))
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        ssum[tn] += acc[tn][tm];
        ssq[tn] = __builtin_elementwise_fma(acc[tn][tm], acc[tn][tm], ssq[tn]);
      }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 v = {acc[tn][tm][4 * j], acc[tn][tm][4 * j + 1], acc[tn][tm][4 * j + 2], acc[tn][tm][4 * j + 3]};
          *reinterpret_cast<half4_t*>(stg + (tm * 32 + rx) * SP + (tn * 32 + 8 * j + 4 * kg) * 2) = __builtin_convertvector(v, half4_t);
        }
    // the wave reads back what its own lanes wrote (LDS executes a wave's accesses in order); the asm also keeps the
    // compiler from moving the differently typed reads above the writes
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (!(p.dbg & 1))
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int q = it * 64 + lane;
      const int row = q >> 3, cc = q & 7;  // row = tm*32 + px
      const int oy = ty * TH + 2 * wid + (row >> 5), ox = tx * TW + (row & 31);
      if (oy < p.H && ox < p.W)
        *reinterpret_cast<uint4*>(y + ((long long)(img * p.H + oy) * p.W + ox) * p.ldo + p.y_coff + cc * 8) =
            *reinterpret_cast<const uint4*>(stg + row * SP + cc * 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging reads done before the next tile's writes
  }

  if (p.stats != nullptr) {
    // fold the 32 pixel lanes of each half-wave (xor < 32 stays inside the half), then 4 waves -> LDS -> one fp64 atomic
    // per channel per workgroup
    float* part = reinterpret_cast<float*>(smem);  // [4 waves][64 channels][2] in the (now idle) patch area
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
}

}  // namespace amp2

// Returns AM_ERR_UNSUPPORTED unless the geometry is exactly a dense 3x3 / stride 1 / pad 1, 64 -> 64 f16 convolution
// (forward packing, tap order kh-major) over a tensor small enough for 30-bit buffer offsets.
int am_conv3x3_c64n64_wreg_f16(const am_conv_geom* g, const void* x, const void* w, const float* bias, int relu, void* y,
                               double* stats, hipStream_t s) {
  using namespace amp2;
  if (g->ntaps != 9 || g->krun != 64 || g->N != 64 || g->pix_shift != 31) return AM_ERR_UNSUPPORTED;
  if (g->iys != 1 || g->ixs != 1 || g->oys != 1 || g->oxs != 1 || g->oy0 != 0 || g->ox0 != 0) return AM_ERR_UNSUPPORTED;
  if (g->MH != g->IH || g->MW != g->IW || g->OH != g->IH || g->OW != g->IW) return AM_ERR_UNSUPPORTED;
  for (int t = 0; t < 9; ++t)
    if (g->dy[t] != t / 3 - 1 || g->dx[t] != t % 3 - 1) return AM_ERR_UNSUPPORTED;
  if (g->IW < TW || (long long)g->B * g->IH * g->IW < 64 * 1024) return AM_ERR_UNSUPPORTED;  // small problems: gather-GEMM
  const long long x_bytes = (long long)g->B * g->IH * g->IW * g->ldi * 2;
  if (x_bytes >= (1ll << 30) || bias != nullptr || relu) return AM_ERR_UNSUPPORTED;  // BN trunk layers only (no bias / ReLU epilogue)
  Patch2Params p;
  p.x = x; p.w = w; p.y = y; p.stats = stats;
  p.B = g->B; p.H = g->IH; p.W = g->IW; p.ldi = g->ldi; p.x_coff = g->x_coff; p.ldo = g->ldo; p.y_coff = g->y_coff;
  p.tiles_y = am_cdiv(g->IH, TH);
  p.tiles_x = am_cdiv(g->IW, TW);
  p.ntiles = p.B * p.tiles_y * p.tiles_x;
  p.x_bytes = (unsigned)x_bytes;
  { const char* e = getenv("AM_PATCH_DEBUG"); p.dbg = e ? atoi(e) : 0; }
  static bool attr_done_dev[AM_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[am_current_device()];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64n64_wreg_k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
      return AM_ERR_LAUNCH;
    attr_done = true;
  }
  const int grid = p.ntiles < 256 ? p.ntiles : 256;  // one persistent workgroup per CU
  g_am_conv_variant = AM_CV_WREG_C64;
  hipLaunchKernelGGL(conv3x3_c64n64_wreg_k, dim3(grid), dim3(256), LDS_BYTES, s, p);
  AM_CHECK_LAUNCH();
  return AM_OK;
}
