// Weights-in-registers 3x3 / stride-1 / pad-1 convolution for 64 -> 64 channels, f16 (ResNet-18 layer1 and its dgrad),
// two persistent workgroups per CU.
//
// Measured on conv_patch.hip (weights in LDS, one wave per SIMD): the 144 MFMAs of a tile run at the matrix-core peak,
// but they are only ~30 % of the tile time -- patch DMA issue, BN statistics, f16 conversion, LDS staging and the global
// stores all run on the same single wave per SIMD, serialised before and after the MFMA phase.  Here
//   * a wave keeps the weights of ONE 32-channel block as MFMA A-fragments in registers (36 k16-steps x half8 = 144
//     VGPRs), so two waves fit on a SIMD (<= 256 VGPRs each) and no weight fragment is ever read from LDS;
//   * a workgroup is four waves (2 pixel halves x 2 channel blocks) with its own stream of 8x16-pixel tiles and its own
//     double-buffered input patch (73 KiB of LDS), and TWO workgroups are resident per CU: their tile loops drift out of
//     phase, so while one runs MFMAs the other runs its epilogue / issues its next patch DMA on the same SIMDs;
//   * input patch: 10 x 18 pixels at a 144-byte LDS pitch (conflict-free ds_read_b128 without a swizzle, every
//     (tap, k16-step) offset an instruction immediate), filled by buffer_load ... lds (out-of-range offsets give the
//     zero padding); in the 16x16x32 form (160-byte pitch) one LDS-DMA wave-instruction delivers six whole pixels, so its
//     patch row / column base are wave-uniform and a lane's chunk is fixed: ~4 instead of ~28 VALU instructions per piece,
//     and the input transform reads its scale / shift once per tile;
//   * MFMAs run "transposed" (A = weights, B = pixels): a lane owns one pixel and four consecutive channels per register
//     quad -- packed adds / FMAs for the BN statistics, v_cvt_pk_f16_f32, 8-byte LDS staging writes, 16-byte stores;
//   * stores are inline-asm buffer stores (see buffer_store16_asm) so that the patch prefetch stays in flight across them.
#include "am_common.h"
#include <cstdlib>

// Timing ablations for scratch/ablate_duo (garbage results with any bit set; the library is built with 0):
//   1 no patch LDS-DMA in the tile loop, 2 no MFMAs, 4 no fragment reads, 8 no global stores
#ifndef AMP3_ABL
#define AMP3_ABL 0
#endif
#ifndef AMP3_ROWREUSE
#define AMP3_ROWREUSE 1  // M16 form: every patch-row fragment read once per (kw, k-half) and reused by the (output row, vertical tap) pairs on it
#endif
#ifndef AMP3_FRING
#define AMP3_FRING 7  // M16 form: prefetch distance of the fragment ring in fragments (0: the step-wise form, four fragments per eight MFMAs)
#endif
#ifndef AMP3_VMCNT4
#define AMP3_VMCNT4 1  // 1 (round 3): the previous tile's four stores stay in flight across the patch wait (vmcnt(4)): 148 -> 142 us at B = 32
#endif

namespace amp3 {

// Diagnostic build only (scratch/ablate_duo, -DAMP3_DIAG): wave 0 of workgroup 0 sums the s_memtime cycles of its tile-loop phases
#ifdef AMP3_DIAG
__device__ long long g_duo_diag[8];
#define AMP3_STAMP(k)                                  \
  do {                                                 \
    const long long t_now_ = __builtin_readcyclecounter(); \
    dg[k] += t_now_ - t_last;                          \
    t_last = t_now_;                                   \
  } while (0)
#else
#define AMP3_STAMP(k) do { } while (0)
#endif

constexpr int TH = 8, TW = 16;             // output tile of one group
constexpr int PH = TH + 2, PW = TW + 2;    // input patch
constexpr int NPIX = PH * PW;              // 180
constexpr int WROW = 1152;                    // bytes per packed weight row (576 halves)
constexpr int SP = 80;                        // staging pitch per output pixel: 32 channels x 2 B + 16 B pad
constexpr int STG_WAVE = 64 * SP;             // 5120
// M16 = false: v_mfma_f32_32x32x16_f16, patch pixels at a 144-byte LDS pitch (9 chunks of 16 B, the last one padding).
// M16 = true:  v_mfma_f32_16x16x32_f16 (less power per FLOP: the launch is power-limited), a pixel fragment is one 16-pixel
//              tile row and 32 channels; its ds_read_b128 is conflict free at a 160-byte pitch (10 chunks, two of padding: a
//              16-lane service group {0-3,12-15 | k-quarter q} + {4-11 | k-quarter q+1} then covers the 16 bank quads exactly once,
//              at 144 bytes it does not for any pixel order).
template <bool M16>
struct Lay {
  static constexpr int CPP = M16 ? 10 : 9;                 // 16-byte chunks per patch pixel
  static constexpr int PP = CPP * 16;                      // LDS bytes per patch pixel
  static constexpr int NINST = (NPIX * CPP + 63) / 64;     // wave-instructions of 64 x 16 B (26 / 29)
  static constexpr int PATCH_SLOT = NINST * 1024;
  static constexpr int IPW = (NINST + 3) / 4;              // instructions per wave of a group (the last ones are skipped)
  // M16 (round 3): a DMA wave-instruction delivers 6 whole patch pixels (60 chunks = 960 B; lanes 60-63 and the two pad chunks of a
  // pixel are masked off) -- 18 = 3 x 6, so piece n is patch row n / 3, columns (n % 3) * 6 + lane / 10: row, column base and LDS
  // address are wave-uniform, a lane's chunk (lane % 10) and byte offset inside the piece never change
  static constexpr int PIECE_PIX = 6, NPIECE = NPIX / PIECE_PIX, PIECE_BYTES = PIECE_PIX * PP;
  static_assert(4 * IPW - 2 == NPIECE || !M16, "the last wave has two pieces less");
  static_assert(!M16 || (NPIX % PIECE_PIX == 0 && PW % PIECE_PIX == 0 && PIECE_PIX * CPP <= 64 && (NPIECE + 3) / 4 == IPW && NPIECE * PIECE_BYTES <= PATCH_SLOT), "pieces");
  static constexpr int PATCH_BYTES = 2 * PATCH_SLOT;       // double buffer
  static constexpr int LDS_BYTES = PATCH_BYTES + 4 * STG_WAVE + 1024;  // 74,752 / 80,896: two workgroups per CU
  static_assert(64 * WROW <= PATCH_BYTES + 4 * STG_WAVE, "weights pass through the patch + staging area once");
  static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};
constexpr unsigned OOB = 0xC0000000u;         // + any tile base stays above num_records (< 2^30)

struct Patch3Params {
  const void* x;
  const void* w;   // packed [>=64][576] halves (gather-GEMM forward packing)
  void* y;
  double* stats;
  int B, H, W, ldi, x_coff, ldo, y_coff;
  int tiles_y, tiles_x, ntiles;
  unsigned x_bytes, w_bytes, y_bytes;
  // optional input transform relu(x * pre_scale[c] + pre_shift[c]) applied to the staged patch: the BatchNorm + ReLU of the
  // producing layer (bn.hip bn_apply_k arithmetic, fp32), so its normalised output never goes through HBM
  const float* pre_scale;
  const float* pre_shift;
  // EPI variant (inference: BatchNorm folded into weights / bias): y = act(conv + bias[n] + res), res laid out like y
  const float* bias;
  const void* res;
  int relu;
};

typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ void buffer_to_lds16(const void* base, unsigned bytes, char* dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000),
                                           (lds_ptr)dst, 16, voff, 0, 0, 0);
}

typedef int rsrc_words_t __attribute__((ext_vector_type(4)));

// 16-byte load through a raw buffer descriptor (offsets >= bytes return zeros); the builtin, not asm: the compiler must know
// when the registers are written
__device__ __forceinline__ rsrc_words_t buffer_load16(const void* base, unsigned bytes, unsigned voff) {
  return __builtin_bit_cast(rsrc_words_t, __builtin_amdgcn_raw_buffer_load_b128(
                                              __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000), voff, 0, 0));
}

// 16-byte store through a raw buffer descriptor, as inline asm on purpose: hipcc's waitcnt pass puts s_waitcnt vmcnt(0)
// in front of every store it can see while an LDS-DMA may be outstanding (it cannot prove the DMA source does not alias
// the store), which would drain the patch prefetch and serialise the stores.  Offsets >= num_records are dropped by the
// hardware, so out-of-image pixels need no branch and every wave issues the same number of stores (the counted vmcnt
// wait below relies on that).
__device__ __forceinline__ void buffer_store16_asm(rsrc_words_t v, rsrc_words_t rsrc, unsigned voff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
}

// EPI = false: raw output + BatchNorm statistics (training / train-mode BatchNorm).  EPI = true: no statistics (their 32
// registers hold the bias and the residual tile instead); the residual is fetched in the store loop's layout (16-byte pieces,
// full 64-byte half rows) at the START of the tile, lands under the MFMA phase, and is added to the f16-rounded conv + bias
// with a packed f16 add (= the fp32 add of two f16 values rounded once), the same two roundings as conv -> f16 -> normalise pass.

template <bool EPI, bool M16>
__global__ __launch_bounds__(256, 2) void conv3x3_c64n64_duo_k(const Patch3Params p) {
  using L = Lay<M16>;
  constexpr int PP = L::PP, CPP = L::CPP, NINST = L::NINST, PATCH_SLOT = L::PATCH_SLOT, IPW = L::IPW, PATCH_BYTES = L::PATCH_BYTES;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  rsrc_words_t yr;
  {
    const unsigned long long ya = reinterpret_cast<unsigned long long>(p.y);
    yr.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ya);
    yr.y = __builtin_amdgcn_readfirstlane((int)((ya >> 32) & 0xffffu));
    yr.z = __builtin_amdgcn_readfirstlane((int)p.y_bytes);
    yr.w = 0x00020000;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w4 = wid, hsel = w4 >> 1, tn = w4 & 1;
  const int rx = lane & 31, kg = lane >> 5;
  // pixel of this lane inside a 2 x 16 pixel fragment.  ds_read_b128 is serviced in the fixed 16-lane groups
  // {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): at the 144-byte pitch a group is conflict free iff its 16 patch pixels
  // differ mod 16, and the second fragment row starts 18 = 2 (mod 16) pixels later -- so row 1 is rotated by two columns.
  const int frow = rx >> 4, fcol = rx < 16 ? rx : (rx + 14) & 15;
  const int c16 = lane & 15, kq = lane >> 4;  // M16: pixel column of the lane inside a 16-pixel row fragment, k quarter (8 channels)

  // ---- weights: global -> LDS (coalesced, once) -> registers.  wf[tap*4 + ks] = A fragment (row = channel tn*32 + rx,
  // k chunk 2*ks + kg) ----
  for (int inst = wid; inst < 64 * WROW / 1024; inst += 4) buffer_to_lds16(p.w, p.w_bytes, smem + inst * 1024, (unsigned)(inst * 1024 + lane * 16));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // M16: wf[(tap*2 + k32 step)*2 + ca] = A fragment (row = channel tn*32 + ca*16 + c16, 8 channels of k quarter kq)
  half8_t wf[36];
#pragma unroll
  for (int s = 0; s < 36; ++s) {
    if constexpr (M16) wf[s] = *reinterpret_cast<const half8_t*>(smem + (tn * 32 + (s & 1) * 16 + c16) * WROW + (s >> 2) * 128 + ((s >> 1) & 1) * 64 + kq * 16);
    else wf[s] = *reinterpret_cast<const half8_t*>(smem + (tn * 32 + rx) * WROW + (s >> 2) * 128 + (s & 3) * 32 + kg * 16);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();  // everyone holds its fragments: the patch area may be overwritten

  const int row_bytes = p.W * p.ldi * 2, pix_bytes = p.ldi * 2;
  const int tiles_per_img = p.tiles_y * p.tiles_x;
  char* const gpatch = smem;
  float* const aff = reinterpret_cast<float*>(smem + PATCH_BYTES + 4 * STG_WAVE);  // [2][64] scale, shift (input transform)
  const bool pre = !EPI && p.pre_scale != nullptr;  // (the host entry refuses an input transform together with the inference epilogue)
  if (pre && tid < 128) aff[tid] = tid < 64 ? p.pre_scale[tid] : p.pre_shift[tid - 64];

  // M16: a lane's fixed place inside every DMA piece -- pixel pl, chunk pcc -- recomputed per call from an opaque copy of the lane id
  // (a handful of instructions; kept across the tile loop they cost registers the kernel does not have)
#define AMP3_LANE_PLACE()                                                  \
  int ln = lane;                                                           \
  asm volatile("" : "+v"(ln));                                             \
  const int pl = ln / 10, pcc = ln - pl * 10;                              \
  const bool dma_lane = ln < L::PIECE_PIX * 10 && pcc < 8;

  auto issue_patch = [&](int tile, int buf) {
    const int img = tile / tiles_per_img;
    const int rem = tile - img * tiles_per_img;
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
    const unsigned tb = (unsigned)((((img * p.H + iy0) * p.W + ix0) * p.ldi + p.x_coff) * 2);  // wraps for the halo row/col: fine
    if constexpr (M16) {
      char* dst = gpatch + buf * PATCH_SLOT;
      AMP3_LANE_PLACE();
      const unsigned lane_off = (unsigned)(pl * pix_bytes + pcc * 16);
      if (dma_lane) {
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
          const int n = w4 * IPW + i;
          if (n < L::NPIECE) {
            const int prow = n / 3, pc0 = (n - prow * 3) * L::PIECE_PIX;  // wave-uniform
            const bool ok = (unsigned)(iy0 + prow) < (unsigned)p.H && (unsigned)(ix0 + pc0 + pl) < (unsigned)p.W;
            const unsigned vo = ok ? tb + (unsigned)(prow * row_bytes + pc0 * pix_bytes) + lane_off : OOB;  // outside the image: zeros
            buffer_to_lds16(p.x, p.x_bytes, dst + n * L::PIECE_BYTES, vo);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      return;
    }
    char* dst = gpatch + buf * PATCH_SLOT + w4 * (IPW * 1024);
    // LDS position q = (w4*IPW + i)*64 + lane -> patch pixel q/CPP, 16-byte chunk q%CPP (chunks >= 8 = pad).  Recomputed per tile
    // from an opaque copy of the lane id: values hipcc hoists out of the tile loop end up spilled, and a spill reload
    // next to the LDS-DMA costs an s_waitcnt vmcnt(0).
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      if (w4 * IPW + i < NINST) {
        const int q = (w4 * IPW + i) * 64 + ln;
        const int pix = q / CPP, cc = q - pix * CPP;
        const int prow = pix / PW, pcol = pix - prow * PW;
        const bool ok = cc < 8 && pix < NPIX && (unsigned)(iy0 + prow) < (unsigned)p.H && (unsigned)(ix0 + pcol) < (unsigned)p.W;
        const unsigned vo = ok ? tb + (unsigned)(prow * row_bytes + pcol * pix_bytes + cc * 16) : OOB;
        buffer_to_lds16(p.x, p.x_bytes, dst + i * 1024, vo);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // Input transform of the patch of `tile` in buffer `buf`: every lane rewrites exactly the 16-byte chunks its own DMA
  // instructions delivered (call after this wave's vmcnt wait, before the barrier that publishes the patch).  Pixels outside
  // the image stay zero (the convolution's padding applies to the TRANSFORMED tensor).
  auto transform_patch = [&](int tile, int buf) {
    const int img = tile / tiles_per_img;
    const int rem = tile - img * tiles_per_img;
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
    if constexpr (M16) {
      AMP3_LANE_PLACE();
      char* dst = gpatch + buf * PATCH_SLOT + ln * 16;
      if (dma_lane) {
        // the lane's eight channels never change: one read of their scale / shift per tile
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(aff + pcc * 8), s1 = *reinterpret_cast<const f32x4*>(aff + pcc * 8 + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(aff + 64 + pcc * 8), h1 = *reinterpret_cast<const f32x4*>(aff + 64 + pcc * 8 + 4);
        // Straight-line code (no per-chunk branch: the scheduler interleaves the eight independent chunks, no dependent packed op
        // waits on its predecessor): all of the lane's chunks are requested first (one LDS round trip per tile; out-of-image chunks
        // hold the DMA's zeros and are read for nothing), transformed, and written back -- an out-of-image chunk's result goes to the
        // unused pad chunk of its pixel instead (the padding must stay zero).
        auto piece = [&](auto first, auto last) {
          constexpr int I0 = decltype(first)::value, I1 = decltype(last)::value;
          half8_t vin[I1 - I0];
#pragma unroll
          for (int i = I0; i < I1; ++i) vin[i - I0] = *reinterpret_cast<const half8_t*>(dst + (w4 * IPW + i) * L::PIECE_BYTES);
#pragma unroll
          for (int i = I0; i < I1; ++i) {
            const int n = w4 * IPW + i;
            const int prow = n / 3, pc0 = (n - prow * 3) * L::PIECE_PIX;
            const bool ok = (unsigned)(iy0 + prow) < (unsigned)p.H && (unsigned)(ix0 + pc0 + pl) < (unsigned)p.W;
            // two channels per instruction: packed fp32 multiply, packed fp32 add (unfused, bn_apply_k's arithmetic), one packed
            // f16 conversion, packed f16 max -- max(round(t), 0) = round(max(t, 0)): rounding is monotone and keeps zero
            const half8_t v = vin[i - I0];
            half8_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const f32x2 xv = {(float)v[2 * e], (float)v[2 * e + 1]};
              const f32x2 sc2 = e < 2 ? f32x2{s0[2 * e], s0[2 * e + 1]} : f32x2{s1[2 * e - 4], s1[2 * e - 3]};
              const f32x2 sh2 = e < 2 ? f32x2{h0[2 * e], h0[2 * e + 1]} : f32x2{h1[2 * e - 4], h1[2 * e - 3]};
              const f32x2 t = xv * sc2 + sh2;
              const half2_t r = __builtin_elementwise_max(__builtin_convertvector(t, half2_t), half2_t{(half_t)0.f, (half_t)0.f});
              o[2 * e] = r[0];
              o[2 * e + 1] = r[1];
            }
            char* slot = dst + n * L::PIECE_BYTES;
            *reinterpret_cast<half8_t*>(ok ? slot : slot + (8 - pcc) * 16) = o;
          }
        };
        piece(std::integral_constant<int, 0>{}, std::integral_constant<int, IPW - 2>{});  // pieces every wave has (4 x 8 - 2 = 30)
        if (w4 * IPW + IPW - 1 < L::NPIECE) piece(std::integral_constant<int, IPW - 2>{}, std::integral_constant<int, IPW>{});
      }
      return;
    }
    char* dst = gpatch + buf * PATCH_SLOT + w4 * (IPW * 1024);
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      if (w4 * IPW + i < NINST) {
        const int q = (w4 * IPW + i) * 64 + ln;
        const int pix = q / CPP, cc = q - pix * CPP;
        const int prow = pix / PW, pcol = pix - prow * PW;
        if (cc < 8 && pix < NPIX && (unsigned)(iy0 + prow) < (unsigned)p.H && (unsigned)(ix0 + pcol) < (unsigned)p.W) {
          half8_t* slot = reinterpret_cast<half8_t*>(dst + i * 1024 + ln * 16);
          const half8_t v = *slot;
          const f32x4 s0 = *reinterpret_cast<const f32x4*>(aff + cc * 8), s1 = *reinterpret_cast<const f32x4*>(aff + cc * 8 + 4);
          const f32x4 h0 = *reinterpret_cast<const f32x4*>(aff + 64 + cc * 8), h1 = *reinterpret_cast<const f32x4*>(aff + 64 + cc * 8 + 4);
          half8_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = (half_t)fmaxf((float)v[e] * s0[e] + h0[e], 0.f);
            o[4 + e] = (half_t)fmaxf((float)v[4 + e] * s1[e] + h1[e], 0.f);
          }
          *slot = o;
        }
      }
    }
  };

  // per-lane partial BN sums of the lane's 16 channels, folded across pixel lanes at the end (EPI: the lane's 16 bias values)
  // (M16: the lane's 8 channels are tn*32 + ca*16 + 4*kq + j -- registers r = ca*4 + j of the same vectors, the rest unused)
  f32x16 ssum, ssq;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int ch = M16 ? tn * 32 + ((r >> 2) & 1) * 16 + 4 * kq + (r & 3) : tn * 32 + 8 * (r >> 2) + 4 * kg + (r & 3);
    ssum[r] = (EPI && p.bias != nullptr) ? p.bias[ch] : 0.f;
    ssq[r] = 0.f;
  }
  const bool has_res = EPI && p.res != nullptr;
  const bool relu_late = EPI && p.relu && has_res, relu_early = EPI && p.relu && !has_res;

  // fragment origin of this lane inside a patch: pixel fragment tm covers rows hsel*4 + tm*2 + frow, cols fcol
  const int fbase = M16 ? ((hsel * 4) * PW + c16) * PP + kq * 16 : ((hsel * 4 + frow) * PW + fcol) * PP + kg * 16;
  char* stg = smem + PATCH_BYTES + wid * STG_WAVE;

  // Tile stream of this workgroup.  Workgroups are dispatched round-robin over the 8 XCDs (id & 7) and each XCD has its
  // own L2, so XCD x works through the contiguous tile range [x*chunk, (x+1)*chunk) -- its resident workgroups sweep it
  // side by side -- and the halo rows/columns shared by neighbouring tiles hit in that L2 instead of crossing the fabric
  // (the mapping only affects speed, never correctness).
  const int chunk = (p.ntiles + 7) >> 3, per_xcd = (int)(gridDim.x + 7) >> 3;
  const int xcd = blockIdx.x & 7;
  const int tend = min((xcd + 1) * chunk, p.ntiles);
  int tile = xcd * chunk + (int)(blockIdx.x >> 3);
  if (tile < tend) issue_patch(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (pre) {
    __syncthreads();  // the scale/shift table is in LDS
    if (tile < tend) transform_patch(tile, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (tile + per_xcd < tend) issue_patch(tile + per_xcd, 1);

  int buf = 0;
#ifdef AMP3_DIAG
  long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t_last = __builtin_readcyclecounter();
#endif
  for (; tile < tend; tile += per_xcd) {
    AMP3_STAMP(6);  // loop overhead / previous iteration's tail
    // ------------------------------- MFMA phase -------------------------------
    // 36 k16 steps (tap-major), software-pipelined by hand: the two pixel-fragment reads of step s+1 are issued before
    // the two MFMAs of step s.  The wait is the builtin so that hipcc's waitcnt pass sees it (by itself it emits
    // lgkmcnt(0) right AFTER the newest reads).
    const char* pt = gpatch + buf * PATCH_SLOT + fbase;
    f32x16 acc[2];  // [tm]
    half8_t fp[2][2];
    f32x4 acc4[4][2];  // M16: [pixel row pb][16-channel half ca]
    half8_t fq[2][4];  // M16: [parity][pb]
    rsrc_words_t resv[4];
    if (EPI && has_res) {  // the store loop's pieces of the residual tile: in flight under the MFMA phase
      const int img = tile / tiles_per_img;
      const int rem = tile - img * tiles_per_img;
      const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int q = it * 64 + lane;
        const int prow = q >> 2, cc = q & 3;
        const int oy = ty * TH + hsel * 4 + (prow >> 5) * 2 + ((prow >> 4) & 1), ox = tx * TW + (prow & 15);
        const unsigned vo = (oy < p.H && ox < p.W) ? (unsigned)((((img * p.H + oy) * p.W + ox) * p.ldo + p.y_coff + tn * 32 + cc * 8) * 2) : 0x80000000u;
        resv[it] = buffer_load16(p.res, p.y_bytes, vo);
      }
    }
    if constexpr (M16 && AMP3_ROWREUSE) {
      // Row-reuse schedule: the wave's four output rows x three vertical taps touch SIX patch rows, and the fragment of patch row R
      // (for one horizontal tap kw and one 32-channel half ks2) serves every (output row pb, vertical tap kh) with pb + kh = R.  So
      // per (kw, ks2) the six row fragments are read ONCE (36 reads per tile instead of 72: the LDS port, which the fragment reads
      // of eight waves saturate exactly when the MFMA pipes are busy, carries half the bytes) and a sliding window of four of them
      // feeds the eight MFMAs of each kh step.  Eight fragment registers: fragment q = 6 * j + R lives in register q & 7; the four
      // reads of (j, kh = 0) fetch rows 4-5 and the next j's rows 0-1, kh = 1 / 2 one row each -- every read is issued at least one
      // step (128 MFMA cycles) before its first use, the compiler's counted lgkmcnt waits leave the younger ones in flight.
      // Accumulation order per output: (kw, ks2) outer, kh inner.
      half8_t fr[8];
      auto frag = [&](int q) {
        const int j = q / 6, R = q - j * 6, kw = j >> 1, ks2 = j & 1;
        return *reinterpret_cast<const half8_t*>(pt + (R * PW + kw) * PP + ks2 * 64);
      };
      if constexpr (!(AMP3_ABL & 4)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) fr[q] = frag(q);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 18; ++s) {
        const int j = s / 3, kh = s - j * 3, kw = j >> 1, ks2 = j & 1;
        if constexpr (!(AMP3_ABL & 4)) {
          if (kh == 0) {
#pragma unroll
            for (int q = 6 * j + 4; q < 6 * j + 8; ++q)
              if (q < 36) fr[q & 7] = frag(q);
          } else if (6 * j + 7 + kh < 36) {
            fr[(6 * j + 7 + kh) & 7] = frag(6 * j + 7 + kh);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if constexpr (!(AMP3_ABL & 2)) {
#pragma unroll
          for (int pb = 0; pb < 4; ++pb)
#pragma unroll
            for (int ca = 0; ca < 2; ++ca)
              acc4[pb][ca] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[((kh * 3 + kw) * 2 + ks2) * 2 + ca], fr[(6 * j + kh + pb) & 7],
                                                                    s == 0 ? z : acc4[pb][ca], 0, 0, 0);
        } else {
#pragma unroll
          for (int pb = 0; pb < 4; ++pb) {
            asm volatile("" ::"v"(fr[(6 * j + kh + pb) & 7]));
            if (s == 0) { acc4[pb][0] = z; acc4[pb][1] = z; }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (M16 && AMP3_FRING > 0) {
      // 72 pixel-row fragments (18 k32 steps x 4 rows, tap-major), each feeding two MFMAs (the wave's two 16-channel halves), through a
      // rolling ring of eight fragment registers: fragment f + AMP3_FRING is requested right before the MFMAs of fragment f, so a read
      // has AMP3_FRING x 32 MFMA cycles to come back (the step-wise form below: 128) -- the ablation showed the MFMA phase waiting on
      // its fragment reads (146 -> 98 us without them).  LDS returns a wave's reads in order: the compiler's counted lgkmcnt waits
      // leave the younger reads in flight.  Accumulation order per accumulator unchanged (bit-identical results).
      constexpr int NF = 72, D = AMP3_FRING;
      static_assert(D >= 1 && D <= 7, "ring of eight registers");
      half8_t fr[8];
      auto frag = [&](int f) {
        const int s = f >> 2, pb = f & 3, tap = s >> 1, ks2 = s & 1, kh = tap / 3, kw = tap - kh * 3;
        return *reinterpret_cast<const half8_t*>(pt + ((pb + kh) * PW + kw) * PP + ks2 * 64);
      };
      if constexpr (!(AMP3_ABL & 4)) {
#pragma unroll
        for (int f = 0; f < D; ++f) fr[f] = frag(f);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (f + D < NF && !(AMP3_ABL & 4)) fr[(f + D) & 7] = frag(f + D);
        __builtin_amdgcn_sched_barrier(0);
        const int s = f >> 2, pb = f & 3;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if constexpr (!(AMP3_ABL & 2)) {
#pragma unroll
          for (int ca = 0; ca < 2; ++ca)
            acc4[pb][ca] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[s * 2 + ca], fr[f & 7], s == 0 ? z : acc4[pb][ca], 0, 0, 0);
        } else {
          asm volatile("" ::"v"(fr[f & 7]));
          if (s == 0) { acc4[pb][0] = z; acc4[pb][1] = z; }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (M16) {
      // 18 k32 steps (tap-major, two per tap): the four pixel-row fragments of step s+1 are requested before the eight MFMAs of step s
#pragma unroll
      for (int s = 0; s < 18 + 1; ++s) {
        if (s >= 1) __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): step s-1's fragments
        __builtin_amdgcn_sched_barrier(0);
        if (s < 18 && !(AMP3_ABL & 4)) {
          const int tap = s >> 1, ks2 = s & 1, kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
          for (int pb = 0; pb < 4; ++pb)
            fq[s & 1][pb] = *reinterpret_cast<const half8_t*>(pt + ((pb + kh) * PW + kw) * PP + ks2 * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s >= 1) {
          const int c = (s - 1) & 1;
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          if constexpr (!(AMP3_ABL & 2)) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb)
#pragma unroll
              for (int ca = 0; ca < 2; ++ca)
                acc4[pb][ca] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[(s - 1) * 2 + ca], fq[c][pb], s == 1 ? z : acc4[pb][ca], 0, 0, 0);
          } else {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
              asm volatile("" ::"v"(fq[c][pb]));
              if (s == 1) { acc4[pb][0] = z; acc4[pb][1] = z; }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 36 + 1; ++s) {
        if (s >= 1) __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): step s-1's fragments (issued one MFMA pair ago)
        __builtin_amdgcn_sched_barrier(0);
        if (s < 36) {
          const int tap = s >> 2, ks = s & 3, kh = tap / 3, kw = tap - kh * 3;
  #pragma unroll
          for (int tm = 0; tm < 2; ++tm)
            fp[s & 1][tm] = *reinterpret_cast<const half8_t*>(pt + ((tm * 2 + kh) * PW + kw) * PP + ks * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s >= 1) {
          const int c = (s - 1) & 1;
          if (s == 1) {  // first step starts from the constant zero: no accumulator clears
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  #pragma unroll
            for (int tm = 0; tm < 2; ++tm) acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0], fp[c][tm], z, 0, 0, 0);
          } else {
  #pragma unroll
            for (int tm = 0; tm < 2; ++tm) acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s - 1], fp[c][tm], acc[tm], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  
    }

    // This wave's share of the next patch (issued a whole epilogue + MFMA phase ago) and the previous tile's stores are
    // done; after the barrier the next patch is complete and nobody reads the current one any more -- so the patch after
    // next goes into the buffer just read, two tiles ahead of its use (an LDS-DMA from HBM takes longer than one MFMA
    // phase under load).  Raw barrier + asm wait: __syncthreads() would do, the explicit form documents what is ordered.
    AMP3_STAMP(0);  // MFMA phase (issue)
#if AMP3_VMCNT4
    // the four stores of the previous tile's epilogue are the wave's youngest vector-memory operations: leave them in flight
    // (the next patch's pieces, issued before them, have landed once only four remain outstanding)
    if (EPI && has_res) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the residual loads of this tile are younger still)
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    AMP3_STAMP(1);  // wait for the next patch's pieces
    if (pre && tile + per_xcd < tend) {
      transform_patch(tile + per_xcd, buf ^ 1);  // this wave's share of the next patch has landed: rewrite it in place
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    AMP3_STAMP(2);  // input transform
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    AMP3_STAMP(3);  // barrier
    if (!(AMP3_ABL & 1) && tile + 2 * per_xcd < tend) issue_patch(tile + 2 * per_xcd, buf);
    buf ^= 1;
    AMP3_STAMP(4);  // patch issue

    // ------------------------------- epilogue -------------------------------
    // acc[tm][r] = out(pixel (row hsel*4 + tm*2 + frow, col fcol), channel tn*32 + 8*(r>>2) + 4*kg + (r&3))
    const int img = tile / tiles_per_img;
    const int rem = tile - img * tiles_per_img;
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    if constexpr (M16) {
      // acc4[pb][ca][j] = out(pixel (row hsel*4 + pb, col c16), channel tn*32 + ca*16 + 4*kq + j); ssum / ssq registers ca*4 + j
      if (!EPI && (ty * TH + TH > p.H || tx * TW + TW > p.W)) {
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
          const bool ok = ty * TH + hsel * 4 + pb < p.H && tx * TW + c16 < p.W;
#pragma unroll
          for (int ca = 0; ca < 2; ++ca)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc4[pb][ca][j] = ok ? acc4[pb][ca][j] : 0.f;
        }
      }
#pragma unroll
      for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int ca = 0; ca < 2; ++ca)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (EPI) {  // ssum holds the bias
              acc4[pb][ca][j] += ssum[ca * 4 + j];
              if (relu_early) acc4[pb][ca][j] = fmaxf(acc4[pb][ca][j], 0.f);
            } else {
              ssum[ca * 4 + j] += acc4[pb][ca][j];
              ssq[ca * 4 + j] = __builtin_fmaf(acc4[pb][ca][j], acc4[pb][ca][j], ssq[ca * 4 + j]);
            }
          }
      // stage 64 pixels x 32 channels in this wave's own area (pixel pb*16 + c16: the store loop's row / column split)
#pragma unroll
      for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int ca = 0; ca < 2; ++ca)
          *reinterpret_cast<half4_t*>(stg + (pb * 16 + c16) * SP + (ca * 16 + 4 * kq) * 2) = __builtin_convertvector(acc4[pb][ca], half4_t);
    } else {
    if (!EPI && (ty * TH + TH > p.H || tx * TW + TW > p.W)) {
        // edge tile: pixels outside the image are not conv outputs -- zero them so they stay out of the statistics
  #pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          const bool ok = ty * TH + hsel * 4 + tm * 2 + frow < p.H && tx * TW + fcol < p.W;
  #pragma unroll
          for (int r = 0; r < 16; ++r) acc[tm][r] = ok ? acc[tm][r] : 0.f;
        }
      }
      if (EPI) {  // ssum holds the bias
  #pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          acc[tm] += ssum;
          if (relu_early) acc[tm] = __builtin_elementwise_max(acc[tm], ssq);  // (ssq stays zero)
        }
      } else {
  #pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          ssum += acc[tm];
          ssq = __builtin_elementwise_fma(acc[tm], acc[tm], ssq);
        }
      }
      // stage 64 pixels x 32 channels in this wave's own area, then 64-byte half-rows go out with 16-byte stores
  #pragma unroll
      for (int tm = 0; tm < 2; ++tm)
  #pragma unroll
        for (int jq = 0; jq < 4; ++jq) {
          const f32x4 v = {acc[tm][4 * jq], acc[tm][4 * jq + 1], acc[tm][4 * jq + 2], acc[tm][4 * jq + 3]};
          *reinterpret_cast<half4_t*>(stg + (tm * 32 + frow * 16 + fcol) * SP + (8 * jq + 4 * kg) * 2) = __builtin_convertvector(v, half4_t);
        }
    }
    // the wave reads back what its own lanes wrote (LDS executes a wave's accesses in order); the asm also keeps the
    // compiler from moving the differently typed reads above the writes
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int q = it * 64 + lane;
      const int prow = q >> 2, cc = q & 3;  // prow = tm*32 + pixel
      const int oy = ty * TH + hsel * 4 + (prow >> 5) * 2 + ((prow >> 4) & 1), ox = tx * TW + (prow & 15);
      const unsigned vo = (oy < p.H && ox < p.W) ? (unsigned)((((img * p.H + oy) * p.W + ox) * p.ldo + p.y_coff + tn * 32 + cc * 8) * 2) : 0x80000000u;
      rsrc_words_t v = *reinterpret_cast<const rsrc_words_t*>(stg + prow * SP + cc * 16);
      if (EPI && has_res) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (int)am_addh2_act((unsigned)v[e], (unsigned)resv[it][e], relu_late);
      }
      if constexpr (!(AMP3_ABL & 8)) buffer_store16_asm(v, yr, vo);
      else asm volatile("" ::"v"(v), "v"(vo));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging reads done before the next tile's writes
    AMP3_STAMP(5);  // epilogue
#ifdef AMP3_DIAG
    dg[7] += 1;
#endif
  }
#ifdef AMP3_DIAG
  if (blockIdx.x == 0 && tid == 0)
    for (int k = 0; k < 8; ++k) g_duo_diag[k] = dg[k];
#endif

  if (!EPI && p.stats != nullptr) {
    // fold the 32 pixel lanes of each half-wave (xor < 32 stays inside the half), then the 4 waves that share a channel
    // block -> LDS -> one fp64 atomic per channel per workgroup
    float* part = reinterpret_cast<float*>(smem);  // [4 waves][32 channels][2] in the (now idle) patch area
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int r = 0; r < (M16 ? 8 : 16); ++r) {
      float sv = ssum[r], qv = ssq[r];
#pragma unroll
      for (int o = (M16 ? 8 : 16); o > 0; o >>= 1) {  // over the lanes that share the channels: 16 pixel columns / 32 pixels
        sv += __shfl_xor(sv, o, 64);
        qv += __shfl_xor(qv, o, 64);
      }
      if (M16 ? c16 == 0 : rx == 0) {
        const int ch = M16 ? (r >> 2) * 16 + 4 * kq + (r & 3) : 8 * (r >> 2) + 4 * kg + (r & 3);
        part[(wid * 32 + ch) * 2 + 0] = sv;
        part[(wid * 32 + ch) * 2 + 1] = qv;
      }
    }
    __syncthreads();
    if (tid < 64) {
      const int ctn = tid >> 5, ch = tid & 31;
      double s = 0.0, q = 0.0;
      for (int h = 0; h < 2; ++h) {
        const int ww = h * 2 + ctn;
        s += (double)part[(ww * 32 + ch) * 2 + 0];
        q += (double)part[(ww * 32 + ch) * 2 + 1];
      }
      double* st = p.stats + (size_t)(blockIdx.x % AM_STATS_REPLICAS) * 2 * 64;
      atomicAdd(st + tid, s);
      atomicAdd(st + 64 + tid, q);
    }
  }
}

}  // namespace amp3

// Returns AM_ERR_UNSUPPORTED unless the geometry is exactly a dense 3x3 / stride 1 / pad 1, 64 -> 64 f16 convolution
// (forward packing, tap order kh-major) over a tensor small enough for 30-bit offsets; with a bias / ReLU / residual epilogue
// (the inference form) there are no statistics and no input transform.
int am_conv3x3_c64n64_duo_pre_f16(const am_conv_geom* g, const void* x, const float* pre_scale, const float* pre_shift, const void* w,
                                  const float* bias, int relu, const void* res, void* y, double* stats, hipStream_t s);

// res (may be null): residual tensor with y's geometry, y = act(conv + bias + res)
int am_conv3x3_c64n64_duo_f16(const am_conv_geom* g, const void* x, const void* w, const float* bias, int relu, const void* res, void* y,
                              double* stats, hipStream_t s) {
  return am_conv3x3_c64n64_duo_pre_f16(g, x, nullptr, nullptr, w, bias, relu, res, y, stats, s);
}

// pre_scale / pre_shift (both or neither): the convolution runs on relu(x * pre_scale[c] + pre_shift[c]).
int am_conv3x3_c64n64_duo_pre_f16(const am_conv_geom* g, const void* x, const float* pre_scale, const float* pre_shift, const void* w,
                                  const float* bias, int relu, const void* res, void* y, double* stats, hipStream_t s) {
  using namespace amp3;
  if (g->ntaps != 9 || g->krun != 64 || g->N != 64 || g->pix_shift != 31) return AM_ERR_UNSUPPORTED;
  if (g->iys != 1 || g->ixs != 1 || g->oys != 1 || g->oxs != 1 || g->oy0 != 0 || g->ox0 != 0) return AM_ERR_UNSUPPORTED;
  if (g->MH != g->IH || g->MW != g->IW || g->OH != g->IH || g->OW != g->IW) return AM_ERR_UNSUPPORTED;
  for (int t = 0; t < 9; ++t)
    if (g->dy[t] != t / 3 - 1 || g->dx[t] != t % 3 - 1) return AM_ERR_UNSUPPORTED;
  if (g->IW < TW || (long long)g->B * g->IH * g->IW < 64 * 1024) return AM_ERR_UNSUPPORTED;  // small problems: gather-GEMM
  const long long x_bytes = (long long)g->B * g->IH * g->IW * g->ldi * 2;
  const long long y_bytes = (long long)g->B * g->OH * g->OW * g->ldo * 2;
  if (x_bytes >= (1ll << 30) || y_bytes >= (1ll << 31)) return AM_ERR_UNSUPPORTED;
  const bool epi = bias != nullptr || relu || res != nullptr;
  if (epi && (stats != nullptr || pre_scale != nullptr)) return AM_ERR_UNSUPPORTED;  // a bias / ReLU / residual epilogue is the inference form: no statistics
  Patch3Params p;
  p.x = x; p.w = w; p.y = y; p.stats = stats;
  p.B = g->B; p.H = g->IH; p.W = g->IW; p.ldi = g->ldi; p.x_coff = g->x_coff; p.ldo = g->ldo; p.y_coff = g->y_coff;
  p.tiles_y = am_cdiv(g->IH, TH);
  p.tiles_x = am_cdiv(g->IW, TW);
  p.ntiles = p.B * p.tiles_y * p.tiles_x;
  p.x_bytes = (unsigned)x_bytes;
  p.w_bytes = 64 * WROW;
  p.pre_scale = pre_scale; p.pre_shift = pre_shift;
  p.y_bytes = (unsigned)y_bytes;
  p.bias = bias; p.res = res; p.relu = relu;
  const bool m16 = am_tuning(AM_TUNE_DUO_MFMA16) != 0;
  const int lds = m16 ? Lay<true>::LDS_BYTES : Lay<false>::LDS_BYTES;
  static bool attr_done_dev[AM_MAX_DEVICES][4] = {};
  bool& attr_done = attr_done_dev[am_current_device()][(epi ? 1 : 0) + (m16 ? 2 : 0)];
  if (!attr_done) {
    const void* fn = m16 ? (epi ? reinterpret_cast<const void*>(conv3x3_c64n64_duo_k<true, true>) : reinterpret_cast<const void*>(conv3x3_c64n64_duo_k<false, true>))
                         : (epi ? reinterpret_cast<const void*>(conv3x3_c64n64_duo_k<true, false>) : reinterpret_cast<const void*>(conv3x3_c64n64_duo_k<false, false>));
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return AM_ERR_LAUNCH;
    attr_done = true;
  }
  const int grid = p.ntiles < 512 ? ((p.ntiles + 7) & ~7) : 512;  // two persistent workgroups per CU, a multiple of the 8 XCDs
  g_am_conv_variant = AM_CV_DUO_C64;
  if (m16) {
    if (epi) hipLaunchKernelGGL((conv3x3_c64n64_duo_k<true, true>), dim3(grid), dim3(256), lds, s, p);
    else hipLaunchKernelGGL((conv3x3_c64n64_duo_k<false, true>), dim3(grid), dim3(256), lds, s, p);
  } else {
    if (epi) hipLaunchKernelGGL((conv3x3_c64n64_duo_k<true, false>), dim3(grid), dim3(256), lds, s, p);
    else hipLaunchKernelGGL((conv3x3_c64n64_duo_k<false, false>), dim3(grid), dim3(256), lds, s, p);
  }
  AM_CHECK_LAUNCH();
  return AM_OK;
}
