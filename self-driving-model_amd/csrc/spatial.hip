// HBM-bound spatial kernels of the expert path: layout conversion, max-pool, global average pool,
// bilinear upsample, pixel-wise cross-entropy.  All coalesced, 16 B per lane where the layout allows.
#include "am_common.h"

namespace {

inline int ew_grid(long long total_threads) {
  long long b = (total_threads + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- NCHW fp32 -> NHWC T with channel padding (zeros) --------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_k(const float* __restrict__ src, T* __restrict__ dst, int C, long long HW,
                                                      int ld, long long total_pix, float mul) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total_pix; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / HW, p = i - b * HW;
    const float* s = src + b * C * HW + p;
    T* d = dst + i * ld;
    constexpr int E = 16 / (int)sizeof(T);
    for (int c0 = 0; c0 < ld; c0 += E) {  // ld is a multiple of E: one 16-byte store per chunk
      uint4 raw;
      T* o = reinterpret_cast<T*>(&raw);
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] = am_from_f32<T>(c0 + e < C ? s[(long long)(c0 + e) * HW] * mul : 0.f);
      *reinterpret_cast<uint4*>(d + c0) = raw;
    }
  }
}

// ---- image boundary: NCHW fp32 [B,C<=4,H,W] (H, W even) -> space-to-depth(2) NHWC [B,H/2,W/2,16] ------------------
// channel (py*2+px)*C + c of output pixel (Y,X) = img[b,c,2Y+py,2X+px]; channels 4*C..15 are zero.  A stride-2 KxK
// first-layer conv on the image is a stride-1 ceil(K/2)+... conv on this tensor with 16 input channels, which gives the
// MFMA kernels 32-byte (f16) pixels instead of 3 useful channels out of 8.
template <typename T>
__global__ __launch_bounds__(256) void image_s2d_k(const float* __restrict__ src, T* __restrict__ dst, int C, int H, int W,
                                                   long long total) {
  const int H2 = H / 2, W2 = W / 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int X = (int)(i % W2);
    long long t = i / W2;
    const int Y = (int)(t % H2);
    const long long b = t / H2;
    T out[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) out[e] = am_from_f32<T>(0.f);
    for (int c = 0; c < C; ++c) {
      const float* pl = src + ((b * C + c) * H + 2 * Y) * (long long)W + 2 * X;
      const float2 r0 = *reinterpret_cast<const float2*>(pl);
      const float2 r1 = *reinterpret_cast<const float2*>(pl + W);
      out[0 * C + c] = am_from_f32<T>(r0.x);
      out[1 * C + c] = am_from_f32<T>(r0.y);
      out[2 * C + c] = am_from_f32<T>(r1.x);
      out[3 * C + c] = am_from_f32<T>(r1.y);
    }
    uint4* d = reinterpret_cast<uint4*>(dst + i * 16);
    const uint4* o = reinterpret_cast<const uint4*>(out);
#pragma unroll
    for (int e = 0; e < (int)(16 * sizeof(T) / 16); ++e) d[e] = o[e];
  }
}

// Two adjacent space-to-depth pixels per thread (W/2 even, C == 3): 16-byte loads, 64 contiguous output bytes, twice the
// bytes in flight per wave -- the pass is a pure copy and should run at HBM speed.
template <typename T>
__global__ __launch_bounds__(256) void image_s2d_x2_k(const float* __restrict__ src, T* __restrict__ dst, int H, int W, long long total2) {
  const int H2 = H / 2, W4 = W / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total2; i += (long long)gridDim.x * blockDim.x) {
    const int X = (int)(i % W4);
    long long t = i / W4;
    const int Y = (int)(t % H2);
    const long long b = t / H2;
    float4 r[3][2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* pl = src + ((b * 3 + c) * H + 2 * Y) * (long long)W + 4 * X;
      r[c][0] = *reinterpret_cast<const float4*>(pl);
      r[c][1] = *reinterpret_cast<const float4*>(pl + W);
    }
    T out[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) out[e] = am_from_f32<T>(0.f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      out[0 * 3 + c] = am_from_f32<T>(r[c][0].x);
      out[1 * 3 + c] = am_from_f32<T>(r[c][0].y);
      out[2 * 3 + c] = am_from_f32<T>(r[c][1].x);
      out[3 * 3 + c] = am_from_f32<T>(r[c][1].y);
      out[16 + 0 * 3 + c] = am_from_f32<T>(r[c][0].z);
      out[16 + 1 * 3 + c] = am_from_f32<T>(r[c][0].w);
      out[16 + 2 * 3 + c] = am_from_f32<T>(r[c][1].z);
      out[16 + 3 * 3 + c] = am_from_f32<T>(r[c][1].w);
    }
    uint4* d = reinterpret_cast<uint4*>(dst + i * 32);
    const uint4* o = reinterpret_cast<const uint4*>(out);
#pragma unroll
    for (int e = 0; e < (int)(32 * sizeof(T) / 16); ++e) d[e] = o[e];
  }
}

// Same boundary for raw camera frames: uint8 NCHW -> (u/255 - mean[c]) / std[c] in fp32 (the reference's loader arithmetic:
// read_image(...).float() / 255.0, then torchvision Normalize = sub mean, div std; bdd_detection_loader.py:54,
// train_bdd100k_ddp.py:471-473) -> space-to-depth NHWC.  One pass, a quarter of the bytes of the fp32 image on the way in.
template <typename T>
__global__ __launch_bounds__(256) void image_u8_s2d_k(const uint8_t* __restrict__ src, T* __restrict__ dst, int C, int H, int W,
                                                      long long total, float m0, float m1, float m2, float m3, float s0, float s1,
                                                      float s2, float s3, int normalize) {
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;  // odd sizes: the missing row / column is zero AFTER the preprocessing
  const float mean[4] = {m0, m1, m2, m3}, stdv[4] = {s0, s1, s2, s3};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int X = (int)(i % W2);
    long long t = i / W2;
    const int Y = (int)(t % H2);
    const long long b = t / H2;
    T out[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) out[e] = am_from_f32<T>(0.f);
    for (int c = 0; c < C; ++c) {
      const uint8_t* pl = src + (b * C + c) * (long long)H * W;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int y = 2 * Y + (k >> 1), x = 2 * X + (k & 1);
        if (y < H && x < W) {
          float v = (float)pl[(long long)y * W + x] / 255.0f;
          if (normalize) v = (v - mean[c]) / stdv[c];
          out[k * C + c] = am_from_f32<T>(v);
        }
      }
    }
    uint4* d = reinterpret_cast<uint4*>(dst + i * 16);
    const uint4* o = reinterpret_cast<const uint4*>(out);
#pragma unroll
    for (int e = 0; e < (int)(16 * sizeof(T) / 16); ++e) d[e] = o[e];
  }
}

// ---- NHWC T -> NCHW fp32 (first C channels) -------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_k(const T* __restrict__ src, float* __restrict__ dst, int C, long long HW,
                                                      int ld, long long total, float mul) {
  // one thread per output element, NCHW order: coalesced writes, strided (L2-absorbed) reads
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i % HW;
    const long long bc = i / HW;
    const int c = (int)(bc % C);
    const long long b = bc / C;
    dst[i] = am_to_f32(src[(b * HW + p) * ld + c]) * mul;
  }
}

// ---- max-pool 3x3 stride 2 pad 1, NHWC ------------------------------------------------------
// scale / shift (am_bn_relu_maxpool3x3s2_fwd): x is a raw conv output and the pooled tensor is MaxPool(relu(x * scale + shift)), the
// activation rounded to T before the comparison exactly as am_bn_apply would have stored it -- the normalised map (the largest
// activation of a ResNet: 472 MB at B = 16) is neither written nor re-read
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_k(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ arg,
                                                     int B, int IH, int IW, int OH, int OW, int C, const float* __restrict__ scale,
                                                     const float* __restrict__ shift) {
  constexpr int E = 16 / (int)sizeof(T);
  const int cpr = C / E;
  const long long total = (long long)B * OH * OW * cpr;
  const bool bn = scale != nullptr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    // (32-bit index arithmetic: the launcher checks total < 2^31; four 64-bit divisions per iteration cost more than the pooling)
    const unsigned iu = (unsigned)i;
    const int ch = (int)(iu % (unsigned)cpr);
    float sc[E], sh[E];
    if (bn) {
#pragma unroll
      for (int e = 0; e < E; e += 4) {
        *reinterpret_cast<f32x4*>(sc + e) = *reinterpret_cast<const f32x4*>(scale + ch * E + e);
        *reinterpret_cast<f32x4*>(sh + e) = *reinterpret_cast<const f32x4*>(shift + ch * E + e);
      }
    }
    unsigned t = iu / (unsigned)cpr;
    const int ox = (int)(t % (unsigned)OW); t /= (unsigned)OW;
    const int oy = (int)(t % (unsigned)OH);
    const int b = (int)(t / (unsigned)OH);
    float best[E];
    int bi[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    // all nine window loads first, at clamped (always valid) addresses, then the comparisons: with the bounds tests as branches
    // around each load the taps ran one memory round trip after the other (3.0 TB/s on the stem map)
    uint4 win[9];
    bool ok[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = oy * 2 - 1 + kh;
      const int iyc = min(max(iy, 0), IH - 1);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ix = ox * 2 - 1 + kw;
        const int ixc = min(max(ix, 0), IW - 1);
        ok[kh * 3 + kw] = (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
        win[kh * 3 + kw] = *reinterpret_cast<const uint4*>(x + (((long long)b * IH + iyc) * IW + ixc) * C + ch * E);
      }
    }
    bool first = true;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if (!ok[k]) continue;
      const T* v = reinterpret_cast<const T*>(&win[k]);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        float f = am_to_f32(v[e]);
        if (bn) f = am_to_f32(am_from_f32<T>(fmaxf(f * sc[e] + sh[e], 0.f)));
        // first maximum in (kh,kw) scan order wins, NaN propagates (torch max_pool2d)
        if (first || f > best[e] || f != f) { best[e] = f; bi[e] = k; }
      }
      first = false;
    }
    uint4 outraw;
    T* o = reinterpret_cast<T*>(&outraw);
#pragma unroll
    for (int e = 0; e < E; ++e) o[e] = am_from_f32<T>(best[e]);
    const long long ooff = (((long long)b * OH + oy) * OW + ox) * C + ch * E;
    *reinterpret_cast<uint4*>(y + ooff) = outraw;
    if (arg) {  // the E codes of this chunk as ONE store (eight byte stores per thread were the kernel's bottleneck: 1.8 TB/s)
      if constexpr (E == 8) {
        uint2 packed;
        packed.x = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
        packed.y = (unsigned)bi[4] | ((unsigned)bi[5] << 8) | ((unsigned)bi[6] << 16) | ((unsigned)bi[7] << 24);
        *reinterpret_cast<uint2*>(arg + ooff) = packed;
      } else {
        *reinterpret_cast<unsigned*>(arg + ooff) = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
      }
    }
  }
}

// Block form of the same gather (what am_maxpool3x3s2_bwd launches): a thread owns the 2x2 block of input pixels
// (2Y + py, 2X + px) of one 16-byte channel chunk.  Window (oy, ox) covers rows 2oy-1 .. 2oy+1, so the block is touched by
// exactly the four windows (Y + wy, X + wx), wy, wx in {0, 1}, and pixel (py, px) sits in window (wy, wx) at kh = py + 1 - 2wy,
// kw = px + 1 - 2wx (valid when 0 <= kh, kw <= 2: an even row / column belongs to one window only).  Four dY loads, four
// arg-max loads, four stores per thread, no data-dependent control flow: 422 -> ~2x faster than one pixel per thread on the stem
// map ([16,360,640,64]: 650 MB of algorithmic traffic).  Contributions are summed in the order of the per-pixel form.
// BNR (am_maxpool3x3s2_bwd_bn): the pooled layer is conv -> BatchNorm -> ReLU (the ResNet stem); while the gradient of a pixel is in
// registers the kernel also reads the raw conv output there and accumulates the BatchNorm-backward sums (sum dz, sum dz * xhat with
// dz = dx masked by the sign of raw * scale + shift: bn.hip bn_bwd_reduce_k's arithmetic on the ROUNDED gradient it would have
// read), so the separate reduce pass over the two full-resolution tensors is not run.  Needs 256 % (C / E) == 0: a thread keeps its
// channel chunk.
template <typename T, bool BNR>
__global__ __launch_bounds__(256) void maxpool_bwd_block_k(const T* __restrict__ dy, const uint8_t* __restrict__ arg, T* __restrict__ dx,
                                                           int B, int IH, int IW, int OH, int OW, int C, const T* __restrict__ raw,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ sg_scale, const float* __restrict__ sg_shift,
                                                           double* __restrict__ sums) {
  constexpr int E = 16 / (int)sizeof(T);
  const int cpr = C / E;
  const int BH = (IH + 1) / 2, BW = (IW + 1) / 2;
  const long long total = (long long)B * BH * BW * cpr;
  using S = typename am_stat_acc<T>::type;  // double in fp32 parity mode (am_common.h)
  S bs[E], bq[E];
  float mu[E], rs[E], ssc[E], ssh[E];
  if (BNR) {
    const int c0 = (int)(threadIdx.x % cpr) * E;  // == (i % cpr) * E for every i of this thread (grid stride is a multiple of 256)
#pragma unroll
    for (int e = 0; e < E; ++e) {
      bs[e] = bq[e] = (S)0;
      mu[e] = mean[c0 + e]; rs[e] = rstd[c0 + e]; ssc[e] = sg_scale[c0 + e]; ssh[e] = sg_shift[c0 + e];
    }
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const unsigned iu = (unsigned)i;  // (32-bit index arithmetic: the launcher checks total < 2^31)
    const int ch = (int)(iu % (unsigned)cpr);
    unsigned t = iu / (unsigned)cpr;
    const int X = (int)(t % (unsigned)BW); t /= (unsigned)BW;
    const int Y = (int)(t % (unsigned)BH);
    const int b = (int)(t / (unsigned)BH);
    float g[2][2][E];
    uint8_t a[2][2][E];
#pragma unroll
    for (int wy = 0; wy < 2; ++wy)
#pragma unroll
      for (int wx = 0; wx < 2; ++wx) {
        const int oy = Y + wy, ox = X + wx;
        if (oy < OH && ox < OW) {
          const long long ooff = (((long long)b * OH + oy) * OW + ox) * C + ch * E;
          const uint4 raw = *reinterpret_cast<const uint4*>(dy + ooff);
          const T* v = reinterpret_cast<const T*>(&raw);
#pragma unroll
          for (int e = 0; e < E; ++e) g[wy][wx][e] = am_to_f32(v[e]);
          if constexpr (E == 8) *reinterpret_cast<uint2*>(a[wy][wx]) = *reinterpret_cast<const uint2*>(arg + ooff);
          else *reinterpret_cast<unsigned*>(a[wy][wx]) = *reinterpret_cast<const unsigned*>(arg + ooff);
        } else {
#pragma unroll
          for (int e = 0; e < E; ++e) { g[wy][wx][e] = 0.f; a[wy][wx][e] = 255; }
        }
      }
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      const int iy = 2 * Y + py;
      if (iy >= IH) continue;
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        const int ix = 2 * X + px;
        if (ix >= IW) continue;
        float acc[E];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.f;
        // per-pixel form order: kh ascending = window row Y+1 (kh = 0) before Y (kh = 1 or 2); kw likewise
#pragma unroll
        for (int wy = 1; wy >= 0; --wy) {
          const int kh = py + 1 - 2 * wy;
          if (kh < 0) continue;
#pragma unroll
          for (int wx = 1; wx >= 0; --wx) {
            const int kw = px + 1 - 2 * wx;
            if (kw < 0) continue;
#pragma unroll
            for (int e = 0; e < E; ++e)
              if (a[wy][wx][e] == kh * 3 + kw) acc[e] += g[wy][wx][e];
          }
        }
        uint4 outraw;
        T* o = reinterpret_cast<T*>(&outraw);
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = am_from_f32<T>(acc[e]);
        const long long xoff = (((long long)b * IH + iy) * IW + ix) * C + ch * E;
        *reinterpret_cast<uint4*>(dx + xoff) = outraw;
        if (BNR) {
          const uint4 rraw = *reinterpret_cast<const uint4*>(raw + xoff);
          const T* rv = reinterpret_cast<const T*>(&rraw);
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const float xf = am_to_f32(rv[e]);
            float dz = am_to_f32(o[e]);
            if (!(xf * ssc[e] + ssh[e] > 0.f)) dz = 0.f;
            bs[e] += (S)dz;
            if constexpr (sizeof(S) == 8) bq[e] += (double)dz * ((double)xf - (double)mu[e]) * (double)rs[e];
            else bq[e] += dz * (xf - mu[e]) * rs[e];
          }
        }
      }
    }
  }
  if (BNR) {
    // block reduction over the threads that share a channel chunk, then fp64 atomics into the replicas (as bn_bwd_reduce_k)
    extern __shared__ char mp_red_raw[];
    S* red = reinterpret_cast<S*>(mp_red_raw);  // [256][2E]
    const int tid = threadIdx.x, rpp = 256 / cpr;
#pragma unroll
    for (int e = 0; e < E; ++e) { red[tid * 2 * E + e] = bs[e]; red[tid * 2 * E + E + e] = bq[e]; }
    __syncthreads();
    for (int o = tid; o < cpr * E * 2; o += 256) {
      const int which = o / (cpr * E), ce = o % (cpr * E);
      const int chn = ce / E, e = ce % E;
      double a = 0.0;
      for (int r = 0; r < rpp; ++r) a += (double)red[(r * cpr + chn) * 2 * E + which * E + e];
      atomicAdd(sums + (size_t)(blockIdx.x % AM_STATS_REPLICAS) * 2 * C + (size_t)which * C + ce, a);
    }
  }
}

// ---- global average pool NHWC [B][P][ld] -> fp32 [B][C] -------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gap_nhwc_fwd_k(const T* __restrict__ x, int ld, float* __restrict__ out, int P, int C,
                                                      int splits) {
  // grid: B * splits blocks; block reduces rows [s*rows_per, ...) of image b; fp32 atomics into out (pre-zeroed)
  constexpr int E = 16 / (int)sizeof(T);
  extern __shared__ float red[];
  const int b = blockIdx.x / splits, sp = blockIdx.x % splits;
  const int cpr = C / E;
  const int rpp = 256 / cpr > 0 ? 256 / cpr : 1;
  const int tid = threadIdx.x, chunk = tid % cpr, rloc = tid / cpr;
  float s[E];
#pragma unroll
  for (int e = 0; e < E; ++e) s[e] = 0.f;
  const int rows_per = (P + splits - 1) / splits;
  const int r0 = sp * rows_per, r1 = min(P, r0 + rows_per);
  if (rloc < rpp) {
    for (int r = r0 + rloc; r < r1; r += rpp) {
      const uint4 raw = *reinterpret_cast<const uint4*>(x + ((long long)b * P + r) * ld + chunk * E);
      const T* v = reinterpret_cast<const T*>(&raw);
#pragma unroll
      for (int e = 0; e < E; ++e) s[e] += am_to_f32(v[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) red[tid * E + e] = s[e];
  __syncthreads();
  for (int ce = tid; ce < C; ce += 256) {
    const int ch = ce / E, e = ce % E;
    float a = 0.f;
    for (int r = 0; r < rpp; ++r) a += red[(r * cpr + ch) * E + e];
    atomicAdd(out + (long long)b * C + ce, a / (float)P);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gap_nhwc_bwd_k(const float* __restrict__ dout, T* __restrict__ dx, int ld, int P, int C,
                                                      long long total, float mul) {
  // dx[b][p][c] = dout[b][c] * mul / P
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long bp = i / C;
    const long long b = bp / P;
    dx[bp * ld + c] = am_from_f32<T>(dout[b * C + c] * mul / (float)P);
  }
}

// ---- global average pool over NCHW fp32 planes: [BC][HW] -> [BC] -----------------------------
__global__ __launch_bounds__(256) void gap_plane_fwd_k(const float* __restrict__ x, float* __restrict__ out, long long HW) {
  __shared__ float red[4];
  const long long plane = blockIdx.x;
  const float4* p4 = reinterpret_cast<const float4*>(x + plane * HW);
  const long long n4 = HW / 4;
  float s = 0.f;
  for (long long i = threadIdx.x; i < n4; i += 256) {
    const float4 v = p4[i];
    s += (v.x + v.y) + (v.z + v.w);
  }
  for (long long i = n4 * 4 + threadIdx.x; i < HW; i += 256) s += x[plane * HW + i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[plane] = (red[0] + red[1] + red[2] + red[3]) / (float)HW;
}

__global__ __launch_bounds__(256) void gap_plane_bwd_k(const float* __restrict__ dout, float* __restrict__ dx, long long HW,
                                                       long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    dx[i] = dout[i / HW] / (float)HW;
}

// ---- bilinear upsample, align_corners=False (F.interpolate semantics) -----------------------
__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_fwd_k(const T* __restrict__ low, int ld, float* __restrict__ out, int B, int C,
                                                      int h, int w, int H, int W, float sh, float sw) {
  // grid-stride over NCHW output; reads of the tiny low-res map hit L1/L2
  const long long total = (long long)B * C * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int X = (int)(i % W);
    long long t = i / W;
    const int Y = (int)(t % H); t /= H;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(Y, sh, h, y0, y1, ly);
    src_index(X, sw, w, x0, x1, lx);
    const T* base = low + (long long)b * h * w * ld + c;
    const float v00 = am_to_f32(base[((long long)y0 * w + x0) * ld]), v01 = am_to_f32(base[((long long)y0 * w + x1) * ld]);
    const float v10 = am_to_f32(base[((long long)y1 * w + x0) * ld]), v11 = am_to_f32(base[((long long)y1 * w + x1) * ld]);
    const float hy = 1.f - ly, hx = 1.f - lx;
    out[i] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
  }
}

// One workgroup per (b, c, low-res row y): accumulates the high-res rows that touch row y.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_k(const float* __restrict__ dout, T* __restrict__ dlow, int ld, int B, int C,
                                                      int h, int w, int H, int W, float sh, float sw, float mul,
                                                      const float* __restrict__ dev_scale) {
  extern __shared__ float bins[];  // [w]
  const int y = blockIdx.x % h;
  const int c = (blockIdx.x / h) % C;
  const int b = blockIdx.x / (h * C);
  for (int i = threadIdx.x; i < w; i += 256) bins[i] = 0.f;
  __syncthreads();
  // candidate high-res rows: those whose y0 or y1 equals y.  Conservative range, exact test inside.
  int Ylo = (int)floorf(((float)y - 1.f + 0.5f) / sh - 0.5f) - 1;
  int Yhi = (int)ceilf(((float)y + 1.f + 0.5f) / sh - 0.5f) + 1;
  if (Ylo < 0) Ylo = 0;
  if (Yhi > H - 1) Yhi = H - 1;
  if (y == 0) Ylo = 0;
  if (y == h - 1) Yhi = H - 1;
  for (int X = threadIdx.x; X < W; X += 256) {
    int x0, x1;
    float lx;
    src_index(X, sw, w, x0, x1, lx);
    float a0 = 0.f, a1 = 0.f;
    for (int Y = Ylo; Y <= Yhi; ++Y) {
      int y0, y1;
      float ly;
      src_index(Y, sh, h, y0, y1, ly);
      float wy = 0.f;
      if (y0 == y) wy += 1.f - ly;
      if (y1 == y) wy += ly;
      if (wy != 0.f) {
        const float g = dout[(((long long)b * C + c) * H + Y) * W + X] * wy;
        a0 += g * (1.f - lx);
        a1 += g * lx;
      }
    }
    atomicAdd(&bins[x0], a0);
    atomicAdd(&bins[x1], a1);
  }
  __syncthreads();
  const float m = mul * (dev_scale ? dev_scale[0] : 1.f);
  for (int i = threadIdx.x; i < w; i += 256) dlow[(((long long)b * h + y) * w + i) * ld + c] = am_from_f32<T>(bins[i] * m);
}

// ---- global-average-pool of a bilinear upsample, without materialising the upsample ---------------
// mean_{Y,X} up(low)[b,c,Y,X] = sum_{y,x} cy[y] * cx[x] * low[b,y,x,c]  with cy/cx the column sums of the (separable)
// interpolation matrices divided by H and W (host-computed with the kernel's own fp32 index arithmetic).
// One workgroup per image; threads = [pixel lanes][ld channels]; LDS reduce over pixel lanes.
template <typename T>
__global__ __launch_bounds__(256) void upsample_gap_fwd_k(const T* __restrict__ low, int ld, const float* __restrict__ cy,
                                                          const float* __restrict__ cx, float* __restrict__ out, int C, int h, int w) {
  extern __shared__ float red[];  // [256]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int lanes = 256 / ld;          // pixel lanes (ld <= 256)
  const int c = tid % ld, pl = tid / ld;
  float acc = 0.f;
  if (pl < lanes) {
    for (int p = pl; p < h * w; p += lanes) {
      const int y = p / w, x = p - y * w;
      acc += cy[y] * cx[x] * am_to_f32(low[((long long)b * h * w + p) * ld + c]);
    }
  }
  red[tid] = acc;
  __syncthreads();
  if (tid < C) {
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += red[l * ld + tid];
    out[(long long)b * C + tid] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_gap_bwd_k(const float* __restrict__ g, const float* __restrict__ cy,
                                                          const float* __restrict__ cx, T* __restrict__ dlow, int ld, int C, int h,
                                                          int w, long long total, float mul) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % ld);
    const long long bp = i / ld;
    const int p = (int)(bp % (h * w));
    const long long b = bp / (h * w);
    const int y = p / w, x = p - y * w;
    dlow[i] = am_from_f32<T>(c < C ? g[b * C + c] * cy[y] * cx[x] * mul : 0.f);
  }
}

// interpolation column sums, one thread: coef[i] = (1/out) * sum_{dst} weight of source i for dst
__global__ void bilinear_colsum_k(float* __restrict__ coef, int in_size, int out_size) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const float scale = (float)in_size / (float)out_size;
  for (int i = 0; i < in_size; ++i) coef[i] = 0.f;
  // fp64 accumulation of the kernel's fp32 weights, rounded once
  // (in_size <= a few hundred: serial is fine, this runs once per shape and is cached by the caller)
  extern __shared__ double acc[];
  for (int i = 0; i < in_size; ++i) acc[i] = 0.0;
  for (int d = 0; d < out_size; ++d) {
    int i0, i1;
    float l1;
    src_index(d, scale, in_size, i0, i1, l1);
    acc[i0] += (double)(1.f - l1);
    acc[i1] += (double)l1;
  }
  for (int i = 0; i < in_size; ++i) coef[i] = (float)(acc[i] / (double)out_size);
}

// ---- pixel-wise cross entropy over NCHW fp32 logits with ignore_index ------------------------
// fwd: acc[0] += sum of -log softmax[target] over valid pixels (fp64), acc[1] += count
__global__ __launch_bounds__(256) void ce2d_fwd_k(const float* __restrict__ logits, const long long* __restrict__ target, int C,
                                                  long long HW, long long total_pix, long long ignore_index,
                                                  double* __restrict__ acc) {
  __shared__ double red[2][4];
  double ls = 0.0, cnt = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total_pix; i += (long long)gridDim.x * blockDim.x) {
    const long long t = target[i];
    if (t == ignore_index) continue;
    const long long b = i / HW, p = i - b * HW;
    const float* l = logits + b * C * HW + p;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, l[(long long)c * HW]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(l[(long long)c * HW] - mx);
    const float lt = (t >= 0 && t < C) ? l[t * HW] : 0.f;
    ls += (double)(logf(se) + mx - lt);
    cnt += 1.0;
  }
  ls = wave_sum(ls);
  cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ls; red[1][threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(acc + 0, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(acc + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// bwd: dlogits = (softmax - onehot) * gout / count   (0 at ignored pixels)
__global__ __launch_bounds__(256) void ce2d_bwd_k(const float* __restrict__ logits, const long long* __restrict__ target, int C,
                                                  long long HW, long long total_pix, long long ignore_index,
                                                  const double* __restrict__ acc, const float* __restrict__ gout,
                                                  float* __restrict__ dlogits) {
  const float g = gout[0] / (float)fmax(acc[1], 1.0);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total_pix; i += (long long)gridDim.x * blockDim.x) {
    const long long t = target[i];
    const long long b = i / HW, p = i - b * HW;
    const float* l = logits + b * C * HW + p;
    float* d = dlogits + b * C * HW + p;
    if (t == ignore_index) {
      for (int c = 0; c < C; ++c) d[(long long)c * HW] = 0.f;
      continue;
    }
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, l[(long long)c * HW]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(l[(long long)c * HW] - mx);
    const float inv = 1.f / se;
    for (int c = 0; c < C; ++c) {
      float pr = expf(l[(long long)c * HW] - mx) * inv;
      if (c == t) pr -= 1.f;
      d[(long long)c * HW] = pr * g;
    }
  }
}

// ---- bilinear upsample + pixel-wise cross entropy without the full-resolution logits -------------------------------------------
// loss = CrossEntropy(F.interpolate(low, (H, W), bilinear), target, ignore_index) and d loss / d low in ONE pass over the
// labels: the [B, C, H, W] fp32 logits (70 MB per image for the segmentation expert) and their gradient are never written.
// One workgroup per (image b, low-resolution row y, group of CB low-resolution columns): it OWNS the gradient cells
// G[b, y, ka..kb) (plain stores, no atomics, no zero-fill, deterministic) and visits every output pixel whose 2x2 source
// neighbourhood touches them -- a pixel is visited by up to two rows (its y0 and y1) and, at group borders, two groups; its
// softmax is recomputed per visit (a few hundred VALU instructions against 8 B of label and no logits traffic).  The pixel's
// loss is counted by the workgroup that owns its (y0, x0) cell.  Interpolation arithmetic = bilinear_fwd_k's, softmax =
// ce2d_*_k's with v_exp_f32; G is the UNNORMALISED gradient (sum over pixels of (softmax - onehot) * weight): the backward
// entry scales it by grad_out / count.
template <typename T, int C>
__global__ __launch_bounds__(256) void upsample_ce2d_k(const T* __restrict__ low, int ld, const long long* __restrict__ target, int h,
                                                       int w, int H, int W, float sh, float sw, long long ignore_index,
                                                       double* __restrict__ acc, float* __restrict__ G, int xgroups, int CB) {
  extern __shared__ float sm[];
  const int tid = threadIdx.x;
  const int xg = blockIdx.x % xgroups;
  const int y = (blockIdx.x / xgroups) % h;
  const int b = blockIdx.x / (xgroups * h);
  const int ka = xg * CB, kb = min(ka + CB, w);
  const int ncell = CB + 2;                      // cells ka-1 .. kb (clamped to the map)
  float* rows = sm;                              // [3][ncell][C]: low rows y-1, y, y+1
  float* part = rows + 3 * ncell * C;            // [256][2C]
  int* px = reinterpret_cast<int*>(part + 256 * 2 * C);  // [256][2]: the owned cells thread tid adds to (-1: none)
  double* red = reinterpret_cast<double*>(px + 512);     // [2][4]
  for (int i = tid; i < 3 * ncell * C; i += 256) {
    const int c = i % C, k = (i / C) % ncell, r = i / (C * ncell);
    const int yy = min(max(y - 1 + r, 0), h - 1), xx = min(max(ka - 1 + k, 0), w - 1);
    rows[i] = am_to_f32(low[(((long long)b * h + yy) * w + xx) * ld + c]);
  }
  // candidate output rows / columns: conservative ranges, exact tests inside (as bilinear_bwd_k)
  int Ylo = (int)floorf(((float)y - 1.f + 0.5f) / sh - 0.5f) - 1, Yhi = (int)ceilf(((float)y + 1.f + 0.5f) / sh - 0.5f) + 1;
  Ylo = max(Ylo, 0); Yhi = min(Yhi, H - 1);
  if (y == 0) Ylo = 0;
  if (y == h - 1) Yhi = H - 1;
  int Xlo = (int)floorf(((float)ka - 1.f + 0.5f) / sw - 0.5f) - 1, Xhi = (int)ceilf(((float)kb + 0.5f) / sw - 0.5f) + 1;
  Xlo = max(Xlo, 0); Xhi = min(Xhi, W - 1);
  if (ka == 0) Xlo = 0;
  if (kb == w) Xhi = W - 1;
  __syncthreads();

  double ls = 0.0, cnt = 0.0;
  float bin = 0.f;  // thread j < (kb-ka)*C owns G[b, y, ka + j / C, j % C]
  for (int Xb = Xlo; Xb <= Xhi; Xb += 256) {
    const int X = Xb + tid;
    int x0 = 0, x1 = 0;
    float lx = 0.f;
    if (X <= Xhi) src_index(X, sw, w, x0, x1, lx);
    const bool own0 = X <= Xhi && x0 >= ka && x0 < kb, own1 = X <= Xhi && x1 >= ka && x1 < kb;
    float a0[C], a1[C];
#pragma unroll
    for (int c = 0; c < C; ++c) a0[c] = a1[c] = 0.f;
    if (own0 || own1) {
      const float hx = 1.f - lx;
      const float* r0 = rows + (x0 - (ka - 1)) * C;  // own1 => x0 >= ka - 1; own0 => x1 <= kb
      const float* r1 = rows + (x1 - (ka - 1)) * C;
      float rm[C], rc[C], rp[C];  // the three low rows interpolated along x at this column
#pragma unroll
      for (int c = 0; c < C; ++c) {
        rm[c] = hx * r0[c] + lx * r1[c];
        rc[c] = hx * r0[ncell * C + c] + lx * r1[ncell * C + c];
        rp[c] = hx * r0[2 * ncell * C + c] + lx * r1[2 * ncell * C + c];
      }
      const long long* tcol = target + (long long)b * H * W + X;
      long long t_next = tcol[(long long)Ylo * W];  // labels one row ahead: the load is in flight under the previous row's softmax
      for (int Y = Ylo; Y <= Yhi; ++Y) {
        const long long t = t_next;
        t_next = tcol[(long long)min(Y + 1, Yhi) * W];
        int y0, y1;
        float ly;
        src_index(Y, sh, h, y0, y1, ly);
        float wy = 0.f;
        if (y0 == y) wy += 1.f - ly;
        if (y1 == y) wy += ly;
        if (wy == 0.f) continue;
        if (t == ignore_index) continue;
        const float hy = 1.f - ly;
        const bool top_c = y0 == y, bot_c = y1 == y;
        float v[C];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          v[c] = hy * (top_c ? rc[c] : rm[c]) + ly * (bot_c ? rc[c] : rp[c]);
          mx = fmaxf(mx, v[c]);
        }
        float se = 0.f, vt = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          vt = (c == (int)t) ? v[c] : vt;
          v[c] = __expf(v[c] - mx);
          se += v[c];
        }
        if (top_c && own0) {  // this workgroup owns the pixel's (y0, x0) cell: its loss is counted here, once
          ls += (double)(__logf(se) + mx - ((t >= 0 && t < C) ? vt : 0.f));
          cnt += 1.0;
        }
        const float inv = 1.f / se;
        const float w0 = wy * (1.f - lx), w1 = wy * lx;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float gpx = v[c] * inv - ((c == (int)t) ? 1.f : 0.f);
          a0[c] += gpx * w0;
          a1[c] += gpx * w1;
        }
      }
    }
    px[2 * tid] = own0 ? x0 : -1;
    px[2 * tid + 1] = own1 ? x1 : -1;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      part[tid * 2 * C + c] = a0[c];
      part[tid * 2 * C + C + c] = a1[c];
    }
    __syncthreads();
    if (tid < (kb - ka) * C) {  // fixed summation order: deterministic
      const int k = ka + tid / C, c = tid % C;
      for (int i = 0; i < 256; ++i) {
        if (px[2 * i] == k) bin += part[i * 2 * C + c];
        if (px[2 * i + 1] == k) bin += part[i * 2 * C + C + c];
      }
    }
    __syncthreads();
  }
  if (tid < (kb - ka) * C) G[(((long long)b * h + y) * w + ka + tid / C) * C + tid % C] = bin;
  ls = wave_sum(ls);
  cnt = wave_sum(cnt);
  if ((tid & 63) == 0) { red[tid >> 6] = ls; red[4 + (tid >> 6)] = cnt; }
  __syncthreads();
  if (tid == 0) {
    const double l = red[0] + red[1] + red[2] + red[3], n = red[4] + red[5] + red[6] + red[7];
    if (n != 0.0) {
      atomicAdd(acc + 0, l);
      atomicAdd(acc + 1, n);
    }
  }
}

// dlow[b, y, x, c] = G * grad_out / count * mul (pad channels of the pixel stride stay as the caller initialised them)
template <typename T>
__global__ __launch_bounds__(256) void upsample_ce2d_bwd_k(const float* __restrict__ G, const double* __restrict__ acc,
                                                           const float* __restrict__ gout, float mul, T* __restrict__ dlow, int ld, int C,
                                                           long long total) {
  const float g = gout[0] / (float)fmax(acc[1], 1.0) * mul;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    dlow[(i / C) * ld + i % C] = am_from_f32<T>(G[i] * g);
}

}  // namespace

#define DT_OK(d) ((d) == AM_F16 || (d) == AM_F32)
#define ST(s) static_cast<hipStream_t>(s)

extern "C" int am_nchw_to_nhwc(int dtype, const float* src, void* dst, int B, int C, int H, int W, int ld, float mul,
                               am_stream_t stream) {
  if (!DT_OK(dtype) || !src || !dst || C > ld || B < 0 || (ld * (dtype == AM_F16 ? 2 : 4)) % 16 != 0) return AM_ERR_ARG;
  const long long HW = (long long)H * W, total = HW * B;
  if (total == 0) return AM_OK;
  if (dtype == AM_F16) hipLaunchKernelGGL(nchw_to_nhwc_k<half_t>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), src, (half_t*)dst, C, HW, ld, total, mul);
  else hipLaunchKernelGGL(nchw_to_nhwc_k<float>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), src, (float*)dst, C, HW, ld, total, mul);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_image_u8_s2d(int dtype, const uint8_t* src, void* dst, int B, int C, int H, int W, const float* mean,
                               const float* stdv, am_stream_t stream) {
  if (!DT_OK(dtype) || !src || !dst || C < 1 || C > 4 || H < 1 || W < 1 || B < 0 || ((mean == nullptr) != (stdv == nullptr)))
    return AM_ERR_ARG;
  const long long total = (long long)B * ((H + 1) / 2) * ((W + 1) / 2);
  if (total == 0) return AM_OK;
  float m[4] = {0.f, 0.f, 0.f, 0.f}, sd[4] = {1.f, 1.f, 1.f, 1.f};
  const int normalize = mean != nullptr;
  for (int c = 0; c < C && normalize; ++c) { m[c] = mean[c]; sd[c] = stdv[c]; }
  if (dtype == AM_F16)
    hipLaunchKernelGGL(image_u8_s2d_k<half_t>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), src, (half_t*)dst, C, H, W, total, m[0], m[1], m[2], m[3], sd[0], sd[1], sd[2], sd[3], normalize);
  else
    hipLaunchKernelGGL(image_u8_s2d_k<float>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), src, (float*)dst, C, H, W, total, m[0], m[1], m[2], m[3], sd[0], sd[1], sd[2], sd[3], normalize);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_nhwc_to_nchw(int dtype, const void* src, float* dst, int B, int C, int H, int W, int ld, float mul,
                               am_stream_t stream) {
  if (!DT_OK(dtype) || !src || !dst || C > ld || B < 0) return AM_ERR_ARG;
  const long long HW = (long long)H * W, total = HW * B * C;
  if (total == 0) return AM_OK;
  if (dtype == AM_F16) hipLaunchKernelGGL(nhwc_to_nchw_k<half_t>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), (const half_t*)src, dst, C, HW, ld, total, mul);
  else hipLaunchKernelGGL(nhwc_to_nchw_k<float>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), (const float*)src, dst, C, HW, ld, total, mul);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_maxpool3x3s2_fwd(int dtype, const void* x, void* y, uint8_t* argmax, int B, int IH, int IW, int C,
                                   am_stream_t stream) {
  const int es = dtype == AM_F16 ? 2 : 4;
  if (!DT_OK(dtype) || !x || !y || (C * es) % 16 != 0) return AM_ERR_ARG;
  const int OH = (IH - 1) / 2 + 1, OW = (IW - 1) / 2 + 1;
  const long long total = (long long)B * OH * OW * (C * es / 16);
  if (total == 0) return AM_OK;
  if (total >= (1ll << 31)) return AM_ERR_UNSUPPORTED;  // 32-bit index arithmetic in the kernel
  if (dtype == AM_F16) hipLaunchKernelGGL(maxpool_fwd_k<half_t>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), (const half_t*)x, (half_t*)y, argmax, B, IH, IW, OH, OW, C, nullptr, nullptr);
  else hipLaunchKernelGGL(maxpool_fwd_k<float>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), (const float*)x, (float*)y, argmax, B, IH, IW, OH, OW, C, nullptr, nullptr);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_bn_relu_maxpool3x3s2_fwd(int dtype, const void* x, const float* scale, const float* shift, void* y, uint8_t* argmax,
                                           int B, int IH, int IW, int C, am_stream_t stream) {
  const int es = dtype == AM_F16 ? 2 : 4;
  if (!DT_OK(dtype) || !x || !y || !scale || !shift || (C * es) % 16 != 0) return AM_ERR_ARG;
  const int OH = (IH - 1) / 2 + 1, OW = (IW - 1) / 2 + 1;
  const long long total = (long long)B * OH * OW * (C * es / 16);
  if (total == 0) return AM_OK;
  if (total >= (1ll << 31)) return AM_ERR_UNSUPPORTED;
  if (dtype == AM_F16) hipLaunchKernelGGL(maxpool_fwd_k<half_t>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), (const half_t*)x, (half_t*)y, argmax, B, IH, IW, OH, OW, C, scale, shift);
  else hipLaunchKernelGGL(maxpool_fwd_k<float>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), (const float*)x, (float*)y, argmax, B, IH, IW, OH, OW, C, scale, shift);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_maxpool3x3s2_bwd(int dtype, const void* dy, const uint8_t* argmax, void* dx, int B, int IH, int IW, int C,
                                   am_stream_t stream) {
  const int es = dtype == AM_F16 ? 2 : 4;
  if (!DT_OK(dtype) || !dy || !dx || !argmax || (C * es) % 16 != 0) return AM_ERR_ARG;
  const int OH = (IH - 1) / 2 + 1, OW = (IW - 1) / 2 + 1;
  const long long total = (long long)B * ((IH + 1) / 2) * ((IW + 1) / 2) * (C * es / 16);
  if (total == 0) return AM_OK;
  if (total >= (1ll << 31)) return AM_ERR_UNSUPPORTED;  // 32-bit index arithmetic in the kernel
  if (dtype == AM_F16) hipLaunchKernelGGL((maxpool_bwd_block_k<half_t, false>), dim3(ew_grid(total)), dim3(256), 0, ST(stream), (const half_t*)dy, argmax, (half_t*)dx, B, IH, IW, OH, OW, C, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  else hipLaunchKernelGGL((maxpool_bwd_block_k<float, false>), dim3(ew_grid(total)), dim3(256), 0, ST(stream), (const float*)dy, argmax, (float*)dx, B, IH, IW, OH, OW, C, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_maxpool3x3s2_bwd_bn(int dtype, const void* dy, const uint8_t* argmax, void* dx, int B, int IH, int IW, int C,
                                      const void* raw, const float* mean, const float* rstd, const float* scale, const float* shift,
                                      double* sums, am_stream_t stream) {
  const int es = dtype == AM_F16 ? 2 : 4;
  if (!DT_OK(dtype) || !dy || !dx || !argmax || !raw || !mean || !rstd || !scale || !shift || !sums || (C * es) % 16 != 0) return AM_ERR_ARG;
  const int cpr = C * es / 16;
  if (cpr > 256 || 256 % cpr != 0) return AM_ERR_UNSUPPORTED;  // caller: am_maxpool3x3s2_bwd + am_bn_bwd_reduce_sign
  const int OH = (IH - 1) / 2 + 1, OW = (IW - 1) / 2 + 1;
  const long long total = (long long)B * ((IH + 1) / 2) * ((IW + 1) / 2) * cpr;
  if (total == 0) return AM_OK;
  if (total >= (1ll << 31)) return AM_ERR_UNSUPPORTED;
  const size_t lds = 256 * 2 * (16 / es) * (dtype == AM_F16 ? sizeof(float) : sizeof(double));
  if (dtype == AM_F16) hipLaunchKernelGGL((maxpool_bwd_block_k<half_t, true>), dim3(ew_grid(total)), dim3(256), lds, ST(stream), (const half_t*)dy, argmax, (half_t*)dx, B, IH, IW, OH, OW, C, (const half_t*)raw, mean, rstd, scale, shift, sums);
  else hipLaunchKernelGGL((maxpool_bwd_block_k<float, true>), dim3(ew_grid(total)), dim3(256), lds, ST(stream), (const float*)dy, argmax, (float*)dx, B, IH, IW, OH, OW, C, (const float*)raw, mean, rstd, scale, shift, sums);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_gap_nhwc_fwd(int dtype, const void* x, int ld, float* out, int B, int P, int C, am_stream_t stream) {
  const int es = dtype == AM_F16 ? 2 : 4;
  if (!DT_OK(dtype) || !x || !out || (C * es) % 16 != 0 || (ld * es) % 16 != 0 || C * es / 16 > 256) return AM_ERR_ARG;
  if (B == 0 || P == 0) return AM_OK;
  hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * C, ST(stream));
  if (e != hipSuccess) return AM_ERR_LAUNCH;
  int splits = 1024 / B;
  if (splits < 1) splits = 1;
  if (splits > (P + 63) / 64) splits = (P + 63) / 64;
  const size_t lds = 256 * (16 / es) * sizeof(float);
  if (dtype == AM_F16) hipLaunchKernelGGL(gap_nhwc_fwd_k<half_t>, dim3(B * splits), dim3(256), lds, ST(stream), (const half_t*)x, ld, out, P, C, splits);
  else hipLaunchKernelGGL(gap_nhwc_fwd_k<float>, dim3(B * splits), dim3(256), lds, ST(stream), (const float*)x, ld, out, P, C, splits);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_gap_nhwc_bwd(int dtype, const float* dout, void* dx, int ld, int B, int P, int C, float mul,
                               am_stream_t stream) {
  if (!DT_OK(dtype) || !dout || !dx || C > ld) return AM_ERR_ARG;
  const long long total = (long long)B * P * C;
  if (total == 0) return AM_OK;
  if (dtype == AM_F16) hipLaunchKernelGGL(gap_nhwc_bwd_k<half_t>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), dout, (half_t*)dx, ld, P, C, total, mul);
  else hipLaunchKernelGGL(gap_nhwc_bwd_k<float>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), dout, (float*)dx, ld, P, C, total, mul);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_gap_plane_fwd(const float* x, float* out, long long planes, long long HW, am_stream_t stream) {
  if (!x || !out || planes < 0 || HW <= 0 || planes > 0x7fffffffLL) return AM_ERR_ARG;
  if (planes == 0) return AM_OK;
  hipLaunchKernelGGL(gap_plane_fwd_k, dim3((unsigned)planes), dim3(256), 0, ST(stream), x, out, HW);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_gap_plane_bwd(const float* dout, float* dx, long long planes, long long HW, am_stream_t stream) {
  if (!dout || !dx || planes < 0 || HW <= 0) return AM_ERR_ARG;
  const long long total = planes * HW;
  if (total == 0) return AM_OK;
  hipLaunchKernelGGL(gap_plane_bwd_k, dim3(ew_grid(total)), dim3(256), 0, ST(stream), dout, dx, HW, total);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_bilinear_up_fwd(int dtype, const void* low, int ld, float* out, int B, int C, int h, int w, int H, int W,
                                  am_stream_t stream) {
  if (!DT_OK(dtype) || !low || !out || C > ld || h <= 0 || w <= 0) return AM_ERR_ARG;
  const long long total = (long long)B * C * H * W;
  if (total == 0) return AM_OK;
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  if (dtype == AM_F16) hipLaunchKernelGGL(bilinear_fwd_k<half_t>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), (const half_t*)low, ld, out, B, C, h, w, H, W, sh, sw);
  else hipLaunchKernelGGL(bilinear_fwd_k<float>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), (const float*)low, ld, out, B, C, h, w, H, W, sh, sw);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_bilinear_up_bwd(int dtype, const float* dout, void* dlow, int ld, int B, int C, int h, int w, int H, int W,
                                  float mul, const float* dev_scale, am_stream_t stream) {
  if (!DT_OK(dtype) || !dout || !dlow || C > ld || h <= 0 || w <= 0) return AM_ERR_ARG;
  if ((long long)B * C * h == 0) return AM_OK;
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const size_t lds = sizeof(float) * (size_t)w;
  if (dtype == AM_F16) hipLaunchKernelGGL(bilinear_bwd_k<half_t>, dim3(B * C * h), dim3(256), lds, ST(stream), dout, (half_t*)dlow, ld, B, C, h, w, H, W, sh, sw, mul, dev_scale);
  else hipLaunchKernelGGL(bilinear_bwd_k<float>, dim3(B * C * h), dim3(256), lds, ST(stream), dout, (float*)dlow, ld, B, C, h, w, H, W, sh, sw, mul, dev_scale);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_ce2d_fwd(const float* logits, const long long* target, int B, int C, long long HW, long long ignore_index,
                           double* acc2, am_stream_t stream) {
  if (!logits || !target || !acc2 || C <= 0) return AM_ERR_ARG;
  hipError_t e = hipMemsetAsync(acc2, 0, 2 * sizeof(double), ST(stream));
  if (e != hipSuccess) return AM_ERR_LAUNCH;
  const long long total = HW * B;
  if (total == 0) return AM_OK;
  hipLaunchKernelGGL(ce2d_fwd_k, dim3(ew_grid(total)), dim3(256), 0, ST(stream), logits, target, C, HW, total, ignore_index, acc2);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_ce2d_bwd(const float* logits, const long long* target, int B, int C, long long HW, long long ignore_index,
                           const double* acc2, const float* grad_out, float* dlogits, am_stream_t stream) {
  if (!logits || !target || !acc2 || !grad_out || !dlogits || C <= 0) return AM_ERR_ARG;
  const long long total = HW * B;
  if (total == 0) return AM_OK;
  hipLaunchKernelGGL(ce2d_bwd_k, dim3(ew_grid(total)), dim3(256), 0, ST(stream), logits, target, C, HW, total, ignore_index, acc2, grad_out, dlogits);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

template <typename T, int C>
static int launch_upsample_ce2d(const void* low, int ld, const long long* target, int B, int h, int w, int H, int W, long long ignore_index,
                                double* acc2, float* G, hipStream_t s) {
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  // columns per workgroup: the widest group whose output footprint ((CB + 1) / sw + 2 columns) fits one 256-thread sweep
  int CB = (int)floorf(254.f * sw) - 1;
  CB = CB < 1 ? 1 : (CB > w ? w : CB);
  if (CB * C > 256) CB = 256 / C;
  const int xgroups = (w + CB - 1) / CB;
  const size_t lds = sizeof(float) * ((size_t)3 * (CB + 2) * C + 256 * 2 * C) + sizeof(int) * 512 + sizeof(double) * 8;
  hipLaunchKernelGGL((upsample_ce2d_k<T, C>), dim3((unsigned)(B * h * xgroups)), dim3(256), lds, s, (const T*)low, ld, target, h, w, H, W, sh, sw,
                     ignore_index, acc2, G, xgroups, CB);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_upsample_ce2d_fwd(int dtype, const void* low, int ld, const long long* target, int B, int C, int h, int w, int H, int W,
                                    long long ignore_index, double* acc2, float* G, am_stream_t stream) {
  if (!DT_OK(dtype) || !low || !target || !acc2 || !G || C <= 0 || C > ld || h <= 0 || w <= 0 || H <= 0 || W <= 0) return AM_ERR_ARG;
  if (C != 3 && C != 19) return AM_ERR_UNSUPPORTED;  // the class counts of the reference's dense experts (register-resident per-class state)
  if ((long long)B * h * w >= (1ll << 31) / 64) return AM_ERR_UNSUPPORTED;
  if (hipMemsetAsync(acc2, 0, 2 * sizeof(double), ST(stream)) != hipSuccess) return AM_ERR_LAUNCH;
  if (B == 0) return AM_OK;
  hipStream_t s = ST(stream);
  if (dtype == AM_F16) return C == 3 ? launch_upsample_ce2d<half_t, 3>(low, ld, target, B, h, w, H, W, ignore_index, acc2, G, s)
                                     : launch_upsample_ce2d<half_t, 19>(low, ld, target, B, h, w, H, W, ignore_index, acc2, G, s);
  return C == 3 ? launch_upsample_ce2d<float, 3>(low, ld, target, B, h, w, H, W, ignore_index, acc2, G, s)
                : launch_upsample_ce2d<float, 19>(low, ld, target, B, h, w, H, W, ignore_index, acc2, G, s);
}

extern "C" int am_upsample_ce2d_bwd(int dtype, const float* G, const double* acc2, const float* grad_out, float mul, void* dlow, int ld,
                                    int B, int C, int h, int w, am_stream_t stream) {
  if (!DT_OK(dtype) || !G || !acc2 || !grad_out || !dlow || C <= 0 || C > ld) return AM_ERR_ARG;
  const long long total = (long long)B * h * w * C;
  if (total == 0) return AM_OK;
  if (dtype == AM_F16) hipLaunchKernelGGL(upsample_ce2d_bwd_k<half_t>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), G, acc2, grad_out, mul, (half_t*)dlow, ld, C, total);
  else hipLaunchKernelGGL(upsample_ce2d_bwd_k<float>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), G, acc2, grad_out, mul, (float*)dlow, ld, C, total);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_bilinear_colsum(float* coef, int in_size, int out_size, am_stream_t stream) {
  if (!coef || in_size <= 0 || out_size <= 0 || in_size > 4096) return AM_ERR_ARG;
  hipLaunchKernelGGL(bilinear_colsum_k, dim3(1), dim3(64), sizeof(double) * (size_t)in_size, ST(stream), coef, in_size, out_size);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_upsample_gap_fwd(int dtype, const void* low, int ld, const float* cy, const float* cx, float* out, int B, int C,
                                   int h, int w, am_stream_t stream) {
  if (!DT_OK(dtype) || !low || !cy || !cx || !out || C > ld || ld > 256 || h <= 0 || w <= 0) return AM_ERR_ARG;
  if (B == 0) return AM_OK;
  if (dtype == AM_F16) hipLaunchKernelGGL(upsample_gap_fwd_k<half_t>, dim3(B), dim3(256), 256 * sizeof(float), ST(stream), (const half_t*)low, ld, cy, cx, out, C, h, w);
  else hipLaunchKernelGGL(upsample_gap_fwd_k<float>, dim3(B), dim3(256), 256 * sizeof(float), ST(stream), (const float*)low, ld, cy, cx, out, C, h, w);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_upsample_gap_bwd(int dtype, const float* g, const float* cy, const float* cx, void* dlow, int ld, int B, int C,
                                   int h, int w, float mul, am_stream_t stream) {
  if (!DT_OK(dtype) || !g || !cy || !cx || !dlow || C > ld || h <= 0 || w <= 0) return AM_ERR_ARG;
  const long long total = (long long)B * h * w * ld;
  if (total == 0) return AM_OK;
  if (dtype == AM_F16) hipLaunchKernelGGL(upsample_gap_bwd_k<half_t>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), g, cy, cx, (half_t*)dlow, ld, C, h, w, total, mul);
  else hipLaunchKernelGGL(upsample_gap_bwd_k<float>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), g, cy, cx, (float*)dlow, ld, C, h, w, total, mul);
  AM_CHECK_LAUNCH();
  return AM_OK;
}

extern "C" int am_image_s2d(int dtype, const float* src, void* dst, int B, int C, int H, int W, am_stream_t stream) {
  if (!DT_OK(dtype) || !src || !dst || C < 1 || C > 4 || (H & 1) || (W & 1) || B < 0) return AM_ERR_ARG;
  const long long total = (long long)B * (H / 2) * (W / 2);
  if (total == 0) return AM_OK;
  if (dtype == AM_F16 && C == 3 && W % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0)
    hipLaunchKernelGGL(image_s2d_x2_k<half_t>, dim3(ew_grid(total / 2)), dim3(256), 0, ST(stream), src, (half_t*)dst, H, W, total / 2);
  else if (dtype == AM_F16) hipLaunchKernelGGL(image_s2d_k<half_t>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), src, (half_t*)dst, C, H, W, total);
  else hipLaunchKernelGGL(image_s2d_k<float>, dim3(ew_grid(total)), dim3(256), 0, ST(stream), src, (float*)dst, C, H, W, total);
  AM_CHECK_LAUNCH();
  return AM_OK;
}
