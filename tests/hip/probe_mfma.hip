// Hardware probe for gfx950 primitives the conv kernels are built on.
// Test infrastructure only: checks, with exact small-integer data, the lane maps of
//   v_mfma_f32_32x32x16_f16, v_mfma_f32_32x32x2_f32, v_mfma_f32_16x16x32_f16,
//   ds_read_b64_tr_b16 and global_load_lds_dwordx4
// against the maps written in self-driving-model_amd/csrc/*.hip. Prints PASS/FAIL per primitive.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2);} } while (0)

// A[32][16], B[16][32] row-major f16 -> C[32][32]
__global__ void k_mfma_32x32x16(const _Float16* A, const _Float16* B, float* C) {
  int l = threadIdx.x;
  half8_t a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[(l & 31) * 16 + 8 * (l >> 5) + j];
    b[j] = B[(8 * (l >> 5) + j) * 32 + (l & 31)];
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    C[row * 32 + (l & 31)] = c[r];
  }
}

// A[32][2], B[2][32] f32
__global__ void k_mfma_32x32x2(const float* A, const float* B, float* C) {
  int l = threadIdx.x;
  float a = A[(l & 31) * 2 + (l >> 5)];
  float b = B[(l >> 5) * 32 + (l & 31)];
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    C[row * 32 + (l & 31)] = c[r];
  }
}

// A[16][32], B[32][16] f16 -> C[16][16]
__global__ void k_mfma_16x16x32(const _Float16* A, const _Float16* B, float* C) {
  int l = threadIdx.x;
  half8_t a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j];
    b[j] = B[(8 * (l >> 4) + j) * 16 + (l & 15)];
  }
  f32x4 c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) * 4 + r;
    C[row * 16 + (l & 15)] = c[r];
  }
}

// Transposed LDS read: LDS image [16 rows(k)][32 cols] of u16, value = row*64+col.
// Each 16-lane group g reads the 4-row x 16-col block with first row r0(g), first col c0(g):
//   lane 4q+p of the group supplies &img[r0+q][c0+4p]; expected: lane i of the group gets
//   img[r0+0..3][c0+i] in elements 0..3.
__global__ void k_tr(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short img[16 * 32];
  int l = threadIdx.x;
  for (int i = l; i < 16 * 32; i += 64) img[i] = (unsigned short)((i / 32) * 64 + (i % 32));
  __syncthreads();
  int g = l >> 4, i16 = l & 15, q = i16 >> 2, p = i16 & 3;
  int r0 = (g >> 1) * 4;       // groups 0,1 -> rows 0..3; groups 2,3 -> rows 4..7
  int c0 = (g & 1) * 16;       // even groups cols 0..15, odd groups cols 16..31
  const unsigned short* addr = &img[(r0 + q) * 32 + c0 + 4 * p];
  s4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)addr);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)v[e];
}

// LDS-DMA: each lane supplies its own global source (16 B); destination = uniform base + lane*16.
__global__ void k_glds(const unsigned* src, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
  int l = threadIdx.x;
  // lane l fetches source chunk (63-l): a per-lane gather
  const unsigned* g = src + (63 - l) * 4;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = lds[l * 4 + e];
}

static int check(const char* name, const std::vector<float>& got, const std::vector<float>& ref) {
  int bad = 0;
  for (size_t i = 0; i < ref.size(); ++i) if (got[i] != ref[i]) { if (bad < 5) printf("  %s mismatch @%zu got %g ref %g\n", name, i, got[i], ref[i]); ++bad; }
  printf("%s: %s (%d bad of %zu)\n", name, bad ? "FAIL" : "PASS", bad, ref.size());
  return bad != 0;
}

int main() {
  int fails = 0;
  {  // 32x32x16 f16
    std::vector<_Float16> A(32 * 16), B(16 * 32);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (_Float16)((i * 3 + k * 5) % 7 - 3);
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (_Float16)((k * 2 + j * 7 + (j > k)) % 5 - 2);
    std::vector<float> ref(32 * 32, 0.f), got(32 * 32);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += (float)A[i * 16 + k] * (float)B[k * 32 + j]; ref[i * 32 + j] = s; }
    _Float16 *dA, *dB; float* dC;
    CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, got.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    k_mfma_32x32x16<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), dC, got.size() * 4, hipMemcpyDeviceToHost));
    fails += check("mfma_f32_32x32x16_f16", got, ref);
  }
  {  // 32x32x2 f32
    std::vector<float> A(32 * 2), B(2 * 32);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 2; ++k) A[i * 2 + k] = (float)((i * 3 + k * 5) % 7 - 3);
    for (int k = 0; k < 2; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)((k * 2 + j * 7 + (j > 3 * k)) % 5 - 2);
    std::vector<float> ref(32 * 32, 0.f), got(32 * 32);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 2; ++k) s += A[i * 2 + k] * B[k * 32 + j]; ref[i * 32 + j] = s; }
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, got.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    k_mfma_32x32x2<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), dC, got.size() * 4, hipMemcpyDeviceToHost));
    fails += check("mfma_f32_32x32x2f32", got, ref);
  }
  {  // 16x16x32 f16
    std::vector<_Float16> A(16 * 32), B(32 * 16);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (_Float16)((i * 3 + k * 5) % 7 - 3);
    for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (_Float16)((k * 2 + j * 7 + (j > k)) % 5 - 2);
    std::vector<float> ref(16 * 16, 0.f), got(16 * 16);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += (float)A[i * 32 + k] * (float)B[k * 16 + j]; ref[i * 16 + j] = s; }
    _Float16 *dA, *dB; float* dC;
    CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, got.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    k_mfma_16x16x32<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), dC, got.size() * 4, hipMemcpyDeviceToHost));
    fails += check("mfma_f32_16x16x32_f16", got, ref);
  }
  {  // tr read
    std::vector<unsigned short> got(64 * 4);
    unsigned short* d; CK(hipMalloc(&d, got.size() * 2));
    k_tr<<<1, 64>>>(d); CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), d, got.size() * 2, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
      int g = l >> 4, i = l & 15; int r0 = (g >> 1) * 4, c0 = (g & 1) * 16;
      for (int e = 0; e < 4; ++e) {
        unsigned short ref = (unsigned short)((r0 + e) * 64 + c0 + i);
        if (got[l * 4 + e] != ref) { if (bad < 8) printf("  tr lane %d elem %d got (r%d,c%d) want (r%d,c%d)\n", l, e, got[l*4+e] / 64, got[l*4+e] % 64, r0 + e, c0 + i); ++bad; }
      }
    }
    printf("ds_read_b64_tr_b16: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad);
    if (bad) { for (int l = 0; l < 64; ++l) { printf("  lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", got[l*4+e] / 64, got[l*4+e] % 64); printf("\n"); } }
    fails += bad != 0;
  }
  {  // glds
    std::vector<unsigned> src(64 * 4), got(64 * 4);
    for (int i = 0; i < 256; ++i) src[i] = 1000 + i;
    unsigned *ds, *dd; CK(hipMalloc(&ds, 1024)); CK(hipMalloc(&dd, 1024));
    CK(hipMemcpy(ds, src.data(), 1024, hipMemcpyHostToDevice));
    k_glds<<<1, 64>>>(ds, dd); CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), dd, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) if (got[l * 4 + e] != src[(63 - l) * 4 + e]) ++bad;
    printf("global_load_lds_dwordx4: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad);
    fails += bad != 0;
  }
  printf("probe: %d failing primitive(s)\n", fails);
  return fails ? 1 : 0;
}
