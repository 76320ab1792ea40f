// Ablation build of conv_band16_k (scratch/ablate_band16/run.sh): the product kernel compiled with -DAMB_ABL=<mask>, behind a
// one-function C entry.  The two symbols the kernel file takes from the library are stubbed here.
#include "../../self-driving-model_amd/csrc/conv_band16.hip"
thread_local int g_am_conv_variant = 0;
int am_tuning(int) { return 1; }
extern "C" int am_conv_npad(int N) { return N > 64 ? (N + 127) / 128 * 128 : (N > 32 ? 64 : 32); }
extern "C" int band16_run(const am_conv_geom* g, const void* x, const void* w, void* y, double* stats, void* stream) {
  return am_conv_band16_f16(g, x, w, nullptr, 0, nullptr, y, stats, static_cast<hipStream_t>(stream));
}
