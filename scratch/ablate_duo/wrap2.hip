// A/B build of conv3x3_c64n64_duo_k: -DDUO_SRC="<path>" picks the source (the tree's or a saved copy); plain and PRE (input transform) entries.
#include DUO_SRC
thread_local int g_am_conv_variant = 0;
int am_tuning(int) { return 1; }
extern "C" int duo_run(const am_conv_geom* g, const void* x, const void* w, void* y, double* stats, void* stream) {
  return am_conv3x3_c64n64_duo_f16(g, x, w, nullptr, 0, nullptr, y, stats, static_cast<hipStream_t>(stream));
}
extern "C" int duo_run_pre(const am_conv_geom* g, const void* x, const float* sc, const float* sh, const void* w, void* y, double* stats, void* stream) {
  return am_conv3x3_c64n64_duo_pre_f16(g, x, sc, sh, w, nullptr, 0, nullptr, y, stats, static_cast<hipStream_t>(stream));
}
#ifdef AMP3_DIAG
extern "C" int duo_diag(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(amp3::g_duo_diag), 8 * sizeof(long long)) == hipSuccess ? 0 : 1; }
#endif
