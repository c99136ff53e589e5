// Kernel micro-benchmark against libsimx_hip.so on the step's real shapes (not part of the product).
// Build: hipcc -O2 tools/kbench.cpp -Iinclude -Lsimxns_amd -lsimx_hip -Wl,-rpath,$PWD/simxns_amd -o tools/kbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <dlfcn.h>
#include "simx.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define SX(x) do { int r = (x); if (r) { printf("simx error %d: %s (line %d)\n", r, simx_last_error(), __LINE__); exit(1);} } while (0)

static int g_dt = SIMX_BF16;                        // KB_F16=1: IEEE-half operands (SIMX_F16)
static void* dalloc(size_t bytes, int fill) {
  if (getenv("KB_ZERO")) fill = 0;                 // power experiment: all-zero operands (DVFS give-back, MI355X_MICROARCH.md)
  void* p; CK(hipMalloc(&p, bytes));
  std::vector<unsigned short> h(1 << 20);
  for (size_t i = 0; i < h.size(); ++i) {
    float f = ((int)((i * 2654435761u) >> 20 & 1023) - 512) / 1024.0f * (fill ? 1.f : 0.f);
    if (g_dt == SIMX_F16) { _Float16 hf = (_Float16)f; memcpy(&h[i], &hf, 2); }
    else { unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
  }
  for (size_t off = 0; off < bytes; off += h.size() * 2) CK(hipMemcpy((char*)p + off, h.data(), std::min(h.size() * 2, bytes - off), hipMemcpyHostToDevice));
  return p;
}
template <typename F> static double timeit(F f, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / iters;
}
// KB_TS=1 on a tools/p3_timeline.py build: cycles per tile of the persistent NT kernel's phases (waves 0 and 5 of one workgroup,
// tiles 2..13 of the last launch): main loop, epilogue, epilogue end -> first stage boundary of the next tile, tile period
static void print_timeline() {
  if (!getenv("KB_TS")) return;
  typedef int (*fn_t)(unsigned long long*);
  fn_t fn = (fn_t)dlsym(RTLD_DEFAULT, "simx_debug_p3_ts");
  if (!fn) { printf("  (no simx_debug_p3_ts in this build)\n"); return; }
  unsigned long long ts[2 * 16 * 4];
  CK(hipDeviceSynchronize());
  if (fn(ts)) return;
  for (int w = 0; w < 2; ++w) {
    double main_ = 0, epi = 0, gap = 0, per = 0; int n = 0;
    for (int t = 2; t < 13; ++t) {
      const unsigned long long* a = ts + (w * 16 + t) * 4; const unsigned long long* b = a + 4;
      if (!a[0] || !b[0] || b[0] < a[0]) continue;
      main_ += (double)(a[1] - a[0]); epi += (double)(a[2] - a[1]); gap += (double)(a[3] - a[2]); per += (double)(b[0] - a[0]); ++n;
    }
    if (n) printf("  timeline wave %d: main loop %.0f  epilogue %.0f  epilogue-end -> next tile's first boundary %.0f  tile period %.0f cycles (%d tiles)\n",
                  w ? 5 : 0, main_ / n, epi / n, gap / n, per / n, n);
  }
}
int main(int argc, char** argv) {
  if (getenv("KB_F16")) g_dt = SIMX_F16;
  const int T = argc > 1 ? atoi(argv[1]) : 262144;
  // KB_H=1024: the recipe teacher's geometry (ernie-2.0-large: H = 1024, F = 4096) -- main NT / TN tables only
  const int H = getenv("KB_H") ? atoi(getenv("KB_H")) : 768, F = 4 * H, iters = 10;
  const bool base = H == 768;
  if (getenv("KB_XP")) {      // the fp32 engine's dense GEMMs on pre-split operand planes (csrc/gemm_xp.hip)
    auto palloc = [&](size_t n, int fmt) { g_dt = fmt; return dalloc(n * 2 * 2, 1); };     // hi plane, lo plane
    auto falloc = [&](size_t n) { float* p; CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4)); return p; };
    void* Ah = palloc((size_t)T * F, SIMX_F16); void* Ab = palloc((size_t)T * F, SIMX_BF16); void* Xb = palloc((size_t)T * F, SIMX_BF16);
    void* Wh = palloc((size_t)F * H, SIMX_F16); void* Wb = palloc((size_t)F * H, SIMX_BF16);
    float* C = falloc((size_t)T * F); float* IN = falloc((size_t)T * F); void* Cp = palloc((size_t)T * F, SIMX_F16);
    float* bias = falloc(F); float* G = falloc((size_t)F * H);
    size_t wsb = 0; { size_t v; v = simx_gemm_tn_planes_workspace_bytes(3*H, H, T); wsb = v; v = simx_gemm_tn_planes_workspace_bytes(F, H, T); if (v > wsb) wsb = v; v = simx_gemm_tn_planes_workspace_bytes(H, F, T); if (v > wsb) wsb = v; }
    void* ws; CK(hipMalloc(&ws, wsb + 256));
    struct S { const char* name; int N, K, epi, res, fmt; } nt[] = {
      {"qkv   fwd  N=2304 K=768 bias", 3 * H, H, 0, 0, SIMX_F16}, {"oproj fwd  N=768  K=768 bias+res", H, H, 0, 1, SIMX_F16},
      {"ffn1  fwd  N=3072 K=768 gelu", F, H, 1, 0, SIMX_F16},     {"ffn2  fwd  N=768  K=3072 bias+res", H, F, 0, 1, SIMX_F16},
      {"ffn2 dgrad N=3072 K=768 dgelu", F, H, 2, 1, SIMX_BF16},   {"ffn1 dgrad N=768  K=3072 +res", H, F, 0, 1, SIMX_BF16},
      {"qkv  dgrad N=768  K=2304 +res", H, 3 * H, 0, 1, SIMX_BF16}, {"ffn1  fwd  N=3072 K=768 gelu(infer)", F, H, 3, 0, SIMX_F16}};
    double tot = 0, totf = 0; int idx = 0;
    for (auto& s : nt) {
      const void* A = s.fmt == SIMX_F16 ? Ah : Ab; const void* W = s.fmt == SIMX_F16 ? Wh : Wb;
      double ms = timeit([&] { SX(simx_gemm_nt_planes(0, s.fmt, s.epi, T, s.N, s.K, A, s.K, (long)T * s.K, W, s.K, (long)s.N * s.K, C, s.N, s.epi == 2 ? nullptr : bias,
                                                     s.res ? IN : nullptr, s.N, Cp, s.N, (long)T * s.N, nullptr)); }, 5);
      double fl = 2.0 * T * s.N * s.K; if (idx++ < 7) { tot += ms; totf += fl; }
      printf("xp gemm_nt %-36s %8.3f ms  %7.1f TF/s\n", s.name, ms, fl / ms / 1e9);
    }
    printf("xp gemm_nt total(7) %.2f ms  avg %.1f TF/s\n", tot, totf / tot / 1e9);
    struct S2 { const char* name; int M, N; } tn[] = {{"wqkv [2304,768]", 3 * H, H}, {"wo [768,768]", H, H}, {"w1 [3072,768]", F, H}, {"w2 [768,3072]", H, F}};
    tot = 0; totf = 0;
    for (auto& s : tn) {
      double ms = timeit([&] { SX(simx_gemm_tn_planes(0, s.M, s.N, T, Ab, s.M, (long)T * s.M, Xb, s.N, (long)T * s.N, G, s.N, 1, ws, wsb, bias)); }, 5);
      double fl = 2.0 * T * s.M * s.N; tot += ms; totf += fl;
      printf("xp gemm_tn+bias %-31s %8.3f ms  %7.1f TF/s\n", s.name, ms, fl / ms / 1e9);
    }
    printf("xp gemm_tn total %.2f ms  avg %.1f TF/s\n", tot, totf / tot / 1e9);
    tot = 0; totf = 0;
    for (auto& s : tn) {
      double ms = timeit([&] { SX(simx_gemm_tn_planes(0, s.M, s.N, T, Ab, s.M, (long)T * s.M, Xb, s.N, (long)T * s.N, G, s.N, 1, ws, wsb, nullptr)); }, 5);
      double fl = 2.0 * T * s.M * s.N; tot += ms; totf += fl;
      printf("xp gemm_tn (no bias) %-26s %8.3f ms  %7.1f TF/s\n", s.name, ms, fl / ms / 1e9);
    }
    printf("xp gemm_tn (no bias) total %.2f ms  avg %.1f TF/s\n", tot, totf / tot / 1e9);
    double ms = timeit([&] { SX(simx_planes_from(0, SIMX_F16, SIMX_BF16, T, F, Ah, F, (long)T * F, Xb, F, (long)T * F)); }, 5);
    printf("xp planes f16->bf16 [T,%d] %8.3f ms  %6.2f TB/s\n", F, ms, 8.0 * T * F / ms / 1e9);
    ms = timeit([&] { SX(simx_planes_from(0, SIMX_F32, SIMX_BF16, T, H, C, H, 0, Xb, H, (long)T * H)); }, 5);
    printf("xp planes f32->bf16 [T,%d] %8.3f ms  %6.2f TB/s\n", H, ms, 8.0 * T * H / ms / 1e9);
    return 0;
  }
  if (getenv("KB_X3")) {      // the fp32 engine's dense GEMMs: f32 tensors, hi+lo split products (SIMX_F32_SPLIT_H / _B)
    auto falloc = [&](size_t n, float scale) { float* p; CK(hipMalloc(&p, n * 4)); std::vector<float> h(1 << 20);
      for (size_t i = 0; i < h.size(); ++i) h[i] = (((int)((i * 2654435761u) >> 20 & 1023)) - 512) / 1024.0f * scale;
      for (size_t off = 0; off < n; off += h.size()) CK(hipMemcpy(p + off, h.data(), std::min(h.size(), n - off) * 4, hipMemcpyHostToDevice)); return p; };
    float* A = falloc((size_t)T * F, 1.f); float* A2 = falloc((size_t)T * F, 1.f); float* C = falloc((size_t)T * F, 0.f); float* C2 = falloc((size_t)T * F, 0.f);
    float* W = falloc((size_t)F * H, 0.05f); float* bias = falloc(F, 0.f); float* G = falloc((size_t)F * H, 0.f);
    size_t wsb = simx_gemm_tn_workspace_bytes(F, H, T); void* ws; CK(hipMalloc(&ws, wsb + 256));
    struct S { const char* name; int N, K, epi, res, code; } nt[] = {
      {"qkv   fwd  N=2304 K=768 bias", 3 * H, H, 0, 0, 3}, {"oproj fwd  N=768  K=768 bias+res", H, H, 0, 1, 3},
      {"ffn1  fwd  N=3072 K=768 gelu", F, H, 1, 0, 3},     {"ffn2  fwd  N=768  K=3072 bias+res", H, F, 0, 1, 3},
      {"ffn2 dgrad N=3072 K=768 dgelu", F, H, 2, 0, 4},    {"ffn1 dgrad N=768  K=3072 +res", H, F, 0, 1, 4},
      {"qkv  dgrad N=768  K=2304 +res", H, 3 * H, 0, 1, 4}};
    double tot = 0, totf = 0;
    for (auto& s : nt) {
      double ms = timeit([&] { SX(simx_gemm_nt(0, s.code, T, s.N, s.K, A, s.K, W, s.K, C, s.N, s.epi == 2 ? nullptr : bias, s.res ? A2 : nullptr, s.N, s.epi, s.epi == 2 ? A2 : nullptr, s.N, s.epi == 1 ? C2 : nullptr, s.N)); }, 5);
      double fl = 2.0 * T * s.N * s.K; tot += ms; totf += fl;
      printf("x3 gemm_nt %-36s %8.3f ms  %7.1f TF/s\n", s.name, ms, fl / ms / 1e9);
    }
    printf("x3 gemm_nt total %.2f ms  avg %.1f TF/s\n", tot, totf / tot / 1e9);
    struct S2 { const char* name; int M, N; } tn[] = {{"wqkv [2304,768]", 3 * H, H}, {"wo [768,768]", H, H}, {"w1 [3072,768]", F, H}, {"w2 [768,3072]", H, F}};
    tot = 0; totf = 0;
    for (auto& s : tn) {
      double ms = timeit([&] { SX(simx_gemm_tn(0, 4, s.M, s.N, T, A, s.M, A2, s.N, G, s.N, 1, ws, wsb)); }, 5);
      double fl = 2.0 * T * s.M * s.N; tot += ms; totf += fl;
      printf("x3 gemm_tn %-36s %8.3f ms  %7.1f TF/s\n", s.name, ms, fl / ms / 1e9);
    }
    printf("x3 gemm_tn total %.2f ms  avg %.1f TF/s\n", tot, totf / tot / 1e9);
    return 0;
  }
  const int pad = getenv("KB_LDA_PAD") ? atoi(getenv("KB_LDA_PAD")) : 0;   // experiment: leading-dimension padding of A
  void* A = dalloc((size_t)T * (F + pad) * 2, 1);      // activations (bf16), up to [T,F]
  void* A2 = dalloc((size_t)(T + 1024) * F * 2, 1);
  void* C = dalloc((size_t)(T + 1024) * F * 2, 0);
  void* C2 = dalloc((size_t)(T + 1024) * F * 2, 0);
  void* W = dalloc((size_t)F * H * 2 * 2, 1);
  float* bias = (float*)dalloc(F * 4 * 2, 0);
  float* G = (float*)dalloc((size_t)F * H * 4, 0);
  size_t wsb = 0; { size_t v; v = simx_gemm_tn_workspace_bytes(3*H, H, T); wsb = v; v = simx_gemm_tn_workspace_bytes(F, H, T); if (v > wsb) wsb = v; v = simx_gemm_tn_workspace_bytes(H, F, T); if (v > wsb) wsb = v; }
  void* ws = dalloc(wsb + 256, 0);
  struct S { const char* name; int N, K, epi, res; } nt[] = {
    {"qkv   fwd  N=2304 K=768 bias", 3 * H, H, 0, 0}, {"oproj fwd  N=768  K=768 bias+res", H, H, 0, 1},
    {"ffn1  fwd  N=3072 K=768 gelu", F, H, 1, 0},     {"ffn2  fwd  N=768  K=3072 bias+res", H, F, 0, 1},
    {"ffn2 dgrad N=3072 K=768 dgelu", F, H, 2, 0},    {"ffn1 dgrad N=768  K=3072 +res", H, F, 0, 1},
    {"qkv  dgrad N=768  K=2304 +res", H, 3 * H, 0, 1}};
  double tot = 0, totf = 0;
  const char* only = getenv("KB_ONLY"); int oi = only ? atoi(only) : -1; int idx = -1;
  for (auto& s : nt) {
    ++idx; if (oi >= 0 && idx != oi) continue;
    double ms = timeit([&] { SX(simx_gemm_nt(0, g_dt, T, s.N, s.K, A, s.K + pad, W, s.K, C, s.N, s.epi == 2 ? nullptr : bias, s.res ? A2 : nullptr, s.N, s.epi, s.epi == 2 ? A2 : nullptr, s.N, s.epi == 1 ? C2 : nullptr, s.N)); }, iters);
    double fl = 2.0 * T * s.N * s.K; tot += ms; totf += fl;
    print_timeline();
    if (base) printf("gemm_nt %-36s %8.3f ms  %7.1f TF/s\n", s.name, ms, fl / ms / 1e9);
    else printf("gemm_nt %-10.10s N=%-5d K=%-5d epi %d res %d %8.3f ms  %7.1f TF/s\n", s.name, s.N, s.K, s.epi, s.res, ms, fl / ms / 1e9);
  }
  printf("gemm_nt total %.2f ms  avg %.1f TF/s\n", tot, totf / tot / 1e9);
  if (oi >= 0) return 0;
  if (base) {   // QKV projection writing the head-major layout ([36][T][64]) and the dgrad reading it
    double ms = timeit([&] { SX(simx_gemm_nt_hm(0, g_dt, T, 3 * H, H, A, H, W, H, C, 64, bias, nullptr, 0, nullptr, 0, T)); }, iters);
    printf("gemm_nt %-36s %8.3f ms  %7.1f TF/s\n", "qkv   fwd  N=2304 K=768 bias -> hm", ms, 2.0 * T * 3 * H * H / ms / 1e9);
    ms = timeit([&] { SX(simx_gemm_nt_hm(0, g_dt, T, H, 3 * H, A, 64, W, 3 * H, C, H, nullptr, A2, H, nullptr, T, 0)); }, iters);
    printf("gemm_nt %-36s %8.3f ms  %7.1f TF/s\n", "qkv  dgrad N=768  K=2304 +res <- hm", ms, 2.0 * T * 3 * H * H / ms / 1e9);
  }
  if (base) {   // how much would plane-blocked [T, 768] outputs be worth?  (no residual: the built flags-2 form)
    double ms = timeit([&] { SX(simx_gemm_nt(0, g_dt, T, H, H, A, H, W, H, C, H, bias, nullptr, 0, 0, nullptr, 0, nullptr, 0)); }, iters);
    printf("gemm_nt %-36s %8.3f ms  %7.1f TF/s\n", "oproj-shape N=768 K=768 bias", ms, 2.0 * T * H * H / ms / 1e9);
    ms = timeit([&] { SX(simx_gemm_nt_pb(0, g_dt, T, H, H, A, H, W, H, C, 64, bias, nullptr, 0, 0, nullptr, 0, nullptr, 2, T)); }, iters);
    printf("gemm_nt %-36s %8.3f ms  %7.1f TF/s\n", "oproj-shape N=768 K=768 bias -> pb", ms, 2.0 * T * H * H / ms / 1e9);
    ms = timeit([&] { SX(simx_gemm_nt(0, g_dt, T, H, F, A, F, W, F, C, H, bias, nullptr, 0, 0, nullptr, 0, nullptr, 0)); }, iters);
    printf("gemm_nt %-36s %8.3f ms  %7.1f TF/s\n", "ffn2-shape N=768 K=3072 bias", ms, 2.0 * T * H * F / ms / 1e9);
    ms = timeit([&] { SX(simx_gemm_nt_pb(0, g_dt, T, H, F, A, F, W, F, C, 64, bias, nullptr, 0, 0, nullptr, 0, nullptr, 2, T)); }, iters);
    printf("gemm_nt %-36s %8.3f ms  %7.1f TF/s\n", "ffn2-shape N=768 K=3072 bias -> pb", ms, 2.0 * T * H * F / ms / 1e9);
    ms = timeit([&] { SX(simx_gemm_nt(0, g_dt, T, F, H, A, H, W, H, C, F, bias, nullptr, 0, 0, nullptr, 0, nullptr, 0)); }, iters);
    printf("gemm_nt %-36s %8.3f ms  %7.1f TF/s\n", "ffn1-shape N=3072 K=768 bias (plain)", ms, 2.0 * T * H * F / ms / 1e9);
    ms = timeit([&] { SX(simx_gemm_nt_pb(0, g_dt, T, F, H, A, H, W, H, C, 64, bias, nullptr, 0, 0, nullptr, 0, nullptr, 2, T)); }, iters);
    printf("gemm_nt %-36s %8.3f ms  %7.1f TF/s\n", "ffn1-shape N=3072 K=768 bias -> pb", ms, 2.0 * T * H * F / ms / 1e9);
  }
  {   // the teacher's FFN-in: GELU without the derivative output (SIMX_EPI_GELU_INFER = 3)
    double ms = timeit([&] { SX(simx_gemm_nt(0, g_dt, T, F, H, A, H + pad, W, H, C, F, bias, nullptr, F, 3, nullptr, F, C2, F)); }, iters);
    printf("gemm_nt ffn1  fwd  N=%-5d K=%-5d gelu (infer)  %8.3f ms  %7.1f TF/s\n", F, H, ms, 2.0 * T * F * H / ms / 1e9);
  }
  struct S2 { const char* name; int M, N; } tn[] = {{"wqkv [2304,768]", 3 * H, H}, {"wo [768,768]", H, H}, {"w1 [3072,768]", F, H}, {"w2 [768,3072]", H, F}};
  tot = 0; totf = 0;
  for (auto& s : tn) {
    double ms = timeit([&] { SX(simx_gemm_tn(0, g_dt, s.M, s.N, T, A, s.M, A2, s.N, G, s.N, 1, ws, wsb)); }, iters);
    double fl = 2.0 * T * s.M * s.N; tot += ms; totf += fl;
    if (base) printf("gemm_tn %-36s %8.3f ms  %7.1f TF/s\n", s.name, ms, fl / ms / 1e9);
    else printf("gemm_tn M=%-5d N=%-5d %8.3f ms  %7.1f TF/s\n", s.M, s.N, ms, fl / ms / 1e9);
  }
  printf("gemm_tn total %.2f ms  avg %.1f TF/s\n", tot, totf / tot / 1e9);
  tot = 0; totf = 0;
  for (auto& s : tn) {
    double ms = timeit([&] { SX(simx_gemm_tn_bias(0, g_dt, s.M, s.N, T, A, s.M, A2, s.N, G, s.N, 1, ws, wsb, bias)); }, iters);
    double fl = 2.0 * T * s.M * s.N; tot += ms; totf += fl;
    printf("gemm_tn+bias %-31s %8.3f ms  %7.1f TF/s\n", s.name, ms, fl / ms / 1e9);
  }
  printf("gemm_tn+bias total %.2f ms  avg %.1f TF/s\n", tot, totf / tot / 1e9);
  if (!base) return 0;
  // attention + LN + colsum at S=128
  const int S = 128, nseq = T / S, heads = 12;
  std::vector<int> cu(nseq + 1); for (int i = 0; i <= nseq; ++i) cu[i] = i * S;
  int* dcu; CK(hipMalloc(&dcu, (nseq + 1) * 4)); CK(hipMemcpy(dcu, cu.data(), (nseq + 1) * 4, hipMemcpyHostToDevice));
  float* lse = (float*)dalloc((size_t)heads * T * 4, 0);
  double ms = timeit([&] { SX(simx_mha_fwd(0, g_dt, nseq, heads, 64, dcu, S, T, A, C, lse)); }, iters);
  printf("mha_fwd S=128 %8.3f ms  %7.1f TF/s\n", ms, 4.0 * T * S * H / ms / 1e9);
  ms = timeit([&] { SX(simx_mha_bwd(0, g_dt, nseq, heads, 64, dcu, S, T, A, C, lse, A2, C2)); }, iters);
  printf("mha_bwd S=128 %8.3f ms  %7.1f TF/s (5-GEMM count)\n", ms, 10.0 * T * S * H / ms / 1e9);
  ms = timeit([&] { SX(simx_ln_fwd(0, g_dt, T, H, A, bias, bias, 1e-12f, C)); }, iters);
  printf("ln_fwd  %8.3f ms  %6.2f TB/s\n", ms, 2.0 * T * H * 2 / ms / 1e9);
  ms = timeit([&] { SX(simx_ln_bwd(0, g_dt, T, H, A, bias, 1e-12f, A2, C, G, G + 1024, G + 2048)); }, iters);
  printf("ln_bwd  %8.3f ms  %6.2f TB/s\n", ms, 3.0 * T * H * 2 / ms / 1e9);
  ms = timeit([&] { SX(simx_colsum(0, g_dt, T, F, A, F, G, 1)); }, iters);
  printf("colsum F %8.3f ms  %6.2f TB/s\n", ms, 1.0 * T * F * 2 / ms / 1e9);
  return 0;
}
