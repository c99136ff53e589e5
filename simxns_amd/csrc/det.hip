// SIMX_DETERMINISTIC=1: run-to-run bit-identical gradients.
//
// The default kernels finish their cross-workgroup reductions (LayerNorm gamma/beta/bias column sums, the bias-gradient
// column sums of the wgrad GEMM, the embedding-table scatter, the four loss scalars) with f32 atomics, whose arrival
// order -- and so the rounding of the sum -- changes from run to run.  With the switch on, every one of those sites
// writes ONE partial per contributor into a per-stream scratch buffer and a second kernel adds the partials in index
// order; the embedding scatter becomes an ownership scan (the wave that owns a table row adds that row's tokens in
// token order).  Split-K slabs, the grad-norm partials and the attention backward were ordered already.  The reference
// gets the same property from torch.use_deterministic_algorithms (it does not turn it on: SimANS/co_training/
// co_training_marco_train.py:33-44 only seeds the RNGs), so this is an operator convenience, priced in DESIGN.md.
#include <mutex>
#include <unordered_map>

#include "common.h"

bool simx_det() {
  static const bool on = [] { const char* e = getenv("SIMX_DETERMINISTIC"); return e && e[0] && e[0] != '0'; }();
  return on;
}

namespace {
struct DetBuf { float* p = nullptr; size_t bytes = 0; };
std::mutex det_mu;
std::unordered_map<hipStream_t, DetBuf> det_bufs;       // one scratch buffer per stream: kernels of one stream are ordered
}  // namespace

// Scratch of the deterministic mode, owned by the library, one buffer per stream, grown on demand (hipFree of the old
// buffer synchronises the device, so kernels still reading it have finished).  Growth is impossible while the stream is
// being captured into a graph: warm the shapes up eagerly first (bench.py and the train scripts do).
float* simx_det_ws(hipStream_t st, size_t bytes) {
  std::lock_guard<std::mutex> g(det_mu);
  DetBuf& b = det_bufs[st];
  if (b.bytes >= bytes) return b.p;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    simx_set_error("deterministic mode: scratch must grow to %zu bytes during graph capture; run the shape eagerly once first", bytes);
    return nullptr;
  }
  const size_t want = bytes + bytes / 4;
  float* np = nullptr;
  if (hipMalloc(&np, want) != hipSuccess) { simx_set_error("deterministic mode: hipMalloc(%zu) failed", want); return nullptr; }
  if (b.p) (void)hipFree(b.p);
  b.p = np;
  b.bytes = want;
  return np;
}

// out[j][c] += inv * sum_b part[b * stride + j * n + c], b ascending.  64 columns x 16 block-groups per workgroup: group q adds
// its contiguous range of partials in order, then the 16 group sums are added in order -- a fixed tree for a fixed launch.
struct DetOuts { float* o[4]; };
__global__ __launch_bounds__(1024) void det_reduce_kernel(const float* __restrict__ part, long stride, int nparts, int n, DetOuts outs,
                                                          const float* __restrict__ gs) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, j = blockIdx.y;
  const int per = (nparts + 15) / 16;
  const int b0 = q * per, b1 = min(nparts, b0 + per);
  float s = 0.f;
  if (c < n) {
    const float* p = part + (long)j * n + c;
    for (int b = b0; b < b1; ++b) s += p[(long)b * stride];
  }
  red[q][cl] = s;
  __syncthreads();
  if (q == 0 && c < n && outs.o[j]) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cl];
    outs.o[j][c] += t * gs_inv(gs);
  }
}

int simx_det_reduce(hipStream_t st, const float* part, long stride, int nparts, int n, float* o0, float* o1, float* o2, const float* gs) {
  DetOuts outs{{o0, o1, o2, nullptr}};
  const int ny = o2 ? 3 : (o1 ? 2 : 1);
  hipLaunchKernelGGL(det_reduce_kernel, dim3(cdiv(n, 64), ny), dim3(1024), 0, st, part, stride, nparts, n, outs, gs);
  SIMX_CHECK_LAUNCH("det_reduce");
  return SIMX_OK;
}

// Ownership scan of a row scatter: table[idx[t]] += rows[t] for t ascending.  Wave g of G owns the table rows with
// idx % G == g, scans all T indices 64 at a time and adds the rows it owns with plain loads and stores -- nobody else
// touches them, and the tokens of one table row are added in token order.
__global__ __launch_bounds__(256) void det_scatter_rows_kernel(int T, int H, const int* __restrict__ idx, const float* __restrict__ rows,
                                                               float* __restrict__ table) {
  const int lane = threadIdx.x & 63;
  const unsigned g = blockIdx.x * 4 + (threadIdx.x >> 6), G = gridDim.x * 4;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    const int id = t < T ? idx[t] : -1;
    unsigned long long m = __ballot(id >= 0 && (unsigned)id % G == g);
    while (m) {
      const int k = __builtin_ctzll(m);
      m &= m - 1;
      const long row = __shfl(id, k, 64);
      const float* src = rows + (long)(t0 + k) * H;
      float* dst = table + row * H;
      for (int c = lane * 4; c < H; c += 256) {
        const float4 a = *reinterpret_cast<const float4*>(src + c);
        float4 d = *reinterpret_cast<float4*>(dst + c);
        d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
        *reinterpret_cast<float4*>(dst + c) = d;
      }
    }
  }
}

int simx_det_scatter_rows(hipStream_t st, int T, int H, const int* idx, const float* rows, float* table, int table_rows) {
  int blocks = cdiv(table_rows, 4);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(det_scatter_rows_kernel, dim3(blocks), dim3(256), 0, st, T, H, idx, rows, table);
  SIMX_CHECK_LAUNCH("det_scatter_rows");
  return SIMX_OK;
}

extern "C" int simx_deterministic(void) { return simx_det() ? 1 : 0; }
