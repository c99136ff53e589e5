// Optional per-launch HIP-event timing (bench.py's live roofline measurement).  Disabled by default:
// the macros cost one branch.  Events are recorded on the SAME stream the kernel is launched on.
#pragma once
#include <hip/hip_runtime.h>

enum { SIMX_K_GEMM_NT = 0, SIMX_K_GEMM_TN, SIMX_K_MHA_FWD, SIMX_K_MHA_BWD, SIMX_K_LN_FWD, SIMX_K_LN_BWD,
       SIMX_K_EMBED_FWD, SIMX_K_EMBED_BWD, SIMX_K_COLSUM, SIMX_K_CAST, SIMX_K_LOSS, SIMX_K_SAMPLER, SIMX_K_ADAMW,
       SIMX_K_OTHER, SIMX_K_COLLATE, SIMX_K_TOPK, SIMX_K_GEMM_NT_P3, SIMX_K_GEMM_TN2, SIMX_K_GEMM_NT_XP, SIMX_K_GEMM_TN_XP, SIMX_K_COUNT };

// begin: returns the record's index (or -1 when recording is off); end: closes that record
int simx_prof_mark(int kernel_id, hipStream_t s, double work, int end_index);
// re-labels the innermost open record (the NT dispatcher tags launches of the persistent 256x256 kernel, the one the
// roofline figure is quoted for, apart from the small-shape kernels behind the same entry point)
void simx_prof_retag(int kernel_id);

struct SimxProfScope {
  int idx; hipStream_t s;
  SimxProfScope(int id_, hipStream_t s_, double w) : s(s_) { idx = simx_prof_mark(id_, s, w, -1); }
  ~SimxProfScope() { if (idx >= 0) simx_prof_mark(0, s, 0.0, idx); }
};
#define SIMX_PROF(id, stream, work) SimxProfScope prof_scope__((id), (hipStream_t)(stream), (double)(work))
