// Optional per-launch HIP-event timing (bench.py's live roofline measurement).  Disabled by default:
// the macros cost one branch.  Events are recorded on the SAME stream the kernel is launched on.
#pragma once
#include <hip/hip_runtime.h>

enum { SIMX_K_GEMM_NT = 0, SIMX_K_GEMM_TN, SIMX_K_MHA_FWD, SIMX_K_MHA_BWD, SIMX_K_LN_FWD, SIMX_K_LN_BWD,
       SIMX_K_EMBED_FWD, SIMX_K_EMBED_BWD, SIMX_K_COLSUM, SIMX_K_CAST, SIMX_K_LOSS, SIMX_K_SAMPLER, SIMX_K_ADAMW,
       SIMX_K_OTHER, SIMX_K_COLLATE, SIMX_K_TOPK, SIMX_K_COUNT };

void simx_prof_mark(int kernel_id, hipStream_t s, double work, int end);

struct SimxProfScope {
  int id; hipStream_t s; double work;
  SimxProfScope(int id_, hipStream_t s_, double w) : id(id_), s(s_), work(w) { simx_prof_mark(id, s, work, 0); }
  ~SimxProfScope() { simx_prof_mark(id, s, work, 1); }
};
#define SIMX_PROF(id, stream, work) SimxProfScope prof_scope__((id), (hipStream_t)(stream), (double)(work))
