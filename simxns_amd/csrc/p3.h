// Shared by the persistent LDS-DMA GEMM kernels (csrc/gemm.hip: gemm_nt_p3_kernel, gemm_tn2_kernel; csrc/gemm_xp.hip: the
// plane-operand kernels of the fp32 engine): XCD-aware tile order, asm forms of the LDS-DMA / global accesses, fragment reads.
#pragma once
#include "common.h"

// ------------------------------------------------------------------------------------------
// XCD-aware tile id: block b runs on XCD b%8 (observed placement, speed only); give every XCD
// a contiguous range of logical tiles.  Bijective for any grid size.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, loc = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

#define V3_READ4(F0, F1, F2, F3, ADDR, O0, O1, O2, O3)                                              \
  asm volatile("ds_read_b128 %0, %4 offset:" #O0 "\n\tds_read_b128 %1, %4 offset:" #O1 "\n\t"       \
               "ds_read_b128 %2, %4 offset:" #O2 "\n\tds_read_b128 %3, %4 offset:" #O3               \
               : "=&v"(F0), "=&v"(F1), "=&v"(F2), "=&v"(F3)                                         \
               : "v"(ADDR)                                                                          \
               : "memory")
#define V3_PIN4(TXT, F0, F1, F2, F3) asm volatile(TXT : "+v"(F0), "+v"(F1), "+v"(F2), "+v"(F3)::"memory")
#define V3_PIN8(TXT, F0, F1, F2, F3, F4, F5, F6, F7) \
  asm volatile(TXT : "+v"(F0), "+v"(F1), "+v"(F2), "+v"(F3), "+v"(F4), "+v"(F5), "+v"(F6), "+v"(F7)::"memory")

#define V5_STAGE 65536
// ------------------------------------------------------------------------------------------
// Shared by the persistent kernels below (gemm_nt_bf16_p3_kernel, gemm_tn2_bf16_kernel): 160 KB of dynamic LDS and global
// accesses in "uniform 64-bit base in SGPRs + 32-bit per-lane offset" form.
// ------------------------------------------------------------------------------------------
#define P_EPI_OFF (2 * V5_STAGE)
#define P_LDS (P_EPI_OFF + 8 * 4096)
// global accesses in "uniform 64-bit base in SGPRs + 32-bit per-lane offset" form: one VGPR per address stream
#define P_GLD4(DST, VOFF, SBASE, OFF) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #OFF : "=&v"(DST) : "v"(VOFF), "s"(SBASE) : "memory")
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// (the trailing s_nop covers the gfx9 "VMEM store of > 64 bits, then a write of its data VGPRs" hazard, which hipcc's
// hazard recognizer cannot see for a store hidden in inline asm.  One wait state was NOT enough on gfx950: with
// s_nop 0 the first data dword of lanes 12-15 of every 16-lane group was still clobbered by the next VALU write in some
// schedules (tests/test_kernels_gpu.py::test_gemm_nt_persistent caught it) -- the store reads its data 16 lanes x 1 dword per
// cycle; 4 wait states cleared it, 6 are used)
// (the persistent kernel's output stores are non-temporal: the tile is not re-read by this kernel, and with `nt` its lines leave
// the L2 before the operand lines the neighbouring CUs still share -- two-output / N = 3072 shapes 2 % faster, the rest equal;
// "sc0 sc1" (write-through) changes nothing.  SIMX_P3_SAMEC, every tile storing to the same rows so that no store reaches HBM,
// shows what is left: the seven shapes run 6 % faster, QKV 11 % -- a store is acknowledged only when the L2 has room, the 32 CUs
// of an XCD store 4 MB = the whole L2 within a few microseconds, and gfx9's single in-order vmcnt makes the next tile's first
// stage boundary wait for those acknowledgements -- or so it seemed: waiting for the next tile's stage 1 BEFORE the stores
// and passing that first boundary with a bare s_barrier changed nothing (6.73 vs 6.73 ms), so the cost is in the store
// path's back-pressure on the issuing waves themselves.  Skewing the XCDs against each other does not help (per-XCD burst unchanged),
// starting the A panels of an XCD in 4 phase groups (N-tiles of a panel in step) costs its 3/4-tile tail and gains nothing.)
#define SIMX_P3_STORE_BITS " nt"
#define P_GST4(VOFF, SBASE, VAL) asm volatile("global_store_dwordx4 %0, %1, %2" SIMX_P3_STORE_BITS "\n\ts_nop 5" ::"v"(VOFF), "v"(VAL), "s"(SBASE) : "memory")
// LDS-DMA in the same form (M0 = wave-uniform LDS byte address of the 1 KB destination).  The persistent kernel
// uses ONLY this form, so the compiler never tracks M0 in it.
#define P_DMA16(VOFF, SBASE, LDSADDR) \
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(VOFF), "s"(SBASE), "s"(LDSADDR) : "memory")
// ------------------------------------------------------------------------------------------
// Persistent NT kernel, three A stages deep.  tools/vmem_bench shows what bounds the two-stage kernel's main loop: the
// LDS-DMA path delivers ~36 B/clk/CU with two 64 KB stages IN FLIGHT but only ~18 with one, and the two-stage ring has
// exactly one in flight (the other is being consumed) -- throughput = bytes in flight / latency.  Here the A operand
// (activations: HBM / fabric latency, L2 never hits) gets three 32 KB slots, B (weights: L2 hits) two: after a stage
// boundary A(st+2), B(st+2) and A(st+3) are in flight = 96 KB, inside the same 160 KB of LDS.  The epilogue has no LDS
// of its own: it borrows the A slot freed by the tile's last stage -- wave w stages through bytes [4w, 4w+4) KB of that
// slot, which are exactly the bytes wave w's own share of the next A load will overwrite, so each wave re-issues its
// share of that load right after its own epilogue, with no cross-wave synchronisation.
// Every stage boundary waits vmcnt(4): the issue order per wave is ... [B(s+1)(4) A(s+2)(4)] so "all but the 4 youngest"
// = stage s+1 complete.  Needs K >= 256.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void p3_half(const bf16_t* __restrict__ G, int ld, int row0, long k0, uint32_t slot, int wave,
                                        uint32_t off0, uint32_t off1) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = wave * 4 + j;
    const char* sg = reinterpret_cast<const char*>(G + (long)(row0 + i * 8) * ld + k0);
    P_DMA16((j & 1) ? off1 : off0, sg, slot + (uint32_t)(i * 1024));
  }
}

// ---- the 256x256 TN (wgrad) kernels' stage helpers (comments at gemm_tn2_kernel, csrc/gemm.hip)
#define TN2_STAGE 65536
#define TN2_LDS (2 * TN2_STAGE)

// (LDS-DMA in the compiler-invisible asm form: with the builtin hipcc assumes every later LDS read may alias the pending
// DMA and waits vmcnt(0) before it; the transpose reads below are BUILTINS so that hipcc allocates both halves of a
// fragment into one 128-bit register tuple and places counted lgkmcnt waits itself -- the inline-asm reads cost 136
// v_mov per stage to assemble the tuples, more VALU time than the MFMAs of a k-step.)
// hm > 0: A is head-major, [M/64][hm][64] (dq/dk/dv, QkvLay in attention.hip): column c of token k sits at
// ((c >> 6) * hm + k) * 64 + (c & 63) -- the row pitch becomes 64 and the column part of the lane offset picks the plane.
__device__ __forceinline__ uint32_t tn2_acol(int c, int hm) { return hm > 0 ? (uint32_t)(c >> 6) * (uint32_t)hm * 64u + (uint32_t)(c & 63) : (uint32_t)c; }
__device__ __forceinline__ void tn2_stage(const bf16_t* __restrict__ A, int lda, int m0, int M,
                                          const bf16_t* __restrict__ B, int ldb, int n0, int N, int k0, int k_end,
                                          char* stage, int wave, int lane, int hm) {
  const uint32_t sbase = (uint32_t)(uintptr_t)stage;
  if (hm > 0) lda = 64;
  const char* ga = reinterpret_cast<const char*>(A + (long)k0 * lda);
  const char* gb = reinterpret_cast<const char*>(B + (long)k0 * ldb);
#pragma unroll
  for (int j = 0; j < 4; ++j) {                 // 32 wave-instructions of 2 k-rows per operand
    const int i = wave * 4 + j;
    const int kr = i * 2 + (lane >> 5);
    const int p16 = lane & 31;
    const int q = (p16 >> 1) ^ (kr & 7);
    int rk = kr;                                 // row relative to k0, clamped into the k-range (ragged tail rows are zeroed later)
    rk = k0 + rk < k_end ? rk : k_end - 1 - k0;
    int ca = m0 + q * 16 + (p16 & 1) * 8, cb = n0 + q * 16 + (p16 & 1) * 8;
    ca = ca + 8 <= M ? ca : M - 8;
    cb = cb + 8 <= N ? cb : N - 8;
    P_DMA16(((uint32_t)(rk * lda) + tn2_acol(ca, hm)) * 2, ga, sbase + (uint32_t)(i * 1024));
    P_DMA16((uint32_t)(rk * ldb + cb) * 2, gb, sbase + (uint32_t)(32768 + i * 1024));
  }
}

// the same for a FULL stage (all 64 k-rows inside the k-range) with the per-lane byte offsets precomputed once per kernel
// (tn2_lane_offsets): the row part of the address rides in the SGPR base, so a stage costs no VALU work at all.
// tn2_stage above recomputes offsets and clamps per stage (~60 VALU per wave) and is kept for the ragged last stage.
__device__ __forceinline__ void tn2_lane_offsets(int lda, int m0, int M, int ldb, int n0, int N, int lane, uint32_t (&oa)[4],
                                                 uint32_t (&ob)[4], int hm) {
  if (hm > 0) lda = 64;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int krl = 2 * j + (lane >> 5);                      // k-row within the wave's 8 rows; (kr & 7) == krl
    const int p16 = lane & 31;
    const int q = (p16 >> 1) ^ (krl & 7);
    int ca = m0 + q * 16 + (p16 & 1) * 8, cb = n0 + q * 16 + (p16 & 1) * 8;
    ca = ca + 8 <= M ? ca : M - 8;
    cb = cb + 8 <= N ? cb : N - 8;
    oa[j] = ((uint32_t)((lane >> 5) * lda) + tn2_acol(ca, hm)) * 2;
    ob[j] = (uint32_t)((lane >> 5) * ldb + cb) * 2;
  }
}
__device__ __forceinline__ void tn2_stage_full(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, int k0,
                                               uint32_t sbase, int wave, const uint32_t (&oa)[4], const uint32_t (&ob)[4]) {
  const char* ga = reinterpret_cast<const char*>(A + (long)(k0 + wave * 8) * lda);
  const char* gb = reinterpret_cast<const char*>(B + (long)(k0 + wave * 8) * ldb);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = wave * 4 + j;
    P_DMA16(oa[j], ga + (long)(2 * j) * lda * 2, sbase + (uint32_t)(i * 1024));
    P_DMA16(ob[j], gb + (long)(2 * j) * ldb * 2, sbase + (uint32_t)(32768 + i * 1024));
  }
}

// one quarter of a full stage (the wave's j-th A and B instruction): issued between MFMA rows instead of as a burst
__device__ __forceinline__ void tn2_stage_piece(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, int k0,
                                                uint32_t sbase, int wave, int j, uint32_t oaj, uint32_t obj) {
  const char* ga = reinterpret_cast<const char*>(A + (long)(k0 + wave * 8 + 2 * j) * lda);
  const char* gb = reinterpret_cast<const char*>(B + (long)(k0 + wave * 8 + 2 * j) * ldb);
  const int i = wave * 4 + j;
  P_DMA16(oaj, ga, sbase + (uint32_t)(i * 1024));
  P_DMA16(obj, gb, sbase + (uint32_t)(32768 + i * 1024));
}

__device__ __forceinline__ void tn2_stage_one(const bf16_t* __restrict__ G, int ld, int k0, uint32_t sbase, int wave, int j, uint32_t off) {
  const char* g = reinterpret_cast<const char*>(G + (long)(k0 + wave * 8 + 2 * j) * ld);
  P_DMA16(off, g, sbase + (uint32_t)((wave * 4 + j) * 1024));
}

// one fragment = two transpose reads (k rows base+4g.. and base+16+4g..)
typedef __attribute__((address_space(3))) bf16x4* tn_lds4_t;
#define TN2_RD(F_LO, F_HI, ADDR_LO, ADDR_HI)                                                        \
  do {                                                                                              \
    F_LO = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn_lds4_t)(uintptr_t)(ADDR_LO));                \
    F_HI = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn_lds4_t)(uintptr_t)(ADDR_HI));                \
  } while (0)

