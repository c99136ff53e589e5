#!/usr/bin/env python
"""Where does a tile of the persistent NT GEMM spend its time?  (measurement tool, not part of the product)

Writes an INSTRUMENTED copy of simxns_amd/csrc/gemm.hip to tools/variants/TS/ (the product source is not touched), in which waves 0
and 5 of one workgroup record clock64() at four points of each of their first 16 tiles --
    0: top of the tile loop            1: last k-step done, epilogue begins
    2: epilogue done (stores issued)   3: first stage boundary of the NEXT tile passed (vmcnt + barrier)
-- builds it into tools/variants/TS/libsimx_hip.so + kbench, and `KB_TS=1 tools/variants/TS/kbench` prints per shape the cycles of the
main loop (0 -> 1), of the epilogue (1 -> 2), from the end of the epilogue to the first boundary of the next tile (2 -> 3), and the
tile period.  Usage (build container): python tools/p3_timeline.py ; (GPU box) KB_F16=1 KB_TS=1 tools/variants/TS/kbench 262144"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "simxns_amd", "csrc", "gemm.hip")).read()

decl = r'''
// ---- tools/p3_timeline.py instrumentation ----
__device__ unsigned long long simx_p3_ts[2 * 16 * 4];
#define P3_TS(P) do { if (ts_on && ts_tile < 16) simx_p3_ts[((wave == 5 ? 16 : 0) + ts_tile) * 4 + (P)] = clock64(); } while (0)
extern "C" int simx_debug_p3_ts(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(simx_p3_ts), sizeof(unsigned long long) * 2 * 16 * 4) == hipSuccess ? 0 : 1;
}
'''
anchor = "template <typename F, int EPI, bool HAS_IN, int HMF = 0>\n__global__ __launch_bounds__(512, 2) void gemm_nt_p3_kernel("
assert src.count(anchor) == 1
src = src.replace(anchor, decl + anchor)


def once(s, old, new):
    assert s.count(old) == 1, (s.count(old), old[:60])
    return s.replace(old, new)


src = once(src, "  int v = blockIdx.x;\n  int tile = xcd_remap(v, ntiles);",
           "  const bool ts_on = blockIdx.x == 11 && lane == 0 && (wave == 0 || wave == 5);\n  int ts_tile = 0;\n  int v = blockIdx.x;\n  int tile = xcd_remap(v, ntiles);")
src = once(src, "    const int vn = v + (int)gridDim.x;\n    const bool has_next = vn < ntiles;", "    P3_TS(0);\n    const int vn = v + (int)gridDim.x;\n    const bool has_next = vn < ntiles;")
src = once(src, "    // ---- epilogue, 8 chunks of 16 rows (= accumulator row-block i), per wave, no barrier", "    P3_TS(1);\n    // ---- epilogue, 8 chunks of 16 rows")
src = once(src, "    // the borrowed A slot is free again (this wave's slice only ever holds this wave's rows): next tile's stage 2", "    P3_TS(2);\n    ++ts_tile;\n    // the borrowed A slot is free again")
# point 3: after the first mid-tile boundary of a tile (st == 0) -- belongs to the PREVIOUS tile's record
src = once(src, "    P3_BOUNDARY_WAIT();                                                                        \\\n    const bool cb__ = st + 2 < nst, ca__ = st + 3 < nst;",
           "    P3_BOUNDARY_WAIT();                                                                        \\\n    if (st == 0 && ts_on && ts_tile >= 1 && ts_tile <= 16) simx_p3_ts[((wave == 5 ? 16 : 0) + ts_tile - 1) * 4 + 3] = clock64(); \\\n    const bool cb__ = st + 2 < nst, ca__ = st + 3 < nst;")

args = sys.argv[1:]
name = args.pop(0) if args and not args[0].startswith("-") else "TS"        # python tools/p3_timeline.py [name] [-D...]
d = os.path.join(ROOT, "tools", "variants", name)
os.makedirs(d, exist_ok=True)
# the copy sits in csrc/ for the duration of the compile so that its relative #includes resolve
tmp = os.path.join(ROOT, "simxns_amd", "csrc", "_gemm_ts_%s.hip" % name)
open(tmp, "w").write(src)
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form".split()
try:
    subprocess.check_call(["hipcc"] + flags + args + ["-c", tmp, "-o", os.path.join(d, "gemm.o")])
finally:
    os.remove(tmp)
objs = [os.path.join(d, "gemm.o")] + [os.path.join(ROOT, "simxns_amd", "csrc", f + ".o") for f in
                                      "gemm_x3 gemm_xp attention attention_f32 attention_x3 layernorm loss sampler optim encoder collate retrieval det".split()]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", os.path.join(d, "libsimx_hip.so")])
subprocess.check_call(["hipcc", "-O2", os.path.join(ROOT, "tools", "kbench.cpp"), "-I" + os.path.join(ROOT, "include"), "-L" + d, "-lsimx_hip", "-ldl",
                       "-Wl,-rpath,$ORIGIN", "-o", os.path.join(d, "kbench")])
print("built", d)
