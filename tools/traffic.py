"""Post-process tools/profile_round.sh output into profiles/<tag>_{kernel_stats.csv,bench.json,traffic.json}.

HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and
WRITE_SIZE are collected in their own passes and reported in KiB; on gfx950 FETCH_SIZE counts 64-B units for what
are 128-B requests on wide coalesced streams, so reads are corrected x2 (calibrated here on ln_fwd, whose
algorithmic read bytes are known exactly: T*H*2); WRITE_SIZE needs no correction (same calibration)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = "gpurun_out/%s" % tag
os.makedirs("profiles", exist_ok=True)


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return agg


fetch, write = per_kernel(src + "/pmc_fetch", "FETCH_SIZE"), per_kernel(src + "/pmc_write", "WRITE_SIZE")
out = {"note": "KiB per launch, raw rocprofv3 values; hbm_bytes_per_launch = FETCH_SIZE*1024*2 (gfx950 correction, "
               "MI355X_MICROARCH.md) + WRITE_SIZE*1024", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f = sum(fetch[k]) / max(1, len(fetch[k]))
    w = sum(write[k]) / max(1, len(write[k]))
    out["kernels"][k] = {"launches": len(fetch[k]) or len(write[k]), "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                         "hbm_bytes_per_launch": round(f * 1024 * 2 + w * 1024)}
json.dump(out, open("profiles/%s_traffic.json" % tag, "w"), indent=1)
for f in glob.glob(src + "/stats/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, "profiles/%s_kernel_stats.csv" % tag)
if os.path.exists(src + "/bench.json"):
    shutil.copy(src + "/bench.json", "profiles/%s_bench.json" % tag)
print(json.dumps({k: v for k, v in out["kernels"].items() if "gemm" in k}, indent=1))
