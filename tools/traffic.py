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
# provenance of every file written here: the commit the profiled tree was built from (the GPU box has no .git: passed in by
# tools/profile_round.sh) and the date of the run -- bench.py prints it beside the counters it quotes from these files
import datetime
SOURCE = {"profile_tag": tag, "commit": sys.argv[2] if len(sys.argv) > 2 else os.environ.get("SIMX_PROFILE_COMMIT", "unknown"),
          "date_utc": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%d %H:%M")}
sys.path.insert(0, os.getcwd())
try:                                        # the kernel sources the counters were taken at (bench.py marks older files stale)
    import bench as _bench
    SOURCE["csrc_digest"] = _bench.csrc_digest()
except Exception:
    pass


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return agg


def traffic_file(fetch_dir, write_dir, name):
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    if not fetch and not write:
        return
    out = {"note": "KiB per launch, raw rocprofv3 values; hbm_bytes_per_launch = FETCH_SIZE*1024*2 (gfx950 correction, "
                   "MI355X_MICROARCH.md) + WRITE_SIZE*1024", "source": SOURCE, "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch[k]) / max(1, len(fetch[k]))
        w = sum(write[k]) / max(1, len(write[k]))
        out["kernels"][k] = {"launches": len(fetch[k]) or len(write[k]), "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                             "hbm_bytes_per_launch": round(f * 1024 * 2 + w * 1024)}
    json.dump(out, open("profiles/%s" % name, "w"), indent=1)


traffic_file(src + "/pmc_fetch", src + "/pmc_write", "%s_traffic.json" % tag)
traffic_file(src + "/fp32_pmc_fetch", src + "/fp32_pmc_write", "%s_fp32_traffic.json" % tag)      # the fp32 engine's kernels
for f in glob.glob(src + "/stats/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, "profiles/%s_kernel_stats.csv" % tag)
if os.path.exists(src + "/bench.json"):
    shutil.copy(src + "/bench.json", "profiles/%s_bench.json" % tag)
if os.path.exists(src + "/bench_line.json"):
    shutil.copy(src + "/bench_line.json", "profiles/%s_bench_line.json" % tag)
for f in glob.glob(src + "/fp32_stats/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, "profiles/%s_fp32_kernel_stats.csv" % tag)
if os.path.exists(src + "/fp32_bench.json"):
    shutil.copy(src + "/fp32_bench.json", "profiles/%s_fp32_bench.json" % tag)

# MFMA utilisation from counters: SQ_VALU_MFMA_BUSY_CYCLES (busy cycles of the matrix pipes summed over the chip's 1024
# SIMDs; 16 per v_mfma_f32_16x16x32_bf16, cross-checked against SQ_INSTS_MFMA) over the cycles the launch had available:
# GRBM_GUI_ACTIVE (summed over the 8 XCDs, separate pass) / 8 x 256 CUs x 4 SIMDs.  Per kernel name the two passes are
# combined through their per-launch averages (the workload is deterministic).  GRBM_GUI_ACTIVE / 8 / duration is the
# shader clock the kernel actually ran at (DVFS: ~1.9 GHz under MFMA load, ~2.3 GHz in the HBM-bound kernels).
def per_kernel_dur(path):
    agg = collections.defaultdict(list)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                agg[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return agg


mb, ga = per_kernel(src + "/pmc_mfma", "SQ_VALU_MFMA_BUSY_CYCLES"), per_kernel(src + "/pmc_active", "GRBM_GUI_ACTIVE")
du = per_kernel_dur(src + "/pmc_active")
if mb and ga:
    util = {"note": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs); clock_ghz = "
                    "GRBM_GUI_ACTIVE / 8 / launch duration; per-launch averages of two rocprofv3 PMC passes of the bench "
                    "command (towers on one stream); step = all kernels of the run, MFMA-free ones included",
            "source": SOURCE, "kernels": {}}
    tb = ta = 0.0
    for k in sorted(set(ga)):
        a_ = sum(ga[k])
        b_ = sum(mb.get(k, [0.0])) * len(ga[k]) / max(1, len(mb.get(k, [0.0])))
        tb += b_
        ta += a_
        if b_ > 0:
            util["kernels"][k] = {"launches": len(ga[k]), "mfma_busy_frac": round(b_ / (a_ * 128), 4)}
            if sum(du[k]) / len(du[k]) > 100e3:        # (short launches: the active count includes dispatch overhead)
                util["kernels"][k]["clock_ghz"] = round(a_ / 8 / max(1, sum(du[k])), 3)
    util["step_mfma_busy_frac"] = round(tb / (ta * 128), 4)
    big = [k for k in ga if sum(du[k]) / len(du[k]) > 100e3]
    util["step_clock_ghz"] = round(sum(sum(ga[k]) for k in big) / 8 / max(1, sum(sum(du[k]) for k in big)), 3)
    json.dump(util, open("profiles/%s_mfma_busy.json" % tag, "w"), indent=1)
    print(json.dumps(util, indent=1))
