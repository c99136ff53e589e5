#!/bin/bash
# Refetch-vs-clock A/B: gemm_nt_p3 under its default XCD-aware row-major tile order (variant A) and with every
# workgroup walking the N-tiles of its own M-panels (variant P, -DSIMX_P3_PANEL_ORDER): time, HBM fetch bytes and clock per launch.
# usage (GPU box, repo root, after tools/build_variant.sh A / P ...): tools/refetch_ab.sh
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/refetch_ab
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export KB_F16=1 SIMX_P5=0
for v in A P; do
  K=$R/tools/variants/$v/kbench
  for i in 1 2 3; do (cd $R/tools/variants/$v && ./kbench 262144 2>&1 | grep -E "^gemm_nt (qkv|ffn1|ffn2|oproj)" ) > $O/time_${v}_$i.log; done
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$v -o p -- $K 262144 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/active_$v -o p -- $K 262144 > /dev/null 2>&1
  ( $K 262144 > /dev/null 2>&1 & sleep 4; rocm-smi --showpower 2>/dev/null | grep -i "power" | head -3 > $O/power_$v.log; wait )
done
python3 - <<PY
import csv, glob, collections, json
out = {}
for v in "AP":
    d = {}
    for tag, cname in (("fetch", "FETCH_SIZE"), ("active", "GRBM_GUI_ACTIVE")):
        f = glob.glob("$O/%s_%s/**/*counter_collection.csv" % (tag, v), recursive=True)
        t = glob.glob("$O/%s_%s/**/*kernel_trace.csv" % (tag, v), recursive=True)
        if not f or not t: continue
        dur = {}
        for r in csv.DictReader(open(t[0])):
            dur.setdefault(r["Kernel_Name"][:48], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == cname: agg[r["Kernel_Name"][:48]].append(float(r["Counter_Value"]))
        for k, vals in agg.items():
            if "gemm_nt_p3" not in k: continue
            e = d.setdefault(k, {})
            e[cname] = sum(vals) / len(vals)
            e[cname + "_ms"] = sum(dur[k]) / len(dur[k]) / 1e6
    out[v] = d
json.dump(out, open("$O/summary.json", "w"), indent=1)
for v in "AP":
    for k, e in sorted(out[v].items()):
        fb = e.get("FETCH_SIZE", 0) * 1024 / 1e9      # KiB -> GB (guide: FETCH_SIZE counts KiB on this part after the /?? correction of tools/traffic.py is NOT applied here: relative use only)
        ghz = e.get("GRBM_GUI_ACTIVE", 0) / 8 / (e.get("GRBM_GUI_ACTIVE_ms", 1) * 1e6)
        print(v, k, "fetch %.3f GB/launch  %.3f ms  clock %.3f GHz" % (fb, e.get("FETCH_SIZE_ms", 0), ghz))
PY
for v in A P; do echo "== $v"; cat $O/time_${v}_*.log | sort | awk '{k=$2" "$3" "$4" "$5; s[k]+=$(NF-3); n[k]++} END{for(k in s) printf "%s  %.4f ms\n", k, s[k]/n[k]}' | sort; cat $O/power_$v.log; done
