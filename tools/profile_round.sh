#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <round-tag, e.g. r01> <commit of the profiled tree>
# Produces under gpurun_out/<tag>/ : bench line (un-profiled), rocprofv3 --kernel-trace --stats of the same command,
# and two separate PMC passes (FETCH_SIZE / WRITE_SIZE) for the HBM traffic of the dominant kernel.
# tools/traffic.py then turns them into profiles/<tag>_*.  PMC passes never combine with trace domains other than
# --kernel-trace (pool rule).
tag=${1:-r01}
commit=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$tag
mkdir -p $O
rm -f $O/bench.json $O/bench_line.json        # never let an earlier run's line sit next to fresh counters
cd /tmp; export TMPDIR=/tmp
# The profiled command is the bench job with the towers on one stream (SIMX_OVERLAP_TOWERS=0) and without the
# realistic-length side measurement, so that every launch of a kernel is the headline workload's and its duration is
# exclusive -- the same conditions as bench.py's own roofline pass, which the averages must agree with.
export SIMX_OVERLAP_TOWERS=0
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-realistic --no-parity"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $CMD > $O/stats.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- $CMD > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- $CMD > $O/pmc_write.log 2>&1
# MFMA utilisation from the hardware counters (BASELINE's "MFMA util %"): busy cycles of the matrix pipes per kernel, and
# the GPU-active cycles of the same launches from a separate pass (per-kernel averages are combined in tools/traffic.py)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA --output-format csv -d $O/pmc_mfma -o p -- $CMD > $O/pmc_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_active -o p -- $CMD > $O/pmc_active.log 2>&1
# the fp32 engine (the arithmetic every shipped recipe selects) as a first-class measurement: its own un-profiled line
# (10 timed steps) and a kernel-trace summary of the same command
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp32_stats -o s -- python $R/bench.py --dtype fp32 --side --steps 3 --warmup 1 > $O/fp32_stats.log 2>&1
# HBM traffic of the fp32 engine's dominant kernel (gemm_nt_xp_kernel): the same two separate PMC passes
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fp32_pmc_fetch -o p -- python $R/bench.py --dtype fp32 --side --steps 2 --warmup 1 --no-prof > $O/fp32_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/fp32_pmc_write -o p -- python $R/bench.py --dtype fp32 --side --steps 2 --warmup 1 --no-prof > $O/fp32_pmc_write.log 2>&1
cd $R && python tools/traffic.py $tag $commit
# the fp32 engine's own line AFTER its counters exist (it quotes profiles/<tag>_fp32_traffic.json; r05's record quoted r04's)
timeout 600 python $R/bench.py --dtype fp32 --side --steps 10 --warmup 2 > $O/fp32_bench.json 2> $O/fp32_bench.err
cp $O/fp32_bench.json $R/profiles/${tag}_fp32_bench.json
# the bench line LAST, so that the counters it quotes (roofline.traffic, mfma_busy_pmc) are the ones just taken at this tree
SIMX_OVERLAP_TOWERS=1 timeout 900 python $R/bench.py --steps 5 --warmup 2 > $O/bench_line.json 2> $O/bench.err
cp $R/gpurun_out/bench_full.json $O/bench.json       # the full record (side-line kernel tables); bench_line.json = the stdout line the driver parses
# (tools/traffic.py ran BEFORE the bench and found neither file: copy them into profiles/ now, next to the counters they quote)
mkdir -p $R/profiles
cp $O/bench.json $R/profiles/${tag}_bench.json
cp $O/bench_line.json $R/profiles/${tag}_bench_line.json
mkdir -p $O/profiles && cp $R/profiles/${tag}_* $O/profiles/      # profiles/ of the box comes back through gpurun_out/
ls -la $O
