#!/bin/bash
# Same-box A/B of one environment switch on the headline job: tools/ab_env.sh NAME "v0 v1 ..." [repeats] [extra bench.py flags]
# (bench.py --side --steps 10 --warmup 3, un-instrumented; one line per run)
R=${GRAFT_REPO_ROOT:-$PWD}
name=$1; vals=$2; reps=${3:-3}; shift 3 2>/dev/null
mkdir -p $R/gpurun_out/ab; cd $R
for i in $(seq 1 $reps); do for v in $vals; do
  env $name=$v python bench.py --side --no-prof --no-parity --no-cpu-baseline --no-realistic --no-fp32-side --steps 10 --warmup 3 "$@" 2>/dev/null \
    | python -c "import sys, json; d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$name=%-6s run $i  %.2f ms/step  %.1f pairs/s' % ('$v', d['ms_per_step'], d['value']))"
done; done | tee $R/gpurun_out/ab/$name.log
