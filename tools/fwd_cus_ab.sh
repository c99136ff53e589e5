#!/bin/bash
# Forward-phase CU budgets A/B (profiles/r06_experiments/04_fwd_cus.md): the frozen teacher's forward and the student's forward run
# on two HIP streams; by default each of their persistent GEMM launches asks for all 256 CUs.  SIMX_FWD_CUS="t,s" gives the
# teacher's launches t CUs and the student's s (bench.py sets the budget per enqueue; the backward keeps every CU).
# usage (GPU box, repo root): tools/fwd_cus_ab.sh [variants...]      default variants below; "0" = no budget
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/fwd_cus_ab
mkdir -p $O
cd $R
V=${@:-"0 128,128 144,112 160,96 112,144 192,192 224,224 0"}
for v in $V; do
  for i in 1 2; do
    SIMX_FWD_CUS=$v python bench.py --side --no-prof --no-parity --no-cpu-baseline --no-realistic --no-fp32-side --steps 10 --warmup 3 2>/dev/null \
      | python -c "import sys, json; d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('SIMX_FWD_CUS=%-8s run $i  %.2f ms/step  %.1f pairs/s' % ('$v', d['ms_per_step'], d['value']))"
  done
done | tee $O/summary.log
