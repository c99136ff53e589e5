#!/bin/bash
# usage: tools/pmc_l2.sh <tag> <cmd...> : one PMC pass with the L2 / fabric read counters
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 120 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/pmc_${tag}_b -o p -- "$@" > /dev/null 2>&1
