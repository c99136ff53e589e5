#!/bin/bash
# usage (GPU box, repo root): tools/profile_sidelines.sh <round-tag> <commit of the profiled tree>
# rocprofv3 --kernel-trace --stats of the bench side lines whose kernels the headline job never launches (the reranker phase: 16-head
# attention at S = 160, H = 1024 / F = 4096 dgrad / wgrad shapes; MS-MARCO Document: BERT-large at S = 512, the chunked attention
# backward, in fp16 and in fp32; PROD: the 6-layer student).  Summaries land in gpurun_out/<tag>_side/<name>_kernel_stats.csv.
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${tag}_side
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export SIMX_OVERLAP_TOWERS=0
run() {
  name=$1; shift
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o s -- python $R/bench.py --side --no-prof "$@" > $O/$name.log 2>&1
  f=$(find $O/$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${name}_kernel_stats.csv
  grep "^{" $O/$name.log | tail -1 > $O/${name}_bench.json
}
run teacher_train_step --dtype fp16 --teacher-step --teacher-arch large --steps 3 --warmup 1
run cfg5_doc_fp16 --dtype fp16 --student-arch large --qlen 128 --plen 512 --celen 512 --negs 7 --batch 16 --grad-ckpt --steps 2 --warmup 1
run cfg5_doc_fp32 --dtype fp32 --student-arch large --qlen 128 --plen 512 --celen 512 --negs 7 --batch 16 --grad-ckpt --steps 2 --warmup 1
run cfg4_prod_B8 --dtype fp16 --student-layers 6 --loss cekd --batch 8 --steps 10 --warmup 3
# the recipe to the letter (train_MS_Pas_AR2.sh: fp32, gradient checkpointing, micro-batch 16 x 16, accumulation 2, ernie-large teacher)
run recipe_of_record --dtype fp32 --grad-ckpt --batch 16 --accum 2 --teacher-arch large --steps 3 --warmup 1
# the kernel sources these traces were taken at (bench.py flags the side stubs `trace_stale` when the tree has moved on)
(cd $R && python -c "import bench, json, time; json.dump({'source': {'csrc_digest': bench.csrc_digest(), 'commit': '${2:-unknown}', 'date_utc': time.strftime('%Y-%m-%dT%H:%M:%SZ', time.gmtime())}}, open('$O/side_source.json', 'w'))")
ls -la $O
