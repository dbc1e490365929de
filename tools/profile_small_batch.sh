#!/bin/bash
# Runs ON THE GPU BOX (gpurun): training-only kernel statistics at the reference scripts' per-GPU batch sizes
# (Bashscript/train/train_vaetf.sh: 128, train_scavaetf.sh: 64) -> gpurun_out/prof_<tag>_b<B>/kernel_stats.csv + bench lines.
#   bash tools/profile_small_batch.sh <tag> [batches...]
set -o pipefail
tag=${1:-r04}; shift
bs=${@:-"64 128"}
root=${GRAFT_REPO_ROOT:-$(pwd)}
args="--steps 30 --warmup 6 --no-cpu-baseline --no-alt-mode --no-decode --no-fixed-len-leg --no-model-types --no-trainer-loop --no-kernel-timing"
cd /tmp && export TMPDIR=/tmp
for b in $bs; do
  out=$root/gpurun_out/prof_${tag}_b$b
  mkdir -p $out
  echo "== batch $b: un-profiled line"
  python3 $root/bench.py --batch $b $args > $out/bench_line.json 2> $out/bench.err || exit 1
  echo "== batch $b: kernel stats"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --batch $b $args > $out/bench_line_profiled.json 2> $out/bench_profiled.err || exit 1
  cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
  find $out -name "*kernel_trace.csv" -delete
  python3 -c "import json;d=json.load(open('$out/bench_line.json'));print('batch $b', d['ms_per_step'],'ms/step', d['value'],'SMILES/s')"
done
