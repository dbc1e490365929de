#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the profiles a round commits under profiles/.
#   bash tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/{stats,pmc_*}/... + bench line
# rocprofv3 gets the program itself after `--` (python3 bench.py ...): no wrappers (the preloaded profiler library
# initialises the GPU before the program starts).  Counter passes are separate from --stats and from each other.
set -o pipefail
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
echo "== kernel stats of the default bench command"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py > $out/bench_line_profiled.json 2> $out/bench_profiled.err || exit 1
echo "== kernel stats of the training step alone (13 + 3 steps, no other leg in the process)"
only="--steps 13 --warmup 3 --no-cpu-baseline --no-alt-mode --no-decode --no-fixed-len-leg --no-model-types --no-trainer-loop --no-kernel-timing"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_only -- python3 $root/bench.py $only > $out/training_only_bench_line.json 2> $out/training_only.err || exit 1
cp $(find $out/stats_only -name "*kernel_stats.csv" | head -1) $out/training_only_kernel_stats.csv
args="--steps 2 --warmup 1 --no-cpu-baseline --no-alt-mode --no-decode --no-fixed-len-leg --no-model-types --no-trainer-loop --no-kernel-timing"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $grp | cut -d' ' -f1)
  echo "== pmc pass $n"
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/pmc_$n -- python3 $root/bench.py $args > $out/pmc_$n.json 2> $out/pmc_$n.err || exit 1
done
cd $root
find $out -name "*kernel_trace.csv" -size +20M -delete      # per-dispatch traces of the long run are not needed
python3 tools/pmc_summary.py --commit "$(cat $root/.gpurun_commit 2>/dev/null || echo unknown)" \
  --command "rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py $args" \
  --out-traffic $out/gemm_x6_traffic.json --out-busy $out/mfma_busy.json $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ_VALU_MFMA_BUSY_CYCLES > $out/pmc_summary.log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
tail -c 1500 $out/pmc_summary.log
