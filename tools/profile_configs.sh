set -e
# usage (on the GPU box): bash tools/profile_configs.sh [tag]  -- rocprofv3 kernel statistics of the other BASELINE configs
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {  # name, bench args...
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$name -o $name -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_bench_$name.json 2> $R/gpurun_out/bench_$name.err
  python3 $R/tools/summarize_prof.py stats $R/gpurun_out/prof_$name/${name}_kernel_stats.csv $R/gpurun_out/${TAG}_${name}_kernel_stats.txt
  echo $name-done
}
run interm10b_b1_recompute --model interm_10b --batch 1 --recompute --steps 3 --warmup 1
run interm117m_32x64_b8 --model interm_117m --grid 32x64 --batch 8 --steps 20 --warmup 5
run interm1b_daymet_96x192_b4 --daymet --grid 96x192 --batch 4 --steps 6 --warmup 2
