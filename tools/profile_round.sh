set -e
# usage (on the GPU box): bash tools/profile_round.sh   -- three rocprofv3 passes of the default bench configuration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r01 -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
echo stats-done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof -o r01_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_fetch.log 2>&1
echo fetch-done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof -o r01_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_write.log 2>&1
echo write-done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof -o r01_mfma -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_mfma.log 2>&1
echo mfma-done
cd $R
python tools/summarize_prof.py stats gpurun_out/prof/r01_kernel_stats.csv gpurun_out/r01_interm1b_b16_kernel_stats.txt
python tools/summarize_prof.py pmc gpurun_out/prof/r01_fetch_counter_collection.csv gpurun_out/r01_interm1b_b16_pmc_traffic.txt
python tools/summarize_prof.py pmc gpurun_out/prof/r01_write_counter_collection.csv gpurun_out/r01_interm1b_b16_pmc_traffic.txt
python tools/summarize_prof.py traffic gpurun_out/prof/r01_fetch_counter_collection.csv gpurun_out/prof/r01_write_counter_collection.csv gpurun_out/r01_traffic.json 16
ls gpurun_out/prof
python tools/summarize_prof.py mfma gpurun_out/prof/r01_mfma_counter_collection.csv gpurun_out/r01_interm1b_b16_mfma_util.txt
