set -e
# usage (on the GPU box): bash tools/profile_round.sh [tag]  -- rocprofv3 passes of the default bench configuration
# (kernel stats; FETCH_SIZE; WRITE_SIZE; MFMA-busy + clock; optional DRAM/MALL counters when the box lists them)
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
rocprofv3 -L > $R/gpurun_out/prof/counters_avail.txt 2>&1 || true
grep -i -E "dram|mall|EA0_RDREQ|EA0_WRREQ|HBM|TCC_EA" $R/gpurun_out/prof/counters_avail.txt | cut -c1-200 | sort -u | head -60 > $R/gpurun_out/${TAG}_counters_memside.txt || true
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o ${TAG} -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs > $R/gpurun_out/prof_bench.log 2>&1
echo stats-done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof -o ${TAG}_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $R/gpurun_out/prof_fetch.log 2>&1
echo fetch-done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof -o ${TAG}_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $R/gpurun_out/prof_write.log 2>&1
echo write-done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof -o ${TAG}_mfma -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $R/gpurun_out/prof_mfma.log 2>&1
echo mfma-done
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum --output-format csv -d $R/gpurun_out/prof -o ${TAG}_mall -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --mall-probe > $R/gpurun_out/prof_mall.log 2>&1 || echo "mall pass failed (see prof_mall.log)"
echo mall-done
cd $R
python tools/summarize_prof.py mall gpurun_out/prof/${TAG}_mall_counter_collection.csv gpurun_out/${TAG}_mall_latency.json > gpurun_out/${TAG}_mall_latency.txt 2>&1 || true
python tools/summarize_prof.py stats gpurun_out/prof/${TAG}_kernel_stats.csv gpurun_out/${TAG}_interm1b_b16_kernel_stats.txt
rm -f gpurun_out/${TAG}_interm1b_b16_pmc_traffic.txt
python tools/summarize_prof.py pmc gpurun_out/prof/${TAG}_fetch_counter_collection.csv gpurun_out/${TAG}_interm1b_b16_pmc_traffic.txt
python tools/summarize_prof.py pmc gpurun_out/prof/${TAG}_write_counter_collection.csv gpurun_out/${TAG}_interm1b_b16_pmc_traffic.txt
python tools/summarize_prof.py traffic gpurun_out/prof/${TAG}_fetch_counter_collection.csv gpurun_out/prof/${TAG}_write_counter_collection.csv gpurun_out/${TAG}_traffic.json 16
python tools/summarize_prof.py mfma gpurun_out/prof/${TAG}_mfma_counter_collection.csv gpurun_out/${TAG}_interm1b_b16_mfma_util.txt
rm -rf gpurun_out/prof/*.csv 2>/dev/null || true
ls gpurun_out | head -50
