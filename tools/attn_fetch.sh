# FETCH_SIZE of the attention kernels at batch 8 and 16 (rocprofv3 --pmc, own pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 8 16; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/af$B -o af -- python3 $R/tools/attn_prof.py $B 24 8192 128 > $R/gpurun_out/af$B.log 2>&1
  python3 $R/tools/summarize_prof.py pmc $R/gpurun_out/af$B/af_counter_collection.csv $R/gpurun_out/af$B.txt
  grep -A2 "attn_fwd" $R/gpurun_out/af$B.txt | head -8
done
