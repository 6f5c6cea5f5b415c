export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp
timeout 300 python $R/bench.py --steps 40 --warmup 5 > $O/r01_bench.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -- python $R/bench.py --steps 40 --warmup 5 --no-cpu > /dev/null 2>&1
find /tmp/p_trace -name "*kernel_stats.csv" -exec cp {} $O/r01_kernel_stats.csv \;
find /tmp/p_trace -name "*kernel_trace.csv" -exec cp {} $O/r01_kernel_trace.csv \;
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python $R/bench.py --steps 10 --warmup 2 --no-cpu --no-events > /dev/null 2>&1
find /tmp/p_fetch -name "*counter_collection.csv" -exec cp {} $O/r01_pmc_fetch_size.csv \;
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- python $R/bench.py --steps 10 --warmup 2 --no-cpu --no-events > /dev/null 2>&1
find /tmp/p_write -name "*counter_collection.csv" -exec cp {} $O/r01_pmc_write_size.csv \;
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/p_mfma -- python $R/bench.py --steps 10 --warmup 2 --no-cpu --no-events > /dev/null 2>&1
find /tmp/p_mfma -name "*counter_collection.csv" -exec cp {} $O/r01_pmc_mfma.csv \;
cd $R
timeout 600 python bench.py --replay 600 > $O/r01_replay600_fp64.json 2>/dev/null
timeout 600 python bench.py --replay 600 --precision 1 > $O/r01_replay600_fp32.json 2>/dev/null
timeout 900 python tools/run_configs.py > $O/r01_configs.txt 2>&1
ls -la $O | tail -15
