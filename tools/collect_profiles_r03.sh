# Round-3 evidence (one gpurun call): bench line, rocprofv3 --kernel-trace --stats of the same command for configs[1], [2], [3],
# separate --pmc passes (FETCH_SIZE / WRITE_SIZE / MFMA counters), replay (fully resident window / slabs / classic), widened rows, config table.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp
timeout 400 python $R/bench.py --steps 20 --warmup 5 > $O/r03_bench.json 2>/dev/null
for cfg in 2 3 4; do
  rm -rf /tmp/p_trace
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -- python $R/bench.py --config $cfg --steps 20 --warmup 3 --no-cpu --no-cfg3 > /dev/null 2>&1
  sfx=""; [ $cfg != 2 ] && sfx="_c$cfg"
  find /tmp/p_trace -name "*kernel_stats.csv" -exec cp {} $O/r03_kernel_stats$sfx.csv \;
  find /tmp/p_trace -name "*kernel_trace.csv" -exec cp {} $O/r03_kernel_trace$sfx.csv \;
done
for cfg in 2 3 4; do
  sfx=""; [ $cfg != 2 ] && sfx="_c$cfg"
  rm -rf /tmp/p_fetch /tmp/p_write
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python $R/bench.py --config $cfg --steps 10 --warmup 2 --no-cpu --no-events --no-cfg3 > /dev/null 2>&1
  find /tmp/p_fetch -name "*counter_collection.csv" -exec cp {} $O/r03_pmc_fetch_size$sfx.csv \;
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- python $R/bench.py --config $cfg --steps 10 --warmup 2 --no-cpu --no-events --no-cfg3 > /dev/null 2>&1
  find /tmp/p_write -name "*counter_collection.csv" -exec cp {} $O/r03_pmc_write_size$sfx.csv \;
done
rm -rf /tmp/p_mfma
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/p_mfma -- python $R/bench.py --steps 10 --warmup 2 --no-cpu --no-events --no-cfg3 > /dev/null 2>&1
find /tmp/p_mfma -name "*counter_collection.csv" -exec cp {} $O/r03_pmc_mfma.csv \;
# 16-ring VGICP pair: HBM traffic of k_vgicp_lin (the line's roofline.traffic)
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_vg
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d /tmp/p_vg -- python $R/bench.py --vgicp --steps 10 --warmup 2 --no-cpu > /dev/null 2>&1
  find /tmp/p_vg -name "*counter_collection.csv" -exec cp {} $O/r03_vgicp16_pmc_$ctr.csv \;
done
cd $R
timeout 600 python bench.py --replay 600 --no-cpu > $O/r03_replay600_window_nocpu.json 2>/dev/null
timeout 900 python bench.py --replay 600 > $O/r03_replay600_fp64.json 2>/dev/null
timeout 900 python bench.py --replay 600 --precision 1 > $O/r03_replay600_fp32.json 2>/dev/null
timeout 600 python bench.py --replay 300 --slabs --no-cpu > $O/r03_replay300_slabs.json 2>/dev/null
timeout 600 python bench.py --replay 300 --classic --no-cpu > $O/r03_replay300_classic.json 2>/dev/null
(cd /tmp; rm -rf /tmp/p_rp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_rp -- python $R/bench.py --replay 200 --no-cpu > /dev/null 2>&1; find /tmp/p_rp -name "*kernel_stats.csv" -exec cp {} $O/r03_replay_kernel_stats.csv \;)
timeout 300 python bench.py --vgicp > $O/r03_vgicp16.json 2>/dev/null
timeout 300 python bench.py --vgicp --vgicp-rings 64 --vgicp-az 2048 > $O/r03_vgicp64.json 2>/dev/null
timeout 300 python bench.py --mapreg > $O/r03_mapreg.json 2>/dev/null
timeout 300 python bench.py --preint > $O/r03_preint.json 2>/dev/null
timeout 900 python tools/run_configs.py > $O/r03_configs.txt 2>&1
for f in $O/r03_kernel_trace*.csv; do python $R/profiles/summarize.py $f > ${f%.csv}_summary.txt 2>&1; done
python $R/profiles/summarize.py $O/r03_kernel_trace.csv $O/r03_pmc_fetch_size.csv $O/r03_pmc_write_size.csv $O/r03_pmc_mfma.csv > $O/r03_summary.txt 2>&1
python $R/profiles/summarize.py $O/r03_kernel_trace_c4.csv $O/r03_pmc_fetch_size_c4.csv $O/r03_pmc_write_size_c4.csv > $O/r03_summary_c4.txt 2>&1
python $R/profiles/summarize.py $O/r03_kernel_trace_c3.csv $O/r03_pmc_fetch_size_c3.csv $O/r03_pmc_write_size_c3.csv > $O/r03_summary_c3.txt 2>&1
ls -la $O | tail -40
