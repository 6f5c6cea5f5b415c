# Round-6 evidence (one gpurun call): bench line, rocprofv3 --kernel-trace --stats of the same command for configs[1], [2], [3], separate --pmc passes
# (FETCH_SIZE / WRITE_SIZE / MFMA counters, each configuration), replay, widened rows, config table.  Collect AFTER the last code change.
# Every pass of a window runs with --no-tracker: the tracker legs' configs[1]-shaped windows launch the same kernel names, and their launches
# polluted the K = 20 / configs[2] averages of the round-4 kernel-trace summaries.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd /tmp
for cfg in 2 3 4; do
  sfx=""; [ $cfg != 2 ] && sfx="_c$cfg"
  rm -rf /tmp/p_trace
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -- python $R/bench.py --config $cfg --steps 20 --warmup 3 --no-cpu --no-cfg3 --no-tracker > /dev/null 2>&1
  find /tmp/p_trace -name "*kernel_stats.csv" -exec cp {} $O/r06_kernel_stats$sfx.csv \;
  find /tmp/p_trace -name "*kernel_trace.csv" -exec cp {} $O/r06_kernel_trace$sfx.csv \;
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p_pmc
    timeout 300 rocprofv3 --pmc $ctr --output-format csv -d /tmp/p_pmc -- python $R/bench.py --config $cfg --steps 10 --warmup 2 --no-cpu --no-events --no-cfg3 --no-tracker > /dev/null 2>&1
    lc=$(echo $ctr | tr A-Z a-z)
    find /tmp/p_pmc -name "*counter_collection.csv" -exec cp {} $O/r06_pmc_${lc}$sfx.csv \;
  done
  rm -rf /tmp/p_mfma
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/p_mfma -- python $R/bench.py --config $cfg --steps 10 --warmup 2 --no-cpu --no-events --no-cfg3 --no-tracker > /dev/null 2>&1
  find /tmp/p_mfma -name "*counter_collection.csv" -exec cp {} $O/r06_pmc_mfma$sfx.csv \;
  python $R/profiles/summarize.py $O/r06_kernel_trace$sfx.csv $O/r06_pmc_fetch_size$sfx.csv $O/r06_pmc_write_size$sfx.csv $O/r06_pmc_mfma$sfx.csv > $O/r06_summary$sfx.txt 2>&1
done
cd $R
if [ "$1" != "quick" ]; then
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_vg
  (cd /tmp; timeout 300 rocprofv3 --pmc $ctr --output-format csv -d /tmp/p_vg -- python $R/bench.py --vgicp --steps 10 --warmup 2 --no-cpu > /dev/null 2>&1)
  find /tmp/p_vg -name "*counter_collection.csv" -exec cp {} $O/r06_vgicp16_pmc_$ctr.csv \;
done
timeout 600 python bench.py --replay 600 --no-cpu > $O/r06_replay600_window_nocpu.json 2>/dev/null
timeout 900 python bench.py --replay 600 > $O/r06_replay600_fp64.json 2>/dev/null
timeout 900 python bench.py --replay 600 --precision 1 > $O/r06_replay600_fp32.json 2>/dev/null
timeout 600 python bench.py --replay 300 --classic --no-cpu > $O/r06_replay300_classic.json 2>/dev/null
(cd /tmp; rm -rf /tmp/p_rp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_rp -- python $R/bench.py --replay 200 --no-cpu > /dev/null 2>&1; find /tmp/p_rp -name "*kernel_stats.csv" -exec cp {} $O/r06_replay_kernel_stats.csv \;)
timeout 300 python bench.py --vgicp > $O/r06_vgicp16.json 2>/dev/null
timeout 300 python bench.py --vgicp --vgicp-rings 64 --vgicp-az 2048 > $O/r06_vgicp64.json 2>/dev/null
timeout 300 python bench.py --mapreg > $O/r06_mapreg.json 2>/dev/null
timeout 300 python bench.py --preint > $O/r06_preint.json 2>/dev/null
fi
# scale sweep: the bench leg + counter passes per size (one launch per iteration everywhere: the counters are then per iteration at every size)
timeout 600 python bench.py --scale-sweep --steps 20 > $O/r06_scale_sweep.json 2>/dev/null
mkdir -p $O/scale
for sc in 1 4 16 64; do
  timeout 300 python bench.py --scale-sweep --scales $sc --launch-mode 4 --steps 12 > $O/scale/scale_$sc.json 2>/dev/null
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p_sc; (cd /tmp; timeout 300 rocprofv3 --pmc $ctr --output-format csv -d /tmp/p_sc -- python $R/bench.py --scale-sweep --scales $sc --launch-mode 4 --steps 8 --no-events > /dev/null 2>&1)
    lc=$(echo $ctr | tr A-Z a-z | sed s/_size//); find /tmp/p_sc -name "*counter_collection.csv" -exec cp {} $O/scale/scale_${sc}_$lc.csv \;
  done
  rm -rf /tmp/p_sc; (cd /tmp; timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d /tmp/p_sc -- python $R/bench.py --scale-sweep --scales $sc --launch-mode 4 --steps 8 --no-events > /dev/null 2>&1)
  find /tmp/p_sc -name "*counter_collection.csv" -exec cp {} $O/scale/scale_${sc}_mfma.csv \;
done
python profiles/scale_summary.py $O/scale > $O/r06_scale_sweep.txt 2>&1
rm -f $O/scale/*.csv
timeout 300 python bench.py --batch --steps 30 > $O/r06_concurrent_windows.json 2>/dev/null
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --batch --steps 30 > $O/r06_concurrent_windows_8queues.json 2>/dev/null
(for c in 2; do echo "== configs[1], the library's choice (persistent solve)"; MODE=0 CFG=$c timeout 120 python tools/probe_timeline.py; echo "== configs[1], one launch per iteration (mode 4)"; MODE=4 CFG=$c timeout 120 python tools/probe_timeline.py; done; echo "== configs[2] (K = 10, 4000 landmarks)"; MODE=0 CFG=3 timeout 120 python tools/probe_timeline.py; echo "== configs[3] (K = 20)"; MODE=0 CFG=4 timeout 120 python tools/probe_timeline.py) 2>&1 | grep -v amdgpu.ids > $O/r06_timeline.txt
timeout 200 python tools/probe_marg.py 2>&1 | grep -v amdgpu.ids > $O/r06_marg_phases.txt
timeout 100 python tools/probe_marg_first.py 2>&1 | grep -v amdgpu.ids >> $O/r06_marg_phases.txt
timeout 900 python tools/run_configs.py > $O/r06_configs.txt 2>&1
for c in 2 3 4; do CFG=$c timeout 120 python tools/probe_phases.py; done > $O/r06_phases.txt 2>&1
timeout 200 python tools/probe_tracker.py > $O/r06_tracker_breakdown.txt 2>&1
# the bench line LAST: it reads the --pmc CSVs of this very collection from profiles/ (copied there for the run)
cp $O/r06_pmc_*.csv $R/profiles/ 2>/dev/null
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $O/r06_bench.json 2>$O/r06_bench.err
ls -la $O | tail -50; cat $O/r06_summary_c4.txt; cat $O/r06_configs.txt
