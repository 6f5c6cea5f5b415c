# bench lines + rocprofv3 kernel stats of the widened rows; run on the GPU box from the repo root
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp
row() { # name, bench args...
  n=$1; shift
  timeout 400 python $R/bench.py "$@" > $O/r01_$n.json 2>/dev/null
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$n -- python $R/bench.py "$@" --no-cpu > /dev/null 2>&1
  find /tmp/p_$n -name "*kernel_stats.csv" -exec cp {} $O/r01_${n}_kernel_stats.csv \;
}
row mapreg --mapreg --steps 200 --warmup 5
row vgicp16 --vgicp --steps 200 --warmup 5
timeout 400 python $R/bench.py --vgicp --vgicp-rings 64 --vgicp-az 2048 --steps 200 --warmup 5 > $O/r01_vgicp64.json 2>/dev/null
row preint --preint --steps 300 --warmup 10
timeout 600 python $R/bench.py --mapreg --map-surf 256000 --map-corner 85000 --scan-surf 25600 --scan-corner 8500 --steps 50 --warmup 3 > $O/r01_mapreg_big.json 2>/dev/null
ls -la $O | tail -12
