export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_m -- python $R/bench.py --mapreg --steps 200 --warmup 5 --no-cpu > /dev/null 2>&1
find /tmp/p_m -name "*kernel_stats.csv" -exec cat {} \; | cut -c1-60,200-400 | head -12
find /tmp/p_m -name "*kernel_stats.csv" -exec awk -F'",' '{print substr($1,1,50), $2, $4}' {} \; | head -12
