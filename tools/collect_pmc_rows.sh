# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes) of the widened rows' dominant kernels; GPU box, repo root
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp
pmc() { # name counter bench-args...
  n=$1; c=$2; shift; shift
  timeout 400 rocprofv3 --pmc $c --output-format csv -d /tmp/q_${n}_$c -- python $R/bench.py "$@" --no-cpu > /dev/null 2>&1
  find /tmp/q_${n}_$c -name "*counter_collection.csv" -exec cp {} $O/r01_${n}_pmc_$c.csv \;
}
for c in FETCH_SIZE WRITE_SIZE; do
  pmc vgicp64 $c --vgicp --vgicp-rings 64 --vgicp-az 2048 --steps 20 --warmup 2
  pmc mapreg $c --mapreg --steps 20 --warmup 2
done
python - <<'PY'
import csv, collections, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
for f in sorted(glob.glob(O + "/r01_*_pmc_*.csv")):
    acc = collections.defaultdict(list); name = None
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"])); name = r["Counter_Name"]
    print(os.path.basename(f))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:5]:
        print("   %-60s launches %5d  mean %10.1f  (%s, raw counter units)" % (k[:60], len(v), sum(v) / len(v), name))
PY
