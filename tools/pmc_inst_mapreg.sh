export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/q_inst -- python $R/bench.py --mapreg --steps 10 --warmup 2 --no-cpu > /dev/null 2>&1
python - <<'PY'
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/q_inst/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "k_map" in k or "pose_solve" in k:
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
