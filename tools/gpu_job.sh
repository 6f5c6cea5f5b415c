#!/bin/bash
# One GPU-box job: what to run is selected by the arguments (tests | probe | bench | prof ...); outputs go to gpurun_out/.
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
for what in "$@"; do
  case "$what" in
    tests) timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/tests.log; tail -5 gpurun_out/tests.log ;;
    tests_all) timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/tests.log; tail -15 gpurun_out/tests.log ;;
    probe) for cfg in 2 4; do CFG=$cfg VIL_LIB=tools/_exp/libvilsolve_stamps.so timeout 300 python tools/probe_step.py > gpurun_out/probe_c$cfg.log 2>&1; cat gpurun_out/probe_c$cfg.log; done ;;
    probe_plain) for cfg in 2 3 4; do CFG=$cfg timeout 300 python tools/probe_step.py 2>&1 | head -1 | tee gpurun_out/probe_plain_c$cfg.log; done ;;
    bench) timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json ;;
    *) echo "unknown job $what" ;;
  esac
done
