cd $GRAFT_REPO_ROOT; T=mvil-fusion_amd/csrc/libvilsolve_tuning.so
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_persist.py tests/test_gpu_edge.py -x -q 2>&1 | tail -3
for rep in 1 2; do for cfg in 2 3 4; do CFG=$cfg VIL_LIB=$T timeout 200 python tools/ab_step.py 2>&1 | grep -v amdgpu.ids; done; done
MODE=0 CFG=2 timeout 120 python tools/probe_timeline.py 2>&1 | grep -v amdgpu.ids | head -6) 2>&1 | tee gpurun_out/exp_wv.log
