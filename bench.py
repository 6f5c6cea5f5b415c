#!/usr/bin/env python3
"""bench.py -- sliding-window solve iterations/s on MI355X (BASELINE.json metric).

A "step" is one full window solve (Estimator::optimization()'s ceres::Solve replacement) of the
BASELINE.json configs[1] workload -- 10 keyframes, 1k landmarks, 30k LiDAR edge/plane points -- with
the factor tables and the state already resident in HBM; value = trust-region iterations executed /
wall time (whole job, all ranks).  One JSON line on rank 0 (contract in the task statement), carrying
  roofline     : the factor-sweep kernel (dominant kernel), HIP-event timed on the library's stream
  cpu_baseline : the CPU restatement of the reference algorithm (oracle/, kind "port"), single thread
                 like the reference's ceres::Solve (no num_threads set, estimator.cpp:1400-1411)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

# the reference's solver limits on the replay configuration: config/mynteye_leishen_indoor.yaml:76-77 -> estimator.cpp:1404 (max_num_iterations), :1411 (max_solver_time_in_seconds)
REF_MAX_ITERATIONS, REF_MAX_TIME_S = 30, 0.05

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


class VilProfile(C.Structure):
    _fields_ = [("sweep_launches", C.c_int64), ("sweep_ms", C.c_double), ("step_launches", C.c_int64), ("step_ms", C.c_double), ("reduce_ms", C.c_double), ("collective_ms", C.c_double)]


def emit(out):
    """The ONE JSON line, and the last thing on stdout: native libraries (RCCL's version banner under a communicator) write through C stdio, which is block-buffered
    on a pipe and would otherwise be flushed at exit, AFTER Python's line."""
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


def algorithmic_bytes(w):
    """SURVEY.md 8(d), fused (read-only) variant of the sweep: bytes one launch must read."""
    n = w.prior.n
    return (132 * len(w.vis_i) + 60 * len(w.plane_pose) + 76 * len(w.edge_pose) + (2296 + 8) * len(w.imu_i)
            + 8 * (n * n + n) + 8 * (16 * w.K + 8))


def pmc_traffic_bytes(files=("r03_pmc_fetch_size.csv", "r03_pmc_write_size.csv"), kernel="k_sweep"):
    """HBM bytes per live launch of `kernel` from the committed rocprofv3 --pmc passes of THIS command (profiles/):
    (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md (HBM section),
    WRITE_SIZE uncalibrated.  None if the CSVs are missing."""
    import csv
    tot = 0.0
    for fn, mult in ((files[0], 2.0), (files[1], 1.0)):
        path = os.path.join(ROOT, "profiles", fn)
        if not os.path.exists(path):
            return None
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"]]
        if not v:
            return None
        live = [x for x in v if x > 0.25 * max(v)]
        tot += mult * 1024.0 * sum(live) / len(live)
    return tot


def pmc_mfma(file, kernel="k_iter"):
    """Counter-based matrix-core figures of `kernel` from the committed rocprofv3 --pmc pass (SQ_INSTS_VALU_MFMA_MOPS_F64, SQ_VALU_MFMA_BUSY_CYCLES,
    SQ_BUSY_CYCLES in ONE pass; profiles/): per live launch the MFMA ops (x 512 = fp64 flop), the cycles the matrix pipes were busy (summed over the
    SIMDs that ran the kernel) and the launch duration.  None if the CSV is missing."""
    import collections, csv
    path = os.path.join(ROOT, "profiles", file)
    if not os.path.exists(path):
        return None
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    if not acc.get("SQ_INSTS_VALU_MFMA_MOPS_F64") or not acc.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        return None
    d = [x[1] for x in acc["SQ_VALU_MFMA_BUSY_CYCLES"]]
    live = [i for i, x in enumerate(d) if x > 0.5 * max(d)]
    mean = lambda name: sum(acc[name][i][0] for i in live) / len(live)
    mops, busy, us = mean("SQ_INSTS_VALU_MFMA_MOPS_F64"), mean("SQ_VALU_MFMA_BUSY_CYCLES"), sum(d[i] for i in live) / len(live) / 1e3
    sq_busy = mean("SQ_BUSY_CYCLES") if acc.get("SQ_BUSY_CYCLES") else None
    return {"mops_per_launch": mops, "flop_per_launch": 512.0 * mops, "mfma_busy_cycles_per_launch": busy, "sq_busy_cycles_per_launch": sq_busy, "launch_us_under_pmc": us, "live_launches": len(live)}


_NATIVE = {}


def native_oracle():
    """BASELINE.md section 2 asks for `-O3 -march=native`: the shipped oracle/liboracle.so is built in the build container with -march=x86-64-v3 (the
    box that runs the bench is another machine), so the CPU leg rebuilds the oracle HERE with -march=native (~25 s, once) and times that; falls back to
    the shipped library -- and says so -- if the host has no compiler."""
    if "so" in _NATIVE:
        return _NATIVE["so"], _NATIVE["note"]
    import subprocess, tempfile
    srcs = ["oracle_factors.cpp", "oracle_solver.cpp", "oracle_marg.cpp", "oracle_api.cpp", "oracle_vgicp.cpp", "oracle_map.cpp"]
    out = os.path.join(tempfile.gettempdir(), "liboracle_native_%d.so" % os.getpid())
    cmd = ["g++", "-O3", "-march=native", "-std=c++17", "-fPIC", "-pthread", "-shared", "-o", out] + [os.path.join(ROOT, "oracle", f) for f in srcs]
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240)
        _NATIVE["so"], _NATIVE["note"] = out, "oracle rebuilt on this host with g++ -O3 -march=native (BASELINE.md section 2)"
    except Exception:
        _NATIVE["so"], _NATIVE["note"] = os.path.join(ROOT, "oracle", "liboracle.so"), "shipped oracle/liboracle.so, built with -O3 -march=x86-64-v3 (no compiler on this host for -march=native)"
    return _NATIVE["so"], _NATIVE["note"]


def cpu_baseline(w, opts, budget_s=10.0):
    """Oracle (CPU restatement, kind 'port') timed on this host: repeated full solves of the same window.
    SURVEY 8(d): warm-up 3, then >= 20 solves; single thread like the reference's ceres::Solve (num_threads is never set,
    estimator.cpp:1400-1411).  A second, shorter leg times the all-cores variant (factor sweep threaded) so that the
    GPU/CPU ratio is not flattered by the single-thread choice."""
    import numpy as np
    from mvil_fusion_amd import lib
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    so_path, build_note = native_oracle()
    orc = lib.Backend(C.CDLL(so_path), "orc_")
    st0 = w.state_copy()

    def leg(threads, budget, min_solves):
        orc.lib.orc_set_threads(threads)
        for _ in range(3):
            w.set_state(st0); orc.solve(w, opts)
        its, ts, t0 = 0, [], time.perf_counter()
        while True:
            w.set_state(st0)
            t1 = time.perf_counter()
            s = orc.solve(w, opts)
            ts.append(time.perf_counter() - t1); its += s.iterations
            if (time.perf_counter() - t0 >= budget and len(ts) >= min_solves) or len(ts) >= 400:
                break
        orc.lib.orc_set_threads(1)
        return its, ts
    its, ts = leg(1, budget_s, 20)
    ncpu = os.cpu_count() or 1
    try:
        ncpu = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # all-cores variant: the factor sweep is ~3 ms of work per linearisation, so more threads is not better -- time a
    # few team sizes and report the best one
    best = None
    for nthr in sorted({min(ncpu, k) for k in (4, 8, 16, 32)}):
        its_k, ts_k = leg(nthr, 1.5, 8)
        rate = its_k / sum(ts_k)
        if best is None or rate > best[0]:
            best = (rate, nthr, len(ts_k))
    mt_rate, nthr, mt_n = best
    w.set_state(st0)
    return {"value": its / sum(ts), "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": "%d full solves (%d trust-region iterations) of the same configs[1] window after 3 warm-up solves, %.1f s; median solve %.2f ms"
                      % (len(ts), its, sum(ts), 1e3 * float(np.median(ts))),
            "all_cores": {"value": mt_rate, "unit": "iterations/s", "cores": nthr,
                          "sample": "%d solves, factor sweep on %d threads = best of the team sizes 4/8/16/32 (Schur complement / Cholesky / dogleg stay serial)" % (mt_n, nthr)},
            "note": "CPU restatement of the reference algorithm (Ceres unavailable); %s; host has %d usable cores" % (build_note, ncpu)}


PHASE_NAMES = ["first workgroup started", "last visual / LiDAR / ICP-LPS role done", "last IMU role done", "chain workgroup saw the IMU / prior records", "chain: W^T complete", "last gather workgroup done",
               "last gather workgroup saw the visual flags", "master started", "master saw the gather's flags", "master saw the W W^T tiles", "dense factorisation done", "x_p published", "master done",
               "last tile workgroup done", "chain: its part of S' gathered", "prior role done", "IMU role 0 entered", "IMU role 0: inputs staged", "IMU role 0: raw blocks done", "IMU role 0: whitened", "IMU role 0: record stores issued", "master: chain back-substituted", "master: step vectors and the helpers' sums in", "master: candidate formed"]


def phases_obj(be):
    """Where a one-launch iteration spends its time: the launch's own 100 MHz wall-clock stamps, averaged over the instrumented pass (vil_profile_phases)."""
    avg = (C.c_double * 32)(); n = C.c_int64(0)
    if be.lib.vil_profile_phases(be.ctx, avg, C.byref(n), 1) != 0 or n.value == 0:
        return None
    order = sorted([q for q in range(1, 24) if avg[q] > 0], key=lambda q: avg[q])
    return {"unit": "us after the launch's first workgroup started", "launches_averaged": int(n.value), "source": "s_memrealtime stamps of the roles (csrc/vil_dev.hpp: prof_stamp); XCD clocks agree to ~1-2 us",
            "stamps": {PHASE_NAMES[q]: round(float(avg[q]), 2) for q in order}}


def roofline_obj(w, prof, measured_on, pmc_files, pmc_note, mfma_file=None, one_launch=True, phases=None, two_launch=None, persistent=False):
    """`roofline` of the dominant kernel.  One-launch iteration (k_iter): the kernel IS the iteration -- factor sweep, gather, chain elimination and trust-region step as roles of
    one grid -- so `achieved` = the sweep's algorithmic bytes over the WHOLE launch's duration (HIP events), which prices a latency-bound launch against the HBM roof; the sweep PHASE of
    the launch (its own clock stamps) and the two-launch structure's k_sweep are reported beside it.  Two / three launches per iteration: the sweep kernel, as in earlier rounds."""
    ab = algorithmic_bytes(w)
    sweep_us = 1e3 * prof.sweep_ms / prof.sweep_launches
    rest_us = 1e3 * prof.step_ms / max(1, prof.step_launches)
    ts = "5" if w.K > 12 else "2"      # (the visual role's accumulator tiles per wave: csrc/vil_sweep.hpp; the counter files may hold other windows' launches too)
    NP, NB = 6 * w.K + 7, 9 * w.K
    chol_flop = w.K * (9 ** 3 / 3.0 + 2.0 * 81 * (NP + 1 + 9)) + 1.0 * (NP + 1) ** 2 * NB + NP ** 3 / 3.0 + 2.0 * (NP * NP + NB * (NP + 9))
    units = 1.0
    if one_launch and persistent:
        # the persistent solve: ONE launch per solve runs every iteration (k_solve, csrc/vil_iter.hpp).  A launch processes `units` sweeps: achieved = units x the sweep's
        # algorithmic bytes over the launch's duration (HIP events around the launch).  Per iteration: the launch's own clock stamps.
        units = prof.sweep_launches / max(1, prof.step_launches)
        launch_us = 1e3 * (prof.step_ms + prof.sweep_ms) / max(1, prof.step_launches)
        rest_us = launch_us / units - sweep_us
    if one_launch:
        kname = ("k_solve<%s>" if persistent else "k_iter<%s>") % ts
        us = (sweep_us + rest_us) * units
        ab1, ab = ab, ab * units           # per sweep / per launch
        ach = ab / (us * 1e-6) / 1e9
        r = {"bound": "hbm", "kernel": kname + (" (ONE resident launch per SOLVE: the roles below loop over the trust-region iterations; per iteration: sweep roles | chain | gather duties | master + helpers | W W^T tiles)" if persistent else " (one launch per trust-region iteration: sweep roles | chain | gather | master + helpers | W W^T tiles)"), "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
             "traffic": pmc_traffic_bytes(pmc_files, kname), "traffic_source": pmc_note, "algorithmic_bytes_per_launch": ab, "avg_launch_us": us, "launches_timed": int(prof.step_launches if persistent else prof.sweep_launches),
             "sweeps_per_launch": units, "avg_iteration_us": us / units,
             "measured_on": measured_on, "variant": "fused sweep (no Jacobian materialisation): read-only bytes, SURVEY 8(d)",
             "note": "the launch is latency-bound (flag hand-offs between roles, then one master workgroup on dependent fp64 chains): the sweep's 2.4 MB are read in the first quarter of it -- sweep_phase prices that quarter, "
                     "critical_path what the remaining three quarters are made of",
             "sweep_phase": {"us": sweep_us, "achieved": ab1 / (sweep_us * 1e-6) / 1e9, "frac": ab1 / (sweep_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s",
                             "what": "first workgroup of the launch (persistent solve: of the iteration) started -> last sweep role's record out, from the launch's own clock stamps"},
             "critical_path": {"us_after_the_sweep_phase": rest_us, "per": "iteration", "bound": "latency: gather of the visual records beside the two-sided 9x9 chain of %d blocks, then a %d-pivot dense Cholesky and the back substitutions on one workgroup (dependent fp64 chains)" % (w.K, NP),
                               "dense_flop_per_launch": chol_flop, "achieved_gflops": chol_flop / (rest_us * 1e-6) / 1e9, "peak_tflops_fp64_matrix": 78.6, "frac": chol_flop / (rest_us * 1e-6) / 78.6e12}}
        r["critical_path_kernel"] = {"kernel": kname, "avg_launch_us": us, "frac": ach / HBM_PEAK_GBS, "see": "roofline.critical_path, roofline.phases"}
        if phases:
            r["phases"] = phases
        if two_launch:
            r["two_launch_structure"] = two_launch
    else:
        kname = "k_sweep<%s>" % ts
        us = sweep_us
        ach = ab / (us * 1e-6) / 1e9
        r = {"bound": "hbm", "kernel": kname, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes(pmc_files, kname), "traffic_source": pmc_note,
             "algorithmic_bytes_per_launch": ab, "avg_launch_us": us, "launches_timed": int(prof.sweep_launches), "gather_plus_step_avg_us": rest_us,
             "measured_on": measured_on, "variant": "fused sweep (no Jacobian materialisation): read-only bytes, SURVEY 8(d)"}
        r["critical_path_kernel"] = {"kernel": "k_step (gather workgroups | chain workgroup | W W^T tile workgroups | master + helpers)", "avg_launch_us": rest_us,
                                     "bound": "latency: gather of the visual records beside the two-sided 9x9 chain of %d blocks, then a %d-pivot dense Cholesky and the back substitutions on one workgroup (dependent fp64 chains)" % (w.K, NP),
                                     "dense_flop_per_launch": chol_flop, "achieved_gflops": chol_flop / (rest_us * 1e-6) / 1e9, "peak_tflops_fp64_matrix": 78.6, "frac": chol_flop / (rest_us * 1e-6) / 78.6e12}
    if mfma_file:
        # the Schur contraction of the landmarks (sum_f Jc^T Jc - sum_l invp e e^T per visual workgroup) runs on the fp64 matrix cores inside k_sweep:
        # utilisation from the COUNTERS of the committed --pmc pass, against the chip's fp64-matrix peak over the sweep's own duration
        m = pmc_mfma(mfma_file, kname)
        if m:
            us_k = m["launch_us_under_pmc"]
            r["mfma"] = {"kernel": kname + " (visual role: Schur contraction of the landmark block)", "bound": "mfma", "unit": "TFLOP/s", "peak": 78.6,
                         "achieved": m["flop_per_launch"] / (us_k * 1e-6) / 1e12, "frac": m["flop_per_launch"] / (us_k * 1e-6) / 78.6e12,
                         "v_mfma_f64_16x16x4_per_launch": m["mops_per_launch"] / 4.0, "flop_per_launch": m["flop_per_launch"],
                         "mfma_busy_cycles_per_launch": m["mfma_busy_cycles_per_launch"], "sq_busy_cycles_per_launch": m["sq_busy_cycles_per_launch"],
                         "launch_us": us_k, "source": "profiles/%s (rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES, one pass; MOPS x 512 flop)" % mfma_file,
                         "note": "a dense window-local contraction spends ~6x the multiply-adds of the block-sparse form; the sweep phase is bound by its longest visual workgroup (evaluation, sums and record around ~4 us of matrix-core work), not by the matrix pipes; "
                                 "counted over the whole launch of the counter pass (one-launch iteration: the dense factorisation's and the W W^T tiles' few matrix-core instructions are in the count)"}
    return r


def tracker_leg(lib, abi, device, n_images=48, warm=6):
    """PCIe-inclusive iterations/s as a tracker gets them: a sequence of configs[1]-shaped windows (K = 10, ~1000 landmarks, 30 k LiDAR
    points: the synthetic replay) driven through the fully resident window (vil_win_*).  Timed per image: the new frame going up
    (vil_win_push_frame: IMU samples, observations, 3000 LiDAR points), the slide (vil_win_drop_frame) and vil_win_solve (small tables + state
    up, solved state back).  The marginalisation between two images is not part of this metric: it is enqueued and waited for outside the clock."""
    import torch
    from mvil_fusion_amd import replay
    be = lib.open_vilsolve(device=device)
    rp = replay.Replay(K=10, n_frames=n_images + warm + 14, L=1000, n_plane=24000, n_edge=6000, seed=20240605, max_iterations=REF_MAX_ITERATIONS, max_time_s=REF_MAX_TIME_S)
    K = rp.K
    be.set_gauge_fix(True); be.win_open(**rp.win_open_args())
    for k in range(K):
        be.win_push_frame(rp.win_frame(k))
    its, el, up_bytes, capped = 0, 0.0, 0, 0
    for step in range(n_images + warm):
        flag = rp.margin_flag()
        w = rp.win_window()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); sg = be.win_solve(w, rp.opts); t1 = time.perf_counter()
        be.win_marginalize(flag, w._icp_marg, w._lps_marg, rp.opts)
        torch.cuda.synchronize()
        t2 = time.perf_counter(); be.win_drop_frame(flag); t3 = time.perf_counter()
        if not rp.absorb(w, None, flag):
            break
        fr = rp.win_frame(K - 1)
        t4 = time.perf_counter(); be.win_push_frame(fr); t5 = time.perf_counter()
        if step >= warm:
            its += sg.iterations; el += (t1 - t0) + (t3 - t2) + (t5 - t4)
            capped += sg.termination in (abi.TERM_NAMES.index("max_iterations"), abi.TERM_NAMES.index("max_time"))
            up_bytes += 8 * (7 * len(fr["dt"]) + 12 + 8 * len(fr["obs"]) + 7 * len(fr["plane"]) + 9 * len(fr["edge"])) + 4 * len(fr["obs_track"]) + 8 * (16 * K + 8 + w.L) + 13 * w.L
    be.close()
    n = max(1, step + 1 - warm)
    return {"value": its / el, "unit": "iterations/s", "ms_per_image": 1e3 * el / n, "images": n, "iterations_per_image": its / n,
            "share_of_images_ended_by_a_cap": capped / n, "caps": {"max_iterations": REF_MAX_ITERATIONS, "max_time_s": REF_MAX_TIME_S},
            "host_to_device_bytes_per_image": int(up_bytes / n),
            "what": "fully resident window (vil_win_*) on a tracker-driven sequence of configs[1]-shaped windows: per image the new frame (IMU samples, observations, 3000 LiDAR points) and "
                    "the window's small tables + state go up, the solved state comes back; timed = vil_win_push_frame + vil_win_drop_frame + vil_win_solve; solver limits = the reference's yaml (max_num_iterations 30, max_solver_time 0.05 s)"}


def replay_mode(args, be, abi, lib):
    """BASELINE.json configs[4] on a SYNTHETIC replay (3indoor.bag is not available offline): the per-image chain
    solve -> gauge fix -> marginalise -> slide is driven by the HIP library; on every frame the CPU restatement gets the
    SAME input window (cpu_baseline leg), so the two latencies and the state difference are per-frame comparable."""
    import numpy as np
    import torch
    from mvil_fusion_amd import replay
    from mvil_fusion_amd.abi import Window
    cap_it, cap_t = (8, 0.0) if args.cap8 else (REF_MAX_ITERATIONS, REF_MAX_TIME_S)      # --cap8: the short-cap variant of rounds 1-4 (NOT the reference's configuration)
    rp = replay.Replay(K=10, n_frames=args.replay + 10, L=1000, n_plane=24000, n_edge=6000, seed=20240605, max_iterations=cap_it, max_time_s=cap_t)
    rp.opts.precision = args.precision
    orc = None
    if not args.no_cpu:
        so_path = os.path.join(ROOT, "oracle", "liboracle.so")
        orc = lib.Backend(C.CDLL(so_path), "orc_")
    # CPU leg: the same iteration cap WITHOUT the wall-clock cap, so that it reproduces the device's solve (state difference) -- how many of its solves the
    # reference's 50 ms cap would have cut short on this host is reported beside it (cpu_solves_over_the_time_cap)
    opts_cpu = abi.default_options(max_iterations=cap_it)
    terms = []
    g_solve, g_marg, g_slide, g_period, c_solve, c_marg, dpos, its, Ls, nvis = [], [], [], [], [], [], [], [], [], []
    mode = "classic" if args.classic else ("slabs" if args.slabs else "window")
    K = rp.K
    if mode == "slabs":                            # round-2 residency: LiDAR frame slabs on the device, gauge fix on the device, marginalisation of the
        be.set_gauge_fix(True); be.lidar_reset()   # resident window; visual / IMU tables, the state and the prior still travel every image
        for k in range(K):
            be.lidar_push(rp.lidar[k][0], rp.lidar[k][1])
    if mode == "window":                           # the fully resident window (vil_win_*): per image the new frame + the small tables go up, the state comes back
        be.set_gauge_fix(True); be.win_open(**rp.win_open_args())
        for k in range(K):
            be.win_push_frame(rp.win_frame(k))
    t_prev = None
    for step in range(args.replay):
        flag = rp.margin_flag()
        wo = None
        if mode == "window":
            w = rp.win_window()
            if orc is not None:
                rp.prior = be.win_prior_download(K).to_prior() or rp.prior           # CPU leg only: the prior never leaves the device on the product path
                wo = Window.from_dict(rp.window().to_dict())
            p0 = w.pose[0].copy()
            t0 = time.perf_counter(); sg = be.win_solve(w, rp.opts); t1 = time.perf_counter()
            be.win_marginalize(flag, w._icp_marg, w._lps_marg, rp.opts)
            if step & 1:                               # every other image waits for the marginalisation kernels inside the timed region: the
                torch.cuda.synchronize()               # serialised figure (no overlap with the host's bookkeeping) next to the overlapped one
            t2 = time.perf_counter()
            pg = None
        elif mode == "slabs":
            w = rp.window(with_lidar=False)
            if orc is not None:
                wo = Window.from_dict(rp.window().to_dict())
            p0 = w.pose[0].copy()
            t0 = time.perf_counter(); sg = be.solve(w, rp.opts); t1 = time.perf_counter()
            pg = be.marginalize_resident(w, flag, w._icp_marg, w._lps_marg, rp.opts); t2 = time.perf_counter()
        else:
            w = rp.window()
            wo = Window.from_dict(w.to_dict()) if orc is not None else None
            p0 = w.pose[0].copy()
            t0 = time.perf_counter(); sg = be.solve(w, rp.opts); be.gauge_fix(p0, w); t1 = time.perf_counter()
            pg = be.marginalize(w, flag, w._icp_marg, w._lps_marg, rp.opts); t2 = time.perf_counter()
        g_solve.append(1e3 * (t1 - t0)); g_marg.append(1e3 * (t2 - t1)); its.append(sg.iterations); Ls.append(w.L); nvis.append(len(w.vis_i)); terms.append(int(sg.termination))
        if orc is not None:
            t0 = time.perf_counter(); orc.solve(wo, opts_cpu); orc.gauge_fix(p0, wo); t1 = time.perf_counter()
            orc.marginalize(wo, flag, w._icp_marg, w._lps_marg, opts_cpu); t2 = time.perf_counter()
            c_solve.append(1e3 * (t1 - t0)); c_marg.append(1e3 * (t2 - t1))
            dpos.append(float(np.abs(w.pose[:, :3] - wo.pose[:, :3]).max()))
        t3 = time.perf_counter()
        if mode == "slabs":
            be.lidar_drop(0 if flag == abi.MARGIN_OLD else K - 2)
        if mode == "window":
            be.win_drop_frame(flag)
        t3b = time.perf_counter()
        if not rp.absorb(w, pg, flag):
            break
        t4 = time.perf_counter()
        if mode == "slabs":
            be.lidar_push(rp.lidar[K - 1][0], rp.lidar[K - 1][1])
        if mode == "window":
            fr = rp.win_frame(K - 1); t4 = time.perf_counter()
            be.win_push_frame(fr)
        if mode != "classic":
            g_slide.append(1e3 * (time.perf_counter() - t4 + t3b - t3))

    def st(v):
        v = np.array(v)
        return {"median": float(np.median(v)), "p95": float(np.percentile(v, 95)), "max": float(v.max())}
    tot_g = np.array(g_solve) + np.array(g_marg)
    if g_slide:
        tot_g = tot_g[:len(g_slide)] + np.array(g_slide)          # dropping a frame and sending the new one up belongs to the image's latency
    tot_all = tot_g
    if mode == "window":                                           # `value`: the SERIALISED images (odd steps: the marginalisation kernels are waited for inside marg_ms)
        tot_g = tot_all[1::2]
    out = {"metric": "per-frame backend latency, synthetic replay (solve + gauge fix + marginalisation, host buffers in, host buffers out)",
           "value": float(np.median(tot_g)), "unit": "ms/frame", "higher_is_better": False, "n_gpus": 1, "frames": len(g_solve), "dtype": "f64" if args.precision == 0 else "f32 eval / f64 accumulate",
           "data": "synthetic replay (3indoor.bag unavailable offline)",
           "config": {"workload": "BASELINE.json configs[4] substitute: K=10, ~%d landmarks / ~%d visual factors per window, 30000 LiDAR points, every 5th image a non-keyframe (MARGIN_SECOND_NEW)" % (int(np.mean(Ls)), int(np.mean(nvis))),
                      "solver_limits": {"max_iterations": cap_it, "max_time_s": cap_t, "source": "short-cap variant (--cap8), not the reference's configuration" if args.cap8 else "config/mynteye_leishen_indoor.yaml:76-77 (estimator.cpp:1404,1411)"},
                      "iterations_per_frame": float(np.mean(its)), "iterations_per_frame_max": int(max(its)),
                      "share_of_frames_ended_by_max_iterations": float(np.mean(np.array(terms) == abi.TERM_NAMES.index("max_iterations"))),
                      "share_of_frames_ended_by_max_time": float(np.mean(np.array(terms) == abi.TERM_NAMES.index("max_time")))},
           "gpu": {"solve_ms": st(g_solve), "marg_ms": st(g_marg), "total_ms": st(tot_g), "note": "includes H2D upload of the window and D2H of the state / prior (PCIe-inclusive)",
                   "mode": {"window": "fully resident window (vil_win_*): observations, IMU samples / records, LiDAR points and the prior stay in HBM; per image the new frame (IMU samples pre-integrated on the device, observations, LiDAR points: slide_push_ms) and the window's small tables go up, the state comes back through pinned memory; the marginalisation is enqueued (marg_ms = the call) and its prior is written device-to-device -- its GPU time is inside the NEXT image's solve_ms",
                            "slabs": "round-2 residency: LiDAR frame slabs stay in HBM, gauge fix on the device, vil_marginalize_resident; visual / IMU tables, state and prior travel every image",
                            "classic": "every table handed over on every image (vil_solve + vil_gauge_fix + vil_marginalize)"}[mode]}}
    if g_slide:
        out["gpu"]["slide_push_ms"] = st(g_slide)
    if mode == "window":
        out["gpu"]["total_overlapped_ms"] = st(tot_all[0::2])
        out["gpu"]["marg_call_ms"] = st(np.array(g_marg)[0::2]); out["gpu"]["marg_serialised_ms"] = st(np.array(g_marg)[1::2])
        out["gpu"]["note"] = ("value / total_ms = images whose marginalisation kernels are waited for inside the timed region (every other image); total_overlapped_ms = the other "
                              "half, where vil_win_marginalize returns once its launches are enqueued and the GPU works while the host does its bookkeeping.  PCIe-inclusive: per image "
                              "the new frame + the small tables up, the state back")
    if orc is not None:
        tot_c = np.array(c_solve) + np.array(c_marg)
        out["cpu_baseline"] = {"solve_ms": st(c_solve), "marg_ms": st(c_marg), "total_ms": st(tot_c), "cores": 4, "kind": "port",
                               "cpu_solves_over_the_time_cap": float(np.mean(np.array(c_solve) > 1e3 * cap_t)) if cap_t > 0 else None,
                               "note": "CPU restatement on the same input windows, same iteration cap, NO wall-clock cap (it has to reproduce the device's solve for the state difference; "
                                       "cpu_solves_over_the_time_cap = share of its solves the reference's 0.05 s limit would have ended early on this host); solve single-threaded, marginalisation 4 threads (marginalization_factor.h:13)"}
        out["speedup_vs_cpu_baseline"] = float(np.median(tot_c) / np.median(tot_g))
        out["max_abs_position_difference_m"] = float(max(dpos))
    emit(out)


def vgicp_mode(args):
    """voxelised GICP scan-to-scan registration (the producer of the LiDAR ICP constraint,
    estimator.cpp:269-300).  Step = one linearisation (correspondences + 6x6 system) of a synthetic 16-ring scan pair with
    the clouds and the voxel map resident in HBM; plus whole alignments per second.  HIP events on the library's stream
    are not exposed for this row yet: the kernel time is the wall time of the call minus the measured D2H + sync floor."""
    import numpy as np
    import torch
    from mvil_fusion_amd import lib, vgicp
    rings, az = args.vgicp_rings, args.vgicp_az
    tx, tc, sx, sc, T_true = vgicp.make_pair(seed=20240606, rings=rings, az=az)
    g = vgicp.Vgicp(lib.load_vilsolve(), "vgicp_")
    g.set_target(tx, tc, 0.5); g.set_source(sx, sc)
    T = np.eye(4)
    for _ in range(args.warmup):
        g.linearize(T)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        e, H, b, nc = g.linearize(T)
    el = time.perf_counter() - t0
    # roofline leg: the same calls with HIP events on the library's stream around k_vgicp_lin
    g.lib.vgicp_profile_enable(g.ctx, 1)
    for _ in range(args.steps):
        g.linearize(T)
    pn, pms = C.c_int64(), C.c_double()
    g.lib.vgicp_profile_read(g.ctx, C.byref(pn), C.byref(pms))
    g.lib.vgicp_profile_enable(g.ctx, 0)
    k_us = 1e3 * pms.value / max(1, pn.value)
    t0 = time.perf_counter(); na = 0
    while time.perf_counter() - t0 < 1.0:
        Tg, sg = g.align(np.eye(4)); na += 1
    el_a = time.perf_counter() - t0
    g.covariances(sx, 20)
    t0 = time.perf_counter()
    for _ in range(5):
        g.covariances(sx, 20)
    cov_ms = 1e3 * (time.perf_counter() - t0) / 5
    n = len(sx)
    # algorithmic bytes of one linearisation (DIRECT1): source point 12 + covariance 72, per correspondence voxel record
    # (num 4 + mean 24 + cov 72) + key probe 8 + slot 4, stored correspondence 4 + 72
    ab = n * (12 + 72 + 4) + nc * (100 + 12 + 72)
    out = {"metric": "VGICP linearisations/sec (scan-to-scan, %d-ring x %d synthetic scan pair, voxel 0.5 m, DIRECT1)" % (rings, az),
           "value": args.steps / el, "unit": "linearisations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "SURVEY 8(f) row 1: %d source points, %d correspondences, target voxel map resident" % (n, nc),
                      "alignments_per_s": na / el_a, "lm_iterations_per_alignment": int(sg.iterations), "covariances_20nn_ms": cov_ms,
                      "translation_error_m": float(np.abs(Tg[:3, 3] - T_true[:3, 3]).max())},
           "roofline": {"bound": "hbm", "kernel": "k_vgicp_lin", "achieved": ab / (k_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ab / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                        "traffic": (pmc_traffic_bytes(("r01_vgicp64_pmc_FETCH_SIZE.csv", "r01_vgicp64_pmc_WRITE_SIZE.csv"), "k_vgicp_lin") if (rings, az) == (64, 2048) else
                                    (pmc_traffic_bytes(("r04_vgicp16_pmc_FETCH_SIZE.csv", "r04_vgicp16_pmc_WRITE_SIZE.csv"), "k_vgicp_lin") if (rings, az) == (16, 1800) else None)),
                        "traffic_source": "profiles/r04_vgicp16_pmc_*.csv (16-ring x 1800 pair, tools/collect_profiles_r04.sh) / profiles/r01_vgicp64_pmc_*.csv (64-ring x 2048 pair, tools/collect_pmc_rows.sh)",
                        "algorithmic_bytes_per_launch": ab, "avg_launch_us": k_us, "launches_timed": int(pn.value),
                        "note": "HIP events on the library's stream around the kernel; a whole vgicp_linearize call is %.1f us of wall time (2 launches + D2H of 29 doubles + stream sync)" % (1e6 * el / args.steps)}}
    if not args.no_cpu:
        orc = vgicp.Vgicp(C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_vgicp_")
        orc.set_target(tx, tc, 0.5); orc.set_source(sx, sc)
        for _ in range(2):
            orc.linearize(T)
        t0 = time.perf_counter(); k = 0
        while time.perf_counter() - t0 < 3.0:
            orc.linearize(T); k += 1
        elc = time.perf_counter() - t0
        t0 = time.perf_counter(); ka = 0
        while time.perf_counter() - t0 < 3.0:
            To, so = orc.align(np.eye(4)); ka += 1
        elca = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": k / elc, "unit": "linearisations/s", "cores": 1, "kind": "port", "sample": "%d linearisations, %.1f s; %d alignments %.1f s" % (k, elc, ka, elca),
                               "alignments_per_s": ka / elca,
                               "note": "single-threaded restatement of fast_gicp's FastVGICP (the reference runs it with OpenMP NumThreads from the yaml)"}
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        out["max_abs_T_difference"] = float(np.abs(Tg - To).max())
    emit(out)
    g.close()


def mapreg_mode(args):
    """LiDAR scan-to-map registration (localMapping.cpp:590-791).  Step = one whole vmap_align (two rounds
    of association + 7-parameter solve) of a synthetic scan against a resident local map; the roofline leg times the surf
    association kernel (10-NN + intensity re-rank + plane fit per scan point) with HIP events on the library's stream."""
    import numpy as np
    import torch
    from mvil_fusion_amd import lib, mapreg
    from mvil_fusion_amd.vgicp import _rot
    cm, sm = mapreg.make_map(seed=20240607, n_surf=args.map_surf, n_corner=args.map_corner)
    R, t = _rot(0.01, -0.015, 0.5), np.array([1.5, -1.0, 0.25])
    sc, ss = mapreg.make_scan(cm, sm, R, t, seed=11, n_surf=args.scan_surf, n_corner=args.scan_corner)
    q0 = mapreg.quat_from_R(R @ _rot(0.004, -0.003, 0.008)); t0 = t + np.array([0.05, -0.04, 0.03])
    so = lib.load_vilsolve()
    be = lib.open_vilsolve()
    g = mapreg.MapReg(so, "vmap_")
    g.set_map(cm, sm)                                                  # first call allocates; every later scan's map reuses the buffers
    t_map0 = time.perf_counter(); g.set_map(cm, sm); t_map = time.perf_counter() - t_map0
    for _ in range(args.warmup):
        g.align(be.ctx, sc, ss, q0, t0)
    torch.cuda.synchronize()
    t0w = time.perf_counter()
    for _ in range(args.steps):
        qg, tg, sg = g.align(be.ctx, sc, ss, q0, t0)
    el = time.perf_counter() - t0w
    t0w = time.perf_counter()
    for _ in range(args.steps):
        eg, pg = g.associate(sc, ss, q0, t0)
    el_assoc = time.perf_counter() - t0w
    g.lib.vmap_profile_enable(g.ctx, 1)
    for _ in range(args.steps):
        g.associate(sc, ss, q0, t0)
    pn, pms = (C.c_int64 * 2)(), (C.c_double * 2)()
    g.lib.vmap_profile_read(g.ctx, pn, pms)
    g.lib.vmap_profile_enable(g.ctx, 0)
    s_us, f_us = 1e3 * pms[0] / max(1, pn[0]), 1e3 * pms[1] / max(1, pn[1])
    # algorithmic bytes of the search kernel per scan point: the point 16, 27 cell probes x (key 8 + start/count 8), the candidates of
    # those cells (the library sizes cells for ~6 points each) x (xyz 12 + index 4), result 44
    nq = len(sc) + len(ss)
    ab = nq * (16 + 27 * 16 + 27 * 6 * 16 + 44)
    out = {"metric": "scan-to-map registrations/sec (lidar_mapping, %d+%d scan points vs %d+%d map points, 2 rounds)" % (len(sc), len(ss), len(cm), len(sm)),
           "value": args.steps / el, "unit": "registrations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 search / f64 fit+solve", "data": "synthetic",
           "config": {"workload": "SURVEY 8(f) row 2: association (%d edge + %d plane factors) + one-launch 6-dof DOGLEG solve, twice, one submission per registration" % (sg.n_edge, sg.n_plane),
                      "associate_ms": 1e3 * el_assoc / args.steps, "set_map_ms": 1e3 * t_map, "solve_iterations_last_round": int(sg.iterations),
                      "translation_error_m": float(np.linalg.norm(tg - t)), "k_map_fit_us": f_us},
           "roofline": {"bound": "hbm", "kernel": "k_map_search", "achieved": ab / (s_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ab / (s_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                        "traffic": pmc_traffic_bytes(("r01_mapreg_pmc_FETCH_SIZE.csv", "r01_mapreg_pmc_WRITE_SIZE.csv"), "k_map_search") if (len(sc), len(ss), len(cm), len(sm)) == (800, 6000, 8000, 60000) else None,
                        "traffic_source": "profiles/r01_mapreg_pmc_*.csv (default sizes only; tools/collect_pmc_rows.sh)",
                        "algorithmic_bytes_per_launch": ab, "avg_launch_us": s_us, "launches_timed": int(pn[0]),
                        "note": "one wave per scan point: 27-cell gather, coalesced candidate loads, sorted list across the wave's lanes"}}
    if not args.no_cpu:
        o = mapreg.MapReg(C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_vmap_")
        o.lib.orc_vmap_set_search(o.ctx, C.c_int32(1))            # kd-tree, as pcl::KdTreeFLANN in the reference
        o.set_map(cm, sm)
        t0w = time.perf_counter(); o.set_map(cm, sm); t_map_c = time.perf_counter() - t0w
        o.align(None, sc, ss, q0, t0)
        t0w = time.perf_counter(); k = 0
        while time.perf_counter() - t0w < 5.0:
            qo, to, so_ = o.align(None, sc, ss, q0, t0); k += 1
        elc = time.perf_counter() - t0w
        out["cpu_baseline"] = {"value": k / elc, "unit": "registrations/s", "cores": 1, "kind": "port", "sample": "%d registrations, %.1f s" % (k, elc), "set_map_ms": 1e3 * t_map_c,
                               "note": "single-threaded restatement: exact kd-tree kNN + the same fits + the oracle's dense solver (the reference uses pcl kd-trees + Ceres, single-threaded per scan)"}
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        out["max_abs_t_difference_m"] = float(np.abs(tg - to).max())
        o.close()
    emit(out)
    g.close(); be.close()


def preint_mode(args):
    """IMU pre-integration (integration_base.h:30-158).  Step = re-propagation of all K-1 = 9
    intervals of a configs[1] window (20-40 samples each at 200 Hz), host buffers in and out; the roofline leg times k_preint
    with HIP events on the library's stream."""
    import numpy as np
    from mvil_fusion_amd import lib, preint
    s = preint.make_stream(n_intervals=args.preint_intervals, samples=(20, 40), seed=20240608)
    ns = int(s[0][-1])
    g = preint.Preint(lib.load_vilsolve(), "vpre_")
    call, rg, jg = g.bind(*s)                                          # arguments marshalled once: the timed region is the C call
    for _ in range(args.warmup):
        call()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        call()
    el = time.perf_counter() - t0
    g.lib.vpre_profile_enable(g.ctx, 1)
    for _ in range(args.steps):
        call()
    pn, pms = C.c_int64(), C.c_double()
    g.lib.vpre_profile_read(g.ctx, C.byref(pn), C.byref(pms))
    k_us = 1e3 * pms.value / max(1, pn.value)
    n = args.preint_intervals
    ab = ns * 56 + n * (96 + 8 * 287 + 8 * 225)                  # samples (dt, acc, gyr) in; per interval acc0/gyr0/ba/bg in, record + jacobian out
    fl = ns * 2 * (3 * 15 ** 3 + 15 * 15 * 18 * 2)               # F J, F P, (F P) F^T, V N V^T
    out = {"metric": "IMU re-propagations/sec (all %d intervals of a window, %d samples)" % (n, ns), "value": args.steps / el, "unit": "windows/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic", "config": {"workload": "SURVEY 8(f) row 3: mid-point pre-integration with 15x15 jacobian / covariance propagation", "samples": ns, "intervals": n},
           "roofline": {"bound": "hbm", "kernel": "k_preint", "achieved": ab / (k_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ab / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                        "traffic": None, "algorithmic_bytes_per_launch": ab, "avg_launch_us": k_us, "launches_timed": int(pn.value), "gflops": fl / (k_us * 1e-6) / 1e9,
                        "note": "one workgroup per interval, batches of 16 samples folded by a pairwise tree of affine maps: latency / LDS bound, neither HBM nor MFMA"}}
    if not args.no_cpu:
        o = preint.Preint(C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_vpre_")
        ocall, ro, jo = o.bind(*s)
        ocall()
        t0 = time.perf_counter(); k = 0
        while time.perf_counter() - t0 < 3.0:
            ocall(); k += 1
        elc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": k / elc, "unit": "windows/s", "cores": 1, "kind": "port", "sample": "%d windows, %.1f s" % (k, elc),
                               "note": "single-threaded restatement with static arrays (the reference allocates MatrixXd F, V per sample)"}
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        out["max_rel_covariance_difference"] = float(np.abs(rg[:, 62:] - ro[:, 62:]).max() / np.abs(ro[:, 62:]).max())
        o.close()
    emit(out)
    g.close()


def solve_batch(bes, opts, abi):
    n = len(bes)
    ctxs = (C.c_void_p * n)(*[b.ctx for b in bes])
    sums = (abi.VilSummary * n)(); st = (C.c_int32 * n)()
    f = bes[0].lib.vil_solve_batch; f.restype = C.c_int
    rc = f(ctxs, C.c_int32(n), C.byref(opts), sums, st)
    if rc != 0:
        raise RuntimeError("vil_solve_batch: status %d" % rc)
    return sum(s.iterations for s in sums)


def concurrent_windows(lib, abi, w, opts, steps, device=0, Bs=(1, 2, 4, 8)):
    """B contexts of the SAME window solved concurrently (vil_solve_batch: a stream + a host thread each, one launch per iteration): aggregate iterations/s.
    One window keeps one master workgroup busy for two thirds of an iteration -- this is what the device does with the rest; the single-GPU form of `replicas`."""
    import torch
    ab = algorithmic_bytes(w)
    bes = [lib.open_vilsolve(device=device) for _ in range(max(Bs))]
    for be in bes:
        be.upload(w)
    rows = []
    for B in Bs:
        grp = bes[:B]
        for _ in range(3):
            for be in grp: be.reset_state()
            solve_batch(grp, opts, abi)
        torch.cuda.synchronize(); t0 = time.perf_counter(); its = 0
        for _ in range(steps):
            for be in grp: be.reset_state()
            its += solve_batch(grp, opts, abi)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        rows.append({"B": B, "aggregate_it_per_s": its / el, "ms_per_batch": 1e3 * el / steps, "hbm_frac_of_the_sweeps": (its + B * steps) * ab / el / 1e9 / HBM_PEAK_GBS})
    for be in bes:
        be.close()
    base = rows[0]["aggregate_it_per_s"]
    for r in rows:
        r["speedup_vs_B1"] = r["aggregate_it_per_s"] / base
    return {"what": "vil_solve_batch: B resident contexts of the configs[1] window, one launch per trust-region iteration each (k_iter), a stream and a host thread per context; every window's result is bit-equal to its solo solve (tests/test_gpu_batch.py)",
            "steps_per_B": steps, "rows": rows, "note": "B = 1 here is one launch per iteration through the batch entry point (a thread per call); the headline `value` is the persistent solve of ONE window"}


def batch_mode(args):
    import torch
    graft.load_package()
    from mvil_fusion_amd import abi, lib, synth
    be0 = lib.open_vilsolve()
    def gpu_prior(pre):
        return be0.marginalize(pre).to_prior()
    w = synth.make_config(2, prior_fn=gpu_prior)
    be0.close()
    out = {"metric": "concurrent windows on one GPU: aggregate solve iterations/sec (10 KF, 1k feat, 30k LiDAR pts per window)", "unit": "iterations/s", "n_gpus": 1, "dtype": "f64", "data": "synthetic", "higher_is_better": True}
    out["concurrent_windows"] = concurrent_windows(lib, abi, w, abi.default_options(), args.steps)
    out["value"] = max(r["aggregate_it_per_s"] for r in out["concurrent_windows"]["rows"])
    emit(out)


def scale_sweep_mode(args):
    """What the kernels do when there IS work: the configs[1] shape with s x the landmarks and LiDAR points, K = 10.  Per size: the sweep phase of the
    iteration (the launch's own clock stamps / the sweep launch's HIP events) against the HBM roof, the whole iteration, and which launch structure the library took.
    The --pmc passes of this command (tools/collect_profiles_r06.sh) add FETCH / WRITE / MFMA per size to profiles/r06_scale_sweep.txt."""
    import torch
    graft.load_package()
    from mvil_fusion_amd import abi, lib, synth
    opts = abi.default_options()
    rows = []
    for sc in [int(v) for v in args.scales.split(",")]:
        w = synth.make_config(2, L=1000 * sc, n_plane=24000 * sc, n_edge=6000 * sc)      # (no marginalisation prior: the generator's synthetic one)
        be = lib.open_vilsolve()
        if args.launch_mode:
            be.lib.vil_debug_set_launch_mode(be.ctx, args.launch_mode)
        try:
            be.upload(w)
        except lib.VilError as e:
            rows.append({"scale": sc, "error": "upload status %d" % e.status}); be.close(); continue
        lpi, one = C.c_int32(0), C.c_int32(0)
        be.lib.vil_debug_get_launch_structure(be.ctx, C.byref(lpi), C.byref(one))
        n = max(3, args.steps // max(1, sc))
        for _ in range(2):
            be.reset_state(); be.solve_resident(opts)
        torch.cuda.synchronize(); t0 = time.perf_counter(); its = 0
        for _ in range(n):
            be.reset_state(); its += be.solve_resident(opts).iterations
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        row = {"scale": sc, "landmarks": w.L, "visual_factors": int(len(w.vis_i)), "lidar_points": int(len(w.plane_pose) + len(w.edge_pose)), "launches_per_iteration": int(lpi.value),
               "iterations_per_s": its / el, "us_per_iteration": 1e6 * el / its, "algorithmic_bytes_per_sweep": algorithmic_bytes(w)}
        if not args.no_events:
            prof = VilProfile()
            be.lib.vil_profile_enable(be.ctx, 2 if lpi.value == 0 else 1)
            be.reset_state(); be.solve_resident(opts); be.lib.vil_profile_read(be.ctx, C.byref(prof), 1)
            for _ in range(n):
                be.reset_state(); be.solve_resident(opts)
            be.lib.vil_profile_read(be.ctx, C.byref(prof), 1)
            be.lib.vil_profile_enable(be.ctx, 0)
            if prof.sweep_launches > 0:
                sweep_us = 1e3 * prof.sweep_ms / prof.sweep_launches
                row["sweep_phase_us"] = sweep_us
                row["sweep_phase_GBps"] = row["algorithmic_bytes_per_sweep"] / (sweep_us * 1e-6) / 1e9
                row["sweep_phase_frac_of_hbm_peak"] = row["sweep_phase_GBps"] / HBM_PEAK_GBS
        rows.append(row)
        be.close()
    emit({"metric": "factor sweep against the HBM roof as the window grows (K = 10)", "unit": "GB/s", "value": max([r.get("sweep_phase_GBps", 0.0) for r in rows] + [0.0]), "n_gpus": 1, "dtype": "f64", "data": "synthetic",
          "higher_is_better": True, "peak": HBM_PEAK_GBS, "scale_sweep": rows,
          "what": "configs[1] shape x scale: the sweep phase = first workgroup started -> last sweep role's record out (one-launch structures: the launch's own 100 MHz stamps; otherwise HIP events around k_sweep)"})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--no-events", action="store_true", help="do not record HIP events around sweep launches in the timed region")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--replay", type=int, default=0, help="config 5: run N images of the synthetic replay and report per-frame latency instead of the headline metric")
    ap.add_argument("--no-cfg3", dest="no_cfg3", action="store_true", help="skip the extra configs[2] (K=10, L=4000, 120k points) leg")
    ap.add_argument("--no-tracker", dest="no_tracker", action="store_true", help="skip the PCIe-inclusive tracker legs (counter passes: their configs[1]-shaped windows launch the same kernels as the window under test)")
    ap.add_argument("--cap8", action="store_true", help="replay mode: the short-cap variant of earlier rounds (8 iterations, no time cap) instead of the reference's limits (30 iterations / 0.05 s)")
    ap.add_argument("--classic", action="store_true", help="replay mode: hand every table over on every image (vil_solve + vil_gauge_fix + vil_marginalize) instead of the resident-window entry points")
    ap.add_argument("--slabs", action="store_true", help="replay mode: round-2 residency (LiDAR frame slabs + vil_marginalize_resident) instead of the fully resident window (vil_win_*)")
    ap.add_argument("--precision", type=int, default=0, help="0 = fp64 (reference arithmetic), 1 = fp32 factor evaluation with fp64 accumulation (replay mode only)")
    ap.add_argument("--vgicp", action="store_true", help="SURVEY 8(f) row 1: bench the voxelised GICP linearisation instead of the headline metric")
    ap.add_argument("--vgicp-rings", type=int, default=16)
    ap.add_argument("--vgicp-az", type=int, default=1800)
    ap.add_argument("--mapreg", action="store_true", help="SURVEY 8(f) row 2: bench the LiDAR scan-to-map registration instead of the headline metric")
    ap.add_argument("--map-surf", type=int, default=60000); ap.add_argument("--map-corner", type=int, default=8000)
    ap.add_argument("--scan-surf", type=int, default=6000); ap.add_argument("--scan-corner", type=int, default=800)
    ap.add_argument("--preint", action="store_true", help="SURVEY 8(f) row 3: bench the IMU pre-integration instead of the headline metric")
    ap.add_argument("--preint-intervals", type=int, default=9)
    ap.add_argument("--batch", action="store_true", help="concurrent windows on one GPU (vil_solve_batch): aggregate iterations/s for B = 1, 2, 4, 8 contexts of the configs[1] window")
    ap.add_argument("--scale-sweep", dest="scale_sweep", action="store_true", help="the configs[1] shape at 1x, 4x, 16x, 64x landmarks and LiDAR points (K = 10): what the sweep does when it has work")
    ap.add_argument("--scales", type=str, default="1,4,16,64")
    ap.add_argument("--launch-mode", dest="launch_mode", type=int, default=0, help="scale sweep: vil_debug_set_launch_mode (4 = one launch per iteration also where the persistent solve would be taken: the counter passes)")
    ap.add_argument("--force-comm", action="store_true", help="test hook: take the multi-GPU code path (process group, communicator, replicas leg) with a single rank")
    ap.add_argument("--replicas", action="store_true", help="N>1: independent replicas instead of sharding one window over RCCL")
    ap.add_argument("--ipc", action="store_true", help="N>1: the library's peer-buffer exchange (IPC-mapped inboxes over xGMI, vil_comm_ipc_*) instead of RCCL for the per-iteration collective")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher -- one rank per GPU under torch.distributed.run, rendezvous on 127.0.0.1
        # (what the driver's own `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` command line does)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" in os.environ and args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    import torch
    dist = None
    cdev = "cuda"          # where the tensors of the harness's own exchanges live (handles, timings)
    n_dev = torch.cuda.device_count()
    oversub = world > max(1, n_dev)      # more ranks than devices (the one-device test of the N > 1 path): ranks share devices, RCCL cannot (one rank per GPU) -> gloo + peer buffers
    local = local % max(1, n_dev)
    if world > 1 or args.force_comm:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if oversub:
            if not (args.ipc or args.replicas):
                raise SystemExit("bench.py: %d ranks on %d device(s) needs --ipc or --replicas (RCCL wants one GPU per rank)" % (world, n_dev))
            cdev = "cpu"
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    graft.load_package()
    from mvil_fusion_amd import abi, lib, synth
    if not os.path.exists(lib.LIB_PATH) or not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        if rank == 0:
            graft.build()              # a checkout without built artefacts (the .so files are git-ignored): compile, do not fall back
        if dist is not None:
            dist.barrier()
    if args.batch:
        return batch_mode(args)
    if args.scale_sweep:
        return scale_sweep_mode(args)
    if args.vgicp:
        vgicp_mode(args)
        return
    if args.mapreg:
        mapreg_mode(args)
        return
    if args.preint:
        preint_mode(args)
        return

    be = lib.open_vilsolve(device=local, rank=rank, world=world)
    opts = abi.default_options()
    if args.replay > 0:
        if world > 1:
            raise SystemExit("--replay is a single-GPU mode")
        replay_mode(args, be, abi, lib)
        be.close()
        return
    sharded = False
    shard_note = None
    if (world > 1 or args.force_comm) and not args.replicas and args.ipc:
        # peer-buffer exchange instead of RCCL (vil_comm_ipc_*): every rank exports its inbox, the 64-byte handles travel through torch.distributed
        h = (C.c_char * 64)()
        fexp = be.lib.vil_comm_ipc_export; fexp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p]
        st = fexp(be.ctx, rank, world, 1 << 19, h)
        mine = torch.frombuffer(bytearray(bytes(h)), dtype=torch.uint8).to(cdev)
        allh = [torch.zeros(64, dtype=torch.uint8, device=cdev) for _ in range(world)]
        dist.all_gather(allh, mine)
        blob = b"".join(bytes(t.cpu().numpy().tobytes()) for t in allh)
        if st == 0:
            st = be.lib.vil_comm_ipc_init(be.ctx, (C.c_char * len(blob)).from_buffer_copy(blob))
        good = torch.tensor([1 if st == 0 else 0], device=cdev, dtype=torch.int32)
        dist.all_reduce(good, op=dist.ReduceOp.MIN)
        if int(good[0]) == 1:
            sharded = True
        else:
            shard_note = "peer-buffer exchange could not be set up (status %d on rank %d): fell back to independent replicas" % (st, rank)
            be.close()
            be = lib.open_vilsolve(device=local, rank=0, world=1)
    elif (world > 1 or args.force_comm) and not args.replicas:
        # one RCCL communicator over xGMI, created inside the library; the 128-byte id travels through torch.distributed.
        # Every rank must agree on the outcome: if any rank cannot join, ALL fall back to independent replicas (reported).
        uid = (C.c_char * 128)()
        ok = 1
        if rank == 0:
            ok = 1 if be.lib.vil_comm_unique_id(uid) == 0 else 0
        t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).cuda()
        dist.broadcast(t, 0)
        flag = torch.tensor([ok], device="cuda", dtype=torch.int32)
        dist.broadcast(flag, 0)
        st = -5
        if int(flag[0]) == 1:
            uid = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
            st = be.lib.vil_comm_init(be.ctx, uid, rank, world)
        good = torch.tensor([1 if st == 0 else 0], device="cuda", dtype=torch.int32)
        dist.all_reduce(good, op=dist.ReduceOp.MIN)
        if int(good[0]) == 1:
            sharded = True
        else:
            shard_note = "RCCL communicator could not be created inside the library (status %d on rank %d): fell back to independent replicas" % (st, rank)
            be.close()
            be = lib.open_vilsolve(device=local, rank=0, world=1)

    # what the LIBRARY's communicator says it spans (not what the launcher hoped for): asserted, and printed in the JSON line
    comm_info = None
    if dist is not None:
        cr, cw, ctp = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
        be.lib.vil_comm_info(be.ctx, C.byref(cr), C.byref(cw), C.byref(ctp))
        mine = torch.tensor([cr.value, cw.value, ctp.value, local], device=cdev, dtype=torch.int32)
        alli = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(alli, mine)
        rows = [[int(v) for v in t_.cpu()] for t_ in alli]
        if sharded and not all(r_[0] == q and r_[1] == world for q, r_ in enumerate(rows)):
            raise SystemExit("bench.py: the library's communicator does not span the %d ranks the launcher started: %r" % (world, rows))
        comm_info = {"columns": ["rank", "world", "transport (0 none, 1 RCCL, 2 in-process, 3 peer buffers)", "device"], "per_rank": rows,
                     "ranks_seen_by_the_library": rows[0][1] if sharded else 1, "devices_visible": n_dev, "harness_backend": "gloo" if cdev == "cpu" else "nccl"}
    got_prior = {"lib": False}

    def gpu_prior(pre):
        try:
            pr = be.marginalize(pre).to_prior()
            got_prior["lib"] = pr is not None
            return pr
        except lib.VilError:
            return None     # -> synthetic prior (stated in config)
    w = synth.make_config(args.config, prior_fn=gpu_prior)
    prior_kind = "vil_marginalize of the preceding synthetic window, on the GPU" if got_prior["lib"] else "synthetic dense prior"
    be.upload(w)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def run(n):
        its, last_ = 0, None
        for _ in range(n):
            be.reset_state()
            last_ = be.solve_resident(opts)
            its += last_.iterations
        return its, last_

    run(args.warmup)
    # the timed region: EXACTLY args.steps steps between barrier + synchronise on both sides -- repeated REPEATS times back to back (an
    # 18 ms region is noisy on a shared host): `value` / `ms_per_step` are the MEDIAN region, all regions are printed in `repeats`
    REPEATS = 5
    regions = []
    for _ in range(REPEATS):
        sync()
        t0 = time.perf_counter()
        iters, last = run(args.steps)
        sync()
        regions.append((time.perf_counter() - t0, iters))
    el, iters = sorted(regions)[len(regions) // 2]
    # roofline leg: the SAME K steps once more with HIP events recorded on the library's stream around every sweep
    # launch.  Kept out of the `value` region because three event records per ~170 us iteration perturb this
    # latency-bound pipeline by several percent.
    prof = VilProfile()
    phases_head = None
    lpi0, one0 = C.c_int32(0), C.c_int32(0)
    be.lib.vil_debug_get_launch_structure(be.ctx, C.byref(lpi0), C.byref(one0))
    persistent = one0.value == 1 and lpi0.value == 0      # the whole solve is one resident launch (k_solve): events around THAT launch + its own stamps (profile mode 2 keeps the launch structure)
    if not args.no_events:
        be.lib.vil_profile_enable(be.ctx, 2 if persistent else 1)
        run(2)
        be.lib.vil_profile_read(be.ctx, C.byref(prof), 1)
        sync()
        t1 = time.perf_counter()
        run(args.steps)
        sync()
        el_events = time.perf_counter() - t1
        be.lib.vil_profile_read(be.ctx, C.byref(prof), 1)
        phases_head = phases_obj(be)
        be.lib.vil_profile_enable(be.ctx, 0)
    lpi, one = C.c_int32(0), C.c_int32(0)
    be.lib.vil_debug_get_launch_structure(be.ctx, C.byref(lpi), C.byref(one))
    one_launch = one.value == 1
    # the same window through the TWO-launch structure of round 4 (k_sweep, then gather + step): the sweep kernel in isolation -- its HIP-event average is what a
    # rocprofv3 --kernel-trace of that structure shows -- and the iteration rate the one-launch iteration is measured against.  Not `value`.
    two_launch = None
    if one_launch and world == 1 and not args.no_events:
        be3 = lib.open_vilsolve(device=local)
        be3.lib.vil_debug_set_launch_mode(be3.ctx, 3)
        be3.upload(w)
        for _ in range(3):
            be3.reset_state(); be3.solve_resident(opts)
        torch.cuda.synchronize(); t3 = time.perf_counter(); it3 = 0
        n3s = max(8, args.steps // 2)
        for _ in range(n3s):
            be3.reset_state(); it3 += be3.solve_resident(opts).iterations
        torch.cuda.synchronize(); el3 = time.perf_counter() - t3
        p3 = VilProfile()
        be3.lib.vil_profile_enable(be3.ctx, 1)
        be3.reset_state(); be3.solve_resident(opts); be3.lib.vil_profile_read(be3.ctx, C.byref(p3), 1)
        for _ in range(n3s):
            be3.reset_state(); be3.solve_resident(opts)
        be3.lib.vil_profile_read(be3.ctx, C.byref(p3), 1)
        be3.close()
        ab3 = algorithmic_bytes(w); us3 = 1e3 * p3.sweep_ms / max(1, p3.sweep_launches)
        two_launch = {"value": it3 / el3, "unit": "iterations/s", "steps": n3s, "what": "vil_debug_set_launch_mode(3): k_sweep, then the merged gather + step launch (round 4's structure), same window, same protocol",
                      "k_sweep": {"kernel": "k_sweep<%s>" % ("5" if w.K > 12 else "2"), "avg_launch_us": us3, "achieved": ab3 / (us3 * 1e-6) / 1e9, "unit": "GB/s", "frac": ab3 / (us3 * 1e-6) / 1e9 / HBM_PEAK_GBS, "launches_timed": int(p3.sweep_launches)},
                      "k_step_avg_us": 1e3 * p3.step_ms / max(1, p3.step_launches)}
    tot_iters, max_el = iters, el
    region_vals = [i_ / e_ for e_, i_ in regions]
    # N > 1: where an iteration's time goes on EVERY rank (HIP events of the instrumented pass) -- the curve should explain itself
    phases = None
    if dist is not None and prof.sweep_launches > 0:
        n_it = max(1, prof.step_launches)
        mine = torch.tensor([1e3 * prof.sweep_ms / max(1, prof.sweep_launches), 1e3 * prof.reduce_ms / n_it, 1e3 * prof.collective_ms / n_it,
                             1e3 * (prof.step_ms - prof.reduce_ms - prof.collective_ms) / n_it], device=cdev, dtype=torch.float64)
        allp = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        phases = {"unit": "us per iteration (HIP events on the library's stream, instrumented pass)", "columns": ["sweep", "gather", "collective", "step"],
                  "per_rank": [[round(float(v), 2) for v in t_.cpu()] for t_ in allp]}
    if dist is not None:
        # every region: MAX over ranks of its time; sharded: all ranks work on the SAME solves (units = its iterations, counted once); replicas: units add up
        tt = torch.tensor([[e_, float(i_)] for e_, i_ in regions], device=cdev, dtype=torch.float64)
        t_max = tt[:, 0].clone(); dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        i_sum = tt[:, 1].clone(); dist.all_reduce(i_sum, op=dist.ReduceOp.SUM)
        units = tt[:, 1] if sharded else i_sum
        region_vals = [float(u / t) for u, t in zip(units, t_max)]
        order = sorted(range(len(regions)), key=lambda q: region_vals[q])
        mid = order[len(order) // 2]
        tot_iters, max_el = float(units[mid]), float(t_max[mid])
    # N > 1, sharded: a second, clearly labelled leg with the SAME K steps run as N independent replicas (one whole window
    # per GPU, no collective) -- the throughput a multi-session deployment gets from the same hardware.  Not `value`.
    replicas_leg = None
    if sharded and dist is not None:
        be2 = lib.open_vilsolve(device=local, rank=0, world=1)
        be2.upload(w)
        for _ in range(max(1, args.warmup)):
            be2.reset_state(); be2.solve_resident(opts)
        sync()
        t2 = time.perf_counter(); its2 = 0
        for _ in range(args.steps):
            be2.reset_state(); its2 += be2.solve_resident(opts).iterations
        sync()
        el2 = time.perf_counter() - t2
        tt2 = torch.tensor([float(its2), el2], device=cdev, dtype=torch.float64)
        s2 = tt2.clone(); dist.all_reduce(s2, op=dist.ReduceOp.SUM)
        m2 = tt2.clone(); dist.all_reduce(m2, op=dist.ReduceOp.MAX)
        replicas_leg = {"value": float(s2[0]) / float(m2[1]), "unit": "iterations/s", "scaling": "weak", "what": "%d independent replicas of the same window, %d steps each, no collective" % (world, args.steps)}
        be2.close()
    # what a tracker gets: a fresh upload per image (vil_solve = pack + H2D + set-up + solve + read-back), never a re-solve of a resident
    # upload -- reported beside `value`, never as `value`
    pcie_leg = None
    if world == 1 and not args.no_tracker:
        saved = w.state_copy()
        for _ in range(2):
            w.set_state(saved); be.solve(w, opts)
        n_p = max(5, args.steps // 2); its_p = 0
        torch.cuda.synchronize(); tp = time.perf_counter()
        for _ in range(n_p):
            w.set_state(saved); its_p += be.solve(w, opts).iterations
        torch.cuda.synchronize(); elp = time.perf_counter() - tp
        w.set_state(saved)
        pcie_classic = {"value": its_p / elp, "unit": "iterations/s", "ms_per_solve": 1e3 * elp / n_p, "host_to_device_bytes_per_solve": int(8 * (14 * len(w.vis_i) + 7 * len(w.plane_pose) + 9 * len(w.edge_pose) + 287 * len(w.imu_i) + w.prior.n ** 2 + 3 * (16 * w.K + 8 + w.L))),
                        "what": "vil_solve per step: EVERY host table packed and uploaded (2.3 MB), solved, state read back (%d solves); first solve of an upload launches directly (no hipGraph)" % n_p}
        pcie_leg = tracker_leg(lib, abi, local)
    # BASELINE.json configs[2] (K = 10, L = 4000, 120 k LiDAR points: the window the 8-GPU sharding is specified on) and configs[3] (K = 20, prior active:
    # the "dense Schur block, MFMA path" window) -- same protocol as the headline, fewer steps; each with its own roofline object
    leg_phases, leg_one = {}, {}
    def window_leg(cfg_id):
        wx = synth.make_config(cfg_id, prior_fn=gpu_prior)
        be.upload(wx)
        for _ in range(2):
            be.reset_state(); be.solve_resident(opts)
        sync()
        nx = max(4, args.steps // 4); tx = time.perf_counter(); itx = 0
        for _ in range(nx):
            be.reset_state(); lx = be.solve_resident(opts); itx += lx.iterations
        sync()
        elx = time.perf_counter() - tx
        if dist is not None:
            ttx = torch.tensor([float(itx), elx], device=cdev, dtype=torch.float64)
            sx = ttx.clone(); dist.all_reduce(sx, op=dist.ReduceOp.SUM); mx = ttx.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            itx, elx = (float(itx) if sharded else float(sx[0])), float(mx[1])
        profx = VilProfile()
        if not args.no_events:
            be.lib.vil_profile_enable(be.ctx, 1)
            be.reset_state(); be.solve_resident(opts)
            be.lib.vil_profile_read(be.ctx, C.byref(profx), 1)
            for _ in range(nx):
                be.reset_state(); be.solve_resident(opts)
            be.lib.vil_profile_read(be.ctx, C.byref(profx), 1)
            leg_phases[cfg_id] = phases_obj(be)
            be.lib.vil_profile_enable(be.ctx, 0)
        lx_, ox_ = C.c_int32(0), C.c_int32(0)
        be.lib.vil_debug_get_launch_structure(be.ctx, C.byref(lx_), C.byref(ox_)); leg_one[cfg_id] = ox_.value == 1
        leg = {"value": itx / elx, "unit": "iterations/s", "n_gpus": world, "ms_per_step": 1e3 * elx / nx, "steps": nx, "iterations_per_solve": lx.iterations,
               "workload": "BASELINE.json configs[%d]: K=%d, L=%d, %d visual factors, %d LiDAR points, prior n=%d, %s" % (cfg_id - 1, wx.K, wx.L, len(wx.vis_i), len(wx.plane_pose) + len(wx.edge_pose), wx.prior.n, "sharded over %d GPUs" % world if sharded else ("1 GPU" if world == 1 else "%d replicas" % world))}
        return leg, wx, profx, nx
    cfg3_leg = cfg4_leg = None
    if args.config == 2 and not args.no_cfg3:
        cfg3_leg, w3, prof3, n3 = window_leg(3)
        cfg4_leg, w4, prof4, n4 = window_leg(4)
    if rank == 0:
        out = {
            "metric": "sliding-window solve iterations/sec (10 KF, 1k feat, 30k LiDAR pts)",
            "value": tot_iters / max_el, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * max_el / args.steps, "higher_is_better": True, "scaling": ("strong" if sharded else "weak"), "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "repeats": {"regions": len(region_vals), "steps_per_region": args.steps, "value_is": "median region", "median": float(sorted(region_vals)[len(region_vals) // 2]),
                        "min": float(min(region_vals)), "max": float(max(region_vals)), "all": [float(v) for v in region_vals]},
            "config": {"workload": "BASELINE.json configs[%d]: K=%d keyframes, L=%d landmarks, %d visual factors, %d plane + %d edge LiDAR points, %d IMU, %d ICP, %d LPS, prior n=%d (%s)"
                       % (args.config - 1, w.K, w.L, len(w.vis_i), len(w.plane_pose), len(w.edge_pose), len(w.imu_i), len(w.icp_ids), len(w.lps_ids), w.prior.n, prior_kind),
                       "iterations_per_solve": last.iterations, "termination": abi.TERM_NAMES[last.termination], "final_cost": last.final_cost,
                       "parallelism": "1 GPU" if world == 1 else (("factor set of ONE window sharded over %d GPUs: visual by landmark owner, LiDAR points in contiguous slices, ONE " % world) + ("peer-buffer exchange (IPC inboxes over xGMI)" if args.ipc else "RCCL all-reduce") + " per iteration of the whole linear-system set ([S|g|cost] + the owners' landmark arrays), then the complete step kernel redundantly on every rank; NO N > 1 curve of this code had been measured on hardware before this run (one GPU per build box)" if sharded else "%d independent replicas" % world),
                       "step": "one full window solve, inputs resident in HBM"},
        }
        if shard_note:
            out["config"]["note"] = shard_note
        if replicas_leg:
            out["replicas"] = replicas_leg
            if phases and phases.get("per_rank"):
                # the first N > 1 record explains itself: what sharding ONE window can and cannot divide (Amdahl), from this very run's numbers
                pr = phases["per_rank"]
                sweep = max(r[0] for r in pr); gather = max(r[1] for r in pr); coll = max(r[2] for r in pr); step = max(r[3] for r in pr)
                one_gpu = replicas_leg["value"] / world
                out["sharding_explained"] = {
                    "one_gpu_it_per_s_in_this_run": one_gpu, "one_gpu_us_per_iteration": 1e6 / one_gpu,
                    "sharded_us_per_iteration_from_the_phases": sweep + gather + coll + step,
                    "divided_by_sharding_us": sweep, "not_divided_us": {"gather": gather, "collective": coll, "step (dense solve, redundant on every rank)": step},
                    "amdahl_bound_it_per_s_at_infinite_ranks": 1e6 / max(1e-9, gather + coll + step),
                    "note": "sharding one window divides the factor sweep only; gather, the one collective and the complete trust-region step repeat on every rank, and a sharded solve runs three launches per "
                            "iteration instead of the single-GPU structure's one resident launch per solve -- `replicas` (and bench.py --batch on one GPU) is the leg that scales"}
        if phases:
            out["phases_per_rank"] = phases
        if comm_info:
            out["communicator"] = comm_info
            if sharded:
                mbb, fbb = C.c_int64(0), C.c_int64(0)
                if be.lib.vil_comm_message_bytes(be.ctx, C.byref(mbb), C.byref(fbb)) == 0:
                    out["communicator"]["message_bytes_per_peer_rank0"] = int(mbb.value); out["communicator"]["full_set_bytes"] = int(fbb.value)
        if prof.sweep_launches > 0:
            out["roofline"] = roofline_obj(w, prof, "second pass of the same %d steps with HIP events enabled (%.1f ms/step instrumented vs %.1f ms/step in the value region)" % (args.steps, 1e3 * el_events / args.steps, 1e3 * max_el / args.steps),
                                           ("r06_pmc_fetch_size.csv", "r06_pmc_write_size.csv"), "profiles/r06_pmc_{fetch,write}_size.csv (separate rocprofv3 --pmc passes of this command)", "r06_pmc_mfma.csv",
                                           one_launch=one_launch, phases=phases_head, two_launch=two_launch, persistent=persistent)
            out["launches_per_iteration"] = int(lpi.value)
        if world == 1 and pcie_leg is not None:
            out["pcie_inclusive"] = pcie_leg
            out["pcie_inclusive_classic"] = pcie_classic
        if cfg3_leg:
            if prof3.sweep_launches > 0:
                cfg3_leg["roofline"] = roofline_obj(w3, prof3, "a further pass of the same %d steps with HIP events enabled" % n3, ("r06_pmc_fetch_size_c3.csv", "r06_pmc_write_size_c3.csv"),
                                                    "profiles/r06_pmc_{fetch,write}_size_c3.csv (rocprofv3 --pmc passes of bench.py --config 3)", "r06_pmc_mfma_c3.csv", one_launch=leg_one.get(3, False), phases=leg_phases.get(3))
            out["configs2_window"] = cfg3_leg
        if cfg4_leg:
            if prof4.sweep_launches > 0:
                cfg4_leg["roofline"] = roofline_obj(w4, prof4, "a further pass of the same %d steps with HIP events enabled" % n4, ("r06_pmc_fetch_size_c4.csv", "r06_pmc_write_size_c4.csv"),
                                                    "profiles/r06_pmc_{fetch,write}_size_c4.csv (rocprofv3 --pmc passes of bench.py --config 4)", "r06_pmc_mfma_c4.csv", one_launch=leg_one.get(4, False), phases=leg_phases.get(4))
            out["configs3_window"] = cfg4_leg
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(w, opts)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            out["speedup_vs_cpu_all_cores"] = out["value"] / out["cpu_baseline"]["all_cores"]["value"]
        emit(out)
    be.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
