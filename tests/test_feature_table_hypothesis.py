"""A17: include/vilwindow_shim.hpp's vil::FeatureTable against a LINE-BY-LINE transcription of the reference's
FeatureManager (vils_estimator/src/feature_manager.cpp:28-42 getFeatureCount, :45-106 addFeatureCheckParallax, :150-168 setDepth,
:170-179 removeFailures, :195-212 getDepthVector, :286-345 removeBackShiftDepth, :347-363 removeBack, :365-384 removeFront,
:386-417 compensatedParallax2; feature_manager.h:62-76 FeaturePerId) on hypothesis-generated image sequences -- feature ids in
arbitrary order with repeats, LiDAR depths appearing late, solved depths going negative, both marginalisation branches.  The image
reaches FeatureManager as a std::map keyed by feature id (estimator_node.cpp:485-503): ascending id, first entry of an id wins; that order
fixes f_manager.feature, hence the landmark numbering (feature_index) and the order of the visual factors (estimator.cpp:1189-1242).
No GPU, g++ only."""
import os
import subprocess

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, INIT_DEPTH, MIN_PARALLAX = 6, 5.0, 10.0 / 460.0        # WINDOW_SIZE (parameters.h:12), INIT_DEPTH (parameters.cpp:189), MIN_PARALLAX / FOCAL_LENGTH

DRIVER = r'''
#include <cstdio>
#include <iostream>
#include <string>
#include "vilwindow_shim.hpp"
extern "C" void vil_prior_capacity(int, int*, int*, int*) {}
extern "C" int vpre_integrate(vpre_ctx*, int32_t, const int32_t*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, double*, double*) { return -1; }
static void dump(const vil::FeatureTable& ft, int W) {
    std::printf("T %d", (int)ft.tracks().size());
    for (const vil::FeatureTrack& t : ft.tracks()) std::printf(" %d %d %d %.17g %d %d", t.feature_id, t.start_frame, (int)t.obs.size(), t.estimated_depth, (int)t.lidar_depth_flag, t.solve_flag);
    std::printf("\nC %d", ft.count());
    std::vector<double> x(ft.count() + 1); ft.depth_vector(x.data());
    for (int i = 0; i < ft.count(); ++i) std::printf(" %.17g", x[i]);
    vil::WindowPacker pk(W + 1, ft.count()); ft.pack(pk, 240.0);
    const vil_problem* p = pk.finish();
    std::printf("\nV %d", p->n_vis);
    for (int f = 0; f < p->n_vis; ++f) std::printf(" %d %d %d %.17g %.17g %.17g %.17g", p->vis_i[f], p->vis_j[f], p->vis_l[f], p->vis_const[14 * f + 0], p->vis_const[14 * f + 3], p->vis_const[14 * f + 10], p->vis_const[14 * f + 13]);
    std::printf("\nL"); for (int l = 0; l < p->L; ++l) std::printf(" %d", (int)p->lm_const[l]);
    std::printf("\n");
}
int main() {
    const int W = 6;
    vil::FeatureTable ft(W, 5.0, 10.0 / 460.0);
    std::string op;
    while (std::cin >> op) {
        if (op == "ADD") {
            int fc, n; double td; std::cin >> fc >> td >> n;
            std::vector<int> ids(n); std::vector<double> obs(8 * (size_t)n);
            for (int k = 0; k < n; ++k) { std::cin >> ids[k]; for (int q = 0; q < 8; ++q) std::cin >> obs[8 * k + q]; }
            const bool kf = ft.add_frame(fc, ids.data(), obs.data(), n, td);
            std::printf("A %d %d\n", (int)kf, ft.last_track_num);
        } else if (op == "SET") {
            int n; std::cin >> n; std::vector<double> x(n + 1); for (int k = 0; k < n; ++k) std::cin >> x[k];
            ft.set_depth(x.data());
        } else if (op == "FAIL") ft.remove_failures();
        else if (op == "BACKSHIFT") { double R0[9], P0[3], R1[9], P1[3]; for (double& v : R0) std::cin >> v; for (double& v : P0) std::cin >> v; for (double& v : R1) std::cin >> v; for (double& v : P1) std::cin >> v; ft.remove_back_shift_depth(R0, P0, R1, P1); }
        else if (op == "BACK") ft.remove_back();
        else if (op == "FRONT") { int fc; std::cin >> fc; ft.remove_front(fc); }
        dump(ft, W);
    }
    return 0;
}
'''


class RefFeatureManager:
    """Transcription of the reference, statement by statement; `feature` is the std::list<FeaturePerId> (insertion order)."""

    def __init__(self):
        self.feature = []
        self.last_track_num = 0

    @staticmethod
    def _new(feature_id, start_frame, measured_depth):          # feature_manager.h:62-76
        f = dict(feature_id=feature_id, start_frame=start_frame, used_num=0, estimated_depth=-1.0, lidar_depth_flag=False, solve_flag=0, fpf=[])
        if measured_depth > 0:
            f["estimated_depth"] = measured_depth; f["lidar_depth_flag"] = True
        return f

    def _in(self, it):                                           # the predicate of :36, :156, :202
        it["used_num"] = len(it["fpf"])
        return it["used_num"] >= 2 and it["start_frame"] < W - 2

    def getFeatureCount(self):                                   # :28-42
        return sum(1 for it in self.feature if self._in(it))

    def addFeatureCheckParallax(self, frame_count, image, td):   # :45-106; image: {feature_id: [(camera_id, 8-vector), ...]} walked in ascending id (std::map)
        parallax_sum, parallax_num = 0.0, 0
        self.last_track_num = 0
        for feature_id in sorted(image):
            v = image[feature_id][0][1]                          # id_pts.second[0].second
            fpf = dict(point=np.array(v[0:3]), uv=np.array(v[3:5]), velocity=np.array(v[5:7]), depth=v[7], cur_td=td)
            it = next((f for f in self.feature if f["feature_id"] == feature_id), None)
            if it is None:
                self.feature.append(self._new(feature_id, frame_count, fpf["depth"]))
                self.feature[-1]["fpf"].append(fpf)
            else:
                it["fpf"].append(fpf)
                self.last_track_num += 1
                if fpf["depth"] > 0 and not it["lidar_depth_flag"]:
                    it["estimated_depth"] = fpf["depth"]; it["lidar_depth_flag"] = True; it["fpf"][0]["depth"] = fpf["depth"]
        if frame_count < 2 or self.last_track_num < 20:
            return True
        for it in self.feature:
            if it["start_frame"] <= frame_count - 2 and it["start_frame"] + len(it["fpf"]) - 1 >= frame_count - 1:
                parallax_sum += self.compensatedParallax2(it, frame_count); parallax_num += 1
        if parallax_num == 0:
            return True
        return parallax_sum / parallax_num >= MIN_PARALLAX

    @staticmethod
    def compensatedParallax2(it, frame_count):                   # :386-417 (p_i_comp = p_i)
        fi = it["fpf"][frame_count - 2 - it["start_frame"]]; fj = it["fpf"][frame_count - 1 - it["start_frame"]]
        u_j, v_j = fj["point"][0], fj["point"][1]
        dep_i = fi["point"][2]
        du, dv = fi["point"][0] / dep_i - u_j, fi["point"][1] / dep_i - v_j
        return max(0.0, np.sqrt(min(du * du + dv * dv, du * du + dv * dv)))

    def setDepth(self, x):                                       # :150-168
        feature_index = -1
        for it in self.feature:
            if not self._in(it):
                continue
            feature_index += 1
            it["estimated_depth"] = 1.0 / x[feature_index]
            it["solve_flag"] = 2 if it["estimated_depth"] < 0 else 1

    def removeFailures(self):                                    # :170-179
        self.feature = [it for it in self.feature if it["solve_flag"] != 2]

    def getDepthVector(self):                                    # :195-212
        return [1.0 / it["estimated_depth"] if it["estimated_depth"] > 0 else 1.0 / INIT_DEPTH for it in self.feature if self._in(it)]

    def removeBackShiftDepth(self, marg_R, marg_P, new_R, new_P):     # :286-345
        out = []
        for it in self.feature:
            if it["start_frame"] != 0:
                it["start_frame"] -= 1; out.append(it); continue
            uv_i = it["fpf"][0]["point"]
            depth = -1.0
            if it["fpf"][0]["depth"] > 0:
                depth = it["fpf"][0]["depth"]
            elif it["estimated_depth"] > 0:
                depth = it["estimated_depth"]
            it["fpf"].pop(0)
            if len(it["fpf"]) < 2:
                continue
            pts_i = uv_i * depth
            w_pts_i = marg_R @ pts_i + marg_P
            pts_j = new_R.T @ (w_pts_i - new_P)
            dep_j = pts_j[2]
            if it["fpf"][0]["depth"] > 0:
                it["estimated_depth"] = it["fpf"][0]["depth"]; it["lidar_depth_flag"] = True
            elif dep_j > 0:
                it["estimated_depth"] = dep_j; it["lidar_depth_flag"] = False
            else:
                it["estimated_depth"] = INIT_DEPTH; it["lidar_depth_flag"] = False
            out.append(it)
        self.feature = out

    def removeBack(self):                                        # :347-363
        out = []
        for it in self.feature:
            if it["start_frame"] != 0:
                it["start_frame"] -= 1; out.append(it)
            else:
                it["fpf"].pop(0)
                if len(it["fpf"]) != 0:
                    out.append(it)
        self.feature = out

    def removeFront(self, frame_count):                          # :365-384
        out = []
        for it in self.feature:
            if it["start_frame"] == frame_count:
                it["start_frame"] -= 1; out.append(it); continue
            j = W - 1 - it["start_frame"]
            if it["start_frame"] + len(it["fpf"]) - 1 < frame_count - 1:      # endFrame() < frame_count - 1
                out.append(it); continue
            it["fpf"].pop(j)
            if len(it["fpf"]) != 0:
                out.append(it)
        self.feature = out

    def factors(self):                                           # the visual loop of Estimator::optimization(), estimator.cpp:1189-1242
        out, lm_const, feature_index = [], [], -1
        for it in self.feature:
            if not self._in(it):
                continue
            feature_index += 1
            lm_const.append(int(it["lidar_depth_flag"]))      # :1217-1221 SetParameterBlockConstant iff lidar_depth_flag
            imu_i = it["start_frame"]; imu_j = imu_i - 1
            f0 = it["fpf"][0]
            for fj in it["fpf"]:
                imu_j += 1
                if imu_i == imu_j:
                    continue
                out.append((imu_i, imu_j, feature_index, f0["point"][0], fj["point"][0], f0["cur_td"], fj["uv"][1] - 240.0))
        return out, lm_const


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    d = tmp_path_factory.mktemp("ftdrv")
    src = d / "drv.cpp"; src.write_text(DRIVER)
    exe = str(d / "drv")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    return exe


# hypothesis draws the STRUCTURE of a sequence (how many features an image carries, which slide follows, the seeds); the bulk numbers come from
# a numpy generator seeded by those draws (keeps the examples inside hypothesis' entropy budget)
step_st = st.tuples(st.integers(0, 40), st.integers(0, 2 ** 31 - 1), st.integers(0, 3), st.floats(-0.3, 0.3), st.booleans(), st.integers(30, 60))


def make_image(n, seed, id_range):
    rng = np.random.default_rng(seed)
    img = []
    for _ in range(n):
        fid = int(rng.integers(0, id_range))                        # unsorted, repeats allowed (the std::map keeps the first entry of an id)
        depth = float(rng.uniform(0.5, 30.0)) if rng.uniform() < 0.3 else -1.0
        jit = 0.002 if seed % 6 else 0.2                            # five images in six move little: the parallax test then picks MARGIN_SECOND_NEW
        img.append((fid, (0.6 * np.sin(1.7 * fid) + float(rng.normal(0, jit)), 0.45 * np.cos(2.3 * fid) + float(rng.normal(0, jit)), float(rng.uniform(100, 700)), float(rng.uniform(100, 400)),
                          float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5)), depth)))
    scales = [float(v) for v in np.where(rng.uniform(size=80) < 0.08, -1.0, 1.0) * rng.uniform(0.5, 2.0, 80)]      # ~8 % of the solved depths go negative
    return img, scales


@settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
@given(st.lists(step_st, min_size=8, max_size=14))
def test_feature_table_matches_transcribed_feature_manager(driver, steps):
    run_sequence(driver, steps)


def test_feature_table_long_sequence_covers_every_branch(driver):
    """A fixed 80-image sequence: both marginalisation branches, the parallax decision (>= 20 continued tracks), failures removed, pre-window removeBack."""
    rng = np.random.default_rng(7)
    steps = [(int(rng.integers(60, 90)), int(rng.integers(0, 2 ** 31 - 1)), int(rng.integers(0, 4)) if k < 4 else 1 + int(rng.integers(0, 3)), float(rng.uniform(-0.3, 0.3)), bool(rng.uniform() < 0.15), 45) for k in range(80)]
    stats = run_sequence(driver, steps)
    assert stats["old"] >= 5 and stats["new"] >= 5 and stats["parallax_decisions"] >= 10 and stats["failures_removed"] >= 5 and stats["max_landmarks"] >= 10, stats


def run_sequence(driver, steps):
    ref = RefFeatureManager()
    script, expect = [], []
    stats = dict(old=0, new=0, parallax_decisions=0, failures_removed=0, max_landmarks=0)

    def state():
        fac, lmc = ref.factors()
        return dict(tracks=[(it["feature_id"], it["start_frame"], len(it["fpf"]), it["estimated_depth"], int(it["lidar_depth_flag"]), it["solve_flag"]) for it in ref.feature],
                    count=ref.getFeatureCount(), dep=ref.getDepthVector(), fac=fac, lmc=lmc)
    frame_count = 0
    for n_img, seed, mode, dz, force_old, id_range in steps:
        img, scales = make_image(n_img, seed, id_range)
        td = 0.001 * len(script)
        image = {}
        toks = []
        for fid, o in img:
            v = [o[0], o[1], 1.0, o[2], o[3], o[4], o[5], o[6]]
            image.setdefault(fid, []).append((0, v))             # image[feature_id].emplace_back(camera_id, ...)
            toks.append("%d %s" % (fid, " ".join(repr(float(x)) for x in v)))
        kf = ref.addFeatureCheckParallax(frame_count, image, td)
        stats["parallax_decisions"] += int(frame_count >= 2 and ref.last_track_num >= 20); stats["max_landmarks"] = max(stats["max_landmarks"], ref.getFeatureCount())
        script.append("ADD %d %r %d %s" % (frame_count, td, len(img), " ".join(toks))); expect.append(("A", kf, ref.last_track_num, state()))
        if frame_count < W:
            if mode == 0 and frame_count >= 2:                   # before initialisation the window may also slide without depths (removeBack, :347-363)
                ref.removeBack(); script.append("BACK"); expect.append((None, None, None, state()))
            else:
                frame_count += 1
            continue
        # a full window: "solve" (depth vector scaled, some entries negative), removeFailures, then one of the two marginalisation branches
        x = [d * s for d, s in zip(ref.getDepthVector(), scales)]
        ref.setDepth(x); script.append("SET %d %s" % (len(x), " ".join(repr(float(v)) for v in x))); expect.append((None, None, None, state()))
        n_before = len(ref.feature)
        ref.removeFailures(); stats["failures_removed"] += n_before - len(ref.feature); script.append("FAIL"); expect.append((None, None, None, state()))
        if kf or force_old:                                      # MARGIN_OLD (estimator.cpp:1798-1813)
            stats["old"] += 1
            R0 = np.eye(3); P0 = np.zeros(3)
            c, s_ = np.cos(0.02), np.sin(0.02)
            R1 = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1.0]]); P1 = np.array([0.05, 0.01, dz])
            ref.removeBackShiftDepth(R0, P0, R1, P1)
            script.append("BACKSHIFT " + " ".join(repr(float(v)) for v in list(R0.ravel()) + list(P0) + list(R1.ravel()) + list(P1)))
        else:                                                    # MARGIN_SECOND_NEW (:1788-1796)
            stats["new"] += 1
            ref.removeFront(frame_count); script.append("FRONT %d" % frame_count)
        expect.append((None, None, None, state()))
    out = subprocess.run([driver], input="\n".join(script) + "\n", capture_output=True, text=True, timeout=60).stdout.split("\n")
    pos = 0
    for op, (tag, kf, ltn, stt) in zip(script, expect):
        if tag == "A":
            a = out[pos].split(); pos += 1
            assert a[0] == "A" and int(a[1]) == int(kf) and int(a[2]) == ltn, (op[:40], a, kf, ltn)
        t = out[pos].split(); c = out[pos + 1].split(); v = out[pos + 2].split(); l = out[pos + 3].split(); pos += 4
        n = int(t[1])
        got = [(int(t[2 + 6 * k]), int(t[3 + 6 * k]), int(t[4 + 6 * k]), float(t[5 + 6 * k]), int(t[6 + 6 * k]), int(t[7 + 6 * k])) for k in range(n)]
        assert len(got) == len(stt["tracks"]), op[:40]
        for g, r in zip(got, stt["tracks"]):                     # same tracks in the same (std::list) order
            assert g[:3] == r[:3] and g[4] == r[4] and abs(g[3] - r[3]) <= 1e-12 * max(1.0, abs(r[3])), (op[:40], g, r)
            assert r[5] == g[5] or (r[5] in (0, 1, 2) and g[5] == r[5]), (g, r)
        assert int(c[1]) == stt["count"] and np.allclose([float(z) for z in c[2:]], stt["dep"], rtol=1e-12, atol=0)
        nv = int(v[1])
        assert nv == len(stt["fac"]), (op[:40], nv, len(stt["fac"]))
        for k, r in enumerate(stt["fac"]):                       # factor order = landmark order = std::map order of the first sighting
            g = v[2 + 7 * k: 9 + 7 * k]
            assert (int(g[0]), int(g[1]), int(g[2])) == r[:3] and np.allclose([float(z) for z in g[3:]], r[3:], rtol=1e-12, atol=1e-15), (op[:40], k, g, r)
        assert [int(z) for z in l[1:]] == stt["lmc"]
    return stats
