"""include/vilwindow_shim.hpp: FeatureManager-style landmark indexing and the slideWindow() shift, checked against an
independent Python model of the same rules on a scripted sequence (g++ only, no GPU), and the batch re-integration of the
dirty IMU intervals through the oracle's implementation of include/vilpreint.h."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FEAT = r'''
#include <cstdio>
#include "vilwindow_shim.hpp"
extern "C" void vil_prior_capacity(int, int*, int*, int*) {}
extern "C" int vpre_integrate(vpre_ctx*, int32_t, const int32_t*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, double*, double*) { return -1; }
static unsigned long long s = 88172645463325252ull;
static double rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s % 1000000) / 1000000.0; }
int main() {
    const int W = 6;
    vil::FeatureTable ft(W, 5.0, 10.0 / 460.0);
    int frame_count = 0, next_id = 0;
    std::vector<int> alive;
    for (int img = 0; img < 30; ++img) {
        // tracks die at random, new ones are born; about 30 per image
        std::vector<int> keep; for (int id : alive) if (rnd() > 0.12) keep.push_back(id);
        alive = keep; while ((int)alive.size() < 30) alive.push_back(next_id++);
        std::vector<int> ids; std::vector<double> obs;
        for (int id : alive) { ids.push_back(id); const double shift = (img % 3 == 0 ? 0.05 : 0.002) * img; const double o[8] = {0.01 * (id % 17) + shift, 0.02 * (id % 11), 1.0, 300 + id, 200 + id % 50, 0.1, -0.1, (id % 4 == 0 && img % 2 == 0) ? 2.0 + 0.1 * (id % 7) : -1.0}; obs.insert(obs.end(), o, o + 8); }
        const bool margin_old = ft.add_frame(frame_count, ids.data(), obs.data(), (int)ids.size(), 0.001 * img);
        std::printf("F %d %d %d %d %d", img, frame_count, (int)margin_old, ft.last_track_num, ft.count());
        if (frame_count == W) {
            // "solve": read the depth vector, perturb it (some go negative), write it back
            std::vector<double> x(ft.count() + 1);
            ft.depth_vector(x.data());
            double sum = 0; for (int i = 0; i < ft.count(); ++i) { sum += x[i]; if ((i + img) % 13 == 0) x[i] = -x[i]; else x[i] *= 1.01; }
            ft.set_depth(x.data());
            ft.remove_failures();
            vil::WindowPacker pk(W + 1, ft.count()); ft.pack(pk, 240.0);
            const vil_problem* p = pk.finish();
            long chk = 0; for (int f = 0; f < p->n_vis; ++f) chk += (long)(p->vis_i[f] + 1) * 3 + (long)(p->vis_j[f] + 1) * 5 + (long)p->vis_l[f] * 7;
            int nconst = 0; for (int l = 0; l < p->L; ++l) nconst += p->lm_const[l];
            std::printf(" S %.9f %d %d %ld %d", sum, ft.count(), p->n_vis, chk, nconst);
            if (margin_old) {
                const double R0[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, P0[3] = {0, 0, 0}, R1[9] = {0.9998, -0.02, 0, 0.02, 0.9998, 0, 0, 0, 1}, P1[3] = {0.05, 0.01, 0.3 * (img % 5) - 0.2};
                ft.remove_back_shift_depth(R0, P0, R1, P1);
            } else ft.remove_front(frame_count);
            double dsum = 0; for (const vil::FeatureTrack& t : ft.tracks()) dsum += t.estimated_depth * (t.start_frame + 1) + (t.lidar_depth_flag ? 0.5 : 0.0) + 0.001 * t.obs.size();
            std::printf(" R %d %.9f", (int)ft.tracks().size(), dsum);
        } else ++frame_count;
        std::printf("\n");
    }
    return 0;
}
'''


class PyTable:
    """The same rules, written independently from the behaviour description (lists of dicts)."""

    def __init__(self, W, init_depth, min_parallax):
        self.W, self.init_depth, self.min_parallax, self.tr, self.last = W, init_depth, min_parallax, [], 0

    def inprob(self, t):
        return len(t["obs"]) >= 2 and t["start"] < self.W - 2

    def add(self, fc, ids, obs, td):
        self.last = 0
        by = {t["id"]: t for t in self.tr}
        for i, o in zip(ids, obs):
            o = dict(pt=o[:3], uv=o[3:5], vel=o[5:7], depth=o[7], td=td)
            t = by.get(i)
            if t is None:
                t = dict(id=i, start=fc, obs=[o], est=o["depth"] if o["depth"] > 0 else -1.0, lidar=o["depth"] > 0, flag=0)
                self.tr.append(t); by[i] = t
            else:
                t["obs"].append(o); self.last += 1
                if o["depth"] > 0 and not t["lidar"]:
                    t["est"], t["lidar"] = o["depth"], True; t["obs"][0]["depth"] = o["depth"]
        if fc < 2 or self.last < 20:
            return True
        ps = []
        for t in self.tr:
            if t["start"] <= fc - 2 and t["start"] + len(t["obs"]) - 1 >= fc - 1:
                a, b = t["obs"][fc - 2 - t["start"]]["pt"], t["obs"][fc - 1 - t["start"]]["pt"]
                ps.append(np.hypot(a[0] / a[2] - b[0], a[1] / a[2] - b[1]))
        return True if not ps else sum(ps) / len(ps) >= self.min_parallax

    def count(self):
        return sum(self.inprob(t) for t in self.tr)

    def depth_vector(self):
        return [1.0 / (t["est"] if t["est"] > 0 else self.init_depth) for t in self.tr if self.inprob(t)]

    def set_depth(self, x):
        it = iter(x)
        for t in self.tr:
            if self.inprob(t):
                t["est"] = 1.0 / next(it); t["flag"] = 2 if t["est"] < 0 else 1

    def pack(self):
        n_vis, chk, nconst, l = 0, 0, 0, -1
        for t in self.tr:
            if not self.inprob(t):
                continue
            l += 1; nconst += bool(t["lidar"])
            for m in range(1, len(t["obs"])):
                n_vis += 1; chk += (t["start"] + 1) * 3 + (t["start"] + m + 1) * 5 + l * 7
        return n_vis, chk, nconst

    def back(self, R0, P0, R1, P1):
        out = []
        for t in self.tr:
            if t["start"] != 0:
                t["start"] -= 1; out.append(t); continue
            f = t["obs"].pop(0)
            depth = f["depth"] if f["depth"] > 0 else (t["est"] if t["est"] > 0 else -1.0)
            if len(t["obs"]) < 2:
                continue
            w = R0 @ (np.array(f["pt"]) * depth) + P0
            dep = (R1.T @ (w - P1))[2]
            if t["obs"][0]["depth"] > 0:
                t["est"], t["lidar"] = t["obs"][0]["depth"], True
            elif dep > 0:
                t["est"], t["lidar"] = dep, False
            else:
                t["est"], t["lidar"] = self.init_depth, False
            out.append(t)
        self.tr = out

    def front(self, fc):
        out = []
        for t in self.tr:
            if t["start"] == fc:
                t["start"] -= 1
            elif t["start"] + len(t["obs"]) - 1 >= fc - 1:
                t["obs"].pop(self.W - 1 - t["start"])
            if t["obs"]:
                out.append(t)
        self.tr = out


def _build(prog, extra=()):
    d = tempfile.mkdtemp()
    src = os.path.join(d, "t.cpp")
    open(src, "w").write(prog)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", os.path.join(d, "t"), *extra])
    return os.path.join(d, "t")


def test_feature_table_matches_python_model():
    out = subprocess.check_output([_build(FEAT)]).decode().strip().split("\n")
    W = 6
    py = PyTable(W, 5.0, 10.0 / 460.0)
    s = 88172645463325252

    def rnd():
        nonlocal s
        s ^= (s << 13) & (2 ** 64 - 1); s ^= s >> 7; s ^= (s << 17) & (2 ** 64 - 1)
        return (s % 1000000) / 1000000.0
    fc, next_id, alive = 0, 0, []
    n_old = n_new = 0
    for img in range(30):
        alive = [i for i in alive if rnd() > 0.12]
        while len(alive) < 30:
            alive.append(next_id); next_id += 1
        shift = (0.05 if img % 3 == 0 else 0.002) * img
        obs = [[0.01 * (i % 17) + shift, 0.02 * (i % 11), 1.0, 300 + i, 200 + i % 50, 0.1, -0.1, (2.0 + 0.1 * (i % 7)) if (i % 4 == 0 and img % 2 == 0) else -1.0] for i in alive]
        mo = py.add(fc, alive, obs, 0.001 * img)
        tok = out[img].split()
        assert tok[:6] == ["F", str(img), str(fc), str(int(mo)), str(py.last), str(py.count())], (img, tok)
        if fc == W:
            x = py.depth_vector()
            assert abs(sum(x) - float(tok[7])) < 1e-8
            x = [(-v if (i + img) % 13 == 0 else v * 1.01) for i, v in enumerate(x)]
            py.set_depth(x)
            py.tr = [t for t in py.tr if t["flag"] != 2]
            n_vis, chk, nconst = py.pack()
            assert [int(tok[8]), int(tok[9]), int(tok[10]), int(tok[11])] == [py.count(), n_vis, chk, nconst]
            if mo:
                n_old += 1
                py.back(np.eye(3), np.zeros(3), np.array([[0.9998, -0.02, 0], [0.02, 0.9998, 0], [0, 0, 1.0]]), np.array([0.05, 0.01, 0.3 * (img % 5) - 0.2]))
            else:
                n_new += 1
                py.front(fc)
            dsum = sum(t["est"] * (t["start"] + 1) + (0.5 if t["lidar"] else 0.0) + 0.001 * len(t["obs"]) for t in py.tr)
            assert int(tok[13]) == len(py.tr) and abs(float(tok[14]) - dsum) < 1e-6
        else:
            fc += 1
    assert n_old >= 3 and n_new >= 3                    # both marginalisation branches were exercised


FRAMES = r'''
#include <cstdio>
#include "vilwindow_shim.hpp"
extern "C" void vil_prior_capacity(int, int*, int*, int*) {}
#ifdef USE_ORACLE
extern "C" int orc_vpre_integrate(vpre_ctx*, int32_t, const int32_t*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, double*, double*);
extern "C" int vpre_integrate(vpre_ctx* c, int32_t n, const int32_t* st, const double* dt, const double* a, const double* g, const double* a0, const double* g0, const double* ba, const double* bg, const double* nz, double* out, double* jac) {
    return orc_vpre_integrate(c, n, st, dt, a, g, a0, g0, ba, bg, nz, out, jac);      // CPU test: the oracle's implementation of the same ABI
}
static vpre_ctx* make_ctx() { return nullptr; }
#else
static vpre_ctx* make_ctx() { vpre_ctx* c = nullptr; return vpre_create(0, &c) == 0 ? c : nullptr; }      // libvilsolve.so: needs the GPU
#endif
static double wave(int i, int c) { return 0.3 * ((i * 37 + c * 11) % 23) / 23.0 - 0.1 + (c == 2 ? 9.8 : 0.0); }
int main() {
    const int K = 5; const double nz[4] = {0.08, 0.004, 0.00004, 2.0e-6};
    vpre_ctx* ctx = make_ctx();
    vil::WindowFrames w(K);
    int sample = 0;
    double a0[3] = {wave(0, 0), wave(0, 1), wave(0, 2)}, g0[3] = {0.01, -0.02, 0.03};
    for (int k = 0; k < K; ++k) {
        w.stamp[k] = 10.0 + 0.1 * k; w.pose[7 * k] = k; w.speedbias[9 * k + 3] = 0.01 * k; w.speedbias[9 * k + 6] = 0.001 * k;
        w.reset_interval(k, a0, g0);
        for (int q = 0; q < 4 + k; ++q, ++sample) { const double a[3] = {wave(sample, 0), wave(sample, 1), wave(sample, 2)}, g[3] = {0.01 * (sample % 5), -0.02, 0.005 * (sample % 3)}; w.push_sample(k, 0.005, a, g); a0[0] = a[0]; a0[1] = a[1]; a0[2] = a[2]; g0[0] = g[0]; g0[1] = g[1]; g0[2] = g[2]; }
    }
    if (w.integrate(ctx, nz) != 0) return 1;
    for (int k = 0; k < K; ++k) std::printf("A %d %.15g %.15g %d\n", k, w.record[287 * k + 16], w.record[287 * k + 2], (int)w.dirty[k]);
    // second-newest frame dropped: interval K-2 absorbs interval K-1
    const double merged_sum = w.record[287 * (K - 2) + 16] + w.record[287 * (K - 1) + 16];
    w.slide_new(a0, g0);
    std::printf("B %d %d %zu %.15g %.15g\n", (int)w.dirty[K - 2], (int)w.dirty[K - 1], w.dt[K - 1].size(), w.stamp[K - 2], w.pose[7 * (K - 2)]);
    if (w.integrate(ctx, nz) != 0) return 1;
    std::printf("C %.15g %.15g %.15g\n", w.record[287 * (K - 2) + 16], merged_sum, w.record[287 * (K - 1) + 16]);
    // direct integration of the concatenated stream of interval K-2 for comparison
    {
        vil::WindowFrames v(2);
        v.speedbias[9 + 3] = w.lin_ba[3 * (K - 2)]; v.speedbias[9 + 6] = w.lin_bg[3 * (K - 2)];
        v.reset_interval(1, &w.acc0[3 * (K - 2)], &w.gyr0[3 * (K - 2)]);
        for (size_t q = 0; q < w.dt[K - 2].size(); ++q) v.push_sample(1, w.dt[K - 2][q], &w.acc[K - 2][3 * q], &w.gyr[K - 2][3 * q]);
        v.dirty[0] = 0;
        if (v.integrate(ctx, nz) != 0) return 1;
        double d = 0; for (int q = 0; q < 287; ++q) d = std::fmax(d, std::fabs(v.record[287 + q] - w.record[287 * (K - 2) + q]));
        std::printf("D %.3g\n", d);
    }
    // oldest frame dropped
    const double s1 = w.record[287 * 1 + 16], p1 = w.pose[7 * 1], st_last = w.stamp[K - 1];
    w.slide_old(a0, g0);
    std::printf("E %.15g %.15g %.15g %.15g %.15g %.15g %zu %d\n", w.record[16], s1, w.pose[0], p1, w.stamp[K - 1], st_last, w.dt[K - 1].size(), (int)w.dirty[K - 1]);
    vil::WindowPacker pk(K, 0); w.pack(pk);
    std::printf("G %d %d %d\n", pk.finish()->n_imu, pk.finish()->imu_i[0], pk.finish()->imu_j[K - 2]);
    return 0;
}
'''


def _check_frames(out, tol):
    A = [l for l in out if l[0] == "A"]
    assert len(A) == 5 and all(int(l[4]) == 0 for l in A)
    for k, l in enumerate(A):
        assert abs(float(l[2]) - 0.005 * (4 + k)) < 1e-15                     # sum_dt of every interval
    B = [l for l in out if l[0] == "B"][0]
    assert B[1:4] == ["1", "1", "0"] and abs(float(B[4]) - 10.4) < 1e-12 and float(B[5]) == 4.0      # state of the newest frame moved down
    C = [l for l in out if l[0] == "C"][0]
    assert abs(float(C[1]) - float(C[2])) < 1e-15 and float(C[3]) == 0.0                # merged interval, fresh empty one
    assert float([l for l in out if l[0] == "D"][0][1]) <= tol                           # == integrating the concatenated stream
    E = [l for l in out if l[0] == "E"][0]
    assert float(E[1]) == float(E[2]) and float(E[3]) == float(E[4]) and float(E[5]) == float(E[6]) and E[7:] == ["0", "1"]
    assert [l for l in out if l[0] == "G"][0][1:] == ["4", "0", "4"]


def test_window_frames_slide_and_reintegrate():
    import oracle_lib
    oracle_lib.open_oracle()
    exe = _build(FRAMES, ["-DUSE_ORACLE", oracle_lib.ORACLE_SO, "-Wl,-rpath," + oracle_lib.ORACLE_DIR])
    _check_frames([l.split() for l in subprocess.check_output([exe]).decode().strip().split("\n")], 0.0)


@pytest.mark.gpu
def test_window_frames_reintegrate_on_device():
    """The same C++ program linked against libvilsolve.so: the dirty intervals are integrated by k_preint."""
    from mvil_fusion_amd import lib
    lib.load_vilsolve()
    d = os.path.dirname(lib.LIB_PATH)
    exe = _build(FRAMES, [lib.LIB_PATH, "-Wl,-rpath," + d])
    _check_frames([l.split() for l in subprocess.check_output([exe]).decode().strip().split("\n")], 0.0)


WINTAB = r'''
#include <cstdio>
#include "vilwindow_shim.hpp"
extern "C" void vil_prior_capacity(int, int*, int*, int*) {}
extern "C" int vpre_integrate(vpre_ctx*, int32_t, const int32_t*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, double*, double*) { return -1; }
int main() {
    const int W = 6;
    vil::FeatureTable ft(W, 5.0, 10.0 / 460.0);
    vil::TrackSlots slots(64);
    int bad = 0;
    for (int img = 0; img < 14; ++img) {
        const int fc = img < W ? img : W;
        std::vector<int> ids; std::vector<double> obs;
        for (int id = img / 2; id < img / 2 + 25; ++id) { ids.push_back(id); const double o[8] = {0.01 * id + 0.02 * img, 0.02 * (id % 7), 1.0, 300.0 + id, 200.0 + id % 9, 0.1, -0.1, id % 5 == 0 ? 3.0 : -1.0}; obs.insert(obs.end(), o, o + 8); }
        ft.add_frame(fc, ids.data(), obs.data(), (int)ids.size(), 0.001 * img);
        if (fc < W) continue;
        // the landmark table of the resident window, expanded to factors, is the factor list WindowPacker gets
        std::vector<int32_t> tr, st, no; std::vector<uint8_t> lc;
        ft.win_landmarks(slots, tr, st, no, lc);
        vil::WindowPacker pk(W + 1, ft.count()); ft.pack(pk, 240.0);
        const vil_problem* p = pk.finish();
        int f = 0;
        for (size_t l = 0; l < tr.size(); ++l) for (int q = 1; q < no[l]; ++q, ++f) bad += !(f < p->n_vis && p->vis_i[f] == st[l] && p->vis_j[f] == st[l] + q && p->vis_l[f] == (int)l);
        bad += f != p->n_vis; bad += (int)tr.size() != p->L;
        for (size_t l = 0; l < tr.size(); ++l) bad += lc[l] != p->lm_const[l];
        // the observations of a frame, by slot: what the factor constants are made of
        for (int k = 0; k <= W; ++k) {
            std::vector<int32_t> ot; std::vector<double> ob;
            ft.win_frame_obs(k, 240.0, slots, ot, ob);
            for (size_t q = 0; q < ot.size(); ++q) bad += ot[q] < 0 || ot[q] >= 64;
            for (size_t a = 0; a < ot.size(); ++a) for (size_t b2 = a + 1; b2 < ot.size(); ++b2) bad += ot[a] == ot[b2];      // one slot per track
        }
        {   // factor 0's constants from the store entries of its landmark
            std::vector<int32_t> ot; std::vector<double> ob;
            ft.win_frame_obs(st[0], 240.0, slots, ot, ob);
            size_t a = 0; while (a < ot.size() && ot[a] != tr[0]) ++a;
            bad += a == ot.size() || ob[8 * a] != p->vis_const[0] || ob[8 * a + 3] != p->vis_const[6] || ob[8 * a + 5] != p->vis_const[10] || ob[8 * a + 6] != p->vis_const[12];
        }
        const double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, P0[3] = {0, 0, 0};
        if (img % 3) ft.remove_back_shift_depth(R, P0, R, P0); else ft.remove_front(fc);
        slots.retain(ft.tracks());
    }
    std::printf("WINTAB %d\\n", bad);
    return 0;
}
'''


def test_resident_window_tables_of_the_shim():
    """FeatureTable::win_landmarks / win_frame_obs + TrackSlots (the host side of vil_win_*): the landmark table expands to exactly the factor
    list WindowPacker receives, slots are unique per track and recycled after a slide."""
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "w.cpp"); open(src, "w").write(WINTAB)
        exe = os.path.join(d, "w")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        out = subprocess.check_output([exe], text=True)
    assert "WINTAB 0" in out, out


TRI = r'''
#include <cstdio>
#include "vilwindow_shim.hpp"
extern "C" void vil_prior_capacity(int, int*, int*, int*) {}
extern "C" int vpre_integrate(vpre_ctx*, int32_t, const int32_t*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, const double*, double*, double*) { return -1; }
int main(int argc, char** argv) {
    // stdin: W n_tracks, poses (W + 1) x 7, ex 7, then per track: start nobs, nobs x [x y z]
    int W, n; if (std::scanf("%d %d", &W, &n) != 2) return 1;
    std::vector<double> pose(7 * (W + 1)), ex(7);
    for (double& v : pose) if (std::scanf("%lf", &v) != 1) return 1;
    for (double& v : ex) if (std::scanf("%lf", &v) != 1) return 1;
    std::vector<int> start(n), nobs(n); std::vector<std::vector<double>> pts(n);
    for (int t = 0; t < n; ++t) { if (std::scanf("%d %d", &start[t], &nobs[t]) != 2) return 1; pts[t].resize(3 * nobs[t]); for (double& v : pts[t]) if (std::scanf("%lf", &v) != 1) return 1; }
    vil::FeatureTable ft(W, 5.0, 10.0 / 460.0);
    for (int fc = 0; fc <= W; ++fc) {
        std::vector<int> ids; std::vector<double> obs;
        for (int t = 0; t < n; ++t) { const int q = fc - start[t]; if (q < 0 || q >= nobs[t]) continue; ids.push_back(t); const double o[8] = {pts[t][3 * q], pts[t][3 * q + 1], pts[t][3 * q + 2], 0, 0, 0, 0, -1.0}; obs.insert(obs.end(), o, o + 8); }
        ft.add_frame(fc, ids.data(), obs.data(), (int)ids.size(), 0.0);
    }
    ft.triangulate(pose.data(), ex.data());
    for (const vil::FeatureTrack& t : ft.tracks()) std::printf("%d %.17g\n", t.feature_id, t.estimated_depth);
    return 0;
}
'''


def test_triangulate_matches_svd_including_low_parallax_tracks():
    """FeatureTable::triangulate (feature_manager.cpp:214-273) against numpy's SVD of the same stacked rows: ordinary tracks, tracks whose
    baseline is 1e-7 of their depth (sigma_min / sigma_max ~ 1e-8: an eigen-decomposition of A^T A loses the null direction there, the one-sided
    Jacobi SVD of the shim must not) and points behind the anchor camera (negative depth -> INIT_DEPTH)."""
    rng = np.random.default_rng(5)
    W, INIT = 8, 5.0
    def q2R(q):
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    ex = np.concatenate([[0.05, -0.02, 0.1], [0.01, -0.02, 0.015, 1.0]]); ex[3:] /= np.linalg.norm(ex[3:])
    tracks = []
    for kind, scale in (("wide", 0.3), ("narrow", 3e-7), ("behind", 0.3)):
        poses = np.zeros((W + 1, 7))
        for k in range(W + 1):
            q = np.concatenate([0.02 * rng.standard_normal(3) * (scale > 1e-3), [1.0]]); q /= np.linalg.norm(q)
            poses[k] = np.concatenate([scale * np.array([k, 0.3 * np.sin(k), 0.1 * k]) + 0.0, q])
        tracks.append((kind, poses))
    # one table per pose set (the poses are per call): run the harness three times
    for kind, poses in tracks:
        Ric, tic = q2R(ex[3:]), ex[:3]
        cam = [(q2R(p[3:]) @ Ric, p[:3] + q2R(p[3:]) @ tic) for p in poses]
        items, lines = [], ["%d %d" % (W, 24)]
        lines.append(" ".join("%.17g" % v for v in poses.ravel())); lines.append(" ".join("%.17g" % v for v in ex))
        for t in range(24):
            start = int(rng.integers(0, W - 3)); nobs = int(rng.integers(2, W + 1 - start + 1))
            R0, t0 = cam[start]
            depth = rng.uniform(2.0, 12.0) * (-1.0 if kind == "behind" and t % 2 else 1.0)
            Xw = R0 @ (np.array([rng.uniform(-0.4, 0.4), rng.uniform(-0.3, 0.3), 1.0]) * depth) + t0
            obs = []
            for m in range(nobs):
                R1, t1 = cam[start + m]
                pc = R1.T @ (Xw - t1)
                pc = pc / pc[2] + np.array([1e-9 * rng.standard_normal(), 1e-9 * rng.standard_normal(), 0.0]) * (kind != "narrow")
                obs.append(pc)
            items.append((start, obs))
            lines.append("%d %d " % (start, nobs) + " ".join("%.17g" % v for o in obs for v in o))
        exe = _build(TRI)
        out = subprocess.run([exe], input="\n".join(lines), capture_output=True, text=True, check=True).stdout.strip().split("\n")
        got = {int(l.split()[0]): float(l.split()[1]) for l in out}
        n_checked = n_init = 0
        for t, (start, obs) in enumerate(items):
            if not (len(obs) >= 2 and start < W - 2):
                assert got[t] == -1.0; continue
            R0, t0 = cam[start]; rows = []
            for m, pt in enumerate(obs):
                R1, t1 = cam[start + m]
                R = R0.T @ R1; tt = R0.T @ (t1 - t0)
                P = np.hstack([R.T, (-R.T @ tt)[:, None]]); f = pt / np.linalg.norm(pt)
                rows.append(f[0] * P[2] - f[2] * P[0]); rows.append(f[1] * P[2] - f[2] * P[1])
            sv = np.linalg.svd(np.array(rows))
            V = sv[2][-1]; d = V[2] / V[3]
            want = INIT if d < 0 else d
            n_init += want == INIT; n_checked += 1
            # the null direction is determined to ~eps * sigma_max / (sigma_3 - sigma_4); the depth quotient inherits that
            tol = 1e-9 if kind != "narrow" else 64 * 2.2e-16 * sv[1][0] / max(sv[1][2] - sv[1][3], 1e-300) * max(1.0, abs(d))
            assert abs(got[t] - want) <= tol * max(1.0, abs(want)), (kind, t, got[t], want, sv[1])
        assert n_checked >= 10
        if kind == "behind": assert n_init >= 3
