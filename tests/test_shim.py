"""The C++ packing shim (include/vilsolve_shim.hpp) compiles with g++ against the C header and produces a
vil_problem whose tables match what was added -- no GPU needed."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = r'''
#include <cstdio>
#include "vilsolve_shim.hpp"
extern "C" void vil_prior_capacity(int K, int* n_max, int* nblk_max, int* x0_max) { *n_max = 6 * K + 16; *nblk_max = K + 4; *x0_max = 7 * K + 33; }
int main() {
    vil::WindowPacker pk(7, 3);
    const double g[3] = {0, 0, 9.8};
    pk.set_constants(g, 460.0, 0.0, true, true);
    double rec[VIL_IMU_CONST] = {0}; rec[16] = 0.1;
    pk.add_imu(0, 1, rec);
    const double pi[3] = {0.1, 0.2, 1}, pj[3] = {0.11, 0.21, 1}, v[2] = {0.01, 0.02};
    pk.add_visual(0, 1, 0, pi, pj, v, v, 0, 0, 5, 6, false);
    pk.add_visual(0, 2, 0, pi, pj, v, v, 0, 0, 5, 6, false);
    pk.add_visual(1, 3, 2, pi, pj, v, v, 0, 0, 5, 6, true);
    const double cp[3] = {1, 2, 3}, n[3] = {0, 0, 1};
    pk.add_plane(4, cp, n, -1.5);
    pk.freeze_frame(5);
    const vil_problem* p = pk.finish();
    vil::PriorStore ps(7);
    ps.out()->n = -1; ps.commit();
    std::printf("%d %d %d %d %d %d %d %.1f %d\n", p->K, p->L, p->n_imu, p->n_vis, p->n_plane, (int)p->lm_const[2], (int)p->pose_const[5], p->sqrt_info_px, ps.prior().n);
    return 0;
}
'''


def test_shim_compiles_and_packs():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        open(src, "w").write(PROG)
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    assert out == ["7", "3", "1", "3", "1", "1", "1", "230.0", "0"]


IDS = r"""
#include <cstdio>
#include "vilsolve_shim.hpp"
extern "C" void vil_prior_capacity(int, int*, int*, int*) {}
int main() {
    const double st[7] = {10.0, 10.1, 10.2, 10.3, 10.4, 10.5, 10.6};
    int a = -9, b = -9;
    bool ok;
    ok = vil::find_nearest_2id(st, 7, 10.25, a, b); std::printf("%d %d %d\n", (int)ok, a, b);      // bracketed
    ok = vil::find_nearest_2id(st, 7, 10.3, a, b);  std::printf("%d %d %d\n", (int)ok, a, b);      // equal to a stamp: that frame is id_b
    ok = vil::find_nearest_2id(st, 7, 9.9, a, b);   std::printf("%d %d %d\n", (int)ok, a, b);      // before the window
    ok = vil::find_nearest_2id(st, 7, 10.0, a, b);  std::printf("%d %d %d\n", (int)ok, a, b);      // equal to frame 0: no left neighbour
    ok = vil::find_nearest_2id(st, 7, 10.7, a, b);  std::printf("%d %d %d\n", (int)ok, a, b);      // after the window
    int ia = 0, ib = 0, ic = 0, id = 0;
    ok = vil::find_windows_id(st, 7, 10.1, 10.2, 10.4, 10.5, ia, ib, ic, id); std::printf("%d %d %d %d %d\n", (int)ok, ia, ib, ic, id);
    ia = ib = ic = id = 0;
    ok = vil::find_windows_id(st, 7, 10.1, 10.2, 10.2, 10.3, ia, ib, ic, id); std::printf("%d %d %d %d %d\n", (int)ok, ia, ib, ic, id);   // shared frame: first bracket shifts down
    ia = ib = ic = id = 0;
    ok = vil::find_windows_id(st, 7, 10.0, 10.1, 10.1, 10.2, ia, ib, ic, id); std::printf("%d %d %d %d %d\n", (int)ok, ia, ib, ic, id);   // shift would leave the window
    ia = ib = ic = id = 0;
    ok = vil::find_windows_id(st, 7, 9.5, 10.1, 10.4, 10.5, ia, ib, ic, id);  std::printf("%d\n", (int)ok);                               // starts before the window
    ia = ib = ic = id = 0;
    ok = vil::find_windows_id(st, 7, 10.0, 10.6, 10.6, 10.6, ia, ib, ic, id); std::printf("%d\n", (int)ok);                               // bracket wider than 0.5 s
    ia = ib = ic = id = 0;
    ok = vil::find_windows_id(st, 7, 10.1, 10.25, 10.4, 10.5, ia, ib, ic, id); std::printf("%d %d %d %d %d\n", (int)ok, ia, ib, ic, id); // tb not a frame stamp: id_b keeps 0
    return 0;
}
"""


def test_window_id_lookup():
    """A13: FindNearest2ID / FindWindowsID semantics (lidar_backend.cpp:3-93), incl. the untouched-id quirk."""
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        open(src, "w").write(IDS)
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().strip().split("\n")
    assert out == ["1 2 3", "1 2 3", "0 -1 0", "0 -1 0", "0 6 7",
                   "1 1 2 4 5", "1 0 1 2 3", "0 -1 0 1 2", "0", "0", "0 1 0 4 5"]
