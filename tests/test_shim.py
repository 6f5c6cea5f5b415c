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
