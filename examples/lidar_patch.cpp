// lidar_patch.cpp -- the LiDAR-side patches of INTEGRATION.md (sections 7 and 8), compiled: scan-to-scan registration as
// Estimator::processLidar uses fast_gicp::FastVGICP (estimator.cpp:269-300), and the scan-to-map loop of lidar_mapping
// (localMapping.cpp:590-791), both through the C-ABI of this library on a synthetic room.
//   g++ -std=c++17 -Iinclude examples/lidar_patch.cpp mvil-fusion_amd/csrc/libvilsolve.so -o lidar_patch && ./lidar_patch
#include <cmath>
#include <cstdio>
#include <vector>

#include "vilmap.h"
#include "vilvgicp.h"

namespace {

unsigned long long rng_state = 88172645463325252ull;
double rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (double)(rng_state % 1000003) / 1000003.0; }
double gauss() { double s = 0; for (int i = 0; i < 12; ++i) s += rnd(); return s - 6.0; }

// points on the walls / floor / ceiling of a 20 x 20 x 5 m room (surf) and along its edges (corner), [x y z intensity]
void make_room(int n_surf, int n_corner, std::vector<float>& surf, std::vector<float>& corner) {
    const double lo[3] = {-10, -10, -1.5}, hi[3] = {10, 10, 3.5};
    for (int i = 0; i < n_surf; ++i) {
        double p[3]; for (int a = 0; a < 3; ++a) p[a] = lo[a] + (hi[a] - lo[a]) * rnd();
        const int ax = (int)(3 * rnd()) % 3, side = rnd() < 0.5;
        p[ax] = side ? hi[ax] : lo[ax];
        for (int a = 0; a < 3; ++a) surf.push_back((float)(p[a] + 0.01 * gauss()));
        surf.push_back((float)(10.0 * (2 * ax + side) + 5 * rnd()));
    }
    for (int i = 0; i < n_corner; ++i) {
        double p[3]; const int ax = (int)(3 * rnd()) % 3; int code = 0;
        for (int a = 0; a < 3; ++a) { if (a == ax) p[a] = lo[a] + (hi[a] - lo[a]) * rnd(); else { const int s = rnd() < 0.5; p[a] = s ? hi[a] : lo[a]; code |= s << a; } }
        for (int a = 0; a < 3; ++a) corner.push_back((float)(p[a] + 0.01 * gauss()));
        corner.push_back((float)(10.0 * ax + code));
    }
}
// every `step`-th map point seen from pose (yaw, t): sensor frame
void make_scan(const std::vector<float>& map, int step, double yaw, const double t[3], std::vector<float>& scan) {
    const double c = std::cos(yaw), s = std::sin(yaw);
    for (size_t i = 0; i < map.size() / 4; i += step) {
        const double d[3] = {map[4 * i] + 0.03 * gauss() - t[0], map[4 * i + 1] + 0.03 * gauss() - t[1], map[4 * i + 2] + 0.03 * gauss() - t[2]};
        scan.push_back((float)(c * d[0] + s * d[1])); scan.push_back((float)(-s * d[0] + c * d[1])); scan.push_back((float)d[2]); scan.push_back(map[4 * i + 3]);
    }
}

}  // namespace

int main() {
    std::vector<float> surf, corner; make_room(30000, 4000, surf, corner);
    const double t_true[3] = {0.6, -0.4, 0.1}, yaw_true = 0.05;
    // ---- scan-to-scan VGICP (estimator.cpp:269-300) ------------------------------------------------
    std::vector<float> txyz, sxyz;
    for (size_t i = 0; i < surf.size() / 4; i += 3) { txyz.push_back(surf[4 * i]); txyz.push_back(surf[4 * i + 1]); txyz.push_back(surf[4 * i + 2]); }
    { std::vector<float> s4; make_scan(surf, 3, yaw_true, t_true, s4); for (size_t i = 0; i < s4.size() / 4; ++i) { sxyz.push_back(s4[4 * i]); sxyz.push_back(s4[4 * i + 1]); sxyz.push_back(s4[4 * i + 2]); } }
    vgicp_ctx* vg = nullptr;
    if (vgicp_create(0, &vg) != 0) { std::fprintf(stderr, "vgicp_create failed (no HIP device: there is no CPU path)\n"); return 3; }
    vgicp_set_target(vg, (int)(txyz.size() / 3), txyz.data(), nullptr, 0.5);      // gicp->setResolution(0.5); setInputTarget (covariances estimated on the device)
    vgicp_set_source(vg, (int)(sxyz.size() / 3), sxyz.data(), nullptr);           // setInputSource
    vgicp_options vo; vgicp_default_options(&vo);
    double guess[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, T[16];
    vgicp_summary vs;
    const int rc1 = vgicp_align(vg, guess, &vo, T, &vs);                           // gicp->align(*aligned, guess); getFinalTransformation()
    std::printf("VGICP %d %d %d %.6f %.6f %.6f %.6f\n", rc1, vs.converged, vs.iterations, T[3], T[7], T[11], std::atan2(T[4], T[0]));
    vgicp_destroy(vg);
    // ---- scan-to-map (localMapping.cpp:590-791) ------------------------------------------------------
    vmap_ctx* vm = nullptr; vil_ctx* vil = nullptr;
    vil_device_cfg cfg = {0, 0, 1, 0};
    if (vmap_create(0, &vm) != 0 || vil_create(&cfg, &vil) != 0) return 3;
    vmap_set_map(vm, (int)(corner.size() / 4), corner.data(), (int)(surf.size() / 4), surf.data());   // kdtree*FromMap->setInputCloud
    std::vector<float> sc, ss; make_scan(corner, 7, yaw_true, t_true, sc); make_scan(surf, 7, yaw_true, t_true, ss);
    vil_options o; vil_default_options(&o); o.max_iterations = 4;
    double q[4] = {0, 0, std::sin(0.5 * (yaw_true + 0.01)), std::cos(0.5 * (yaw_true + 0.01))}, t[3] = {t_true[0] + 0.05, t_true[1] - 0.04, t_true[2] + 0.03};   // q_w_curr / t_w_curr guess
    vmap_summary ms;
    const int rc2 = vmap_align(vm, vil, (int)(sc.size() / 4), sc.data(), (int)(ss.size() / 4), ss.data(), q, t, &o, &ms);
    std::printf("VMAP %d %d %d %d %.6f %.6f %.6f %.6f\n", rc2, ms.rounds, ms.n_edge, ms.n_plane, t[0], t[1], t[2], 2.0 * std::atan2(q[2], q[3]));
    vmap_destroy(vm); vil_destroy(vil);
    return (rc1 == 0 && rc2 == 0) ? 0 : 4;
}
