// vilformat.hpp -- header-only C++ for the wire / on-disk formats either side of the estimator:
//   * the /feature_tracker_/feature sensor_msgs::PointCloud: points[i] = (x, y, 1) float32, channels (float32)
//     0 id * NUM_OF_CAM + camera, 1 u, 2 v, 3 velocity_x, 4 velocity_y, 5 depth
//     (encode feature_tracker_node.cpp:127-177, decode estimator_node.cpp:485-503);
//   * the trajectory log VINS_RESULT_PATH ("Frontend.txt"): one line per image, TUM order
//     "stamp px py pz qx qy qz qw", fixed notation, 9 digits for the stamp and 5 for the rest (visualization.cpp:199-212).
// Plain arrays in and out; feeds vil::FeatureTable::add_frame (vilwindow_shim.hpp).  Host logic only.
#ifndef VILFORMAT_HPP
#define VILFORMAT_HPP

#include <algorithm>
#include <cstdio>
#include <numeric>
#include <vector>

namespace vil {

struct FeatureFrame { std::vector<int> ids, camera_ids; std::vector<double> obs8; };   // obs8: [x y z u v vx vy depth] per entry

// estimator_node.cpp:485-503.  The reference collects the points in a std::map keyed by feature id and processImage uses
// the FIRST camera entry of every id, so the result is sorted by feature id (stable within an id) and, with
// first_camera_only, holds one observation per id.  Returns false if a point violates z == 1 (the reference asserts).
inline bool decode_feature_cloud(int n, const float* points_xyz, const float* const channels[6], int num_of_cam, bool first_camera_only, FeatureFrame& out) {
    std::vector<int> fid(n), cam(n), order(n);
    for (int i = 0; i < n; ++i) {
        const int v = (int)(channels[0][i] + 0.5);                    // float -> int by adding one half and truncating
        fid[i] = v / num_of_cam; cam[i] = v % num_of_cam;
        if (points_xyz[3 * i + 2] != 1.0f) return false;
    }
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return fid[a] < fid[b]; });
    out.ids.clear(); out.camera_ids.clear(); out.obs8.clear();
    for (int q = 0; q < n; ++q) {
        const int i = order[q];
        if (first_camera_only && !out.ids.empty() && out.ids.back() == fid[i]) continue;
        out.ids.push_back(fid[i]); out.camera_ids.push_back(cam[i]);
        const double o[8] = {points_xyz[3 * i], points_xyz[3 * i + 1], points_xyz[3 * i + 2], channels[1][i], channels[2][i], channels[3][i], channels[4][i], channels[5][i]};
        out.obs8.insert(out.obs8.end(), o, o + 8);
    }
    return true;
}

// feature_tracker_node.cpp:127-177: the tracker side of the same message (ids below 2^24 / NUM_OF_CAM survive the float channel)
inline void encode_feature_cloud(const FeatureFrame& in, int num_of_cam, std::vector<float>& points_xyz, std::vector<float> channels[6]) {
    const size_t n = in.ids.size();
    points_xyz.resize(3 * n);
    for (int c = 0; c < 6; ++c) channels[c].resize(n);
    for (size_t i = 0; i < n; ++i) {
        const double* o = &in.obs8[8 * i];
        points_xyz[3 * i] = (float)o[0]; points_xyz[3 * i + 1] = (float)o[1]; points_xyz[3 * i + 2] = 1.0f;
        channels[0][i] = (float)(in.ids[i] * num_of_cam + in.camera_ids[i]);
        channels[1][i] = (float)o[3]; channels[2][i] = (float)o[4]; channels[3][i] = (float)o[5]; channels[4][i] = (float)o[6]; channels[5][i] = (float)o[7];
    }
}

// visualization.cpp:199-212: returns the number of characters written (excluding the terminator), line ends with '\n'
inline int format_trajectory_line(double stamp, const double P[3], const double q_xyzw[4], char* buf, size_t cap) {
    return std::snprintf(buf, cap, "%.9f %.5f %.5f %.5f %.5f %.5f %.5f %.5f\n", stamp, P[0], P[1], P[2], q_xyzw[0], q_xyzw[1], q_xyzw[2], q_xyzw[3]);
}
inline bool parse_trajectory_line(const char* line, double& stamp, double P[3], double q_xyzw[4]) {
    return std::sscanf(line, "%lf %lf %lf %lf %lf %lf %lf %lf", &stamp, &P[0], &P[1], &P[2], &q_xyzw[0], &q_xyzw[1], &q_xyzw[2], &q_xyzw[3]) == 8;
}

}  // namespace vil
#endif
