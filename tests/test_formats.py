"""include/vilformat.hpp and its Python mirror: feature PointCloud channel layout and the trajectory log."""
import os
import subprocess
import tempfile

import numpy as np

from mvil_fusion_amd import formats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = r'''
#include <cstdio>
#include <cstdlib>
#include "vilformat.hpp"
int main(int argc, char** argv) {
    // input: n, num_of_cam, then n rows: x y chan0..chan5
    int n, ncam; if (std::scanf("%d %d", &n, &ncam) != 2) return 1;
    std::vector<float> pts(3 * n), ch[6]; for (auto& c : ch) c.resize(n);
    for (int i = 0; i < n; ++i) { if (std::scanf("%f %f", &pts[3 * i], &pts[3 * i + 1]) != 2) return 1; pts[3 * i + 2] = 1.0f; for (int c = 0; c < 6; ++c) if (std::scanf("%f", &ch[c][i]) != 1) return 1; }
    const float* cp[6] = {ch[0].data(), ch[1].data(), ch[2].data(), ch[3].data(), ch[4].data(), ch[5].data()};
    vil::FeatureFrame f;
    if (!vil::decode_feature_cloud(n, pts.data(), cp, ncam, true, f)) return 2;
    for (size_t i = 0; i < f.ids.size(); ++i) { std::printf("D %d %d", f.ids[i], f.camera_ids[i]); for (int q = 0; q < 8; ++q) std::printf(" %.9g", f.obs8[8 * i + q]); std::printf("\n"); }
    std::vector<float> p2, c2[6];
    vil::encode_feature_cloud(f, ncam, p2, c2);
    vil::FeatureFrame g;
    const float* cq[6] = {c2[0].data(), c2[1].data(), c2[2].data(), c2[3].data(), c2[4].data(), c2[5].data()};
    vil::decode_feature_cloud((int)f.ids.size(), p2.data(), cq, ncam, true, g);
    std::printf("R %d\n", (int)(g.ids == f.ids && g.camera_ids == f.camera_ids && g.obs8 == f.obs8));
    char buf[256];
    const double P[3] = {1.234567891, -20.5, 0.000004}, q[4] = {0.1, -0.2, 0.3, 0.9273618495495704};
    vil::format_trajectory_line(1403636579.763555527, P, q, buf, sizeof buf);
    std::printf("T %s", buf);
    double st, P2[3], q2[4];
    const bool okp = vil::parse_trajectory_line(buf, st, P2, q2);
    std::printf("P %d %.9f %.5f %.5f\n", (int)okp, st, P2[0], q2[3]);
    pts[2] = 0.5f;
    std::printf("Z %d\n", (int)vil::decode_feature_cloud(n, pts.data(), cp, ncam, true, f));
    return 0;
}
'''


def test_cpp_and_python_agree():
    rng = np.random.default_rng(5)
    n, ncam = 40, 2
    ids = rng.permutation(60)[:n] + 1000
    cams = rng.integers(0, ncam, n)
    ids[5], cams[5] = ids[4], 1 - cams[4]                      # the same feature seen by both cameras
    xy = rng.normal(0, 0.3, (n, 2)).astype(np.float32)
    ch = [(ids * ncam + cams).astype(np.float32)] + [rng.normal(100, 50, n).astype(np.float32) for _ in range(2)] + [rng.normal(0, 1, n).astype(np.float32) for _ in range(2)]
    ch.append(np.where(rng.random(n) < 0.4, rng.uniform(1, 20, n), -1.0).astype(np.float32))
    text = "%d %d\n" % (n, ncam) + "\n".join(" ".join("%.9g" % v for v in [xy[i, 0], xy[i, 1]] + [c[i] for c in ch]) for i in range(n))
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        open(src, "w").write(PROG)
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", os.path.join(d, "t")])
        out = subprocess.run([os.path.join(d, "t")], input=text.encode(), capture_output=True, check=True).stdout.decode().strip().split("\n")
    pts = np.column_stack([xy, np.ones(n, np.float32)])
    pid, pcam, pobs = formats.decode_feature_cloud(pts, ch, ncam)
    D = [l.split() for l in out if l.startswith("D ")]
    assert len(D) == len(pid) == n - 1                                             # the duplicate id collapsed onto its first entry
    assert [int(l[1]) for l in D] == list(pid) == sorted(pid) and [int(l[2]) for l in D] == list(pcam)
    assert np.allclose(np.array([[float(v) for v in l[3:]] for l in D]), pobs, rtol=1e-8, atol=0)
    assert "R 1" in out and "Z 0" in out
    T = [l for l in out if l.startswith("T ")][0][2:] + "\n"
    assert T == formats.format_trajectory_line(1403636579.763555527, [1.234567891, -20.5, 0.000004], [0.1, -0.2, 0.3, 0.9273618495495704])
    assert T == "1403636579.763555527 1.23457 -20.50000 0.00000 0.10000 -0.20000 0.30000 0.92736\n"
    assert [l for l in out if l.startswith("P ")][0] == "P 1 1403636579.763555527 1.23457 0.92736"
    rows = formats.parse_trajectory(T + T)
    assert rows.shape == (2, 8) and rows[1, 7] == 0.92736
    # encode -> decode round trip in Python
    p2, c2 = formats.encode_feature_cloud(pid, pcam, pobs, ncam)
    i2, k2, o2 = formats.decode_feature_cloud(p2, c2, ncam)
    assert np.array_equal(i2, pid) and np.array_equal(k2, pcam) and np.array_equal(o2, pobs)
