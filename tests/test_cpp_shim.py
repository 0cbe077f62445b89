"""The C++ shim (include/posehip.hpp): it must compile without OpenCV (CPU suite)
and the host example built on it must run on the GPU (GPU suite)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shim_compiles_without_opencv(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "posehip.hpp"\n'
                   'int main() { posehip::DMatch m; std::vector<posehip::DMatch> v; (void)m; '
                   'return (int)v.size(); }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only",
                           "-I" + os.path.join(ROOT, "include"), str(src)])
    # and as plain C for the C ABI header
    csrc = tmp_path / "t.c"
    csrc.write_text('#include "posehip.h"\nint main(void) { return sizeof(ph_dmatch) == 16 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only",
                           "-I" + os.path.join(ROOT, "include"), str(csrc)])


# A stand-in for <opencv2/core.hpp>: just enough declarations (no definitions) of the OpenCV
# types the shim's POSEHIP_WITH_OPENCV overloads touch, with OpenCV's own member signatures,
# so that those overloads can be compile-checked (syntax only) where OpenCV is not installed.
OPENCV_STAND_IN = r"""
#pragma once
#include <cstddef>
#include <vector>
#define CV_8UC1 0
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
namespace cv {
class _InputArray; class _OutputArray;
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
struct MatStep { operator std::size_t() const; };
class Mat {
  public:
    Mat();
    unsigned char *data; int rows, cols; MatStep step;
    bool empty() const; int type() const; std::size_t total() const;
    int checkVector(int elemChannels, int depth = -1, bool requireContinuous = true) const;
    Mat reshape(int cn, int rows = 0) const;
    void convertTo(OutputArray m, int rtype, double alpha = 1, double beta = 0) const;
    template <typename T> T &at(int i, int j); template <typename T> const T &at(int i, int j) const;
};
class _InputArray {
  public:
    _InputArray(); _InputArray(const Mat &m);
    template <typename T> _InputArray(const std::vector<T> &v);
    Mat getMat(int idx = -1) const; bool empty() const;
};
class _OutputArray : public _InputArray {
  public:
    _OutputArray(); _OutputArray(Mat &m); template <typename T> _OutputArray(std::vector<T> &v);
    void create(int rows, int cols, int type) const; bool needed() const;
};
const _OutputArray &noArray();
struct DMatch { int queryIdx, trainIdx, imgIdx; float distance; };
template <typename T> struct Point3_ { T x, y, z; }; typedef Point3_<float> Point3f;
template <typename T> struct Point_ { T x, y; }; typedef Point_<float> Point2f;
template <typename T> struct Ptr { T *operator->() const; };
}
"""


def test_shim_opencv_overloads_compile_against_a_stand_in(tmp_path):
    """The reference's own lines (src/cereal_box.cpp:51-53,344 and the EPnP branch's
    solvePnPRansac call with OpenCV's positional arguments) must compile against the shim
    with POSEHIP_WITH_OPENCV, on cv::Mat / std::vector<cv::DMatch> / cv::InputArray."""
    inc = tmp_path / "opencv2"
    inc.mkdir()
    (inc / "core.hpp").write_text(OPENCV_STAND_IN)
    src = tmp_path / "ref_lines.cpp"
    src.write_text(r"""
#define POSEHIP_WITH_OPENCV
#include "posehip.hpp"
using namespace cv;
Mat descriptors, descriptorsOld;
std::vector<DMatch> matches, good_matches;                       // src/cereal_box.cpp:51-52
auto matcher = posehip::DescriptorMatcher::create("BruteForce-Hamming");   // :53
void computeORBfeatureMatches()
{
    matcher->match(descriptors, descriptorsOld, matches);        // :344, unchanged
    std::vector<std::vector<DMatch>> knn;
    matcher->knnMatch(descriptors, descriptorsOld, knn, 2);
}
bool pose(const std::vector<Point3f> &objectPoints, const std::vector<Point2f> &imagePoints,
          const Mat &cameraMatrix, const Mat &distCoeffs, Mat &rvec, Mat &tvec, std::vector<int> &inliers)
{
    bool a = posehip::solvePnPRansac(objectPoints, imagePoints, cameraMatrix, distCoeffs, rvec, tvec, false,
                                     100, 8.0f, 0.99, inliers, posehip::SOLVEPNP_EPNP);
    bool b = posehip::solvePnPRansac(objectPoints, imagePoints, cameraMatrix, noArray(), rvec, tvec);
    return a && b;
}
int main() { return 0; }
""")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only",
                           "-I" + str(tmp_path), "-I" + os.path.join(ROOT, "include"), str(src)])


def test_shim_solvepnp_positional_signature_without_opencv(tmp_path):
    """OpenCV's positional argument order (distCoeffs after cameraMatrix, inliers before flags)
    on the shim's own types; too few points are refused before any pointer is formed."""
    src = tmp_path / "t.cpp"
    src.write_text(r"""
#include "posehip.hpp"
using namespace posehip;
int main()
{
    std::vector<Point3f> obj; std::vector<Point2f> img; double K[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, r[3], t[3];
    std::vector<int> inliers; std::vector<double> dist(4, 0.0);
    try {
        solvePnPRansac(obj, img, K, noArray(), r, t, false, 100, 8.0f, 0.99, inliers, SOLVEPNP_EPNP);
        return 1;
    } catch (const Error &e) {
        if (e.status != PH_ERR_INVALID_ARG) return 2;
    }
    try {
        dist[1] = 0.1; obj.resize(6); img.resize(6);
        solvePnPRansac(obj, img, K, dist, r, t);
        return 3;
    } catch (const Error &e) {
        if (e.status != PH_ERR_INVALID_ARG) return 4;
    }
    return 0;
}
""")
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           str(src), "-L" + os.path.join(ROOT, "poseestimation_amd"), "-lposehip",
                           "-Wl,-rpath," + os.path.join(ROOT, "poseestimation_amd"),
                           "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(exe)])
    # both refusals happen in the shim, before the library is asked for a GPU
    assert subprocess.run([str(exe)]).returncode == 0


def _read_dump(path):
    """Arrays as examples/cereal_box_matcher.cpp dumps them: int64 count + raw elements."""
    import numpy as np
    raw = open(path, "rb").read()
    dtypes = [np.uint8, np.uint8, np.int32, np.int32, np.int32, np.int32, np.float32, np.float32,
              np.float64, np.float64, np.float64, np.int32]
    out, pos = [], 0
    for dt in dtypes:
        n = int(np.frombuffer(raw, np.int64, 1, pos)[0])
        pos += 8
        out.append(np.frombuffer(raw, dt, n, pos).copy())
        pos += n * np.dtype(dt).itemsize
    assert pos == len(raw)
    return out


def _dmatches(a):
    """(queryIdx, trainIdx, imgIdx, distance) rows of a dumped DMatch array."""
    import numpy as np
    a = a.reshape(-1, 4)
    return [(int(r[0]), int(r[1]), int(r[2]), float(np.array([r[3]], np.int32).view(np.float32)[0]))
            for r in a]


@pytest.mark.gpu
def test_cpp_host_example_runs(tmp_path):
    """The reference's matcher + pose flow on the C++ shim, compiled with plain g++: every
    result the program computed on the GPU is compared with the CPU oracle on the program's
    own inputs (match lists exactly; inliers exactly; pose within 1e-4 relative error)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_bind as orc
    exe = os.path.join(ROOT, "examples", "cereal_box_matcher")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    dump = tmp_path / "dump.bin"
    out = subprocess.run([exe, str(dump)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"matches (\d+) good (\d+) crosscheck (\d+) ratio (\d+)", out.stdout)
    assert m, out.stdout
    n_match, n_good, n_cc, n_ratio = map(int, m.groups())
    assert n_match == 480 and 0 < n_good <= 480 and 0 < n_cc <= 480 and 0 < n_ratio <= 480
    assert "pnp ok 1" in out.stdout
    (desc, desc_old, matches, good, cc, ratio, obj, img, K, rvec, tvec, inliers) = _read_dump(dump)
    q, t = desc.reshape(-1, 32), desc_old.reshape(-1, 32)
    want = orc.match(q, t)
    assert _dmatches(matches) == want and len(want) == n_match
    assert _dmatches(good) == orc.min_dist_filter(want)
    assert _dmatches(cc) == orc.match(q, t, cross_check=True)
    ridx, rdist = orc.knn(q, t, 2)
    assert _dmatches(ratio) == orc.ratio_filter(ridx, rdist, 0.75)
    obj, img = obj.reshape(-1, 3), img.reshape(-1, 2)
    ok0, r0, t0, inl0, _ = orc.solvepnp_ransac(obj, img, K.reshape(3, 3), iterations=100)
    assert ok0 and np.array_equal(inliers, inl0)
    Ra, Rb = orc.rodrigues(rvec), orc.rodrigues(r0)
    assert np.linalg.norm(Rb - Ra) / np.linalg.norm(Ra) <= 1e-4
    assert np.linalg.norm(t0 - tvec) / np.linalg.norm(tvec) <= 1e-4
