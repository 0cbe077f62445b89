// posehip.hpp -- C++17 shim over the C ABI of posehip.h that keeps the call
// signatures of the OpenCV classes the reference uses on its hot path, so that
// the reference's call sites change by a type name only:
//
//   src/cereal_box.cpp:53    Ptr<DescriptorMatcher> matcher = DescriptorMatcher::create("BruteForce-Hamming");
//                         -> auto matcher = posehip::DescriptorMatcher::create("BruteForce-Hamming");
//   src/cereal_box.cpp:344   matcher->match(descriptors, descriptorsOld, matches);      (unchanged)
//   src/orb_feature_matching/matcher.py:18,21  BFMatcher(NORM_HAMMING, crossCheck=true).match(des1, des2)
//   EPnP branch (docs/index.md:75,120)  solvePnPRansac(obj, img, K, dist, rvec, tvec, false, iters, err, conf, inliers, SOLVEPNP_EPNP)
//
// The header needs no OpenCV.  Define POSEHIP_WITH_OPENCV before including it
// (with <opencv2/core.hpp> on the include path) to get overloads on OpenCV's own types:
// match / knnMatch taking cv::Mat and std::vector<cv::DMatch>, and solvePnPRansac on
// cv::InputArray / cv::OutputArray.  Errors: the C ABI never throws;
// this shim converts a non-zero status into posehip::Error (the reference lets
// cv::Exception propagate the same way).
#ifndef POSEHIP_HPP
#define POSEHIP_HPP

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "posehip.h"

#ifdef POSEHIP_WITH_OPENCV
#include <opencv2/core.hpp>
#endif

namespace posehip {

struct Error : std::runtime_error {
    ph_status status;
    Error(ph_status s, const char *msg) : std::runtime_error(msg), status(s) {}
};

inline void check(ph_status s)
{
    if (s != PH_OK)
        throw Error(s, ph_last_error_string());
}

// A binding of the library to one GPU beside the default one (ph_ctx_create): its own stream,
// workspace and lock.  No counterpart in OpenCV or the reference, which know one device; a
// host that deals frames to several GPUs -- or lets several workers share one -- creates one
// per worker.  Released with the object.
class Context {
  public:
    explicit Context(int device) { check(ph_ctx_create(device, &ctx_)); }
    ~Context() { (void)ph_ctx_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    ph_ctx *handle() const { return ctx_; }
    int device() const { return ph_ctx_device(ctx_); }

  private:
    ph_ctx *ctx_ = nullptr;
};

// Binds a context to the calling thread for the guard's lifetime: every posehip call of the
// thread (BFMatcher::match, solvePnPRansac, the C entry points) then runs in it.
class CurrentContext {
  public:
    explicit CurrentContext(const Context &c) : before_(ph_ctx_current()) { check(ph_ctx_make_current(c.handle())); }
    ~CurrentContext() { (void)ph_ctx_make_current(before_); }
    CurrentContext(const CurrentContext &) = delete;
    CurrentContext &operator=(const CurrentContext &) = delete;

  private:
    ph_ctx *before_;
};

// cv::DMatch, field for field (16 bytes).
struct DMatch {
    int queryIdx = -1;
    int trainIdx = -1;
    int imgIdx = -1;
    float distance = 3.402823466e+38f;
    DMatch() = default;
    DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(-1), distance(d) {}
    DMatch(int q, int t, int i, float d) : queryIdx(q), trainIdx(t), imgIdx(i), distance(d) {}
    bool operator<(const DMatch &m) const { return distance < m.distance; }
};
static_assert(sizeof(DMatch) == sizeof(ph_dmatch), "DMatch must match the C ABI layout");

// What the matcher needs of a cv::Mat of ORB descriptors: CV_8UC1, 32 columns.
struct Descriptors {
    const std::uint8_t *data = nullptr;
    int rows = 0;
    std::size_t step = 32; // bytes between rows (cv::Mat::step)
    Descriptors() = default;
    Descriptors(const std::uint8_t *d, int r, std::size_t s = 32) : data(d), rows(r), step(s) {}
#ifdef POSEHIP_WITH_OPENCV
    Descriptors(const cv::Mat &m) // NOLINT: implicit, like cv::InputArray
        : data(m.data), rows(m.rows), step(m.empty() ? 32 : (std::size_t)m.step)
    {
        if (!m.empty() && (m.type() != CV_8UC1 || m.cols != 32))
            throw Error(PH_ERR_INVALID_ARG, "descriptors must be CV_8UC1 with 32 columns (ORB)");
    }
#endif
};

enum NormTypes { NORM_HAMMING = 6 };
enum SolvePnPMethod { SOLVEPNP_EPNP = 1 };

// cv::BFMatcher / cv::DescriptorMatcher for NORM_HAMMING on 256-bit descriptors.
class BFMatcher {
  public:
    explicit BFMatcher(int normType = NORM_HAMMING, bool crossCheck = false, int device = 0)
        : crossCheck_(crossCheck)
    {
        if (normType != NORM_HAMMING)
            throw Error(PH_ERR_INVALID_ARG, "only NORM_HAMMING is implemented");
        if (ph_device() < 0)
            check(ph_init(device));
    }

    // DescriptorMatcher::match(query, train, matches): src/cereal_box.cpp:344.
    void match(const Descriptors &query, const Descriptors &train, std::vector<DMatch> &matches) const
    {
        matches.clear();
        if (query.rows == 0 || train.rows == 0)
            return;
        matches.resize((std::size_t)query.rows);
        int n = 0;
        auto *out = reinterpret_cast<ph_dmatch *>(matches.data());
        check(crossCheck_ ? ph_match_crosscheck_hamming256(query.data, query.rows, query.step, train.data,
                                                           train.rows, train.step, out, &n)
                          : ph_match_hamming256(query.data, query.rows, query.step, train.data, train.rows,
                                                train.step, out, &n));
        matches.resize((std::size_t)n);
    }

    // DescriptorMatcher::knnMatch(query, train, matches, k), k = 1 or 2.
    void knnMatch(const Descriptors &query, const Descriptors &train, std::vector<std::vector<DMatch>> &matches,
                  int k) const
    {
        matches.clear();
        if (crossCheck_) { // BFMatcher asserts K == 1 with crossCheck
            if (k != 1)
                throw Error(PH_ERR_INVALID_ARG, "crossCheck requires k == 1");
            std::vector<DMatch> flat;
            match(query, train, flat);
            for (const DMatch &m : flat)
                matches.push_back({m});
            return;
        }
        if (query.rows == 0 || train.rows == 0)
            return;
        std::vector<std::int32_t> idx((std::size_t)query.rows * k), dist((std::size_t)query.rows * k);
        check(ph_knn_hamming256(query.data, query.rows, query.step, train.data, train.rows, train.step, k,
                                idx.data(), dist.data()));
        matches.resize((std::size_t)query.rows);
        for (int i = 0; i < query.rows; i++)
            for (int r = 0; r < k && idx[(std::size_t)i * k + r] >= 0; r++)
                matches[(std::size_t)i].emplace_back(i, idx[(std::size_t)i * k + r], 0,
                                                     (float)dist[(std::size_t)i * k + r]);
    }

    bool isCrossCheck() const { return crossCheck_; }

#ifdef POSEHIP_WITH_OPENCV
    // The same two calls on OpenCV's own match type, so that the reference's
    // `std::vector<DMatch> matches; matcher->match(descriptors, descriptorsOld, matches);`
    // (src/cereal_box.cpp:51,344, with `using namespace cv`) compiles unchanged: cv::Mat
    // converts to Descriptors, and cv::DMatch has the layout of posehip::DMatch.
    void match(const Descriptors &query, const Descriptors &train, std::vector<cv::DMatch> &matches) const
    {
        std::vector<DMatch> tmp;
        match(query, train, tmp);
        copyMatches(tmp, matches);
    }
    void knnMatch(const Descriptors &query, const Descriptors &train,
                  std::vector<std::vector<cv::DMatch>> &matches, int k) const
    {
        std::vector<std::vector<DMatch>> tmp;
        knnMatch(query, train, tmp, k);
        matches.resize(tmp.size());
        for (std::size_t i = 0; i < tmp.size(); i++)
            copyMatches(tmp[i], matches[i]);
    }
    static void copyMatches(const std::vector<DMatch> &from, std::vector<cv::DMatch> &to)
    {
        static_assert(sizeof(cv::DMatch) == sizeof(DMatch), "cv::DMatch layout");
        to.resize(from.size());
        if (!from.empty())
            std::memcpy(static_cast<void *>(to.data()), from.data(), from.size() * sizeof(DMatch));
    }
#endif

    // ---- the train collection of cv::DescriptorMatcher: add() one descriptor set per
    // train image, train() uploads their concatenation to the GPU once, and
    // match(query, matches) / knnMatch(query, matches, k) then send only the query.
    void add(const Descriptors &descriptors)
    {
        imgStart_.push_back((int)(rows_.size() / 32));
        for (int i = 0; i < descriptors.rows; i++)
            rows_.insert(rows_.end(), descriptors.data + (std::size_t)i * descriptors.step,
                         descriptors.data + (std::size_t)i * descriptors.step + 32);
        db_.reset();
    }
    void clear()
    {
        rows_.clear();
        imgStart_.clear();
        db_.reset();
    }
    bool empty() const { return rows_.empty(); }
    void train()
    {
        if (db_ && ph_db_rows(db_.get()) >= 0)
            return;
        ph_db *h = nullptr;
        static const std::uint8_t none = 0;
        check(ph_db_upload(rows_.empty() ? &none : rows_.data(), (std::int64_t)(rows_.size() / 32), 32, &h));
        db_ = std::shared_ptr<ph_db>(h, [](ph_db *p) {
            if (ph_db_rows(p) >= 0) // not already released by ph_shutdown
                ph_db_free(p);
        });
    }
    void knnMatch(const Descriptors &query, std::vector<std::vector<DMatch>> &matches, int k)
    {
        matches.clear();
        if (k != 1 && k != 2)
            throw Error(PH_ERR_INVALID_ARG, "k must be 1 or 2");
        if (crossCheck_) { // needs the swapped search too: one call on the concatenation
            if (k != 1)
                throw Error(PH_ERR_INVALID_ARG, "crossCheck requires k == 1");
            std::vector<DMatch> flat;
            match(query, Descriptors(rows_.data(), (int)(rows_.size() / 32), 32), flat);
            for (DMatch &m : flat) {
                localise(m);
                matches.push_back({m});
            }
            return;
        }
        if (query.rows == 0 || rows_.empty())
            return;
        train();
        std::vector<std::int32_t> idx((std::size_t)query.rows * k), dist((std::size_t)query.rows * k);
        check(ph_db_knn_hamming256(db_.get(), query.data, query.rows, query.step, k, idx.data(), dist.data(),
                                   nullptr, 0.f));
        matches.resize((std::size_t)query.rows);
        for (int i = 0; i < query.rows; i++)
            for (int r = 0; r < k && idx[(std::size_t)i * k + r] >= 0; r++) {
                DMatch m(i, idx[(std::size_t)i * k + r], 0, (float)dist[(std::size_t)i * k + r]);
                localise(m);
                matches[(std::size_t)i].push_back(m);
            }
    }
    void match(const Descriptors &query, std::vector<DMatch> &matches)
    {
        std::vector<std::vector<DMatch>> rows;
        knnMatch(query, rows, 1);
        matches.clear();
        for (const auto &r : rows)
            if (!r.empty())
                matches.push_back(r[0]);
    }

  private:
    // index into the concatenation -> (imgIdx, index within that image)
    void localise(DMatch &m) const
    {
        int img = 0;
        while (img + 1 < (int)imgStart_.size() && imgStart_[(std::size_t)img + 1] <= m.trainIdx)
            img++;
        m.imgIdx = img;
        m.trainIdx -= imgStart_.empty() ? 0 : imgStart_[(std::size_t)img];
    }
    bool crossCheck_;
    std::vector<std::uint8_t> rows_; // the added descriptor sets, packed
    std::vector<int> imgStart_;      // first row of every train image
    std::shared_ptr<ph_db> db_;      // their resident copy
};

// DescriptorMatcher::create("BruteForce-Hamming"): src/cereal_box.cpp:53.
struct DescriptorMatcher {
    static std::shared_ptr<BFMatcher> create(const std::string &type)
    {
        if (type != "BruteForce-Hamming")
            throw Error(PH_ERR_INVALID_ARG, "only \"BruteForce-Hamming\" is implemented");
        return std::make_shared<BFMatcher>(NORM_HAMMING, false);
    }
};

// knnMatch(k = 2) + Lowe ratio test (benchmark configuration; not in the reference).
inline void ratioTestMatch(const Descriptors &query, const Descriptors &train, float ratio,
                           std::vector<DMatch> &good)
{
    good.clear();
    if (query.rows == 0 || train.rows == 0)
        return;
    if (ph_device() < 0)
        check(ph_init(0));
    good.resize((std::size_t)query.rows);
    int n = 0;
    check(ph_knn2_ratio_hamming256(query.data, query.rows, query.step, train.data, train.rows, train.step, ratio,
                                   reinterpret_cast<ph_dmatch *>(good.data()), &n));
    good.resize((std::size_t)n);
}

// The good_matches filter of src/cereal_box.cpp:345-356 (explicit bounds; the
// output starts empty instead of growing across frames).
inline void minDistFilter(const std::vector<DMatch> &matches, std::vector<DMatch> &good)
{
    good.resize(matches.size());
    int n = 0;
    check(ph_min_dist_filter(reinterpret_cast<const ph_dmatch *>(matches.data()), (int)matches.size(),
                             reinterpret_cast<ph_dmatch *>(good.data()), &n));
    good.resize((std::size_t)n);
}

struct Point3f {
    float x, y, z;
};
struct Point2f {
    float x, y;
};

// Stand-ins for the optional arrays of cv::solvePnPRansac.  distCoeffs: noArray(), an
// empty vector or all zeros (the reference renders with an ideal pinhole,
// src/cereal_box.cpp:222); anything else is refused, lens distortion being outside this
// path.  inliers: noArray() or a std::vector<int> to fill.
struct NoArray {};
inline NoArray noArray() { return {}; }
struct DistCoeffs {
    bool zero = true;
    DistCoeffs() = default;
    DistCoeffs(NoArray) {}        // NOLINT: implicit, like cv::InputArray
    DistCoeffs(std::nullptr_t) {} // NOLINT
    template <typename T> DistCoeffs(const std::vector<T> &v) // NOLINT
    {
        for (const T &c : v)
            zero = zero && c == T(0);
    }
};
struct InliersOut {
    std::vector<int> *v = nullptr;
    InliersOut(NoArray) {}                    // NOLINT: implicit, like cv::OutputArray
    InliersOut(std::nullptr_t) {}             // NOLINT
    InliersOut(std::vector<int> &x) : v(&x) {} // NOLINT
    InliersOut(std::vector<int> *x) : v(x) {}  // NOLINT
};

// cv::solvePnPRansac(objectPoints, imagePoints, cameraMatrix, distCoeffs, rvec, tvec,
// useExtrinsicGuess = false, iterationsCount = 100, reprojectionError = 8.0,
// confidence = 0.99, inliers = noArray(), flags): OpenCV's positional signature, the call
// of the reference's EPnP branch (docs/index.md:75,120; TODO at src/cereal_box.cpp:292-295).
// cameraMatrix: row-major 3 x 3.  rvec / tvec: 3 doubles each.  Only SOLVEPNP_EPNP is
// implemented, so it is the default here (OpenCV's default is SOLVEPNP_ITERATIVE; the
// reference passes SOLVEPNP_EPNP).  Fewer than 5 correspondences: posehip::Error, where
// OpenCV throws cv::Exception (or takes P3P at exactly 4).
inline bool solvePnPRansac(const std::vector<Point3f> &objectPoints, const std::vector<Point2f> &imagePoints,
                           const double cameraMatrix[9], const DistCoeffs &distCoeffs, double rvec[3],
                           double tvec[3], bool useExtrinsicGuess = false, int iterationsCount = 100,
                           float reprojectionError = 8.0f, double confidence = 0.99,
                           InliersOut inliers = noArray(), int flags = SOLVEPNP_EPNP)
{
    if (flags != SOLVEPNP_EPNP || useExtrinsicGuess)
        throw Error(PH_ERR_INVALID_ARG, "only SOLVEPNP_EPNP without an extrinsic guess is implemented");
    if (!distCoeffs.zero)
        throw Error(PH_ERR_INVALID_ARG, "lens distortion is not implemented (pass noArray() or zeros)");
    if (objectPoints.size() != imagePoints.size())
        throw Error(PH_ERR_INVALID_ARG, "objectPoints and imagePoints differ in length");
    if (objectPoints.size() < 5)
        throw Error(PH_ERR_INVALID_ARG, "RANSAC-EPnP needs at least 5 correspondences");
    if (ph_device() < 0)
        check(ph_init(0));
    const int n = (int)objectPoints.size();
    std::vector<std::int32_t> inl((std::size_t)n);
    int n_inl = 0, found = 0;
    check(ph_solvepnp_ransac_epnp(&objectPoints.data()->x, &imagePoints.data()->x, n, cameraMatrix,
                                  iterationsCount, reprojectionError, confidence, 0xFFFFFFFFFFFFFFFFull, rvec,
                                  tvec, inl.data(), &n_inl, &found, nullptr));
    if (inliers.v) {
        if (found)
            inliers.v->assign(inl.begin(), inl.begin() + n_inl);
        else
            inliers.v->clear();
    }
    return found != 0;
}

// The same without the distCoeffs slot (zero distortion).
inline bool solvePnPRansac(const std::vector<Point3f> &objectPoints, const std::vector<Point2f> &imagePoints,
                           const double cameraMatrix[9], double rvec[3], double tvec[3],
                           bool useExtrinsicGuess = false, int iterationsCount = 100,
                           float reprojectionError = 8.0f, double confidence = 0.99,
                           InliersOut inliers = noArray(), int flags = SOLVEPNP_EPNP)
{
    return solvePnPRansac(objectPoints, imagePoints, cameraMatrix, DistCoeffs(), rvec, tvec, useExtrinsicGuess,
                          iterationsCount, reprojectionError, confidence, inliers, flags);
}

// solvePnPRansac for a batch of frames in one call (shared camera and RANSAC
// parameters); element p of the result is what solvePnPRansac on frame p returns.
struct PnPResult {
    bool found = false;
    double rvec[3] = {0, 0, 0}, tvec[3] = {0, 0, 0};
    std::vector<int> inliers;
    int iterations = 0;
};

inline std::vector<PnPResult> solvePnPRansacBatch(const std::vector<std::vector<Point3f>> &objectPoints,
                                                  const std::vector<std::vector<Point2f>> &imagePoints,
                                                  const double cameraMatrix[9], int iterationsCount = 100,
                                                  float reprojectionError = 8.0f, double confidence = 0.99)
{
    if (objectPoints.size() != imagePoints.size())
        throw Error(PH_ERR_INVALID_ARG, "objectPoints and imagePoints differ in length");
    if (ph_device() < 0)
        check(ph_init(0));
    const int P = (int)objectPoints.size();
    std::vector<PnPResult> out((std::size_t)P);
    if (P == 0)
        return out;
    std::vector<std::int32_t> off((std::size_t)P + 1, 0);
    for (int p = 0; p < P; p++) {
        if (objectPoints[p].size() != imagePoints[p].size())
            throw Error(PH_ERR_INVALID_ARG, "objectPoints and imagePoints of a frame differ in length");
        off[p + 1] = off[p] + (std::int32_t)objectPoints[p].size();
    }
    std::vector<Point3f> obj;
    std::vector<Point2f> img;
    obj.reserve((std::size_t)off[P] + 1);
    img.reserve((std::size_t)off[P] + 1);
    for (int p = 0; p < P; p++) {
        obj.insert(obj.end(), objectPoints[p].begin(), objectPoints[p].end());
        img.insert(img.end(), imagePoints[p].begin(), imagePoints[p].end());
    }
    if (obj.empty()) { // keep the pointers valid
        obj.push_back({0, 0, 0});
        img.push_back({0, 0});
    }
    std::vector<double> rv((std::size_t)P * 3), tv((std::size_t)P * 3);
    std::vector<std::int32_t> inl(obj.size()), n_inl((std::size_t)P), found((std::size_t)P), iters((std::size_t)P);
    check(ph_solvepnp_ransac_epnp_batched(&obj[0].x, &img[0].x, off.data(), P, cameraMatrix, iterationsCount,
                                          reprojectionError, confidence, 0xFFFFFFFFFFFFFFFFull, rv.data(),
                                          tv.data(), inl.data(), n_inl.data(), found.data(), iters.data()));
    for (int p = 0; p < P; p++) {
        PnPResult &r = out[(std::size_t)p];
        r.found = found[p] != 0;
        r.iterations = iters[p];
        std::memcpy(r.rvec, &rv[(std::size_t)p * 3], sizeof(r.rvec));
        std::memcpy(r.tvec, &tv[(std::size_t)p * 3], sizeof(r.tvec));
        if (r.found)
            r.inliers.assign(inl.begin() + off[p], inl.begin() + off[p] + n_inl[p]);
    }
    return out;
}

#ifdef POSEHIP_WITH_OPENCV
// Drop-in overloads on OpenCV's own types.
inline void match(const BFMatcher &m, const cv::Mat &query, const cv::Mat &train, std::vector<cv::DMatch> &out)
{
    m.match(Descriptors(query), Descriptors(train), out);
}

// cv::solvePnPRansac on OpenCV's array types: objectPoints N x 3 (or N-vector of Point3f),
// imagePoints N x 2 (or N-vector of Point2f), cameraMatrix 3 x 3, distCoeffs empty or
// zeros; rvec / tvec come back as 3 x 1 CV_64F, inliers as an N x 1 CV_32S column.
inline bool solvePnPRansac(cv::InputArray objectPoints, cv::InputArray imagePoints, cv::InputArray cameraMatrix,
                           cv::InputArray distCoeffs, cv::OutputArray rvec, cv::OutputArray tvec,
                           bool useExtrinsicGuess = false, int iterationsCount = 100,
                           float reprojectionError = 8.0f, double confidence = 0.99,
                           cv::OutputArray inliers = cv::noArray(), int flags = SOLVEPNP_EPNP)
{
    const cv::Mat obj_in = objectPoints.getMat(), img_in = imagePoints.getMat();
    cv::Mat obj, img, K, dist;
    const int n = obj_in.checkVector(3), n2 = img_in.checkVector(2);
    if (n < 0 || n2 != n)
        throw Error(PH_ERR_INVALID_ARG, "objectPoints must hold N 3-D points and imagePoints N 2-D points");
    obj_in.reshape(1, n).convertTo(obj, CV_32F); // N x 3 / N x 2 float
    img_in.reshape(1, n).convertTo(img, CV_32F);
    cameraMatrix.getMat().convertTo(K, CV_64F);
    if (K.rows != 3 || K.cols != 3)
        throw Error(PH_ERR_INVALID_ARG, "cameraMatrix must be 3 x 3");
    const double k9[9] = {K.at<double>(0, 0), K.at<double>(0, 1), K.at<double>(0, 2),
                          K.at<double>(1, 0), K.at<double>(1, 1), K.at<double>(1, 2),
                          K.at<double>(2, 0), K.at<double>(2, 1), K.at<double>(2, 2)};
    std::vector<double> d;
    if (!distCoeffs.empty()) {
        distCoeffs.getMat().convertTo(dist, CV_64F);
        for (int i = 0; i < (int)dist.total(); i++)
            d.push_back(dist.reshape(1, 1).at<double>(0, i));
    }
    std::vector<Point3f> o((std::size_t)n);
    std::vector<Point2f> p((std::size_t)n);
    for (int i = 0; i < n; i++) {
        o[(std::size_t)i] = {obj.at<float>(i, 0), obj.at<float>(i, 1), obj.at<float>(i, 2)};
        p[(std::size_t)i] = {img.at<float>(i, 0), img.at<float>(i, 1)};
    }
    double r[3], t[3];
    std::vector<int> inl;
    const bool ok = solvePnPRansac(o, p, k9, DistCoeffs(d), r, t, useExtrinsicGuess, iterationsCount,
                                   reprojectionError, confidence, InliersOut(inl), flags);
    if (ok) {
        rvec.create(3, 1, CV_64F);
        tvec.create(3, 1, CV_64F);
        cv::Mat rm = rvec.getMat(), tm = tvec.getMat();
        for (int i = 0; i < 3; i++) {
            rm.at<double>(i, 0) = r[i];
            tm.at<double>(i, 0) = t[i];
        }
    }
    if (inliers.needed()) {
        inliers.create((int)inl.size(), 1, CV_32S);
        cv::Mat im = inliers.getMat();
        for (int i = 0; i < (int)inl.size(); i++)
            im.at<int>(i, 0) = inl[(std::size_t)i];
    }
    return ok;
}
#endif

} // namespace posehip

#endif // POSEHIP_HPP
