// The matcher flow of the reference's main program, on the posehip shim.
//
// src/cereal_box.cpp keeps the previous frame's ORB descriptors, matches the
// current frame against them (:344), filters by distance (:345-356), and -- on
// its EPnP branch -- feeds 2D-3D pairs to solvePnPRansac.  This program does the
// same with synthetic frames (ORB detection and the OpenGL renderer stay on the
// host in the reference and are not part of this library).  With a file name as its
// argument it also writes its inputs and results there (raw little-endian arrays, in the
// order of the dump() calls below); tests/test_cpp_shim.py reads that file and checks every
// result against the CPU oracle.
//
// build: g++ -std=c++17 -O2 examples/cereal_box_matcher.cpp -Iinclude -Lposeestimation_amd -lposehip
//            -Wl,-rpath,'$ORIGIN/../poseestimation_amd' -o examples/cereal_box_matcher
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "posehip.hpp"

using namespace posehip;

static std::FILE *g_dump = nullptr;
template <typename T> static void dump(const T *p, std::size_t n)
{
    if (g_dump) {
        const std::int64_t count = (std::int64_t)n;
        std::fwrite(&count, sizeof(count), 1, g_dump);
        std::fwrite(p, sizeof(T), n, g_dump);
    }
}

static std::vector<std::uint8_t> random_descriptors(int n, unsigned seed)
{
    std::mt19937 rng(seed);
    std::vector<std::uint8_t> d((std::size_t)n * 32);
    for (auto &b : d)
        b = (std::uint8_t)(rng() & 0xFF);
    return d;
}

int main(int argc, char **argv)
{
    if (argc > 1 && !(g_dump = std::fopen(argv[1], "wb")))
        return 2;
    // matching global variables of the reference (src/cereal_box.cpp:47-53)
    std::vector<DMatch> matches, good_matches;
    auto matcher = DescriptorMatcher::create("BruteForce-Hamming");

    const int n_old = 500, n_new = 480;
    std::vector<std::uint8_t> descriptorsOld = random_descriptors(n_old, 1);
    // the new frame: most descriptors of the old one with a few flipped bits, shuffled
    std::vector<std::uint8_t> descriptors((std::size_t)n_new * 32);
    std::mt19937 rng(2);
    for (int i = 0; i < n_new; i++) {
        const int src = (int)((i * 7919u) % n_old);
        for (int b = 0; b < 32; b++)
            descriptors[(std::size_t)i * 32 + b] = descriptorsOld[(std::size_t)src * 32 + b];
        for (int f = 0; f < (int)(rng() % 24); f++) {
            const unsigned bit = rng() % 256;
            descriptors[(std::size_t)i * 32 + bit / 8] ^= (std::uint8_t)(1u << (bit % 8));
        }
    }

    // computeORBfeatureMatches(), src/cereal_box.cpp:341-358
    matcher->match(Descriptors(descriptors.data(), n_new), Descriptors(descriptorsOld.data(), n_old), matches);
    minDistFilter(matches, good_matches);

    // matcher.py:18-24: cross-check matcher, sorted by distance
    std::vector<DMatch> cc;
    BFMatcher(NORM_HAMMING, true).match(Descriptors(descriptors.data(), n_new),
                                        Descriptors(descriptorsOld.data(), n_old), cc);

    // k = 2 + ratio test
    std::vector<DMatch> ratio;
    ratioTestMatch(Descriptors(descriptors.data(), n_new), Descriptors(descriptorsOld.data(), n_old), 0.75f, ratio);

    // the train set resident on the GPU (add + train): the same matches as passing it per call
    BFMatcher resident;
    resident.add(Descriptors(descriptorsOld.data(), n_old));
    resident.train();
    std::vector<DMatch> matches_resident;
    resident.match(Descriptors(descriptors.data(), n_new), matches_resident);
    bool resident_same = matches_resident.size() == matches.size();
    for (std::size_t i = 0; resident_same && i < matches.size(); i++)
        resident_same = matches_resident[i].trainIdx == matches[i].trainIdx &&
                        matches_resident[i].distance == matches[i].distance && matches_resident[i].imgIdx == 0;
    std::printf("resident train set gives the same matches: %d\n", (int)resident_same);
    if (!resident_same)
        return 1;

    dump(descriptors.data(), descriptors.size());
    dump(descriptorsOld.data(), descriptorsOld.size());
    dump(reinterpret_cast<const std::int32_t *>(matches.data()), matches.size() * 4);
    dump(reinterpret_cast<const std::int32_t *>(good_matches.data()), good_matches.size() * 4);
    dump(reinterpret_cast<const std::int32_t *>(cc.data()), cc.size() * 4);
    dump(reinterpret_cast<const std::int32_t *>(ratio.data()), ratio.size() * 4);

    unsigned long long digest = 0; // wraps modulo 2^64
    for (const DMatch &m : matches)
        digest = digest * 1000003ull + (unsigned long long)(m.queryIdx * 131 + m.trainIdx * 17 + (int)m.distance);
    std::printf("matches %zu good %zu crosscheck %zu ratio %zu digest %llu\n", matches.size(), good_matches.size(),
                cc.size(), ratio.size(), digest);
    for (int i = 0; i < 5; i++)
        std::printf("match[%d] = (%d, %d, %g)\n", i, matches[i].queryIdx, matches[i].trainIdx, matches[i].distance);

    // pose: solvePnPRansac on synthetic 2D-3D pairs of the box (EPnP branch)
    const double f = 300.0 / std::tan(22.5 * M_PI / 180.0);
    const double K[9] = {f, 0, 400, 0, f, 300, 0, 0, 1};
    std::vector<Point3f> obj;
    std::vector<Point2f> img;
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    const double ang = 0.5;
    for (int i = 0; i < 400; i++) {
        const float X = U(rng), Y = 1.2934f * U(rng), Z = 0.41149f * U(rng);
        const double x = std::cos(ang) * X + std::sin(ang) * Z, y = Y, z = -std::sin(ang) * X + std::cos(ang) * Z + 6;
        obj.push_back({X, Y, Z});
        if (i % 4 == 3)
            img.push_back({400 + 400 * U(rng), 300 + 300 * U(rng)}); // outlier
        else
            img.push_back({(float)(x / z * f + 400), (float)(y / z * f + 300)});
    }
    double rvec[3], tvec[3];
    std::vector<int> inliers;
    // OpenCV's positional arguments: (obj, img, K, distCoeffs, rvec, tvec, useExtrinsicGuess,
    // iterationsCount, reprojectionError, confidence, inliers, flags)
    const bool ok = solvePnPRansac(obj, img, K, noArray(), rvec, tvec, false, 100, 8.0f, 0.99, inliers,
                                   SOLVEPNP_EPNP);
    dump(&obj[0].x, obj.size() * 3);
    dump(&img[0].x, img.size() * 2);
    dump(K, 9);
    dump(rvec, 3);
    dump(tvec, 3);
    dump(inliers.data(), inliers.size());
    std::printf("pnp ok %d inliers %zu rvec %.6f %.6f %.6f tvec %.6f %.6f %.6f\n", (int)ok, inliers.size(), rvec[0],
                rvec[1], rvec[2], tvec[0], tvec[1], tvec[2]);
    // the same frame three times as one batched call: every result equals the single call
    const std::vector<PnPResult> batch = solvePnPRansacBatch({obj, obj, obj}, {img, img, img}, K);
    bool same = batch.size() == 3;
    for (const PnPResult &r : batch)
        same = same && r.found && r.inliers == inliers && std::memcmp(r.rvec, rvec, sizeof(rvec)) == 0 &&
               std::memcmp(r.tvec, tvec, sizeof(tvec)) == 0;
    std::printf("pnp batch of 3 identical to the single call: %d\n", (int)same);
    ph_shutdown();
    if (g_dump)
        std::fclose(g_dump);
    return ok && same && inliers.size() >= 300 && inliers.size() <= 305 ? 0 : 1;
}
