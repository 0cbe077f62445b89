// The reference's main loop without a window: src/cereal_box.cpp:243-301 (render loop with
// the asynchronous post-processing worker) and :315-358 (drawORBfeatures ->
// computeORBfeatures + computeORBfeatureMatches), continued through the step the reference
// leaves as a TODO (:292-295, pixel -> 3D point) into solvePnPRansac(SOLVEPNP_EPNP) and
// the model matrix that replaces the "TEMPORARY" ground truth on the wireframe (:286).
//
// What stays outside this program is what the library does not cover: the OpenGL renderer
// and ORB.  A FrameSource stands in for the pair (glReadPixels :279 + detect/compute
// :336-338): the box carries fixed feature points with fixed 256-bit descriptors, and a
// frame is what ORB::create() would report of them at the frame's model matrix -- at most
// 500 keypoints (pixel rows from the top, detection noise) with their descriptors (a few
// flipped bits) plus clutter.  cv::imshow / waitKey (:275-276) become one log line per
// post-processed frame, so the program runs on a machine without a display.
//
// The loop keeps the reference's structure: a frame is captured only when the previous
// post-processing has finished (postProcessingDone, :273-277), the worker runs on another
// thread through std::async (:280) and calls the matcher and the pose solver from there, and
// the main thread keeps "rendering" (advancing the box by 50 degrees per second, :263).
//
//   headless_pipeline [--frames N] [--fps F] [--lockstep] [--stride S] [--track] [--start T] [--dump FILE]
//     --start T    the clock (glfwGetTime, :263) at the first frame, default 0.6 s = 30 degrees, where
//                  two faces of the box are in view; frames that show a single face give EPnP
//                  coplanar points, which OpenCV's EPnP (four control points, no planar variant)
//                  does not solve reliably -- such frames are marked "one face" and summarised apart
//     --lockstep   wait for the worker in every iteration and advance S rendered frames
//                  between captures: the run is then deterministic (used by the test suite)
//     --track      lift the old frame's keypoints with the pose ESTIMATED for it instead of
//                  its ground-truth model matrix (frame 0 always starts from ground truth)
//     --dump FILE  per post-processed frame: inputs and results as raw arrays (int64 count +
//                  elements), which tests/test_headless_pipeline.py replays on the CPU oracle
//
// build: g++ -std=c++17 -O2 examples/headless_pipeline.cpp -Iinclude -Lposeestimation_amd -lposehip
//            -pthread -Wl,-rpath,'$ORIGIN/../poseestimation_amd' -o examples/headless_pipeline
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "posehip.hpp"
#include "posehip_geometry.hpp"

using namespace posehip;
namespace geo = posehip::geometry;

struct KeyPoint { // what this path reads of a cv::KeyPoint
    Point2f pt;
};

// ---- stand-in for renderer + ORB ---------------------------------------------------
class FrameSource {
  public:
    static constexpr int N_FEATURES = 1500, MAX_KEYPOINTS = 500, N_CLUTTER = 40;

    FrameSource()
    {
        std::mt19937 rng(7);
        std::uniform_real_distribution<double> U(-1.0, 1.0);
        const double *h = geo::BOX_HALF_EXTENTS;
        // area of the face pairs normal to x, y, z
        const double area[3] = {h[1] * h[2], h[0] * h[2], h[0] * h[1]};
        std::discrete_distribution<int> axis_of({area[0], area[1], area[2]});
        for (int i = 0; i < N_FEATURES; i++) {
            const int a = axis_of(rng);
            const double s = rng() & 1 ? 1.0 : -1.0;
            double p[3] = {U(rng) * h[0], U(rng) * h[1], U(rng) * h[2]};
            p[a] = s * h[a];
            for (int j = 0; j < 3; j++) {
                point_.push_back(p[j]);
                normal_.push_back(j == a ? s : 0.0);
            }
            for (int b = 0; b < 32; b++)
                descriptor_.push_back((std::uint8_t)(rng() & 0xFF));
        }
        geo::cameraMatrix(K_);
    }

    const double *K() const { return K_; }

    // Keypoints and descriptors of the frame rendered with `model` (column-major GL matrix).
    void detectAndCompute(const double model[16], unsigned frame_seed, std::vector<KeyPoint> &keypoints,
                          std::vector<std::uint8_t> &descriptors) const
    {
        double view[16], R[9], t[3];
        geo::viewMatrix(view);
        geo::poseFromModelMatrix(model, view, R, t);
        std::mt19937 rng(1000 + frame_seed);
        std::normal_distribution<float> noise(0.f, 0.5f);
        struct Seen {
            unsigned response;
            int feature;
            float u, v;
        };
        std::vector<Seen> seen;
        for (int i = 0; i < N_FEATURES; i++) {
            const double *p = &point_[(std::size_t)3 * i], *n = &normal_[(std::size_t)3 * i];
            double pc[3], nc[3];
            for (int r = 0; r < 3; r++) {
                pc[r] = R[3 * r] * p[0] + R[3 * r + 1] * p[1] + R[3 * r + 2] * p[2] + t[r];
                nc[r] = R[3 * r] * n[0] + R[3 * r + 1] * n[1] + R[3 * r + 2] * n[2];
            }
            // a face is drawn where it looks at the camera (which sits at the origin)
            if (nc[0] * pc[0] + nc[1] * pc[1] + nc[2] * pc[2] >= -0.05 * pc[2] || pc[2] <= 0.1)
                continue;
            const float u = (float)(K_[0] * pc[0] / pc[2] + K_[2]) + noise(rng);
            const float v = (float)(K_[4] * pc[1] / pc[2] + K_[5]) + noise(rng);
            if (u < 16 || v < 16 || u > geo::SCR_WIDTH - 16 || v > geo::SCR_HEIGHT - 16)
                continue; // ORB keeps away from the border (edgeThreshold 31)
            seen.push_back({(unsigned)rng(), i, u, v});
        }
        // the strongest responses survive, as with ORB's nfeatures = 500
        std::sort(seen.begin(), seen.end(), [](const Seen &a, const Seen &b) { return a.response > b.response; });
        if ((int)seen.size() > MAX_KEYPOINTS - N_CLUTTER)
            seen.resize(MAX_KEYPOINTS - N_CLUTTER);
        keypoints.clear();
        descriptors.clear();
        for (const Seen &s : seen) {
            keypoints.push_back({{s.u, s.v}});
            const std::uint8_t *d = &descriptor_[(std::size_t)32 * s.feature];
            const std::size_t at = descriptors.size();
            descriptors.insert(descriptors.end(), d, d + 32);
            for (int f = (int)(rng() % 20); f > 0; f--) {
                const unsigned bit = rng() % 256;
                descriptors[at + bit / 8] ^= (std::uint8_t)(1u << (bit % 8));
            }
        }
        std::uniform_real_distribution<float> Upx(0.f, 1.f);
        for (int c = 0; c < N_CLUTTER; c++) { // background corners that belong to no feature
            keypoints.push_back({{Upx(rng) * geo::SCR_WIDTH, Upx(rng) * geo::SCR_HEIGHT}});
            for (int b = 0; b < 32; b++)
                descriptors.push_back((std::uint8_t)(rng() & 0xFF));
        }
    }

  private:
    std::vector<double> point_, normal_;
    std::vector<std::uint8_t> descriptor_;
    double K_[9];
};

// ---- the reference's globals (src/cereal_box.cpp:39-53) -----------------------------
static FrameSource g_source;
static std::vector<KeyPoint> keypoints, keypointsOld;
static std::vector<std::uint8_t> descriptors, descriptorsOld;
static std::vector<DMatch> matches, good_matches;
static std::shared_ptr<BFMatcher> matcher;
static std::atomic<bool> postProcessingDone{true};
static bool firstFrame = false; // the reference's name: true once a previous frame exists
static std::future<int> async_out;

// what glReadPixels hands the worker (:279): here the frame's model matrix and number
static double g_captured_model[16];
static unsigned g_captured_index = 0;
// the previous frame's pose, from which its keypoints are lifted to the box
static double g_model_old[16];
static bool g_track = false;

struct FrameResult {
    unsigned index = 0;
    int n_keypoints = 0, n_matches = 0, n_good = 0, n_pairs = 0, n_inliers = 0;
    bool matched = false, found = false;
    bool one_face = false; // every 2D-3D pair lies on one face of the box
    double rot_err = 0, trans_err = 0, model_err = 0, worker_ms = 0;
    double model_est[16];
};
static FrameResult g_result;
static std::FILE *g_dump = nullptr;

template <typename T> static void dump(const T *p, std::size_t n)
{
    if (g_dump) {
        const std::int64_t count = (std::int64_t)n;
        std::fwrite(&count, sizeof(count), 1, g_dump);
        if (n)
            std::fwrite(p, sizeof(T), n, g_dump);
    }
}

static void computeORBfeatures() // :333-339
{
    g_source.detectAndCompute(g_captured_model, g_captured_index, keypoints, descriptors);
}

static void computeORBfeatureMatches() // :341-358
{
    if (firstFrame) {
        matcher->match(Descriptors(descriptors.data(), (int)keypoints.size()),
                       Descriptors(descriptorsOld.data(), (int)keypointsOld.size()), matches);
        // the reference's loop runs to descriptorsOld.rows over matches (one per QUERY row) and
        // appends to a vector it never clears; the filter here has the same rule on the
        // match list's own length and starts empty
        minDistFilter(matches, good_matches);
    }
}

// The step behind the TODO (:292-295): the old frame's matched keypoints lifted to the box,
// paired with the new frame's pixels, solvePnPRansac on the pairs, and the estimate turned
// into the wireframe's model matrix (:286).
static void estimatePose(FrameResult &res)
{
    double view[16], R_old[9], t_old[3];
    geo::viewMatrix(view);
    geo::poseFromModelMatrix(g_model_old, view, R_old, t_old);
    std::vector<double> px(good_matches.size() * 2), lifted(good_matches.size() * 3);
    for (std::size_t i = 0; i < good_matches.size(); i++) {
        px[2 * i] = keypointsOld[(std::size_t)good_matches[i].trainIdx].pt.x;
        px[2 * i + 1] = keypointsOld[(std::size_t)good_matches[i].trainIdx].pt.y;
    }
    const std::vector<unsigned char> hit =
        geo::liftPixelsToBox(px.data(), good_matches.size(), R_old, t_old, g_source.K(), lifted.data());
    std::vector<Point3f> obj;
    std::vector<Point2f> img;
    for (std::size_t i = 0; i < good_matches.size(); i++)
        if (hit[i]) {
            obj.push_back({(float)lifted[3 * i], (float)lifted[3 * i + 1], (float)lifted[3 * i + 2]});
            img.push_back(keypoints[(std::size_t)good_matches[i].queryIdx].pt);
        }
    res.n_pairs = (int)obj.size();
    // Pairs on a single face are coplanar.  OpenCV's EPnP always places four control points
    // and has no planar variant, so on such a frame it returns a pose that reprojects within
    // the threshold but can sit far from the truth; the library reproduces that, and the
    // frame is marked so that the summary can tell the two kinds of frame apart.
    int on_face[6] = {0, 0, 0, 0, 0, 0};
    for (const Point3f &p : obj) {
        const float c[3] = {p.x, p.y, p.z};
        for (int a = 0; a < 3; a++)
            if (std::fabs(std::fabs(c[a]) - geo::BOX_HALF_EXTENTS[a]) < 1e-4)
                on_face[2 * a + (c[a] > 0)]++;
    }
    res.one_face = !obj.empty() && *std::max_element(on_face, on_face + 6) >= 0.97 * obj.size();
    double rvec[3] = {0, 0, 0}, tvec[3] = {0, 0, 0};
    std::vector<int> inliers;
    res.found = obj.size() >= 5 && solvePnPRansac(obj, img, g_source.K(), noArray(), rvec, tvec, false, 100, 8.0f,
                                                  0.99, inliers, SOLVEPNP_EPNP);
    res.n_inliers = (int)inliers.size();
    dump(obj.empty() ? nullptr : &obj[0].x, obj.size() * 3);
    dump(img.empty() ? nullptr : &img[0].x, img.size() * 2);
    dump(rvec, 3);
    dump(tvec, 3);
    dump(inliers.data(), inliers.size());
    if (!res.found)
        return;
    geo::modelMatrixFromPose(rvec, tvec, view, res.model_est);
    // against the ground truth the reference still draws: the pose errors of
    // src/evaluation/Evaluation.py:5-15 and the largest difference of the model matrices
    double R_gt[9], t_gt[3], R_est[9];
    geo::poseFromModelMatrix(g_captured_model, view, R_gt, t_gt);
    geo::rodrigues(rvec, R_est);
    double dr = 0, nr = 0, dt = 0, nt = 0;
    for (int i = 0; i < 9; i++) {
        dr += (R_gt[i] - R_est[i]) * (R_gt[i] - R_est[i]);
        nr += R_est[i] * R_est[i];
    }
    for (int i = 0; i < 3; i++) {
        dt += (t_gt[i] - tvec[i]) * (t_gt[i] - tvec[i]);
        nt += tvec[i] * tvec[i];
    }
    res.rot_err = std::sqrt(dr / nr);
    res.trans_err = std::sqrt(dt / nt);
    for (int i = 0; i < 16; i++)
        res.model_err = std::max(res.model_err, std::fabs(res.model_est[i] - g_captured_model[i]));
}

static int drawORBfeatures() // :315-331, on the worker thread
{
    const auto t0 = std::chrono::steady_clock::now();
    FrameResult res;
    res.index = g_captured_index;
    computeORBfeatures();
    computeORBfeatureMatches();
    res.n_keypoints = (int)keypoints.size();
    res.matched = firstFrame;
    if (firstFrame) {
        res.n_matches = (int)matches.size();
        res.n_good = (int)good_matches.size();
        dump(descriptors.data(), descriptors.size());
        dump(descriptorsOld.data(), descriptorsOld.size());
        dump(reinterpret_cast<const std::int32_t *>(matches.data()), matches.size() * 4);
        dump(reinterpret_cast<const std::int32_t *>(good_matches.data()), good_matches.size() * 4);
        estimatePose(res);
    }
    // cv::drawKeypoints / cv::flip (:321-322) prepared the window's image; nothing to draw here
    firstFrame = true;
    keypointsOld = keypoints;
    descriptorsOld = descriptors;
    // the next frame lifts this frame's keypoints: with its ground-truth pose (a reference
    // image with known 3D coordinates, as the TODO plans) or, tracking, with its estimate
    std::memcpy(g_model_old, g_track && res.found ? res.model_est : g_captured_model, sizeof(g_model_old));
    res.worker_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    g_result = res;
    postProcessingDone = true;
    return 0;
}

int main(int argc, char **argv)
{
    int n_frames = 30, stride = 6;
    double fps = 60.0, start = 0.6;
    bool lockstep = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--frames" && i + 1 < argc)
            n_frames = std::atoi(argv[++i]);
        else if (a == "--fps" && i + 1 < argc)
            fps = std::atof(argv[++i]);
        else if (a == "--stride" && i + 1 < argc)
            stride = std::atoi(argv[++i]);
        else if (a == "--start" && i + 1 < argc)
            start = std::atof(argv[++i]);
        else if (a == "--lockstep")
            lockstep = true;
        else if (a == "--track")
            g_track = true;
        else if (a == "--dump" && i + 1 < argc) {
            if (!(g_dump = std::fopen(argv[++i], "wb")))
                return 2;
        } else {
            std::fprintf(stderr, "usage: %s [--frames N] [--fps F] [--lockstep] [--stride S] [--track] [--start SECONDS] [--dump FILE]\n",
                         argv[0]);
            return 2;
        }
    }
    matcher = DescriptorMatcher::create("BruteForce-Hamming"); // :53 (initialises the GPU library)
    dump(g_source.K(), 9);

    const double axis[3] = {0.3, 1.0, 0.0};
    double wire_model[16]; // what the wire shader is given (:286): the latest estimate
    double sim_time = start;
    int processed = 0, posed = 0, rendered = 0, one_face_frames = 0;
    double worst_rot = 0, worst_trans = 0, worst_model = 0, worker_ms_sum = 0;
    const auto wall0 = std::chrono::steady_clock::now();
    geo::modelMatrix(0.0, axis, wire_model);
    while (processed < n_frames + 1) { // the first frame only seeds descriptorsOld
        // render (:262-270): the box at 50 degrees per second about (0.3, 1, 0)
        double model_mat[16];
        geo::modelMatrix(sim_time * 50.0 * geo::PI / 180.0, axis, model_mat);
        rendered++;
        if (postProcessingDone) { // :273
            if (async_out.valid()) {
                async_out.get();
                const FrameResult &r = g_result; // in place of imshow (:275): one line per frame
                if (r.matched) {
                    std::printf("frame %4u: %3d keypoints %3d matches %3d good %3d pairs | pose %d inliers %3d "
                                "rot_err %.2e trans_err %.2e model_err %.2e%s | worker %.2f ms\n",
                                r.index, r.n_keypoints, r.n_matches, r.n_good, r.n_pairs, (int)r.found, r.n_inliers,
                                r.rot_err, r.trans_err, r.model_err, r.one_face ? " (one face)" : "", r.worker_ms);
                    worker_ms_sum += r.worker_ms;
                    if (r.found)
                        posed++;
                    if (r.found && r.one_face)
                        one_face_frames++;
                    if (r.found && !r.one_face) {
                        worst_rot = std::max(worst_rot, r.rot_err);
                        worst_trans = std::max(worst_trans, r.trans_err);
                        worst_model = std::max(worst_model, r.model_err);
                    }
                    if (r.found)
                        std::memcpy(wire_model, r.model_est, sizeof(wire_model));
                }
                processed++;
            }
            if (processed < n_frames + 1) {
                postProcessingDone = false; // :277
                std::memcpy(g_captured_model, model_mat, sizeof(g_captured_model)); // glReadPixels, :279
                g_captured_index = (unsigned)rendered;
                async_out = std::async(std::launch::async, drawORBfeatures); // :280
            }
        }
        // the wireframe is drawn with wire_model (:283-290); swap buffers (:299) paces the loop
        if (lockstep) {
            if (async_out.valid())
                async_out.wait();
            sim_time += stride / 60.0;
            rendered += stride - 1;
        } else {
            sim_time += 1.0 / fps;
            std::this_thread::sleep_for(std::chrono::duration<double>(1.0 / fps));
        }
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
    std::printf("summary: %d frames post-processed of %d rendered in %.2f s, %d poses, worst rot_err %.3e "
                "trans_err %.3e model_err %.3e, %d one-face frames apart, mean worker %.2f ms\n",
                n_frames, rendered, wall, posed, worst_rot, worst_trans, worst_model, one_face_frames,
                n_frames ? worker_ms_sum / n_frames : 0.0);
    matcher.reset();
    ph_shutdown();
    if (g_dump)
        std::fclose(g_dump);
    return posed == n_frames ? 0 : 1;
}
