// Host-side stress of the context machinery of libposehip.so through the C ABI: several
// threads create contexts on GPU 0, bind them, upload a resident train set, run matcher and
// pose calls, share a train set across contexts, and destroy the contexts again -- while
// another thread keeps calling through the default context and a third polls the
// registry's read-only queries.  Every result is compared with the one the default context
// computed up front.  Meant to be built twice more with -fsanitize=address,undefined and
// -fsanitize=thread on the host side (tests/tools/build_sanitized.sh) so that the locking and
// lifetime logic runs under the sanitizers on a GPU box.  Exit status 0 iff everything matched.
//
// build: g++ -std=c++17 -O2 -pthread tests/context_stress.cpp -Iinclude -Lposeestimation_amd -lposehip
//            -Wl,-rpath,'$ORIGIN/../poseestimation_amd' -o tests/context_stress
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "posehip.h"

static std::atomic<int> g_failures{0};
#define REQUIRE(cond)                                                                             \
    do {                                                                                          \
        if (!(cond)) {                                                                            \
            std::fprintf(stderr, "%s:%d: %s failed (%s)\n", __FILE__, __LINE__, #cond,            \
                         ph_last_error_string());                                                 \
            g_failures++;                                                                         \
            return;                                                                               \
        }                                                                                         \
    } while (0)

static std::vector<uint8_t> rows(int n, unsigned seed)
{
    std::mt19937 g(seed);
    std::vector<uint8_t> d((size_t)n * 32);
    for (auto &b : d)
        b = (uint8_t)(g() & 0xFF);
    return d;
}

struct Scene {
    std::vector<float> obj, img;
    double K[9];
};

// points in a box seen by a pinhole camera 6 units away, a quarter of them outliers
static Scene scene(int n, unsigned seed)
{
    Scene s;
    std::mt19937 g(seed);
    std::uniform_real_distribution<double> u(-1.0, 1.0), px(0.0, 800.0);
    const double f = 724.26, c = std::cos(0.5), sn = std::sin(0.5);
    const double K[9] = {f, 0, 400, 0, f, 300, 0, 0, 1};
    std::memcpy(s.K, K, sizeof(K));
    for (int i = 0; i < n; i++) {
        const double X = u(g), Y = 1.29 * u(g), Z = 0.41 * u(g);
        const double x = c * X + sn * Z, y = Y, z = -sn * X + c * Z + 6.0;
        s.obj.insert(s.obj.end(), {(float)X, (float)Y, (float)Z});
        if (i % 4 == 3)
            s.img.insert(s.img.end(), {(float)px(g), (float)(0.75 * px(g))});
        else
            s.img.insert(s.img.end(), {(float)(f * x / z + 400 + 0.3 * u(g)), (float)(f * y / z + 300 + 0.3 * u(g))});
    }
    return s;
}

struct Expected {
    std::vector<int32_t> idx, dist;
    std::vector<ph_dmatch> cc;
    double rvec[3], tvec[3];
    std::vector<int32_t> inliers;
    int iterations = 0;
};

static const int NQ = 700, NT = 900, NP = 400;

static void run_calls(const std::vector<uint8_t> &q, const std::vector<uint8_t> &t, const Scene &sc,
                      const ph_db *shared_db, Expected *fill, const Expected *want)
{
    std::vector<int32_t> idx((size_t)NQ * 2), dist((size_t)NQ * 2);
    REQUIRE(ph_knn_hamming256(q.data(), NQ, 32, t.data(), NT, 32, 2, idx.data(), dist.data()) == PH_OK);
    std::vector<ph_dmatch> cc((size_t)NQ);
    int n_cc = 0;
    REQUIRE(ph_match_crosscheck_hamming256(q.data(), NQ, 32, t.data(), NT, 32, cc.data(), &n_cc) == PH_OK);
    cc.resize((size_t)n_cc);
    double rvec[3], tvec[3];
    std::vector<int32_t> inl((size_t)NP);
    int n_inl = 0, found = 0, iters = 0;
    REQUIRE(ph_solvepnp_ransac_epnp(sc.obj.data(), sc.img.data(), NP, sc.K, 100, 8.0f, 0.99,
                                    0xFFFFFFFFFFFFFFFFull, rvec, tvec, inl.data(), &n_inl, &found,
                                    &iters) == PH_OK);
    REQUIRE(found == 1);
    inl.resize((size_t)n_inl);
    if (shared_db) { // a train set another context uploaded
        std::vector<int32_t> i2((size_t)NQ * 2), d2((size_t)NQ * 2);
        REQUIRE(ph_db_knn_hamming256(shared_db, q.data(), NQ, 32, 2, i2.data(), d2.data(), nullptr, 0.f) == PH_OK);
        REQUIRE(i2 == idx && d2 == dist);
    }
    if (fill) {
        fill->idx = idx;
        fill->dist = dist;
        fill->cc = cc;
        std::memcpy(fill->rvec, rvec, sizeof(rvec));
        std::memcpy(fill->tvec, tvec, sizeof(tvec));
        fill->inliers = inl;
        fill->iterations = iters;
    }
    if (want) {
        REQUIRE(idx == want->idx && dist == want->dist);
        REQUIRE(cc.size() == want->cc.size() &&
                std::memcmp(cc.data(), want->cc.data(), cc.size() * sizeof(ph_dmatch)) == 0);
        REQUIRE(inl == want->inliers && iters == want->iterations);
        REQUIRE(std::memcmp(rvec, want->rvec, sizeof(rvec)) == 0 && std::memcmp(tvec, want->tvec, sizeof(tvec)) == 0);
    }
}

int main(int argc, char **argv)
{
    const int n_threads = argc > 1 ? std::atoi(argv[1]) : 4, reps = argc > 2 ? std::atoi(argv[2]) : 10;
    const auto q = rows(NQ, 1), t = rows(NT, 2);
    const Scene sc = scene(NP, 3);
    if (ph_init(0) != PH_OK) {
        std::fprintf(stderr, "ph_init: %s\n", ph_last_error_string());
        return 2;
    }
    Expected want;
    run_calls(q, t, sc, nullptr, &want, nullptr);
    ph_db *shared = nullptr; // owned by the default context, used from every worker's context
    if (ph_db_upload(t.data(), NT, 32, &shared) != PH_OK)
        return 2;
    std::atomic<bool> stop{false};
    std::thread default_user([&] {
        while (!stop.load())
            run_calls(q, t, sc, shared, nullptr, &want);
    });
    std::thread poller([&] {
        while (!stop.load()) {
            (void)ph_device();
            (void)ph_knn_last_path();
            (void)ph_db_rows(shared);
            (void)ph_ctx_current();
            std::this_thread::yield();
        }
    });
    std::vector<std::thread> workers;
    for (int w = 0; w < n_threads; w++)
        workers.emplace_back([&, w] {
            for (int r = 0; r < reps; r++) {
                ph_ctx *ctx = nullptr;
                REQUIRE(ph_ctx_create(0, &ctx) == PH_OK);
                REQUIRE(ph_ctx_make_current(ctx) == PH_OK);
                REQUIRE(ph_ctx_current() == ctx && ph_device() == 0);
                if ((w + r) % 3 == 0)
                    REQUIRE(ph_knn_set_path(2) == PH_OK); // this context only: matrix cores
                ph_db *mine = nullptr;
                REQUIRE(ph_db_upload(t.data(), NT, 32, &mine) == PH_OK);
                run_calls(q, t, sc, (w + r) % 2 ? shared : mine, nullptr, &want);
                if (r % 2) // released with its context otherwise
                    REQUIRE(ph_db_free(mine) == PH_OK);
                if (r % 3 == 1)
                    REQUIRE(ph_ctx_make_current(nullptr) == PH_OK);
                REQUIRE(ph_ctx_destroy(ctx) == PH_OK);
                REQUIRE(ph_ctx_current() == nullptr && ph_ctx_device(ctx) == -1);
                if (!(r % 2))
                    REQUIRE(ph_db_rows(mine) == -1);
            }
        });
    for (auto &th : workers)
        th.join();
    stop = true;
    default_user.join();
    poller.join();
    if (ph_db_free(shared) != PH_OK)
        g_failures++;
    ph_shutdown();
    std::printf("{\"threads\": %d, \"contexts_per_thread\": %d, \"failures\": %d}\n", n_threads, reps,
                g_failures.load());
    return g_failures.load() ? 1 : 0;
}
