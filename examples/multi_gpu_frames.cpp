// A batch of frames dealt to several GPUs from the threads of ONE C++ process -- the
// frame-sharded form of the matcher for a host like the reference's, which is a C++ program
// with worker threads (src/cereal_box.cpp:280) and no Python around it.
//
// Every worker thread owns a posehip::Context (ph_ctx_create: its own stream, workspace and
// lock on one GPU), binds it to itself, uploads the model database to its GPU once
// (DescriptorMatcher::add + train), and matches its contiguous block of frames against it
// with knnMatch(k = 2) + ratio test, one frame per call as the reference's loop does
// (src/cereal_box.cpp:341-344).  Frames are independent: the workers exchange nothing.
//
//   multi_gpu_frames [--devices 0,1,2,...] [--frames N] [--rows R] [--db ROWS] [dump-file]
//
// --devices lists one GPU per worker; a GPU may be listed more than once (workers then share
// it, each through its own context; that is how a one-GPU box exercises this program).
// With a dump file the inputs and every frame's result are written there (raw little-endian
// arrays in the order of the dump() calls below) for tests/test_multi_gpu_frames.py, which
// replays every frame on the CPU oracle.
//
// build: g++ -std=c++17 -O2 -pthread examples/multi_gpu_frames.cpp -Iinclude -Lposeestimation_amd
//            -lposehip -Wl,-rpath,'$ORIGIN/../poseestimation_amd' -o examples/multi_gpu_frames
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "posehip.hpp"

using namespace posehip;

static std::vector<std::uint8_t> random_descriptors(std::size_t n, unsigned seed)
{
    std::mt19937 rng(seed);
    std::vector<std::uint8_t> d(n * 32);
    for (std::size_t i = 0; i + 4 <= d.size(); i += 4) {
        const std::uint32_t w = rng();
        std::memcpy(&d[i], &w, 4);
    }
    return d;
}

template <typename T> static void dump(std::FILE *f, const T *p, std::size_t n)
{
    const std::int64_t count = (std::int64_t)n;
    std::fwrite(&count, sizeof(count), 1, f);
    std::fwrite(p, sizeof(T), n, f);
}

struct WorkerResult {
    int first_frame = 0, n_frames = 0;
    double seconds = 0;
    std::string error;
};

int main(int argc, char **argv)
{
    std::vector<int> devices = {0};
    int n_frames = 64, rows = 2000, db_rows = 50000;
    const char *dump_path = nullptr;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--devices" && i + 1 < argc) {
            devices.clear();
            for (char *tok = std::strtok(argv[++i], ","); tok; tok = std::strtok(nullptr, ","))
                devices.push_back(std::atoi(tok));
        } else if (a == "--frames" && i + 1 < argc) {
            n_frames = std::atoi(argv[++i]);
        } else if (a == "--rows" && i + 1 < argc) {
            rows = std::atoi(argv[++i]);
        } else if (a == "--db" && i + 1 < argc) {
            db_rows = std::atoi(argv[++i]);
        } else {
            dump_path = argv[i];
        }
    }
    if (devices.empty() || n_frames < 1 || rows < 1 || db_rows < 2)
        return 2;
    const int n_workers = (int)devices.size();
    const float ratio = 0.75f;

    // the model database and the frames' descriptors; every frame holds a few exact copies of
    // database rows, so that the ratio test keeps something
    const std::vector<std::uint8_t> db = random_descriptors((std::size_t)db_rows, 1);
    std::vector<std::uint8_t> frames = random_descriptors((std::size_t)n_frames * rows, 2);
    for (int f = 0; f < n_frames; f++)
        for (int r = 0; r < rows; r += 97)
            std::memcpy(&frames[((std::size_t)f * rows + r) * 32],
                        &db[((std::size_t)(f * 131 + r * 7) % db_rows) * 32], 32);

    // results of the whole batch, frame after frame: idx[rows][2], dist[rows][2], keep[rows]
    std::vector<std::int32_t> idx((std::size_t)n_frames * rows * 2), dist(idx.size());
    std::vector<std::uint8_t> keep((std::size_t)n_frames * rows);
    std::vector<WorkerResult> res((std::size_t)n_workers);

    auto worker = [&](int w) {
        WorkerResult &out = res[(std::size_t)w];
        // contiguous block of frames of worker w
        const int lo = (int)((long long)n_frames * w / n_workers);
        const int hi = (int)((long long)n_frames * (w + 1) / n_workers);
        out.first_frame = lo;
        out.n_frames = hi - lo;
        try {
            Context ctx(devices[(std::size_t)w]);
            CurrentContext bind(ctx);
            ph_db *model = nullptr; // this GPU's copy of the database, resident for all frames
            check(ph_db_upload(db.data(), db_rows, 32, &model));
            const auto t0 = std::chrono::steady_clock::now();
            for (int f = lo; f < hi; f++) {
                const std::size_t r0 = (std::size_t)f * rows;
                check(ph_db_knn_hamming256(model, &frames[r0 * 32], rows, 32, 2, &idx[r0 * 2],
                                           &dist[r0 * 2], &keep[r0], ratio));
            }
            out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            check(ph_db_free(model));
        } catch (const std::exception &e) {
            out.error = e.what();
        }
    };
    std::vector<std::thread> threads;
    const auto t0 = std::chrono::steady_clock::now();
    for (int w = 0; w < n_workers; w++)
        threads.emplace_back(worker, w);
    for (std::thread &t : threads)
        t.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    long long kept = 0;
    for (std::uint8_t k : keep)
        kept += k;
    for (int w = 0; w < n_workers; w++) {
        const WorkerResult &r = res[(std::size_t)w];
        if (!r.error.empty()) {
            std::fprintf(stderr, "worker %d (device %d): %s\n", w, devices[(std::size_t)w], r.error.c_str());
            return 1;
        }
        std::printf("worker %d device %d frames %d..%d seconds %.6f\n", w, devices[(std::size_t)w],
                    r.first_frame, r.first_frame + r.n_frames, r.seconds);
    }
    std::printf("frames %d rows %d db %d workers %d kept %lld wall %.6f evals_per_s %.4g\n", n_frames, rows,
                db_rows, n_workers, kept, wall, (double)n_frames * rows * db_rows / wall);
    if (dump_path) {
        std::FILE *f = std::fopen(dump_path, "wb");
        if (!f)
            return 2;
        dump(f, db.data(), db.size());
        dump(f, frames.data(), frames.size());
        dump(f, idx.data(), idx.size());
        dump(f, dist.data(), dist.size());
        dump(f, keep.data(), keep.size());
        std::fclose(f);
    }
    return 0;
}
