/*
 * posehip.h -- C ABI of libposehip.so: the MI355X (gfx950) implementation of the
 * correspondence-and-pose hot path of seqwalt/PoseEstimation.
 *
 * The reference has no FFI layer of its own; the boundary it would bind is the
 * pair of OpenCV calls on its hot path.  Every entry point names the reference
 * call site it replaces (paths relative to the reference tree):
 *
 *   src/cereal_box.cpp:53        DescriptorMatcher::create("BruteForce-Hamming")
 *   src/cereal_box.cpp:344       matcher->match(descriptors, descriptorsOld, matches)
 *   src/cereal_box.cpp:345-356   min-distance filter on the match list
 *   src/orb_feature_matching/matcher.py:18,21   BFMatcher(NORM_HAMMING, crossCheck=True).match
 *   src/cereal_box.cpp:292-295 + docs/index.md:75,120
 *                                cv::solvePnPRansac(..., SOLVEPNP_EPNP) (EPnP branch)
 *
 * Conventions
 *   - plain C types only; the caller owns every buffer it passes in;
 *   - no call throws; each returns a ph_status, and ph_last_error_string()
 *     describes the last failure on the calling thread;
 *   - "host" entry points take host pointers and copy in/out themselves;
 *     "_dev" entry points take device pointers (HBM-resident) and a HIP stream
 *     (NULL = the library's own stream) and return after enqueueing unless
 *     stated otherwise;
 *   - descriptors are 256-bit ORB rows: 32 bytes each, row stride >= 32;
 *   - calls may come from any thread (the reference calls from a std::async
 *     worker, src/cereal_box.cpp:280).  Every call runs in a context: the
 *     default one of ph_init, or one made with ph_ctx_create and bound to the
 *     calling thread with ph_ctx_make_current.  Calls through one context are
 *     serialised by its lock; calls through different contexts run concurrently
 *     (own stream, own workspace), on the same GPU or on different ones.
 */
#ifndef POSEHIP_H
#define POSEHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PH_OK = 0,
    PH_ERR_INVALID_ARG = 1,
    PH_ERR_HIP = 2,
    PH_ERR_NO_DEVICE = 3,
    PH_ERR_NOT_INITIALIZED = 4,
    PH_ERR_TOO_LARGE = 5
} ph_status;

/* Layout-compatible with cv::DMatch (16 bytes). */
typedef struct {
    int32_t queryIdx;
    int32_t trainIdx;
    int32_t imgIdx;
    float distance;
} ph_dmatch;

#define PH_NO_MATCH_IDX (-1)
#define PH_NO_MATCH_DIST 2147483647

/* ---- lifetime ------------------------------------------------------------ */

/* Creates the default context: binds it to one GPU and creates its stream and
 * workspace (further GPUs, or further concurrent callers: ph_ctx_create below).
 * Replaces the matcher construction at src/cereal_box.cpp:53.  May be called
 * again with the same device (no-op).  ph_shutdown releases the default context
 * and the resident train sets uploaded through it. */
ph_status ph_init(int device);
void ph_shutdown(void);
const char *ph_last_error_string(void);
/* Device of the calling thread's context (the default one unless
 * ph_ctx_make_current bound another), or -1. */
int ph_device(void);

/* ---- more than one GPU, or more than one caller, in one process ------------ */
/* The reference is one process with worker threads (src/cereal_box.cpp:280) and has
 * no notion of a device; a host like it that deals frames to the 8 GPUs of a node
 * needs one binding per GPU.  A context is such a binding: its own stream,
 * workspace, staging buffers, settings (k-NN path, profiling) and lock.  ph_init
 * creates the default context; ph_ctx_create makes another on any GPU, also on one
 * that already has a context (two workers sharing a GPU then overlap instead of
 * queueing behind one lock).
 *
 * Every ph_* entry point acts on the calling thread's CURRENT context:
 * ph_ctx_make_current(ctx) binds one to the calling thread (like hipSetDevice, the
 * binding is per thread), ph_ctx_make_current(NULL) returns the thread to the
 * default context.  A worker thread per GPU therefore calls ph_ctx_make_current
 * once and then uses the same entry points as single-GPU code.  Device pointers and
 * streams passed to a call must belong to the context's GPU.  A resident train set
 * (ph_db_upload) belongs to the context that uploaded it, is released with it, and
 * serves calls of every context on the same GPU.
 *
 * ph_ctx_destroy waits for calls in flight through the context, then releases what
 * it holds; threads that still have it current get PH_ERR_INVALID_ARG from their
 * next call.  ph_shutdown releases the default context only.  The scratch of the
 * allocation-free entry point is per GPU, not per context. */
typedef struct ph_ctx ph_ctx;
ph_status ph_ctx_create(int device, ph_ctx **ctx);
ph_status ph_ctx_destroy(ph_ctx *ctx);
ph_status ph_ctx_make_current(ph_ctx *ctx);
ph_ctx *ph_ctx_current(void);          /* NULL: the default context */
int ph_ctx_device(const ph_ctx *ctx);  /* -1 for a dead handle */

/* ---- descriptor matching, host buffers ----------------------------------- */

/* knnMatch(query, train, k) for k = 1 or 2 (cv::DescriptorMatcher::knnMatch;
 * k=1 is the arithmetic of src/cereal_box.cpp:344).  out_idx/out_dist hold
 * nq*k entries, row-major per query, best first; lowest train index wins among
 * equal distances.  Slots beyond min(k, nt) hold PH_NO_MATCH_IDX /
 * PH_NO_MATCH_DIST.  nq == 0 or nt == 0 is not an error. */
ph_status ph_knn_hamming256(const uint8_t *query, int nq, size_t q_step, const uint8_t *train,
                            int nt, size_t t_step, int k, int32_t *out_idx, int32_t *out_dist);

/* match(query, train) without cross-check: src/cereal_box.cpp:344.
 * out holds nq entries; *n_out = nq when nt > 0, else 0. */
ph_status ph_match_hamming256(const uint8_t *query, int nq, size_t q_step, const uint8_t *train,
                              int nt, size_t t_step, ph_dmatch *out, int *n_out);

/* match(query, train) of a matcher built with crossCheck=true:
 * src/orb_feature_matching/matcher.py:18,21.  A pair (q, t) is returned iff t is q's
 * nearest train row AND q is t's nearest query row, each with the lowest-index rule on
 * ties (cv::batchDistance's crosscheck branch since the fix of OpenCV issue 11855 in 2018, pinned by its
 * regression test Features2d_DMatch.issue_11855).  out holds up to nq entries in
 * ascending queryIdx; queries without a consistent pair are dropped. */
ph_status ph_match_crosscheck_hamming256(const uint8_t *query, int nq, size_t q_step,
                                         const uint8_t *train, int nt, size_t t_step,
                                         ph_dmatch *out, int *n_out);

/* knnMatch(k=2) followed by Lowe's ratio test (keep m[0] iff
 * m[0].distance < ratio * m[1].distance, float arithmetic).  The reference has
 * no ratio test; the benchmark configuration "kNN k=2 ratio test" defines it.
 * out holds up to nq entries in ascending queryIdx. */
ph_status ph_knn2_ratio_hamming256(const uint8_t *query, int nq, size_t q_step,
                                   const uint8_t *train, int nt, size_t t_step, float ratio,
                                   ph_dmatch *out, int *n_out);

/* The filter of src/cereal_box.cpp:345-356 with explicit bounds: keep
 * distance <= max(2*min_dist, 30.0).  Host-only (at most a few hundred
 * elements in the reference).  out holds up to n entries; out may equal m. */
ph_status ph_min_dist_filter(const ph_dmatch *m, int n, ph_dmatch *out, int *n_out);

/* The Lowe ratio test on the rows of a k = 2 result (idx / dist as written by
 * ph_knn_hamming256): keep row i iff it has two neighbours and
 * (float)dist[2i] < ratio * (float)dist[2i+1].  Host-only; out holds up to nq
 * entries.  (Defined by the benchmark configuration, not by the reference; the
 * device entry points apply the same test on the GPU through d_keep.) */
ph_status ph_ratio_filter(const int32_t *idx, const int32_t *dist, int nq, float ratio,
                          ph_dmatch *out, int *n_out);

/* ---- a train set resident in HBM ------------------------------------------- */
/* DescriptorMatcher::add(descriptors) + train() of OpenCV's matcher interface
 * (the reference passes the train set on every call, src/cereal_box.cpp:344): a
 * model database that many frames are matched against is uploaded once.
 * ph_db_upload copies nt rows (stride t_step >= 32) to the GPU and returns a
 * handle; ph_db_free releases it (ph_shutdown releases all live handles);
 * ph_db_device_ptr is the packed device buffer for the device entry points
 * below.  ph_db_knn_hamming256 is ph_knn_hamming256 against the resident set
 * (host queries, host results; keep may be NULL, otherwise k must be 2 and
 * keep[i] receives the ratio test's verdict); ph_db_knn_hamming256_dev is
 * ph_knn_hamming256_dev against it (device queries and results, only enqueues).
 * Both keep whatever the distance kernel needs beside the rows (the FP4 image of
 * the matrix-core path, 128 bytes per row, built by the first large call) with
 * the handle, so a call pays for the queries only. */
typedef struct ph_db ph_db;
ph_status ph_db_upload(const uint8_t *train, int64_t nt, size_t t_step, ph_db **db);
ph_status ph_db_free(ph_db *db);
int64_t ph_db_rows(const ph_db *db);           /* -1 for a dead handle */
const void *ph_db_device_ptr(const ph_db *db); /* NULL for a dead handle or an empty set */
ph_status ph_db_knn_hamming256(const ph_db *db, const uint8_t *query, int64_t nq, size_t q_step,
                               int k, int32_t *out_idx, int32_t *out_dist, uint8_t *keep,
                               float ratio);
ph_status ph_db_knn_hamming256_dev(const ph_db *db, const void *d_query, int64_t nq, int k,
                                   int32_t *d_out_idx, int32_t *d_out_dist, uint8_t *d_keep,
                                   float ratio, void *stream);

/* ---- descriptor matching, device buffers --------------------------------- */

/* Same result as ph_knn_hamming256 with HBM-resident inputs and outputs.
 * d_query / d_train: packed rows (stride 32 bytes), 16-byte aligned.
 * train_index_base is added to every emitted train index (train-set shards
 * report global indices).  d_keep may be NULL; otherwise k must be 2 and
 * d_keep[i] (one byte per query) is set to 1 iff the ratio test with `ratio`
 * passes.  Enqueues on `stream` (a hipStream_t, NULL = library stream) and
 * returns without waiting.  A context's workspace is shared by its calls: a call
 * on another stream than the context's previous one is ordered after that one's
 * work (calls through ONE context never run concurrently with each other; calls
 * through different contexts do). */
ph_status ph_knn_hamming256_dev(const void *d_query, int64_t nq, const void *d_train, int64_t nt,
                                int k, int32_t train_index_base, int32_t *d_out_idx,
                                int32_t *d_out_dist, uint8_t *d_keep, float ratio, void *stream);

/* ph_knn_hamming256_dev for callers that cannot allocate: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart
 * in the reference.  Same arguments, kernels and results as
 * ph_knn_hamming256_dev, but the call's scratch (FP4 image of the train rows,
 * per-split keys) lives in a fixed device array of PH_NOALLOC_SCRATCH_BYTES that
 * every GPU the library runs on holds until the process ends, and the call
 * makes no hipMalloc / hipFree or other allocating call.  A problem whose
 * scratch does not fit (roughly nt * 128 + splits * nq * 8 bytes; 100k x 100k
 * fits, 1M x 50k does not) is refused with PH_ERR_TOO_LARGE and a message naming
 * the bytes needed.  Every call on a GPU shares that GPU's array: calls that
 * overlap on one GPU (different streams without an ordering between them) race;
 * the library does not order them.  Profiling events are not recorded. */
#define PH_NOALLOC_SCRATCH_BYTES ((size_t)32 << 20)
ph_status ph_knn_hamming256_noalloc_dev(const void *d_query, int64_t nq, const void *d_train,
                                        int64_t nt, int k, int32_t train_index_base,
                                        int32_t *d_out_idx, int32_t *d_out_dist, uint8_t *d_keep,
                                        float ratio, void *stream);

/* A batch of frames in one launch: frame f matches query rows
 * [q_offsets[f], q_offsets[f+1]) of d_query against train rows
 * [t_offsets[f], t_offsets[f+1]) of d_train, i.e. n_frames calls of
 * matcher->knnMatch(descriptors, descriptorsOld, ...) as the reference issues one
 * per rendered frame (src/cereal_box.cpp:341-344), fused so that small frames fill
 * the GPU.  q_offsets / t_offsets: HOST arrays of n_frames + 1 ascending row
 * offsets (copied before the call returns).  d_query and d_train may be the same
 * buffer (frame f against frame f-1 of one descriptor stream).  Outputs are
 * indexed by query row of d_query (rows q_offsets[0] .. q_offsets[n_frames]);
 * train indices are relative to the frame's first train row.  A frame without
 * train rows yields PH_NO_MATCH_* rows.  Train offsets that are multiples of 4
 * rows run fastest.  Otherwise as ph_knn_hamming256_dev. */
ph_status ph_knn_hamming256_frames_dev(const void *d_query, const int32_t *q_offsets,
                                       const void *d_train, const int32_t *t_offsets,
                                       int n_frames, int k, int32_t *d_out_idx,
                                       int32_t *d_out_dist, uint8_t *d_keep, float ratio,
                                       void *stream);

/* Cross-check match for a batch of frames in one launch: per frame the result of
 * ph_match_crosscheck_hamming256_dev (BFMatcher(NORM_HAMMING, crossCheck=true)
 * .match of src/orb_feature_matching/matcher.py:18,21), frames laid out as for
 * ph_knn_hamming256_frames_dev.  d_out_idx / d_out_dist are indexed by query row of
 * d_query (not compacted); train indices are relative to the frame. */
ph_status ph_match_crosscheck_hamming256_frames_dev(const void *d_query, const int32_t *q_offsets,
                                                    const void *d_train, const int32_t *t_offsets,
                                                    int n_frames, int32_t *d_out_idx,
                                                    int32_t *d_out_dist, void *stream);

/* Merge per-shard k-NN candidate lists (the step after the all-gather of a
 * train-sharded search).  d_cand_idx / d_cand_dist: [n_shards][nq][k], shards
 * ordered by ascending train index range, indices global.  Output as
 * ph_knn_hamming256_dev.  Lowest index wins among equal distances. */
ph_status ph_knn_merge_dev(const int32_t *d_cand_idx, const int32_t *d_cand_dist, int n_shards,
                           int64_t nq, int k, int32_t *d_out_idx, int32_t *d_out_dist,
                           uint8_t *d_keep, float ratio, void *stream);

/* Cross-check match on device buffers: d_out_idx[q] = matched train index or
 * PH_NO_MATCH_IDX, d_out_dist[q] its distance (not compacted). */
ph_status ph_match_crosscheck_hamming256_dev(const void *d_query, int64_t nq, const void *d_train,
                                             int64_t nt, int32_t *d_out_idx, int32_t *d_out_dist,
                                             void *stream);

/* The cross-check matcher when the train set is sharded over GPUs (no counterpart in the
 * reference, which is single-process; the arithmetic is that of
 * src/orb_feature_matching/matcher.py:18,21).  With the queries replicated, every train
 * row's nearest query is known to the shard that owns the row.  This call returns, per
 * query, its nearest train row WITHIN this shard as a candidate for ph_knn_merge_dev:
 * d_out_idx[q] = 2 * (index + train_index_base) + flag, flag = 1 iff q is that row's
 * nearest query (PH_NO_MATCH_* for a shard without rows); indices are therefore limited
 * to 2^30.  Over the whole train set: all-gather the lists, fold them with
 * ph_knn_merge_dev(k = 1) -- the smallest (distance, index) wins and its flag rides
 * along -- and unpack with ph_match_crosscheck_unpack_dev. */
ph_status ph_match_crosscheck_hamming256_shard_dev(const void *d_query, int64_t nq,
                                                   const void *d_train_shard, int64_t nt,
                                                   int32_t train_index_base, int32_t *d_out_idx,
                                                   int32_t *d_out_dist, void *stream);

/* Last step of the sharded cross-check (no counterpart in the reference): in place,
 * d_idx[q] = 2 * index + flag becomes index where the flag is set and PH_NO_MATCH_IDX /
 * PH_NO_MATCH_DIST where it is not (the query's nearest train row has another nearest
 * query). */
ph_status ph_match_crosscheck_unpack_dev(int32_t *d_idx, int32_t *d_dist, int64_t nq,
                                         void *stream);

/* ---- RANSAC-EPnP pose ------------------------------------------------------- */
/* Replaces cv::solvePnPRansac(obj, img, K, noArray(), rvec, tvec, false,
 * iterationsCount, reprojectionError, confidence, inliers, SOLVEPNP_EPNP), the
 * call the reference's EPnP branch makes where this checkout has the TODO at
 * src/cereal_box.cpp:292-295 (docs/index.md:75,120).  Zero lens distortion (the
 * reference renders with an ideal pinhole, src/cereal_box.cpp:222).
 *
 * obj: n x 3 float model points, img: n x 2 float pixels, K: row-major 3 x 3.
 * rng_state seeds OpenCV's multiply-with-carry cv::RNG; OpenCV itself always
 * uses 0xFFFFFFFFFFFFFFFF.  *found = 1 iff a pose was found (OpenCV's return
 * value); rvec/tvec then hold the EPnP refit over all inliers (OpenCV hands
 * that solve the consensus set as CV_64F points, so the normalised image
 * points are not rounded to float on the way in, unlike in the per-sample
 * solves; reproduced), inliers (n entries, may be NULL) their ascending row
 * indices.  iterations_run (may be
 * NULL) = RANSAC iterations consumed after the adaptive budget cut.  n < 5 is
 * PH_ERR_INVALID_ARG (OpenCV asserts n >= 4 and uses P3P at n == 4, which is
 * outside this path). */
ph_status ph_solvepnp_ransac_epnp(const float *obj, const float *img, int n, const double *K,
                                  int iterations_count, float reproj_err, double confidence,
                                  uint64_t rng_state, double *rvec, double *tvec,
                                  int32_t *inliers, int *n_inliers, int *found,
                                  int *iterations_run);

/* The call above for a batch of independent problems (one per frame) that share K
 * and the RANSAC parameters: problem p owns rows [point_offsets[p],
 * point_offsets[p+1]) of obj / img (n_problems + 1 ascending host offsets) and gets
 * exactly the result a separate ph_solvepnp_ransac_epnp call on its rows would
 * give (the sampler restarts from rng_state for every problem, as OpenCV
 * constructs its RNG inside every call).  The hypotheses of all problems are
 * solved and scored together, a GPU-full at a time, so a batch costs about the
 * latency of one call.  rvecs / tvecs: 3 doubles per problem; found,
 * n_inliers (may be NULL), iterations_run (may be NULL): one int32 per problem;
 * inliers (may be NULL): as many entries as rows, problem p's ascending,
 * problem-relative inlier rows start at inliers[point_offsets[p]].  A problem
 * with fewer than 5 rows gets found = 0 (where the single call reports an
 * error).  No counterpart in the reference, which solves one frame at a time. */
ph_status ph_solvepnp_ransac_epnp_batched(const float *obj, const float *img,
                                          const int32_t *point_offsets, int n_problems,
                                          const double *K, int iterations_count, float reproj_err,
                                          double confidence, uint64_t rng_state, double *rvecs,
                                          double *tvecs, int32_t *inliers, int32_t *n_inliers,
                                          int32_t *found, int32_t *iterations_run);

/* cv::solvePnP(obj, img, K, noArray(), rvec, tvec, false, SOLVEPNP_EPNP) on
 * n >= 4 correspondences given as CV_32F points (what RANSAC's per-sample solve
 * is; the final refit of the call above differs only in taking its points as
 * CV_64F). */
ph_status ph_solvepnp_epnp(const float *obj, const float *img, int n, const double *K,
                           double *rvec, double *tvec);

/* The first n_subsets 5-point samples RANSACPointSetRegistrator::run draws from
 * n_points correspondences with cv::RNG state rng_state: out[5*s .. 5*s+5).
 * Host-only. */
ph_status ph_ransac_subsets(uint64_t rng_state, int n_points, int n_subsets, int32_t *out);

/* Hypothesis generation + scoring on device buffers: for each of n_hyp 5-point
 * samples (d_subsets, row indices) solve EPnP -> d_models[6h..6h+6) = rvec|tvec
 * (double) and count the inliers of that model over all n correspondences ->
 * d_counts[h].  K is a host pointer.  Enqueues on `stream` and returns. */
ph_status ph_ransac_epnp_hypotheses_dev(const float *d_obj, const float *d_img, int n,
                                        const double *K, const int32_t *d_subsets, int n_hyp,
                                        float reproj_err, double *d_models, int32_t *d_counts,
                                        void *stream);

/* The same for a batch of independent problems (frames) in one launch: problem
 * p owns correspondence rows [d_point_offsets[p], d_point_offsets[p+1]) of
 * d_obj / d_img (n_problems + 1 int32 offsets on the device), all problems share
 * K and have hyp_per_problem hypotheses; d_subsets[(p*hyp_per_problem + h)*5 ..]
 * index rows RELATIVE to the problem's range; outputs are indexed
 * p*hyp_per_problem + h.  No counterpart in the reference (it solves one frame
 * at a time); it exists so that many frames fill the GPU. */
ph_status ph_ransac_epnp_hypotheses_batched_dev(const float *d_obj, const float *d_img,
                                                const int32_t *d_point_offsets, int n_problems,
                                                const double *K, const int32_t *d_subsets,
                                                int hyp_per_problem, float reproj_err,
                                                double *d_models, int32_t *d_counts, void *stream);

/* ---- measurement aids (no counterpart in the reference) -------------------- */

/* With profiling on, every k-NN call records HIP events around its distance
 * kernel and its merge kernel on the stream it uses; ph_knn_last_timing waits
 * for the last such call and returns both durations in milliseconds. */
ph_status ph_set_profiling(int enable);
ph_status ph_knn_last_timing(float *distance_kernel_ms, float *merge_kernel_ms);
/* Duration of the bit -> FP4 pre-pass of that call (0 when it ran the popcount kernel). */
ph_status ph_knn_last_prepass_timing(float *prepass_ms);
/* Duration of an empty event-to-event interval recorded right after that call:
 * the fixed cost the two figures above include on top of the kernels' own time. */
ph_status ph_knn_last_timing_overhead(float *empty_interval_ms);

/* The distance kernel exists twice: a popcount kernel on the vector ALUs and a
 * matrix-core kernel (descriptor bits as FP4 +-4, Hamming distance from the dot
 * product), bit-identical in their results.  mode 0 (default) picks by problem
 * size, 1 always takes the popcount kernel, 2 always the matrix cores.  May be
 * called before ph_init; ph_shutdown resets it to 0.  ph_init also reads the
 * environment variable POSEHIP_KNN_PATH ("popcount" / "matrix").  ph_knn_last_path: which
 * one the last k-NN call of this process ran (0 popcount, 1 matrix cores). */
ph_status ph_knn_set_path(int mode);
int ph_knn_last_path(void);

/* Sustained VALU rate, in lane-operations per second, of the instruction pair
 * the distance loop is made of (v_xor_b32 with a scalar operand + accumulating
 * v_bcnt_u32_b32), 8 waves per SIMD on every CU, `iters` x 128 operations per
 * lane.  One Hamming-256 evaluation is 16 such lane-operations, so the
 * matcher's roofline is this figure / 16.  Blocks until measured. */
ph_status ph_valu_popcount_peak(int iters, double *lane_ops_per_s);

/* Sustained rate, in FLOP per second, of the matrix-core distance kernel's one MFMA
 * (v_mfma_scale_f32_32x32x64_f8f6f4 on FP4 operands of +-4 at random) with nothing
 * else in the loop: four independent chains, two waves per SIMD on every CU, `iters` x
 * 16 MFMAs per wave.  The dense-FP4 spec peak assumes the nominal clock; this is what
 * the chip sustains at the clock it holds under that load, i.e. the ceiling of the
 * matrix-core matcher on this GPU.  Blocks until measured. */
ph_status ph_mfma_fp4_peak(int iters, double *flop_per_s);

/* Sustained fp64 VALU rates, in lane-operations per second, with 8 waves per SIMD on
 * every CU: v_fma_f64 alone (two FLOP per lane-operation: the fp64 vector peak), and the
 * v_mul_f64 + v_add_f64 mix (one FLOP each) that the RANSAC-EPnP solve is made of, being
 * built without contraction so that it rounds like the sequential statement.  Blocks
 * until measured. */
ph_status ph_valu_fp64_peak(int iters, double *fma_lane_ops_per_s,
                            double *mul_add_lane_ops_per_s);

/* The EPnP solve of a batch of more than 64 hypotheses per CU exists twice, with
 * bit-identical models: as one kernel (mode 0 / 1, the default: it measured fastest), and
 * as two (mode 2: the 12 x 12 decomposition at two waves per SIMD, then the rest, handing
 * over through a library workspace; mode 3: the second half at two waves per SIMD as well).
 * A measurement and test aid, like ph_knn_set_path. */
ph_status ph_ransac_set_solve_path(int mode);

/* With profiling on (ph_set_profiling), ph_ransac_epnp_hypotheses[_batched]_dev records HIP
 * events on its stream around the solve kernel of its first round of hypotheses, around
 * the solve kernel of its last round (a call of at most one GPU-full -- 128 hypotheses per
 * CU -- is a single round and both are the same launch) and around the last round's
 * scoring kernel; this waits for that call and returns the three durations in
 * milliseconds (any pointer may be NULL). */
ph_status ph_ransac_last_timing(float *first_solve_ms, float *last_solve_ms,
                                float *last_score_ms);

/* Blocks until everything enqueued on `stream` (NULL = library stream) is done. */
ph_status ph_stream_synchronize(void *stream);

#ifdef __cplusplus
}
#endif
#endif /* POSEHIP_H */
