/*
 * gslam_hip.h — plain-C ABI of libgslam_hip.so: the MI355X (gfx950) hot path of GSLAM.
 *
 * This is the drop-in boundary underneath the GSLAM plugin shims (libgslam_optimizer.so,
 * libgslam_featuredetector.so).  Everything here is `extern "C"`, plain pointers + sizes,
 * POD structs, int status codes, no exceptions, no torch / C++ types.
 *
 * Reference interfaces each entry point stands in for (paths relative to the GSLAM tree):
 *   gh_bf_*      GSLAM/core/Vocabulary.h:485-491  (DistanceFactory::hamming32: 4 x u64 XOR + popcount)
 *                GSLAM/core/Vocabulary.h:1712-1725 (first strict minimum wins ties -> lowest index)
 *                GSLAM/core/Map.h:252-258         (FrameConnection::getMatches vector<pair<int,int>>)
 *   gh_orb_*     GSLAM/core/Map.h:122-195         (KeyPoint, 28-byte POD == gh_keypoint)
 *                GSLAM/core/Map.h:309-321         (MapFrame::setKeyPoints(kps, N x 32 8UC1 GImage))
 *                GSLAM/core/GImage.h:160-190,378-382 (dense row-major u8 image / descriptor matrix)
 *   gh_ba_*      GSLAM/core/Optimizer.h:102-182   (BundleGraph, BundleEdge, KeyFrameEstimzation,
 *                                                  MapPointEstimation, OptimzeConfig)
 *                GSLAM/core/Optimizer.h:229       (Optimizer::optimize(BundleGraph&))
 *                GSLAM/core/Optimizer.h:202-207   (Optimizer::optimizePnP)
 *                GSLAM/core/SE3.h:100-103,257-287 (inverse / exp used by the pose update)
 *
 * Conventions
 *   - `*_dev` pointers are DEVICE (HBM) addresses; everything else is host memory.
 *   - Every call enqueues on the context's stream; `gh_ctx_sync` waits for it.  Host-buffer
 *     convenience entry points (`*_host`) synchronise before returning.
 *   - One gh_ctx per host thread; a ctx may be created on one thread and used on another
 *     (GSLAM constructs optimizers on one thread and calls them on Messenger workers).
 *   - No CPU fallback exists: every entry point fails with GH_ERR_HIP if the device is absent.
 */
#ifndef GSLAM_HIP_H_
#define GSLAM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gh_ctx gh_ctx;

typedef enum gh_status {
  GH_OK = 0,
  GH_ERR_ARG = 1,      /* bad argument (null pointer, negative size, capacity overflow) */
  GH_ERR_HIP = 2,      /* a HIP runtime call failed; see gh_last_error */
  GH_ERR_NOMEM = 3,    /* device or host allocation failed */
  GH_ERR_NUMERIC = 4,  /* BA: reduced system not positive definite even at maximum damping */
  GH_ERR_UNSUPPORTED = 5
} gh_status;

/* ------------------------------------------------------------------ context ---------- */
int gh_abi_version(void); /* bumps when a struct in this header changes layout or entry points change meaning; 2 since round 5
                             (GH_ABI_VERSION below is what this header describes: a host compares the two at start-up) */
#define GH_ABI_VERSION 2
gh_status gh_ctx_create(int device, gh_ctx** out);
void gh_ctx_destroy(gh_ctx* ctx);
const char* gh_last_error(const gh_ctx* ctx);
/* Use an externally owned hipStream_t (e.g. the caller's current stream).  NULL is the legacy
 * default stream (what torch's default stream is); gh_ctx_use_own_stream restores the private one. */
gh_status gh_ctx_set_stream(gh_ctx* ctx, void* hip_stream);
/* Frees what the context has grown for past calls (device scratch, pinned staging, the arenas of gh_ba_solve and of the
 * graph solvers): a single large graph would otherwise pin its high-water mark for the life of the context.  Waits for the
 * context's stream; everything is re-grown on demand; gh_ba_graph objects own their memory and are not touched. */
gh_status gh_ctx_trim(gh_ctx* ctx);
/* Linear solver of the reduced camera system in gh_ba_solve / gh_ba_graph_solve: GH_BA_SOLVER_AUTO (default) takes the
 * band solver (block cyclic reduction, chol_cr.hip) when every point is seen from cameras at most 31 indices apart, the
 * system has at least four superblocks of 64 * ceil((6 span + 5) / 64) columns, else the
 * dense MFMA factorisation; _DENSE forces the dense path (what BASELINE's C5 names); _BAND asks for the band solver and
 * falls back to dense when the graph is not a band.  The test is made on the solver's own camera order (gh_ba_camera_order),
 * not on the caller's.  LOOP CLOSURES: under _AUTO / _BAND the cameras that long-range points
 * tie to far-away cameras (at most 1024 of them) are numbered last inside the solver, and the system is solved as a band +
 * dense border (GH_BA_SOLVER_ARROW in gh_ctx_last_ba_solver; gh_arrow_solve_dev) instead of falling to the dense path.
 * The environment variable GSLAM_HIP_BA_SOLVER=dense|band|auto overrides it (A/B measurements). */
#define GH_BA_SOLVER_AUTO 0
#define GH_BA_SOLVER_DENSE 1
#define GH_BA_SOLVER_BAND 2
#define GH_BA_SOLVER_ARROW 3 /* reported only: band + dense border */
gh_status gh_ctx_set_ba_solver(gh_ctx* ctx, int solver);
/* What the last gh_ba_solve / gh_ba_graph_solve on this context used: GH_BA_SOLVER_DENSE or _BAND (0 before the first
 * solve); *band_tiles = 64-column tiles per superblock of the band solver (0 for dense), *cam_span = the largest distance
 * in camera indices between two observers of one point -- border cameras of the arrow ordering left out -- (the
 * half-bandwidth of the band part is 6 * span + 5).
 * Either pointer may be NULL. */
int gh_ctx_last_ba_solver(gh_ctx* ctx, int* band_tiles, int* cam_span);
/* The camera order behind that solve (round 6): *border_cams = cameras of the arrowhead border (0 for band / dense),
 * *reordered = 1 when the solver replaced the caller's camera order by its own bandwidth-reducing order (gh_ba_camera_order).
 * Returns 1 after a solve, 0 before the first.  Either pointer may be NULL. */
int gh_ctx_last_ba_order(gh_ctx* ctx, int* border_cams, int* reordered);
int gh_ctx_last_ba_border_points(gh_ctx* ctx); /* long-range points in the border of that solve (0: a camera border or none) */
gh_status gh_ctx_use_own_stream(gh_ctx* ctx);
void* gh_ctx_stream(gh_ctx* ctx);
gh_status gh_ctx_sync(gh_ctx* ctx);
/* Device facts as the runtime reports them. */
gh_status gh_device_info(gh_ctx* ctx, int* cu_count, int* clock_khz, size_t* hbm_bytes, char* name, int name_cap);

/* Raw device memory helpers, so C/C++ hosts need no HIP headers. */
gh_status gh_dev_alloc(gh_ctx* ctx, size_t bytes, void** out_dev);
gh_status gh_dev_free(gh_ctx* ctx, void* dev);
gh_status gh_dev_upload(gh_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
gh_status gh_dev_download(gh_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
gh_status gh_dev_memset(gh_ctx* ctx, void* dst_dev, int value, size_t bytes);

/* Live per-kernel timing with HIP events on the context's stream (bench.py's roofline). */
typedef struct gh_prof_entry {
  char name[48];
  uint64_t launches;
  double total_ms;
} gh_prof_entry;
gh_status gh_prof_enable(gh_ctx* ctx, int on); /* on=1 also clears accumulated entries */
/* Resolves pending events (synchronises), writes up to cap entries, returns count in *n. */
gh_status gh_prof_collect(gh_ctx* ctx, gh_prof_entry* out, int cap, int* n);

/* ------------------------------------------------------------------ BF matcher ------- */
/* 256-bit descriptors, 32 bytes each, dense rows (GImage N x 32, 8UC1).
 * For query i:  idx1[i] = argmin_j hamming(q_i, t_j), first minimum (lowest j) on ties;
 *               d1[i]   = that distance; d2[i] = min over j != idx1[i] (65535 if nt < 2).
 * nt == 0  ->  idx1 = -1, d1 = d2 = 65535.   Any nt: train sets beyond 65535 rows (a frame against a local map) are swept
 * in chunks and folded with the same strict '<', so the result is the one of a single sweep.  (The batched pair entries
 * below keep cap <= 65535: their rows are frames.) */
gh_status gh_bf_match_dev(gh_ctx* ctx, const uint8_t* q_dev, int nq, const uint8_t* t_dev, int nt,
                          int32_t* idx1_dev, uint16_t* d1_dev, uint16_t* d2_dev);
gh_status gh_bf_match_host(gh_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt,
                           int32_t* idx1, uint16_t* d1, uint16_t* d2);

/* Batched over frame pairs.  desc_dev: F frames x cap rows x 32 B; counts_dev[f] <= cap valid rows.
 * Pair p matches frame pair_q[p] (queries) against frame pair_t[p] (train).
 * Outputs are P x cap (rows >= counts[pair_q[p]] get idx1 = -1, d = 65535).
 * Dispatches on the amount of pair work: small batches run the popcount kernel (gh_bf_match_pairs_popc_dev), batches
 * that fill the chip the exact MFMA formulation (gh_bf_match_pairs_mfma_dev); the results are bit-identical. */
gh_status gh_bf_match_pairs_dev(gh_ctx* ctx, const uint8_t* desc_dev, const int32_t* counts_dev, int cap,
                                const int32_t* pair_q_dev, const int32_t* pair_t_dev, int npairs,
                                int32_t* idx1_dev, uint16_t* d1_dev, uint16_t* d2_dev);

/* The same matchers for descriptors of ANY width that is a multiple of 8 bytes (8 .. 256): the reference chooses its distance
 * by the descriptor's width -- GSLAM/core/Vocabulary.h:565-567: hamming32 for 32 bytes, hamming64 (:493-500) for 64,
 * hamming8x (:502-513) for other multiples of 8 -- and so do these entries (32 bytes run the kernels above, MFMA formulation
 * included).  Same outputs, same first-minimum rule; train sets beyond 65535 rows are swept in chunks for every width (round 6). */
gh_status gh_bf_match_bytes_dev(gh_ctx* ctx, const uint8_t* q_dev, int nq, const uint8_t* t_dev, int nt, int desc_bytes,
                                int32_t* idx1_dev, uint16_t* d1_dev, uint16_t* d2_dev);
gh_status gh_bf_match_bytes_host(gh_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int desc_bytes, int32_t* idx1,
                                 uint16_t* d1, uint16_t* d2); /* host arrays in and out, like gh_bf_match_host */
gh_status gh_bf_match_pairs_bytes_dev(gh_ctx* ctx, const uint8_t* desc_dev, const int32_t* counts_dev, int cap, int desc_bytes,
                                      const int32_t* pair_q_dev, const int32_t* pair_t_dev, int npairs, int32_t* idx1_dev,
                                      uint16_t* d1_dev, uint16_t* d2_dev);

/* The popcount formulation, always (north_star's contract kernel: v_xor + v_bcnt, GSLAM/core/Vocabulary.h:485-491 as
 * written; the kernel bench.py prices against the VALU issue ceiling). */
gh_status gh_bf_match_pairs_popc_dev(gh_ctx* ctx, const uint8_t* desc_dev, const int32_t* counts_dev, int cap,
                                     const int32_t* pair_q_dev, const int32_t* pair_t_dev, int npairs,
                                     int32_t* idx1_dev, uint16_t* d1_dev, uint16_t* d2_dev);

/* The same contract through an exact integer MFMA formulation, always: hamming = |a| + |b| - 2 |a & b| with |a & b| as a
 * dot product of the descriptors expanded to one byte per bit on v_mfma_i32_16x16x64_i8; three VALU instructions per
 * pair instead of nineteen.  Results are bit-identical to gh_bf_match_pairs_popc_dev. */
gh_status gh_bf_match_pairs_mfma_dev(gh_ctx* ctx, const uint8_t* desc_dev, const int32_t* counts_dev, int cap,
                                     const int32_t* pair_q_dev, const int32_t* pair_t_dev, int npairs,
                                     int32_t* idx1_dev, uint16_t* d1_dev, uint16_t* d2_dev);

/* Match masks (integer only):  keep[i] = d1 <= max_dist  &&  d1 * ratio_den < ratio_num * d2
 *                                        && (!cross_check || back_idx1[idx1[i]] == i).
 * ratio_num <= 0 disables the ratio test.  back_idx1_dev may be NULL when cross_check == 0. */
gh_status gh_match_mask_dev(gh_ctx* ctx, const int32_t* idx1_dev, const uint16_t* d1_dev,
                            const uint16_t* d2_dev, int nq, const int32_t* back_idx1_dev, int nt,
                            int max_dist, int ratio_num, int ratio_den, int cross_check,
                            uint8_t* keep_dev);

/* VALU ceiling probe: runs a register-resident xor+popcount chain and reports the measured
 * rate in 256-bit pair-equivalents per second (16 VALU ops each). */
gh_status gh_bf_valu_probe(gh_ctx* ctx, double* pairs_per_s);

/* Issue-rate probes per VALU instruction class of the ORB / matcher inner loops (register-only chains, 8 waves per
 * SIMD): wave-instructions per second over the whole device for op = 0, 1, ... ; GH_ERR_ARG past the last op.  name (may
 * be NULL) receives the instruction class.  bench.py reports them as the MEASURED issue ceiling of the VALU-bound
 * kernels instead of an assumed cycles-per-instruction figure. */
gh_status gh_valu_issue_probe(gh_ctx* ctx, int op, double* wave_insts_per_s, char* name, int name_cap);

/* Host only (no GPU needed): the 32-bit multiplier m with (n * m) >> 32 == n / d for EVERY n <= n_max, or 0 when no such
 * constant is guaranteed (the kernels then divide).  orb_fast_cells turns its tile id into (frame, tile row, tile column)
 * with two of these instead of two integer divisions per wave; exported so that the guarantee can be tested exhaustively. */
uint32_t gh_magic_div(uint32_t d, uint32_t n_max);

/* ------------------------------------------------------------------ ORB front end ---- */
/* Layout-identical to GSLAM::KeyPoint (GSLAM/core/Map.h:122-195, sizeof == 28). */
typedef struct gh_keypoint {
  float x, y;     /* level-0 pixel coordinates */
  float size;     /* 31 * scale^octave */
  float angle;    /* degrees, [0,360]: a multiple of 12 (30 orientation bins), or continuous (gh_orb_plan_set_steering) */
  float response; /* FAST corner score */
  int32_t octave;
  int32_t class_id; /* -1 */
} gh_keypoint;

/* Row-band restricted variant for stereo left-right matching (BASELINE configs[2]): train row j is a
 * candidate of query i only if |kps[i].y - kps[j].y| <= kps[i].size * band_per_size (single fp32 multiply).
 * kps_dev: F x cap gh_keypoint records parallel to desc_dev.  Outputs as gh_bf_match_pairs_dev. */
gh_status gh_bf_match_band_pairs_dev(gh_ctx* ctx, const uint8_t* desc_dev, const gh_keypoint* kps_dev,
                                     const int32_t* counts_dev, int cap, const int32_t* pair_q_dev,
                                     const int32_t* pair_t_dev, int npairs, float band_per_size,
                                     int32_t* idx1_dev, uint16_t* d1_dev, uint16_t* d2_dev);

typedef struct gh_orb_params {
  int32_t n_features;   /* K: total keypoint quota per frame (default 1000) */
  int32_t n_levels;     /* pyramid levels, 1..8 (default 8); scale factor fixed at 1.2 */
  int32_t ini_th_fast;  /* default 20 */
  int32_t min_th_fast;  /* default 7 */
} gh_orb_params;
void gh_orb_default_params(gh_orb_params* p);

typedef struct gh_orb_plan gh_orb_plan; /* owns pyramid + workspace for (w, h, batch) */
gh_status gh_orb_plan_create(gh_ctx* ctx, int width, int height, int max_batch,
                             const gh_orb_params* params, gh_orb_plan** out);
void gh_orb_plan_destroy(gh_orb_plan* plan);
/* Replace the plan's BRIEF test pattern: 256 tests x (ax, ay, bx, by) int8, relative to the keypoint, unrotated.  The
 * 30-bin steered look-up table is rebuilt from it (12-degree steps, round half away from zero).  Lets a deployer load
 * the canonical ORB / ORB-SLAM `bit_pattern_31_` (after gh_orb_plan_set_steering(plan, 1): it has points of radius 18.4) so
 * that descriptors are compatible with existing ORB vocabularies; the built-in default is a seeded pattern of this
 * repository (the canonical table is not available offline).  In the 30-bin mode every rotated point must stay inside the
 * +-13 px blurred patch (radius <= 13.49), in the continuous mode inside radius 19.49; otherwise GH_ERR_ARG. */
gh_status gh_orb_plan_set_pattern(gh_orb_plan* plan, const int8_t* pattern_256x4);
/* Orientation / steering mode.  0 (default): 30 orientation bins of 12 degrees with a precomputed rotated pattern per
 * bin, KeyPoint.angle a multiple of 12.  1: continuous -- angle = the fp32 polynomial arctangent of OpenCV's fastAtan2
 * (what ORB-SLAM's IC_Angle calls) on the same integer moments, every test point rotated by that angle and rounded to the
 * nearest pixel (ties to even, as cvRound): the steering of OpenCV / ORB-SLAM's computeOrbDescriptor.  In this mode
 * gh_orb_plan_set_pattern takes points up to radius 19.49 (the canonical ORB `bit_pattern_31_` reaches (-13, -13)), the
 * 7x7 blur reads up to 22 px from the keypoint and pixels beyond the image border are mirrored (BORDER_REFLECT_101, what
 * ORB-SLAM's padded pyramid holds there).  With the canonical pattern installed the descriptors are the ones ORB
 * vocabularies were trained on up to the image arithmetic (integer pyramid / FAST score / blur here, float there). */
gh_status gh_orb_plan_set_steering(gh_orb_plan* plan, int mode);
/* Keypoint distribution.  0 (default): 32 x 32 cells, in-cell rank, per-level quota by rank order (oracle steps 3-5; the
 * throughput path).  1: what ORB-SLAM's ORBextractor does (ComputeKeyPointsOctTree + DistributeOctTree) -- FAST per ~30 x 30
 * cell (W' = w - 32: nCols = W' / 30, wCell = ceil(W' / nCols)) with non-maximum suppression INSIDE the cell and the
 * cell's own fall-back from ini_th_fast to min_th_fast, every surviving corner a candidate (no per-cell cap), then per
 * (frame, level) the quadtree: nodes split into four until there are n_l of them (largest first in the last stage), the
 * strongest corner of each node kept.  Specification: oracle/orb_oracle.c steps 4' and 5' (bit-exact; ties that ORB-SLAM
 * leaves to heap addresses are fixed there).  Output rows per level are in (y, x) order.  Limits: levels up to 4096 x 4096,
 * per-level quota <= 2045.  The candidate lists hold the exact worst case (one strict maximum per 2 x 2 pixels of a cell) while
 * that fits 4 GB over max_batch frames (1080p: up to ~400 frames per call) and calls are asynchronous like the default mode's;
 * a plan whose budget had to cut the lists makes every call wait for its kernels, and a frame with more FAST maxima than a
 * list holds fails with GH_ERR_NOMEM, never silently.
 * Together with gh_orb_plan_set_steering(plan, 1) and the canonical pattern this is the ORB-SLAM extraction up to the image
 * arithmetic (integer pyramid and blur here, float resize there; KeyPoint.response = FAST score, OpenCV's is score - 1). */
gh_status gh_orb_plan_set_distribution(gh_orb_plan* plan, int mode);
/* Level geometry of the plan, for tests. */
gh_status gh_orb_plan_level(const gh_orb_plan* plan, int level, int* w, int* h, int* quota);
size_t gh_orb_plan_device_bytes(const gh_orb_plan* plan);

/* Extract from `batch` gray u8 frames resident in HBM (frame f at gray_dev + f*frame_stride,
 * rows `row_stride` bytes apart).  Outputs, each with capacity K = params.n_features per frame:
 *   kps_dev  batch x K gh_keypoint, desc_dev batch x K x 32 B, counts_dev batch int32.
 * A 16-byte aligned buffer (pointer, row_stride, frame_stride) is read in place, in 16-byte windows over the whole
 * row_stride x height extent of every frame: the padding of the rows must be allocated.  For a single frame whose
 * allocation ends at (height-1) * row_stride + width (an ROI view) pass batch = 1 and frame_stride = 0: the frame
 * is then copied row by row into the plan's own pyramid slab first. */
gh_status gh_orb_extract_dev(gh_orb_plan* plan, const uint8_t* gray_dev, int batch, size_t frame_stride,
                             int row_stride, gh_keypoint* kps_dev, uint8_t* desc_dev, int32_t* counts_dev);
/* Single host frame convenience (uploads, extracts, downloads, synchronises). */
gh_status gh_orb_extract_host(gh_orb_plan* plan, const uint8_t* gray, int row_stride, gh_keypoint* kps,
                              uint8_t* desc, int32_t* count);
/* Fixed-point luma (B*1868 + G*9617 + R*4899 + 8192) >> 14 for 3- or 4-channel BGR(A) input. */
gh_status gh_bgr_to_gray_dev(gh_ctx* ctx, const uint8_t* bgr_dev, int width, int height, int channels,
                             int src_row_stride, uint8_t* gray_dev, int dst_row_stride);
/* The same for n_frames frames laid out src_frame_stride / dst_frame_stride bytes apart (one launch). */
gh_status gh_bgr_to_gray_batch_dev(gh_ctx* ctx, const uint8_t* bgr_dev, int width, int height, int channels,
                                   int src_row_stride, size_t src_frame_stride, int n_frames, uint8_t* gray_dev,
                                   int dst_row_stride, size_t dst_frame_stride);

/* ---- Host-fed streaming extraction: frames arrive in HOST memory, results are wanted in HOST memory -------------------
 * What a GSLAM process sees: a dataset / camera thread hands over frames one by one (GSLAM/plugins/play/main.cpp:99-155)
 * and MapFrame::setKeyPoints wants std::vector<KeyPoint> + an N x 32 descriptor matrix on the host (GSLAM/core/Map.h:
 * 309-321).  A ring of `depth` slots of up to `chunk_frames` frames each; per submit ONE flat host-to-device DMA, the
 * extraction, and a pack kernel that writes per-frame offsets + only the valid records straight into pinned host memory --
 * on three HIP streams chained by events, so the copies of neighbouring chunks run under the kernels of this one.  No
 * torch, no Python: this is the entry a C++ host uses.  Frame layout is fixed per stream: `channels` interleaved u8
 * (1 gray, 3 BGR, 4 BGRA: luma as gh_bgr_to_gray_dev), rows row_stride bytes apart, frames frame_stride bytes apart.
 *   gh_orb_stream_staging  pinned staging block (chunk_frames x frame_stride bytes) of the slot the NEXT submit uses: a
 *                          producer that decodes / captures straight into it saves the host copy (submit with NULL).
 *   gh_orb_stream_submit   frames_host = NULL (the staging block) or the caller's own buffer (pinned: DMA in place;
 *                          pageable: the runtime stages it and the call returns when that is done).  Returns at once
 *                          otherwise; blocks only while the slot's previous ticket (ticket - depth) is still running.
 *                          The results of ticket t are overwritten by ticket t + depth: collect before that.
 *   gh_orb_stream_collect  waits for the ticket; *out points into the slot's pinned result block (valid until the slot
 *                          is reused): frame f owns records offsets[f] .. offsets[f + 1] - 1 of kps / desc.
 *                          Thread-safe against a concurrent submit (producer / consumer threads).
 *   gh_orb_stream_poll     *ready = 1 when collect would not block. */
typedef struct gh_orb_stream gh_orb_stream;
typedef struct gh_orb_stream_result {
  int32_t n_frames;
  const int32_t* offsets;   /* n_frames + 1 */
  const gh_keypoint* kps;   /* offsets[n_frames] records, frame after frame */
  const uint8_t* desc;      /* offsets[n_frames] x 32 B */
  float gpu_ms;             /* first byte on the link -> last result byte in host memory, for this ticket */
} gh_orb_stream_result;
gh_status gh_orb_stream_create(gh_ctx* ctx, int width, int height, int channels, int row_stride, size_t frame_stride,
                               int chunk_frames, int depth, const gh_orb_params* params, gh_orb_stream** out);
void gh_orb_stream_destroy(gh_orb_stream* stream);
/* The extraction plan behind the stream (owned by it), for gh_orb_plan_set_pattern / gh_orb_plan_set_steering: call them
 * while no ticket is outstanding. */
gh_orb_plan* gh_orb_stream_plan(gh_orb_stream* stream);
gh_status gh_orb_stream_staging(gh_orb_stream* stream, uint8_t** host_pinned);
gh_status gh_orb_stream_submit(gh_orb_stream* stream, const uint8_t* frames_host, int n_frames, int64_t* ticket);
gh_status gh_orb_stream_poll(gh_orb_stream* stream, int64_t ticket, int* ready);
gh_status gh_orb_stream_collect(gh_orb_stream* stream, int64_t ticket, gh_orb_stream_result* out);
/* Pinned (page-locked, DMA-able) host memory for callers without HIP headers. */
gh_status gh_host_alloc_pinned(gh_ctx* ctx, size_t bytes, void** out_host);
gh_status gh_host_free_pinned(gh_ctx* ctx, void* host);

/* Test-only branch census.  enable != 0 makes the following extractions of this plan count how often the rarely taken
 * paths run; out16 (may be NULL) receives and clears the 16 counters accumulated so far:
 *   [0] cells processed  [1] cells with > 64 scored pixels (list branch)  [2] cells that wrote overflow entries (8th..)
 *   [3] cells at the 32-entry cap  [4] cells with > 32 candidates (ranks dropped)  [5] cells where a strong corner
 *   silenced weaker ones  [6] longest pass-1 queue  [7] most scored pixels in a cell  [8] level selections cut by the
 *   quota  [9] ... with the cut-off bin split among ties  [10] cells whose overflow entries the selection read
 *   [11] streaming selection variant used  [12] zero-filled output rows  [13] level selections below quota.
 * tests/test_orb_adversarial_gpu.py uses it to prove that its inputs reach those branches. */
gh_status gh_orb_plan_debug_counters(gh_orb_plan* plan, int enable, uint32_t* out16);
/* Debug/test access: copy pyramid level `level` of batch slot `slot` to host (w*h bytes, dense). */
gh_status gh_orb_debug_level(gh_orb_plan* plan, int slot, int level, uint8_t* out_host);

/* Deterministic synthetic frames (integer procedural texture, seed = base_seed + frame index);
 * bit-identical to oracle/synth.c.  Test/bench input generator, resident in HBM. */
gh_status gh_synth_frames_dev(gh_ctx* ctx, uint8_t* gray_dev, int width, int height, int row_stride,
                              size_t frame_stride, int first_frame, int n_frames, uint32_t base_seed);

/* ------------------------------------------------------------------ multi-GPU exchange */
/* Frames shard over the GPUs of one node (one process per GPU); the only data-path exchange is an
 * all-gather of the per-frame records after extraction and, optionally, of the match rows after matching.  The
 * reference has no transport (GSLAM's Messenger is in-process, GSLAM/core/Messenger.h:455-468,687-715), so these entry
 * points are what a multi-GPU C++ host adds; no Python / torch is involved.
 *   RCCL transport: rank 0 calls gh_comm_unique_id and hands the 128 bytes to the other ranks by any out-of-band means
 *     (file, socket, MPI, torch.distributed broadcast), then every rank calls gh_comm_create_rccl.  ncclAllGather over
 *     xGMI on the communicator's own stream.
 *   IPC transport (same node): every rank calls gh_comm_create_ipc with the same rendezvous name; peers' gathered
 *     buffers are mapped through HIP IPC and each rank pushes its slice into all of them.  Also works when several ranks
 *     share one GPU (RCCL refuses that).  Asynchronous like RCCL: the ranks' streams order themselves through flags in
 *     the rendezvous segment that bounded one-wave kernels set and poll; no host thread waits during an exchange.
 * A gather is ordered after the work already enqueued on the context's stream and runs beside it; gh_comm_wait orders
 * the context's stream after the gather and returns at once.  A rank that gave up waiting for a peer (timeout
 * GSLAM_HIP_COMM_TIMEOUT_S, default 60) shows in gh_comm_status and in the next gh_allgather* call of every rank.
 * Gathered buffers must come from gh_comm_buffer (a collective call). */
typedef struct gh_comm gh_comm;
gh_status gh_comm_unique_id(uint8_t id_out[128]);
gh_status gh_comm_create_rccl(gh_ctx* ctx, int rank, int world, const uint8_t unique_id[128], gh_comm** out);
gh_status gh_comm_create_ipc(gh_ctx* ctx, int rank, int world, const char* rendezvous_name, gh_comm** out);
void gh_comm_destroy(gh_comm* comm);
int gh_comm_rank(const gh_comm* comm);
int gh_comm_world(const gh_comm* comm);
gh_status gh_comm_buffer(gh_comm* comm, size_t bytes_per_rank, void** gathered_dev); /* world x bytes_per_rank */
gh_status gh_allgather(gh_comm* comm, const void* send_dev, void* gathered_dev, size_t bytes_per_rank);
/* This rank's `frames` frames of gh_orb_extract_dev output into slot `rank` of the gathered arrays
 * (world x frames x cap records).  kps_dev / g_kps_dev may both be NULL when only descriptors are needed. */
gh_status gh_allgather_features(gh_comm* comm, int frames, int cap, const gh_keypoint* kps_dev, const uint8_t* desc_dev,
                                const int32_t* counts_dev, gh_keypoint* g_kps_dev, uint8_t* g_desc_dev,
                                int32_t* g_counts_dev);
/* `rows` match rows of this rank (cap entries each) into slot `rank`; d1 / d2 (and their gathered arrays) may be NULL. */
gh_status gh_allgather_matches(gh_comm* comm, int rows, int cap, const int32_t* idx1_dev, const uint16_t* d1_dev,
                               const uint16_t* d2_dev, int32_t* g_idx1_dev, uint16_t* g_d1_dev, uint16_t* g_d2_dev);
gh_status gh_comm_wait(gh_comm* comm);
gh_status gh_comm_status(gh_comm* comm); /* GH_OK, or GH_ERR_HIP once any rank abandoned an exchange */

/* ------------------------------------------------------------------ BoW transform ---- */
/* GSLAM::Vocabulary image -> BoW vector (GSLAM/core/Vocabulary.h:1558-1621, per-feature descent :1695-1736,
 * normalisation :386-408), for 32-byte binary descriptors.  The vocabulary is given in the in-memory layout of the
 * reference's .gbow file (:1843-1932): nodes = {uint32 childNum; float weight}[nnodes], node_desc = nnodes x 32 B,
 * children of node p at p*k+1 .. p*k+childNum.  weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY;
 * scoring: 0 L1_NORM, 1 L2_NORM, 2 CHI_SQUARE, 3 KL, 4 BHATTACHARYYA, 5 DOT_PRODUCT (enum order of the reference). */
typedef struct gh_bow_vocab gh_bow_vocab;
gh_status gh_bow_vocab_create(gh_ctx* ctx, int k, int L, int weighting, int scoring, uint32_t nnodes,
                              const void* nodes, const uint8_t* node_desc, gh_bow_vocab** out);
/* Binary descriptors of any multiple of 8 bytes (Vocabulary.h:560-568: 32 -> hamming32, 64 -> hamming64, others ->
 * hamming8x); gh_bow_vocab_create is desc_bytes = 32.  The transforms take descriptors of the vocabulary's width. */
gh_status gh_bow_vocab_create_bytes(gh_ctx* ctx, int k, int L, int weighting, int scoring, uint32_t nnodes,
                                    const void* nodes, const uint8_t* node_desc, int desc_bytes, gh_bow_vocab** out);
/* Float vocabularies (SIFT / SURF style): node_desc n_nodes x dims floats, dims a multiple of 8 -- the dimensions for which
 * the reference's DistanceFactory picks l2generic whatever ISA it was built for (Vocabulary.h:550-560,569-578): squared L2,
 * float accumulation in index order, no fused multiply-add.  The transforms then take descriptors of dims floats. */
gh_status gh_bow_vocab_create_f32(gh_ctx* ctx, int k, int L, int weighting, int scoring, uint32_t nnodes, const void* nodes,
                                  const float* node_desc, int dims, gh_bow_vocab** out);
void gh_bow_vocab_destroy(gh_bow_vocab* vocab);
/* Batched over images: desc_dev n_images x cap x 32 B, counts_dev (may be NULL = cap rows each).  Per feature:
 * word id, word weight, node id at level L - levelsup (0xFFFFFFFF / 0 / 0xFFFFFFFF for rows >= count).  Per image:
 * bow_word (ascending, 0xFFFFFFFF padded) / bow_val (normalised as the reference does) / bow_n.  cap <= 16384. */
gh_status gh_bow_transform_dev(gh_bow_vocab* vocab, const uint8_t* desc_dev, const int32_t* counts_dev, int cap,
                               int n_images, int levelsup, uint32_t* word_dev, float* weight_dev, uint32_t* node_dev,
                               uint32_t* bow_word_dev, float* bow_val_dev, int32_t* bow_n_dev);
gh_status gh_bow_transform_host(gh_bow_vocab* vocab, const uint8_t* desc, int n, int levelsup, uint32_t* word,
                                float* weight, uint32_t* node, uint32_t* bow_word, float* bow_val, int32_t* bow_n);

/* Batched GSLAM::Vocabulary::score(a, b) = m_scoring_object->score(a, b) (GSLAM/core/Vocabulary.h:691-979, one class
 * per ScoringType): every query BowVector against every database BowVector, scores_dev[q * n_db + j] = score(query q,
 * database j) as the reference's double.  Vectors are in the padded layout gh_bow_transform_dev writes (word ids
 * ascending, *_n valid entries of a row of capacity cap_*), so a block of images transformed on the GPU is scored with
 * no host copy.  L1 / L2 / chi-square / Bhattacharyya / dot product are bit-identical to the reference (same float
 * terms, same ascending-id double accumulation); KL goes through logf and matches to 1e-6 relative. */
gh_status gh_bow_score_dev(gh_ctx* ctx, int scoring, const uint32_t* q_word_dev, const float* q_val_dev,
                           const int32_t* q_n_dev, int n_q, int cap_q, const uint32_t* db_word_dev,
                           const float* db_val_dev, const int32_t* db_n_dev, int n_db, int cap_db, double* scores_dev);
/* One query against n_db host vectors in CSR form (db_off[n_db + 1] offsets into db_word / db_val): the loop-closure
 * candidate scan of a detector that keeps std::map BowVectors on the host. */
gh_status gh_bow_score_host(gh_ctx* ctx, int scoring, const uint32_t* q_word, const float* q_val, int q_n,
                            const uint32_t* db_word, const float* db_val, const int64_t* db_off, int n_db,
                            double* scores);

/* ------------------------------------------------------------------ undistortion ----- */
/* GSLAM::Undistorter::undistort / undistortFast (GSLAM/core/Undistorter.h:206-348) with the remap tables its
 * prepareReMap builds on the host (:120-203): per output pixel remapX, remapFast, remapIdx[4], remapCoef[4].
 * u8 images, 1 / 3 (/ 4 for `fast`) interleaved channels, dense rows.  Pixels the reference leaves unwritten are 0. */
typedef struct gh_undist_plan gh_undist_plan;
gh_status gh_undist_plan_create(gh_ctx* ctx, int w_in, int h_in, int w_out, int h_out, const float* remapX,
                                const int32_t* remapFast, const int32_t* remapIdx, const float* remapCoef,
                                gh_undist_plan** out);
void gh_undist_plan_destroy(gh_undist_plan* plan);
gh_status gh_undistort_dev(gh_undist_plan* plan, const uint8_t* img_dev, int channels, int batch,
                           size_t in_frame_stride, uint8_t* out_dev, size_t out_frame_stride, int fast);
gh_status gh_undistort_host(gh_undist_plan* plan, const uint8_t* img, int channels, uint8_t* out, int fast);

/* ------------------------------------------------------------------ RANSAC estimation - */
/* Robust model fitting with an inlier mask, behind GSLAM::Estimator (GSLAM/core/Estimator.h:100-169; interface only in
 * the reference).  Deterministic: 2048 hypotheses drawn from `seed` (gh_ransac_estimate_conf: the prefix of them that
 * the requested confidence asks for); winner = most correspondences with squared error
 * <= threshold^2, lowest hypothesis index on ties.  model_out must hold 12 doubles; mask_out (n bytes, may be NULL) gets
 * 1 for inliers; *inliers_out = 0 means no model.
 *   0 homography   findHomography      src, dst n x 2; model 9 row-major, h33 = 1; forward transfer error
 *   1 affine 2D    findAffine2D        src, dst n x 2; model 6 = 2 x 3
 *   2 fundamental  findFundamental     src, dst n x 2; dst^T F src = 0; model 9; Sampson error; rank not enforced
 *   3 affine 3D    findAffine3D        src, dst n x 3; model 12 = 3 x 4
 *   4 essential    findEssentialMatrix src, dst n x 2 NORMALISED image coordinates; hypotheses as (2), the winner is
 *                                      projected onto the essential manifold (singular values s, s, 0); model 9
 *   5 SIM3         findSIM3            src, dst n x 3; dst ~ s R src + t by Horn's closed form on 3 pairs;
 *                                      model 8 = [qx qy qz qw tx ty tz s]
 *   6 plane        findPlane           src n x 3 (dst ignored, pass src); model 4 = [nx ny nz d], n . x + d = 0, |n| = 1
 *   7 PnP          findPnP             src n x 3 object points, dst n x 2 normalised image points; 6-point DLT
 *                                      hypotheses (coplanar object points are degenerate); model 12 = [R row-major | t],
 *                                      X_cam = R X + t.  Refine on the inliers with gh_ba_pnp. */
#define GH_MODEL_HOMOGRAPHY 0
#define GH_MODEL_AFFINE2D 1
#define GH_MODEL_FUNDAMENTAL 2
#define GH_MODEL_AFFINE3D 3
#define GH_MODEL_ESSENTIAL 4
#define GH_MODEL_SIM3 5
#define GH_MODEL_PLANE 6
#define GH_MODEL_PNP 7
gh_status gh_ransac_estimate(gh_ctx* ctx, int model, const double* src, const double* dst, int n, double threshold,
                             uint64_t seed, double* model_out, uint8_t* mask_out, int* inliers_out);
/* The same with GSLAM::Estimator's `confidence` argument honoured (Estimator.h:100-169 pass threshold AND confidence):
 * the winner is the best among the hypotheses a SEQUENTIAL adaptive RANSAC with the same draws would have examined -- in
 * index order, after every strict improvement of the best inlier count c the needed number of hypotheses becomes
 * ceil(log(1 - confidence) / log(1 - (c / n)^s)) (s = sample size of the model), and the walk ends when it is reached.
 * *hypotheses_used_out (may be NULL) reports that number.  confidence <= 0 or >= 1: all 2048 (== gh_ransac_estimate). */
gh_status gh_ransac_estimate_conf(gh_ctx* ctx, int model, const double* src, const double* dst, int n, double threshold,
                                  double confidence, uint64_t seed, double* model_out, uint8_t* mask_out,
                                  int* inliers_out, int* hypotheses_used_out);
/* The same with GSLAM::EstimatorMethod's sampling flag (Estimator.h:86-89):
 *   GH_SAMPLE_RANSAC  as gh_ransac_estimate_conf.
 *   GH_SAMPLE_LMEDS   least median of squares over the same 2048 hypotheses: the winner has the smallest MEDIAN squared error
 *                     (the element of rank n / 2, ascending; lowest hypothesis index on ties), `confidence` is not used; the
 *                     mask holds the correspondences within max(threshold, 2.5 * 1.4826 * (1 + 5 / (n - s)) * sqrt(median))
 *                     (Rousseeuw's robust standard deviation, the rule of OpenCV's LMEDS); no model when even the best
 *                     median is undefined.
 *   GH_SAMPLE_NONE    NOSAMPLE: no hypotheses -- the algebraic least-squares model of ALL correspondences (normal equations of
 *                     the minimal solver's rows for H / affine, smallest eigenvector of A^T A for F / E / PnP, Horn over all
 *                     pairs for SIM3, principal plane); mask / inliers by `threshold`; *hypotheses_used_out = 1. */
#define GH_SAMPLE_RANSAC 0
#define GH_SAMPLE_LMEDS 1
#define GH_SAMPLE_NONE 2
gh_status gh_ransac_estimate_ex(gh_ctx* ctx, int model, const double* src, const double* dst, int n, double threshold,
                                double confidence, uint64_t seed, int sampling, double* model_out, uint8_t* mask_out,
                                int* inliers_out, int* hypotheses_used_out);
/* Midpoint triangulation, one correspondence per thread (GSLAM::Estimator::trianglate, Estimator.h:164-168): the point of
 * the REFERENCE frame closest to the two rays ref_dir and cur_dir (camera.UnProject of the two pixels), with
 * X_cur = T_ref2cur X_ref, pose = [qx qy qz qw tx ty tz].  pose_stride = 7: one pose per correspondence; 0: one pose for
 * all.  ok[i] = 0 when the rays are parallel or the point lies behind either camera. */
gh_status gh_triangulate(gh_ctx* ctx, const double* ref2cur_pose, int pose_stride, const double* ref_dir,
                         const double* cur_dir, int n, double* ref_points, uint8_t* ok);

/* ------------------------------------------------------------------ bundle adjustment - */
/* DOF bits follow GSLAM::KeyFrameEstimzationDOF (GSLAM/core/Optimizer.h:70-84). */
#define GH_KF_X 1
#define GH_KF_Y 2
#define GH_KF_Z 4
#define GH_KF_RX 8
#define GH_KF_RY 16
#define GH_KF_RZ 32
#define GH_KF_SE3 63
#define GH_KF_SCALE 64 /* UPDATE_KF_SCALE: only the pose-graph solver and gh_align_sim3 have a scale to update */
#define GH_KF_SIM3 127

typedef struct gh_ba_problem {
  int32_t n_cams, n_points, n_obs;
  /* T_wc (camera -> world) per keyframe: 7 doubles [qx,qy,qz,qw,tx,ty,tz]  (in/out) */
  double* cam_pose;
  const int32_t* cam_dof; /* GH_KF_* bitmask per camera; 0 = fixed */
  double* point_xyz;      /* 3 doubles per point, world frame  (in/out) */
  const uint8_t* point_free; /* 1 = optimise, 0 = fixed; NULL = all free */
  const int32_t* obs_cam;
  const int32_t* obs_point;
  const double* obs_xy;   /* 2 doubles per observation: normalised (x,y) on the z=1 plane */
  const double* obs_info; /* 4 doubles (row-major 2x2) per observation or NULL = identity */
} gh_ba_problem;

typedef struct gh_ba_options {
  double huber_delta;        /* OptimzeConfig::projectErrorHuberThreshold; <= 0 disables */
  int32_t max_iterations;    /* OptimzeConfig::maxIterations */
  double initial_radius;     /* 1e4 */
  double function_tolerance; /* 1e-6 */
  double gradient_tolerance; /* 1e-10 */
  double min_relative_decrease; /* 1e-3 */
  int32_t verbose;
  int32_t deterministic;     /* 1 (default): reproducible sums -- gh_ba_solve: ordered segmented Schur accumulation; gh_graph_solve:
                                the landmark part pre-rounded so that its atomics add exactly (bit-identical from run to run) on
                                the DENSE path (two more n x lda accumulators, one fold pass per iteration); graphs large enough
                                for the block-sparse Cholesky (gh_graph_summary: sparse) keep plain atomics -- the option is
                                IGNORED there and the last bits vary between runs;
                                0: plain f64 atomics (faster assembly, last bits vary between runs) */
} gh_ba_options;
void gh_ba_default_options(gh_ba_options* o);

#define GH_BA_MAX_TRACE 512
typedef struct gh_ba_summary {
  int32_t iterations;      /* LM iterations executed (accepted + rejected) */
  int32_t accepted;
  int32_t termination;     /* 0 max_iterations, 1 function_tolerance, 2 gradient_tolerance, 3 failure */
  double initial_cost, final_cost;
  double solve_ms_total;   /* host time from the dense reduced-camera solve to the candidate cost (per-iteration sync) */
  double total_ms;
  int32_t trace_len;
  double trace_cost[GH_BA_MAX_TRACE];   /* candidate cost evaluated at each iteration */
  double trace_radius[GH_BA_MAX_TRACE]; /* trust-region radius used at each iteration */
  uint8_t trace_accepted[GH_BA_MAX_TRACE];
} gh_ba_summary;

/* Host-struct entry point (what the Optimizer plugin calls): uploads, runs LM on the GPU,
 * writes cam_pose / point_xyz back in place. */
gh_status gh_ba_solve(gh_ctx* ctx, gh_ba_problem* problem, const gh_ba_options* options,
                      gh_ba_summary* summary);

/* Device-resident graph across solves.  A windowed SLAM back end calls Optimizer::optimize (GSLAM/core/Optimizer.h:229)
 * every few keyframes on a graph whose TOPOLOGY (which camera observes which point) changes far less often than its
 * values; gh_ba_solve rebuilds the index lists, the Schur pair lists and the chunk tables and uploads every array each
 * time (1.1-2.4 ms at C4 against 12 x 1.0 ms of iterations).  A gh_ba_graph keeps all of that in HBM:
 *   gh_ba_graph_create   validates + uploads the problem, builds the lists and tables (what gh_ba_solve does before its
 *                        first iteration).  options->deterministic is fixed here (it decides whether pair lists exist).
 *   gh_ba_graph_update   new VALUES for the same topology; any pointer may be NULL = unchanged.  obs_info / point_free
 *                        can only be updated if the graph was created with them.
 *   gh_ba_graph_solve    LM iterations on the resident state (same algorithm, same results as gh_ba_solve on the same
 *                        values); the state stays on the device: consecutive solves continue from the last accepted one.
 *   gh_ba_graph_read     cam_pose (n_cams x 7) / point_xyz (n_points x 3) back to the host; either may be NULL.
 * The Optimizer plugin caches one graph keyed on the topology of the BundleGraph it is given. */
typedef struct gh_ba_graph gh_ba_graph;
gh_status gh_ba_graph_create(gh_ctx* ctx, const gh_ba_problem* problem, const gh_ba_options* options, gh_ba_graph** out);
void gh_ba_graph_destroy(gh_ba_graph* graph);
gh_status gh_ba_graph_update(gh_ba_graph* graph, const double* cam_pose, const double* point_xyz, const double* obs_xy,
                             const double* obs_info, const int32_t* cam_dof, const uint8_t* point_free);
gh_status gh_ba_graph_solve(gh_ba_graph* graph, const gh_ba_options* options, gh_ba_summary* summary);
gh_status gh_ba_graph_read(gh_ba_graph* graph, double* cam_pose, double* point_xyz);

/* Pose-only (motion-only BA) on 3D-2D matches: pose = T_wc 7 doubles in/out;
 * information_out (36 doubles, row-major 6x6 J^T J at the solution) may be NULL. */
gh_status gh_ba_pnp(gh_ctx* ctx, const double* points_xyz, const double* obs_xy, int n, double* pose,
                    int dof, const gh_ba_options* options, double* information_out, gh_ba_summary* summary);

/* Optimizer::magin (GSLAM/core/Optimizer.h:230-232, "Convert bundle graph to pose graph"; the reference ships the
 * declaration only): one SE3 edge per pair of cameras (first < second) that share at least min_shared points, with the 6x6
 * information (row-major, [v, w] order of SE3::exp) of the second camera's pose relative to the first held fixed -- the
 * Schur complement of the two-view problem over the shared points at the current estimate, Huber-weighted like
 * gh_ba_solve (specification: oracle_ba_marginalize in oracle/ba_oracle.c).  The measurement of the edge is
 * T_first^-1 T_second of the current poses (the caller forms it).  Edges come sorted by (first, second); *n_edges is
 * always the number found; with max_edges == 0 nothing else is written (size query), with 0 < max_edges < *n_edges the
 * call fails.  edge_shared (may be NULL) = shared points per edge. */
/* The camera order gh_ba_solve / gh_ba_graph_create would use INSIDE the solver for this graph -- host code, no device work, the
 * context plays no part (tests, tools, a caller that wants to know whether its graph reaches the band solver).  BundleGraph::keyframes
 * is a plain vector (GSLAM/core/Optimizer.h:116-119,150-157): nothing says a caller fills it in trajectory order, so the solver
 * orders the reduced camera system for itself, as Ceres' SPARSE_SCHUR does behind Optimizer::optimize (Optimizer.h:229):
 *   the caller's order when it is a band (every point's observers at most 31 camera indices apart) or a band + border as it stands;
 *   otherwise a weighted maximum-adjacency order of the camera co-visibility graph (from a sample of the points), refined by two
 *   barycentre sweeps, then the border of the cameras that long-range points tie to far-away ones.
 * perm_out[n_cams]: old camera of every new position (the identity when the caller's order stands); *n_border: cameras of the dense
 * border at the end of the order; *cam_span: largest distance in new positions between two band observers of one point (the solver
 * takes the band / arrowhead path when 6 * span + 5 <= 192 and at least four superblocks remain); *reordered: 1 when the
 * bandwidth-reducing order was applied.  *n_border_points (NULL: the camera-border choice only): when the long-range POINTS make
 * the smaller border (3 unknowns each against 6 per far camera) they are kept out of the Schur complement and solved for together
 * with the cameras -- then *n_border = 0, no camera moves for them, and *cam_span leaves them out.
 * problem: only n_cams, n_points, n_obs, obs_cam, obs_point are read.
 * GH_ERR_ARG for null pointers or indices out of range.  The environment variable GSLAM_HIP_BA_REORDER=0 keeps gh_ba_solve on the
 * caller's order (A/B measurements); this function always reports the order the default would choose. */
gh_status gh_ba_camera_order(const gh_ba_problem* problem, int32_t* perm_out, int32_t* n_border, int32_t* cam_span,
                             int32_t* reordered, int32_t* n_border_points);

gh_status gh_ba_marginalize(gh_ctx* ctx, const gh_ba_problem* problem, double huber_delta, int32_t min_shared,
                            int32_t max_edges, int32_t* edge_first, int32_t* edge_second, int32_t* edge_shared,
                            double* edge_info, int32_t* n_edges);

/* ---- Pose graph: Optimizer::optimize(BundleGraph&) with se3Graph / sim3Graph / gpsGraph edges ---------------------------
 * Data contract GSLAM/core/Optimizer.h:127-148,162-167: SE3Edge {firstId, secondId, measurement SE3_12 := SE3_1^-1 SE3_2,
 * information 6x6}, SIM3Edge {.., SIM3_12 := SIM3_1^-1 SIM3_2, 7x7}, GPSEdge {frameId, SE3_gps := SE3_frame, 6x6};
 * keyframes are SIM3 T_wc with KeyFrameEstimzationDOF masks incl. UPDATE_KF_SCALE.  The reference ships no solver; the
 * specification is the header of oracle/pg_oracle.c: residual = log(M^-1 * S_first^-1 * S_second) (SE3 edges on the (R, t)
 * part, GPS edges log(M^-1 * T_frame)), cost 1/2 sum r^T Lambda r, update S <- S * SIM3::exp(delta) restricted to the dof
 * mask, central-difference Jacobians, Levenberg-Marquardt on the dense normal equations with the trust-region strategy of
 * gh_ba_solve (options / summary are shared; huber_delta and deterministic are not used: the assembly is always
 * deterministic).  Arrays: frame_sim3 n_frames x 8 [qx qy qz qw tx ty tz s] (in/out); *_meas n x 7 (SE3, GPS:
 * [qx qy qz qw tx ty tz]) or n x 8 (SIM3); *_info row-major 6x6 / 7x7 per edge, NULL = identity. */
typedef struct gh_pg_problem {
  int32_t n_frames;
  double* frame_sim3;
  const int32_t* frame_dof; /* GH_KF_* bits; 0 = fixed (at least one frame should be: the gauge) */
  int32_t n_se3;
  const int32_t *se3_first, *se3_second;
  const double *se3_meas, *se3_info;
  int32_t n_sim3;
  const int32_t *sim3_first, *sim3_second;
  const double *sim3_meas, *sim3_info;
  int32_t n_gps;
  const int32_t* gps_frame;
  const double *gps_meas, *gps_info;
} gh_pg_problem;
gh_status gh_pg_solve(gh_ctx* ctx, gh_pg_problem* problem, const gh_ba_options* options, gh_ba_summary* summary);

/* The GENERAL BundleGraph of Optimizer::optimize (GSLAM/core/Optimizer.h:150-172,229): the keyframes and pose-graph edges
 * of gh_pg_problem PLUS landmarks -- XYZ map points (mappoints, :113-114,155) and inverse-depth points (invDepths,
 * :106-111,152-153: host keyframe, anchor in the host camera, idepth) -- with their pinhole observations (BundleEdge,
 * :121-125,160-161).  This is the path for what gh_ba_solve (SE3 cameras + XYZ points only, the fast path) does not take:
 * inverse-depth points, keyframes with a free scale, pose-graph edges mixed with observations.  The reference holds no
 * implementation: specification and cross-checks in oracle/graph_oracle.c (parity unpinned).  options->huber_delta is
 * OptimzeConfig::projectErrorHuberThreshold (0 = no robust kernel).  Landmarks are eliminated by a Schur complement, the
 * keyframe system (7 n_frames square) is dense.  In/out: pg.frame_sim3, xyz, idp_rho.
 *   xyz n_xyz x 3 world points, xyz_free NULL = all free;  idp_host / idp_anchor (n x 3, pinhole anchors (x, y, 1)) /
 *   idp_rho > 0 / idp_free;  obs_kind 0 = XYZ point, 1 = inverse-depth point; obs_point indexes the respective array;
 *   obs_xy n_obs x 2 normalised image coordinates; obs_info n_obs x 4 row-major 2x2 or NULL.
 * Sphere projection: the residual is the predicted bearing in the tangent plane of the measured one (2-vector, same
 * information / Huber), dropped on the opposite hemisphere. */
typedef struct gh_graph_problem {
  gh_pg_problem pg;
  int32_t n_xyz;
  double* xyz;
  const uint8_t* xyz_free;
  int32_t n_idp;
  const int32_t* idp_host;
  const double* idp_anchor;
  double* idp_rho;
  const uint8_t* idp_free;
  int32_t n_obs;
  const int32_t *obs_kind, *obs_point, *obs_frame;
  const double *obs_xy, *obs_info;
  int32_t projection;        /* 0 = PROJECTION_PINHOLE (obs_xy), 1 = PROJECTION_SPHERE (obs_bearing; Optimizer.h:58-61,175) */
  const double* obs_bearing; /* n_obs x 3 unit bearings, sphere only: anchors are unit bearings too, idepth = 1 / range */
  /* Camera self-calibration -- BundleGraph::camera + cameraDOF (Optimizer.h:86-100,169-171: "Invalid camera indicates idea
   * camera").  intrinsics = fx fy cx cy k1 k2 p1 p2 k3 of GSLAM's OpenCV camera model (GSLAM/core/Camera.h:386-407; a
   * pinhole camera :213-227 has k = p = 0), in / out; NULL = ideal camera.  With intrinsics, obs_xy are PIXELS, the residual
   * is Project(Y) - obs_xy (Huber threshold and obs_info in pixel units), and the parameters whose bit is set in
   * intrinsics_free (bit i = parameter i; CameraEstimationDOF: FOCAL -> bits 0-1, CENTER -> 2-3, K1 4, K2 5, P1 6, P2 7,
   * K3 8) are unknowns of the same Levenberg-Marquardt problem (a 9-row block behind the keyframes in the reduced system).
   * Anchors of inverse-depth points stay normalised.  Pinhole projection only (GH_ERR_ARG with projection = 1). */
  double* intrinsics;
  int32_t intrinsics_free;
} gh_graph_problem;
gh_status gh_graph_solve(gh_ctx* ctx, gh_graph_problem* problem, const gh_ba_options* options, gh_ba_summary* summary);

/* 3-D alignment dst_k ~ s R src_k + t over n correspondences (n x 3 doubles each), Horn's closed form: what
 * Optimizer::optimizeICP (3D-3D correspondences, Optimizer.h:210-217) and Optimizer::fitSim3 (translations of two
 * synchronised trajectories, :220-225) compute.  dof & GH_KF_SCALE decides whether s is estimated (else s = 1).
 * sim3_out 8 doubles [qx qy qz qw tx ty tz s]; information_out (49 doubles, may be NULL) = J^T J of the residuals
 * dst - S exp(delta) src at the solution, rows / columns of masked dof zero; ssq_out (may be NULL) = sum of squared
 * residuals; *ok_out = 0 for fewer than 3 or degenerate (coincident) correspondences. */
gh_status gh_align_sim3(gh_ctx* ctx, const double* src, const double* dst, int n, int dof, double* sim3_out,
                        double* information_out, double* ssq_out, int* ok_out);

/* Dense SPD solve used by the reduced camera system, exposed for tests and the C5 bench:
 * A_dev is n x n column-major/symmetric (lower triangle read, overwritten by L), b_dev in, x out.
 * *info: 0 = ok; 1 .. n = the matrix is not positive definite (first bad 64-column block, 1-based first column);
 * > n = a single-launch kernel (n <= ~3300 with lda % 16 == 0: factorisation; n <= 8192: back-substitution) could not
 * get all its workgroups resident in bounded time -- rebuild A and retry with GSLAM_HIP_CHOL_FLOW=0 / GSLAM_HIP_BWD_CHAIN=0
 * in the environment (gh_ba_solve does this by itself). */
/* With lda > n the right-hand side rides through the factorisation as row n of A (one launch less per solve, as in the
 * bundle adjustment): the padding rows n .. lda - 1 are scratch then. */
gh_status gh_potrf_solve_dev(gh_ctx* ctx, double* A_dev, int n, int lda, double* b_dev, int* info);

/* The same solve for a BAND matrix by block cyclic reduction (gslam_amd/csrc/chol_cr.hip): what gh_ba_solve uses for the
 * reduced camera system when the cameras only share points with their neighbours along the trajectory (the SPARSE_SCHUR
 * case of the Ceres solve behind GSLAM/core/Optimizer.h:229).  A_dev n x n column-major, lower triangle read and
 * overwritten, A[r][c] = 0 (stored zeros) for r - c > half_bandwidth; lda > n (row n is scratch for the right-hand side);
 * b_dev in, x out.  half_bandwidth <= 192 and n >= 4 superblocks of 64 * ceil(half_bandwidth / 64) columns, else
 * GH_ERR_ARG.  The result equals gh_potrf_solve_dev's up to rounding (a Cholesky factorisation of the odd-even permuted
 * matrix).  The reduction stops when two superblocks survive (four with a border: gh_arrow_solve_dev) and hands the system they
 * form to the dense path (GSLAM_HIP_CR_TOP = 1 .. 16 survivors; 1 = reduce down to the first superblock).
 * *info: 0 = ok; 1 .. n = first column + 1 of a diagonal tile that is not positive definite; > n = a bounded wait inside the dense
 * path's single-launch kernels expired (as gh_potrf_solve_dev reports it; A is overwritten either way). */
gh_status gh_band_solve_dev(gh_ctx* ctx, double* A_dev, int n, int lda, int half_bandwidth, double* b_dev, int* info);

/* ... and for an ARROWHEAD matrix, a band with a dense border: the reduced camera system of a trajectory with a few loop
 * closures once the cameras that long-range points tie to far-away ones are numbered last (what gh_ba_solve does by itself:
 * gh_ba_options.solver / the arrow ordering of gslam_amd/csrc/ba.hip; the graphs global BA exists for,
 * GSLAM/core/Optimizer.h:127-148,229).  The first n_band unknowns form the band (A[r][c] = 0 for r - c > half_bandwidth,
 * r < n_band), rows n_band .. n - 1 are dense.  Block cyclic reduction on the band with the border rows riding along as extra
 * rows of every eliminated superblock, then the dense system of the surviving superblocks + the border through the dense
 * factorisation.  Same limits on the band as gh_band_solve_dev; the result equals gh_potrf_solve_dev's up to rounding. */
gh_status gh_arrow_solve_dev(gh_ctx* ctx, double* A_dev, int n, int lda, int n_band, int half_bandwidth, double* b_dev,
                             int* info);
/* The COMPACT storage gh_ba_solve keeps its reduced camera system in when the band / arrowhead solver runs (round 6): a column
 * holds the rows block cyclic reduction ever touches -- 0.93 GB instead of the 28.8 GB of the dense lower triangle at C5.
 * gh_cr_compact_layout: for n_band band unknowns of `half_bandwidth` and nbr border unknowns, *lda = doubles per column, *m =
 * superblock columns, *brow = local row of the first border row.  Element (r, c), r >= c (plus the strictly upper part of a
 * column's own 64 x 64 diagonal tile, which is never read as data), at A[c * lda + local(r, c)]:
 *     r >= n_band                          brow + (r - n_band)        border rows; the right-hand side is local row brow + nbr
 *     I - J <= 1  (I = r / m, J = c / m)    r - J * m                  the band
 *     I - J = 2^k, k >= 1                   (1 + k) * m + (r - I * m)  the fill of the reduction: ZERO on entry
 * Returns 0 when the band does not fit the solver (more than 192 wide or fewer than four superblocks).
 * gh_arrow_solve_compact_dev: gh_arrow_solve_dev on such a matrix (n_band == n: a band without a border); A_dev is overwritten,
 * b_dev (n doubles) receives x.  Test / tool entries: gh_ba_solve builds the layout itself. */
int gh_cr_compact_layout(int n_band, int half_bandwidth, int nbr, int* lda, int* m, int* brow);
gh_status gh_arrow_solve_compact_dev(gh_ctx* ctx, double* A_dev, int n, int lda, int n_band, int half_bandwidth, double* b_dev,
                                     int* info);

/* Structure of the border through the reduction (host only, no GPU; what gh_ba_solve computes once per topology when the
 * border block has 4 M entries or more, so that the border kernels of the arrowhead solver skip what stays zero).
 * tiles = 64-column tiles per superblock (1 .. 3, as the solver picks them for the half-bandwidth); with m = 64 tiles,
 * N = ceil(n_band / m), nbs = ceil(nbr / 16), ntr = ceil((nbr + 1) / 64):
 *   init[i * nbs + t] != 0  iff  the 16-row strip t of the border rows has a structural non-zero among the band columns of
 *                                superblock i (a border camera and a band camera that see a common point);
 *   out: N * nbs bytes -- the strips that can be non-zero in superblock i at the moment the reduction eliminates it (all set
 *        for the superblocks that survive into the dense system) -- then N * ntr bytes, the same per 64-row tile of the
 *        corner (the tile that holds the right-hand-side row always set).
 * Returns the bytes `out` needs (0 = bad arguments); fills `out` when it is non-NULL and out_bytes suffices. */
size_t gh_cr_border_structure(int n_band, int tiles, int nbr, const uint8_t* init, uint8_t* out, size_t out_bytes);

/* Block-sparse Cholesky with a dense root: the linear solver gh_pg_solve uses for LARGE pose graphs
 * (GSLAM/core/Optimizer.h:127-148,162-167 -- se3Graph / sim3Graph / gpsGraph over thousands of keyframes; the system has
 * one 7 x 7 block per keyframe and one per edge).  Exposed for tests and tools:
 * gh_bs_symbolic (host only, no GPU): the elimination order of the keyframe graph given its distinct frame pairs --
 *   rounds of independent low-degree keyframes (cyclic reduction on an odometry chain), the rest is the dense root.
 *   counts_out[5] = {sparse columns, root keyframes, rounds, below-diagonal blocks incl. fill, 7 x 7 block products};
 *   pos_out[n_frames] frame -> position; round_ptr_out[rounds + 1]; colptr_out[sparse columns + 1]; rows_out = row
 *   positions of the blocks of each sparse column, ascending (any of the arrays may be NULL).
 * gh_bs_solve_host: (H + clamp(diag H, 1e-6, 1e32) / radius) x = -g for a symmetric positive definite H given by its
 *   diagonal blocks diag[n_frames][49] and off-diagonal blocks off[n_pairs][49] (block (row frame prow[k], column frame
 *   pcol[k]), both column-major); g, x_out[7 n_frames].  *info: 0 ok, > 0 a non-positive pivot.
 * root_min: keyframes kept for the dense root at least; max_rounds: bound on the elimination rounds. */
gh_status gh_bs_symbolic(int n_frames, int n_pairs, const int32_t* prow, const int32_t* pcol, int root_min, int max_rounds,
                         int32_t* pos_out, int64_t* counts_out, int32_t* round_ptr_out, int round_cap, int32_t* colptr_out,
                         int32_t* rows_out, int rows_cap);
gh_status gh_bs_solve_host(gh_ctx* ctx, int n_frames, int n_pairs, const int32_t* prow, const int32_t* pcol, const double* diag,
                           const double* off, const double* g, double radius, int root_min, int max_rounds, double* x_out,
                           int* info);

#ifdef __cplusplus
}
#endif
#endif /* GSLAM_HIP_H_ */
