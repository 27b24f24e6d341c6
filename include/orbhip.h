/* orbhip — C ABI of the MI355X (gfx950) ORB front end.
 *
 * Drop-in boundary for the two classes of the reference that form its data-parallel hot path:
 *   ORB_SLAM::ORBextractor  (include/ORBextractor.h:32-77, src/ORBextractor.cc)
 *   ORB_SLAM::ORBmatcher::DescriptorDistance + the best/second-best scan every Search* method
 *   runs (include/ORBmatcher.h:46, src/ORBmatcher.cc:461-487,1805-1821)
 *
 * Conventions: plain pointers and sizes, int status codes (0 = ok, < 0 = error), nothing throws
 * across the boundary.  One call in flight per handle (the reference's extractor is not
 * re-entrant either: include/ORBextractor.h:74-75 holds mutable pyramid scratch); stateless
 * functions (orbhip_hamming_distance) are re-entrant.  All work of a handle is issued on the
 * handle's own HIP stream; the *_device entry points are asynchronous on it.
 */
#ifndef ORBHIP_H
#define ORBHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBHIP_OK 0
#define ORBHIP_EINVAL (-1)   /* bad argument */
#define ORBHIP_ENOMEM (-2)   /* device or host allocation failed */
#define ORBHIP_EHIP (-3)     /* a HIP runtime call or kernel launch failed */
#define ORBHIP_ECAP (-4)     /* caller's output capacity / handle's max size too small */
#define ORBHIP_EGEOM (-5)    /* image too small for the cell grid the parameters imply */
#define ORBHIP_ENODEV (-6)   /* no usable GPU */
#define ORBHIP_EUNSUPPORTED (-7)

#define ORBHIP_MAX_LEVELS 16
#define ORBHIP_HARRIS_SCORE 0 /* include/ORBextractor.h:36 */
#define ORBHIP_FAST_SCORE 1

/* Field-for-field cv::KeyPoint of OpenCV 2.4 (28 bytes): what operator() fills
 * (src/ORBextractor.cc:769-777). */
typedef struct OrbKeypoint {
    float x, y;      /* pt, level coordinates times mvScaleFactor[octave] */
    float size;      /* (int)(31 * mvScaleFactor[octave]) */
    float angle;     /* degrees in [0, 360) */
    float response;  /* FAST score */
    int32_t octave;
    int32_t class_id; /* always -1 */
} OrbKeypoint;

typedef struct orbhip_extractor orbhip_extractor;

/* Replaces ORBextractor::ORBextractor (include/ORBextractor.h:38, src/ORBextractor.cc:457-511).
 * device: HIP device ordinal.  max_width/max_height/max_batch size the resident device arenas
 * (pyramids of max_batch frames stay in HBM); larger batches are processed in chunks. */
int orbhip_extractor_create(orbhip_extractor** out, int nfeatures, float scale_factor, int nlevels,
                            int score_type, int fast_th, int device, int max_width, int max_height,
                            int max_batch);
void orbhip_extractor_destroy(orbhip_extractor* e);

/* Replace ORBextractor::GetLevels / GetScaleFactor (include/ORBextractor.h:47-51). */
int orbhip_levels(const orbhip_extractor* e);
float orbhip_scale_factor(const orbhip_extractor* e);
/* Constructor tables, for callers that mirror them (Frame recomputes scale factors from
 * GetScaleFactor, src/Frame.cc:93-104).  which: 0 = features per level (int32[nlevels]),
 * 1 = mvScaleFactor (float[nlevels]), 2 = mvInvScaleFactor (float[nlevels]).  Returns count. */
int orbhip_table(const orbhip_extractor* e, int which, void* out);

/* Replaces ORBextractor::operator() (include/ORBextractor.h:43-45, src/ORBextractor.cc:718-779)
 * for one 8-bit single-channel image in host memory.  kp_out[cap], desc_out[cap*32] are host
 * buffers, cap >= nfeatures.  The mask argument of the reference has no effect on its results
 * (src/ORBextractor.cc:601-603 builds cellMask and never uses it) and is therefore not taken.
 * rows == 0 || cols == 0 returns ORBHIP_OK without touching the outputs (:721-722).  Blocking. */
int orbhip_extract(orbhip_extractor* e, const uint8_t* img, int rows, int cols, ptrdiff_t row_stride,
                   OrbKeypoint* kp_out, uint8_t* desc_out, int cap, int* n_out);

/* The same for a batch of equally sized frames in host memory (frame f starts at
 * imgs + f * frame_stride).  Outputs are per frame at stride cap: kp_out[batch*cap],
 * desc_out[batch*cap*32], n_out[batch].  Blocking.  Packed frames (row_stride == cols, frame_stride ==
 * rows * cols) that lie in page-locked host memory are copied to the device from where they lie; any other
 * input is first packed into the handle's own page-locked staging (about twice the time per call).  Chunks of
 * 512 frames and more run in pieces whose copies overlap the kernels of the piece before. */
int orbhip_extract_batch(orbhip_extractor* e, const uint8_t* imgs, int batch, int rows, int cols,
                         ptrdiff_t row_stride, ptrdiff_t frame_stride, OrbKeypoint* kp_out,
                         uint8_t* desc_out, int cap, int* n_out);

/* The same with every buffer resident in device memory.  Asynchronous on the handle's stream:
 * call orbhip_extractor_synchronize (or synchronise the device) before reading the outputs. */
int orbhip_extract_batch_device(orbhip_extractor* e, const uint8_t* d_imgs, int batch, int rows,
                                int cols, ptrdiff_t row_stride, ptrdiff_t frame_stride,
                                OrbKeypoint* d_kp_out, uint8_t* d_desc_out, int cap, int* d_n_out);
int orbhip_extractor_synchronize(orbhip_extractor* e);
/* The handle's hipStream_t, for callers that order their own work after the extraction. */
void* orbhip_extractor_stream(orbhip_extractor* e);

/* Stage taps of the last chunk processed (test instrumentation; blocking copies to host).
 * stage: 0 = unblurred padded plane of (frame, level)  -> u8 (h+32) x (w+32), tightly packed
 *        1 = blurred padded plane
 *        2 = FAST corner measure plane (same shape; only the interior is defined)
 *        3 = per-cell candidate counts of the level -> int32[nCells]
 *        4 = per-cell nToRetain of the level -> int32[nCells]
 *        5 = level keypoints after selection -> int32 triples x, y, score; returns count
 *        6 = per-cell candidates, cell-major raster order -> int32 triples x, y (cell coords), score
 *        (score: the FAST score; with HARRIS_SCORE the IEEE bits of the float response)
 * Returns the number of bytes (stages 0-2) or elements written, or < 0. */
int orbhip_debug_get(orbhip_extractor* e, int stage, int frame, int level, void* buf, size_t buf_bytes);
/* Level geometry of the current image size: out[0]=w, [1]=h, [2]=levelCols, [3]=levelRows,
 * [4]=cellW, [5]=cellH, [6]=nfeaturesCell, [7]=quota. */
int orbhip_level_geometry(const orbhip_extractor* e, int level, int32_t* out8);
/* Kernel time of the last extract call on this handle, measured with HIP events on the handle's
 * stream (milliseconds, whole call). */
float orbhip_last_elapsed_ms(orbhip_extractor* e);

/* Which build of OpenCV 2.4 the 7x7 Gaussian of src/ORBextractor.cc:760 is reproduced for.  The row pass and
 * the integer taps are the same in both; the column pass differs only where its sum is an exact tie
 * (sum mod 2^16 == 2^15): ORBHIP_GAUSS_SCALAR (default, the profile the oracle defines) rounds it up,
 * (sum + 2^15) >> 16, as the plain C++ column filter does; ORBHIP_GAUSS_SSE2 rounds it to the even neighbour
 * in the first w & ~3 columns of a level, as the fp32 / cvtps2dq column filter of an SSE2 build does
 * (about 6 blurred pixels per million and 1 descriptor bit per 2 million differ between the two). */
#define ORBHIP_GAUSS_SCALAR 0
#define ORBHIP_GAUSS_SSE2 1
int orbhip_extractor_set_gaussian_profile(orbhip_extractor* e, int profile);
/* Per-stage kernel timing (bench instrumentation).  With profiling enabled every extract call
 * brackets its kernel groups with HIP events on the handle's stream; after
 * orbhip_extractor_synchronize, stage_times returns for the last call the summed duration (ms) and
 * the launch count of each group: 0 level-0 border copy, 1 pyramid resize, 2 FAST corner measure,
 * 3 per-cell NMS + compaction, 4 quota + retainBest selection, 5 orientation, 6 Gaussian blur,
 * 7 descriptors + keypoint records.  Returns the number of groups (8). */
int orbhip_extractor_set_profiling(orbhip_extractor* e, int enable);
int orbhip_extractor_stage_times(orbhip_extractor* e, float* ms_out8, int* launches_out8);

/* Replaces ORBmatcher::DescriptorDistance (include/ORBmatcher.h:46, src/ORBmatcher.cc:1805-1821):
 * Hamming distance of two 32-byte descriptors in host memory.  A single pair is never worth a
 * kernel launch, so this is host popcount; re-entrant and lock-free. */
int orbhip_hamming_distance(const void* a, const void* b);

typedef struct orbhip_matcher orbhip_matcher;
int orbhip_matcher_create(orbhip_matcher** out, int device);
void orbhip_matcher_destroy(orbhip_matcher* m);
int orbhip_matcher_synchronize(orbhip_matcher* m);
void* orbhip_matcher_stream(orbhip_matcher* m);

/* Replaces the best / second-best scan of src/ORBmatcher.cc:461-485 run over the full train set
 * (identity candidate list), for npairs independent (query set, train set) pairs:
 * pair p has queries q + p*q_pair_stride (nq rows of 32 B) and train rows t + p*t_pair_stride.
 * Outputs per pair and query: idx (first train index attaining the minimum, + idx_base),
 * d1 (minimum distance), d2 (second order statistic with multiplicity).  nt == 0 gives
 * idx = -1, d1 = d2 = INT_MAX; nt == 1 gives d2 = INT_MAX, as the reference's INT_MAX
 * initialisation does.  Device pointers; asynchronous on the matcher's stream. */
int orbhip_knn2_device(orbhip_matcher* m, const uint8_t* d_q, int nq, ptrdiff_t q_pair_stride,
                       const uint8_t* d_t, int nt, ptrdiff_t t_pair_stride, int npairs, int idx_base,
                       int32_t* d_idx, int32_t* d_d1, int32_t* d_d2);
/* No counterpart in the reference.  The same scan as orbhip_knn2_device, same arguments, results and
 * asynchrony, for callers that cannot allocate device memory while they call: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  The partial results that the train
 * chunks of one call exchange live in a fixed array in device memory (orbhip_knn2_scratch_bytes()
 * = 16 MiB per GPU, held until the process ends) instead of the handle's growable workspace; this
 * call makes no hipMalloc / hipFree.  A call whose partial results (npairs * chunks * nq * 8 bytes) exceed the
 * array returns ORBHIP_ECAP and launches nothing.  The chunk count is the launch plan's: with few (query block,
 * pair) units the train set is cut so that the grid fills the chip (4 workgroups per CU for the matrix-core scan, 16
 * for the popcount scan, at least 2048 rows per chunk), and a chunk never exceeds the 4,194,272 rows the packed
 * keys can index, so chunks = max(chip-filling count, ceil(nt / 4194272)); orbhip_knn2_partial_bytes returns the
 * bytes a shape needs (0 for a single chunk) without launching anything, so a caller can test a shape beforehand.
 * All matchers of a GPU share that GPU's array.  Calls are ordered against each other on the device, whichever
 * handle and stream they come from (each waits for the previous user's kernels through a per-device event; neither
 * step allocates), so matchers of different host threads may call concurrently; the exception is a call made
 * under stream capture, which takes no part in that ordering.  Nothing in the array needs clearing between
 * calls: every entry a call reads it has written first. */
int orbhip_knn2_device_noalloc(orbhip_matcher* m, const uint8_t* d_q, int nq, ptrdiff_t q_pair_stride,
                               const uint8_t* d_t, int nt, ptrdiff_t t_pair_stride, int npairs, int idx_base,
                               int32_t* d_idx, int32_t* d_d1, int32_t* d_d2);
int orbhip_knn2_scratch_bytes(void);
/* Bytes of partial results orbhip_knn2_device / _noalloc would use for this shape on this handle (0: one chunk, no
 * scratch needed; -1: invalid arguments).  Host arithmetic; launches nothing. */
long long orbhip_knn2_partial_bytes(orbhip_matcher* m, int nq, int nt, int npairs);
/* Host-pointer convenience form of orbhip_knn2_device (copies in and out; blocking). */
int orbhip_knn2(orbhip_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt, int npairs,
                int32_t* idx, int32_t* d1, int32_t* d2);
/* Merge of per-shard results of index-contiguous train shards (shard g owns a contiguous index
 * range that precedes shard g+1's): parts[g*3*nq ...] holds idx, d1, d2 arrays of shard g.
 * Reproduces the serial ascending scan exactly.  Device pointers; asynchronous. */
int orbhip_knn2_merge_device(orbhip_matcher* m, const int32_t* d_parts, int nshards, int nq,
                             int32_t* d_idx, int32_t* d_d1, int32_t* d_d2);
/* Host twin of orbhip_knn2_merge_device, same function on records in host memory (no GPU involved): for a host that
 * moves the per-shard records through a transport of its own (MPI, gloo, a socket) instead of RCCL. */
int orbhip_knn2_merge_host(const int32_t* parts, int nshards, int nq, int32_t* idx, int32_t* d1, int32_t* d2);
/* Layout of the all-gather of orbhip_knn2_sharded: every rank contributes one record of *record_ints int32 (idx[nq],
 * d1[nq], d2[nq]); rank r's record lies at int32 offset *slot_offset_ints of the receive buffer of *total_bytes
 * bytes — the layout orbhip_knn2_merge_device / _host take as `parts`.  Pure host arithmetic. */
int orbhip_knn2_gather_layout(int nq, int world, int rank, size_t* record_ints, size_t* slot_offset_ints, size_t* total_bytes);
/* ---- Frame-versus-map matching over a map sharded across GPUs --------------------------------------
 * One process per GPU.  The map descriptors are split into index-contiguous shards (rank g owns the
 * global rows [idx_base_g, idx_base_g + nt_shard_g), rank order = index order); the queries are
 * replicated.  A communicator wraps one RCCL communicator (ncclComm_t); RCCL is bound at run time
 * (the copy the process has loaded, else ORBHIP_RCCL_LIB, else librccl.so.1). */
typedef struct orbhip_comm orbhip_comm;
#define ORBHIP_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */
/* ncclGetUniqueId: one rank calls it and hands the bytes to every rank by its own means (a file, a
 * socket, MPI, torch.distributed ...). */
int orbhip_comm_unique_id(void* id_out, size_t id_bytes);
/* ncclCommInitRank on `device`: collective over all `world` ranks of the id. */
int orbhip_comm_create(orbhip_comm** out, const void* id, size_t id_bytes, int rank, int world, int device);
/* Wraps a communicator the host program already owns (ncclComm_t as void*); it is not destroyed by
 * orbhip_comm_destroy. */
int orbhip_comm_adopt(orbhip_comm** out, void* nccl_comm, int rank, int world, int device);
void orbhip_comm_destroy(orbhip_comm* c);
int orbhip_comm_rank(const orbhip_comm* c);
int orbhip_comm_world(const orbhip_comm* c);
/* Replaces the best / second-best scan of src/ORBmatcher.cc:461-485 over the WHOLE map for nq queries
 * (the scans Tracking::SearchReferencePointsInFrustum, src/Tracking.cc:880-885, and
 * LoopClosing::ComputeSim3, src/LoopClosing.cc:263-347, run over their candidate map points, here with
 * the identity candidate list).  Every rank calls it with the same queries and its own shard; on return
 * (asynchronously, on the matcher's stream) every rank holds the same d_idx (global row of the first
 * minimum, or -1 for an empty map), d_d1, d_d2 — bit for bit what one serial ascending scan of the
 * unsharded map gives.  Steps, all on the matcher's stream: scan of the shard, ONE ncclAllGather of
 * 3 * nq int32 per rank, ordered merge.  comm == NULL stands for a single shard (no collective).
 * Collective: every rank of the communicator must call it, in the same order relative to other
 * collectives on that communicator. */
int orbhip_knn2_sharded(orbhip_matcher* m, orbhip_comm* comm, const uint8_t* d_q, int nq, const uint8_t* d_t_shard,
                        int nt_shard, int idx_base, int32_t* d_idx, int32_t* d_d1, int32_t* d_d2);

/* Acceptance predicates of src/ORBmatcher.cc evaluated per query on the device:
 * form 0: (float)d1 < ratio*(float)d2 (:237,:665,:804), 1: d1 <= d2*ratio (:487,:596);
 * accepted iff the ratio form holds and d1 <= th (TH_LOW 20 / TH_HIGH 100, :42-45).
 * accept[i] = 1/0.  d2 == INT_MAX passes every ratio form. */
int orbhip_ratio_test_device(orbhip_matcher* m, const int32_t* d_d1, const int32_t* d_d2, int n,
                             float ratio, int form, int th, uint8_t* d_accept);
float orbhip_matcher_last_elapsed_ms(orbhip_matcher* m);
/* Sustained integer-VALU rate of the device in lane-operations per second, measured live with HIP
 * events over a saturating launch (bench instrumentation; the denominator of the matcher's
 * roofline): mode 0 = v_bcnt_u32_b32, 1 = the matcher's v_xor_b32 + v_bcnt_u32_b32 mix,
 * 2 = v_add_u32.  Returns 0 on failure. */
double orbhip_valu_rate(orbhip_matcher* m, int mode);

/* ---- Feature grid of a frame and the windowed matchers that take their candidates from it -------
 * Batched over independent frames / frame pairs; every pointer is a device pointer and the calls
 * are asynchronous on the matcher's stream unless stated otherwise.  Frame f has d_n[f] keypoints
 * at d_kp + f*cap (records as written by orbhip_extract_batch_device) and descriptors at
 * d_desc + f*cap*32.  cap <= ORBHIP_FRAME_MAX_KEYPOINTS. */
#define ORBHIP_FRAME_GRID_COLS 64 /* include/Frame.h:38 */
#define ORBHIP_FRAME_GRID_ROWS 48 /* include/Frame.h:37 */
#define ORBHIP_FRAME_GRID_CELLS (ORBHIP_FRAME_GRID_COLS * ORBHIP_FRAME_GRID_ROWS)
#define ORBHIP_FRAME_MAX_KEYPOINTS 8192
/* Frame::mnMinX, mnMinY, mnMaxX, mnMaxY (src/Frame.cc:322-350): {0, 0, cols, rows} for an
 * undistorted camera. */
typedef struct OrbFrameBounds {
    int32_t min_x, min_y, max_x, max_y;
} OrbFrameBounds;

/* Replaces the grid assignment of the Frame constructor (src/Frame.cc:78-79,110-124) and
 * Frame::PosInGrid (:268-278).  mGrid[ix][iy] of frame f is
 * d_cell_items[f*cap + d_cell_start[f*(CELLS+1) + c] ... d_cell_start[f*(CELLS+1) + c + 1]) with
 * c = ix*ORBHIP_FRAME_GRID_ROWS + iy, in push_back (ascending keypoint index) order; keypoints that
 * PosInGrid rejects are in no cell. */
int orbhip_frame_grid_device(orbhip_matcher* m, const OrbKeypoint* d_kp, const int32_t* d_n, int batch, int cap,
                             OrbFrameBounds bounds, int32_t* d_cell_start, int32_t* d_cell_items);

/* One call of Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel); minLevel = maxLevel = -1
 * switches the level check off. */
typedef struct OrbAreaQuery {
    float x, y, r;
    int32_t min_level, max_level;
} OrbAreaQuery;
/* Replaces Frame::GetFeaturesInArea (src/Frame.cc:201-266), nq queries per frame
 * (d_queries[f*nq + q]).  The indices query q returns, in the reference's order (grid column
 * outer, grid row inner, push order inside a cell), are
 * d_out[f*out_cap + d_out_start[f*(nq+1) + q] ... d_out_start[f*(nq+1) + q + 1]).  A frame whose
 * queries return more than out_cap indices in total makes the next orbhip_matcher_synchronize
 * fail with ORBHIP_ECAP (the starts are still complete, so the caller can size a retry). */
int orbhip_features_in_area_device(orbhip_matcher* m, const OrbKeypoint* d_kp, const int32_t* d_n, int batch, int cap,
                                   OrbFrameBounds bounds, const int32_t* d_cell_start, const int32_t* d_cell_items,
                                   const OrbAreaQuery* d_queries, int nq, int32_t* d_out_start, int32_t* d_out,
                                   int out_cap);

/* Which kernel scans the train rows of orbhip_knn2_device, orbhip_knn2_device_noalloc, orbhip_knn2 and
 * orbhip_knn2_sharded.  ORBHIP_SCAN_MATRIX (the default; ORBHIP_KNN_KERNEL=popcount in the environment makes the
 * other one the default of new handles): descriptor bits as 4-bit matrix operands, the distances of 32 rows x 32
 * queries from four v_mfma_f32_32x32x64_f8f6f4.  ORBHIP_SCAN_POPCOUNT: xor + popcount per lane, rows through scalar
 * loads.  Both return the same integers (ties to the lower index); no counterpart in the reference, whose
 * DescriptorDistance (src/ORBmatcher.cc:1805-1821) is a scalar bit trick. */
#define ORBHIP_SCAN_MATRIX 0
#define ORBHIP_SCAN_POPCOUNT 1
int orbhip_matcher_set_scan_kernel(orbhip_matcher* m, int kind);

/* Asynchronous form of orbhip_search_for_initialization_device (orbhip_window_search_device and the projection
 * matchers keep no candidate lists and are always asynchronous).  Its
 * candidate lists (one 4-byte record per
 * feature that a GetFeaturesInArea call returns) live in a workspace of the handle.  Without a reservation
 * (entries = 0, the default) each call sizes that workspace from a count it reads back from the device, and
 * blocks.  With a reservation of `entries` records the calls read nothing back and return as soon as their
 * kernels are queued on the matcher's stream; a call whose lists need more than the reservation produces no
 * matches (all -1, counts 0) and makes the next orbhip_matcher_synchronize fail with ORBHIP_ECAP, the message
 * naming the count that was needed.  No counterpart in the reference (its matchers are host loops). */
int orbhip_matcher_reserve_candidates(orbhip_matcher* m, long long entries);

/* Replaces ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:609-724) for `batch` independent
 * frame pairs (F1 = *_1, F2 = *_2 with its grid).  d_prev_matched (batch*cap*2 floats, x then y) is
 * vbPrevMatched: read as the window centres and updated for the matched keypoints.  Outputs
 * d_matches12 (batch*cap, vnMatches12: index in F2 or -1) and d_nmatches (batch).  The sequential
 * dependence of the reference (a later keypoint of F1 can take a feature of F2 away from an
 * earlier one, rotation histogram over every accepted match) is reproduced exactly.
 * Blocking (sizes an internal workspace from a device count) unless a reservation is in place
 * (orbhip_matcher_reserve_candidates). */
int orbhip_search_for_initialization_device(orbhip_matcher* m, const OrbKeypoint* d_kp1, const uint8_t* d_desc1,
                                            const int32_t* d_n1, const OrbKeypoint* d_kp2, const uint8_t* d_desc2,
                                            const int32_t* d_n2, const int32_t* d_cell_start2,
                                            const int32_t* d_cell_items2, int batch, int cap, OrbFrameBounds bounds,
                                            float* d_prev_matched, int window_size, float nnratio,
                                            int check_orientation, int32_t* d_matches12, int32_t* d_nmatches);

/* Replaces ORBmatcher::WindowSearch (src/ORBmatcher.cc:420-527).  d_valid1[f*cap + i] != 0 stands
 * for "F1.mvpMapPoints[i] exists and is not bad"; min_level <= 0 and max_level == INT_MAX switch
 * the level filters off as in the reference.  Output d_matches21 (batch*cap): the F1 index whose
 * map point is assigned to each F2 feature (vpMapPointMatches2[i2] = F1.mvpMapPoints[that]) or -1;
 * d_nmatches (batch).  Asynchronous on the matcher's stream; no candidate lists (the search has the shape of the
 * projection matchers and runs through the same two kernels, F1's keypoints as the probes). */
int orbhip_window_search_device(orbhip_matcher* m, const OrbKeypoint* d_kp1, const uint8_t* d_desc1,
                                const int32_t* d_n1, const uint8_t* d_valid1, const OrbKeypoint* d_kp2,
                                const uint8_t* d_desc2, const int32_t* d_n2, const int32_t* d_cell_start2,
                                const int32_t* d_cell_items2, int batch, int cap, OrbFrameBounds bounds,
                                int window_size, int min_level, int max_level, float nnratio, int check_orientation,
                                int32_t* d_matches21, int32_t* d_nmatches);

/* One projected map point of the projection matchers: where it falls in the target frame, the
 * radius and level arguments of the GetFeaturesInArea call the reference makes for it, and the angle
 * of the keypoint it was observed at (rotation histogram).  valid = 0 stands for every `continue` the
 * reference takes before that call (no map point, bad, already found, outside the image, ...). */
typedef struct OrbProbe {
    float u, v, r;
    int32_t min_level, max_level;
    float angle;
    int32_t valid;
    int32_t reserved;
} OrbProbe;
#define ORBHIP_MAX_PROBES 32768
/* Acceptance rules of the five ORBmatcher::SearchByProjection overloads */
#define ORBHIP_PROJ_LOCAL_MAP 0 /* (Frame&, vector<MapPoint*>&, th)  src/ORBmatcher.cc:52-128 */
#define ORBHIP_PROJ_RATIO 1     /* (Frame&, Frame&, windowSize, ...)  :530-605 */
#define ORBHIP_PROJ_BEST 2      /* (Frame&, const Frame&, th) :1518-1631; (Frame&, KeyFrame*, ...) :1633-1757;
                                   (KeyFrame*, Scw, ...) :297-418 — best distance <= th_dist */
/* Search and assignment loops of ORBmatcher::SearchByProjection for `batch` independent target
 * frames.  Frame f receives d_nprobes[f] probes d_probes[f*probe_cap ...] with descriptors
 * d_probe_desc[(f*probe_cap + i)*32 ...], processed in index order as the reference walks its map
 * points: candidates from GetFeaturesInArea(u, v, r, min_level, max_level) on the frame's grid,
 * features that already hold a map point skipped — d_taken[f*cap + i2] != 0 on entry (may be NULL)
 * or assigned by an earlier probe — Hamming distances, acceptance by `rule`, and for
 * check_orientation != 0 the rotation histogram / three-maxima filter over the new matches.
 * Output d_match_of_feature (batch*cap): the probe assigned to each feature by this call or -1;
 * d_nmatches (batch).  probe_cap <= ORBHIP_MAX_PROBES.  Asynchronous on the matcher's stream: no candidate
 * lists are stored (per probe its four best free candidates; a probe that loses them to earlier probes has its
 * area enumerated again), nothing is read back. */
int orbhip_search_by_projection_device(orbhip_matcher* m, const OrbProbe* d_probes, const uint8_t* d_probe_desc,
                                       const int32_t* d_nprobes, int probe_cap, const OrbKeypoint* d_kp2,
                                       const uint8_t* d_desc2, const int32_t* d_n2, const int32_t* d_cell_start2,
                                       const int32_t* d_cell_items2, int batch, int cap, OrbFrameBounds bounds,
                                       const uint8_t* d_taken, int rule, float nnratio, int th_dist, int check_orientation,
                                       int32_t* d_match_of_feature, int32_t* d_nmatches);

/* Host-pointer form for ONE target frame (copy in, build the frame's grid, match, copy out; blocking):
 * what a forwarding ORBmatcher::SearchByProjection calls after it has projected its map points with
 * the reference's own pose arithmetic. */
int orbhip_search_by_projection(orbhip_matcher* m, const OrbProbe* probes, const uint8_t* probe_desc, int nprobes,
                                const OrbKeypoint* kp2, const uint8_t* desc2, int n2, OrbFrameBounds bounds,
                                const uint8_t* taken, int rule, float nnratio, int th_dist, int check_orientation,
                                int32_t* match_of_feature, int32_t* nmatches);

/* ---- Probes from poses and map points -------------------------------------------------------------
 * A camera as the projection matchers use it: rotation, translation and centre of the pose (CV_32F) and the
 * intrinsics.  orbhip_projection_from_pose fills it from a 4x4 row-major Tcw as Frame::UpdatePoseMatrices
 * (src/Frame.cc:131-136) and KeyFrame::SetPose (src/KeyFrame.cc:102-109) do (Ow = -Rcw.t()*tcw);
 * orbhip_projection_from_sim3 decomposes a Sim3 Scw first (src/ORBmatcher.cc:305-310, :1156-1160).  Host
 * computations. */
typedef struct OrbProjection {
    float Rcw[9], tcw[3], Ow[3];
    float fx, fy, cx, cy;
} OrbProjection;
int orbhip_projection_from_pose(const float* Tcw, float fx, float fy, float cx, float cy, OrbProjection* out);
int orbhip_projection_from_sim3(const float* Scw, float fx, float fy, float cx, float cy, OrbProjection* out);
#define ORBHIP_PROBES_LAST_FRAME 0 /* SearchByProjection(CurrentFrame, LastFrame, th)  src/ORBmatcher.cc:1531-1556 */
#define ORBHIP_PROBES_KEYFRAME 1   /* SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist)  :1651-1690 */
#define ORBHIP_PROBES_FRUSTUM 2    /* Frame::isInFrustum (src/Frame.cc:138-199) + SearchByProjection(F, vpMapPoints, th) :63-81 */
#define ORBHIP_PROBES_FUSE 3       /* Fuse, both forms (:1054-1094, :1180-1223); SearchByProjection(pKF, Scw, ...) (:326-375) */
#define ORBHIP_PROBES_WINDOW 4     /* SearchByProjection(F1, F2, windowSize, vpMapPointMatches2) (:545-566): radius = th =
                                      windowSize, the source keypoint's octave the only level, no image test */
#define ORBHIP_PROBES_SIM3 5       /* SearchBySim3, either direction (:1330-1375, :1415-1460); two cameras:
                                      orbhip_build_probes_sim3_device */
/* Replaces, for `batch` target frames, the lines of those methods between their pointer tests and their
 * GetFeaturesInArea call: frame f projects its d_npoints[f] map points (world positions d_pos, 3 floats each;
 * normals d_normal; GetMinDistanceInvariance / GetMaxDistanceInvariance in d_min_dist / d_max_dist; all indexed
 * f*probe_cap + i) with camera d_cams[f] and writes one OrbProbe per point — valid = 0 where the reference takes a
 * `continue` (d_skip[f*probe_cap + i] != 0 stands for its pointer tests: NULL, isBad(), outlier, already found).
 * d_src_kp (LAST_FRAME: LastFrame.mvKeys[i] / mvKeysUn[i]; KEYFRAME: pKF->GetKeyPointUn(i)) supplies the octave and
 * the angle; th is the method's th (FRUSTUM: the th of SearchByProjection, the radius being
 * RadiusByViewingCos(viewCos) [* th if th != 1] * scale); viewing_cos_limit is isInFrustum's (0.5 in
 * src/Tracking.cc).  Arrays a kind does not read may be NULL (LAST_FRAME: normals and distances; KEYFRAME: normals
 * and d_max_dist).  The probes feed orbhip_search_by_projection_device (LAST_FRAME and KEYFRAME with
 * ORBHIP_PROJ_BEST, FRUSTUM with ORBHIP_PROJ_LOCAL_MAP, FUSE with ORBHIP_PROJ_BEST for the Sim3 overload) or
 * orbhip_best_in_radius_device (FUSE).  Device pointers; asynchronous on the matcher's stream. */
int orbhip_build_probes_device(orbhip_matcher* m, int kind, const OrbProjection* d_cams, OrbFrameBounds bounds,
                               const float* d_scale_factors, int nlevels, const float* d_pos, const float* d_normal,
                               const float* d_min_dist, const float* d_max_dist, const OrbKeypoint* d_src_kp,
                               const uint8_t* d_skip, const int32_t* d_npoints, int batch, int probe_cap, float th,
                               float viewing_cos_limit, OrbProbe* d_probes);

/* ORBHIP_PROBES_SIM3: one direction of ORBmatcher::SearchBySim3.  A map point of the source keyframe goes into that
 * keyframe's camera (d_cams_from[f]: its pose) and from there through the similarity (d_cams_to[f]: Rcw = s*R,
 * tcw = t, and the intrinsics the projection uses — orbhip_projection_sim3_pair builds both directions); depth,
 * image and distance-range tests, predicted level and radius th * scale as in the reference; no viewing-angle test.
 * The probes feed orbhip_best_in_radius_device.  Device pointers; asynchronous. */
int orbhip_build_probes_sim3_device(orbhip_matcher* m, const OrbProjection* d_cams_from, const OrbProjection* d_cams_to,
                                    OrbFrameBounds bounds, const float* d_scale_factors, int nlevels, const float* d_pos,
                                    const float* d_min_dist, const float* d_max_dist, const uint8_t* d_skip,
                                    const int32_t* d_npoints, int batch, int probe_cap, float th, OrbProbe* d_probes);
/* The two similarities of ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1290-1292) from s12, R12 (9 floats, row-major) and
 * t12: *to2 carries sR21 = (1/s12) R12^T and t21 = -sR21 t12 (keyframe 1's points into camera 2), *to1 carries
 * sR12 = s12 R12 and t12; both get the intrinsics (the reference projects both directions with pKF1's).  Host
 * computation, rounded where cv::Mat arithmetic rounds. */
int orbhip_projection_sim3_pair(float s12, const float* R12, const float* t12, float fx, float fy, float cx, float cy,
                                OrbProjection* to2, OrbProjection* to1);
/* Probes of ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th) (src/ORBmatcher.cc:52-81)
 * from the fields Frame::isInFrustum left in each map point: proj_xy = (mTrackProjX, mTrackProjY), view_cos =
 * mTrackViewCos, level = mnTrackScaleLevel; skip[i] != 0 stands for !mbTrackInView || isBad().  Radius =
 * RadiusByViewingCos(view_cos) [* th if th != 1] * scale_factors[level], levels level-1 .. level.  Host computation
 * (one multiply per point); a level outside [0, nlevels) is ORBHIP_EINVAL. */
int orbhip_probes_from_tracked(int n, const float* proj_xy, const float* view_cos, const int32_t* level, const uint8_t* skip,
                               float th, const float* scale_factors, int nlevels, OrbProbe* probes);

/* The search the reference's Fuse and SearchBySim3 make per projected map point — "match to the most similar
 * keypoint in the radius" (ORBmatcher::Fuse src/ORBmatcher.cc:1096-1124 and :1225-1253, ORBmatcher::SearchBySim3
 * :1376-1404 and :1461-1489): KeyFrame::GetFeaturesInArea(u, v, r) (src/KeyFrame.cc:638-683), the level test
 * kpLevel in [min_level, max_level] of the probe, and the first candidate in list order with the least Hamming
 * distance.  Unlike the SearchByProjection overloads the probes are independent (a feature can be the best of
 * several probes; nothing is skipped), which is what these methods need: what they do with the result (Replace /
 * AddObservation, the <= TH_LOW or <= TH_HIGH gate, the mutual-agreement check) stays host code in the
 * reference's order.  Frame f: d_nprobes[f] probes at d_probes[f*probe_cap ...] with descriptors
 * d_probe_desc[(f*probe_cap + i)*32 ...]; outputs (batch*probe_cap each) d_best_idx = bestIdx or -1 when the
 * probe is not valid or no candidate passes the level test, d_best_dist = bestDist or INT_MAX.  Device pointers;
 * asynchronous on the matcher's stream; no workspace, nothing read back. */
int orbhip_best_in_radius_device(orbhip_matcher* m, const OrbProbe* d_probes, const uint8_t* d_probe_desc,
                                 const int32_t* d_nprobes, int probe_cap, const OrbKeypoint* d_kp2, const uint8_t* d_desc2,
                                 const int32_t* d_cell_start2, const int32_t* d_cell_items2, int batch, int cap,
                                 OrbFrameBounds bounds, int32_t* d_best_idx, int32_t* d_best_dist);
/* Host-pointer form for ONE keyframe (copy in, build its grid, search, copy out; blocking): what the forwarding
 * bodies of ORBmatcher::Fuse (src/ORBmatcher.cc:1027-1145, :1147-1276) and ORBmatcher::SearchBySim3 (:1278-1516)
 * call after they have projected their map points with the reference's own pose arithmetic. */
int orbhip_best_in_radius(orbhip_matcher* m, const OrbProbe* probes, const uint8_t* probe_desc, int nprobes,
                          const OrbKeypoint* kp2, const uint8_t* desc2, int n2, OrbFrameBounds bounds, int32_t* best_idx,
                          int32_t* best_dist);

/* ---- Map points in, matches out: what the forwarding ORBmatcher methods call ---------------------------------
 * The map points of one call as flat host arrays of n entries each: what the reference reads out of its MapPoint /
 * Frame / KeyFrame objects between its pointer tests and its GetFeaturesInArea call.  Arrays a probe kind does not
 * read (see orbhip_build_probes_device) may be NULL. */
typedef struct OrbPointSet {
    int32_t n;
    const float* pos;          /* 3 per point: MapPoint::GetWorldPos() */
    const float* normal;       /* 3 per point: MapPoint::GetNormal() */
    const float* min_dist;     /* MapPoint::GetMinDistanceInvariance() */
    const float* max_dist;     /* MapPoint::GetMaxDistanceInvariance() */
    const OrbKeypoint* src_kp; /* the keypoint the point was observed at: octave and angle */
    const uint8_t* skip;       /* != 0: the reference's pointer tests drop the point (NULL, isBad(), outlier, already found) */
    const uint8_t* desc;       /* 32 per point: MapPoint::GetDescriptor(), or the source frame's descriptor row */
} OrbPointSet;
/* One call of a projection matcher for ONE target frame, host pointers, blocking: the points are projected with `cam`
 * (orbhip_build_probes_device, kinds LAST_FRAME / KEYFRAME / FRUSTUM / FUSE / WINDOW), the probes stay on the device and
 * feed orbhip_search_by_projection_device over the frame's grid.  match_of_feature (n2): the point assigned to each
 * feature by this call or -1; *nmatches: the method's return value.  probes_out (may be NULL): the n probes, for
 * inspection.  Replaces src/ORBmatcher.cc:297-418 (FUSE + ORBHIP_PROJ_BEST, th_dist TH_LOW), :530-605 (WINDOW +
 * ORBHIP_PROJ_RATIO, TH_HIGH), :1518-1631 (LAST_FRAME + ORBHIP_PROJ_BEST, TH_HIGH), :1633-1757 (KEYFRAME +
 * ORBHIP_PROJ_BEST, ORBdist). */
int orbhip_project_and_search(orbhip_matcher* m, int kind, const OrbProjection* cam, const float* scale_factors, int nlevels,
                              const OrbPointSet* pts, float th, float viewing_cos_limit, const OrbKeypoint* kp2,
                              const uint8_t* desc2, int n2, OrbFrameBounds bounds, const uint8_t* taken, int rule, float nnratio,
                              int th_dist, int check_orientation, int32_t* match_of_feature, int32_t* nmatches,
                              OrbProbe* probes_out);
/* The same for the independent search of Fuse (kind FUSE, cam_to NULL; src/ORBmatcher.cc:1054-1124, :1180-1253) and of
 * one direction of SearchBySim3 (kind SIM3, cam = the source keyframe's pose, cam_to = the similarity; :1330-1404,
 * :1415-1489): best_idx / best_dist per point as orbhip_best_in_radius_device returns them. */
int orbhip_project_and_best_in_radius(orbhip_matcher* m, int kind, const OrbProjection* cam, const OrbProjection* cam_to,
                                      const float* scale_factors, int nlevels, const OrbPointSet* pts, float th,
                                      const OrbKeypoint* kp2, const uint8_t* desc2, int n2, OrbFrameBounds bounds,
                                      int32_t* best_idx, int32_t* best_dist, OrbProbe* probes_out);

/* Host-pointer forms of the two matchers for ONE frame pair (copy in, build F2's grid, match, copy
 * out; blocking) — what a forwarding ORBmatcher::SearchForInitialization / WindowSearch calls with
 * the contents of F1.mvKeysUn / mDescriptors and F2's. */
int orbhip_search_for_initialization(orbhip_matcher* m, const OrbKeypoint* kp1, const uint8_t* desc1, int n1,
                                     const OrbKeypoint* kp2, const uint8_t* desc2, int n2, OrbFrameBounds bounds,
                                     float* prev_matched, int window_size, float nnratio, int check_orientation,
                                     int32_t* matches12, int32_t* nmatches);
int orbhip_window_search(orbhip_matcher* m, const OrbKeypoint* kp1, const uint8_t* desc1, int n1, const uint8_t* valid1,
                         const OrbKeypoint* kp2, const uint8_t* desc2, int n2, OrbFrameBounds bounds, int window_size,
                         int min_level, int max_level, float nnratio, int check_orientation, int32_t* matches21,
                         int32_t* nmatches);

/* Replaces the descriptor choice of MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:221-250)
 * for npoints map points at once: point p observes the descriptors (rows of 32 bytes)
 * d_desc[32*d_start[p] ... 32*d_start[p+1]); d_best_idx[p] = the row (relative to d_start[p]) with
 * the least median Hamming distance to all rows of the point, the first one on ties;
 * d_best_median[p] = that median (the element of rank (size_t)(0.5*(N-1)) of the sorted row).  A
 * point without observations gives -1 / INT_MAX.  Points with more than 128 observations make the
 * next orbhip_matcher_synchronize fail with ORBHIP_ECAP.  Device pointers; asynchronous. */
int orbhip_distinctive_descriptors_device(orbhip_matcher* m, const uint8_t* d_desc, const int32_t* d_start,
                                          int npoints, int32_t* d_best_idx, int32_t* d_best_median);

/* ---- Keypoint undistortion -----------------------------------------------------------------------
 * Camera of a frame as the reference holds it: mK (fx, fy, cx, cy; CV_32F) and mDistCoef (k1, k2, p1,
 * p2; CV_32F), src/Tracking.cc (settings file) and include/Frame.h. */
typedef struct OrbCamera {
    float fx, fy, cx, cy;
    float k1, k2, p1, p2;
} OrbCamera;
/* Replaces Frame::UndistortKeyPoints (src/Frame.cc:290-320) for `batch` frames: d_kp_un[f*cap + i] is
 * keypoint record i of frame f with pt replaced by cv::undistortPoints(pt, mK, mDistCoef, Mat(), mK) —
 * double arithmetic, five iterations, as OpenCV 2.4 evaluates it — or an unchanged copy when
 * cam.k1 == 0 (:292-296).  In-place (d_kp_un == d_kp) is allowed.  Device pointers; asynchronous on
 * the matcher's stream. */
int orbhip_undistort_keypoints_device(orbhip_matcher* m, const OrbKeypoint* d_kp, const int32_t* d_n, int batch, int cap,
                                      OrbCamera cam, OrbKeypoint* d_kp_un);
/* Replaces Frame::ComputeImageBounds (src/Frame.cc:322-350): the bounds of the undistorted image from
 * its four corners (floor / ceil of the float results), or {0, 0, cols, rows} when cam.k1 == 0.  Host
 * computation; no GPU involved. */
int orbhip_image_bounds(OrbCamera cam, int cols, int rows, OrbFrameBounds* out);

/* ---- Bag-of-words transform (vocabulary tree) ----------------------------------------------------
 * The tree of Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h (Node, :296-330) as arrays in node-id
 * order, node 0 = root: parent[i] < i, the node's 32-byte descriptor, its weight (read at leaves).
 * Children attach to their parent in id order and word ids number the leaves in id order, as
 * loadFromTextFile (:1337-1420) and the tree builder do.  L = depth of the tree (m_L).  The arrays
 * are copied to the device; host pointers. */
typedef struct orbhip_vocabulary orbhip_vocabulary;
int orbhip_vocabulary_create(orbhip_vocabulary** out, int device, int nnodes, const int32_t* parent, const uint8_t* desc,
                             const double* weight, int L);
void orbhip_vocabulary_destroy(orbhip_vocabulary* v);
int orbhip_vocabulary_words(const orbhip_vocabulary* v);
/* Replaces TemplatedVocabulary::transform(feature, word_id, weight, &nid, levelsup) (:1218-1259)
 * for n descriptors (rows of 32 bytes): per descriptor the word id, the id of its ancestor at level
 * L - levelsup (0 = root when that level is <= 0) and the word's weight.  The BowVector /
 * FeatureVector of transform(features, v, fv, levelsup) (:1125-1194) are these triples accumulated
 * in feature order (v.addWeight(word, weight) / fv.addFeature(node, i) for weight > 0).  Device
 * pointers; asynchronous on the matcher's stream. */
int orbhip_bow_transform_device(orbhip_matcher* m, const orbhip_vocabulary* v, const uint8_t* d_desc, int n, int levelsup,
                                int32_t* d_word, int32_t* d_node, double* d_weight);

/* FeatureVector of each of `batch` frames from the per-descriptor outputs of
 * orbhip_bow_transform_device (frame f: d_n[f] descriptors at d_node + f*cap, d_weight + f*cap), as
 * transform(features, v, fv, levelsup) fills it (TemplatedVocabulary.h:1150-1163, FeatureVector.cpp:
 * 31-45): descriptors with weight > 0 grouped by node, nodes ascending, feature indices ascending.
 * Frame f: d_fv_count[f] nodes d_fv_nodes[f*node_cap ...], features of node j at
 * d_fv_feat[f*cap + d_fv_start[f*(node_cap+1) + j] ... d_fv_start[f*(node_cap+1) + j + 1]).  More than
 * node_cap distinct nodes make the next orbhip_matcher_synchronize fail with ORBHIP_ECAP.
 * Device pointers; asynchronous. */
int orbhip_feature_vector_device(orbhip_matcher* m, const int32_t* d_node, const double* d_weight, const int32_t* d_n,
                                 int batch, int cap, int node_cap, int32_t* d_fv_nodes, int32_t* d_fv_start,
                                 int32_t* d_fv_feat, int32_t* d_fv_count);
/* Replaces ORBmatcher::SearchByBoW for `batch` independent pairs, in both of its forms:
 *   th_low_inclusive = 1, d_valid2 = NULL: SearchByBoW(KeyFrame* pKF, Frame &F, ...) (src/ORBmatcher.cc:
 *     158-295) with 1 = the keyframe, 2 = the frame; vpMapPointMatches[iF] = the keyframe's map point
 *     d_matches21[iF] (or NULL where that is -1);
 *   th_low_inclusive = 0: SearchByBoW(KeyFrame*, KeyFrame*, ...) (:726-861); vpMatches12[i1] = map point
 *     d_matches12[i1] of keyframe 2.
 * d_valid1 / d_valid2 (cap bytes per frame): the keyframe's map point exists and is not bad.
 * FeatureVectors in the layout of orbhip_feature_vector_device.  d_nmatches (batch): the return value.
 * Device pointers; asynchronous. */
int orbhip_search_by_bow_device(orbhip_matcher* m, const OrbKeypoint* d_kp1, const uint8_t* d_desc1, const int32_t* d_n1,
                                const uint8_t* d_valid1, const int32_t* d_fv1_nodes, const int32_t* d_fv1_start,
                                const int32_t* d_fv1_feat, const int32_t* d_fv1_count, const OrbKeypoint* d_kp2,
                                const uint8_t* d_desc2, const int32_t* d_n2, const uint8_t* d_valid2,
                                const int32_t* d_fv2_nodes, const int32_t* d_fv2_start, const int32_t* d_fv2_feat,
                                const int32_t* d_fv2_count, int batch, int cap, int node_cap, float nnratio,
                                int check_orientation, int th_low_inclusive, int32_t* d_matches12, int32_t* d_matches21,
                                int32_t* d_nmatches);

/* Replaces ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:863-1025) for `batch` independent keyframe
 * pairs: features of keyframe 1 and keyframe 2 that hold no map point (d_free1 / d_free2 non-zero, cap bytes per
 * frame) and lie under the same vocabulary node are paired — candidates at most TH_LOW away, ordered by (distance,
 * index), the first within twice the least distance that passes ORBmatcher::CheckDistEpipolarLine (:139-156) —
 * a feature of keyframe 2 at most once, then the rotation-histogram filter.  d_F12: the fundamental matrix of each
 * pair (9 floats, row-major, as LocalMapping::ComputeF12 returns it); d_level_sigma2: KeyFrame::GetSigma2(level) for
 * level < nlevels, shared by all pairs (octaves outside are clamped).  FeatureVectors in the layout of
 * orbhip_feature_vector_device.  Outputs: d_matches12 (batch*cap) = vMatches12 (the vMatchedPairs of :1014-1022 are
 * its entries >= 0 in index order), d_matches21 (batch*cap, the inverse), d_nmatches (batch) = the return value.
 * Device pointers; asynchronous. */
int orbhip_search_for_triangulation_device(orbhip_matcher* m, const OrbKeypoint* d_kp1, const uint8_t* d_desc1,
                                           const int32_t* d_n1, const uint8_t* d_free1, const int32_t* d_fv1_nodes,
                                           const int32_t* d_fv1_start, const int32_t* d_fv1_feat, const int32_t* d_fv1_count,
                                           const OrbKeypoint* d_kp2, const uint8_t* d_desc2, const int32_t* d_n2,
                                           const uint8_t* d_free2, const int32_t* d_fv2_nodes, const int32_t* d_fv2_start,
                                           const int32_t* d_fv2_feat, const int32_t* d_fv2_count, int batch, int cap, int node_cap,
                                           const float* d_F12, const float* d_level_sigma2, int nlevels, int check_orientation,
                                           int32_t* d_matches12, int32_t* d_matches21, int32_t* d_nmatches);
/* Host-pointer form for ONE pair (copy in, match, copy out; blocking): what a forwarding
 * ORBmatcher::SearchForTriangulation calls.  An octave of keyframe 2 outside [0, nlevels) is ORBHIP_EINVAL. */
int orbhip_search_for_triangulation(orbhip_matcher* m, const OrbKeypoint* kp1, const uint8_t* desc1, int n1, const uint8_t* free1,
                                    const int32_t* fv1_nodes, const int32_t* fv1_start, const int32_t* fv1_feat, int fv1_count,
                                    const OrbKeypoint* kp2, const uint8_t* desc2, int n2, const uint8_t* free2,
                                    const int32_t* fv2_nodes, const int32_t* fv2_start, const int32_t* fv2_feat, int fv2_count,
                                    const float* F12, const float* level_sigma2, int nlevels, int check_orientation,
                                    int32_t* matches12, int32_t* nmatches);

/* Host-pointer form of orbhip_search_by_bow_device for ONE pair (copy in, match, copy out; blocking): what a
 * forwarding ORBmatcher::SearchByBoW (src/ORBmatcher.cc:158-295 with th_low_inclusive = 1 and valid2 = NULL,
 * :726-861 with th_low_inclusive = 0) calls.  A DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>,
 * Thirdparty/DBoW2/DBoW2/FeatureVector.h) arrives flattened in map order: fv_nodes[j] ascending, the features of
 * node j at fv_feat[fv_start[j] ... fv_start[j+1]), fv_start[0] = 0.  matches12 (n1) / matches21 (n2) as in the
 * device form. */
int orbhip_search_by_bow(orbhip_matcher* m, const OrbKeypoint* kp1, const uint8_t* desc1, int n1, const uint8_t* valid1,
                         const int32_t* fv1_nodes, const int32_t* fv1_start, const int32_t* fv1_feat, int fv1_count,
                         const OrbKeypoint* kp2, const uint8_t* desc2, int n2, const uint8_t* valid2, const int32_t* fv2_nodes,
                         const int32_t* fv2_start, const int32_t* fv2_feat, int fv2_count, float nnratio, int check_orientation,
                         int th_low_inclusive, int32_t* matches12, int32_t* matches21, int32_t* nmatches);
/* Host-pointer form of orbhip_distinctive_descriptors_device (blocking): the descriptor choice of
 * MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:221-250) for npoints map points, start[0] = 0. */
int orbhip_distinctive_descriptors(orbhip_matcher* m, const uint8_t* desc, const int32_t* start, int npoints, int32_t* best_idx,
                                   int32_t* best_median);
/* Host-pointer form of orbhip_undistort_keypoints_device for ONE frame (blocking): Frame::UndistortKeyPoints
 * (src/Frame.cc:290-320), mvKeysUn from mvKeys.  kp_un == kp is allowed. */
int orbhip_undistort_keypoints(orbhip_matcher* m, const OrbKeypoint* kp, int n, OrbCamera cam, OrbKeypoint* kp_un);
/* Host-pointer form of orbhip_bow_transform_device (blocking): the per-feature part of
 * mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4) in Frame::ComputeBoW (src/Frame.cc:281-288);
 * the caller accumulates BowVector / FeatureVector from the triples in feature order
 * (TemplatedVocabulary.h:1150-1163). */
int orbhip_bow_transform(orbhip_matcher* m, const orbhip_vocabulary* v, const uint8_t* desc, int n, int levelsup, int32_t* word,
                         int32_t* node, double* weight);

/* Device self-tests of the arithmetic helpers (test instrumentation, host pointers, blocking):
 * which 0: n (y, x) float pairs -> n floats, the device's fastAtan2 in degrees
 *       1: n float angles (radians) -> n (cos, sin) float pairs as the descriptor kernel computes them
 *       2: n uint32 keys (score in bits 31..24), aux = nth -> the keys as the device's
 *          nth_element leaves them
 *       3: n 7x7 u8 patches -> n int32 FAST corner measures of the patch centres
 *       4: as 2 with the wave-cooperative selection the extractor runs (n <= 4096)
 *       5: as 4 with the introselect depth limit forced to 1 (reaches the heap-select branch) */
int orbhip_selftest(int which, const void* in, int n, int aux, void* out);

/* Library/device facts. */
int orbhip_device_count(void);
const char* orbhip_version(void);
const char* orbhip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
