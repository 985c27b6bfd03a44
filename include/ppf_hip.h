/*
 * ppf_hip.h — C-ABI of the MI355X-native PPF matching / voting engine (libppf_hip.so).
 *
 * This is the drop-in boundary for the one path the reference accelerates badly: everything
 * /root/reference/include/CloudProcessing.h does through cv::ppf_match_3d::PPF3DDetector.
 * Plain pointers and sizes only; no C++/torch/OpenCV/PCL types.  The header-only C++ facades
 * (include/ppf_match_3d.hpp — OpenCV-shaped, what the reference calls; include/ppf_pcl.hpp —
 * PCL-shaped, what BASELINE.json's north_star names) sit on top of exactly these entry points.
 *
 *   entry point                     replaces (reference call site)
 *   ------------------------------  ---------------------------------------------------------------
 *   ppf_model_train                 PPF3DDetector(relSampling, relDistance) + trainModel(Mat)
 *                                   CloudProcessing.h:205,217,234 (ctor), :236 (trainModel)
 *   ppf_model_retain/_release       by-value detector copies and explicit dtor calls
 *                                   CloudProcessing.h:81,206,218,240,432,485
 *   ppf_model_save / ppf_model_load detector.write(FileStorage) :250 / detector.read(FileNode) :112
 *   ppf_match                       detector.match(scene, results, step, dist) :442  (edge == NULL)
 *                                   detector.match_S2B(scene, edge, results, step, dist) :495
 *   ppf_raw_votes                   the per-reference-point argmax inside match() — the bit-exact
 *                                   parity surface {refIndMax, alphaIndMax, maxVotes}
 *   ppf_match_device (+workspace)   same as ppf_match with clouds already resident in HBM and an
 *                                   explicit HIP stream: the entry bench.py times
 *   ppf_match_batch                 the per-object loop around Matching (CloudProcessing.h:41-45 holds one cloud
 *                                   per detected object, :58 one detector per model): crops x models
 *   ppf_pair_features               pcl::PPFEstimation::compute (north_star's PCL names; the reference never calls it)
 *   ppf_sample_cloud                samplePCByQuantization inside trainModel/match (A2)
 *   ppf_transform_pc_pose           transformPCPose, src/YOLO_cropping_ppf_test.cpp:125
 *   ppf_icp_refine (+_device)       ICP icp(100, 0.005f, 2.5f, 8); icp.registerModelToScene(model, scene, poses)
 *                                   CloudProcessing.h:465-470 (Matching), :518-523 (Matching_S2B)
 *   ppf_icp_register                ICP::registerModelToScene(src, dst, residual, pose), the single-pose overload
 *   ppf_prep_crop / _voxel_grid /   CloudProcessor::SceneCropping :263, Subsampling :361, OutlierProcessing :341,
 *   _outlier_removal / _normals /   NormalEstimation :381, EdgeExtraction :406, PointCloudXYZNormalToMat :163
 *   _edges / _to_mat (+ppf_cloud_*) (the PCL stages that produce the matcher's input)
 *
 * Conventions
 *   - A cloud argument is (pointer, rows, stride, normal_offset): float32 rows whose first three floats
 *     are x y z and whose normal sits `normal_offset` floats into the row; `stride` is the row pitch in
 *     floats.  The N x 6 CV_32FC1 Mat that CloudProcessing.h:163-190 builds is (6, PPF_NOFF_MAT = 3);
 *     a pcl::PointCloud<pcl::PointNormal> -- x y z 1 | nx ny nz 0 | curvature pad pad pad, the input of
 *     that function -- is (12, PPF_NOFF_PCL = 4).  Both host layouts pass without a repack.
 *   - Every function returns a ppf_status; no exception crosses this boundary.
 *     ppf_last_error() returns the calling thread's last message.
 *   - "No pose found" is not an error: *n_out = 0 (the wrapper handles it, :450-454).
 *   - Matching with an untrained/NULL model fails with PPF_ERR_NOT_TRAINED before any work
 *     (the wrapper's pre-check, :435-439).
 *   - Handles are immutable after training and reference counted: concurrent ppf_match calls
 *     on one model from several host threads / streams are safe; each call (or workspace)
 *     owns its scratch memory.
 *   - There is NO CPU fallback: without a usable HIP device every compute entry point
 *     returns PPF_ERR_HIP.
 */
#ifndef PPF_HIP_H
#define PPF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPF_ABI_VERSION 4 /* 4: every cloud argument carries its normal offset (pcl::PointNormal without a repack) */

#define PPF_NOFF_MAT 3 /* x y z nx ny nz */
#define PPF_NOFF_PCL 4 /* pcl::PointNormal: x y z pad nx ny nz pad curvature pad pad pad (stride 12) */

typedef enum ppf_status {
  PPF_OK = 0,
  PPF_ERR_INVALID = 1,     /* bad argument (CV_Assert in the reference's library) */
  PPF_ERR_NOT_TRAINED = 2, /* match on an untrained model */
  PPF_ERR_HIP = 3,         /* HIP runtime failure / no device */
  PPF_ERR_NOMEM = 4,
  PPF_ERR_IO = 5,
  PPF_ERR_CAPACITY = 6     /* caller's output buffer too small; *n_out holds the needed count */
} ppf_status;

typedef struct ppf_model ppf_model;         /* opaque, ref-counted, device-resident model table */
typedef struct ppf_workspace ppf_workspace; /* opaque per-caller scratch + result buffers */
typedef struct ppf_batch ppf_batch;         /* opaque: streams + workspaces of a batched call (crops x models) */
typedef struct ppf_cloud ppf_cloud;         /* opaque device-resident cloud: rows x y z nx ny nz + curvature */

typedef struct ppf_train_params {
  double relative_sampling_step;  /* PPF3DDetector ctor arg 1 (reference: 0.025, CloudProcessing.h:64) */
  double relative_distance_step;  /* ctor arg 2 (reference: 0.05, :65; TrainDetector default 0.5, :223) */
  double num_angles;              /* ctor arg 3, default 30 */
  int32_t presampled;             /* 1: rows are already the sampled model, skip samplePCByQuantization */
  int32_t distance_from_distance_step; /* 0 (default): distance step = diameter*relative_sampling_step,
                                          as the reference's library computes it; 1: use relative_distance_step */
  int32_t max_tile_refs;          /* 0 = auto: model reference points per LDS accumulator tile */
  int32_t key_equality;           /* PPF_KEY_BUCKET (0, the reference's library): a scene pair votes for every model pair
                                     in its hash bucket, colliding keys included; PPF_KEY_EXACT (1, PCL PPFHashMapSearch):
                                     only for model pairs with the same quantised key */
  int32_t feature;                /* PPF_FEATURE_PPF (0, the reference's library): three acos angles + distance, truncated;
                                     PPF_FEATURE_DARBOUX (1, PCL PPFEstimation / pcl::computePairFeatures): atan2 angle and
                                     two cosines of the Darboux frame + distance, keys floor(f / step).  A property of the
                                     trained table; matching follows the model. */
  int32_t reserved;               /* 0 */
} ppf_train_params;
#define PPF_KEY_BUCKET 0
#define PPF_KEY_EXACT 1
#define PPF_FEATURE_PPF 0
#define PPF_FEATURE_DARBOUX 1

typedef struct ppf_match_params {
  double relative_scene_sample_step; /* match() arg 3: every (int)(1/x)-th sampled scene point is a reference */
  double relative_scene_distance;    /* match() arg 4: scene sampling step, relative to the scene bbox */
  double position_threshold;         /* setSearchParams; < 0 = default (relative_sampling_step) */
  double rotation_threshold;         /* setSearchParams; < 0 = default ((360/angle_step)/180*pi) */
  int32_t use_weighted_avg;          /* setSearchParams */
  int32_t presampled;                /* 1: scene (and edge) rows are already sampled: every row votes */
  int32_t ref_offset;                /* sharding over ranks: vote reference points ref_offset, */
  int32_t ref_stride;                /*   ref_offset+ref_stride, ... of the reference list (1 = all) */
  int32_t skip_clustering;           /* 1: stop after per-reference poses */
  int32_t vote_mode;                 /* PPF_VOTE_AUTO (0): runs of many hits vote through per-run count tables;
                                        PPF_VOTE_DIRECT (1): every (entry, hit) pair casts its own atomic.  Same results. */
  /* PCL-semantics policy switches (pcl::PPFRegistration; zero = what the reference's library does) */
  double pair_radius;                /* > 0: a reference point is only paired with points at most this far away (PCL searches
                                        model_diameter / 2 around it); <= 0: with every point */
  int32_t rot_metric_relative;       /* 1: poses cluster when the angle of their RELATIVE rotation is below
                                        rotation_threshold (PCL); 0: when their rotation angles differ by less (OpenCV) */
  int32_t alpha_range_2pi;           /* 1: alpha_m - alpha_s is wrapped into [-pi, pi] and binned over 2 pi, numAngles bins of
                                        2 pi / numAngles (PCL); 0: the unwrapped difference over 4 pi (OpenCV) */
} ppf_match_params;
#define PPF_VOTE_AUTO 0
#define PPF_VOTE_DIRECT 1

/* cv::ppf_match_3d::Pose3D fields the reference reads (src/YOLO_cropping_ppf_test.cpp:124-125). */
typedef struct ppf_pose {
  double pose[16]; /* 4x4 row-major, model -> scene */
  double q[4];     /* quaternion [w x y z] */
  double t[3];
  double angle;
  double alpha;
  double residual;
  uint32_t model_index;
  uint32_t num_votes;
} ppf_pose;

/* argmax of one scene reference point's accumulator */
typedef struct ppf_vote {
  uint32_t ref_ind_max;
  uint32_t alpha_ind_max;
  uint32_t max_votes;
} ppf_vote;

typedef struct ppf_model_info {
  int32_t n_ref;         /* sampled model points N_m */
  int32_t num_angles;    /* floor(2*pi/angle_step) */
  uint32_t slots;        /* next_pow2(N_m^2) hash slots of the reference's table */
  uint32_t n_buckets;    /* non-empty slots */
  uint64_t n_entries;    /* N_m*(N_m-1) (+ spill duplicates) */
  int32_t n_tiles;       /* accumulator tiles */
  int32_t tile_refs;     /* model reference points per tile */
  double angle_step;     /* radians */
  double distance_step;  /* metres (float-rounded like the reference) */
  double diameter;       /* model bbox diagonal */
  double position_threshold_default;
  double rotation_threshold_default;
  uint64_t device_bytes; /* HBM held by the model */
} ppf_model_info;

/* counters of the last match executed in a workspace */
typedef struct ppf_match_stats {
  int32_t n_scene_sampled; /* rows after scene sampling */
  int32_t n_paired;        /* rows of the paired cloud (== n_scene_sampled unless S2B) */
  int32_t n_ref;           /* reference points voted by this call */
  int32_t n_poses;         /* clustered poses */
  uint64_t n_pairs;        /* scene pairs hashed and looked up */
  uint64_t n_votes;        /* accumulator increments == pair-matches */
  float ms_vote_kernel;    /* device time of the voting kernel, summed over the batches of the call (timing enabled) */
  float ms_pair_kernel;    /* pair kernel, likewise */
  float ms_total_device;   /* first kernel start -> last kernel end */
  float ms_group_kernel;   /* hit grouping (k_group, the two rankings, the count tables), likewise */
  uint64_t n_hits;         /* scene pairs that found a non-empty bucket */
  uint64_t n_lds_atomics;  /* LDS atomic lane-operations the voting kernel issued (<= n_votes when runs vote by counts) */
  uint64_t scratch_bytes;  /* device scratch of the call: hit pools, run table, frames */
  int32_t n_batches;       /* batches of reference points the call was cut into */
  int32_t n_retries;       /* repeats because the hit pools (sized from earlier calls) were too small */
  uint64_t n_acc32_items;  /* (reference point, accumulator tile)s voted with 32-bit cells: those whose 16-bit cells overflowed, or all
                              of them once a workspace has seen a twentieth of a call's votes cast twice (or with PPF_OPT_ACC32 = 1) */
  uint64_t n_tables;       /* count tables built for the runs of many hits (one per 191 hits of such a run) */
  uint64_t phase_clocks[8]; /* zero, except in a diagnostic build of the library (-DPPF_PHASE_CLOCKS): shader clocks the voting kernel's waves spent per phase */
} ppf_match_stats;

/* totals of one ppf_batch_run */
typedef struct ppf_batch_stats {
  uint64_t n_pairs, n_votes, n_hits, n_lds_atomics;
  int32_t n_matches; /* crops x models */
  int32_t n_retries;
  int32_t lanes;
  float ms_wall;     /* host wall clock of the call */
  /* with ppf_batch_enable_timing: device time of the kernels summed over all matches of the run (HIP events on each lane's
   * stream; lanes overlap, so the sums exceed the wall clock) */
  float ms_vote_kernel, ms_pair_kernel, ms_group_kernel;
  int32_t reserved;
} ppf_batch_stats;

/* cv::ppf_match_3d::ICP constructor arguments (uniform sampling, one correspondence per point) */
typedef struct ppf_icp_params {
  int32_t iterations;    /* ICP ctor arg 1 (reference: 100, CloudProcessing.h:465) */
  float tolerance;       /* arg 2 (0.005f) */
  float rejection_scale; /* arg 3 (2.5f); <= 0 disables the median+MAD rejection */
  int32_t num_levels;    /* arg 4 (8) */
  int32_t flags;         /* 0: all poses of a call advance through the same launches, two per iteration, neighbours from a
                            grid search.  PPF_ICP_LEGACY: the earlier schedule (one stream per pose, seven launches per
                            iteration, exhaustive neighbour search); with it PPF_ICP_NO_SMALL_LEVELS (coarse levels kernel by
                            kernel instead of in one workgroup) and PPF_ICP_ONE_STREAM (the poses share the caller's stream).
                            Same results whichever way (tests/test_gpu_robustness.py). */
  int32_t reserved[3];
} ppf_icp_params;
#define PPF_ICP_NO_SMALL_LEVELS 1
#define PPF_ICP_ONE_STREAM 2
#define PPF_ICP_LEGACY 4
#define PPF_ICP_GRID_ALWAYS 8 /* test knob: the grid neighbour search on every level (by default levels of at most 1,024 scene rows scan them all) */

void ppf_default_train_params(ppf_train_params* p);
void ppf_default_match_params(ppf_match_params* p);
void ppf_default_icp_params(ppf_icp_params* p); /* 100, 0.005f, 2.5f, 8: the reference's ICP object */
int ppf_abi_version(void);
/* copies the calling thread's last error text; returns its length */
int ppf_last_error(char* buf, int cap);
/* number of visible HIP devices (0 when there is none); never fails */
int ppf_device_count(void);

/* ---- model ------------------------------------------------------------------------------ */
ppf_status ppf_model_train(const float* xyzn, int n, int stride, int normal_offset, const ppf_train_params* params, ppf_model** out);
ppf_status ppf_model_retain(ppf_model* m);
ppf_status ppf_model_release(ppf_model* m);
ppf_status ppf_model_get_info(const ppf_model* m, ppf_model_info* info);
/* HIP device the model's table lives on (the current device of the thread that trained or loaded it) */
ppf_status ppf_model_get_device(const ppf_model* m, int* device);
/* pcl::PPFHashMapSearch::nearestNeighborSearch(f1, f2, f3, f4, indices) (north_star's PCL names; the reference never calls it):
 * the pairs (i, j) of the trained model's sampled points whose quantised feature equals the quantised f4[0..3] -- what a hash
 * map keyed on the quantised feature holds under that key -- ascending by (i, j), as pairs_ij[2q], pairs_ij[2q + 1].  The
 * feature kind and the steps are the model's (ppf_train_params.feature, ppf_model_info.angle_step / distance_step).
 * cap_pairs = 0 only counts (*n_out). */
ppf_status ppf_model_nearest_pairs(const ppf_model* m, const float* f4, uint32_t* pairs_ij, int cap_pairs, int* n_out);
/* The host-buffer entries (ppf_match, ppf_raw_votes, ppf_match_clouds) keep warm contexts with the model -- one per call that
 * has been in flight at once (at least 2, at most 16), each holding a stream, pinned staging and the scratch of its last call
 * (0.45 GB for a 50,000-point crop).  This releases the idle ones beyond `keep` (0: all) and returns how many went; calls in
 * flight are not touched, the next call on a model without an idle context simply starts cold.  The reference has no
 * counterpart: its detector frees nothing until it is destroyed (/root/reference/include/CloudProcessing.h:79-83). */
ppf_status ppf_model_trim_contexts(const ppf_model* m, int keep, int* released);
/* sampled model cloud (n_ref x 6 floats) */
ppf_status ppf_model_get_sampled(const ppf_model* m, float* out, int cap_rows);
/* CSR dump for inspection/tests: any pointer may be NULL. bucket_off has n_tiles*(n_buckets+1) u32,
 * entries n_entries x {int32 cell_base, float alpha_m}. */
ppf_status ppf_model_get_table(const ppf_model* m, uint32_t* bucket_slot, uint32_t* bucket_off, int32_t* entry_cell,
                               float* entry_alpha);
ppf_status ppf_model_save(const ppf_model* m, const char* path);
/* every field and table of the file is validated before use: a truncated or corrupt file is PPF_ERR_IO */
ppf_status ppf_model_load(const char* path, ppf_model** out);
/* The same byte stream to and from memory (what cv::FileStorage carries for detector.write / detector.read,
 * CloudProcessing.h:112,250): buf == NULL queries *size; a too small buffer is PPF_ERR_CAPACITY with *size set.
 * ppf_model_load_mem validates exactly like ppf_model_load. */
ppf_status ppf_model_save_mem(const ppf_model* m, void* buf, size_t cap, size_t* size);
ppf_status ppf_model_load_mem(const void* buf, size_t size, ppf_model** out);
/* the same validation without a device (host only): PPF_OK or PPF_ERR_IO */
ppf_status ppf_model_check_file(const char* path);

/* pcl::PPFEstimation::compute: the n x n pair features of a cloud as float32 rows of five, row i*n + j =
 * {f1, f2, f3, f4, alpha_m} of the pair (i, j) -- `feature` PPF_FEATURE_PPF: the three acos angles and the distance of the
 * reference's library; PPF_FEATURE_DARBOUX: pcl::computePairFeatures' values; alpha_m in the engine's frame convention.
 * Rows i == j and degenerate pairs are NaN.  cap_rows: rows `out` can hold (>= n*n).  Computed on the device. */
ppf_status ppf_pair_features(const float* xyzn, int n, int stride, int normal_offset, int feature, float* out, size_t cap_rows);

/* ---- matching, host buffers ------------------------------------------------------------- */
ppf_status ppf_match(const ppf_model* m, const float* scene, int ns, int sstride, int snoff, const float* edge, int ne,
                     int estride, int enoff, const ppf_match_params* params, ppf_pose* out, int cap, int* n_out);
ppf_status ppf_raw_votes(const ppf_model* m, const float* scene, int ns, int sstride, int snoff, const float* edge, int ne,
                         int estride, int enoff, const ppf_match_params* params, ppf_vote* votes, ppf_pose* raw_poses, int cap,
                         int* n_ref, ppf_match_stats* stats);

/* Many crops x many models (BASELINE config C5): every scene is uploaded and sampled once and matched against
 * every model.  out holds n_scenes x n_models blocks of `cap` poses (best first), n_out the count of each block.
 * (= ppf_batch_create(min(4, n_scenes)) + ppf_batch_run + ppf_batch_destroy) */
ppf_status ppf_match_batch(const ppf_model* const* models, int n_models, const float* const* scenes, const int* ns,
                           int sstride, int snoff, int n_scenes, const ppf_match_params* params, ppf_pose* out, int cap, int* n_out);
/* The reusable form: `lanes` HIP streams, each with its own workspace and pinned staging.  Crop c runs on lane c mod
 * lanes, its matches against all models back to back without host involvement; one synchronisation at the end.
 * scenes_on_device != 0: `scenes` are device pointers (no staging).  out / n_out / stats may be NULL. */
ppf_status ppf_batch_create(int lanes, ppf_batch** out);
ppf_status ppf_batch_destroy(ppf_batch* b);
/* record HIP events around the kernels of every match of the following runs (ppf_batch_stats.ms_*_kernel) */
ppf_status ppf_batch_enable_timing(ppf_batch* b, int on);
ppf_status ppf_batch_run(ppf_batch* b, const ppf_model* const* models, int n_models, const float* const* scenes, const int* ns,
                         int sstride, int snoff, int n_scenes, int scenes_on_device, const ppf_match_params* params, ppf_pose* out,
                         int cap, int* n_out, ppf_batch_stats* stats);
/* device block of the last run: n_scenes * n_models * cap pose records (zero rows past each count), e.g. for a gather */
ppf_status ppf_batch_device_block(ppf_batch* b, void** d_poses, int* n_records);
/* device-to-device copy of the first n_records of that block into d_dst, enqueued on `stream` */
ppf_status ppf_batch_copy_block(ppf_batch* b, void* d_dst, int n_records, void* stream);

/* ---- matching, device-resident clouds + explicit stream ---------------------------------- */
ppf_status ppf_workspace_create(ppf_workspace** out);
ppf_status ppf_workspace_destroy(ppf_workspace* ws);
/* Tuning / test knobs of a workspace.  PPF_OPT_HIT_FRACTION: expected hits per scene pair, which sizes the hit pools of
 * the next call (normally learned from the previous calls; a too small value only costs a repeat of the call).
 * PPF_OPT_GROUP_ROUND_BUCKETS: bucket ids grouped per pass over a reference point's hits (0 = as many as fit LDS). */
#define PPF_OPT_HIT_FRACTION 1
#define PPF_OPT_GROUP_ROUND_BUCKETS 2
#define PPF_OPT_CLUSTER_SERIAL 3 /* != 0: the serial greedy cluster assignment (the path for > 11,520 poses) for any size */
#define PPF_OPT_ACC32 4          /* 0 (default): 16-bit accumulator cells first, 32-bit cells for the (reference point, tile)s whose cells overflow;
                                   a workspace that sees a twentieth of a call's votes cast twice that way goes to 32-bit cells for
                                   everything; 1: 32-bit cells for everything from the first call; 2: 16-bit cells first, always;
                                   3: the (reference point, tile)s that will cast more votes than a limit learned from the previous
                                   call go straight to 32-bit cells (measured slower than 0 on BASELINE's C4: kept for the comparison) */
#define PPF_OPT_TABLE_FRACTION 5 /* expected count tables per hit (sizes the table pool of the next call; learned from then on) */
#define PPF_OPT_BATCH_REFS 6     /* > 0: at most this many reference points per batch of a call (default: what 4 GB of hit scratch hold); a test knob */
#define PPF_OPT_RUN_STAGING 7    /* > 0: the vote kernel stages at most this many runs of a reference point per segment (rounded down to a multiple of 64, at
                                   least 64; default: what the LDS holds next to the model's accumulator tile, 704 .. 1,024); a test knob: several segments
                                   per reference point on small scenes */
ppf_status ppf_workspace_set_option(ppf_workspace* ws, int option, double value);
/* record HIP events around the kernels of each call (read back through ppf_workspace_results' stats) */
ppf_status ppf_workspace_enable_timing(ppf_workspace* ws, int on);
/* Enqueue sampling + voting + pose assembly (+ clustering) on `stream` (a hipStream_t, NULL = default
 * stream).  d_scene/d_edge are DEVICE pointers.  Returns after enqueueing when everything could be
 * sized without a host round trip (presampled clouds); results stay in the workspace. */
ppf_status ppf_match_device(const ppf_model* m, ppf_workspace* ws, const float* d_scene, int ns, int sstride, int snoff,
                            const float* d_edge, int ne, int estride, int enoff, const ppf_match_params* params, void* stream);
/* Wait for the workspace's last call and copy results out (any pointer may be NULL). */
ppf_status ppf_workspace_results(ppf_workspace* ws, ppf_vote* votes, ppf_pose* raw_poses, int cap_ref, int* n_ref,
                                 ppf_pose* poses, int cap_poses, int* n_poses, ppf_match_stats* stats);
/* exact per-reference-point counters of the last call: accumulator increments and pairs hashed */
ppf_status ppf_workspace_ref_counters(ppf_workspace* ws, uint64_t* votes_per_ref, uint64_t* pairs_per_ref, int cap);
/* Full accumulators (n_ref x n_model*num_angles u32, the reference's `accumulator` array before its
 * argmax scan) of the voted reference points; presampled clouds only.  Debug / parity surface. */
ppf_status ppf_debug_accumulators(const ppf_model* m, const float* scene, int ns, int sstride, int snoff, const float* edge, int ne,
                                  int estride, int enoff, const ppf_match_params* params, uint32_t* acc, size_t cap_words,
                                  int* n_ref);
/* Size of the cached device block a request of `bytes` is served from (host only, no device needed): classes of 1/8
 * octave, so at most 12.5 % more than asked for.  Test surface of the block cache's keying. */
size_t ppf_debug_block_size(size_t bytes);
/* Evaluate include/ppf_detmath.h on the device: fn 0 acos(x), 1 sin(x), 2 cos(x), 3 atan2(x, y), 4 sqrt(x),
 * 5 x / y.  The bit patterns must equal the host's (tests/test_gpu_detmath.py). */
ppf_status ppf_debug_device_math(int fn, const double* x, const double* y, double* out, int n);
/* device pointer to the per-reference pose records of the last call (n_ref x ppf_pose), for a
 * collective gather without a host copy */
ppf_status ppf_workspace_device_poses(ppf_workspace* ws, void** d_raw_poses, int* n_ref);
/* Device-side result blocks, enqueued on `stream` (call ppf_workspace_results first: it is what notices and repeats a
 * call whose hit pools were too small).  d_dst receives k (cap) ppf_pose records, zero rows past the available count:
 * the best k clustered poses / the per-reference poses.  For collectives that gather from device memory. */
ppf_status ppf_workspace_copy_top_poses(ppf_workspace* ws, void* d_dst, int k, void* stream);
ppf_status ppf_workspace_copy_raw_poses(ppf_workspace* ws, void* d_dst, int cap, void* stream);
/* clusterPoses on a DEVICE pose list (e.g. the all-gathered per-reference poses of all ranks), enqueued on `stream`;
 * the clusters are fetched with ppf_workspace_results(poses) or ppf_workspace_copy_top_poses */
ppf_status ppf_cluster_poses_device(const ppf_model* m, ppf_workspace* ws, const void* d_in, int n, int num_poses,
                                    const ppf_match_params* params, void* stream);
/* cluster a caller-supplied pose list (e.g. the all-gathered per-reference poses of all ranks) */
ppf_status ppf_cluster_poses(const ppf_model* m, const ppf_pose* in, int n, int num_poses,
                             const ppf_match_params* params, ppf_pose* out, int cap, int* n_out);

/* ---- helpers on the path's edges --------------------------------------------------------- */
/* samplePCByQuantization: returns rows through *n_out (out may be NULL to query) */
ppf_status ppf_sample_cloud(const float* xyzn, int n, int stride, int normal_offset, double relative_step, float* out, int cap_rows,
                            int* n_out);
/* out: n x 6 packed rows */
ppf_status ppf_transform_pc_pose(const float* xyzn, int n, int stride, int normal_offset, const double* pose16, float* out);


/* ---- ICP refinement of matched poses (the step right after the path) ---------------------- */
/* registerModelToScene(model, scene, poses): every pose moves the model, a multi-level point-to-plane ICP registers
 * the moved model to the scene, and the pose becomes poseICP * pose (pose/q/t/angle/residual are rewritten, votes
 * and model_index kept).  iterations_out (optional) receives the iterations spent per pose.  Host clouds. */
ppf_status ppf_icp_refine(const float* model, int n_model, int mstride, int mnoff, const float* scene, int n_scene, int sstride,
                          int snoff, const ppf_icp_params* params, ppf_pose* poses_io, int n_poses, int* iterations_out);
/* same with DEVICE-resident clouds and an explicit hipStream_t (poses_io stays on the host) */
ppf_status ppf_icp_refine_device(const float* d_model, int n_model, int mstride, int mnoff, const float* d_scene, int n_scene,
                                 int sstride, int snoff, const ppf_icp_params* params, ppf_pose* poses_io, int n_poses,
                                 int* iterations_out, void* stream);
/* registerModelToScene(src, dst, residual, pose): one registration without an initial pose */
ppf_status ppf_icp_register(const float* src, int n_src, int sstride, int snoff, const float* dst, int n_dst, int dstride, int dnoff,
                            const ppf_icp_params* params, double* pose16_out, double* residual_out, int* iterations_out);

/* ---- the producers of the N x 6 input (CloudProcessing.h:263-427), device-resident between stages ---------- */
/* cols = 3 (xyz; normals zero; normal_offset ignored) or 6 (xyz + the normal at normal_offset), `stride` floats between rows */
ppf_status ppf_cloud_upload(const float* rows, int n, int stride, int normal_offset, int cols, ppf_cloud** out);
ppf_status ppf_cloud_release(ppf_cloud* c);
ppf_status ppf_cloud_size(const ppf_cloud* c, int* n);
/* rows6: n x 6 floats, curvature: n floats; either may be NULL */
ppf_status ppf_cloud_download(const ppf_cloud* c, float* rows6, float* curvature, int cap_rows);
/* device pointer to the packed n x 6 rows (valid while the cloud lives): feeds ppf_match_device / ppf_icp_refine_device */
ppf_status ppf_cloud_device_rows(const ppf_cloud* c, const float** d_rows6, int* n);
/* SceneCropping (:263-339) for one box {x, y, width, height}: +-30 px, mean corner depth, corners pushed 0.15 m back,
 * points inside the pyramid {camera centre, 4 corners} are kept.  depth: HOST image (rows x cols float, metres),
 * intr = {fx, fy, ppx, ppy}. */
ppf_status ppf_prep_crop(const ppf_cloud* in, const int* box_xywh, const float* depth, int depth_rows, int depth_cols,
                         const double* intr, ppf_cloud** out);
/* Subsampling (:361-380): pcl::VoxelGrid, cubic leaf */
ppf_status ppf_prep_voxel_grid(const ppf_cloud* in, double leaf, ppf_cloud** out);
/* OutlierProcessing (:341-360): pcl::StatisticalOutlierRemoval(meanK, stddevMul) */
ppf_status ppf_prep_outlier_removal(const ppf_cloud* in, int mean_k, double stddev_mul, ppf_cloud** out);
/* NormalEstimation (:381-405): k-neighbour plane fit, normals towards the camera centre, curvature */
ppf_status ppf_prep_normals(const ppf_cloud* in, int k, ppf_cloud** out);
/* EdgeExtraction (:406-427): curvature > threshold */
ppf_status ppf_prep_edges(const ppf_cloud* in, float curvature_threshold, ppf_cloud** out);
/* PointCloudXYZNormalToMat (:163-190): rows with re-normalised normals */
ppf_status ppf_prep_to_mat(const ppf_cloud* in, ppf_cloud** out);
/* match / match_S2B (:442, :495) and the ICP step (:465-470, :518-523) on clouds that are already resident (the
 * outputs of ppf_prep_to_mat): crop -> ... -> edges -> match -> ICP without a host copy of any cloud */
ppf_status ppf_match_clouds(const ppf_model* m, const ppf_cloud* scene, const ppf_cloud* edge, const ppf_match_params* params,
                            ppf_pose* out, int cap, int* n_out);
ppf_status ppf_icp_refine_clouds(const ppf_cloud* model, const ppf_cloud* scene, const ppf_icp_params* params, ppf_pose* poses_io,
                                 int n_poses, int* iterations_out);
/* exact neighbour lists (parity surface): idx, d2 are [n][k], ascending (distance, index) */
ppf_status ppf_prep_knn(const ppf_cloud* in, int k, int* idx, float* d2);

#ifdef __cplusplus
}
#endif
#endif /* PPF_HIP_H */
