/*
 * ppf_hip.hip — MI355X (gfx950) implementation of the C-ABI in include/ppf_hip.h.
 *
 * Path: A2 cloud sampling -> A3 pair feature + key hash -> A5-train model table ->
 * A5-match per-reference-point Hough vote over the discretised alpha -> argmax -> A8 pose assembly ->
 * A7 pose clustering.  Reference call sites: /root/reference/include/CloudProcessing.h:236
 * (trainModel), :442 (match), :495 (match_S2B).
 *
 * Data layout in HBM:
 *   clouds        SoA  x[] y[] z[] nx[] ny[] nz[]  (f32, coalesced 256 B per wave-instruction)
 *   slot map      slots/64 x {u64 occupancy bits, u32 rank, u32 pad}: hash slot -> dense bucket id
 *                 in one 16-byte load (the reference indexes 2^k slots by hash % slots and never
 *                 compares keys, so only "which slots are non-empty" has to be kept)
 *   bucket_off    n_tiles x (n_buckets+1) u32 CSR offsets, one CSR per accumulator tile
 *   records       pair records {row_a, row_b, alpha_a, alpha_b}: row = u32 code of the model ref's accumulator row (LDS byte
 *                 offset (guard + word_row*numAngles)*4, the half of the word its 16-bit cells use in bit 0, the entry's
 *                 count-table X and cell above bit 17: vote_row_code, agg_cell_bits), alpha = f32 alpha_m; 8 B per model
 *                 pair; per (tile, bucket) low-half rows first, bank- and phase-interleaved inside each half (see
 *                 place_entry); bucket_off counts records
 *   accumulator   LDS, ceil(tile_refs/2) x numAngles words of two 16-bit cells per workgroup (one workgroup = one scene
 *                 reference point x one tile of model reference points; 32-bit cells: one half of the tile's rows)
 *
 * This file is the one translation unit of libppf_hip.so; the code lives in the headers included at the bottom:
 *   kernels   ppf_train_kernels.h (table build)  ppf_sample_kernels.h (A2)  ppf_match_kernels.h (k_frames / k_pairs /
 *             k_group / k_vote)  ppf_pose_kernels.h (k_finalize, k_rank, clustering)  ppf_icp_kernels.h  ppf_prep_kernels.h
 *   host      ppf_device_mem.h (errors, block cache)  ppf_host_common.h (scans, sorts, model / workspace structs)
 *             ppf_model_host.h  ppf_match_host.h  ppf_batch_host.h  ppf_icp_host.h  ppf_prep_host.h  (the C-ABI)
 *
 * Compile: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see __graft_entry__.build()).
 * No CPU fallback exists: without a HIP device the compute entry points return PPF_ERR_HIP.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/ppf_hip.h"
#include "ppf_core.h"

#include "ppf_device_mem.h"    /* fail(), HIPCHK, DevPool, DevBuf */
#include "ppf_train_kernels.h" /* accumulator geometry, CloudSoA, table-build kernels, scan */
#include "ppf_sample_kernels.h"
#include "ppf_match_kernels.h"
#include "ppf_icp_kernels.h"
#include "ppf_prep_kernels.h"
#include "ppf_pose_kernels.h"  /* k_finalize, k_rank, clustering, result blocks */
#include "ppf_host_common.h"   /* scans, sorts, sampling, ppf_model / ppf_workspace, enqueue_cluster */
#include "ppf_model_host.h"    /* C-ABI: defaults, training, handles, model file */
#include "ppf_match_host.h"    /* C-ABI: workspaces, ppf_match_device, ppf_match, ppf_raw_votes, clustering */
#include "ppf_batch_host.h"    /* C-ABI: crops x models */
#include "ppf_icp_host.h"      /* row N2: ICP refinement, host side */
#include "ppf_prep_host.h"     /* row N4: cloud stages, host side */
