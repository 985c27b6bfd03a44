/*
 * ppf_match_kernels.h — the matching hot path on gfx950, included by
 * ppf_hip.hip.  Reference call sites: /root/reference/include/CloudProcessing.h:442 (match), :495
 * (match_S2B).
 *
 * Four kernels per batch of scene reference points:
 *
 *   k_frames   one thread per reference point: the rotation/translation (Rsg, tsg) that takes the
 *              reference point to the origin with its normal on +x (fp64, 12 doubles per point).
 *
 *   k_pairs    one thread per scene pair (s_r, s_i): pair feature (fp64, deterministic math) ->
 *              4 x int32 key -> MurmurHash3 -> slot -> dense bucket id through the slot map (one
 *              16-byte load).  Only ~1-2 % of the pairs of a real crop land in a non-empty slot;
 *              for those the lane also computes alpha_s and the wave appends a 16-byte hit record
 *              {bucket, alpha_s} to the reference point's hit list (ballot + one atomic per wave).
 *              VALU(fp64)-bound; reads 24 B per pair from the L2-resident scene SoA.
 *
 *   k_group    one workgroup per reference point: stable LSD radix sort (8-bit digits, in L2-resident
 *              global memory) of the point's hit list by bucket id.  On a real crop many pairs of one
 *              reference point fall into the same few heavy buckets (measured: votes / distinct
 *              bucket entries = 31), so grouping them lets k_vote read a bucket once for all of them.
 *
 *   k_vote     one workgroup per (reference point, accumulator tile).  The tile's Hough accumulator
 *              lives in LDS.  Sorted hits form runs (one bucket, m hits with different alpha_s); a run
 *              is cut into work items of <= VOTE_CHUNK table entries x <= VOTE_MAX_HITS hits.  A wave
 *              loads a batch of 64 x VOTE_UNROLL entries (coalesced 8-byte loads, next batch
 *              prefetched into a second register set) and votes it once per hit of the item straight
 *              from registers: HBM/L2 traffic per vote drops from 8 B to 8/m B and the kernel is bound
 *              by the LDS atomic rate (one ds_add_u32 per vote, bank and same-cell conflicts included;
 *              measured: halving the VALU work per vote does not change its time).
 *
 * Alpha bin, exactly: bin = (int)(A*(alpha_m - alpha_s + 2pi)/(4pi)) in fp64 is what the reference
 * computes.  The fast path evaluates q = (alpha_m - alpha_s)*A/(4pi) + A/2 in fp32 (|error| <= 9e-8*A,
 * folded as alpha_m*S + (A/2 - alpha_s*S): |error| <= 1.1e-7*A) and takes trunc(q)
 * whenever q is farther than G = 5e-7*A from an integer; otherwise
 * (about 3e-5 of the votes) the lane re-evaluates the fp64 chain.  Both paths give the same integer.
 */
#ifndef PPF_MATCH_KERNELS_H
#define PPF_MATCH_KERNELS_H

#ifndef PPF_GUARD_REL
#define PPF_GUARD_REL 5e-7f /* alpha-bin guard band relative to A; the fp32 error bound is 1.1e-7 */
#endif
#ifndef PPF_ABL
#define PPF_ABL 0 /* diagnostic ablations of k_vote; 0 in every shipped build */
#endif

constexpr int PAIR_BLOCK = 256;
constexpr int PAIRS_PER_THREAD = 8;   /* one k_pairs workgroup covers 2048 paired points of one reference point */
constexpr int VOTE_BLOCK = 1024;
constexpr int VOTE_WAVES = VOTE_BLOCK / 64;
constexpr int VOTE_UNROLL = 4;        /* pair records (2 entries each) loaded per lane per batch */
#ifndef PPF_VOTE_CHUNK_BATCHES
#define PPF_VOTE_CHUNK_BATCHES 4
#endif
#ifndef PPF_VOTE_MAX_HITS
#define PPF_VOTE_MAX_HITS 16
#endif
#ifndef PPF_VOTE_DYNAMIC
#define PPF_VOTE_DYNAMIC 1
#endif
#ifndef PPF_PIPE_VALU
#define PPF_PIPE_VALU 4 /* VALU instructions scheduled between two LDS atomics of the pipelined vote loop */
#endif
#ifndef PPF_VOTE_FIXED
#define PPF_VOTE_FIXED 0 /* 1: 16.16 fixed-point alpha bins, 2.5-2.75 VALU per vote instead of 4.9 (vote_hits_fx).  Bit-exact (same
                          parity tests), but NOT faster on gfx950 today: with the VALU work halved the kernel sits on its LDS-atomic
                          bound (13.8 ms either way; 11.7 ms with conflict-free addresses), so the plain fp32 path stays the default */
#endif
#ifndef PPF_VOTE_PIPE
#define PPF_VOTE_PIPE 1 /* atomics of hit h issued under the arithmetic of hit h+1 (vote_hits) */
#endif
constexpr int VOTE_CHUNK = 64 * VOTE_UNROLL * PPF_VOTE_CHUNK_BATCHES; /* pair records per work item (1024 = 2048 entries) */
constexpr int VOTE_MAX_HITS = PPF_VOTE_MAX_HITS; /* hits of one bucket run voted per work item */
constexpr int GROUP_BLOCK = 1024;
constexpr int VOTE_SEG = VOTE_BLOCK;  /* hits staged in LDS per segment: one per thread */
constexpr int LDS_HEADER = 256;       /* bytes: reduction scratch (16 words) + run-start masks (16 x u64) */

struct HitRec {
  uint32_t bucket;   /* dense bucket id */
  uint32_t alpha32;  /* k_pairs: index j of the paired point; after k_group: (float)alpha_s bits for the fp32 vote path */
  double alpha_s;    /* exact alpha_s (k_group) */
};

/* fp32 acos for BIN SELECTION only: acos(|x|) = sqrt(1-|x|) * P(|x|), degree-7 least-squares/minimax fit,
 * measured max error 3.4e-7 rad including fp32 evaluation (tests/test_gpu_fastkeys.py re-checks the keys
 * against the exact path).  `t` = 1-|x| is formed in fp64 so the estimate stays relative-accurate near |x| = 1. */
__device__ __forceinline__ float acos32_estimate(double x) {
  const double ax = ppf_fabs(x);
  const float xf = (float)ax;
  const float t = fmaxf((float)(1.0 - ax), 0.0f);
  float p = -0.001441536471247673f;
  p = __builtin_fmaf(p, xf, 0.007245631422847509f);
  p = __builtin_fmaf(p, xf, -0.01780921407043934f);
  p = __builtin_fmaf(p, xf, 0.03133561089634895f);
  p = __builtin_fmaf(p, xf, -0.0503128282725811f);
  p = __builtin_fmaf(p, xf, 0.08899927139282227f);
  p = __builtin_fmaf(p, xf, -0.21459989249706268f);
  p = __builtin_fmaf(p, xf, 1.5707963705062866f);
  const float a = __builtin_sqrtf(t) * p;
  return x >= 0.0 ? a : 3.14159274101257324f - a;
}

/* Quantised key of a scene pair, exactly the integers of ppf_hash_feature(ppf_pair_feature(...)).
 * Fast path (every lane, branch-free): angles binned from the fp32 estimate, distance binned with a
 * reciprocal multiply; a lane whose value lies within a guard band of a bin edge (or is degenerate /
 * out of acos range) recomputes the fp64 chain.  Guards: angle 2e-5 bins-units-equivalent >> the 1.6e-6
 * estimate error; distance 1e-9 >> 1e-13. */
struct FastKeyConsts {
  float rstep32;   /* 1 / angle_step */
  float gq;        /* angle guard in bin units */
  double rdstep;   /* 1 / dist_step */
};

/* the four quantised features of a scene pair (the 16-byte key the reference hashes) */
__device__ __forceinline__ void pair_key(const ppf_vec3& p1, const ppf_vec3& n1, const ppf_vec3& p2, const ppf_vec3& n2,
                                         const double angle_step, const double dist_step, const FastKeyConsts& fk, int32_t (&k)[4]) {
  const double dx = p2.x - p1.x, dy = p2.y - p1.y, dz = p2.z - p1.z;
  const double f3 = ppf_sqrt(dx * dx + dy * dy + dz * dz);
  double rinv = __builtin_amdgcn_rcp(f3);
  rinv = rinv * (2.0 - f3 * rinv); /* one Newton step: ~1e-16 relative, far inside the guard */
  const double x0 = (n1.x * dx + n1.y * dy + n1.z * dz) * rinv;
  const double x1 = (n2.x * dx + n2.y * dy + n2.z * dz) * rinv;
  const double x2 = ppf_dot3(n1, n2); /* same expression as the exact path: bit-identical */
  bool slow = !(f3 > PPF_EPS) || !(ppf_fabs(x0) <= 1.0 - 1e-12) || !(ppf_fabs(x1) <= 1.0 - 1e-12) || !(ppf_fabs(x2) <= 1.0);
  {
    const float q0 = acos32_estimate(x0) * fk.rstep32, q1 = acos32_estimate(x1) * fk.rstep32,
                q2 = acos32_estimate(x2) * fk.rstep32;
    k[0] = (int)q0; k[1] = (int)q1; k[2] = (int)q2;
    const float lim = 0.5f - fk.gq;
    slow |= (__builtin_fabsf(__builtin_amdgcn_fractf(q0) - 0.5f) > lim) |
            (__builtin_fabsf(__builtin_amdgcn_fractf(q1) - 0.5f) > lim) |
            (__builtin_fabsf(__builtin_amdgcn_fractf(q2) - 0.5f) > lim);
    const double q3 = f3 * fk.rdstep;
    k[3] = (int)q3;
    const double fr3 = q3 - (double)k[3];
    slow |= !(fr3 > 1e-9 && fr3 < 1.0 - 1e-9) || !(q3 < 2.0e9);
  }
  if (slow) {
    double f[4] = {0, 0, 0, 0};
    ppf_pair_feature(p1, n1, p2, n2, f);
    k[0] = ppf_d2i(f[0] / angle_step); k[1] = ppf_d2i(f[1] / angle_step); k[2] = ppf_d2i(f[2] / angle_step);
    k[3] = ppf_d2i(f[3] / dist_step);
  }
}
__device__ __forceinline__ uint32_t pair_slot_hash(const ppf_vec3& p1, const ppf_vec3& n1, const ppf_vec3& p2,
                                                   const ppf_vec3& n2, const double angle_step, const double dist_step,
                                                   const FastKeyConsts& fk) {
  int32_t k[4];
  pair_key(p1, n1, p2, n2, angle_step, dist_step, fk, k);
  return ppf_murmur_key16(k[0], k[1], k[2], k[3]);
}

struct MatchArgs {
  CloudSoA surf;   /* reference points come from here */
  CloudSoA paired; /* second points of the pairs (== surf for match, the edge cloud for match_S2B) */
  int same_cloud;
  int scene_step, ref_offset, ref_stride; /* reference point r (global) -> row (ref_offset + r*ref_stride)*scene_step */
  int ref_base, n_ref;                    /* this batch: global r = ref_base + local r */
  /* model table */
  const SlotWord* slotmap;
  uint32_t slot_mask;
  /* key -> dense bucket id (-1: empty slot), indexed ((k0*lut_na + k1)*lut_na + k2)*lut_nd + k3: the hash of every
   * quantised key a scene can produce, tabulated once per model; keys outside the table take the hash path */
  const int32_t* key_lut;
  int lut_na, lut_nd;
  const uint32_t* bucket_off;
  int n_buckets;
  const uint4* records;    /* pair records {row_a, row_b, alpha_a, alpha_b}; bucket_off counts records */
  const uint4* records_fx; /* the same records with alpha_m as signed 16.16 fixed point of alpha_m*A/(4pi): what k_vote adds */
  int n_tiles, tile_refs, num_angles, n_model;
  double angle_step, dist_step;
  /* per-batch scratch */
  double* frames;          /* [n_ref][12] */
  HitRec* hits;            /* [n_ref][hit_cap] */
  uint32_t* hit_count;     /* [n_ref] */
  uint2* keys_a;           /* [n_ref][hit_cap] {bucket, hit index}: sorted by bucket after k_group */
  uint2* keys_b;           /* [n_ref][hit_cap] ping-pong */
  const uint2* keys_sorted; /* where k_group leaves the grouped keys: keys_b on the LDS path, keys_a on the radix path */
  int hit_cap;
  int key_bits;            /* bits of a bucket id */
  int group_lds_buckets;   /* n_buckets when one LDS counter per bucket fits (single-pass grouping), else 0 */
  const uint32_t* bucket_total; /* [n_buckets] entries of a bucket over all tiles */
  unsigned long long* work;     /* [n_ref] votes the reference point will cast (sum of its hits' bucket sizes) */
  uint32_t* perm;               /* [n_ref] reference points ordered by work, heaviest first (k_rank) */
  const uint32_t* perm_group;   /* [n_ref] reference points ordered by hit count, for k_group (may be NULL) */
  /* results, indexed by global r */
  uint2* partial;               /* [n_ref_all * n_tiles] {max votes, local flat index} */
  unsigned long long* cellsum;  /* [n_ref_all * n_tiles] sum of the tile's accumulator == votes cast */
  unsigned long long* pairs;    /* [n_ref_all] pairs hashed */
  uint32_t* acc_dump;           /* optional [n_ref_all][n_model*num_angles] full accumulators (debug/tests) */
  int ablate;                   /* PPF_ABLATE env (diagnostic builds only): 1 conflict-free atomics, 2 no atomics, 3 no entry loads */
};

__device__ __forceinline__ int ref_row(const MatchArgs& a, int r_local) {
  return (a.ref_offset + (a.ref_base + r_local) * a.ref_stride) * a.scene_step;
}

__global__ __launch_bounds__(64) void k_frames(MatchArgs a) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.n_ref) return;
  const int i = ref_row(a, r);
  double R[9], t[3];
  ppf_transform_rt(ld3(a.surf.x, a.surf.y, a.surf.z, i), ld3(a.surf.nx, a.surf.ny, a.surf.nz, i), R, t);
  double* f = a.frames + (size_t)r * 12;
#pragma unroll
  for (int k = 0; k < 9; k++) f[k] = R[k];
#pragma unroll
  for (int k = 0; k < 3; k++) f[9 + k] = t[k];
}

/* grid: x = chunks of PAIR_BLOCK*PAIRS_PER_THREAD paired points, y = reference point of the batch */
__global__ __launch_bounds__(PAIR_BLOCK) void k_pairs(MatchArgs a) {
  __shared__ uint2 stash[PAIRS_PER_THREAD][PAIR_BLOCK]; /* {bucket, j} of this thread's hits, one slot per iteration */
  const int r = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int i_ref = ref_row(a, r);
  const ppf_vec3 p1 = ld3(a.surf.x, a.surf.y, a.surf.z, i_ref), n1 = ld3(a.surf.nx, a.surf.ny, a.surf.nz, i_ref);
  FastKeyConsts fk;
  fk.rstep32 = (float)(1.0 / a.angle_step);
  fk.gq = 4.0e-6f * fk.rstep32; /* 4e-6 rad: > 10x the 3.4e-7 rad estimate error; 1.9e-5 bins at 12 degrees */
  fk.rdstep = 1.0 / a.dist_step;
  unsigned long long my_pairs = 0;
  uint32_t hit_mask = 0;
  const int j0 = blockIdx.x * (PAIR_BLOCK * PAIRS_PER_THREAD) + tid;
  const int n = a.paired.n;
  /* the point of the next iteration is fetched while the current pair is hashed */
  float nx0 = 0, nx1 = 0, nx2 = 0, nx3 = 0, nx4 = 0, nx5 = 0;
  {
    const int jc = min(j0, n - 1);
    nx0 = a.paired.x[jc]; nx1 = a.paired.y[jc]; nx2 = a.paired.z[jc];
    nx3 = a.paired.nx[jc]; nx4 = a.paired.ny[jc]; nx5 = a.paired.nz[jc];
  }
#pragma unroll 1
  for (int it = 0; it < PAIRS_PER_THREAD; it++) {
    const int j = j0 + it * PAIR_BLOCK;
    const ppf_vec3 p2 = ppf_mk3((double)nx0, (double)nx1, (double)nx2), n2 = ppf_mk3((double)nx3, (double)nx4, (double)nx5);
    {
      const int jn = min(j + PAIR_BLOCK, n - 1);
      nx0 = a.paired.x[jn]; nx1 = a.paired.y[jn]; nx2 = a.paired.z[jn];
      nx3 = a.paired.nx[jn]; nx4 = a.paired.ny[jn]; nx5 = a.paired.nz[jn];
    }
    if (j < n && !(a.same_cloud && j == i_ref)) {
      /* match_S2B: the reference point itself is never paired, even when the edge cloud contains it
       * (bit-identical row), so edge == scene reduces exactly to match().  Values came from floats, so
       * comparing the doubles compares the float bits (no NaN/-0 cases in finite clouds). */
      const bool self_pair = !a.same_cloud && p2.x == p1.x && p2.y == p1.y && p2.z == p1.z && n2.x == n1.x &&
                             n2.y == n1.y && n2.z == n1.z;
      if (!self_pair) {
        int32_t key[4];
        pair_key(p1, n1, p2, n2, a.angle_step, a.dist_step, fk, key);
        int b;
        if (((uint32_t)key[0] < (uint32_t)a.lut_na) & ((uint32_t)key[1] < (uint32_t)a.lut_na) & ((uint32_t)key[2] < (uint32_t)a.lut_na) &
            ((uint32_t)key[3] < (uint32_t)a.lut_nd)) {
          b = a.key_lut[(size_t)((key[0] * a.lut_na + key[1]) * a.lut_na + key[2]) * a.lut_nd + key[3]];
        } else { /* NaN features (INT_MIN bins) or pairs farther apart than the table covers */
          b = slot_to_bucket(a.slotmap, ppf_murmur_key16(key[0], key[1], key[2], key[3]) & a.slot_mask);
        }
        /* The reference skips a pair whose alpha_s is NaN; for finite clouds it never is.  alpha_s itself is
         * computed later (k_group), only for the ~6 % of pairs that found a bucket. */
        my_pairs += 1u;
        if (b >= 0) {
          stash[it][tid] = make_uint2((uint32_t)b, (uint32_t)j);
          hit_mask |= 1u << it;
        }
      }
    }
  }
  /* one returned atomic per wave: wave-wide exclusive scan of the per-lane hit counts */
  const uint32_t mine = (uint32_t)__popc(hit_mask);
  uint32_t incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(incl, o);
    if (lane >= o) incl += y;
  }
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  if (total) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&a.hit_count[r], total);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    uint32_t pos = base + incl - mine;
    HitRec* __restrict__ hits = a.hits + (size_t)r * a.hit_cap;
    uint2* __restrict__ keys = a.keys_a + (size_t)r * a.hit_cap;
#pragma unroll 1
    for (int it = 0; it < PAIRS_PER_THREAD; it++) {
      if (hit_mask & (1u << it)) {
        const uint2 h = stash[it][tid];
        HitRec rec;
        rec.bucket = h.x; rec.alpha32 = h.y; rec.alpha_s = 0.0;
        hits[pos] = rec;
        keys[pos] = make_uint2(h.x, pos);
        pos++;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) my_pairs += __shfl_down(my_pairs, o);
  if (lane == 0 && my_pairs) atomicAdd(&a.pairs[a.ref_base + r], my_pairs);
}

/*
 * k_group: stable LSD radix sort of one reference point's {bucket, hit index} keys by bucket id.
 * Elements are taken 1024 at a time in list order; inside a tile, wave w owns elements 64w..64w+63,
 * so (wave, lane) order is list order.  Rank of an element among equal digits = digits before it in
 * earlier tiles (running base) + in earlier waves of the tile (wave counts) + in lower lanes of its
 * wave (ballot match).  The result always ends in keys_a.
 */
__global__ __launch_bounds__(GROUP_BLOCK) void k_group(MatchArgs a) {
  __shared__ uint32_t base[256];
  __shared__ unsigned long long wsum[GROUP_BLOCK / 64];
  extern __shared__ uint32_t gcnt[]; /* LDS path: one counter per bucket (+ wave totals); radix path: wcnt[16][256] */
  const int r = a.perm_group ? (int)a.perm_group[blockIdx.x] : (int)blockIdx.x; /* most hits first: no long block at the tail */
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t n = a.hit_count[r];
  const bool lds_path = a.group_lds_buckets > 0;
  const int nb1 = a.group_lds_buckets + 1;
  if (lds_path) {
    for (int k = tid; k < nb1; k += GROUP_BLOCK) gcnt[k] = 0;
    __syncthreads();
  }
  {
    /* ONE pass over the hits: alpha_s of every hit (dense: only pairs that found a bucket) = angle of
     * (tsg + Rsg p2) about x; the exact number of votes this reference point will cast (used to launch the heaviest
     * first); and, on the LDS path, the per-bucket histogram.  A NaN alpha (non-finite cloud) makes the reference
     * skip the pair: the hit is retired by emptying its key. */
    const double* __restrict__ fr = a.frames + (size_t)r * 12;
    const double R10 = fr[3], R11 = fr[4], R12 = fr[5], R20 = fr[6], R21 = fr[7], R22 = fr[8], ty = fr[10], tz = fr[11];
    HitRec* __restrict__ hits = a.hits + (size_t)r * a.hit_cap;
    uint2* kk = a.keys_a + (size_t)r * a.hit_cap;
    unsigned long long w = 0;
    for (uint32_t i = tid; i < n; i += GROUP_BLOCK) {
      const int j = (int)hits[i].alpha32;
      uint32_t key = kk[i].x;
      const uint32_t total = a.bucket_total[min(key, (uint32_t)(a.n_buckets - 1))]; /* independent gather, issued with the others */
      const ppf_vec3 p2 = ld3(a.paired.x, a.paired.y, a.paired.z, j);
      const double qy = ty + (R10 * p2.x + R11 * p2.y + R12 * p2.z);
      const double qz = tz + (R20 * p2.x + R21 * p2.y + R22 * p2.z);
      double as = 0.0;
      if (ppf_alpha_in_frame(qy, qz, &as)) {
        hits[i].alpha32 = __float_as_uint((float)as);
        hits[i].alpha_s = as;
        w += total;
      } else {
        hits[i].alpha32 = 0; hits[i].alpha_s = 0.0;
        key = 0xFFFFFFFFu;
        kk[i].x = key; /* sorts last; k_vote gives it no entries */
      }
      if (lds_path) atomicAdd(&gcnt[min(key, (uint32_t)(nb1 - 1))], 1u);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o);
    if (lane == 0) wsum[wave] = w;
    __syncthreads();
    if (tid == 0) {
      unsigned long long t = 0;
      for (int k = 0; k < GROUP_BLOCK / 64; k++) t += wsum[k];
      a.work[r] = t;
    }
  }
  uint2* src = a.keys_a + (size_t)r * a.hit_cap;
  uint2* dst = a.keys_b + (size_t)r * a.hit_cap;
  if (n < 2 && !lds_path) return;
  if (lds_path) {
    /* Grouping only needs equal buckets to be adjacent (any order inside a bucket: votes commute), so when one
     * counter per bucket fits in LDS the histogram of the pass above, a scan and one scatter through cursors do it.
     * Retired hits (key 0xFFFFFFFF) go to the extra last counter.  The grouped keys stay in keys_b (k_vote reads
     * a.keys_sorted). */
    uint32_t* wtot = gcnt + nb1;
    /* exclusive scan of nb1 counters: each thread owns a contiguous slice */
    const int per = (nb1 + GROUP_BLOCK - 1) / GROUP_BLOCK;
    const int k0 = tid * per, k1 = min(k0 + per, nb1);
    uint32_t tsum = 0;
    for (int k = k0; k < k1; k++) tsum += gcnt[k];
    uint32_t incl = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(incl, o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    uint32_t run = incl - tsum;
    for (int w = 0; w < wave; w++) run += wtot[w];
    for (int k = k0; k < k1; k++) { const uint32_t c = gcnt[k]; gcnt[k] = run; run += c; }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += GROUP_BLOCK) {
      const uint2 key = src[i];
      dst[atomicAdd(&gcnt[min(key.x, (uint32_t)(nb1 - 1))], 1u)] = key;
    }
    return;
  }
  uint32_t (*wcnt)[256] = reinterpret_cast<uint32_t (*)[256]>(gcnt);
  const int passes = (a.key_bits + 7) / 8;
  for (int pass = 0; pass < passes; pass++) {
    const int shift = pass * 8;
    if (tid < 256) base[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += GROUP_BLOCK) atomicAdd(&base[(src[i].x >> shift) & 255u], 1u);
    __syncthreads();
    if (wave == 0) { /* exclusive scan of the 256 digit counts: 4 per lane */
      uint32_t c[4], tsum = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) { c[k] = base[lane * 4 + k]; tsum += c[k]; }
      uint32_t incl = tsum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
      }
      uint32_t ex = incl - tsum;
#pragma unroll
      for (int k = 0; k < 4; k++) { base[lane * 4 + k] = ex; ex += c[k]; }
    }
    __syncthreads();
    for (uint32_t t0 = 0; t0 < n; t0 += GROUP_BLOCK) {
      for (int k = tid; k < (GROUP_BLOCK / 64) * 256; k += GROUP_BLOCK) (&wcnt[0][0])[k] = 0;
      __syncthreads();
      const uint32_t i = t0 + tid;
      const bool valid = i < n;
      uint2 key = make_uint2(0, 0);
      uint32_t d = 0;
      if (valid) { key = src[i]; d = (key.x >> shift) & 255u; }
      /* lanes of this wave with the same digit */
      unsigned long long same = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < 8; bit++) {
        const unsigned long long bb = __ballot((d >> bit) & 1u);
        same &= ((d >> bit) & 1u) ? bb : ~bb;
      }
      const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
      if (valid && rank == 0) wcnt[wave][d] = (uint32_t)__popcll(same);
      __syncthreads();
      if (valid) {
        uint32_t pos = base[d] + rank;
        for (int w = 0; w < wave; w++) pos += wcnt[w][d];
        dst[pos] = key;
      }
      __syncthreads();
      if (tid < 256) {
        uint32_t add = 0;
#pragma unroll
        for (int w = 0; w < GROUP_BLOCK / 64; w++) add += wcnt[w][tid];
        base[tid] += add;
      }
      __syncthreads();
    }
    uint2* tmp = src; src = dst; dst = tmp;
  }
  if (passes & 1) { /* result sits in keys_b: copy back */
    uint2* ka = a.keys_a + (size_t)r * a.hit_cap;
    const uint2* kb = a.keys_b + (size_t)r * a.hit_cap;
    for (uint32_t i = tid; i < n; i += GROUP_BLOCK) ka[i] = kb[i];
  }
}

/*
 * Accumulator layout in LDS (words):  [guard: VOTE_GUARD(P)] [tile_refs x P cells]
 *   P = (A+1)|1 is the row pitch: A alpha bins + the "bin == A" spill cell of the row (the reference
 *   indexes corrI*A + alpha_index without a range check, so alpha_index == A lands on the next model
 *   reference point's bin 0; the spill cell is folded into that bin when the accumulator is scanned).
 *   An odd pitch also spreads rows over all 32 LDS banks.
 *   The guard words below cell 0 take every vote that must not count: mirrored spill entries
 *   (word offset GW-A) with any bin other than A, and the lanes past the end of a bucket (word = lane).
 *   With it the vote needs no range check at all.  Entry offsets are bytes from the guard's start.
 *
 * Two votes (one pair record) = 2 x v_fma_f32 (q'), 2 x v_cvt_i32_f32 (k), 2 x v_fract_f32 + a shared v_min3
 * reduction (guard band), 2 x v_lshl_add_u32 (byte address), 2 x ds_add_u32.
 *   q' = alpha_m*S + Ohg,  Ohg = A/2 - alpha_s*S + G  (folded once per hit), k = trunc(q')
 *   k is the reference's integer whenever fract(q') >= 2G; otherwise the lane
 *   re-evaluates the fp64 chain.  That happens for ~3e-5 of the votes, so the re-evaluation is
 *   taken once per batch of U entries and only when some lane of the wave needs it.
 */

template <int U>
__device__ __forceinline__ void load_records(uint4* rec, const uint4* __restrict__ src, const uint32_t e0, const int lane) {
#pragma unroll
  for (int u = 0; u < U; u++) {
#if PPF_ABL == 3 || PPF_ABL == 5 /* diagnostic: no record loads */
    const uint32_t x = e0 + u * 64 + lane;
    rec[u] = make_uint4(((x * 2u) & 1023u) * 124u + 504u, ((x * 2u + 1u) & 1023u) * 124u + 504u,
                        0x3a000000u + (x & 0xffffu) * 64u, 0x3a000000u + (x & 0xffffu) * 64u + 32u);
#else
    rec[u] = src[e0 + u * 64 + lane];
#endif
  }
}

/* U pair records per lane -> 2U votes per lane.  n_valid = number of 64-record groups (of the U) holding data. */
template <int U>
__device__ __forceinline__ void cast_votes(unsigned char* __restrict__ acc_bytes, const uint4* rec, const int n_valid,
                                           const float S, const float Ohg, const double* __restrict__ asd_lds,
                                           const float G2, const int A) {
  int ka[U], kb[U];
  float fa[U], fb[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    /* two scalar v_fma_f32: measured faster than one v_pk_fma_f32 on gfx950 (compute-only k_vote 10.0 vs 13.0 ms) */
    const float qa = __builtin_fmaf(__uint_as_float(rec[u].z), S, Ohg);
    const float qb = __builtin_fmaf(__uint_as_float(rec[u].w), S, Ohg);
    ka[u] = (int)qa;
    kb[u] = (int)qb;
    fa[u] = __builtin_amdgcn_fractf(qa);
    fb[u] = __builtin_amdgcn_fractf(qb);
  }
  /* smallest fractional part of the 2U votes with as few v_min3_f32 as possible (3 inputs each) */
  float frmin = fa[0];
  {
    float pend[2 * U];
    int np = 0;
#pragma unroll
    for (int u = 0; u < U; u++) { if (u) pend[np++] = fa[u]; pend[np++] = fb[u]; }
    int i = 0;
#pragma unroll
    for (; i + 2 <= np; i += 2) frmin = __builtin_fminf(__builtin_fminf(frmin, pend[i]), pend[i + 1]);
    if (i < np) frmin = __builtin_fminf(frmin, pend[i]);
  }
  if (__builtin_expect(__any(frmin < G2), 0)) {
    const double asd = *asd_lds; /* exact alpha_s of this hit, only needed here */
#pragma unroll
    for (int u = 0; u < U; u++) {
      uint32_t za = rec[u].z, zb = rec[u].w;
      asm volatile("" : "+v"(za), "+v"(zb)); /* keep the fp64 conversions of the rare path out of the hot loop */
      const float aa = __uint_as_float(za), ab = __uint_as_float(zb);
      if (__builtin_amdgcn_fractf(__builtin_fmaf(aa, S, Ohg)) < G2) ka[u] = ppf_alpha_bin_exact(aa, asd, A);
      if (__builtin_amdgcn_fractf(__builtin_fmaf(ab, S, Ohg)) < G2) kb[u] = ppf_alpha_bin_exact(ab, asd, A);
    }
  }
#pragma unroll
  for (int u = 0; u < U; u++) {
    if (u < n_valid) {
      int adr_a = (int)rec[u].x + ka[u] * 4, adr_b = (int)rec[u].y + kb[u] * 4;
#if PPF_ABL == 1 /* diagnostic: same instruction stream, conflict-free addresses */
      asm volatile("" ::"v"(adr_a), "v"(adr_b));
      adr_a = (int)((threadIdx.x & 63) * 4 + u * 512 + 1024);
      adr_b = adr_a + 256;
#endif
#if PPF_ABL == 2 || PPF_ABL == 5 /* diagnostic: no atomics */
      asm volatile("" ::"v"(adr_a), "v"(adr_b));
#else
      atomicAdd(reinterpret_cast<uint32_t*>(acc_bytes + adr_a), 1u);
      atomicAdd(reinterpret_cast<uint32_t*>(acc_bytes + adr_b), 1u);
#endif
    }
  }
}

/* ---- the same votes with the LDS atomics of hit h issued in the shadow of hit h+1's arithmetic ------------------
 * cast_votes emits [39 VALU][8 ds_add] per hit; with 4 waves per SIMD the waves bunch up at their LDS phases and the
 * two pipes alternate instead of overlapping.  Here the addresses of a hit are kept in registers and its atomics are
 * interleaved (sched_group_barrier: 5 VALU, 1 DS, ...) with the bin arithmetic of the next hit of the item. */
typedef __attribute__((address_space(3))) unsigned char lds_byte; /* explicit LDS pointers: 32-bit arithmetic, ds_* atomics */
typedef __attribute__((address_space(3))) uint32_t lds_u32;
template <int U>
__device__ __forceinline__ void vote_bins(const uint4* rec, const float S, const float Ohg, const double* __restrict__ asd_lds,
                                          const float G2, const int A, int (&ka)[U], int (&kb)[U], float& frmin_out) {
  float fa[U], fb[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const float qa = __builtin_fmaf(__uint_as_float(rec[u].z), S, Ohg);
    const float qb = __builtin_fmaf(__uint_as_float(rec[u].w), S, Ohg);
    ka[u] = (int)qa;
    kb[u] = (int)qb;
    fa[u] = __builtin_amdgcn_fractf(qa);
    fb[u] = __builtin_amdgcn_fractf(qb);
  }
  float frmin = fa[0];
  float pend[2 * U];
  int np = 0;
#pragma unroll
  for (int u = 0; u < U; u++) { if (u) pend[np++] = fa[u]; pend[np++] = fb[u]; }
  int i = 0;
#pragma unroll
  for (; i + 2 <= np; i += 2) frmin = __builtin_fminf(__builtin_fminf(frmin, pend[i]), pend[i + 1]);
  if (i < np) frmin = __builtin_fminf(frmin, pend[i]);
  frmin_out = frmin;
  (void)asd_lds; (void)G2; (void)A;
}
/* rare path: votes within the guard band of a bin edge get the exact fp64 bin */
template <int U>
__device__ __forceinline__ void vote_fix(const uint4* rec, const float S, const float Ohg, const double* __restrict__ asd_lds,
                                         const float G2, const int A, const float frmin, int (&ka)[U], int (&kb)[U]) {
  if (__builtin_expect(__any(frmin < G2), 0)) {
    const double asd = *asd_lds; /* exact alpha_s of this hit, only needed here */
#pragma unroll
    for (int u = 0; u < U; u++) {
      uint32_t za = rec[u].z, zb = rec[u].w;
      asm volatile("" : "+v"(za), "+v"(zb)); /* keep the fp64 conversions of the rare path out of the hot loop */
      const float aa = __uint_as_float(za), ab = __uint_as_float(zb);
      if (__builtin_amdgcn_fractf(__builtin_fmaf(aa, S, Ohg)) < G2) ka[u] = ppf_alpha_bin_exact(aa, asd, A);
      if (__builtin_amdgcn_fractf(__builtin_fmaf(ab, S, Ohg)) < G2) kb[u] = ppf_alpha_bin_exact(ab, asd, A);
    }
  }
}
/* the 2U atomics of one hit: LDS address = row's bin 0 + 4k (one v_lshl_add_u32), ds_add_u32 */
template <int U>
__device__ __forceinline__ void vote_issue(const uint32_t (&pa)[U], const uint32_t (&pb)[U], const int (&ka)[U], const int (&kb)[U],
                                           const int n_valid) {
#pragma unroll
  for (int u = 0; u < U; u++) {
    if (u < n_valid) {
      (void)__hip_atomic_fetch_add((lds_u32*)(uintptr_t)(pa[u] + ((uint32_t)ka[u] << 2)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      (void)__hip_atomic_fetch_add((lds_u32*)(uintptr_t)(pb[u] + ((uint32_t)kb[u] << 2)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}
/* one pipeline stage: the atomics of the previous hit (bins pka/pkb) under the bin arithmetic of hit hh (-> nka/nkb) */
template <int U>
__device__ __forceinline__ void vote_stage(const uint4* rec, const int n_valid, const float S, const float ohg_v, const int hh,
                                           const double* __restrict__ asd_lds, const float G2, const int A, const uint32_t (&pa)[U],
                                           const uint32_t (&pb)[U], const int (&pka)[U], const int (&pkb)[U], int (&nka)[U], int (&nkb)[U]) {
  float frmin;
  const float Ohg = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ohg_v), hh));
  vote_issue<U>(pa, pb, pka, pkb, n_valid);
  vote_bins<U>(rec, S, Ohg, asd_lds + hh, G2, A, nka, nkb, frmin);
#pragma unroll
  for (int i = 0; i < 2 * U; i++) {
    __builtin_amdgcn_sched_group_barrier(0x002, PPF_PIPE_VALU, 0); /* VALU */
    __builtin_amdgcn_sched_group_barrier(0x080, 1, 0); /* DS */
  }
  vote_fix<U>(rec, S, Ohg, asd_lds + hh, G2, A, frmin, nka, nkb);
}
/* all hits of a work item against one register batch of records; two sets of bins alternate so that a set is only
 * overwritten a full stage after the atomics that used it were issued */
template <int U>
__device__ __forceinline__ void vote_hits(unsigned char* __restrict__ acc_bytes, const uint4* rec, const int n_valid, const float S,
                                          const float ohg_v, const int nh, const double* __restrict__ asd_lds, const float G2,
                                          const int A) {
  uint32_t pa[U], pb[U];
  int ka0[U], kb0[U], ka1[U], kb1[U];
  const uint32_t base = (uint32_t)(uintptr_t)(lds_byte*)acc_bytes; /* LDS byte address of the guard region */
#pragma unroll
  for (int u = 0; u < U; u++) {
    pa[u] = base + rec[u].x;
    pb[u] = base + rec[u].y;
    asm volatile("" : "+v"(pa[u]), "+v"(pb[u])); /* computed once per batch, not rematerialised per vote */
  }
  {
    float frmin;
    const float Ohg = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ohg_v), 0));
    vote_bins<U>(rec, S, Ohg, asd_lds, G2, A, ka0, kb0, frmin);
    vote_fix<U>(rec, S, Ohg, asd_lds, G2, A, frmin, ka0, kb0);
  }
  int hh = 1;
  for (; hh + 1 < nh; hh += 2) {
    vote_stage<U>(rec, n_valid, S, ohg_v, hh, asd_lds, G2, A, pa, pb, ka0, kb0, ka1, kb1);
    vote_stage<U>(rec, n_valid, S, ohg_v, hh + 1, asd_lds, G2, A, pa, pb, ka1, kb1, ka0, kb0);
  }
  if (hh < nh) {
    vote_stage<U>(rec, n_valid, S, ohg_v, hh, asd_lds, G2, A, pa, pb, ka0, kb0, ka1, kb1);
    vote_issue<U>(pa, pb, ka1, kb1, n_valid);
  } else {
    vote_issue<U>(pa, pb, ka0, kb0, n_valid);
  }
}

/* Buckets of at most 32 pair records (64 entries: more than half of all (tile, bucket) runs): one ENTRY per lane
 * instead of one pair record per lane, so the 64 lanes of the single group are filled twice as well and a hit costs
 * one fma/cvt/fract/lshl_add/ds_add instead of two of each.  Same bins, same guard band, same exact fallback. */
__device__ __forceinline__ void vote_hits_single(unsigned char* __restrict__ acc_bytes, const uint32_t row_bytes, const uint32_t alpha_bits,
                                                 const float S, const float ohg_v, const int nh, const double* __restrict__ asd_lds,
                                                 const float G2, const int A) {
  uint32_t pr = (uint32_t)(uintptr_t)(lds_byte*)acc_bytes + row_bytes;
  asm volatile("" : "+v"(pr));
  const float am = __uint_as_float(alpha_bits);
  uint32_t adr_prev = 0;
  for (int hh = 0; hh < nh; hh++) {
    const float Ohg = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ohg_v), hh));
    if (hh) (void)__hip_atomic_fetch_add((lds_u32*)(uintptr_t)adr_prev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const float q = __builtin_fmaf(am, S, Ohg);
    int k = (int)q;
    if (__builtin_expect(__any(__builtin_amdgcn_fractf(q) < G2), 0)) {
      const double asd = asd_lds[hh];
      uint32_t z = alpha_bits;
      asm volatile("" : "+v"(z));
      const float az = __uint_as_float(z);
      if (__builtin_amdgcn_fractf(__builtin_fmaf(az, S, Ohg)) < G2) k = ppf_alpha_bin_exact(az, asd, A);
    }
    adr_prev = pr + ((uint32_t)k << 2);
  }
  if (nh > 0) (void)__hip_atomic_fetch_add((lds_u32*)(uintptr_t)adr_prev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

/* ---- 16.16 fixed-point votes --------------------------------------------------------------------------------
 * bin = floor(x), x = (alpha_m - alpha_s)*A/(4pi) + A/2.  Per entry fx = rint(alpha_m * A/(4pi) * 65536) (signed, built
 * with the table), per hit C = rint((A/2 - alpha_s*A/(4pi)) * 65536) + 2 (fp64, once per staged hit): the integer sum
 * s = fx + C equals x*65536 + 2 + e with |e| <= 1.01 (two roundings; the fp64 products are exact to 1e-9 units), so
 * whenever the 16 fraction bits of s are >= 4, floor(x) = s >> 16 -- and it is also what the reference's fp64 chain
 * gives, which differs from x by ~1e-14.  Otherwise (6e-5 of the votes) the lane evaluates that fp64 chain.
 * Cost per vote: v_add_u32 (s), v_mad_u32_u16 (LDS address = hi16(s)*4 + row: the shift, the mask and the add in
 * one instruction), half a v_min3_u16 (the guard: running minimum of the low halves), ds_add_u32. */
#ifndef PPF_FX_ASM
#define PPF_FX_ASM 1 /* v_mad_u32_u16 / v_min3_u16 through inline asm (the compiler does not select them from C++) */
#endif
__device__ __forceinline__ uint32_t fx_min3(uint32_t m, uint32_t a, uint32_t b) {
#if PPF_FX_ASM
  uint32_t r;
  asm("v_min3_u16 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
  return r;
#else /* 16-bit minimum of the low halves: selected as v_min3_u16 */
  const unsigned short x = (unsigned short)m, y = (unsigned short)a, z = (unsigned short)b;
  const unsigned short t = x < y ? x : y;
  return (uint32_t)(t < z ? t : z);
#endif
}
__device__ __forceinline__ uint32_t fx_addr(uint32_t s, uint32_t row) { /* hi16(s) * 4 + row */
#if PPF_FX_ASM
  uint32_t r;
  asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[1,0,0,0]" : "=v"(r) : "v"(s), "v"(row));
  return r;
#else
  return (uint32_t)(unsigned short)(s >> 16) * (uint32_t)(unsigned short)4 + row;
#endif
}
template <int U>
__device__ __forceinline__ void fx_sums(const uint4* rec, const uint32_t C, uint32_t (&sa)[U], uint32_t (&sb)[U], uint32_t& guard) {
  uint32_t m = 0xFFFFu;
#pragma unroll
  for (int u = 0; u < U; u++) {
    sa[u] = rec[u].z + C;
    sb[u] = rec[u].w + C;
    m = fx_min3(m, sa[u], sb[u]);
  }
  guard = m;
}
/* rare path: votes whose fraction is inside the guard get the exact fp64 bin (alpha_m comes from the float records) */
template <int U>
__device__ __forceinline__ void fx_fix(const uint4* __restrict__ srcf, const uint32_t e0, const uint32_t c, const int lane,
                                       const double* __restrict__ asd_lds, const int A, const uint32_t T, const uint32_t guard,
                                       uint32_t (&sa)[U], uint32_t (&sb)[U]) {
  if (__builtin_expect(__any((guard & 0xFFFFu) < T), 0)) {
    const double asd = *asd_lds; /* exact alpha_s of this hit */
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (__any(((sa[u] & 0xFFFFu) < T) | ((sb[u] & 0xFFFFu) < T))) {
        const uint4 rf = srcf[min(e0 + (uint32_t)(u * 64 + lane), c - 1u)];
        if ((sa[u] & 0xFFFFu) < T) sa[u] = (uint32_t)ppf_alpha_bin_exact(__uint_as_float(rf.z), asd, A) << 16;
        if ((sb[u] & 0xFFFFu) < T) sb[u] = (uint32_t)ppf_alpha_bin_exact(__uint_as_float(rf.w), asd, A) << 16;
      }
    }
  }
}
template <int U>
__device__ __forceinline__ void fx_issue(const uint32_t (&pa)[U], const uint32_t (&pb)[U], const uint32_t (&sa)[U], const uint32_t (&sb)[U],
                                         const int n_valid) {
#pragma unroll
  for (int u = 0; u < U; u++) {
    if (u < n_valid) {
#ifdef PPF_FX_NOCONFLICT /* diagnostic: same instruction stream, conflict-free addresses */
      uint32_t aa = fx_addr(sa[u], pa[u]), ab = fx_addr(sb[u], pb[u]);
      asm volatile("" ::"v"(aa), "v"(ab));
      aa = (uint32_t)((threadIdx.x & 63) * 4 + u * 512 + 29184 + 1024);
      ab = aa + 256;
      (void)__hip_atomic_fetch_add((lds_u32*)(uintptr_t)aa, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      (void)__hip_atomic_fetch_add((lds_u32*)(uintptr_t)ab, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
      (void)__hip_atomic_fetch_add((lds_u32*)(uintptr_t)fx_addr(sa[u], pa[u]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      (void)__hip_atomic_fetch_add((lds_u32*)(uintptr_t)fx_addr(sb[u], pb[u]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    }
  }
}
template <int U>
__device__ __forceinline__ void fx_stage(const uint4* rec, const int n_valid, const uint32_t cfix_v, const int hh,
                                         const double* __restrict__ asd_lds, const int A, const uint32_t T, const uint4* __restrict__ srcf,
                                         const uint32_t e0, const uint32_t c, const int lane, const uint32_t (&pa)[U], const uint32_t (&pb)[U],
                                         const uint32_t (&psa)[U], const uint32_t (&psb)[U], uint32_t (&nsa)[U], uint32_t (&nsb)[U]) {
  const uint32_t C = (uint32_t)__builtin_amdgcn_readlane((int)cfix_v, hh);
  uint32_t guard;
  fx_issue<U>(pa, pb, psa, psb, n_valid);
  fx_sums<U>(rec, C, nsa, nsb, guard);
#pragma unroll
  for (int i = 0; i < 2 * U; i++) {
    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); /* VALU */
    __builtin_amdgcn_sched_group_barrier(0x080, 1, 0); /* DS */
  }
  fx_fix<U>(srcf, e0, c, lane, asd_lds + hh, A, T, guard, nsa, nsb);
}
/* all hits of a work item against one register batch of records (rec: fixed-point records e0.. of the run) */
template <int U>
__device__ __forceinline__ void vote_hits_fx(unsigned char* __restrict__ acc_bytes, const uint4* rec, const int n_valid, const uint32_t cfix_v,
                                             const int nh, const double* __restrict__ asd_lds, const int A, const uint32_t T,
                                             const uint4* __restrict__ srcf, const uint32_t e0, const uint32_t c, const int lane) {
  uint32_t pa[U], pb[U], sa0[U], sb0[U], sa1[U], sb1[U];
  const uint32_t base = (uint32_t)(uintptr_t)(lds_byte*)acc_bytes; /* LDS byte address of the guard region */
#pragma unroll
  for (int u = 0; u < U; u++) {
    pa[u] = base + rec[u].x;
    pb[u] = base + rec[u].y;
    asm volatile("" : "+v"(pa[u]), "+v"(pb[u])); /* computed once per batch, not rematerialised per vote */
  }
  {
    uint32_t guard;
    fx_sums<U>(rec, (uint32_t)__builtin_amdgcn_readlane((int)cfix_v, 0), sa0, sb0, guard);
    fx_fix<U>(srcf, e0, c, lane, asd_lds, A, T, guard, sa0, sb0);
  }
  int hh = 1;
  for (; hh + 1 < nh; hh += 2) {
    fx_stage<U>(rec, n_valid, cfix_v, hh, asd_lds, A, T, srcf, e0, c, lane, pa, pb, sa0, sb0, sa1, sb1);
    fx_stage<U>(rec, n_valid, cfix_v, hh + 1, asd_lds, A, T, srcf, e0, c, lane, pa, pb, sa1, sb1, sa0, sb0);
  }
  if (hh < nh) {
    fx_stage<U>(rec, n_valid, cfix_v, hh, asd_lds, A, T, srcf, e0, c, lane, pa, pb, sa0, sb0, sa1, sb1);
    fx_issue<U>(pa, pb, sa1, sb1, n_valid);
  } else {
    fx_issue<U>(pa, pb, sa0, sb0, n_valid);
  }
}
/* runs of at most 32 pair records: one entry per lane (see vote_hits_single) */
__device__ __forceinline__ void vote_hits_single_fx(unsigned char* __restrict__ acc_bytes, const uint32_t row_bytes, const uint32_t fxv,
                                                    const uint32_t alpha_bits, const uint32_t cfix_v, const int nh,
                                                    const double* __restrict__ asd_lds, const int A, const uint32_t T) {
  uint32_t pr = (uint32_t)(uintptr_t)(lds_byte*)acc_bytes + row_bytes;
  asm volatile("" : "+v"(pr));
  uint32_t s_prev = 0;
  for (int hh = 0; hh < nh; hh++) {
    const uint32_t C = (uint32_t)__builtin_amdgcn_readlane((int)cfix_v, hh);
    if (hh) (void)__hip_atomic_fetch_add((lds_u32*)(uintptr_t)fx_addr(s_prev, pr), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    uint32_t sv = fxv + C;
    if (__builtin_expect(__any((sv & 0xFFFFu) < T), 0)) {
      if ((sv & 0xFFFFu) < T) sv = (uint32_t)ppf_alpha_bin_exact(__uint_as_float(alpha_bits), asd_lds[hh], A) << 16;
    }
    s_prev = sv;
  }
  if (nh > 0) (void)__hip_atomic_fetch_add((lds_u32*)(uintptr_t)fx_addr(s_prev, pr), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ __launch_bounds__(VOTE_BLOCK) void k_vote(MatchArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t* red = reinterpret_cast<uint32_t*>(smem);                               /* LDS_HEADER */
  uint32_t* seg_prefix = reinterpret_cast<uint32_t*>(smem + LDS_HEADER);          /* VOTE_SEG + 64 */
  uint32_t* seg_off = seg_prefix + (VOTE_SEG + 64);                                /* VOTE_SEG */
  uint32_t* seg_cnt = seg_off + VOTE_SEG;                                          /* VOTE_SEG */
  uint32_t* seg_m = seg_cnt + VOTE_SEG;                                            /* VOTE_SEG: run length at run starts */
  uint32_t* seg_a32 = seg_m + VOTE_SEG;                                            /* VOTE_SEG */
  double* seg_a64 = reinterpret_cast<double*>(seg_a32 + VOTE_SEG);                 /* VOTE_SEG */
  const int A = a.num_angles;
  const int P = vote_pitch(A);
  const int GW = vote_guard(A);
  uint32_t* lds_acc = reinterpret_cast<uint32_t*>(seg_a64 + VOTE_SEG);             /* guard + cells */
  uint32_t* acc = lds_acc + GW;
  /* table entries carry byte offsets relative to the start of the guard ((GW + local_ref*P)*4), so the
   * vote address is entry.x + k*4 on top of a compile-time LDS offset */
  unsigned char* acc_bytes = reinterpret_cast<unsigned char*>(lds_acc);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); /* scalar: the work-item loop is wave-uniform */
  const int slot = blockIdx.x / a.n_tiles, tile = blockIdx.x - slot * a.n_tiles;
  const int r = (int)a.perm[slot]; /* heaviest reference points first */
  const int rg = a.ref_base + r;
  const int tile_base = tile * a.tile_refs;
  const int refs_here = min(a.tile_refs, a.n_model - tile_base);
  const int words = GW + refs_here * P;
  { /* clear guard + cells with 16-byte LDS stores (the region starts 16-byte aligned) */
    uint4* z = reinterpret_cast<uint4*>(lds_acc);
    for (int c = tid; c < words / 4; c += VOTE_BLOCK) z[c] = make_uint4(0u, 0u, 0u, 0u);
    for (int c = (words & ~3) + tid; c < words; c += VOTE_BLOCK) lds_acc[c] = 0u;
  }

  const uint32_t* __restrict__ boff = a.bucket_off + (size_t)tile * (a.n_buckets + 1);
#if PPF_ABL == 0 && PPF_VOTE_PIPE && PPF_VOTE_FIXED
  const uint4* __restrict__ records = a.records_fx;
  const uint4* __restrict__ records_f = a.records; /* float alpha_m: only read on the guard path */
#else
  const uint4* __restrict__ records = a.records;
#endif
  const HitRec* __restrict__ hits = a.hits + (size_t)r * a.hit_cap;
  const int n_hits = (int)a.hit_count[r];
  const float S = (float)((double)A / (4 * PPF_PI));
#ifdef PPF_FORCE_EXACT
  const float G = 1.0f;
#else
  const float G = PPF_GUARD_REL * (float)A;
#endif
  const float G2 = 2.0f * G;
  (void)G2;
  const float Og = 0.5f * (float)A + G;
  const uint32_t tail_bytes = (uint32_t)(lane * 4); /* per-lane guard word for lanes past the end of a bucket */

  const uint2* __restrict__ keys = a.keys_sorted + (size_t)r * a.hit_cap;
  unsigned long long* start_mask = reinterpret_cast<unsigned long long*>(red + 16); /* VOTE_WAVES x u64 */

  for (int seg0 = 0; seg0 < n_hits; seg0 += VOTE_SEG) {
    /* Stage a segment of the bucket-sorted hit list: this tile's bucket range, alpha_s, run
     * structure (a run = consecutive hits of one bucket), work items per run, exclusive scan. */
    __syncthreads(); /* previous segment fully consumed (and the accumulator clear, first time) */
    const int n_seg = min(VOTE_SEG, n_hits - seg0);
    uint32_t off = 0, cnt = 0;
    bool is_start = true; /* positions past the end terminate the last run */
    if (tid < n_seg) {
      const uint2 key = keys[seg0 + tid];
      const HitRec h = hits[key.y];
      if (key.x != 0xFFFFFFFFu) {
        off = boff[key.x];
        cnt = boff[key.x + 1] - off;
      }
#if PPF_ABL == 0 && PPF_VOTE_PIPE && PPF_VOTE_FIXED
      /* C = rint((A/2 - alpha_s * A/(4 pi)) * 65536) + 2 (the folded guard), fp64 */
      seg_a32[tid] = (uint32_t)(long long)__builtin_rint((0.5 * (double)A - h.alpha_s * ((double)A / (4 * PPF_PI))) * 65536.0) + 2u;
#else
      seg_a32[tid] = h.alpha32;
#endif
      seg_a64[tid] = h.alpha_s;
      is_start = (tid == 0) || (keys[seg0 + tid - 1].x != key.x);
    }
    const unsigned long long smask = __ballot(is_start);
    if (lane == 0) start_mask[wave] = smask;
    __syncthreads();
    uint32_t items = 0, m = 0;
    if (tid < n_seg && is_start && cnt > 0) {
      /* run length: distance to the next run start (or to the end of the segment) */
      int next = VOTE_SEG;
      unsigned long long rest = (lane == 63) ? 0ull : (smask >> (lane + 1));
      if (rest) {
        next = tid + 1 + (__ffsll((long long)rest) - 1);
      } else {
        for (int w = wave + 1; w < VOTE_WAVES; w++) {
          const unsigned long long mw = start_mask[w];
          if (mw) { next = w * 64 + (__ffsll((long long)mw) - 1); break; }
        }
      }
      m = (uint32_t)(min(next, n_seg) - tid);
      items = ((m + VOTE_MAX_HITS - 1) / VOTE_MAX_HITS) * ((cnt + VOTE_CHUNK - 1) / VOTE_CHUNK);
    }
    seg_off[tid] = off;
    seg_cnt[tid] = cnt;
    seg_m[tid] = m;
    uint32_t incl = items;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(incl, o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) red[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int k = 0; k < VOTE_WAVES; k++) {
      const uint32_t w = red[k];
      if (k < wave) woff += w;
      total += w;
    }
    seg_prefix[tid] = woff + incl - items; /* exclusive */
    if (tid < 64) seg_prefix[VOTE_SEG + tid] = total; /* sentinel + padding for the 64-wide look-ahead */
#if PPF_VOTE_DYNAMIC
    if (tid == 0) red[48] = VOTE_WAVES; /* next unclaimed work item (each wave starts with item == its id) */
#endif
    __syncthreads();

    /* waves take work items round-robin; the owning run start is found with a 64-wide look-ahead
     * from the previous one (positions that start no run in this tile have 0 items and are skipped) */
    int h = 0;
#if PPF_VOTE_DYNAMIC
    /* work items are claimed from an LDS counter as waves become free (items of one segment differ by
     * orders of magnitude in size); a wave's items still come in increasing order */
    for (uint32_t item = wave; item < total;
         item = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lane == 0 ? atomicAdd(&red[48], 1u) : 0u))) {
#else
    for (uint32_t item = wave; item < total; item += VOTE_WAVES) {
#endif
      while (true) { /* advance h to the last position with prefix <= item */
        const uint32_t pv = seg_prefix[min(h + 1 + lane, VOTE_SEG + 63)];
        const unsigned long long le = __ballot(pv <= item);
        const int adv = __popcll(le);
        h += adv;
        if (adv < 64) break;
      }
      /* the staged values are the same in every lane: move them to scalar registers so the item
       * runs on scalar control flow and scalar base addresses */
      const uint32_t local = item - (uint32_t)__builtin_amdgcn_readfirstlane((int)seg_prefix[h]);
      const uint32_t c_all = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg_cnt[h]);
      const uint32_t m_all = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg_m[h]);
      const uint32_t nchunk = (c_all + VOTE_CHUNK - 1) / VOTE_CHUNK;
      const uint32_t sub = local / nchunk, chunk = local - sub * nchunk;
      const uint32_t o = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg_off[h]) + chunk * VOTE_CHUNK;
#if PPF_ABL == 6 /* diagnostic: empty work items */
      const uint32_t c = 0;
#else
      const uint32_t c = min((uint32_t)VOTE_CHUNK, c_all - chunk * VOTE_CHUNK);
#endif
      const int h0 = h + (int)(sub * VOTE_MAX_HITS);
      const int nh = min((int)VOTE_MAX_HITS, h + (int)m_all - h0);
      /* lane l holds the folded offset of hit h0+l: Ohg = A/2 + G - alpha_s*S */
      float ohg_v = 0.f;
      if (lane < nh) ohg_v = Og - __uint_as_float(seg_a32[h0 + lane]) * S;
      const uint4* __restrict__ src = records + o;
#if PPF_ABL == 0 && PPF_VOTE_PIPE && PPF_VOTE_FIXED
      const uint32_t cfix_v = lane < nh ? seg_a32[h0 + lane] : 0u; /* lane l: fixed-point offset of hit h0+l */
      const uint4* __restrict__ srcf = records_f + o;
#ifdef PPF_FORCE_EXACT
      const uint32_t T = 0x10000u;
#else
      const uint32_t T = 4u;
#endif
      (void)ohg_v;
#endif
      constexpr uint32_t B = 64 * VOTE_UNROLL; /* records per batch */
      const uint32_t nfull = c / B;
      /* full batches, software-pipelined over two register sets: the loads of batch b+1 are in
       * flight while batch b is voted for every hit of the item.  The prefetch is unconditional
       * (the last one re-reads the final batch): a conditional one would merge two control-flow
       * paths and force the compiler into a vmcnt that also waits for the prefetch. */
      uint4 ea[VOTE_UNROLL], eb[VOTE_UNROLL];
      if (nfull) load_records<VOTE_UNROLL>(ea, src, 0, lane);
      uint32_t b = 0;
      while (b < nfull) {
        load_records<VOTE_UNROLL>(eb, src, min(b + 1, nfull - 1) * B, lane);
#if PPF_ABL == 0 && PPF_VOTE_PIPE && PPF_VOTE_FIXED
        vote_hits_fx<VOTE_UNROLL>(acc_bytes, ea, VOTE_UNROLL, cfix_v, nh, &seg_a64[h0], A, T, srcf, b * B, c, lane);
#elif PPF_ABL == 0 && PPF_VOTE_PIPE
        vote_hits<VOTE_UNROLL>(acc_bytes, ea, VOTE_UNROLL, S, ohg_v, nh, &seg_a64[h0], G2, A);
#else
        for (int hh = 0; hh < nh; hh++) {
          const float Ohg = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ohg_v), hh));
          cast_votes<VOTE_UNROLL>(acc_bytes, ea, VOTE_UNROLL, S, Ohg, &seg_a64[h0 + hh], G2, A);
        }
#endif
        if (++b >= nfull) break;
        load_records<VOTE_UNROLL>(ea, src, min(b + 1, nfull - 1) * B, lane);
#if PPF_ABL == 0 && PPF_VOTE_PIPE && PPF_VOTE_FIXED
        vote_hits_fx<VOTE_UNROLL>(acc_bytes, eb, VOTE_UNROLL, cfix_v, nh, &seg_a64[h0], A, T, srcf, b * B, c, lane);
#elif PPF_ABL == 0 && PPF_VOTE_PIPE
        vote_hits<VOTE_UNROLL>(acc_bytes, eb, VOTE_UNROLL, S, ohg_v, nh, &seg_a64[h0], G2, A);
#else
        for (int hh = 0; hh < nh; hh++) {
          const float Ohg = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ohg_v), hh));
          cast_votes<VOTE_UNROLL>(acc_bytes, eb, VOTE_UNROLL, S, Ohg, &seg_a64[h0 + hh], G2, A);
        }
#endif
        ++b;
      }
      const uint32_t e0 = nfull * B;
#if PPF_ABL == 0 && PPF_VOTE_PIPE
      if (e0 < c && c - e0 <= 32) { /* at most 64 entries left: one entry per lane */
        const uint32_t e = e0 + ((uint32_t)lane >> 1);
        const uint4 r = src[min(e, c - 1)];
        const bool second = (lane & 1) != 0;
        const uint32_t row_bytes = e < c ? (second ? r.y : r.x) : tail_bytes;
#if PPF_VOTE_FIXED
        const uint4 rf = srcf[min(e, c - 1)];
        vote_hits_single_fx(acc_bytes, row_bytes, second ? r.w : r.z, second ? rf.w : rf.z, cfix_v, nh, &seg_a64[h0], A, T);
#else
        vote_hits_single(acc_bytes, row_bytes, second ? r.w : r.z, S, ohg_v, nh, &seg_a64[h0], G2, A);
#endif
      } else
#endif
      if (e0 < c) { /* tail: clamped addresses; lanes past the end vote into their guard word */
#pragma unroll
        for (int u = 0; u < VOTE_UNROLL; u++) {
          const uint32_t e = e0 + u * 64 + lane;
          ea[u] = src[min(e, c - 1)];
          if (e >= c) { ea[u].x = tail_bytes; ea[u].y = tail_bytes; }
        }
        const int n_valid = (int)((c - e0 + 63) / 64);
#if PPF_ABL == 0 && PPF_VOTE_PIPE
        /* most buckets are smaller than a batch (median 18 records): only the 64-record groups that hold data get
         * their bin arithmetic, through an instantiation per group count */
        static_assert(VOTE_UNROLL == 4, "tail dispatch below assumes 4 groups per batch");
#if PPF_VOTE_FIXED
        switch (n_valid) {
          case 1: vote_hits_fx<1>(acc_bytes, ea, 1, cfix_v, nh, &seg_a64[h0], A, T, srcf, e0, c, lane); break;
          case 2: vote_hits_fx<2>(acc_bytes, ea, 2, cfix_v, nh, &seg_a64[h0], A, T, srcf, e0, c, lane); break;
          case 3: vote_hits_fx<3>(acc_bytes, ea, 3, cfix_v, nh, &seg_a64[h0], A, T, srcf, e0, c, lane); break;
          default: vote_hits_fx<4>(acc_bytes, ea, 4, cfix_v, nh, &seg_a64[h0], A, T, srcf, e0, c, lane); break;
        }
#else
        switch (n_valid) {
          case 1: vote_hits<1>(acc_bytes, ea, 1, S, ohg_v, nh, &seg_a64[h0], G2, A); break;
          case 2: vote_hits<2>(acc_bytes, ea, 2, S, ohg_v, nh, &seg_a64[h0], G2, A); break;
          case 3: vote_hits<3>(acc_bytes, ea, 3, S, ohg_v, nh, &seg_a64[h0], G2, A); break;
          default: vote_hits<4>(acc_bytes, ea, 4, S, ohg_v, nh, &seg_a64[h0], G2, A); break;
        }
#endif
#else
        for (int hh = 0; hh < nh; hh++) {
          const float Ohg = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ohg_v), hh));
          cast_votes<VOTE_UNROLL>(acc_bytes, ea, n_valid, S, Ohg, &seg_a64[h0 + hh], G2, A);
        }
#endif
      }
    }
  }
  __syncthreads();

  /* Scan in the reference's order (model ref ascending, alpha bin ascending, strict >) == smallest
   * upstream flat index ref*A + bin among the maxima; the spill cell of row ref-1 is folded into
   * (ref, bin 0) on the way.  Also the exact vote total of the tile. */
  uint32_t* dump = a.acc_dump ? a.acc_dump + (size_t)rg * a.n_model * A + (size_t)tile_base * A : nullptr;
  uint32_t bv = 0, bi = 0xFFFFFFFFu;
  unsigned long long sum = 0;
  /* one thread per accumulator row: consecutive threads read consecutive rows, pitch P is odd -> no bank
   * conflicts, no integer division; bins ascending with strict > keeps the row's first maximum */
  for (int ref = tid; ref < refs_here; ref += VOTE_BLOCK) {
    const uint32_t* row = acc + ref * P;
    for (int bin = 0; bin < A; bin++) {
      uint32_t v = row[bin];
      if (bin == 0 && ref > 0) v += row[A - P]; /* spill cell of the previous row */
      if (dump) dump[ref * A + bin] = v;
      sum += v;
      if (v > bv) { bv = v; bi = (uint32_t)(ref * A + bin); }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t v2 = __shfl_down(bv, o), i2 = __shfl_down(bi, o);
    sum += __shfl_down(sum, o);
    if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
  }
  uint32_t* red_v = seg_prefix; /* staging arrays are free now */
  uint32_t* red_i = seg_prefix + VOTE_WAVES;
  __syncthreads();
  if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
  __syncthreads();
  if (wave == 0) {
    uint32_t v = (lane < VOTE_WAVES) ? red_v[lane] : 0u;
    uint32_t ix = (lane < VOTE_WAVES) ? red_i[lane] : 0xFFFFFFFFu;
#pragma unroll
    for (int o = VOTE_WAVES / 2; o > 0; o >>= 1) {
      const uint32_t v2 = __shfl_down(v, o), i2 = __shfl_down(ix, o);
      if (v2 > v || (v2 == v && i2 < ix)) { v = v2; ix = i2; }
    }
    if (lane == 0) a.partial[(size_t)rg * a.n_tiles + tile] = make_uint2(v, ix);
  }
  if (lane == 0 && sum) atomicAdd(&a.cellsum[(size_t)rg * a.n_tiles + tile], sum);
}

/* fixed LDS of k_vote: header + hit staging (the guard and the cells are sized per model) */
constexpr size_t VOTE_LDS_FIXED = LDS_HEADER + (size_t)(VOTE_SEG + 64) * 4 + (size_t)VOTE_SEG * 4 * 4 + (size_t)VOTE_SEG * 8;

#endif /* PPF_MATCH_KERNELS_H */
