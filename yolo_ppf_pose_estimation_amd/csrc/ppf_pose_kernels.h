/*
 * ppf_pose_kernels.h — what follows the vote: tile merge + pose assembly (k_finalize, rows A5 tail / A8), ranking (k_rank), pose clustering
 * (row A7: k_clm_*, k_cluster_*) and the device-side result blocks.  Included by ppf_hip.hip after the match kernels.
 */
#ifndef PPF_POSE_KERNELS_H
#define PPF_POSE_KERNELS_H

/* ---- diagnostic: evaluate the deterministic math and the pair feature on the device ----------------- */
__global__ void k_debug_math(int fn, const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ out,
                             int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  switch (fn) {
    case 0: out[i] = ppf_acos(x[i]); break;
    case 1: out[i] = ppf_sin(x[i]); break;
    case 2: out[i] = ppf_cos(x[i]); break;
    case 3: out[i] = ppf_atan2(x[i], y[i]); break;
    case 4: out[i] = ppf_sqrt(x[i]); break;
    default: out[i] = x[i] / y[i]; break;
  }
}

/* ---- finalize: merge tiles, assemble the raw pose (rows A5 tail + A8) ---------------------- */
struct FinalArgs {
  CloudSoA surf, model;
  int scene_step, ref_offset, ref_stride, n_ref;
  int n_tiles, tile_refs, num_angles;
  int alpha_2pi; /* PCL's alpha binning: the winning bin stands for idx * 2pi/A - pi */
  int acc32;     /* every (reference point, tile) was voted with 32-bit cells: two partial results per tile (one per half of
                    its rows) and the edge values; otherwise only those flagged in ovf_items */
  const uint32_t* ovf_items;
  const uint32_t* edge;
  const uint2* partial;
  const unsigned long long* cellsum;
  const unsigned long long* pairs;
  ppf_vote* votes;
  ppf_pose* poses;
  unsigned long long* totals; /* [0] votes, [1] pairs */
  const unsigned long long* item_votes; /* votes per (reference point, tile), or NULL */
  unsigned long long* need_hist;        /* [2][ACC_HIST]: votes of the items that needed 32-bit cells / that did not, by size class of the item */
};

/* size classes of a (reference point, tile) by the votes it casts: four per octave */
constexpr int ACC_HIST = 256;
__host__ __device__ inline int acc_hist_class(unsigned long long v) {
  if (v < 4ull) return (int)v;
  int lg = 63;
  while (!((v >> lg) & 1ull)) lg--;
  return lg * 4 + (int)((v >> (lg - 2)) & 3ull);
}
__host__ __device__ inline unsigned long long acc_hist_lower(int cls) { /* smallest count of a class */
  if (cls < 8) return cls < 4 ? (unsigned long long)cls : 4ull; /* classes 4..7 are empty (lg = 1 gives 4 + ...: never reached below lg 2) */
  const int lg = cls / 4, fr = cls % 4;
  return (unsigned long long)(4 + fr) << (lg - 2);
}

__global__ __launch_bounds__(64) void k_finalize(FinalArgs a) {
  /* the call's totals: a workgroup (ONE wave) adds its reference points' votes and pairs up in LDS and sends one pair of atomics
   * (2,500 threads adding to two addresses were most of this kernel's 15 us) */
  __shared__ unsigned long long s_tot[2];
  if (threadIdx.x == 0) { s_tot[0] = 0ull; s_tot[1] = 0ull; }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.n_ref) return;
  uint32_t maxVotes = 0, flat = 0;
  unsigned long long nv = 0;
  for (int t = 0; t < a.n_tiles; t++) {
    const size_t slot = ((size_t)r * a.n_tiles + t) * 2;
    const uint2 p = a.partial[slot];
    nv += a.cellsum[(size_t)r * a.n_tiles + t];
    if (p.x > maxVotes) { maxVotes = p.x; flat = (uint32_t)(t * a.tile_refs * a.num_angles) + p.y; }
    uint32_t cell_max = p.x;
    const bool wide = a.acc32 || a.ovf_items[(size_t)r * a.n_tiles + t];
    if (wide) cell_max = max(cell_max, a.partial[slot + 1].x);
    if (a.item_votes && a.need_hist) { /* what the host learns the 32-bit limit from: did an item of this size need 32-bit cells? */
      const unsigned long long V = a.item_votes[(size_t)r * a.n_tiles + t];
      const bool needed = wide && cell_max > 65535u;
      if (V) atomicAdd(&a.need_hist[(needed ? 0 : ACC_HIST) + acc_hist_class(V)], V);
    }
    if (wide) { /* the high-half rows come after the low-half rows; bin 0 of their first row still lacks the spill
                      cell of the row before it, which the low halves' workgroup counted */
      uint2 q = a.partial[slot + 1];
      const uint32_t carry = a.edge[slot];
      if (carry) {
        const uint32_t cv = a.edge[slot + 1] + carry, ci = (uint32_t)(vote_half_rows(a.tile_refs) * a.num_angles);
        if (cv > q.x || (cv == q.x && ci <= q.y)) q = make_uint2(cv, ci);
      }
      if (q.x > maxVotes) { maxVotes = q.x; flat = (uint32_t)(t * a.tile_refs * a.num_angles) + q.y; }
    }
  }
  const uint32_t refIndMax = maxVotes ? flat / (uint32_t)a.num_angles : 0u;
  const uint32_t alphaIndMax = maxVotes ? flat % (uint32_t)a.num_angles : 0u;
  ppf_vote v;
  v.ref_ind_max = refIndMax; v.alpha_ind_max = alphaIndMax; v.max_votes = maxVotes;
  a.votes[r] = v;
  atomicAdd(&s_tot[0], nv);
  atomicAdd(&s_tot[1], a.pairs[r]);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (threadIdx.x == 0) { /* lane 0 is a reference point in every workgroup of the grid */
    atomicAdd(&a.totals[0], s_tot[0]);
    atomicAdd(&a.totals[1], s_tot[1]);
  }

  const int i_ref = (a.ref_offset + r * a.ref_stride) * a.scene_step;
  double Rsg[9], tsg[3], RInv[9], tInv[3], Rmg[9], tmg[3];
  ppf_transform_rt(ld3(a.surf.x, a.surf.y, a.surf.z, i_ref), ld3(a.surf.nx, a.surf.ny, a.surf.nz, i_ref), Rsg, tsg);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) RInv[i * 3 + j] = Rsg[j * 3 + i];
  ppf_vec3 rt = ppf_mul33(RInv, ppf_mk3(tsg[0], tsg[1], tsg[2]));
  tInv[0] = -rt.x; tInv[1] = -rt.y; tInv[2] = -rt.z;
  ppf_transform_rt(ld3(a.model.x, a.model.y, a.model.z, (int)refIndMax),
                   ld3(a.model.nx, a.model.ny, a.model.nz, (int)refIndMax), Rmg, tmg);
  double TsgInv[16], Tmg[16], Talpha[16], tmp[16], raw[16];
  ppf_rt_to_pose(RInv, tInv, TsgInv);
  ppf_rt_to_pose(Rmg, tmg, Tmg);
  const double alpha = a.alpha_2pi ? ((int)alphaIndMax * (2 * PPF_PI)) / a.num_angles - PPF_PI
                                   : ((int)alphaIndMax * (4 * PPF_PI)) / a.num_angles - 2 * PPF_PI;
  const double sx = ppf_sin(alpha), cx = ppf_cos(alpha);
  const double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
  const double t0[3] = {0, 0, 0};
  ppf_rt_to_pose(Rx, t0, Talpha);
  ppf_mat44_mul(Talpha, Tmg, tmp);
  ppf_mat44_mul(TsgInv, tmp, raw);
  ppf_pose P;
  for (int k = 0; k < 16; k++) P.pose[k] = raw[k];
  double Rr[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Rr[i * 3 + j] = raw[i * 4 + j];
  P.t[0] = raw[3]; P.t[1] = raw[7]; P.t[2] = raw[11];
  P.angle = ppf_angle_from_trace(Rr[0] + Rr[4] + Rr[8]);
  ppf_dcm_to_quat(Rr, P.q);
  P.alpha = alpha; P.residual = 0; P.model_index = refIndMax; P.num_votes = maxVotes;
  a.poses[r] = P;
}

/* ---- pose clustering (row A7: clusterPoses / matchPose / PoseCluster3D), one workgroup ---------------
 * 1. rank poses by (votes desc, input index asc)   [the reference's std::sort is not stable; this total
 *    order is the frozen one]                       O(n^2) counting, n <= a few thousand
 * 2. greedy: in rank order, a pose joins the FIRST cluster (creation order) whose first pose is within
 *    position_threshold (|dt|) and rotation_threshold (|angle difference|), else it opens a cluster.
 *    Sequential over poses, parallel over cluster heads (min-reduce of the matching cluster index).
 * 3. per cluster: quaternion / translation sums taken in joining order (fp64, same order as the CPU
 *    restatement, so results are bit-identical), plain or vote-weighted mean, pose rebuilt from the mean
 *    quaternion; cluster votes = sum of member votes.
 * 4. clusters ranked by (votes desc, creation order asc) and written out.
 */
__global__ __launch_bounds__(256) void k_widen_u32(const uint32_t* __restrict__ in, int n, unsigned long long* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}

/* Generic ranking: perm[rank] = i and rank_of[i] = rank for keys sorted (key desc, index asc).  n may live on the
 * device (n_dev != nullptr).  O(n^2) spread wide: a workgroup ranks 16 keys, 16 threads per key each counting every
 * 16th key of a 1024-key LDS tile, partial counts added by shuffles. */
constexpr int RANK_KEYS = 16; /* keys per workgroup of 256 threads */
__global__ __launch_bounds__(256) void k_rank(const unsigned long long* __restrict__ keys, int n_host,
                                              const uint32_t* __restrict__ n_dev, uint32_t* __restrict__ perm,
                                              uint32_t* __restrict__ rank_of) {
  __shared__ unsigned long long tile[1024];
  const int n = n_dev ? (int)*n_dev : n_host;
  if ((int)(blockIdx.x * RANK_KEYS) >= n) return; /* whole workgroup out of range */
  const int i = blockIdx.x * RANK_KEYS + (threadIdx.x >> 4), part = threadIdx.x & 15;
  const unsigned long long ki = i < n ? keys[i] : 0ull;
  uint32_t rank = 0;
  for (int j0 = 0; j0 < n; j0 += 1024) {
    const int cnt = min(1024, n - j0);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) tile[t] = keys[j0 + t];
    __syncthreads();
    for (int t = part; t < cnt; t += 16) {
      const unsigned long long kj = tile[t];
      rank += (kj > ki || (kj == ki && j0 + t < i)) ? 1u : 0u; /* keys before position i win ties */
    }
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) rank += (uint32_t)__shfl_xor((int)rank, o);
  if (i >= n || part != 0) return;
  if (perm) perm[rank] = (uint32_t)i;
  if (rank_of) rank_of[i] = rank;
}

struct ClusterArgs {
  const ppf_pose* in;
  int n, num_poses;
  double pos_thr, rot_thr;
  int weighted;
  int rot_relative;      /* rotation test on the relative rotation of two poses (PCL) instead of their angle difference */
  double cos_half_rot;   /* cos(rot_thr / 2): |qa . qb| above it <=> relative angle below rot_thr */
  /* global scratch */
  const uint32_t* order; /* [n] rank -> pose (k_rank on the vote keys) */
  uint32_t* assign;   /* [n] rank position -> cluster */
  uint32_t* head;     /* [n] cluster -> pose index of its first member */
  uint32_t* crank;    /* [n] cluster -> output slot (k_rank on the cluster votes) */
  uint32_t* coff;     /* [n+1] cluster -> first member slot */
  uint32_t* gvotes;   /* [n] votes of the members, in member-slot order */
  uint32_t* g_sizes;  /* [n] cluster sizes when the LDS variant does not fit */
  unsigned long long* cvotes; /* [n] */
  double* gq;         /* [7n] q0..q3,t0..t2 of the members, in member-slot order */
  double* g_heads;    /* [4n] cluster heads when the LDS variant does not fit */
  ppf_pose* out;      /* [n] */
  uint32_t* n_out;
};

/*
 * Greedy first-match assignment (step 2 of clusterPoses) without one barrier round per pose: poses are
 * taken CL_BLOCK at a time in rank order.
 *   A. the workgroup looks every pose of the round up among the clusters that existed BEFORE the round; a hit
 *      there is final, because clusters opened later have larger indices and the rule is "first cluster".
 *   B. one wave then walks the round's unmatched poses in order; each is compared only with the clusters
 *      opened inside this round (held in registers), joins the first match or opens one.
 * The serial part is proportional to the number of clusters opened and only touches LDS: cluster heads
 * (32 B per pose, IN_LDS when n <= CLUSTER_LDS_MAX) and a 1024-pose exchange buffer.  Sizes, joining order and
 * votes are computed afterwards in parallel (k_cluster_sizes / _offsets / _members).
 */
constexpr int CLUSTER_LDS_MAX = 3600;

/* matchPose(): |dt| < position_threshold && |angle difference| < rotation_threshold, with the reference's
 * sqrt only evaluated when the squared distance is within 1e-12 (relative) of the squared threshold */
__device__ __forceinline__ bool pose_matches(double hx, double hy, double hz, double ha, double tx, double ty, double tz,
                                             double ang, double pos_thr, double pos_thr2, double rot_thr) {
  const double dx = hx - tx, dy = hy - ty, dz = hz - tz;
  const double d2 = dx * dx + dy * dy + dz * dz;
  const double phi = ppf_fabs(ang - ha);
  if (!(phi < rot_thr)) return false;
  if (d2 < pos_thr2 * (1.0 - 1e-12)) return true;
  if (d2 > pos_thr2 * (1.0 + 1e-12)) return false;
  return ppf_sqrt(d2) < pos_thr;
}

#ifndef PPF_CL_BLOCK
#define PPF_CL_BLOCK 64
#endif
constexpr int CL_BLOCK = PPF_CL_BLOCK;   /* poses resolved per round (64/128/256) */
constexpr int CL_PARTS = 1024 / CL_BLOCK; /* threads per pose in step A */
constexpr int CL_SLOTS = CL_BLOCK / 64;  /* clusters opened in a round, held in registers: slot i of lane l = i*64 + l */

template <bool IN_LDS>
__global__ __launch_bounds__(1024) void k_cluster_assign(ClusterArgs a) {
  extern __shared__ __align__(16) unsigned char csm[];
  __shared__ uint32_t s_nclusters;
  __shared__ uint32_t s_match[CL_BLOCK], s_order[CL_BLOCK];
  __shared__ double s_pose[4][CL_BLOCK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = a.n;
  const int np = min(a.num_poses, n);
  double* hx;
  if constexpr (IN_LDS) hx = reinterpret_cast<double*>(csm);
  else hx = a.g_heads;
  double* hy = hx + n; double* hz = hy + n; double* ha = hz + n;
  const double pos_thr2 = a.pos_thr * a.pos_thr;
  if (tid == 0) s_nclusters = 0;
  __syncthreads();
  for (int s0 = 0; s0 < np; s0 += CL_BLOCK) {
    const uint32_t nc0 = s_nclusters;
    /* A. clusters that existed before this round: CL_PARTS threads per pose, each scanning every CL_PARTS-th
     *    cluster in ascending order; the first match overall is the minimum over them (atomicMin). */
    const int pl = tid & (CL_BLOCK - 1), part = tid / CL_BLOCK;
    const int s = s0 + pl;
    if (part == 0) {
      if (s < np) {
        const uint32_t pi = a.order[s];
        s_pose[0][pl] = a.in[pi].t[0]; s_pose[1][pl] = a.in[pi].t[1]; s_pose[2][pl] = a.in[pi].t[2];
        s_pose[3][pl] = a.in[pi].angle;
        s_order[pl] = pi;
      }
      s_match[pl] = 0xFFFFFFFFu;
    }
    __syncthreads();
    if (s < np) {
      const double tx = s_pose[0][pl], ty = s_pose[1][pl], tz = s_pose[2][pl], ang = s_pose[3][pl];
      for (uint32_t c = (uint32_t)part; c < nc0; c += CL_PARTS) {
        if (pose_matches(hx[c], hy[c], hz[c], ha[c], tx, ty, tz, ang, a.pos_thr, pos_thr2, a.rot_thr)) {
          atomicMin(&s_match[pl], c);
          break;
        }
      }
    }
    __syncthreads();
    if (s < np && part == 0 && s_match[pl] != 0xFFFFFFFFu) a.assign[s] = s_match[pl]; /* final: older clusters win */
    /* B. one wave resolves, in rank order, the poses no older cluster took, against the clusters opened in
     *    this round (kept in registers) */
    if (wave == 0) {
      const int cnt = min(CL_BLOCK, np - s0);
      double rx[CL_SLOTS], ry[CL_SLOTS], rz[CL_SLOTS], ra[CL_SLOTS];
#pragma unroll
      for (int i = 0; i < CL_SLOTS; i++) { rx[i] = 0; ry[i] = 0; rz[i] = 0; ra[i] = 0; }
      uint32_t nnew = 0;
      for (int c0 = 0; c0 < cnt; c0 += 64) {
        const int kk = min(c0 + lane, CL_BLOCK - 1);
        const bool un = (c0 + lane) < cnt && s_match[kk] == 0xFFFFFFFFu;
        unsigned long long mask = __ballot(un);
        while (mask) {
          const int l = __ffsll((long long)mask) - 1;
          mask &= mask - 1;
          /* pose l of the chunk, read by every lane from the same LDS address (broadcast) */
          const int kl = c0 + l;
          const double tx = s_pose[0][kl], ty = s_pose[1][kl], tz = s_pose[2][kl], ang = s_pose[3][kl];
          uint32_t m = 0xFFFFFFFFu;
#pragma unroll
          for (int i = 0; i < CL_SLOTS; i++) {
            const bool live = (uint32_t)(i * 64 + lane) < nnew;
            const bool hit = live && pose_matches(rx[i], ry[i], rz[i], ra[i], tx, ty, tz, ang, a.pos_thr, pos_thr2, a.rot_thr);
            const unsigned long long bal = __ballot(hit);
            if (bal && m == 0xFFFFFFFFu) m = nc0 + (uint32_t)(i * 64) + (uint32_t)(__ffsll((long long)bal) - 1);
          }
          if (m == 0xFFFFFFFFu) { /* open a cluster: lane (nnew & 63) keeps it in slot nnew >> 6 */
            m = nc0 + nnew;
#pragma unroll
            for (int i = 0; i < CL_SLOTS; i++) {
              if ((uint32_t)(i * 64 + lane) == nnew) { rx[i] = tx; ry[i] = ty; rz[i] = tz; ra[i] = ang; }
            }
            if (lane == 0) {
              a.head[m] = s_order[kl];
              hx[m] = tx; hy[m] = ty; hz[m] = tz; ha[m] = ang;
            }
            nnew++;
          }
          if (lane == 0) a.assign[s0 + c0 + l] = m;
        }
      }
      if (lane == 0) s_nclusters = nc0 + nnew;
    }
    __threadfence_block();
    __syncthreads();
  }
  const int nc = (int)s_nclusters;
  if (tid == 0) *a.n_out = (uint32_t)nc;
  for (int c = tid; c < n; c += 1024) { a.cvotes[c] = 0; a.g_sizes[c] = 0; }
}

/* ---- the same greedy assignment through a match matrix ------------------------------------------------------
 * "Pose i joins the first cluster whose head matches it, else opens one" only depends on heads, and the head of a
 * cluster is its first pose.  So: (1) all pairwise tests head-candidate j < i against pose i in parallel into a bit
 * matrix (row i, bit j); (2) ONE wave walks the rows in rank order keeping the set of heads as a bit mask in
 * registers: i is a head iff row_i & heads == 0 -- a ballot per pose instead of a scan of the clusters; rows are
 * staged through LDS a block at a time by the whole workgroup; (3) every pose finds its cluster in parallel: the
 * lowest set bit of row_i & heads, numbered by the heads before it.  Same predicate, same order, same result as
 * k_cluster_assign, 0.72 ms -> tens of microseconds at 2,500 poses. */
constexpr int CLM_MAX_WORDS = 180;  /* 64 staged rows must fit the LDS window: up to 11,520 poses */
constexpr int CLM_LDS_BYTES = 96 * 1024;

/* relative-rotation variant of matchPose() (PCL's posesWithinErrorBounds): the angle of Ra^T Rb is 2 acos(|qa . qb|) */
__device__ __forceinline__ bool pose_matches_rel(double hx, double hy, double hz, const double* hq, double tx, double ty, double tz,
                                                 const double* q, double pos_thr, double pos_thr2, double cos_half_rot) {
  const double dx = hx - tx, dy = hy - ty, dz = hz - tz;
  const double d2 = dx * dx + dy * dy + dz * dz;
  const double d = ppf_fabs(hq[0] * q[0] + hq[1] * q[1] + hq[2] * q[2] + hq[3] * q[3]);
  if (!(d > cos_half_rot)) return false;
  if (d2 < pos_thr2 * (1.0 - 1e-12)) return true;
  if (d2 > pos_thr2 * (1.0 + 1e-12)) return false;
  return ppf_sqrt(d2) < pos_thr;
}

/* poses in rank order as SoA (x, y, z, angle, q0..q3) in g_heads; cluster counters cleared */
__global__ __launch_bounds__(256) void k_clm_gather(ClusterArgs a) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = a.n;
  if (s >= n) return;
  a.cvotes[s] = 0;
  a.g_sizes[s] = 0;
  if (s >= min(a.num_poses, n)) return;
  const ppf_pose& p = a.in[a.order[s]];
  a.g_heads[s] = p.t[0]; a.g_heads[(size_t)n + s] = p.t[1]; a.g_heads[2 * (size_t)n + s] = p.t[2]; a.g_heads[3 * (size_t)n + s] = p.angle;
  if (a.rot_relative)
    for (int k = 0; k < 4; k++) a.g_heads[(size_t)(4 + k) * n + s] = p.q[k];
}
/* bits[i*words + w] bit b = pose 64w+b (< i) matches pose i; one workgroup per pose i, one wave per word */
__global__ __launch_bounds__(256) void k_clm_matrix(ClusterArgs a, unsigned long long* __restrict__ bits, int words) {
  const int i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = a.n;
  const double* px = a.g_heads; const double* py = px + n; const double* pz = py + n; const double* pa = pz + n;
  const double tx = px[i], ty = py[i], tz = pz[i], ang = pa[i];
  const double pos_thr2 = a.pos_thr * a.pos_thr;
  const double* pq = pa + n; /* q0[n] q1[n] q2[n] q3[n], relative metric only */
  double qi[4] = {0, 0, 0, 0};
  if (a.rot_relative)
    for (int k = 0; k < 4; k++) qi[k] = pq[(size_t)k * n + i];
  for (int w = wave; w <= (i >> 6); w += 4) {
    const int j = (w << 6) + lane;
    bool hit = false;
    if (j < i) {
      if (a.rot_relative) {
        const double qj[4] = {pq[j], pq[(size_t)n + j], pq[2 * (size_t)n + j], pq[3 * (size_t)n + j]};
        hit = pose_matches_rel(px[j], py[j], pz[j], qj, tx, ty, tz, qi, a.pos_thr, pos_thr2, a.cos_half_rot);
      } else {
        hit = pose_matches(px[j], py[j], pz[j], pa[j], tx, ty, tz, ang, a.pos_thr, pos_thr2, a.rot_thr);
      }
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) bits[(size_t)i * words + w] = m;
  }
}
/* heads[w] = bit mask of the poses that open a cluster, prefix[w] = clusters opened before word w, *n_out = clusters.
 * Rows are staged through LDS in rounds (multiples of 64 rows, odd pitch against bank conflicts); all waves first AND
 * every row of the round with the head words of the earlier rounds.  One wave then walks the round 64 rows at a time
 * with ONE ROW PER LANE: each lane ANDs its row with the head words of the round's earlier groups, then the 64 rows of
 * the group are resolved among themselves: row r opens a cluster iff it matched no earlier head and none of the group's
 * rows before it that opened one. */
__global__ __launch_bounds__(1024) void k_clm_heads(ClusterArgs a, const unsigned long long* __restrict__ bits, int words, int pitch,
                                                    int rows_per_round, unsigned long long* __restrict__ heads, uint32_t* __restrict__ prefix) {
  extern __shared__ unsigned long long clm_lds[]; /* head words [words] | rows [rows_per_round][pitch] */
  unsigned long long* s_heads = clm_lds;
  unsigned long long* s_rows = clm_lds + words;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int np = min(a.num_poses, a.n);
  for (int w = tid; w < words; w += 1024) s_heads[w] = 0ull;
  for (int s0 = 0; s0 < np; s0 += rows_per_round) {
    const int cnt = min(rows_per_round, np - s0);
    const int wmax = ((s0 + cnt - 1) >> 6) + 1; /* words any row of the round can use */
    __syncthreads();
    for (int e = tid; e < cnt * wmax; e += 1024) {
      const int r = e / wmax, w = e - r * wmax;
      s_rows[r * pitch + w] = w <= ((s0 + r) >> 6) ? bits[(size_t)(s0 + r) * words + w] : 0ull;
    }
    __syncthreads();
    /* every row against the heads of the EARLIER ROUNDS (all known): one row per thread, all sixteen waves; the verdict
     * replaces the row's word 0, which nobody reads again.  What is left for the one wave that walks the groups are the
     * head words of this round's own groups. */
    const int W0 = s0 >> 6;
    if (W0 > 0) {
      for (int r = tid; r < cnt; r += 1024) {
        unsigned long long* row = s_rows + r * pitch;
        bool p = false;
        for (int w = 0; w < W0; w++) p |= (row[w] & s_heads[w]) != 0ull;
        row[0] = p ? 1ull : 0ull;
      }
      __syncthreads();
    }
    if (wave == 0) {
      for (int g0 = 0; g0 < cnt; g0 += 64) { /* s0 and g0 are multiples of 64: the group is word G of the mask */
        const int G = (s0 + g0) >> 6;
        const int r = g0 + lane;
        const bool valid = r < cnt;
        const unsigned long long* row = s_rows + (valid ? r : g0) * pitch;
        bool pre = !valid || (W0 > 0 && row[0] != 0ull);
        for (int w = W0; w < G; w++) pre |= (row[w] & s_heads[w]) != 0ull;
        const unsigned long long blk = valid ? row[G] : 0ull; /* matches among the rows of this group (lower ones) */
        const unsigned long long taken = __ballot(pre);
        /* the group among itself, all lanes at once: a row still undecided opens a cluster when none of the rows before it
         * that it matches is a head or still undecided, and is taken when one of them is a head; the lowest undecided row is
         * always decided, most rounds decide nearly all of them */
        unsigned long long gh = 0ull, cand = ~taken;
        while (cand) {
          const bool und = (cand >> lane) & 1ull;
          const unsigned long long nh = __ballot(und && (blk & (gh | cand)) == 0ull);
          const unsigned long long nt = __ballot(und && (blk & gh) != 0ull);
          gh |= nh;
          cand &= ~(nh | nt);
        }
        if (lane == 0) s_heads[G] = gh;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  __syncthreads();
  for (int w = tid; w < words; w += 1024) heads[w] = s_heads[w];
  if (tid == 0) {
    uint32_t run = 0;
    for (int w = 0; w < words; w++) { prefix[w] = run; run += (uint32_t)__popcll(s_heads[w]); }
    *a.n_out = run;
  }
}
/* assign[s] = cluster of pose s; head[c] = pose index of the cluster's first member */
__global__ __launch_bounds__(256) void k_clm_assign(ClusterArgs a, const unsigned long long* __restrict__ bits, int words,
                                                    const unsigned long long* __restrict__ heads, const uint32_t* __restrict__ prefix) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  const int np = min(a.num_poses, a.n);
  if (s >= np) return;
  const int ws = s >> 6;
  const unsigned long long hw = heads[ws];
  if ((hw >> (s & 63)) & 1ull) {
    const uint32_t c = prefix[ws] + (uint32_t)__popcll(hw & ((1ull << (s & 63)) - 1ull));
    a.assign[s] = c;
    a.head[c] = a.order[s];
    return;
  }
  for (int w = 0; w <= ws; w++) {
    const unsigned long long h = heads[w];
    const unsigned long long v = bits[(size_t)s * words + w] & h;
    if (v) {
      const int b = __ffsll((long long)v) - 1;
      a.assign[s] = prefix[w] + (uint32_t)__popcll(h & ((1ull << b) - 1ull));
      return;
    }
  }
}

/* cluster sizes and votes (integer atomics: order-free) */
__global__ __launch_bounds__(256) void k_cluster_sizes(ClusterArgs a) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  const int np = min(a.num_poses, a.n);
  if (s >= np) return;
  const uint32_t c = a.assign[s];
  atomicAdd(&a.g_sizes[c], 1u);
  atomicAdd(&a.cvotes[c], (unsigned long long)a.in[a.order[s]].num_votes);
}

/* exclusive scan of the cluster sizes -> member-slot offsets (one workgroup, chunks of 1024) */
__global__ __launch_bounds__(1024) void k_cluster_offsets(ClusterArgs a) {
  __shared__ uint32_t wtot[16];
  __shared__ uint32_t carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nc = (int)*a.n_out;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int c0 = 0; c0 < nc; c0 += 1024) {
    const int c = c0 + tid;
    const uint32_t v = c < nc ? a.g_sizes[c] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(incl, o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    uint32_t woff = carry;
    for (int k = 0; k < wave; k++) woff += wtot[k];
    if (c < nc) a.coff[c] = woff + incl - v;
    __syncthreads();
    if (tid == 1023) carry = woff + incl;
    __syncthreads();
  }
  if (tid == 0) a.coff[nc] = carry;
}

/* gather the members' q, t, votes into member-slot order = joining order: the joining index of a pose is
 * the number of earlier (rank order) poses of the same cluster */
__global__ __launch_bounds__(256) void k_cluster_members(ClusterArgs a) {
  __shared__ uint32_t tile[1024];
  const int np = min(a.num_poses, a.n);
  if ((int)(blockIdx.x * blockDim.x) >= np) return; /* whole workgroup out of range */
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t c = s < np ? a.assign[s] : 0xFFFFFFFFu;
  uint32_t rin = 0; /* earlier poses of the same cluster = this pose's position among the members */
  const int last = min(np, (int)((blockIdx.x + 1) * blockDim.x)); /* no thread of the workgroup looks past its own s */
  for (int j0 = 0; j0 < last; j0 += 1024) {
    const int cnt = min(1024, last - j0);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) tile[t] = a.assign[j0 + t];
    __syncthreads();
    const int upto = min(max(s - j0, 0), cnt);
    for (int t = 0; t < upto; t++) rin += tile[t] == c ? 1u : 0u;
  }
  if (s >= np) return;
  const uint32_t slot = a.coff[c] + rin;
  const ppf_pose& p = a.in[a.order[s]];
  double* g = a.gq + (size_t)slot * 7;
  g[0] = p.q[0]; g[1] = p.q[1]; g[2] = p.q[2]; g[3] = p.q[3]; g[4] = p.t[0]; g[5] = p.t[1]; g[6] = p.t[2];
  a.gvotes[slot] = p.num_votes;
}

/* steps 3 + 4 of clusterPoses: means in joining order (fp64, sequential per cluster: bit-identical to the CPU
 * restatement), pose rebuilt from the mean quaternion, written to the cluster's rank */
__global__ __launch_bounds__(64) void k_cluster_finish(ClusterArgs a) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= (int)*a.n_out) return;
  double q[4] = {0, 0, 0, 0}, t[3] = {0, 0, 0}, wsum = 0;
  const uint32_t k0 = a.coff[c], k1 = a.coff[c + 1];
  for (uint32_t k = k0; k < k1; k++) {
    const double* g = a.gq + (size_t)k * 7;
    if (a.weighted) {
      const double w = (double)a.gvotes[k];
      for (int j = 0; j < 4; j++) q[j] += w * g[j];
      for (int j = 0; j < 3; j++) t[j] += w * g[4 + j];
      wsum += w;
    } else {
      for (int j = 0; j < 4; j++) q[j] += g[j];
      for (int j = 0; j < 3; j++) t[j] += g[4 + j];
    }
  }
  const double inv = a.weighted ? 1.0 / wsum : 1.0 / (int)(k1 - k0);
  for (int j = 0; j < 3; j++) t[j] *= inv;
  for (int j = 0; j < 4; j++) q[j] *= inv;
  ppf_pose P = a.in[a.head[c]];
  double R[9];
  ppf_quat_to_dcm(q, R);
  for (int j = 0; j < 4; j++) P.q[j] = q[j];
  for (int j = 0; j < 3; j++) P.t[j] = t[j];
  ppf_rt_to_pose(R, t, P.pose);
  P.angle = ppf_angle_from_trace(R[0] + R[4] + R[8]);
  P.num_votes = (uint32_t)a.cvotes[c];
  a.out[a.crank[c]] = P;
}

__global__ void k_vote_keys(const ppf_pose* __restrict__ in, int n, unsigned long long* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[i] = in[i].num_votes;
}

/* the call's summary for the host, straight into pinned memory: out[0..15] totals, out[16] overflow word, out[17] clustered poses */
__global__ void k_summary(const unsigned long long* __restrict__ totals, const uint32_t* __restrict__ overflow, const uint32_t* __restrict__ n_final,
                          unsigned long long* __restrict__ out) {
  const int t = threadIdx.x;
  if (t < 16) out[t] = totals[t];
  else if (t == 16) out[16] = (unsigned long long)*overflow;
  else if (t == 17) out[17] = n_final ? (unsigned long long)*n_final : 0ull;
}

#endif /* PPF_POSE_KERNELS_H */
