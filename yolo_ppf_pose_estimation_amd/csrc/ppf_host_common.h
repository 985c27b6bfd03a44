/*
 * ppf_host_common.h — host-side plumbing shared by the C-ABI files: scans, sorts, device cloud sampling, the model / workspace
 * structs and the clustering launch sequence.  Included by ppf_hip.hip.
 */
#ifndef PPF_HOST_COMMON_H
#define PPF_HOST_COMMON_H

/* ============================================================================================ */
/* host side                                                                                      */
/* ============================================================================================ */
namespace {

/* a cloud argument's layout: x y z at floats 0..2, the normal at noff..noff+2, inside rows of `stride` floats */
bool bad_layout(int stride, int noff) { return stride < 6 || noff < 3 || noff + 3 > stride; }

uint32_t next_pow2(uint32_t v) {
  v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v++;
  return v;
}

void bbox_host(const float* pc, int n, int stride, float lo[3], float hi[3]) {
  for (int k = 0; k < 3; k++) { lo[k] = pc[k]; hi[k] = pc[k]; }
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) {
      const float v = pc[(size_t)i * stride + k];
      lo[k] = v < lo[k] ? v : lo[k];
      hi[k] = v > hi[k] ? v : hi[k];
    }
}

bool have_device() {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

constexpr size_t SCAN_ONE_MAX = 40960; /* elements k_scan_one takes (ten rounds) */
ppf_status device_exclusive_scan(const uint32_t* in, uint32_t* out, size_t n, hipStream_t st) {
  if (n == 0) return PPF_OK;
  if (n > 1024 && n <= SCAN_ONE_MAX) { /* one workgroup, one launch (<= 1,024 elements are one block of the general kernel anyway) */
    k_scan_one<<<dim3(1), dim3(1024), 0, st>>>(in, out, n);
    HIPCHK(hipGetLastError());
    return PPF_OK;
  }
  const size_t nb = (n + 1023) / 1024;
  DevBuf<uint32_t> sums, sums_scan;
  if (nb > 1) {
    HIPCHK(sums.reserve(nb));
    HIPCHK(sums_scan.reserve(nb));
  }
  k_scan_block<<<dim3((unsigned)nb), dim3(256), 0, st>>>(in, out, nb > 1 ? sums.p : nullptr, n);
  HIPCHK(hipGetLastError());
  if (nb > 1) {
    ppf_status s = device_exclusive_scan(sums.p, sums_scan.p, nb, st);
    if (s != PPF_OK) return s;
    k_scan_add<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(out, sums_scan.p, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st)); /* sums buffers die at scope exit */
  }
  return PPF_OK;
}

struct CloudDev {
  DevBuf<float> buf; /* 6 planes of `pitch` floats */
  int n = 0, pitch = 0;
  CloudSoA view() const {
    CloudSoA c;
    c.x = buf.p; c.y = buf.p + pitch; c.z = buf.p + 2 * (size_t)pitch;
    c.nx = buf.p + 3 * (size_t)pitch; c.ny = buf.p + 4 * (size_t)pitch; c.nz = buf.p + 5 * (size_t)pitch;
    c.n = n;
    return c;
  }
  /* from a device AoS cloud */
  ppf_status load_device(const float* d_src, int rows, int stride, int noff, hipStream_t st) {
    n = rows;
    pitch = (rows + 63) & ~63;
    HIPCHK(buf.reserve((size_t)6 * std::max(pitch, 64)));
    if (rows > 0) {
      k_aos_to_soa<<<dim3((rows + 255) / 256), dim3(256), 0, st>>>(d_src, rows, stride, noff, buf.p, pitch);
      HIPCHK(hipGetLastError());
    }
    return PPF_OK;
  }
  /* from a host AoS cloud (packed rows of 6) */
  ppf_status load_host(const float* h_src, int rows, hipStream_t st) {
    n = rows;
    pitch = (rows + 63) & ~63;
    HIPCHK(buf.reserve((size_t)6 * std::max(pitch, 64)));
    std::vector<float> soa((size_t)6 * pitch, 0.f);
    for (int i = 0; i < rows; i++)
      for (int k = 0; k < 6; k++) soa[(size_t)k * pitch + i] = h_src[(size_t)i * 6 + k];
    HIPCHK(hipMemcpyAsync(buf.p, soa.data(), soa.size() * sizeof(float), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    return PPF_OK;
  }
};

/* Stable LSD radix sort of (key, value) pairs by key (8-bit digits, as many passes as max_key needs), then the
 * starts of the runs of equal keys.  On return *vals_sorted points at the sorted values (one of the two buffers),
 * starts[0..*n_runs) are the run starts.  One 4-byte read-back sizes `starts`. */
ppf_status sort_segments(DevBuf<uint32_t>& keys, DevBuf<uint32_t>& vals, DevBuf<uint32_t>& keys2, DevBuf<uint32_t>& vals2, int n,
                         unsigned long long max_key, DevBuf<uint32_t>& starts, uint32_t** vals_sorted, uint32_t* n_runs,
                         hipStream_t st) {
  DevBuf<uint32_t> hist, offs, flags, segid;
  const unsigned nb256 = (unsigned)((n + 255) / 256);
  int bits = 1;
  while (bits < 32 && (1ull << bits) <= max_key) bits++;
  const int nblk = (n + RS_BLOCK - 1) / RS_BLOCK;
  HIPCHK(hist.reserve((size_t)256 * nblk)); HIPCHK(offs.reserve((size_t)256 * nblk));
  uint32_t *ka = keys.p, *va = vals.p, *kb = keys2.p, *vb = vals2.p;
  for (int shift = 0; shift < bits; shift += 8) {
    k_rs_hist<<<dim3(nblk), dim3(RS_BLOCK), 0, st>>>(ka, n, shift, nblk, hist.p);
    HIPCHK(hipGetLastError());
    ppf_status s = device_exclusive_scan(hist.p, offs.p, (size_t)256 * nblk, st);
    if (s != PPF_OK) return s;
    k_rs_scatter<<<dim3(nblk), dim3(RS_BLOCK), 0, st>>>(ka, va, n, shift, nblk, offs.p, kb, vb);
    HIPCHK(hipGetLastError());
    std::swap(ka, kb); std::swap(va, vb);
  }
  HIPCHK(flags.reserve((size_t)n + 1)); HIPCHK(segid.reserve((size_t)n + 1));
  HIPCHK(hipMemsetAsync(flags.p + n, 0, sizeof(uint32_t), st));
  k_seg_flags<<<dim3(nb256), dim3(256), 0, st>>>(ka, n, flags.p);
  HIPCHK(hipGetLastError());
  ppf_status s = device_exclusive_scan(flags.p, segid.p, (size_t)n + 1, st);
  if (s != PPF_OK) return s;
  uint32_t n_rows = 0;
  HIPCHK(hipMemcpyAsync(&n_rows, segid.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(starts.reserve(std::max<uint32_t>(n_rows, 1)));
  k_seg_starts<<<dim3(nb256), dim3(256), 0, st>>>(flags.p, segid.p, n, starts.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st)); /* flags/segid die at scope exit */
  *vals_sorted = va;
  *n_runs = n_rows;
  return PPF_OK;
}

/* Row A2 on the device: d_src is a device AoS cloud; the sampled rows land in `dst` (SoA) and, when asked for,
 * in `host_rows` (N' x 6).  One 4-byte read-back sizes the output. */
ppf_status device_sample_cloud(const float* d_src, int n, int stride, int noff, float step, CloudDev& dst,
                               std::vector<float>* host_rows, hipStream_t st) {
  const int ns = (int)(1.0 / step);
  DevBuf<uint32_t> bbox, keys, vals, keys2, vals2, starts;
  HIPCHK(bbox.reserve(6));
  const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
  HIPCHK(hipMemcpyAsync(bbox.p, init, sizeof(init), hipMemcpyHostToDevice, st));
  const unsigned nb256 = (unsigned)((n + 255) / 256);
  k_bbox<<<dim3(std::max(1u, std::min((unsigned)((n + 2047) / 2048), 256u))), dim3(256), 0, st>>>(d_src, n, stride, bbox.p);
  HIPCHK(hipGetLastError());
  HIPCHK(keys.reserve(n)); HIPCHK(vals.reserve(n)); HIPCHK(keys2.reserve(n)); HIPCHK(vals2.reserve(n));
  k_cell_keys<<<dim3(nb256), dim3(256), 0, st>>>(d_src, n, stride, bbox.p, ns, keys.p, vals.p);
  HIPCHK(hipGetLastError());
  const unsigned long long max_key = (unsigned long long)ns * ns * ns + (unsigned long long)ns * ns + ns;
  uint32_t n_rows = 0;
  uint32_t* va = nullptr;
  ppf_status s = sort_segments(keys, vals, keys2, vals2, n, max_key, starts, &va, &n_rows, st);
  if (s != PPF_OK) return s;
  dst.n = (int)n_rows;
  dst.pitch = ((int)n_rows + 63) & ~63;
  HIPCHK(dst.buf.reserve((size_t)6 * std::max(dst.pitch, 64)));
  DevBuf<float> aos;
  if (host_rows) HIPCHK(aos.reserve((size_t)std::max<uint32_t>(n_rows, 1) * 6));
  if (n_rows) {
    k_seg_sum<<<dim3((n_rows + 63) / 64), dim3(64), 0, st>>>(d_src, stride, noff, va, starts.p, (int)n_rows, n, dst.buf.p, dst.pitch,
                                                          host_rows ? aos.p : nullptr);
    HIPCHK(hipGetLastError());
  }
  if (host_rows) {
    host_rows->resize((size_t)n_rows * 6);
    if (n_rows) HIPCHK(hipMemcpyAsync(host_rows->data(), aos.p, host_rows->size() * sizeof(float), hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipStreamSynchronize(st)); /* the scratch buffers above die at scope exit */
  return PPF_OK;
}

}  // namespace

struct HostCtx; /* warm context of the host-buffer entries (ppf_match_host.h) */

inline uint64_t next_model_serial() {
  static std::atomic<uint64_t> n{0};
  return ++n;
}

struct ppf_model {
  std::atomic<int> refcount{1};
  const uint64_t serial = next_model_serial(); /* what workspaces remember a model by (an address can be reused by a later model) */
  ppf_train_params params{};
  ppf_model_info info{};
  std::vector<float> sampled; /* host copy, n_ref x 6 */
  CloudDev cloud;
  DevBuf<SlotWord> slotmap;
  DevBuf<uint32_t> bucket_off;  /* n_tiles * (n_buckets + 1) */
  DevBuf<uint32_t> bucket_slot; /* n_buckets: hash slot of each dense bucket id */
  DevBuf<uint32_t> bucket_total; /* n_buckets: entries over all tiles */
  DevBuf<uint32_t> bucket_mid;   /* n_tiles * n_buckets: see k_bucket_mid */
  DevBuf<uint4> records;          /* pair records, see place_entry */
  DevBuf<int32_t> key_lut;        /* quantised key -> bucket, see k_build_key_lut */
  KeyDims kd{};
  uint64_t n_records = 0;
  int device = 0;
  /* Idle contexts of ppf_match / ppf_raw_votes / ppf_match_clouds on this model: workspace (learned pool sizes, scratch),
   * stream, pinned staging.  The table itself stays immutable; this list is the only state calls share, under its mutex. */
  mutable std::mutex ctx_mu;
  mutable std::vector<HostCtx*> ctx_idle;
  mutable size_t ctx_out = 0, ctx_peak = 0; /* calls in flight on this model now / at most so far (since the last trim) */
  ~ppf_model();
};

struct ppf_workspace {
  CloudDev surf, edge;
  DevBuf<float> staging;
  DevBuf<uint2> partial;
  DevBuf<uint32_t> half_edge; /* see MatchArgs::edge */
  DevBuf<uint32_t> ovf_items, ovf_list; /* see MatchArgs */
  /* hit scratch of one batch of reference points (see ppf_match_kernels.h) */
  DevBuf<double> frames;
  DevBuf<uint2> raw;                     /* striped pool of {bucket, j} */
  DevBuf<uint32_t> cursors;              /* CUR_WORDS */
  DevBuf<uint2> chunk_desc;
  DevBuf<unsigned long long> hit_count;
  DevBuf<double> s_a64;
  DevBuf<uint16_t> s_cell;
  DevBuf<uint4> runs;
  DevBuf<unsigned char> tables;          /* count tables of a batch's many-hit runs (k_tables -> k_vote), TBL_BYTES each */
  DevBuf<uint2> table_desc;              /* what each table covers: {first sorted hit, hits} */
  DevBuf<uint2> run_blocks;
  DevBuf<unsigned long long> work;
  DevBuf<uint32_t> perm;
  DevBuf<uint32_t> perm_group;
  DevBuf<unsigned long long> counters; /* cellsum[n_ref*T] | pairs[n_ref] | totals[2] | tally[14] */
  DevBuf<ppf_vote> votes;
  DevBuf<ppf_pose> raw_poses;
  DevBuf<ppf_pose> d_final;
  DevBuf<uint32_t> cl_u32;              /* order | assign | head | crank | n_out */
  DevBuf<unsigned long long> cl_votes;
  DevBuf<double> cl_soa;
  DevBuf<unsigned long long> cl_bits;   /* match matrix rows | head mask | (u32) head prefix */
  std::vector<ppf_pose> final_poses;
  bool clustered = false;
  ppf_match_stats stats{};
  ppf_model* model = nullptr;           /* retained while the workspace may still read it */
  bool model_owns_me = false;           /* a context workspace of model->ctx_idle: lives inside its model, holds no reference to it */
  ppf_match_params params{};
  int n_ref = 0, n_ref_total = 0, rows = 0;
  hipStream_t stream = nullptr;
  bool timing = false;
  hipEvent_t ev[2] = {nullptr, nullptr}; /* first kernel start, last kernel end */
  std::vector<hipEvent_t> batch_ev;      /* 4 per batch: k_pairs start / end, k_vote start / end */
  size_t ev_base = 0;                    /* first event of the current call in batch_ev (a batch context keeps one set per match of a run) */
  int n_batches = 0;
  bool pending = false;
  bool checked = false;                  /* the overflow flag of the pending call has been read */
  bool pools_failed = false;             /* a call ran out of hit pools at their worst-case size: the context is dropped, not kept warm */
  bool has_edge = false;
  double hit_frac = 0.25;                /* expected hits per scene pair: sizes the hit pools, learned from every call */
  bool frac_known = false;               /* false: the next call first COUNTS its hits (one extra pair pass and one wait) */
  double run_frac = 0.4;                 /* expected runs (distinct buckets hit by a reference point) per hit, learned likewise */
  double tbl_frac = TBL_FRAC_START;      /* expected count tables per hit, learned likewise (at most TBL_FRAC_MAX) */
  struct Learned { uint64_t model_serial; double hit, run, tbl; };
  std::vector<Learned> frac_by_model;    /* the three fractions remembered per model, at most 16 (batches alternate models): workspace_learned() */
  int batch_refs_cap = 0;                /* 0 = what the scratch budget holds; tests lower it to force several batches per call */
  int run_seg_cap = 0;                   /* 0 = what the LDS holds (vote_run_seg); tests lower it to force several staging segments per reference point */
  int round_buckets_cap = 0;             /* 0 = GROUP_MAX_BUCKETS; tests lower it to force several k_group rounds */
  bool acc32 = false;                    /* a call on this model cast more than PPF_ACC32_SWITCH of its votes twice (16-bit cells overflowed): 32-bit cells until the model changes */
  bool force_acc32 = false;              /* PPF_OPT_ACC32 = 1: 32-bit cells for every (reference point, tile) */
  int acc32_policy = 0;                  /* PPF_OPT_ACC32: 0 the switch above, 2 never switch (16-bit cells first, always), 3 a per-item limit (heavy_votes) */
  unsigned long long heavy_votes = ~0ull; /* policy 3: a (reference point, tile) that will cast at least this many votes goes straight to 32-bit cells:
                                            learned per model from what the previous call's items of each size needed (workspace_finish) */
  DevBuf<unsigned long long> item_votes, need_hist;
  bool cluster_serial = false;           /* force the serial greedy assignment (otherwise only used above 11,520 poses) */
  int device = -1;
  uint32_t* acc_dump = nullptr; /* set by ppf_debug_accumulators for one call */
  /* pinned: what the host reads of a finished call before anything else -- the 16 totals, the pools' overflow word, the number of
   * clustered poses -- written there by the call's last kernel (k_summary), so that the wait for the stream is the only round
   * trip (four synchronous 4- to 128-byte copies were 40 us of a 0.5 ms match) */
  unsigned long long* h_sum = nullptr;
  bool sum_valid = false; /* the pending call ended with k_summary */
  ~ppf_workspace();
};

namespace {

ppf_status enqueue_cluster(ppf_workspace* ws, const ppf_pose* d_in, int n, int num_poses, double pos, double rot,
                           bool weighted, hipStream_t st, bool rot_relative = false);

void resolve_thresholds(const ppf_model* m, const ppf_match_params* p, double* pos, double* rot) {
  *pos = p->position_threshold < 0 ? m->info.position_threshold_default : p->position_threshold;
  *rot = p->rotation_threshold < 0 ? m->info.rotation_threshold_default : p->rotation_threshold;
}

ppf_status enqueue_cluster(ppf_workspace* ws, const ppf_pose* d_in, int n, int num_poses, double pos, double rot,
                           bool weighted, hipStream_t st, bool rot_relative) {
  const size_t nn = (size_t)std::max(n, 1);
  HIPCHK(ws->d_final.reserve(nn));
  HIPCHK(ws->cl_u32.reserve(8 * nn + 4)); /* n_out | order | assign | rin | head | crank | gvotes | sizes | coff[n+1] */
  HIPCHK(ws->cl_votes.reserve(2 * nn));   /* cluster votes | pose vote keys */
  HIPCHK(ws->cl_soa.reserve(15 * nn));    /* member q,t 7n | heads 4n (+ 4n quaternions for the relative rotation metric) */
  ClusterArgs ca;
  ca.in = d_in; ca.n = n; ca.num_poses = num_poses; ca.pos_thr = pos; ca.rot_thr = rot; ca.weighted = weighted ? 1 : 0;
  ca.rot_relative = rot_relative ? 1 : 0;
  ca.cos_half_rot = ppf_cos(0.5 * rot);
  uint32_t* u = ws->cl_u32.p;
  ca.n_out = u; u += 1;
  uint32_t* order = u; u += n;
  ca.order = order; ca.assign = u; u += n; ca.head = u; u += n; ca.crank = u; u += n;
  ca.gvotes = u; u += n; ca.g_sizes = u; u += n; ca.coff = u;
  ca.cvotes = ws->cl_votes.p;
  unsigned long long* vkeys = ws->cl_votes.p + nn;
  ca.gq = ws->cl_soa.p; ca.g_heads = ca.gq + 7 * nn;
  ca.out = ws->d_final.p;
  HIPCHK(hipMemsetAsync(ca.n_out, 0, sizeof(uint32_t), st));
  if (n > 0) {
    static std::once_flag once_c;
    static hipError_t attr_c = hipSuccess;
    std::call_once(once_c, [] {
      attr_c = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cluster_assign<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 116 * 1024);
    });
    HIPCHK(attr_c);
    const unsigned nb = (unsigned)((n + 255) / 256);
    k_vote_keys<<<dim3(nb), dim3(256), 0, st>>>(d_in, n, vkeys);
    k_rank<<<dim3((unsigned)((n + RANK_KEYS - 1) / RANK_KEYS)), dim3(256), 0, st>>>(vkeys, n, nullptr, order, nullptr);            /* (votes desc, index asc) */
    const int np = std::min(num_poses, n);
    const int words = (np + 63) / 64;
    const size_t matrix_words = (size_t)np * words;
    if (rot_relative && np > 0 && words > CLM_MAX_WORDS)
      return fail(PPF_ERR_INVALID, "clustering: the relative rotation metric handles up to %d poses", CLM_MAX_WORDS * 64);
    if (np > 0 && words <= CLM_MAX_WORDS && (!ws->cluster_serial || rot_relative)) {
      /* match matrix + one wave walking the rows (see k_clm_heads) */
      static std::once_flag once_h;
      static hipError_t attr_h = hipSuccess;
      std::call_once(once_h, [] {
        attr_h = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_clm_heads), hipFuncAttributeMaxDynamicSharedMemorySize, CLM_LDS_BYTES);
      });
      HIPCHK(attr_h);
      HIPCHK(ws->cl_bits.reserve(matrix_words + 2 * (size_t)words + 2));
      unsigned long long* bits = ws->cl_bits.p;
      unsigned long long* heads = bits + matrix_words;
      uint32_t* prefix = reinterpret_cast<uint32_t*>(heads + words);
      const int pitch = words | 1; /* odd pitch (in 8-byte words): lanes reading the same word of different rows spread over the banks */
      const int rows_per_round = std::max(64, std::min(512, (int)((CLM_LDS_BYTES - (size_t)words * 8) / ((size_t)pitch * 8)) / 64 * 64));
      k_clm_gather<<<dim3(nb), dim3(256), 0, st>>>(ca);
      k_clm_matrix<<<dim3((unsigned)np), dim3(256), 0, st>>>(ca, bits, words);
      k_clm_heads<<<dim3(1), dim3(1024), ((size_t)rows_per_round * pitch + words) * 8, st>>>(ca, bits, words, pitch, rows_per_round, heads, prefix);
      k_clm_assign<<<dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st>>>(ca, bits, words, heads, prefix);
    } else if (n <= CLUSTER_LDS_MAX) {
      k_cluster_assign<true><<<dim3(1), dim3(1024), (size_t)n * 32 + 64, st>>>(ca);
    } else {
      k_cluster_assign<false><<<dim3(1), dim3(1024), 0, st>>>(ca);
    }
    k_cluster_sizes<<<dim3(nb), dim3(256), 0, st>>>(ca);
    k_cluster_offsets<<<dim3(1), dim3(1024), 0, st>>>(ca);
    k_cluster_members<<<dim3(nb), dim3(256), 0, st>>>(ca);
    k_rank<<<dim3((unsigned)((n + RANK_KEYS - 1) / RANK_KEYS)), dim3(256), 0, st>>>(ca.cvotes, 0, ca.n_out, nullptr, ca.crank);    /* (cluster votes desc, creation asc) */
    k_cluster_finish<<<dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st>>>(ca);
    HIPCHK(hipGetLastError());
  }
  return PPF_OK;
}

}  // namespace

#endif /* PPF_HOST_COMMON_H */
