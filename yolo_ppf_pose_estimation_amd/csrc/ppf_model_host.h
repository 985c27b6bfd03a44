/*
 * ppf_model_host.h — C-ABI, model side: defaults, ppf_model_train (table build), handles, table dump, the versioned model file.
 * Reference call sites: /root/reference/include/CloudProcessing.h:205-261 (ctor, trainModel, write), :106-121 (read).
 */
#ifndef PPF_MODEL_HOST_H
#define PPF_MODEL_HOST_H

/* ============================================================================================ */
/* C-ABI                                                                                          */
/* ============================================================================================ */
extern "C" {

void ppf_default_train_params(ppf_train_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->relative_sampling_step = 0.05;
  p->relative_distance_step = 0.05;
  p->num_angles = 30;
}
void ppf_default_match_params(ppf_match_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->relative_scene_sample_step = 1.0 / 5.0;
  p->relative_scene_distance = 0.03;
  p->position_threshold = -1;
  p->rotation_threshold = -1;
  p->ref_stride = 1;
}
int ppf_abi_version(void) { return PPF_ABI_VERSION; }
int ppf_last_error(char* buf, int cap) {
  if (buf && cap > 0) {
    strncpy(buf, g_last_error.c_str(), (size_t)cap - 1);
    buf[cap - 1] = 0;
  }
  return (int)g_last_error.size();
}
int ppf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

/* ---- model ------------------------------------------------------------------------------------ */
/* model reference points whose accumulator rows fit one k_vote workgroup's LDS next to its fixed part */
static int max_tile_rows(int num_angles) {
  const long budget = (long)LDS_BYTES - (long)VOTE_LDS_FIXED - 4L * (vote_guard(num_angles) + 1);
  return budget <= 0 ? 0 : (int)(budget / (4L * vote_pitch(num_angles)));
}

/* tabulate hash -> bucket for every key with angle bins 0..floor(pi/angle_step)+1 and distance bins 0..1023 (pairs up
 * to ~1000 distance steps apart: tens of model diameters); everything else keeps the hash path in k_pairs.
 * PPF_FEATURE_DARBOUX: the angle key spans -pi..pi and the two cosine keys -1..1, all divided by the angle step and floored. */
static void key_lut_dims(ppf_model* m) {
  KeyDims& d = m->kd;
  if (m->params.feature == PPF_FEATURE_DARBOUX) {
    d.o0 = (int)std::floor(PPF_PI / m->info.angle_step) + 2;
    d.o1 = d.o2 = (int)std::floor(1.0 / m->info.angle_step) + 2;
    d.n0 = 2 * d.o0 + 1;
    d.n1 = d.n2 = 2 * d.o1 + 1;
  } else {
    d.o0 = d.o1 = d.o2 = 0;
    d.n0 = d.n1 = d.n2 = (int)std::floor(PPF_PI / m->info.angle_step) + 2;
  }
  d.nd = 1024;
  const size_t per_dist = (size_t)d.n0 * d.n1 * d.n2;
  if (per_dist * d.nd > ((size_t)1 << 26)) /* very fine angle steps: shrink the distance range to keep the table at 256 MiB */
    d.nd = (int)std::max<size_t>(1, ((size_t)1 << 26) / per_dist);
}
/* hash slots of the table: next_pow2(N^2) like the reference's library, or (PPF_KEY_EXACT) one slot per quantised key */
static uint32_t table_slots(const ppf_model* m) {
  if (m->params.key_equality == PPF_KEY_EXACT) return next_pow2(std::max<uint32_t>((uint32_t)key_table_size(m->kd), 16u));
  return next_pow2(std::max<uint32_t>((uint32_t)((size_t)m->info.n_ref * m->info.n_ref), 16u));
}
static ppf_status build_key_lut(ppf_model* m, hipStream_t st) {
  key_lut_dims(m);
  const size_t n = key_table_size(m->kd);
  HIPCHK(m->key_lut.reserve(n));
  k_build_key_lut<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(m->slotmap.p, m->info.slots - 1, m->params.key_equality == PPF_KEY_EXACT,
                                                                           m->kd, m->key_lut.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  return PPF_OK;
}

static ppf_status build_table(ppf_model* m, hipStream_t st) {
  const int N = m->info.n_ref;
  const size_t NN = (size_t)N * N;
  const uint32_t slots = m->info.slots;
  const size_t words = (slots + 63) / 64;
  DevBuf<uint32_t> pair_slot, word_cnt, word_rank, counts, offsets;
  DevBuf<float> pair_alpha;
  DevBuf<unsigned long long> bits;
  HIPCHK(pair_slot.reserve(NN));
  HIPCHK(pair_alpha.reserve(NN));
  HIPCHK(bits.reserve(words));
  HIPCHK(hipMemsetAsync(bits.p, 0, words * sizeof(unsigned long long), st));
  k_train_pairs<<<dim3(N), dim3(256), 0, st>>>(m->cloud.view(), m->info.angle_step, m->info.distance_step, slots - 1,
                                               m->params.key_equality == PPF_KEY_EXACT, m->params.feature == PPF_FEATURE_DARBOUX, m->kd,
                                               pair_slot.p, pair_alpha.p, bits.p);
  HIPCHK(hipGetLastError());
  HIPCHK(word_cnt.reserve(words + 1));
  HIPCHK(word_rank.reserve(words + 1));
  HIPCHK(hipMemsetAsync(word_cnt.p, 0, (words + 1) * sizeof(uint32_t), st));
  k_popcount_words<<<dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st>>>(bits.p, word_cnt.p, words);
  HIPCHK(hipGetLastError());
  ppf_status s = device_exclusive_scan(word_cnt.p, word_rank.p, words + 1, st);
  if (s != PPF_OK) return s;
  uint32_t n_buckets = 0;
  HIPCHK(hipMemcpyAsync(&n_buckets, word_rank.p + words, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  m->info.n_buckets = n_buckets;
  HIPCHK(m->slotmap.reserve(words));
  k_pack_slotmap<<<dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st>>>(bits.p, word_rank.p, m->slotmap.p, words);
  HIPCHK(hipGetLastError());

  const int T = m->info.n_tiles;
  const size_t ncnt = (size_t)T * n_buckets + 1;
  DevBuf<uint32_t> rec_cnt;
  HIPCHK(counts.reserve(ncnt));
  HIPCHK(rec_cnt.reserve(ncnt));
  HIPCHK(offsets.reserve(ncnt));
  HIPCHK(m->bucket_slot.reserve(std::max<uint32_t>(n_buckets, 1)));
  HIPCHK(hipMemsetAsync(counts.p, 0, ncnt * sizeof(uint32_t), st));
  /* Dealing order inside a bucket (see "table layout" above): two levels (the accumulator-word half of the entry's row),
   * a level cut into up to gmax cell groups, inside a group every bank's entries in phase order, spread evenly over it. */
  const int levels = 2;
  uint32_t gmax = (uint32_t)AGG_Q; /* one group per count-table cell at most; fewer groups when the class keys would not fit 32 bits or the group counters would be unreasonably big */
  while (gmax > 1 && (ncnt * (size_t)levels * gmax * DEAL_BANKS >= 0xFFFFFFF0ull || ncnt * (size_t)levels * gmax * sizeof(uint32_t) > ((size_t)1 << 30)))
    gmax >>= 1;
  if (ncnt * (size_t)levels * gmax * DEAL_BANKS >= 0xFFFFFFF0ull)
    return fail(PPF_ERR_INVALID, "ppf_model_train: %u buckets x %d tiles are more than the table build can key", n_buckets, T);
  DevBuf<uint32_t> level_cnt, mirror_cur, group_cnt;
  HIPCHK(level_cnt.reserve(ncnt * levels));
  HIPCHK(mirror_cur.reserve(ncnt));
  HIPCHK(group_cnt.reserve(ncnt * levels * gmax));
  HIPCHK(hipMemsetAsync(level_cnt.p, 0, ncnt * levels * sizeof(uint32_t), st));
  HIPCHK(hipMemsetAsync(mirror_cur.p, 0, ncnt * sizeof(uint32_t), st));
  HIPCHK(hipMemsetAsync(group_cnt.p, 0, ncnt * levels * gmax * sizeof(uint32_t), st));
  const unsigned nblk = (unsigned)((NN + 255) / 256);
  k_train_bin<<<dim3(nblk), dim3(256), 0, st>>>(pair_slot.p, pair_alpha.p, N, m->slotmap.p, (int)n_buckets,
                                                m->info.tile_refs, T, m->info.num_angles, levels, counts.p, nullptr,
                                                level_cnt.p, mirror_cur.p, nullptr, gmax, nullptr, m->bucket_slot.p, 0);
  HIPCHK(hipGetLastError());
  /* real entries (N(N-1) + mirrored spill entries) */
  s = device_exclusive_scan(counts.p, offsets.p, ncnt, st);
  if (s != PPF_OK) return s;
  uint32_t n_entries = 0;
  HIPCHK(hipMemcpyAsync(&n_entries, offsets.p + (ncnt - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  /* record offsets */
  k_record_counts<<<dim3((unsigned)((ncnt + 255) / 256)), dim3(256), 0, st>>>(counts.p, rec_cnt.p, ncnt - 1);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemsetAsync(rec_cnt.p + (ncnt - 1), 0, sizeof(uint32_t), st));
  s = device_exclusive_scan(rec_cnt.p, offsets.p, ncnt, st);
  if (s != PPF_OK) return s;
  uint32_t n_records = 0;
  HIPCHK(hipMemcpyAsync(&n_records, offsets.p + (ncnt - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  m->info.n_entries = n_entries;
  m->n_records = n_records;
  /* per-tile CSR rows of n_buckets+1 RECORD offsets: row t = offsets[t*NB .. t*NB+NB] (the next row's first
   * element closes the last bucket), materialised with an explicit copy per tile */
  HIPCHK(m->bucket_off.reserve((size_t)T * (n_buckets + 1)));
  for (int t = 0; t < T; t++)
    HIPCHK(hipMemcpyAsync(m->bucket_off.p + (size_t)t * (n_buckets + 1), offsets.p + (size_t)t * n_buckets,
                          (size_t)(n_buckets + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
  HIPCHK(m->records.reserve(std::max<uint32_t>(n_records, 1)));
  if (n_records) {
    k_record_init<<<dim3((n_records + 255) / 256), dim3(256), 0, st>>>(m->records.p, n_records, m->info.num_angles);
    HIPCHK(hipGetLastError());
  }
  DevBuf<uint32_t> pair_pos; /* dealing position of every pair inside its (tile, bucket, level, cell group) */
  {
    DevBuf<uint32_t> kcls, kph, v1, kt, v2, starts, pair_rank, pair_n;
    HIPCHK(kcls.reserve(NN)); HIPCHK(kph.reserve(NN)); HIPCHK(v1.reserve(NN)); HIPCHK(kt.reserve(NN)); HIPCHK(v2.reserve(NN));
    HIPCHK(pair_rank.reserve(NN)); HIPCHK(pair_n.reserve(NN)); HIPCHK(pair_pos.reserve(NN));
    const uint32_t invalid = (uint32_t)((size_t)T * n_buckets * 2 * gmax * DEAL_BANKS);
    k_train_keys<<<dim3(nblk), dim3(256), 0, st>>>(pair_slot.p, pair_alpha.p, N, m->slotmap.p, (int)n_buckets, m->info.tile_refs,
                                                   m->info.num_angles, level_cnt.p, gmax, group_cnt.p, invalid, kcls.p, kph.p, v1.p);
    HIPCHK(hipGetLastError());
    /* (1) rank of every pair by phase inside its (tile, bucket, level, bank): two stable sorts (phase, then group) */
    uint32_t* va = nullptr;
    uint32_t n_runs = 0;
    s = sort_segments(kph, v1, kt, v2, (int)NN, 65535ull, starts, &va, &n_runs, st);
    if (s != PPF_OK) return s;
    k_gather_u32<<<dim3(nblk), dim3(256), 0, st>>>(kcls.p, va, NN, kph.p); /* group keys in phase order */
    HIPCHK(hipGetLastError());
    {
      DevBuf<uint32_t>& vin = va == v1.p ? v1 : v2;
      DevBuf<uint32_t>& vtmp = va == v1.p ? v2 : v1;
      s = sort_segments(kph, vin, kt, vtmp, (int)NN, (unsigned long long)invalid, starts, &va, &n_runs, st);
      if (s != PPF_OK) return s;
    }
    k_train_ranks<<<dim3(nblk), dim3(256), 0, st>>>(va, starts.p, n_runs, NN, pair_rank.p, pair_n.p);
    HIPCHK(hipGetLastError());
    /* (2) position inside the cell group: the banks' entries spread evenly, i.e. sorted by (rank + 1/2) / class size; ties keep
     * the pair order (stable sorts from the identity) */
    k_train_spread<<<dim3(nblk), dim3(256), 0, st>>>(kcls.p, pair_rank.p, pair_n.p, invalid, NN, kph.p, kt.p, v1.p);
    HIPCHK(hipGetLastError());
    {
      DevBuf<uint32_t> kseg;
      HIPCHK(kseg.reserve(NN));
      HIPCHK(hipMemcpyAsync(kseg.p, kt.p, NN * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
      s = sort_segments(kph, v1, kt, v2, (int)NN, 0xFFFFFFFFull, starts, &va, &n_runs, st);
      if (s != PPF_OK) return s;
      k_gather_u32<<<dim3(nblk), dim3(256), 0, st>>>(kseg.p, va, NN, kph.p); /* level keys in fraction order */
      HIPCHK(hipGetLastError());
      DevBuf<uint32_t>& vin = va == v1.p ? v1 : v2;
      DevBuf<uint32_t>& vtmp = va == v1.p ? v2 : v1;
      s = sort_segments(kph, vin, kt, vtmp, (int)NN, (unsigned long long)(invalid / DEAL_BANKS), starts, &va, &n_runs, st);
      if (s != PPF_OK) return s;
      k_train_ranks<<<dim3(nblk), dim3(256), 0, st>>>(va, starts.p, n_runs, NN, pair_pos.p);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(st));
    }
    HIPCHK(hipStreamSynchronize(st)); /* the sort scratch dies here */
  }
  k_train_bin<<<dim3(nblk), dim3(256), 0, st>>>(pair_slot.p, pair_alpha.p, N, m->slotmap.p, (int)n_buckets,
                                                m->info.tile_refs, T, m->info.num_angles, levels, counts.p, offsets.p,
                                                level_cnt.p, mirror_cur.p, group_cnt.p, gmax, m->records.p, nullptr, 1, pair_pos.p);
  HIPCHK(hipGetLastError());
  HIPCHK(m->bucket_total.reserve(std::max<uint32_t>(n_buckets, 1)));
  if (n_buckets) {
    k_bucket_total<<<dim3((n_buckets + 255) / 256), dim3(256), 0, st>>>(m->bucket_off.p, (int)n_buckets, T, m->bucket_total.p);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(m->bucket_mid.reserve(std::max<size_t>((size_t)n_buckets * T, 1)));
  if (n_buckets) {
    k_bucket_mid<<<dim3((unsigned)(((size_t)n_buckets * T + 255) / 256)), dim3(256), 0, st>>>(
        m->bucket_off.p, (int)n_buckets, T, m->records.p, (uint32_t)((vote_guard(m->info.num_angles) - m->info.num_angles) * 4), m->bucket_mid.p);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(st));
  ppf_status sl = build_key_lut(m, st);
  if (sl != PPF_OK) return sl;
  m->info.device_bytes = m->cloud.buf.bytes() + m->slotmap.bytes() + m->bucket_off.bytes() + m->bucket_slot.bytes() +
                         m->records.bytes() + m->key_lut.bytes();
  return PPF_OK;
}

ppf_status ppf_model_train(const float* xyzn, int n, int stride, int noff, const ppf_train_params* params, ppf_model** out) {
  if (!out) return fail(PPF_ERR_INVALID, "ppf_model_train: out is NULL");
  *out = nullptr;
  if (!xyzn || n <= 1 || bad_layout(stride, noff) || !params) return fail(PPF_ERR_INVALID, "ppf_model_train: bad argument");
  if (!(params->relative_sampling_step > 0) || !(params->num_angles >= 1))
    return fail(PPF_ERR_INVALID, "ppf_model_train: bad parameters");
  if (!have_device()) return fail(PPF_ERR_HIP, "ppf_model_train: no HIP device (this engine has no CPU fallback)");
  std::unique_ptr<ppf_model> owner(new ppf_model()); /* released on every early return */
  ppf_model* m = owner.get();
  m->params = *params;
  HIPCHK(hipGetDevice(&m->device));
  /* ctor + setSearchParams defaults of the reference's detector */
  const double angle_step = (360.0 / params->num_angles) * PPF_PI / 180.0;
  float lo[3], hi[3];
  bbox_host(xyzn, n, stride, lo, hi);
  const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  const float diameter = std::sqrt(dx * dx + dy * dy + dz * dz);
  const float dist_step = (float)(diameter * (params->distance_from_distance_step ? params->relative_distance_step
                                                                                  : params->relative_sampling_step));
  if (params->presampled) {
    m->sampled.resize((size_t)n * 6);
    for (int i = 0; i < n; i++) {
      memcpy(&m->sampled[(size_t)i * 6], xyzn + (size_t)i * stride, 12);
      memcpy(&m->sampled[(size_t)i * 6 + 3], xyzn + (size_t)i * stride + noff, 12);
    }
  }
  hipStream_t st = nullptr;
  {
    ppf_status s0;
    if (params->presampled) {
      s0 = m->cloud.load_host(m->sampled.data(), n, st);
    } else { /* A2 on the device: upload the raw model cloud, sample, keep a host copy of the sampled rows */
      DevBuf<float> d_raw;
      hipError_t e = d_raw.reserve((size_t)n * stride);
      if (e == hipSuccess) e = hipMemcpy(d_raw.p, xyzn, (size_t)n * stride * sizeof(float), hipMemcpyHostToDevice);
      s0 = e == hipSuccess ? device_sample_cloud(d_raw.p, n, stride, noff, (float)params->relative_sampling_step, m->cloud, &m->sampled, st)
                           : fail(PPF_ERR_HIP, "ppf_model_train: upload failed: %s", hipGetErrorString(e));
    }
    if (s0 != PPF_OK) return s0;
  }
  const int N = (int)(m->sampled.size() / 6);
  if (N < 2 || (uint64_t)N * N > 0x7FFFFFFFull) {
    return fail(PPF_ERR_INVALID, "ppf_model_train: %d sampled model points unsupported", N);
  }
  m->info.n_ref = N;
  m->info.num_angles = (int)std::floor(2 * PPF_PI / angle_step);
  m->info.angle_step = angle_step;
  m->info.distance_step = dist_step;
  m->info.diameter = diameter;
  key_lut_dims(m);
  if (key_table_size(m->kd) > ((size_t)1 << 30)) /* k_pairs indexes the key table with 32-bit arithmetic (and 4 GiB of keys would be pointless) */
    return fail(PPF_ERR_INVALID, "ppf_model_train: num_angles %g is too fine for the key table", params->num_angles);
  if (params->key_equality != PPF_KEY_BUCKET && params->key_equality != PPF_KEY_EXACT)
    return fail(PPF_ERR_INVALID, "ppf_model_train: key_equality must be PPF_KEY_BUCKET or PPF_KEY_EXACT");
  if (params->feature != PPF_FEATURE_PPF && params->feature != PPF_FEATURE_DARBOUX)
    return fail(PPF_ERR_INVALID, "ppf_model_train: feature must be PPF_FEATURE_PPF or PPF_FEATURE_DARBOUX");
  if (params->key_equality == PPF_KEY_EXACT && (double)diameter / (double)dist_step + 2.0 > (double)m->kd.nd)
    return fail(PPF_ERR_INVALID, "ppf_model_train: PPF_KEY_EXACT needs diameter / distance step (%g) below %d", (double)diameter / dist_step, m->kd.nd);
  m->info.slots = table_slots(m);
  m->info.position_threshold_default = params->relative_sampling_step;
  m->info.rotation_threshold_default = ((360 / angle_step) / 180.0 * PPF_PI);
  const int A = m->info.num_angles;
  int max_refs = 2 * max_tile_rows(A); /* 16-bit cells: two rows per accumulator word */
  if (params->max_tile_refs > 0) max_refs = std::min(max_refs, params->max_tile_refs);
  if (max_refs < 1) {
    return fail(PPF_ERR_INVALID, "ppf_model_train: num_angles %d too large for the LDS accumulator", A);
  }
  m->info.n_tiles = (N + max_refs - 1) / max_refs;
  m->info.tile_refs = (N + m->info.n_tiles - 1) / m->info.n_tiles;
  ppf_status s = build_table(m, st);
  if (s != PPF_OK) {
    return s;
  }
  *out = owner.release();
  return PPF_OK;
}

ppf_status ppf_pair_features(const float* xyzn, int n, int stride, int noff, int feature, float* out, size_t cap_rows) {
  if (!xyzn || n <= 0 || bad_layout(stride, noff) || !out || (feature != PPF_FEATURE_PPF && feature != PPF_FEATURE_DARBOUX))
    return fail(PPF_ERR_INVALID, "ppf_pair_features: bad argument");
  const size_t rows = (size_t)n * (size_t)n;
  if (cap_rows < rows) return fail(PPF_ERR_CAPACITY, "ppf_pair_features: need room for %zu rows", rows);
  if (!have_device()) return fail(PPF_ERR_HIP, "ppf_pair_features: no HIP device (this engine has no CPU fallback)");
  DevBuf<float> d_raw, d_out;
  CloudDev cloud;
  HIPCHK(d_raw.reserve((size_t)n * stride));
  HIPCHK(hipMemcpy(d_raw.p, xyzn, (size_t)n * stride * sizeof(float), hipMemcpyHostToDevice));
  ppf_status s = cloud.load_device(d_raw.p, n, stride, noff, nullptr);
  if (s != PPF_OK) return s;
  HIPCHK(d_out.reserve(rows * 5));
  k_pair_features<<<dim3((unsigned)n), dim3(256)>>>(cloud.view(), feature == PPF_FEATURE_DARBOUX ? 1 : 0, d_out.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, d_out.p, rows * 5 * sizeof(float), hipMemcpyDeviceToHost));
  return PPF_OK;
}

ppf_status ppf_model_nearest_pairs(const ppf_model* m, const float* f4, uint32_t* pairs_ij, int cap_pairs, int* n_out) {
  if (!m || !f4 || !n_out || cap_pairs < 0 || (cap_pairs > 0 && !pairs_ij)) return fail(PPF_ERR_INVALID, "ppf_model_nearest_pairs: bad argument");
  if (!have_device()) return fail(PPF_ERR_HIP, "ppf_model_nearest_pairs: no HIP device (this engine has no CPU fallback)");
  const bool darboux = m->params.feature == PPF_FEATURE_DARBOUX;
  const double as = m->info.angle_step, ds = m->info.distance_step;
  int32_t k[4];
  for (int c = 0; c < 4; c++) { /* the key of the query: the quantisation the table's pairs went through */
    const double q = (double)f4[c] / (c < 3 ? as : ds);
    k[c] = darboux ? ppf_floor_key(q) : ppf_d2i(q);
  }
  const int n = m->info.n_ref;
  DevBuf<uint2> d_out;
  DevBuf<uint32_t> d_cur;
  const uint32_t cap = (uint32_t)std::min<uint64_t>((uint64_t)n * (uint64_t)n, 1u << 26); /* 512 MB at most */
  HIPCHK(d_out.reserve(std::max<uint32_t>(cap, 1u)));
  HIPCHK(d_cur.reserve(1));
  HIPCHK(hipMemset(d_cur.p, 0, sizeof(uint32_t)));
  k_key_pairs<<<dim3((unsigned)n), dim3(256)>>>(m->cloud.view(), darboux ? 1 : 0, as, ds, k[0], k[1], k[2], k[3], d_out.p, cap, d_cur.p);
  HIPCHK(hipGetLastError());
  uint32_t found = 0;
  HIPCHK(hipMemcpy(&found, d_cur.p, sizeof(found), hipMemcpyDeviceToHost));
  if (found > cap) return fail(PPF_ERR_CAPACITY, "ppf_model_nearest_pairs: %u pairs share this key, more than one call returns", found);
  *n_out = (int)found;
  if ((int)found > cap_pairs) return cap_pairs == 0 ? PPF_OK : fail(PPF_ERR_CAPACITY, "ppf_model_nearest_pairs: need room for %u pairs", found);
  std::vector<uint2> h(found);
  if (found) HIPCHK(hipMemcpy(h.data(), d_out.p, (size_t)found * sizeof(uint2), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end(), [](const uint2& a, const uint2& b) { return a.x != b.x ? a.x < b.x : a.y < b.y; });
  for (uint32_t q = 0; q < found; q++) { pairs_ij[2 * q] = h[q].x; pairs_ij[2 * q + 1] = h[q].y; }
  return PPF_OK;
}

ppf_status ppf_model_retain(ppf_model* m) {
  if (!m) return fail(PPF_ERR_INVALID, "ppf_model_retain: NULL");
  m->refcount.fetch_add(1);
  return PPF_OK;
}
ppf_status ppf_model_release(ppf_model* m) {
  if (!m) return PPF_OK;
  if (m->refcount.fetch_sub(1) == 1) {
    sync_device(m->device); /* the table returns to the block cache: no match may still be reading it */
    delete m;
  }
  return PPF_OK;
}
ppf_status ppf_model_get_info(const ppf_model* m, ppf_model_info* info) {
  if (!m || !info) return fail(PPF_ERR_INVALID, "ppf_model_get_info: NULL");
  *info = m->info;
  return PPF_OK;
}
ppf_status ppf_model_get_device(const ppf_model* m, int* device) {
  if (!m || !device) return fail(PPF_ERR_INVALID, "ppf_model_get_device: NULL");
  *device = m->device;
  return PPF_OK;
}
ppf_status ppf_model_get_sampled(const ppf_model* m, float* out, int cap_rows) {
  if (!m || !out) return fail(PPF_ERR_INVALID, "ppf_model_get_sampled: NULL");
  if (cap_rows < m->info.n_ref) return fail(PPF_ERR_CAPACITY, "ppf_model_get_sampled: need %d rows", m->info.n_ref);
  memcpy(out, m->sampled.data(), m->sampled.size() * sizeof(float));
  return PPF_OK;
}
ppf_status ppf_model_get_table(const ppf_model* m, uint32_t* bucket_slot, uint32_t* bucket_off, int32_t* entry_cell,
                               float* entry_alpha) {
  if (!m) return fail(PPF_ERR_INVALID, "ppf_model_get_table: NULL");
  const size_t nb = m->info.n_buckets, nr = m->n_records;
  const int T = m->info.n_tiles, A = m->info.num_angles;
  if (bucket_slot) HIPCHK(hipMemcpy(bucket_slot, m->bucket_slot.p, nb * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (!bucket_off && !entry_cell && !entry_alpha) return PPF_OK;
  /* decode the pair records: per (tile, bucket) the real entries in storage order; dummies (row in the first
   * guard words) are skipped; the CSR handed out counts ENTRIES */
  std::vector<uint32_t> roff((size_t)T * (nb + 1));
  std::vector<uint4> rec(nr);
  HIPCHK(hipMemcpy(roff.data(), m->bucket_off.p, roff.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (nr) HIPCHK(hipMemcpy(rec.data(), m->records.p, nr * sizeof(uint4), hipMemcpyDeviceToHost));
  const uint32_t first_real = (uint32_t)((vote_guard(A) - A) * 4);
  size_t k = 0;
  for (int t = 0; t < T; t++) {
    for (size_t b = 0; b < nb; b++) {
      if (bucket_off) bucket_off[(size_t)t * (nb + 1) + b] = (uint32_t)k;
      for (uint32_t r = roff[(size_t)t * (nb + 1) + b]; r < roff[(size_t)t * (nb + 1) + b + 1]; r++) {
        const uint32_t rows[2] = {rec[r].x & ROW_CODE_MASK, rec[r].y & ROW_CODE_MASK}, al[2] = {rec[r].z, rec[r].w};
        for (int sl = 0; sl < 2; sl++) {
          if (rows[sl] < first_real) continue;
          if (k >= m->info.n_entries) return fail(PPF_ERR_INVALID, "ppf_model_get_table: more entries than counted");
          if (entry_cell) { /* byte offset of the row -> reference layout local_ref*numAngles (mirrored spill entries: -numAngles) */
            const int32_t w = (int32_t)(rows[sl] / 4) - vote_guard(A);
            entry_cell[k] = w < 0 ? -A : (w / vote_pitch(A) + (int32_t)(rows[sl] & 1u) * vote_half_rows(m->info.tile_refs)) * A;
          }
          if (entry_alpha) memcpy(&entry_alpha[k], &al[sl], 4);
          k++;
        }
      }
    }
    if (bucket_off) bucket_off[(size_t)t * (nb + 1) + nb] = (uint32_t)k;
  }
  return PPF_OK;
}

/* ---- model (de)serialisation: versioned binary CSR (the reference's XML format is defined by a
 * private OpenCV patch and unknown) ------------------------------------------------- */
static const char PPF_MAGIC[8] = {'P', 'P', 'F', 'H', 'I', 'P', '0', '5'}; /* 02: pair-record table; 03: ppf_train_params.feature; 04: 64 count-table cells (7-bit cell field), cell-grouped dealing; 05: the build's count-table cell count in the header */
/* what the records' count-table bits were computed with: a file written by a build with another PPF_AGG_Q carries cells of
 * another width in its row codes and would be voted through the wrong cells; it is refused */
struct ModelBuildWord { uint32_t agg_q, reserved; };

/* the model as a byte stream (what a model file holds), written to an open FILE */
static ppf_status model_write(const ppf_model* m, FILE* f, const char* path) {
  const size_t words = ((size_t)m->info.slots + 63) / 64, nb = m->info.n_buckets, ne = m->n_records;
  std::vector<SlotWord> slotmap(words);
  std::vector<uint32_t> boff((size_t)m->info.n_tiles * (nb + 1)), bslot(nb);
  std::vector<uint4> ent(ne);
  HIPCHK(hipMemcpy(slotmap.data(), m->slotmap.p, words * sizeof(SlotWord), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(boff.data(), m->bucket_off.p, boff.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (nb) HIPCHK(hipMemcpy(bslot.data(), m->bucket_slot.p, nb * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (ne) HIPCHK(hipMemcpy(ent.data(), m->records.p, ne * sizeof(uint4), hipMemcpyDeviceToHost));
  bool ok = fwrite(PPF_MAGIC, 1, 8, f) == 8;
  const ModelBuildWord bw = {(uint32_t)AGG_Q, 0u};
  ok = ok && fwrite(&bw, sizeof(bw), 1, f) == 1;
  ok = ok && fwrite(&m->params, sizeof(m->params), 1, f) == 1;
  ok = ok && fwrite(&m->info, sizeof(m->info), 1, f) == 1;
  ok = ok && fwrite(&m->n_records, sizeof(m->n_records), 1, f) == 1;
  ok = ok && fwrite(m->sampled.data(), sizeof(float), m->sampled.size(), f) == m->sampled.size();
  ok = ok && fwrite(slotmap.data(), sizeof(SlotWord), words, f) == words;
  ok = ok && fwrite(boff.data(), sizeof(uint32_t), boff.size(), f) == boff.size();
  ok = ok && fwrite(bslot.data(), sizeof(uint32_t), nb, f) == nb;
  ok = ok && fwrite(ent.data(), sizeof(uint4), ne, f) == ne;
  if (!ok) return fail(PPF_ERR_IO, "ppf_model_save: short write to %s", path);
  return PPF_OK;
}

ppf_status ppf_model_save(const ppf_model* m, const char* path) {
  if (!m || !path) return fail(PPF_ERR_INVALID, "ppf_model_save: NULL");
  FILE* f = fopen(path, "wb");
  if (!f) return fail(PPF_ERR_IO, "ppf_model_save: cannot open %s", path);
  ppf_status s = model_write(m, f, path);
  if (fclose(f) != 0 && s == PPF_OK) s = fail(PPF_ERR_IO, "ppf_model_save: short write to %s", path);
  return s;
}

/* the same bytes into the caller's memory: no file in between (the FileStorage overloads of the C++ facade) */
ppf_status ppf_model_save_mem(const ppf_model* m, void* buf, size_t cap, size_t* size) {
  if (!m || !size) return fail(PPF_ERR_INVALID, "ppf_model_save_mem: NULL");
  char* mem = nullptr;
  size_t len = 0;
  FILE* f = open_memstream(&mem, &len);
  if (!f) return fail(PPF_ERR_NOMEM, "ppf_model_save_mem: out of memory");
  ppf_status s = model_write(m, f, "memory");
  if (fclose(f) != 0 && s == PPF_OK) s = fail(PPF_ERR_NOMEM, "ppf_model_save_mem: out of memory");
  if (s == PPF_OK) {
    *size = len;
    if (buf && cap >= len) memcpy(buf, mem, len);
    else if (buf) s = fail(PPF_ERR_CAPACITY, "ppf_model_save_mem: need %zu bytes", len);
  }
  free(mem);
  return s;
}

/* Everything read from the file is checked before it reaches a kernel: header fields against each other and against
 * the file size, the CSR rows, the record rows (LDS byte offsets k_vote adds to) and alphas, the slot map's ranks.  A file
 * that fails any check is PPF_ERR_IO; no exception leaves this function. */
static ppf_status model_load_stream(FILE* f, const char* path, ppf_model** out, bool check_only);

static ppf_status model_load_impl(const char* path, ppf_model** out, bool check_only) {
  FILE* f = fopen(path, "rb");
  if (!f) return fail(PPF_ERR_IO, "ppf_model_load: cannot open %s", path);
  struct Closer { FILE* f; ~Closer() { if (f) fclose(f); } } closer{f};
  return model_load_stream(f, path, out, check_only);
}

static ppf_status model_load_stream(FILE* f, const char* path, ppf_model** out, bool check_only) {
  auto bad = [&](const char* what) { return fail(PPF_ERR_IO, "ppf_model_load: %s is not a valid model file (%s)", path, what); };
  if (fseek(f, 0, SEEK_END) != 0) return bad("seek");
  const long fsize = ftell(f);
  if (fsize < 0 || fseek(f, 0, SEEK_SET) != 0) return bad("seek");
  char magic[8];
  std::unique_ptr<ppf_model> owner(new ppf_model()); /* released on every early return */
  ppf_model* m = owner.get();
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, PPF_MAGIC, 8) != 0) return bad("magic");
  ModelBuildWord bw;
  if (fread(&bw, sizeof(bw), 1, f) != 1) return bad("header");
  if (bw.agg_q != (uint32_t)AGG_Q || bw.reserved != 0u) return bad("written by a build with another count-table cell count");
  if (fread(&m->params, sizeof(m->params), 1, f) != 1 || fread(&m->info, sizeof(m->info), 1, f) != 1 ||
      fread(&m->n_records, sizeof(m->n_records), 1, f) != 1)
    return bad("header");
  const ppf_model_info& I = m->info;
  const uint64_t N = (uint64_t)(I.n_ref > 0 ? I.n_ref : 0);
  if (I.n_ref < 2 || N * N > 0x7FFFFFFFull) return bad("n_ref");
  if (!(I.num_angles >= 1 && I.num_angles <= 4096) || !(I.angle_step > 1e-4) || !(I.distance_step > 0) || !std::isfinite(I.diameter)) return bad("steps");
  if (m->params.key_equality != PPF_KEY_BUCKET && m->params.key_equality != PPF_KEY_EXACT) return bad("key_equality");
  if (m->params.feature != PPF_FEATURE_PPF && m->params.feature != PPF_FEATURE_DARBOUX) return bad("feature");
  key_lut_dims(m);
  if (key_table_size(m->kd) > ((size_t)1 << 30)) return bad("angle step too fine for the key table");
  if (I.slots != table_slots(m)) return bad("slots");
  if (I.num_angles != (int)std::floor(2 * PPF_PI / I.angle_step)) return bad("num_angles");
  const int A = I.num_angles, GW = vote_guard(A);
  if (I.n_tiles < 1 || I.tile_refs < 1 || (uint64_t)I.n_tiles * I.tile_refs < N || (uint64_t)(I.n_tiles - 1) * I.tile_refs >= N)
    return bad("tiles");
  if (I.tile_refs > 2 * max_tile_rows(A)) return bad("tile does not fit this build's LDS accumulator");
  if (I.n_buckets > I.slots || (uint64_t)I.n_buckets > N * N) return bad("n_buckets");
  if (I.n_entries > N * N + N) return bad("n_entries");
  const uint64_t nb = I.n_buckets, ne = m->n_records, T = (uint64_t)I.n_tiles;
  if (ne > I.n_entries / 2 + 32ull * T * nb + 64 || ne >= 0xFFFFFFFFull) return bad("n_records");
  const uint64_t words = ((uint64_t)I.slots + 63) / 64;
  const uint64_t expect = 8 + sizeof(ModelBuildWord) + sizeof(m->params) + sizeof(m->info) + sizeof(m->n_records) + N * 6 * sizeof(float) +
                          words * sizeof(SlotWord) + T * (nb + 1) * sizeof(uint32_t) + nb * sizeof(uint32_t) + ne * sizeof(uint4);
  if ((uint64_t)fsize != expect) return bad("file size does not match its header");
  std::vector<SlotWord> slotmap(words);
  std::vector<uint32_t> boff(T * (nb + 1)), bslot(nb);
  std::vector<uint4> ent(ne);
  m->sampled.resize(N * 6);
  bool ok = fread(m->sampled.data(), sizeof(float), m->sampled.size(), f) == m->sampled.size();
  ok = ok && fread(slotmap.data(), sizeof(SlotWord), words, f) == words;
  ok = ok && fread(boff.data(), sizeof(uint32_t), boff.size(), f) == boff.size();
  ok = ok && (nb == 0 || fread(bslot.data(), sizeof(uint32_t), nb, f) == nb);
  ok = ok && (ne == 0 || fread(ent.data(), sizeof(uint4), ne, f) == ne);
  if (!ok) return bad("short read");
  for (float v : m->sampled)
    if (!std::isfinite(v)) return bad("sampled cloud");
  { /* slot map: ranks are the running popcount, which ends at n_buckets */
    uint64_t run = 0;
    for (uint64_t w = 0; w < words; w++) {
      if (slotmap[w].rank != run) return bad("slot map ranks");
      run += (uint64_t)__builtin_popcount(slotmap[w].bits_lo) + (uint64_t)__builtin_popcount(slotmap[w].bits_hi);
    }
    if (run != nb) return bad("slot map population");
  }
  for (uint64_t k = 0; k < nb; k++)
    if (bslot[k] >= I.slots) return bad("bucket slots");
  { /* per-tile CSR rows over the records: monotone, chained tile to tile, ending at n_records */
    uint32_t prev = 0;
    for (uint64_t t = 0; t < T; t++) {
      const uint32_t* row = &boff[t * (nb + 1)];
      if (row[0] != prev) return bad("bucket offsets (tile start)");
      for (uint64_t k = 0; k < nb; k++)
        if (row[k + 1] < row[k]) return bad("bucket offsets (order)");
      prev = row[nb];
    }
    if (prev != ne) return bad("bucket offsets (total)");
  }
  { /* records: LDS byte offsets inside guard + the tile's word rows (a vote adds up to A*4 bytes) with the half of the
     * word in bit 0, finite alphas within (-pi, pi) */
    const uint32_t limit_words = (uint32_t)vote_lds_words(I.tile_refs, A);
    for (uint64_t k = 0; k < ne; k++) {
      const uint32_t codes[2] = {ent[k].x, ent[k].y}, al[2] = {ent[k].z, ent[k].w};
      for (int sl = 0; sl < 2; sl++) {
        const uint32_t row = codes[sl] & ROW_CODE_MASK, cx = (codes[sl] >> ROW_X_SHIFT) & 31u, cq = (codes[sl] >> ROW_Q_SHIFT) & ROW_Q_MASK;
        if ((row & 2u) || row / 4 + (uint32_t)A + 1 > limit_words) return bad("record row"); /* bin A of the last row: the word behind the rows */
        if ((codes[sl] >> 30) || cx > (uint32_t)A || cq > (uint32_t)AGG_Q) return bad("record cell");
        float av;
        memcpy(&av, &al[sl], 4);
        if (!(std::fabs(av) <= 3.1416f)) return bad("record alpha");
      }
    }
    /* every (tile, bucket) in dealing order (position j = record 32*(j/64) + j%32, slot (j%64)/32): entries of low-half
     * rows, entries of high-half rows, padding -- what k_bucket_mid and the 32-bit passes of k_vote rely on */
    const uint32_t first_real = (uint32_t)((GW - A) * 4);
    for (uint64_t t = 0; t < T; t++)
      for (uint64_t b = 0; b < nb; b++) {
        const uint32_t off = boff[t * (nb + 1) + b], cnt = boff[t * (nb + 1) + b + 1] - off;
        int state = 0; /* 0: low halves, 1: high halves, 2: padding */
        for (uint32_t j = 0; j < 64u * ((cnt + 31u) / 32u); j++) {
          const uint32_t r = 32u * (j / 64u) + (j % 32u);
          if (r >= cnt) continue;
          const uint32_t code = (((j % 64u) / 32u) ? ent[off + r].y : ent[off + r].x) & ROW_CODE_MASK;
          const int kind = code < first_real ? 2 : (int)(code & 1u);
          if (kind < state) return bad("record halves (order)");
          state = kind;
        }
      }
  }
  if (check_only) return PPF_OK;
  m->refcount = 1;
  HIPCHK(hipGetDevice(&m->device));
  ppf_status s = m->cloud.load_host(m->sampled.data(), I.n_ref, nullptr);
  auto up = [&](auto& dst, const auto& src) -> ppf_status {
    HIPCHK(dst.reserve(std::max<size_t>(src.size(), 1)));
    if (!src.empty()) HIPCHK(hipMemcpy(dst.p, src.data(), src.size() * sizeof(src[0]), hipMemcpyHostToDevice));
    return PPF_OK;
  };
  if (s == PPF_OK) s = up(m->slotmap, slotmap);
  if (s == PPF_OK) s = up(m->bucket_off, boff);
  if (s == PPF_OK) s = up(m->bucket_slot, bslot);
  if (s == PPF_OK) s = up(m->records, ent);
  if (s == PPF_OK) {
    hipError_t e = m->bucket_total.reserve(std::max<uint32_t>(I.n_buckets, 1));
    if (e != hipSuccess) s = fail(PPF_ERR_HIP, "ppf_model_load: %s", hipGetErrorString(e));
    else if (I.n_buckets) {
      k_bucket_total<<<dim3((I.n_buckets + 255) / 256), dim3(256)>>>(m->bucket_off.p, (int)I.n_buckets, I.n_tiles, m->bucket_total.p);
      if (hipDeviceSynchronize() != hipSuccess) s = fail(PPF_ERR_HIP, "ppf_model_load: bucket totals failed");
    }
  }
  if (s == PPF_OK) {
    hipError_t e = m->bucket_mid.reserve(std::max<size_t>((size_t)I.n_buckets * I.n_tiles, 1));
    if (e != hipSuccess) s = fail(PPF_ERR_HIP, "ppf_model_load: %s", hipGetErrorString(e));
    else if (I.n_buckets) {
      k_bucket_mid<<<dim3((unsigned)(((size_t)I.n_buckets * I.n_tiles + 255) / 256)), dim3(256)>>>(
          m->bucket_off.p, (int)I.n_buckets, I.n_tiles, m->records.p, (uint32_t)((GW - A) * 4), m->bucket_mid.p);
      if (hipDeviceSynchronize() != hipSuccess) s = fail(PPF_ERR_HIP, "ppf_model_load: bucket halves failed");
    }
  }
  if (s == PPF_OK) s = build_key_lut(m, nullptr); /* not stored in the file: rebuilt from the slot map */
  if (s != PPF_OK) return s;
  m->info.device_bytes = m->cloud.buf.bytes() + m->slotmap.bytes() + m->bucket_off.bytes() + m->bucket_slot.bytes() +
                         m->records.bytes() + m->key_lut.bytes();
  *out = owner.release();
  return PPF_OK;
}

ppf_status ppf_model_load(const char* path, ppf_model** out) {
  if (!path || !out) return fail(PPF_ERR_INVALID, "ppf_model_load: NULL");
  *out = nullptr;
  if (!have_device()) return fail(PPF_ERR_HIP, "ppf_model_load: no HIP device");
  try {
    return model_load_impl(path, out, false);
  } catch (const std::bad_alloc&) {
    return fail(PPF_ERR_NOMEM, "ppf_model_load: out of host memory reading %s", path);
  } catch (...) {
    return fail(PPF_ERR_IO, "ppf_model_load: %s is not a valid model file", path);
  }
}

ppf_status ppf_model_load_mem(const void* buf, size_t size, ppf_model** out) {
  if (!buf || !out) return fail(PPF_ERR_INVALID, "ppf_model_load_mem: NULL");
  *out = nullptr;
  if (!have_device()) return fail(PPF_ERR_HIP, "ppf_model_load_mem: no HIP device");
  if (size == 0) return fail(PPF_ERR_IO, "ppf_model_load: memory is not a valid model file (empty)");
  FILE* f = fmemopen(const_cast<void*>(buf), size, "rb");
  if (!f) return fail(PPF_ERR_NOMEM, "ppf_model_load_mem: out of memory");
  struct Closer { FILE* f; ~Closer() { if (f) fclose(f); } } closer{f};
  try {
    return model_load_stream(f, "memory", out, false);
  } catch (const std::bad_alloc&) {
    return fail(PPF_ERR_NOMEM, "ppf_model_load_mem: out of host memory");
  } catch (...) {
    return fail(PPF_ERR_IO, "ppf_model_load_mem: not a valid model");
  }
}

ppf_status ppf_model_check_file(const char* path) {
  if (!path) return fail(PPF_ERR_INVALID, "ppf_model_check_file: NULL");
  ppf_model* none = nullptr;
  try {
    return model_load_impl(path, &none, true);
  } catch (const std::bad_alloc&) {
    return fail(PPF_ERR_NOMEM, "ppf_model_check_file: out of host memory reading %s", path);
  } catch (...) {
    return fail(PPF_ERR_IO, "ppf_model_check_file: %s is not a valid model file", path);
  }
}

}  // extern "C"

#endif /* PPF_MODEL_HOST_H */
