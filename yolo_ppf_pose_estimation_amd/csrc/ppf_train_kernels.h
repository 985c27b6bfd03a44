/*
 * ppf_train_kernels.h — accumulator geometry, cloud / slot-map types and the kernels that build the model table (row A5-train:
 * /root/reference/include/CloudProcessing.h:236 trainModel), plus the generic exclusive scan.  Included by ppf_hip.hip.
 */
#ifndef PPF_TRAIN_KERNELS_H
#define PPF_TRAIN_KERNELS_H

/* ============================================================================================ */
/* device code                                                                                    */
/* ============================================================================================ */

/* LDS accumulator geometry (see ppf_match_kernels.h): row pitch in words and guard words below cell 0 */
__host__ __device__ constexpr int vote_pitch(int A) { return A; } /* rows follow each other without a gap: bin A of a row (the reference's spill) IS bin 0 of the next row */
__host__ __device__ constexpr int vote_guard(int A) { return 64 + 2 * ((A + 1) | 1); }
/* accumulator words of a tile: guard, ceil(tile_refs / 2) word rows, and one word behind them for the spill of each half's last row */
__host__ __device__ constexpr int vote_lds_words(int tile_refs, int A) { return vote_guard(A) + ((tile_refs + 1) / 2) * vote_pitch(A) + 1; }
/* A tile of R model rows keeps 16-bit cells, two rows per 32-bit word: row r < H = ceil(R/2) in the low halves, row r + H
 * in the high halves.  A pair record names a row by the byte offset of its bin 0 with the half in bit 0. */
__host__ __device__ constexpr int vote_half_rows(int tile_refs) { return (tile_refs + 1) / 2; }
__host__ __device__ inline uint32_t vote_row_code(int row_local, int tile_refs, int A) {
  const int H = vote_half_rows(tile_refs);
  const int hf = row_local >= H ? 1 : 0;
  return (uint32_t)((vote_guard(A) + (row_local - hf * H) * vote_pitch(A)) * 4) | (uint32_t)hf;
}

struct CloudSoA {
  const float *x, *y, *z, *nx, *ny, *nz;
  int n;
};

struct SlotWord {
  uint32_t bits_lo, bits_hi, rank, pad;
};

/* rows (x y z at 0, normal at noff, pitch stride) -> six planes */
__global__ void k_aos_to_soa(const float* __restrict__ src, int n, int stride, int noff, float* __restrict__ dst, int pitch) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = src + (size_t)i * stride;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    dst[(size_t)k * pitch + i] = p[k];
    dst[(size_t)(3 + k) * pitch + i] = p[noff + k];
  }
}

/* row i of three arrays of a cloud.  The index goes in as a 32-bit BYTE offset next to each (scalar) base pointer, which a
 * global load takes as it is; from `a[i]` the compiler builds a 64-bit address per array (clouds have far fewer than 2^30 rows) */
__device__ __forceinline__ ppf_vec3 ld3(const float* a, const float* b, const float* c, int i) {
  const uint32_t o = (uint32_t)i * 4u;
  return ppf_mk3((double)*reinterpret_cast<const float*>(reinterpret_cast<const char*>(a) + o),
                 (double)*reinterpret_cast<const float*>(reinterpret_cast<const char*>(b) + o),
                 (double)*reinterpret_cast<const float*>(reinterpret_cast<const char*>(c) + o));
}

/* ---- training: one workgroup per model reference point i, threads sweep j (row A5-train) ---- */
__global__ __launch_bounds__(256) void k_train_pairs(CloudSoA m, double angle_step, double dist_step,
                                                     uint32_t slot_mask, int key_exact, int darboux, KeyDims kd,
                                                     uint32_t* __restrict__ pair_slot, float* __restrict__ pair_alpha,
                                                     unsigned long long* __restrict__ slot_bits) {
  __shared__ double frame[12];
  const int i = blockIdx.x;
  const ppf_vec3 p1 = ld3(m.x, m.y, m.z, i), n1 = ld3(m.nx, m.ny, m.nz, i);
  if (threadIdx.x == 0) ppf_transform_rt(p1, n1, frame, frame + 9);
  __syncthreads();
  double R[9], t[3];
  for (int k = 0; k < 9; k++) R[k] = frame[k];
  for (int k = 0; k < 3; k++) t[k] = frame[9 + k];
  for (int j = threadIdx.x; j < m.n; j += blockDim.x) {
    const size_t idx = (size_t)i * m.n + j;
    if (j == i) {
      pair_slot[idx] = 0xFFFFFFFFu;
      pair_alpha[idx] = 0.f;
      continue;
    }
    const ppf_vec3 p2 = ld3(m.x, m.y, m.z, j), n2 = ld3(m.nx, m.ny, m.nz, j);
    double f[4] = {0, 0, 0, 0};
    int32_t k0, k1, k2, k3;
    if (darboux) { /* PPF_FEATURE_DARBOUX: PCL's feature, floor() keys; degenerate pairs are left out of the table */
      if (!ppf_pair_feature_darboux(p1, n1, p2, n2, f)) { pair_slot[idx] = 0xFFFFFFFFu; pair_alpha[idx] = 0.f; continue; }
      k0 = ppf_floor_key(f[0] / angle_step); k1 = ppf_floor_key(f[1] / angle_step); k2 = ppf_floor_key(f[2] / angle_step);
      k3 = ppf_floor_key(f[3] / dist_step);
    } else {
      ppf_pair_feature(p1, n1, p2, n2, f);
      k0 = ppf_d2i(f[0] / angle_step); k1 = ppf_d2i(f[1] / angle_step); k2 = ppf_d2i(f[2] / angle_step);
      k3 = ppf_d2i(f[3] / dist_step);
    }
    uint32_t slot;
    if (key_exact) { /* PPF_KEY_EXACT: the "slot" is the quantised key itself (its index in the key table) */
      size_t ki;
      if (!key_index(kd, k0, k1, k2, k3, &ki) && !key_index_nan(kd, k0, k1, k2, k3, &ki)) { /* cannot happen: ppf_model_train checks the range */
        pair_slot[idx] = 0xFFFFFFFFu; pair_alpha[idx] = 0.f; continue;
      }
      slot = (uint32_t)ki;
    } else {
      slot = ppf_murmur_key16(k0, k1, k2, k3) & slot_mask; /* hash % slots, slots a power of two */
    }
    pair_slot[idx] = slot;
    pair_alpha[idx] = (float)ppf_model_alpha(R, t, p2);
    atomicOr(&slot_bits[slot >> 6], 1ull << (slot & 63));
  }
}

/* pcl::PPFEstimation::compute: row i*n + j = {f1, f2, f3, f4, alpha_m} (float32) of the pair (i, j); NaN rows for i == j and
 * for the pairs the Darboux feature leaves undefined */
__global__ __launch_bounds__(256) void k_pair_features(CloudSoA m, int darboux, float* __restrict__ out) {
  __shared__ double frame[12];
  const int i = blockIdx.x;
  const ppf_vec3 p1 = ld3(m.x, m.y, m.z, i), n1 = ld3(m.nx, m.ny, m.nz, i);
  if (threadIdx.x == 0) ppf_transform_rt(p1, n1, frame, frame + 9);
  __syncthreads();
  double R[9], t[3];
  for (int k = 0; k < 9; k++) R[k] = frame[k];
  for (int k = 0; k < 3; k++) t[k] = frame[9 + k];
  const float nanf_ = __uint_as_float(0x7FC00000u);
  for (int j = threadIdx.x; j < m.n; j += blockDim.x) {
    float* o = out + ((size_t)i * m.n + j) * 5;
    const ppf_vec3 p2 = ld3(m.x, m.y, m.z, j), n2 = ld3(m.nx, m.ny, m.nz, j);
    double f[4] = {0, 0, 0, 0};
    bool ok = j != i;
    if (ok) {
      if (darboux) ok = ppf_pair_feature_darboux(p1, n1, p2, n2, f) != 0;
      else ppf_pair_feature(p1, n1, p2, n2, f);
    }
    if (!ok) {
      for (int k = 0; k < 5; k++) o[k] = nanf_;
      continue;
    }
    for (int k = 0; k < 4; k++) o[k] = (float)f[k];
    o[4] = (float)ppf_model_alpha(R, t, p2);
  }
}

/* pcl::PPFHashMapSearch::nearestNeighborSearch: the model pairs (i, j) whose quantised feature IS the key -- what a hash map
 * keyed on the quantised feature holds under that key (the voting table itself does not keep j: a vote does not need it).
 * One workgroup per first point; the pairs found are appended through one cursor (any order: the host sorts them). */
__global__ __launch_bounds__(256) void k_key_pairs(CloudSoA m, int darboux, double angle_step, double dist_step, int k0, int k1, int k2, int k3,
                                                   uint2* __restrict__ out, uint32_t cap, uint32_t* __restrict__ cursor) {
  const int i = blockIdx.x;
  const ppf_vec3 p1 = ld3(m.x, m.y, m.z, i), n1 = ld3(m.nx, m.ny, m.nz, i);
  for (int j = threadIdx.x; j < m.n; j += blockDim.x) {
    if (j == i) continue;
    const ppf_vec3 p2 = ld3(m.x, m.y, m.z, j), n2 = ld3(m.nx, m.ny, m.nz, j);
    double f[4] = {0, 0, 0, 0};
    int32_t k[4];
    if (darboux) {
      if (!ppf_pair_feature_darboux(p1, n1, p2, n2, f)) continue;
      k[0] = ppf_floor_key(f[0] / angle_step); k[1] = ppf_floor_key(f[1] / angle_step); k[2] = ppf_floor_key(f[2] / angle_step);
      k[3] = ppf_floor_key(f[3] / dist_step);
    } else {
      ppf_pair_feature(p1, n1, p2, n2, f);
      k[0] = ppf_d2i(f[0] / angle_step); k[1] = ppf_d2i(f[1] / angle_step); k[2] = ppf_d2i(f[2] / angle_step); k[3] = ppf_d2i(f[3] / dist_step);
    }
    if (k[0] == k0 && k[1] == k1 && k[2] == k2 && k[3] == k3) {
      const uint32_t pos = atomicAdd(cursor, 1u);
      if (pos < cap) out[pos] = make_uint2((uint32_t)i, (uint32_t)j);
    }
  }
}

__global__ void k_bucket_total(const uint32_t* __restrict__ bucket_off, int n_buckets, int n_tiles, uint32_t* __restrict__ total) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_buckets) return;
  uint32_t t = 0;
  for (int k = 0; k < n_tiles; k++) {
    const uint32_t* row = bucket_off + (size_t)k * (n_buckets + 1);
    t += row[b + 1] - row[b];
  }
  total[b] = t;
}

/* Number of entries of every (tile, bucket) that belong to low-half rows.  With the dealing order above they are the
 * first n0 dealing positions (position j = record 32*(j/64) + j%32, slot (j%64)/32), the high-half rows' entries follow,
 * padding comes last: a 32-bit pass over the low halves needs records [0, 32*(n0/64) + min(n0%64, 32)), one over the
 * high halves [32*(n0/64) + max(n0%64 - 32, 0), end) -- see k_vote. */
__global__ void k_bucket_mid(const uint32_t* __restrict__ bucket_off, int n_buckets, int n_tiles, const uint4* __restrict__ records,
                             uint32_t first_real, uint32_t* __restrict__ mid) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n_buckets * n_tiles) return;
  const size_t t = idx / n_buckets, b = idx % n_buckets;
  const uint32_t* row = bucket_off + t * ((size_t)n_buckets + 1);
  const uint32_t off = row[b], cnt = row[b + 1] - off;
  uint32_t lo = 0, hi = 64u * ((cnt + 31u) / 32u); /* positions < lo are low-half entries, positions >= hi are not */
  while (lo < hi) {
    const uint32_t j = (lo + hi) >> 1;
    const uint32_t r = 32u * (j / 64u) + (j % 32u);
    bool low = false;
    if (r < cnt) {
      const uint4 rec = records[off + r];
      const uint32_t code = (((j % 64u) / 32u) ? rec.y : rec.x) & ROW_CODE_MASK;
      low = code >= first_real && !(code & 1u);
    }
    if (low) lo = j + 1; else hi = j;
  }
  mid[idx] = lo;
}

__global__ void k_popcount_words(const unsigned long long* __restrict__ bits, uint32_t* __restrict__ cnt, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cnt[i] = (uint32_t)__popcll(bits[i]);
}

__global__ void k_pack_slotmap(const unsigned long long* __restrict__ bits, const uint32_t* __restrict__ rank,
                               SlotWord* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long b = bits[i];
  SlotWord w;
  w.bits_lo = (uint32_t)b; w.bits_hi = (uint32_t)(b >> 32); w.rank = rank[i]; w.pad = 0;
  out[i] = w;
}

__device__ __forceinline__ int slot_to_bucket(const SlotWord* __restrict__ slotmap, uint32_t slot) {
  const SlotWord w = slotmap[slot >> 6];
  const unsigned long long bits = (unsigned long long)w.bits_lo | ((unsigned long long)w.bits_hi << 32);
  const uint32_t bit = slot & 63;
  if (!((bits >> bit) & 1ull)) return -1;
  return (int)(w.rank + (uint32_t)__popcll(bits & ((1ull << bit) - 1ull)));
}

/* key_lut[key_index(k0..k3)] = dense bucket of hash(k0..k3) % slots, or -1 */
__global__ __launch_bounds__(256) void k_build_key_lut(const SlotWord* __restrict__ slotmap, uint32_t slot_mask, int key_exact, KeyDims kd,
                                                       int32_t* __restrict__ lut) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= key_table_size(kd)) return;
  const int k3 = (int)(idx % kd.nd);
  size_t t = idx / kd.nd;
  const int k2 = (int)(t % kd.n2) - kd.o2; t /= kd.n2;
  const int k1 = (int)(t % kd.n1) - kd.o1;
  const int k0 = (int)(t / kd.n1) - kd.o0;
  lut[idx] = slot_to_bucket(slotmap, key_exact ? (uint32_t)idx : (ppf_murmur_key16(k0, k1, k2, k3) & slot_mask));
}

/* phase 0: count entries per (tile, bucket); phase 1: scatter through cursors */
/* ---- table layout (see also ppf_match_kernels.h) -------------------------------------------------------------
 * The entries of a (tile, bucket) are stored as PAIR RECORDS {row_a, row_b, alpha_a, alpha_b} (16 B): a lane of
 * k_vote loads one record (global_load_dwordx4), computes both alpha bins with one v_pk_fma_f32 and casts two
 * LDS atomics.  One ds_add_u32 wave-instruction therefore covers the a-slots (or the b-slots) of 64 consecutive
 * records; the LDS serves it as two halves of 32 lanes over 32 banks (bank = word address mod 32), one array cycle per
 * distinct address on the fullest bank of a half, and charges the wave max(4 cycles, array cycles) -- two addresses on a
 * bank are free, the third is not (profiles/r03_ubench_lds_counters.md; round 2 assumed 64 banks and groups of 16 lanes).
 * Entries are therefore put in a DEALING ORDER in which the 32 consecutive entries of a half rarely put three on a bank:
 *   - bank  c = (row_word + bin0(alpha_m)) mod 32: the bank of the vote when alpha_s == 0, and of the counted adds of
 *     the count-table path up to a constant; for another alpha_s all bins shift together, up to one bin of jitter
 *     decided by where alpha_m sits inside its bin;
 *   - level lv = the half of the accumulator words the entry's row owns (vote_row_code): the rows of the low halves are
 *     dealt first, so the records of a (tile, bucket) are those of its low-half rows, at most 32 mixed records, those of
 *     its high-half rows -- the launch with 32-bit cells walks only its half's share (k_bucket_mid);
 *   - inside a level the k-th entry (by phase) of a bank that holds n of them gets the key (k + 1/2) / n, and the
 *     entries are dealt in key order: every bank's entries are spread evenly, a bank with many entries (a model row
 *     that owns much of the bucket) as well as one with few -- dealing the banks round robin left the tail to the few
 *     heavy banks;
 *   - the level is first cut into up to 64 CELL GROUPS of at least PPF_DEAL_GROUP_MIN (256) entries each (deal_groups):
 *     group = the entry's count-table cell q (the 1/64 of an alpha bin its alpha_m sits in) scaled to the number of
 *     groups; the spreading above happens inside each group.  A level of 8,192 entries or more is thereby sorted by cell
 *     exactly, so the 128 entries of a record block share one cell (two at a group boundary): the lanes of k_vote's
 *     own-cell loop then walk the same hits for the same number of steps and read the same table rows (round 2 ordered
 *     a whole level by the per-bank phase quantile, which scattered a block over 3-5 cells: the loop ran as long as the
 *     fullest of them).  Smaller groups sort smaller levels exactly too but lose more to bank conflicts at the group
 *     boundaries than the loop gains (profiles/r03_vote_variants.md).
 * Dealing position j -> record 32*(j/64) + j%32, slot (j%64)/32: 32 consecutive dealing positions share a slot of
 * 32 consecutive records (one half of a wave-instruction is 32 consecutive dealing positions).  Unused slots of the last
 * records hold dummies that vote into the LDS guard words.
 */

__device__ uint32_t agg_cell_bits(float am, int A); /* ppf_match_kernels.h */
#ifndef PPF_DEAL_BANKS
#define PPF_DEAL_BANKS 32
#endif
constexpr uint32_t DEAL_BANKS = PPF_DEAL_BANKS; /* LDS banks the dealing order spreads a bucket's entries over */

/* cell groups of a level of n entries: the largest power of two <= gmax that leaves every group PPF_DEAL_GROUP_MIN entries on
 * average (a group must be several 32-entry windows long for the bank spreading inside it to work) */
#ifndef PPF_DEAL_GROUP_MIN
#define PPF_DEAL_GROUP_MIN 256
#endif
__host__ __device__ __forceinline__ uint32_t deal_groups(uint32_t n_level, uint32_t gmax) {
  uint32_t g = 1;
  while (2u * g <= gmax && n_level >= 2u * PPF_DEAL_GROUP_MIN * g) g <<= 1; /* a power of two, at most gmax */
  return g;
}
/* group of an entry: its count-table cell (0..AGG_Q-1, AGG_Q = votes one by one: last group) scaled to `groups` */
__device__ __forceinline__ uint32_t deal_group_of(float alpha_m, int num_angles, uint32_t groups) {
  const uint32_t q = min((agg_cell_bits(alpha_m, num_angles) >> ROW_Q_SHIFT) & ROW_Q_MASK, (uint32_t)AGG_Q - 1u);
  return (q * groups) / (uint32_t)AGG_Q;
}

__device__ __forceinline__ void entry_class_level(uint32_t row_bytes, float alpha_m, int num_angles, int levels,
                                                  uint32_t* cls, uint32_t* lvl) {
  const float q = alpha_m * (float)((double)num_angles / (4 * PPF_PI)) + 0.5f * (float)num_angles;
  const float fl = floorf(q);
  *cls = (row_bytes / 4u + (uint32_t)(int)fl) & (DEAL_BANKS - 1u);
  *lvl = levels > 1 ? (row_bytes & 1u) : 0u; /* the half of the accumulator word the entry's row owns: low-half rows are dealt first */
}

__host__ __device__ __forceinline__ uint32_t records_for(uint32_t n_entries) {
  return 32u * (n_entries / 64u) + min(32u, n_entries % 64u);
}

/* phase 0: count; phase 1: place (rec_off = record offset of the (tile, bucket)).
 * pos = the entry's dealing position inside its level (k_train_spread + sorts); mirror: one of the few mirrored spill
 * entries, which are not part of the sorts: they take the last positions of the low-half level */
/* level_cnt[tb*2 + level]: entries of the level (mirrored ones included, they sit at the end of level 0);
 * group_cnt[(tb*2 + level)*gmax + group]: sorted entries of each cell group (k_train_keys); mirror_cur[tb]: cursor of the mirrored
 * entries.  pos = the entry's position inside its group (k_train_spread + sorts). */
__device__ __forceinline__ void place_entry(int phase, size_t tb, uint32_t row_bytes, float am, int num_angles, int levels,
                                            uint32_t* __restrict__ counts, const uint32_t* __restrict__ rec_off,
                                            uint32_t* __restrict__ level_cnt, uint32_t* __restrict__ mirror_cur,
                                            const uint32_t* __restrict__ group_cnt, uint32_t gmax,
                                            uint4* __restrict__ records, bool mirror, uint32_t pos) {
  uint32_t c, lv;
  entry_class_level(row_bytes, am, num_angles, levels, &c, &lv);
  if (phase == 0) {
    atomicAdd(&counts[tb], 1u);
    atomicAdd(&level_cnt[tb * 2 + lv], 1u);
    return;
  }
  const uint32_t n0 = level_cnt[tb * 2]; /* entries of the low-half level */
  uint32_t j;
  if (mirror) {
    j = n0 - 1u - atomicAdd(&mirror_cur[tb], 1u);
  } else {
    /* the same group as k_train_keys gave the entry: both size the groups from the level's count */
    const uint32_t groups = deal_groups(level_cnt[tb * 2 + lv], gmax), g = deal_group_of(am, num_angles, groups);
    uint32_t before = 0;
    const uint32_t* gc = group_cnt + (tb * 2 + lv) * (size_t)gmax;
    for (uint32_t k = 0; k < g; k++) before += gc[k];
    j = (lv ? n0 : 0u) + before + pos;
  }
  uint32_t* rec = reinterpret_cast<uint32_t*>(&records[rec_off[tb] + 32u * (j / 64u) + (j % 32u)]);
  const uint32_t slot = (j % 64u) / 32u;
  rec[slot] = row_bytes | agg_cell_bits(am, num_angles);
  rec[2 + slot] = __float_as_uint(am);
}

__global__ void k_train_bin(const uint32_t* __restrict__ pair_slot, const float* __restrict__ pair_alpha, int n_model,
                            const SlotWord* __restrict__ slotmap, int n_buckets, int tile_refs, int n_tiles,
                            int num_angles, int levels, uint32_t* __restrict__ counts, const uint32_t* __restrict__ rec_off,
                            uint32_t* __restrict__ level_cnt, uint32_t* __restrict__ mirror_cur,
                            const uint32_t* __restrict__ group_cnt, uint32_t gmax,
                            uint4* __restrict__ records, uint32_t* __restrict__ bucket_slot, int phase,
                            const uint32_t* __restrict__ pair_rank = nullptr) {
  const size_t total = (size_t)n_model * n_model;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const uint32_t slot = pair_slot[idx];
  if (slot == 0xFFFFFFFFu) return;
  const int i = (int)(idx / n_model);
  const int b = slot_to_bucket(slotmap, slot);
  const int tile = i / tile_refs;
  const float am = pair_alpha[idx];
  if (phase == 0 && bucket_slot) bucket_slot[b] = slot;
  place_entry(phase, (size_t)tile * n_buckets + b, vote_row_code(i - tile * tile_refs, tile_refs, num_angles), am, num_angles,
              levels, counts, rec_off, level_cnt, mirror_cur, group_cnt, gmax, records, false, pair_rank ? pair_rank[idx] : 0u);
  /* alpha bin == numAngles spills into the next model reference point's bin 0 (see k_vote); when
   * that point lives in the next tile, the entry is mirrored there: bin A -> cell 0, others -> guard. */
  if (am >= SPILL_ALPHA_MIN && tile + 1 < n_tiles && i == (tile + 1) * tile_refs - 1)
    place_entry(phase, (size_t)(tile + 1) * n_buckets + b, (uint32_t)((vote_guard(num_angles) - num_angles) * 4), am,
                num_angles, levels, counts, rec_off, level_cnt, mirror_cur, group_cnt, gmax, records, true, 0u);
}

/* sort keys of the model pairs for the dealing order: key_class = (((tile*n_buckets + bucket)*2 + level)*gmax + cell group)*DEAL_BANKS
 * + bank (invalid pairs: `invalid`), key_phase = position of alpha_m inside its bin, 16 bits; also counts the entries of every
 * cell group */
__global__ __launch_bounds__(256) void k_train_keys(const uint32_t* __restrict__ pair_slot, const float* __restrict__ pair_alpha, int n_model,
                                                    const SlotWord* __restrict__ slotmap, int n_buckets, int tile_refs, int num_angles,
                                                    const uint32_t* __restrict__ level_cnt, uint32_t gmax, uint32_t* __restrict__ group_cnt,
                                                    uint32_t invalid, uint32_t* __restrict__ key_class, uint32_t* __restrict__ key_phase,
                                                    uint32_t* __restrict__ vals) {
  const size_t total = (size_t)n_model * n_model;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  vals[idx] = (uint32_t)idx;
  const uint32_t slot = pair_slot[idx];
  if (slot == 0xFFFFFFFFu) { key_class[idx] = invalid; key_phase[idx] = 0; return; }
  const int i = (int)(idx / n_model);
  const int b = slot_to_bucket(slotmap, slot);
  const int tile = i / tile_refs;
  const float am = pair_alpha[idx];
  const uint32_t row_bytes = vote_row_code(i - tile * tile_refs, tile_refs, num_angles);
  uint32_t c, lv;
  entry_class_level(row_bytes, am, num_angles, 2, &c, &lv);
  const float q = am * (float)((double)num_angles / (4 * PPF_PI)) + 0.5f * (float)num_angles;
  const size_t seg = ((size_t)tile * n_buckets + b) * 2 + lv;
  const uint32_t g = deal_group_of(am, num_angles, deal_groups(level_cnt[seg], gmax));
  atomicAdd(&group_cnt[seg * gmax + g], 1u);
  key_class[idx] = (uint32_t)((seg * gmax + g) * DEAL_BANKS + c);
  key_phase[idx] = min((uint32_t)((q - floorf(q)) * 65536.0f), 65535u);
}
/* second stage of the dealing order: key_frac = (k + 1/2) / n as a 32-bit fraction, k = the pair's rank by phase inside its
 * (tile, bucket, level, cell group, bank) and n that class's size; key_seg = ((tile*n_buckets + bucket)*2 + level)*gmax + cell group */
__global__ __launch_bounds__(256) void k_train_spread(const uint32_t* __restrict__ key_class, const uint32_t* __restrict__ pair_rank,
                                                      const uint32_t* __restrict__ pair_n, uint32_t invalid, size_t n,
                                                      uint32_t* __restrict__ key_frac, uint32_t* __restrict__ key_seg,
                                                      uint32_t* __restrict__ vals) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  vals[idx] = (uint32_t)idx;
  const uint32_t kc = key_class[idx];
  if (kc == invalid) { key_frac[idx] = 0u; key_seg[idx] = invalid / DEAL_BANKS; return; }
  const uint32_t nb = pair_n[idx]; /* >= rank + 1 */
  key_frac[idx] = (uint32_t)((((unsigned long long)(2u * pair_rank[idx] + 1u)) << 31) / nb);
  key_seg[idx] = kc / DEAL_BANKS;
}
__global__ __launch_bounds__(256) void k_gather_u32(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, size_t n,
                                                    uint32_t* __restrict__ dst) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) dst[p] = src[idx[p]];
}
/* rank of every pair inside its run of equal class keys: pair_rank[vals[p]] = p - start of p's run; pair_n (optional) = the
 * run's length */
__global__ __launch_bounds__(256) void k_train_ranks(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ starts, uint32_t n_runs,
                                                     size_t n, uint32_t* __restrict__ pair_rank, uint32_t* __restrict__ pair_n = nullptr) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint32_t lo = 0, hi = n_runs; /* last run start <= p */
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((size_t)starts[mid] <= p) lo = mid; else hi = mid;
  }
  pair_rank[vals[p]] = (uint32_t)(p - starts[lo]);
  if (pair_n) pair_n[vals[p]] = (lo + 1 < n_runs ? starts[lo + 1] : (uint32_t)n) - starts[lo];
}

__global__ void k_record_counts(const uint32_t* __restrict__ counts, uint32_t* __restrict__ rec_cnt, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rec_cnt[i] = records_for(counts[i]);
}

/* every slot starts as a dummy: row = one of the first 64 guard words (never a cell), alpha = 0.00327 (any value whose
 * alpha*A/(4 pi) sits in the middle of a 1/64 cell for the usual A: the count-table path of k_vote then treats it like any
 * other entry instead of taking its on-a-cell-boundary route) */
__global__ void k_record_init(uint4* __restrict__ records, size_t n, int num_angles) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint32_t w = (uint32_t)(i & 63u) * 4u | agg_cell_bits(0.00327f, num_angles);
    const uint32_t al = __float_as_uint(0.00327f);
    records[i] = make_uint4(w, w, al, al);
  }
}

/* ---- exclusive scan (u32), 1024 elements per block ------------------------------------------- */
__global__ __launch_bounds__(256) void k_scan_block(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                    uint32_t* __restrict__ block_sums, size_t n) {
  __shared__ uint32_t wave_tot[4];
  const size_t base = (size_t)blockIdx.x * 1024 + (size_t)threadIdx.x * 4;
  uint32_t v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = (base + k < n) ? in[base + k] : 0u;
  uint32_t s = v[0] + v[1] + v[2] + v[3];
  uint32_t incl = s;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t y = __shfl_up(incl, o);
    if (lane >= o) incl += y;
  }
  if (lane == 63) wave_tot[wv] = incl;
  __syncthreads();
  uint32_t woff = 0;
  for (int k = 0; k < wv; k++) woff += wave_tot[k];
  uint32_t excl = woff + incl - s;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (base + k < n) out[base + k] = excl;
    excl += v[k];
  }
  if (threadIdx.x == 255 && block_sums) block_sums[blockIdx.x] = woff + incl;
}
/* the same scan by ONE workgroup in rounds of 4,096 elements: up to a few ten thousand elements (the flags, cell counts and
 * digit tables of the preparation stages and of a crop's sampling) one launch instead of three, which cost more than they compute */
__global__ __launch_bounds__(1024) void k_scan_one(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
  __shared__ uint32_t wave_tot[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t carry = 0;
  for (size_t r0 = 0; r0 < n; r0 += 4096) {
    const size_t base = r0 + (size_t)threadIdx.x * 4;
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = (base + k < n) ? in[base + k] : 0u;
    const uint32_t s = v[0] + v[1] + v[2] + v[3];
    uint32_t incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t y = __shfl_up(incl, o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { const uint32_t w = wave_tot[k]; if (k < wv) woff += w; total += w; }
    uint32_t excl = carry + woff + incl - s;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (base + k < n) out[base + k] = excl;
      excl += v[k];
    }
    carry += total;
    __syncthreads(); /* wave_tot is written again in the next round */
  }
}
__global__ void k_scan_add(uint32_t* __restrict__ out, const uint32_t* __restrict__ block_off, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += block_off[i / 1024];
}

#endif /* PPF_TRAIN_KERNELS_H */
