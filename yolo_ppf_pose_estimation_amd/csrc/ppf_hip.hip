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

/* ============================================================================================ */
/* errors                                                                                         */
/* ============================================================================================ */
namespace {

thread_local std::string g_last_error;

ppf_status fail(ppf_status st, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  /* error paths return while kernels of the failed call may still run; their scratch goes back to the block cache
   * (DevPool) as the locals unwind, so drain the device first.  Errors are rare: the cost does not matter. */
  if (st == PPF_ERR_HIP || st == PPF_ERR_NOMEM || st == PPF_ERR_CAPACITY) (void)hipDeviceSynchronize();
  return st;
}

#define HIPCHK(expr)                                                                                       \
  do {                                                                                                     \
    hipError_t e__ = (expr);                                                                               \
    if (e__ != hipSuccess)                                                                                 \
      return fail(PPF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

/* Device memory for scratch and results comes from a process-wide cache of freed blocks (power-of-two size classes
 * per device): hipMalloc costs tens of microseconds and hipFree synchronises the whole device, which is most of the
 * time of the small stages (cloud stages, ICP set-up).  A block is only released by a DevBuf whose last user has been
 * synchronised with (every entry point waits for its kernels before its scratch goes out of scope), so a reused block
 * is never still in flight.  PPF_NO_POOL=1 turns the cache off. */
class DevPool {
 public:
  static DevPool& get() {
    static DevPool* p = new DevPool(); /* never destroyed: no hipFree after the runtime is gone */
    return *p;
  }
  /* size classes: 8 per octave (1, 1.125, ... 1.875 x 2^k), so a block wastes at most 12.5 % of what was asked for */
  static size_t class_size(int cls) { return ((size_t)8 + (size_t)(cls & 7)) << (cls >> 3); }
  static int class_of(size_t bytes) {
    int cls = 5 * 8; /* 256 B */
    while (class_size(cls) < bytes) cls++;
    return cls;
  }
  hipError_t acquire(size_t bytes, void** out, size_t* granted, int* device) {
    *out = nullptr;
    const size_t want = std::max<size_t>(bytes, 256);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    *device = dev; /* a block goes back to the list of the device it was allocated on, whatever is current then */
    if (off_) { *granted = want; return hipMalloc(out, want); }
    const int cls = class_of(want);
    {
      std::lock_guard<std::mutex> g(mu_);
      auto& lst = free_[key(dev, cls)];
      if (!lst.empty()) { *out = lst.back(); lst.pop_back(); *granted = class_size(cls); return hipSuccess; }
    }
    *granted = class_size(cls);
    e = hipMalloc(out, *granted);
    if (e != hipSuccess) { /* out of memory: drop the cache and retry once */
      trim();
      e = hipMalloc(out, *granted);
    }
    return e;
  }
  void release(void* p, size_t granted, int dev) {
    if (!p) return;
    if (off_) { (void)hipFree(p); return; }
    const int cls = class_of(granted);
    std::lock_guard<std::mutex> g(mu_);
    free_[key(dev, cls)].push_back(p);
  }
  void trim() {
    std::lock_guard<std::mutex> g(mu_);
    for (auto& kv : free_) {
      for (void* p : kv.second) (void)hipFree(p);
      kv.second.clear();
    }
  }

 private:
  DevPool() : off_(getenv("PPF_NO_POOL") != nullptr) {}
  static int key(int dev, int cls) { return dev * 1024 + cls; }
  std::mutex mu_;
  std::map<int, std::vector<void*>> free_;
  bool off_;
};

void sync_device_of_block(int dev); /* = sync_device, defined below */

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;      /* elements usable */
  size_t granted = 0;  /* bytes of the block behind p */
  int device = 0;      /* device the block lives on */
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { DevPool::get().release(p, granted, device); }
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) { /* growing a live buffer (rare): earlier asynchronous work may still use the old block */
      sync_device_of_block(device);
      DevPool::get().release(p, granted, device); p = nullptr; cap = 0; granted = 0;
    }
    void* q = nullptr;
    hipError_t e = DevPool::get().acquire(std::max<size_t>(n, 1) * sizeof(T), &q, &granted, &device);
    if (e == hipSuccess) { p = static_cast<T*>(q); cap = n; }
    return e;
  }
  /* like reserve, but a block more than twice as big as needed (and above 16 MiB) is traded for a fitting one: the hit
   * pools of a workspace shrink again after an unusually dense scene */
  hipError_t fit(size_t n) {
    if (p && granted > ((size_t)16 << 20) && granted > 2 * std::max<size_t>(n, 1) * sizeof(T)) {
      sync_device_of_block(device);
      DevPool::get().release(p, granted, device); p = nullptr; cap = 0; granted = 0;
    }
    return reserve(n);
  }
  size_t bytes() const { return granted; }
};

void sync_device(int dev);
void sync_device_of_block(int dev) { sync_device(dev); }

/* wait for everything enqueued on device `dev` (the device a buffer lives on, which need not be the current one) */
void sync_device(int dev) {
  int cur = -1;
  if (dev < 0 || hipGetDevice(&cur) != hipSuccess || cur == dev) { (void)hipDeviceSynchronize(); return; }
  if (hipSetDevice(dev) == hipSuccess) {
    (void)hipDeviceSynchronize();
    (void)hipSetDevice(cur);
  }
}

constexpr int LDS_BYTES = 160 * 1024;              /* LDS per CU == per k_vote workgroup */
constexpr size_t HIT_BYTES_BUDGET = 4ull << 30;    /* hit scratch per batch of reference points */
constexpr float SPILL_ALPHA_MIN = 3.1415f;         /* entries with alpha_m >= this can reach alpha bin == numAngles */

}  // namespace

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

__global__ void k_aos_to_soa(const float* __restrict__ src, int n, int stride, float* __restrict__ dst, int pitch) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = src + (size_t)i * stride;
#pragma unroll
  for (int k = 0; k < 6; k++) dst[(size_t)k * pitch + i] = p[k];
}

__device__ __forceinline__ ppf_vec3 ld3(const float* a, const float* b, const float* c, int i) {
  return ppf_mk3((double)a[i], (double)b[i], (double)c[i]);
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
      if (!key_index(kd, k0, k1, k2, k3, &ki)) { pair_slot[idx] = 0xFFFFFFFFu; pair_alpha[idx] = 0.f; continue; } /* cannot happen: ppf_model_train checks the range */
      slot = (uint32_t)ki;
    } else {
      slot = ppf_murmur_key16(k0, k1, k2, k3) & slot_mask; /* hash % slots, slots a power of two */
    }
    pair_slot[idx] = slot;
    pair_alpha[idx] = (float)ppf_model_alpha(R, t, p2);
    atomicOr(&slot_bits[slot >> 6], 1ull << (slot & 63));
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
 * records; the LDS pipe takes them in groups of 16 lanes and serialises the lanes of a group that meet in one of its 64
 * banks (profiles/r02_ubench_lds_ops.txt).  Entries are therefore put in a DEALING ORDER in which 16 consecutive entries
 * hit (almost always) 16 different banks:
 *   - bank  c = (row_word + bin0(alpha_m)) mod 64: the bank of the vote when alpha_s == 0, and of the counted adds of
 *     the count-table path up to a constant; for another alpha_s all bins shift together, up to one bin of jitter
 *     decided by where alpha_m sits inside its bin;
 *   - level lv = the half of the accumulator words the entry's row owns (vote_row_code): the rows of the low halves are
 *     dealt first, so the records of a (tile, bucket) are those of its low-half rows, at most 32 mixed records, those of
 *     its high-half rows -- the launch with 32-bit cells walks only its half's share (k_bucket_mid);
 *   - inside a level the k-th entry (by phase) of a bank that holds n of them gets the key (k + 1/2) / n, and the
 *     entries are dealt in key order: every bank's entries are spread evenly over the level, a bank with many
 *     entries (a model row that owns much of the bucket) as well as one with few -- dealing the banks round robin
 *     left the tail of a level to the few heavy banks (1.32 serialised passes per 16 counted adds on the headline
 *     table, against 1.09 now; direct votes with their jitter 1.58 -> 1.31).
 * Dealing position j -> record 32*(j/64) + j%32, slot (j%64)/32: 32 consecutive dealing positions share a slot of
 * 32 consecutive records (lanes 16g .. 16g+15 of a wave-instruction are 16 consecutive dealing positions).  Unused slots of the last records hold dummies that vote into the LDS guard words.
 */

__device__ uint32_t agg_cell_bits(float am, int A); /* ppf_match_kernels.h */
constexpr uint32_t DEAL_BANKS = 64; /* LDS banks the dealing order spreads a bucket's entries over */

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
__device__ __forceinline__ void place_entry(int phase, size_t tb, uint32_t row_bytes, float am, int num_angles, int levels,
                                            uint32_t* __restrict__ counts, const uint32_t* __restrict__ rec_off,
                                            uint32_t* __restrict__ class_cnt, uint32_t* __restrict__ class_cur,
                                            uint4* __restrict__ records, bool mirror, uint32_t pos) {
  uint32_t c, lv;
  entry_class_level(row_bytes, am, num_angles, levels, &c, &lv);
  const size_t cbase = tb * (size_t)levels * DEAL_BANKS;
  if (phase == 0) {
    atomicAdd(&counts[tb], 1u);
    atomicAdd(&class_cnt[cbase + lv * DEAL_BANKS + c], 1u);
    return;
  }
  uint32_t n0 = 0; /* entries of the low-half level */
  if (lv || mirror)
    for (uint32_t cc = 0; cc < DEAL_BANKS; cc++) n0 += class_cnt[cbase + cc];
  const uint32_t j = mirror ? n0 - 1u - atomicAdd(&class_cur[cbase], 1u) : (lv ? n0 : 0u) + pos;
  uint32_t* rec = reinterpret_cast<uint32_t*>(&records[rec_off[tb] + 32u * (j / 64u) + (j % 32u)]);
  const uint32_t slot = (j % 64u) / 32u;
  rec[slot] = row_bytes | agg_cell_bits(am, num_angles);
  rec[2 + slot] = __float_as_uint(am);
}

__global__ void k_train_bin(const uint32_t* __restrict__ pair_slot, const float* __restrict__ pair_alpha, int n_model,
                            const SlotWord* __restrict__ slotmap, int n_buckets, int tile_refs, int n_tiles,
                            int num_angles, int levels, uint32_t* __restrict__ counts, const uint32_t* __restrict__ rec_off,
                            uint32_t* __restrict__ class_cnt, uint32_t* __restrict__ class_cur,
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
              levels, counts, rec_off, class_cnt, class_cur, records, false, pair_rank ? pair_rank[idx] : 0u);
  /* alpha bin == numAngles spills into the next model reference point's bin 0 (see k_vote); when
   * that point lives in the next tile, the entry is mirrored there: bin A -> cell 0, others -> guard. */
  if (am >= SPILL_ALPHA_MIN && tile + 1 < n_tiles && i == (tile + 1) * tile_refs - 1)
    place_entry(phase, (size_t)(tile + 1) * n_buckets + b, (uint32_t)((vote_guard(num_angles) - num_angles) * 4), am,
                num_angles, levels, counts, rec_off, class_cnt, class_cur, records, true, 0u);
}

/* sort keys of the model pairs for the dealing order: key_class = ((tile*n_buckets + bucket)*2 + level)*64 + bank (invalid pairs:
 * `invalid`), key_phase = position of alpha_m inside its bin, 16 bits */
__global__ __launch_bounds__(256) void k_train_keys(const uint32_t* __restrict__ pair_slot, const float* __restrict__ pair_alpha, int n_model,
                                                    const SlotWord* __restrict__ slotmap, int n_buckets, int tile_refs, int num_angles,
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
  key_class[idx] = (uint32_t)((((size_t)tile * n_buckets + b) * 2 + lv) * DEAL_BANKS + c);
  key_phase[idx] = min((uint32_t)((q - floorf(q)) * 65536.0f), 65535u);
}
/* second stage of the dealing order: key_frac = (k + 1/2) / n as a 32-bit fraction, k = the pair's rank by phase inside its
 * (tile, bucket, level, bank) and n that group's size; key_seg = (tile*n_buckets + bucket)*2 + level */
__global__ __launch_bounds__(256) void k_train_spread(const uint32_t* __restrict__ key_class, const uint32_t* __restrict__ pair_rank,
                                                      const uint32_t* __restrict__ class_cnt, uint32_t invalid, size_t n,
                                                      uint32_t* __restrict__ key_frac, uint32_t* __restrict__ key_seg,
                                                      uint32_t* __restrict__ vals) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  vals[idx] = (uint32_t)idx;
  const uint32_t kc = key_class[idx];
  if (kc == invalid) { key_frac[idx] = 0u; key_seg[idx] = invalid / DEAL_BANKS; return; }
  const uint32_t nb = class_cnt[kc]; /* >= rank + 1 */
  key_frac[idx] = (uint32_t)((((unsigned long long)(2u * pair_rank[idx] + 1u)) << 31) / nb);
  key_seg[idx] = kc / DEAL_BANKS;
}
__global__ __launch_bounds__(256) void k_gather_u32(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, size_t n,
                                                    uint32_t* __restrict__ dst) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) dst[p] = src[idx[p]];
}
/* rank of every pair inside its run of equal class keys: pair_rank[vals[p]] = p - start of p's run */
__global__ __launch_bounds__(256) void k_train_ranks(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ starts, uint32_t n_runs,
                                                     size_t n, uint32_t* __restrict__ pair_rank) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint32_t lo = 0, hi = n_runs; /* last run start <= p */
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((size_t)starts[mid] <= p) lo = mid; else hi = mid;
  }
  pair_rank[vals[p]] = (uint32_t)(p - starts[lo]);
}

__global__ void k_record_counts(const uint32_t* __restrict__ counts, uint32_t* __restrict__ rec_cnt, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rec_cnt[i] = records_for(counts[i]);
}

/* every slot starts as a dummy: row = one of the first 64 guard words (never a cell), alpha = 0.0065 (any value whose
 * alpha*A/(4 pi) sits in the middle of a 1/32 cell for the usual A: the count-table path of k_vote then treats it like any
 * other entry instead of taking its on-a-cell-boundary route) */
__global__ void k_record_init(uint4* __restrict__ records, size_t n, int num_angles) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint32_t w = (uint32_t)(i & 63u) * 4u | agg_cell_bits(0.0065f, num_angles);
    const uint32_t al = __float_as_uint(0.0065f);
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
__global__ void k_scan_add(uint32_t* __restrict__ out, const uint32_t* __restrict__ block_off, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += block_off[i / 1024];
}

#include "ppf_sample_kernels.h"
#include "ppf_match_kernels.h"
#include "ppf_icp_kernels.h"
#include "ppf_prep_kernels.h"

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
};

__global__ void k_finalize(FinalArgs a) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.n_ref) return;
  uint32_t maxVotes = 0, flat = 0;
  unsigned long long nv = 0;
  for (int t = 0; t < a.n_tiles; t++) {
    const size_t slot = ((size_t)r * a.n_tiles + t) * 2;
    const uint2 p = a.partial[slot];
    nv += a.cellsum[(size_t)r * a.n_tiles + t];
    if (p.x > maxVotes) { maxVotes = p.x; flat = (uint32_t)(t * a.tile_refs * a.num_angles) + p.y; }
    if (a.acc32 || a.ovf_items[(size_t)r * a.n_tiles + t]) { /* the high-half rows come after the low-half rows; bin 0 of their first row still lacks the spill
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
  atomicAdd(&a.totals[0], nv);
  atomicAdd(&a.totals[1], a.pairs[r]);

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

/* ============================================================================================ */
/* host side                                                                                      */
/* ============================================================================================ */
namespace {

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

ppf_status device_exclusive_scan(const uint32_t* in, uint32_t* out, size_t n, hipStream_t st) {
  if (n == 0) return PPF_OK;
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
  ppf_status load_device(const float* d_src, int rows, int stride, hipStream_t st) {
    n = rows;
    pitch = (rows + 63) & ~63;
    HIPCHK(buf.reserve((size_t)6 * std::max(pitch, 64)));
    if (rows > 0) {
      k_aos_to_soa<<<dim3((rows + 255) / 256), dim3(256), 0, st>>>(d_src, rows, stride, buf.p, pitch);
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
ppf_status device_sample_cloud(const float* d_src, int n, int stride, float step, CloudDev& dst,
                               std::vector<float>* host_rows, hipStream_t st) {
  const int ns = (int)(1.0 / step);
  DevBuf<uint32_t> bbox, keys, vals, keys2, vals2, starts;
  HIPCHK(bbox.reserve(6));
  const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
  HIPCHK(hipMemcpyAsync(bbox.p, init, sizeof(init), hipMemcpyHostToDevice, st));
  const unsigned nb256 = (unsigned)((n + 255) / 256);
  k_bbox<<<dim3(std::min(nb256, 2048u)), dim3(256), 0, st>>>(d_src, n, stride, bbox.p);
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
    k_seg_sum<<<dim3((n_rows + 63) / 64), dim3(64), 0, st>>>(d_src, stride, va, starts.p, (int)n_rows, n, dst.buf.p, dst.pitch,
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

struct ppf_model {
  std::atomic<int> refcount{1};
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
  DevBuf<uint2> run_blocks;
  DevBuf<unsigned long long> work;
  DevBuf<uint32_t> perm;
  DevBuf<uint32_t> perm_group;
  DevBuf<unsigned long long> counters; /* cellsum[n_ref*T] | pairs[n_ref] | totals[2] | tally[5] */
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
  ppf_match_params params{};
  int n_ref = 0, n_ref_total = 0, rows = 0;
  hipStream_t stream = nullptr;
  bool timing = false;
  hipEvent_t ev[2] = {nullptr, nullptr}; /* first kernel start, last kernel end */
  std::vector<hipEvent_t> batch_ev;      /* 4 per batch: k_pairs start / end, k_vote start / end */
  int n_batches = 0;
  bool pending = false;
  bool checked = false;                  /* the overflow flag of the pending call has been read */
  bool has_edge = false;
  double hit_frac = 0.25;                /* expected hits per scene pair: sizes the hit pools, learned from every call */
  bool frac_known = false;               /* false: the next call first COUNTS its hits (one extra pair pass and one wait) */
  double run_frac = 0.4;                 /* expected runs (distinct buckets hit by a reference point) per hit, learned likewise */
  struct Learned { const ppf_model* model; double hit, run; };
  std::vector<Learned> frac_by_model;    /* the two fractions remembered per model (batches alternate models) */
  int round_buckets_cap = 0;             /* 0 = GROUP_MAX_BUCKETS; tests lower it to force several k_group rounds */
  bool acc32 = false;                    /* a 16-bit accumulator cell overflowed with this model: 32-bit cells until the model changes */
  bool force_acc32 = false;              /* PPF_OPT_ACC32 */
  bool cluster_serial = false;           /* force the serial greedy assignment (otherwise only used above 11,520 poses) */
  int device = -1;
  uint32_t* acc_dump = nullptr; /* set by ppf_debug_accumulators for one call */
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
   * inside a level every bank's entries in phase order, spread evenly over the level. */
  const int levels = 2;
  const size_t ncls = ncnt * (size_t)levels * DEAL_BANKS;
  DevBuf<uint32_t> class_cnt, class_cur;
  HIPCHK(class_cnt.reserve(ncls));
  HIPCHK(class_cur.reserve(ncls));
  HIPCHK(hipMemsetAsync(class_cnt.p, 0, ncls * sizeof(uint32_t), st));
  HIPCHK(hipMemsetAsync(class_cur.p, 0, ncls * sizeof(uint32_t), st));
  const unsigned nblk = (unsigned)((NN + 255) / 256);
  k_train_bin<<<dim3(nblk), dim3(256), 0, st>>>(pair_slot.p, pair_alpha.p, N, m->slotmap.p, (int)n_buckets,
                                                m->info.tile_refs, T, m->info.num_angles, levels, counts.p, nullptr,
                                                class_cnt.p, class_cur.p, nullptr, m->bucket_slot.p, 0);
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
  DevBuf<uint32_t> pair_pos; /* dealing position of every pair inside its (tile, bucket, level) */
  {
    DevBuf<uint32_t> kcls, kph, v1, kt, v2, starts, pair_rank;
    HIPCHK(kcls.reserve(NN)); HIPCHK(kph.reserve(NN)); HIPCHK(v1.reserve(NN)); HIPCHK(kt.reserve(NN)); HIPCHK(v2.reserve(NN));
    HIPCHK(pair_rank.reserve(NN)); HIPCHK(pair_pos.reserve(NN));
    const uint32_t invalid = (uint32_t)((size_t)T * n_buckets * 2 * DEAL_BANKS);
    k_train_keys<<<dim3(nblk), dim3(256), 0, st>>>(pair_slot.p, pair_alpha.p, N, m->slotmap.p, (int)n_buckets, m->info.tile_refs,
                                                   m->info.num_angles, invalid, kcls.p, kph.p, v1.p);
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
    k_train_ranks<<<dim3(nblk), dim3(256), 0, st>>>(va, starts.p, n_runs, NN, pair_rank.p);
    HIPCHK(hipGetLastError());
    /* (2) position inside the level: the banks' entries spread evenly, i.e. sorted by (rank + 1/2) / bank size; ties keep
     * the pair order (stable sorts from the identity) */
    k_train_spread<<<dim3(nblk), dim3(256), 0, st>>>(kcls.p, pair_rank.p, class_cnt.p, invalid, NN, kph.p, kt.p, v1.p);
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
                                                class_cnt.p, class_cur.p, m->records.p, nullptr, 1, pair_pos.p);
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

ppf_status ppf_model_train(const float* xyzn, int n, int stride, const ppf_train_params* params, ppf_model** out) {
  if (!out) return fail(PPF_ERR_INVALID, "ppf_model_train: out is NULL");
  *out = nullptr;
  if (!xyzn || n <= 1 || stride < 6 || !params) return fail(PPF_ERR_INVALID, "ppf_model_train: bad argument");
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
    for (int i = 0; i < n; i++) memcpy(&m->sampled[(size_t)i * 6], xyzn + (size_t)i * stride, 24);
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
      s0 = e == hipSuccess ? device_sample_cloud(d_raw.p, n, stride, (float)params->relative_sampling_step, m->cloud, &m->sampled, st)
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

/* ---- workspace / matching --------------------------------------------------------------------- */
ppf_status ppf_workspace_create(ppf_workspace** out) {
  if (!out) return fail(PPF_ERR_INVALID, "ppf_workspace_create: NULL");
  *out = new (std::nothrow) ppf_workspace();
  if (!*out) return fail(PPF_ERR_NOMEM, "ppf_workspace_create: out of memory");
  return PPF_OK;
}
ppf_status ppf_workspace_destroy(ppf_workspace* ws) {
  if (!ws) return PPF_OK;
  delete ws; /* the destructor drains the device its buffers live on before they return to the block cache */
  return PPF_OK;
}
ppf_status ppf_workspace_set_option(ppf_workspace* ws, int option, double value) {
  if (!ws) return fail(PPF_ERR_INVALID, "ppf_workspace_set_option: NULL");
  switch (option) {
    case PPF_OPT_HIT_FRACTION:
      if (!(value > 0 && value <= 1)) return fail(PPF_ERR_INVALID, "ppf_workspace_set_option: hit fraction must be in (0, 1]");
      ws->hit_frac = value;
      ws->frac_known = true;
      return PPF_OK;
    case PPF_OPT_GROUP_ROUND_BUCKETS:
      if (!(value >= 0 && value <= 1e9)) return fail(PPF_ERR_INVALID, "ppf_workspace_set_option: bad bucket count");
      ws->round_buckets_cap = (int)value;
      return PPF_OK;
    case PPF_OPT_CLUSTER_SERIAL:
      ws->cluster_serial = value != 0;
      return PPF_OK;
    case PPF_OPT_ACC32:
      ws->force_acc32 = value != 0;
      return PPF_OK;
    default:
      return fail(PPF_ERR_INVALID, "ppf_workspace_set_option: unknown option %d", option);
  }
}
ppf_status ppf_workspace_enable_timing(ppf_workspace* ws, int on) {
  if (!ws) return fail(PPF_ERR_INVALID, "ppf_workspace_enable_timing: NULL");
  if (on && !ws->ev[0])
    for (auto& e : ws->ev) HIPCHK(hipEventCreate(&e));
  ws->timing = on != 0;
  return PPF_OK;
}

static ppf_status check_match_args(const ppf_model* m, const void* scene, int ns, int sstride, const void* edge, int ne,
                                   int estride, const ppf_match_params* p) {
  if (!m) return fail(PPF_ERR_NOT_TRAINED, "The model is not trained. Cannot match without training");
  if (!scene || ns <= 0 || sstride < 6 || !p) return fail(PPF_ERR_INVALID, "match: bad scene argument");
  if (edge && (ne <= 0 || estride < 6)) return fail(PPF_ERR_INVALID, "match: bad edge argument");
  if (!(p->relative_scene_sample_step <= 1 && p->relative_scene_sample_step > 0))
    return fail(PPF_ERR_INVALID, "match: relativeSceneSampleStep must be in (0, 1]");
  if (!p->presampled && !(p->relative_scene_distance > 0)) return fail(PPF_ERR_INVALID, "match: relativeSceneDistance must be > 0");
  if (p->ref_stride < 1 || p->ref_offset < 0) return fail(PPF_ERR_INVALID, "match: bad ref_offset/ref_stride");
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (dev != m->device)
    return fail(PPF_ERR_INVALID, "match: the model lives on device %d, the calling thread's current device is %d", m->device, dev);
  return PPF_OK;
}

/* A2: sample the scene (and edge) cloud into the workspace, or take the rows as they are */
static ppf_status prepare_scene(ppf_workspace* ws, const float* d_scene, int ns, int sstride, const float* d_edge, int ne,
                                int estride, const ppf_match_params* params, hipStream_t st) {
  auto load = [&](CloudDev& dst, const float* d_src, int rows, int stride) -> ppf_status {
    if (params->presampled) return dst.load_device(d_src, rows, stride, st);
    return device_sample_cloud(d_src, rows, stride, (float)params->relative_scene_distance, dst, nullptr, st);
  };
  ppf_status s = load(ws->surf, d_scene, ns, sstride);
  if (s != PPF_OK) return s;
  if (d_edge) s = load(ws->edge, d_edge, ne, estride);
  ws->has_edge = d_edge != nullptr;
  return s;
}

static ppf_status match_prepared(const ppf_model* m, ppf_workspace* ws, const ppf_match_params* params, hipStream_t st, bool retry = false);

ppf_status ppf_match_device(const ppf_model* m, ppf_workspace* ws, const float* d_scene, int ns, int sstride,
                            const float* d_edge, int ne, int estride, const ppf_match_params* params, void* stream) {
  if (!ws) return fail(PPF_ERR_INVALID, "ppf_match_device: workspace is NULL");
  if (!m) return fail(PPF_ERR_NOT_TRAINED, "The model is not trained. Cannot match without training");
  if (!have_device()) return fail(PPF_ERR_HIP, "ppf_match_device: no HIP device (this engine has no CPU fallback)");
  ppf_status s = check_match_args(m, d_scene, ns, sstride, d_edge, ne, estride, params);
  if (s != PPF_OK) return s;
  hipStream_t st = (hipStream_t)stream;
  s = prepare_scene(ws, d_scene, ns, sstride, d_edge, ne, estride, params, st);
  if (s != PPF_OK) return s;
  return match_prepared(m, ws, params, st);
}

/* store the current model's learned hit fraction (at most 16 models are remembered) */
static void workspace_remember_frac(ppf_workspace* ws) {
  if (!ws->model || !ws->frac_known) return;
  for (auto& fm : ws->frac_by_model)
    if (fm.model == ws->model) { fm.hit = ws->hit_frac; fm.run = ws->run_frac; return; }
  if (ws->frac_by_model.size() >= 16) ws->frac_by_model.erase(ws->frac_by_model.begin());
  ws->frac_by_model.push_back({ws->model, ws->hit_frac, ws->run_frac});
}

/* the workspace keeps the model alive until its next call (or its destruction): results are fetched later */
static void workspace_hold_model(ppf_workspace* ws, const ppf_model* m) {
  if (ws->model == m) return;
  ppf_model* old = ws->model;
  ws->model = const_cast<ppf_model*>(m);
  if (ws->model) ws->model->refcount.fetch_add(1);
  if (old) (void)ppf_model_release(old);
}

ppf_workspace::~ppf_workspace() {
  sync_device(device); /* buffers return to the block cache: nothing may still be using them */
  for (auto& e : ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : batch_ev)
    if (e) (void)hipEventDestroy(e);
  if (model) (void)ppf_model_release(model);
}

/* bytes of hit scratch one hit costs: raw {bucket, j} + sorted payload (alpha_s, cell) + its share of the run table */
constexpr double HIT_SCRATCH_BYTES = 8.0 + 8.0 + 2.0 + 16.0 / 6.0;

/* k_pairs<pair feature, surface-to-boundary>: same_cloud == 0 is match_S2B (the paired points come from the edge cloud) */
static void launch_pairs(const MatchArgs& va, bool darboux, hipStream_t st) {
  const dim3 grid(va.pair_chunks, va.n_ref), block(PAIR_BLOCK);
  if (darboux) {
    if (va.same_cloud) k_pairs<true, false><<<grid, block, 0, st>>>(va);
    else k_pairs<true, true><<<grid, block, 0, st>>>(va);
  } else {
    if (va.same_cloud) k_pairs<false, false><<<grid, block, 0, st>>>(va);
    else k_pairs<false, true><<<grid, block, 0, st>>>(va);
  }
}

/* everything after A2: frames -> pairs -> group -> rank -> vote -> finalize -> cluster, on the clouds held by ws.
 * Nothing here waits for the device: the hit pools are sized from ws->hit_frac (hits per scene pair, learned from the
 * previous calls); a pool that turns out too small raises a device flag, which the first accessor of the results reads
 * (workspace_finish) and answers by repeating the call with bigger pools. */
static ppf_status match_prepared(const ppf_model* m, ppf_workspace* ws, const ppf_match_params* params, hipStream_t st, bool retry) {
  ppf_status s = PPF_OK;
  const bool d_edge = ws->has_edge;
  if (ws->model != m) { /* another model: its own hit density (remembered if it has been here before) */
    workspace_remember_frac(ws);
    ws->acc32 = false;
    bool found = false;
    for (auto& fm : ws->frac_by_model)
      if (fm.model == m) { ws->hit_frac = fm.hit; ws->run_frac = fm.run; found = true; }
    if (found) ws->frac_known = true;
    else if (!ws->frac_by_model.empty()) ws->frac_known = false; /* a model this workspace has not met: count first */
  }
  workspace_hold_model(ws, m);
  HIPCHK(hipGetDevice(&ws->device));
  ws->params = *params;
  ws->stream = st;
  ws->clustered = false;
  ws->checked = false;
  ws->final_poses.clear();
  const int retries = retry ? ws->stats.n_retries : 0;
  memset(&ws->stats, 0, sizeof(ws->stats));
  ws->stats.n_retries = retries;
  const int rows = ws->surf.n;
  const int scene_step = (int)(1.0 / params->relative_scene_sample_step);
  const int n_ref_total = (rows + scene_step - 1) / scene_step;
  const int n_ref = n_ref_total > params->ref_offset ? (n_ref_total - params->ref_offset + params->ref_stride - 1) / params->ref_stride : 0;
  ws->rows = rows;
  ws->n_ref_total = n_ref_total;
  ws->n_ref = n_ref;
  ws->n_batches = 0;
  ws->stats.n_scene_sampled = rows;
  ws->stats.n_paired = d_edge ? ws->edge.n : rows;
  ws->stats.n_ref = n_ref;
  ws->pending = true;
  if (n_ref == 0) return PPF_OK;

  const int T = m->info.n_tiles;
  HIPCHK(ws->partial.reserve((size_t)n_ref * T * 2));
  HIPCHK(ws->half_edge.reserve((size_t)n_ref * T * 2));
  HIPCHK(ws->ovf_items.reserve((size_t)n_ref * T));
  const size_t n_cnt = (size_t)n_ref * T + n_ref + 7; /* cellsum | pairs | totals[2] | tally[5]: LDS operations, hits, runs, 32-bit items, votes cast twice */
  HIPCHK(ws->counters.reserve(n_cnt));
  HIPCHK(ws->votes.reserve(n_ref));
  HIPCHK(ws->raw_poses.reserve(n_ref));
  if (ws->timing) HIPCHK(hipEventRecord(ws->ev[0], st));
  HIPCHK(hipMemsetAsync(ws->counters.p, 0, n_cnt * sizeof(unsigned long long), st));
  HIPCHK(hipMemsetAsync(ws->ovf_items.p, 0, (size_t)n_ref * T * sizeof(uint32_t), st));

  MatchArgs va;
  memset(&va, 0, sizeof(va));
  va.surf = ws->surf.view();
  va.paired = d_edge ? ws->edge.view() : ws->surf.view();
  va.same_cloud = d_edge ? 0 : 1;
  va.scene_step = scene_step; va.ref_offset = params->ref_offset; va.ref_stride = params->ref_stride;
  va.slotmap = m->slotmap.p; va.slot_mask = m->info.slots - 1;
  va.key_lut = m->key_lut.p; va.kd = m->kd;
  va.bucket_off = m->bucket_off.p; va.n_buckets = (int)m->info.n_buckets;
  va.records = m->records.p;
  va.n_tiles = T; va.tile_refs = m->info.tile_refs; va.num_angles = m->info.num_angles; va.n_model = m->info.n_ref;
  va.angle_step = m->info.angle_step; va.dist_step = m->info.distance_step;
  va.partial = ws->partial.p;
  va.edge = ws->half_edge.p;
  va.ovf_items = ws->ovf_items.p;
  va.cellsum = ws->counters.p;
  va.pairs = ws->counters.p + (size_t)n_ref * T;
  va.tally = ws->counters.p + (size_t)n_ref * T + n_ref + 2;
  va.acc_dump = ws->acc_dump;
  va.bucket_total = m->bucket_total.p;
  va.bucket_mid = m->bucket_mid.p;
  va.key_exact = m->params.key_equality == PPF_KEY_EXACT;
  const bool acc32_all = ws->acc32 || ws->force_acc32; /* otherwise: 16-bit cells, then 32-bit cells for the (reference point, tile)s that overflowed */
  const bool darboux = m->params.feature == PPF_FEATURE_DARBOUX;
  va.pair_radius = params->pair_radius;
  va.agg_min_hits = (params->vote_mode == PPF_VOTE_DIRECT || params->alpha_range_2pi || m->info.num_angles > AGG_MAX_ANGLES) ? 0 : PPF_AGG_MIN_HITS;
  const int n_paired = va.paired.n;
  va.pair_chunks = (n_paired + PAIR_BLOCK * PAIRS_PER_THREAD - 1) / (PAIR_BLOCK * PAIRS_PER_THREAD);
  const uint32_t round_cap = ws->round_buckets_cap > 0 ? (uint32_t)std::min(ws->round_buckets_cap, GROUP_MAX_BUCKETS) : (uint32_t)GROUP_MAX_BUCKETS;
  va.n_rounds = std::max(1, (int)((m->info.n_buckets + round_cap - 1) / round_cap));
  va.round_buckets = (int)std::min<uint32_t>(std::max<uint32_t>(m->info.n_buckets, 1u), round_cap);

  HIPCHK(ws->cursors.reserve(CUR_WORDS));
  va.cursors = ws->cursors.p;
  if (!ws->frac_known) {
    /* Cold workspace: nothing is known about this scene's hit density, so the pair kernel first only counts its hits
     * (same arithmetic, nothing stored) and the pools are sized from the exact number.  Costs one extra pair pass and
     * one wait for the device, once: later calls size their pools from what the previous call saw. */
    HIPCHK(hipMemsetAsync(ws->cursors.p, 0, CUR_WORDS * sizeof(uint32_t), st));
    va.count_only = 1;
    va.stripe_bits = 6;
    for (int base = 0; base < n_ref; base += 32768) {
      va.ref_base = base;
      va.n_ref = std::min(32768, n_ref - base);
      launch_pairs(va, darboux, st);
      HIPCHK(hipGetLastError());
    }
    va.count_only = 0;
    std::vector<uint32_t> cw(CUR_SORTED);
    HIPCHK(hipMemcpyAsync(cw.data(), ws->cursors.p, cw.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    unsigned long long hits = 0;
    for (int sidx = 0; sidx < POOL_STRIPES; sidx++)
      hits += (unsigned long long)cw[sidx * CUR_STRIDE] | ((unsigned long long)cw[sidx * CUR_STRIDE + 1] << 32);
    const double pairs_total = (double)n_ref * (double)n_paired;
    ws->hit_frac = std::min(1.0, std::max(1e-3, 1.06 * (double)hits / std::max(1.0, pairs_total)));
    ws->frac_known = true;
  }
  /* batch of reference points: its expected hits fit the scratch budget (and 32-bit pool offsets) */
  const double frac = std::min(1.0, std::max(ws->hit_frac, 1e-3));
  const double hits_per_ref = std::max(64.0, frac * (double)n_paired);
  int batch = (int)std::min<double>((double)n_ref, std::max(1.0, (double)HIT_BYTES_BUDGET / (hits_per_ref * HIT_SCRATCH_BYTES)));
  batch = (int)std::min<double>((double)batch, std::max(1.0, 2.0e9 / hits_per_ref));
  batch = std::min(batch, 32768); /* grid.y of k_pairs */
  const bool worst_case = frac >= 1.0;
  const double est = hits_per_ref * (double)batch;
  /* a stripe receives whole workgroups of up to PAIR_BLOCK*PAIRS_PER_THREAD hits; at worst-case size every workgroup of
   * the batch could be full and land anywhere, otherwise 4 % + two workgroups of slack over an even share */
  const uint32_t wg_hits = PAIR_BLOCK * PAIRS_PER_THREAD;
  int stripe_bits = 6; /* 64 stripes, fewer while a stripe would average over less than 512 workgroups */
  while (stripe_bits > 0 && ((size_t)batch * va.pair_chunks >> stripe_bits) < 512) stripe_bits--;
  if (worst_case) stripe_bits = 0; /* one stripe that holds every pair of the batch: nothing can overflow */
  const uint32_t n_stripes = 1u << stripe_bits;
  const uint32_t stripe_cap = worst_case ? (uint32_t)std::min<double>(4.0e9 / n_stripes, (double)batch * va.pair_chunks * wg_hits)
                                         : (uint32_t)(est / n_stripes * 1.04) + 2 * wg_hits;
  const uint32_t sorted_cap = (uint32_t)std::min(4.0e9, est + 4096.0);
  const uint32_t run_cap = worst_case ? sorted_cap : (uint32_t)std::min<double>((double)sorted_cap, std::max(est * std::min(1.0, ws->run_frac), 64.0 * batch) + 1024.0);
  HIPCHK(ws->frames.reserve((size_t)batch * 12));
  HIPCHK(ws->raw.fit((size_t)stripe_cap * n_stripes));
  HIPCHK(ws->chunk_desc.reserve((size_t)batch * va.pair_chunks));
  HIPCHK(ws->hit_count.reserve(batch));
  HIPCHK(ws->s_a64.fit(sorted_cap));
  HIPCHK(ws->s_cell.fit(sorted_cap));
  HIPCHK(ws->runs.fit(run_cap));
  HIPCHK(ws->run_blocks.reserve((size_t)batch * va.n_rounds));
  HIPCHK(ws->work.reserve(batch));
  HIPCHK(ws->perm.reserve(batch));
  HIPCHK(ws->perm_group.reserve(batch));
  HIPCHK(ws->ovf_list.reserve((size_t)batch * T));
  va.ovf_list = ws->ovf_list.p;
  ws->stats.scratch_bytes = ws->frames.bytes() + ws->raw.bytes() + ws->cursors.bytes() + ws->chunk_desc.bytes() + ws->hit_count.bytes() +
                            ws->s_a64.bytes() + ws->s_cell.bytes() + ws->runs.bytes() + ws->run_blocks.bytes() +
                            ws->work.bytes() + ws->perm.bytes() + ws->perm_group.bytes();
  va.frames = ws->frames.p;
  va.raw = ws->raw.p; va.stripe_cap = stripe_cap; va.stripe_bits = stripe_bits;
  va.chunk_desc = ws->chunk_desc.p; va.hit_count = ws->hit_count.p;
  va.s_a64 = ws->s_a64.p; va.s_cell = ws->s_cell.p; va.sorted_cap = sorted_cap;
  va.runs = ws->runs.p; va.run_cap = run_cap; va.run_blocks = ws->run_blocks.p;
  va.work = ws->work.p; va.perm = ws->perm.p; va.perm_group = ws->perm_group.p;

  const size_t lds = VOTE_LDS_FIXED + (size_t)vote_lds_words(m->info.tile_refs, m->info.num_angles) * 4;
  if (lds > (size_t)LDS_BYTES) return fail(PPF_ERR_INVALID, "match: model tile of %d reference points does not fit the LDS accumulator", m->info.tile_refs);
  /* k_group's dynamic LDS: one counter per bucket of a round, the prefix of the pool pieces */
  const size_t group_lds = (size_t)((va.round_buckets + 1) & ~1) * sizeof(uint32_t) + (size_t)((va.pair_chunks + 2) & ~1) * sizeof(uint32_t);
  if (group_lds + 2048 > (size_t)LDS_BYTES) return fail(PPF_ERR_INVALID, "match: %d paired points are more than one call can group", n_paired);
  static std::once_flag once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(once, [] {
    const void* votes[4] = {reinterpret_cast<const void*>(&k_vote<false, false>), reinterpret_cast<const void*>(&k_vote<true, false>),
                            reinterpret_cast<const void*>(&k_vote<false, true>), reinterpret_cast<const void*>(&k_vote<true, true>)};
    for (const void* f : votes)
      if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr_err == hipSuccess)
      attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_group), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES - 1024);
  });
  HIPCHK(attr_err);
  const int n_batches = (n_ref + batch - 1) / batch;
  ws->n_batches = n_batches;
  ws->stats.n_batches = n_batches;
  if (ws->timing)
    while (ws->batch_ev.size() < (size_t)n_batches * 4) {
      hipEvent_t e = nullptr;
      HIPCHK(hipEventCreate(&e));
      ws->batch_ev.push_back(e);
    }
  HIPCHK(hipMemsetAsync(ws->cursors.p, 0, CUR_WORDS * sizeof(uint32_t), st));
  for (int bi = 0; bi < n_batches; bi++) {
    const int base = bi * batch;
    va.ref_base = base;
    va.n_ref = std::min(batch, n_ref - base);
    if (bi) HIPCHK(hipMemsetAsync(ws->cursors.p, 0, CUR_OVERFLOW * sizeof(uint32_t), st)); /* the overflow word lives on */
    k_frames<<<dim3((va.n_ref + 63) / 64), dim3(64), 0, st>>>(va);
    HIPCHK(hipGetLastError());
    if (ws->timing) HIPCHK(hipEventRecord(ws->batch_ev[bi * 4 + 0], st));
    launch_pairs(va, darboux, st);
    HIPCHK(hipGetLastError());
    if (ws->timing) HIPCHK(hipEventRecord(ws->batch_ev[bi * 4 + 1], st));
    k_ref_hits<<<dim3((va.n_ref + 255) / 256), dim3(256), 0, st>>>(va);
    /* k_group takes the reference points with the most hits first */
    k_rank<<<dim3((va.n_ref + RANK_KEYS - 1) / RANK_KEYS), dim3(256), 0, st>>>(va.hit_count, va.n_ref, nullptr, ws->perm_group.p, nullptr);
    k_group<<<dim3(va.n_ref), dim3(GROUP_BLOCK), group_lds, st>>>(va);
    HIPCHK(hipGetLastError());
    /* k_vote takes the reference points that will cast the most votes first */
    k_rank<<<dim3((va.n_ref + RANK_KEYS - 1) / RANK_KEYS), dim3(256), 0, st>>>(va.work, va.n_ref, nullptr, va.perm, nullptr);
    HIPCHK(hipGetLastError());
    if (ws->timing) HIPCHK(hipEventRecord(ws->batch_ev[bi * 4 + 2], st));
    const dim3 grid16((unsigned)((size_t)va.n_ref * T)), grid32((unsigned)((size_t)va.n_ref * T * 2));
    if (!acc32_all) {
      va.acc32 = 0;
      if (params->alpha_range_2pi) k_vote<true, false><<<grid16, dim3(VOTE_BLOCK), lds, st>>>(va);
      else k_vote<false, false><<<grid16, dim3(VOTE_BLOCK), lds, st>>>(va);
      HIPCHK(hipGetLastError());
    }
    va.acc32 = acc32_all ? 1 : 2; /* 2: the (reference point, tile)s the 16-bit launch listed, over a grid that does not depend on their number */
    const dim3 g32 = acc32_all ? grid32 : dim3(std::min(grid32.x, 1024u));
    if (params->alpha_range_2pi) k_vote<true, true><<<g32, dim3(VOTE_BLOCK), lds, st>>>(va);
    else k_vote<false, true><<<g32, dim3(VOTE_BLOCK), lds, st>>>(va);
    va.acc32 = 0;
    HIPCHK(hipGetLastError());
    if (ws->timing) HIPCHK(hipEventRecord(ws->batch_ev[bi * 4 + 3], st));
  }

  FinalArgs fa;
  fa.surf = ws->surf.view(); fa.model = m->cloud.view();
  fa.scene_step = scene_step; fa.ref_offset = params->ref_offset; fa.ref_stride = params->ref_stride; fa.n_ref = n_ref;
  fa.n_tiles = T; fa.tile_refs = m->info.tile_refs; fa.num_angles = m->info.num_angles;
  fa.alpha_2pi = params->alpha_range_2pi != 0;
  fa.acc32 = acc32_all ? 1 : 0; fa.ovf_items = ws->ovf_items.p; fa.edge = ws->half_edge.p;
  fa.partial = ws->partial.p; fa.cellsum = va.cellsum; fa.pairs = va.pairs;
  fa.votes = ws->votes.p; fa.poses = ws->raw_poses.p;
  fa.totals = ws->counters.p + (size_t)n_ref * T + n_ref;
  k_finalize<<<dim3((n_ref + 63) / 64), dim3(64), 0, st>>>(fa);
  HIPCHK(hipGetLastError());
  if (!params->skip_clustering) {
    double pos, rot;
    resolve_thresholds(m, params, &pos, &rot);
    /* the reference clusters sampled.rows / sceneSamplingStep poses (integer division: the lowest-voted
     * pose is dropped when the stride does not divide the row count); a shard clusters its own share */
    const int num = (params->ref_stride == 1 && params->ref_offset == 0) ? rows / scene_step : n_ref;
    s = enqueue_cluster(ws, ws->raw_poses.p, n_ref, num, pos, rot, params->use_weighted_avg != 0, st, params->rot_metric_relative != 0);
    if (s != PPF_OK) return s;
    ws->clustered = true;
  }
  if (ws->timing) HIPCHK(hipEventRecord(ws->ev[1], st));
  return PPF_OK;
}

/* Wait for the workspace's pending call and make sure it ran with big enough hit pools: when a pool overflowed (device
 * flag), the call is repeated on the same stream with a doubled estimate, until it fits (at hit_frac == 1 the pools
 * hold every scene pair).  Also reads the counters and learns hit_frac for the next call. */
static ppf_status workspace_finish(ppf_workspace* ws) {
  if (!ws->pending) return fail(PPF_ERR_INVALID, "no call in this workspace");
  HIPCHK(hipStreamSynchronize(ws->stream));
  if (ws->checked || ws->n_ref == 0) { ws->checked = true; return PPF_OK; }
  for (;;) {
    const int T = ws->model->info.n_tiles;
    unsigned long long tot[7];
    uint32_t ovf = 0;
    HIPCHK(hipMemcpy(tot, ws->counters.p + (size_t)ws->n_ref * T + ws->n_ref, sizeof(tot), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&ovf, ws->cursors.p + CUR_OVERFLOW, sizeof(ovf), hipMemcpyDeviceToHost));
    if (!ovf) {
      ws->stats.n_votes = tot[0];
      ws->stats.n_pairs = tot[1];
      ws->stats.n_lds_atomics = tot[2];
      ws->stats.n_hits = tot[3];
      ws->stats.n_acc32_items = tot[5];
      /* The 16-bit launch is about 12 % cheaper than the 32-bit one, and what it flags is voted twice: a scene that casts more
       * than a tenth of its votes in (reference point, tile)s that overflow goes straight to 32-bit cells from now on. */
      if (!ws->acc32 && (double)tot[6] > 0.10 * (double)tot[0]) ws->acc32 = true;
      if (tot[1]) ws->hit_frac = std::min(1.0, std::max(1e-3, 1.06 * (double)tot[3] / (double)tot[1]));
      if (tot[3]) ws->run_frac = std::min(1.0, std::max(0.02, 1.10 * (double)tot[4] / (double)tot[3]));
      ws->frac_known = true;
      if (ws->clustered) {
        uint32_t nf = 0;
        HIPCHK(hipMemcpy(&nf, ws->cl_u32.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
        ws->stats.n_poses = (int)nf;
      }
      if (ws->timing) {
        float pr = 0, gr = 0, vo = 0;
        for (int b = 0; b < ws->n_batches; b++) {
          float t0 = 0, t1 = 0, t2 = 0;
          HIPCHK(hipEventElapsedTime(&t0, ws->batch_ev[b * 4 + 0], ws->batch_ev[b * 4 + 1]));
          HIPCHK(hipEventElapsedTime(&t1, ws->batch_ev[b * 4 + 1], ws->batch_ev[b * 4 + 2]));
          HIPCHK(hipEventElapsedTime(&t2, ws->batch_ev[b * 4 + 2], ws->batch_ev[b * 4 + 3]));
          pr += t0; gr += t1; vo += t2;
        }
        ws->stats.ms_pair_kernel = pr; ws->stats.ms_group_kernel = gr; ws->stats.ms_vote_kernel = vo;
        HIPCHK(hipEventElapsedTime(&ws->stats.ms_total_device, ws->ev[0], ws->ev[1]));
      }
      ws->checked = true;
      return PPF_OK;
    }
    if (ws->hit_frac >= 1.0) return fail(PPF_ERR_CAPACITY, "match: hit pools overflowed at worst-case size (flags %u)", ovf);
    if (ovf & 3u) ws->hit_frac = std::min(1.0, ws->hit_frac * 2.0); /* raw or sorted hit pool */
    if (ovf & 4u) ws->run_frac = std::min(1.0, ws->run_frac * 2.0); /* run table */
    ws->stats.n_retries++;
    const ppf_match_params p = ws->params;
    ppf_model* m = ws->model;
    ppf_status s = match_prepared(m, ws, &p, ws->stream, true);
    if (s != PPF_OK) return s;
    HIPCHK(hipStreamSynchronize(ws->stream));
  }
}

ppf_status ppf_workspace_results(ppf_workspace* ws, ppf_vote* votes, ppf_pose* raw_poses, int cap_ref, int* n_ref,
                                 ppf_pose* poses, int cap_poses, int* n_poses, ppf_match_stats* stats) {
  if (!ws || !ws->pending) return fail(PPF_ERR_INVALID, "ppf_workspace_results: no call in this workspace");
  ppf_status sf = workspace_finish(ws);
  if (sf != PPF_OK) return sf;
  const int nr = ws->n_ref;
  if (n_ref) *n_ref = nr;
  if ((votes || raw_poses) && cap_ref < nr) return fail(PPF_ERR_CAPACITY, "ppf_workspace_results: need room for %d reference points", nr);
  if (nr > 0) {
    if (votes) HIPCHK(hipMemcpy(votes, ws->votes.p, (size_t)nr * sizeof(ppf_vote), hipMemcpyDeviceToHost));
    if (raw_poses) HIPCHK(hipMemcpy(raw_poses, ws->raw_poses.p, (size_t)nr * sizeof(ppf_pose), hipMemcpyDeviceToHost));
  }
  if ((poses || n_poses) && ws->clustered) {
    if (ws->final_poses.empty() && nr > 0) {
      uint32_t nf = 0;
      HIPCHK(hipMemcpy(&nf, ws->cl_u32.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
      ws->final_poses.resize(nf);
      if (nf) HIPCHK(hipMemcpy(ws->final_poses.data(), ws->d_final.p, (size_t)nf * sizeof(ppf_pose), hipMemcpyDeviceToHost));
    }
    ws->stats.n_poses = (int)ws->final_poses.size();
    if (n_poses) *n_poses = (int)ws->final_poses.size();
    if (poses) {
      if (cap_poses < (int)ws->final_poses.size())
        return fail(PPF_ERR_CAPACITY, "ppf_workspace_results: need room for %d poses", (int)ws->final_poses.size());
      memcpy(poses, ws->final_poses.data(), ws->final_poses.size() * sizeof(ppf_pose));
    }
  } else if (n_poses) {
    *n_poses = 0;
  }
  if (stats) *stats = ws->stats;
  return PPF_OK;
}

static ppf_status run_host(const ppf_model* m, const float* scene, int ns, int sstride, const float* edge, int ne,
                           int estride, const ppf_match_params* params, ppf_workspace* ws);

ppf_status ppf_workspace_ref_counters(ppf_workspace* ws, uint64_t* votes_per_ref, uint64_t* pairs_per_ref, int cap) {
  if (!ws || !ws->pending) return fail(PPF_ERR_INVALID, "ppf_workspace_ref_counters: no call in this workspace");
  ppf_status sf = workspace_finish(ws);
  if (sf != PPF_OK) return sf;
  const int nr = ws->n_ref;
  if (cap < nr) return fail(PPF_ERR_CAPACITY, "ppf_workspace_ref_counters: need room for %d reference points", nr);
  if (nr == 0) return PPF_OK;
  const int T = ws->model->info.n_tiles;
  std::vector<unsigned long long> h((size_t)nr * T + nr);
  HIPCHK(hipMemcpy(h.data(), ws->counters.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  for (int r = 0; r < nr; r++) {
    unsigned long long v = 0;
    for (int t = 0; t < T; t++) v += h[(size_t)r * T + t];
    if (votes_per_ref) votes_per_ref[r] = v;
    if (pairs_per_ref) pairs_per_ref[r] = h[(size_t)nr * T + r];
  }
  return PPF_OK;
}

ppf_status ppf_debug_accumulators(const ppf_model* m, const float* scene, int ns, int sstride, const float* edge, int ne,
                                  int estride, const ppf_match_params* params, uint32_t* acc, size_t cap_words,
                                  int* n_ref) {
  if (!m || !acc || !params) return fail(PPF_ERR_INVALID, "ppf_debug_accumulators: bad argument");
  ppf_match_params p = *params;
  const size_t per_ref = (size_t)m->info.n_ref * m->info.num_angles;
  const int scene_step = (int)(1.0 / p.relative_scene_sample_step);
  if (!p.presampled) return fail(PPF_ERR_INVALID, "ppf_debug_accumulators: presampled clouds only");
  const int n_ref_total = (ns + scene_step - 1) / scene_step;
  const int nr = n_ref_total > p.ref_offset ? (n_ref_total - p.ref_offset + p.ref_stride - 1) / p.ref_stride : 0;
  if (cap_words < per_ref * nr) return fail(PPF_ERR_CAPACITY, "ppf_debug_accumulators: need %zu words", per_ref * nr);
  DevBuf<uint32_t> dump;
  HIPCHK(dump.reserve(std::max<size_t>(per_ref * nr, 1)));
  HIPCHK(hipMemset(dump.p, 0, per_ref * nr * sizeof(uint32_t)));
  ppf_workspace ws;
  ws.acc_dump = dump.p;
  p.skip_clustering = 1;
  ppf_status s = run_host(m, scene, ns, sstride, edge, ne, estride, &p, &ws);
  if (s != PPF_OK) return s;
  HIPCHK(hipMemcpy(acc, dump.p, per_ref * nr * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (n_ref) *n_ref = nr;
  return PPF_OK;
}

/* the block cache's size class for a request (host only): what DevBuf is granted for `bytes` */
size_t ppf_debug_block_size(size_t bytes) {
  return DevPool::class_size(DevPool::class_of(std::max<size_t>(bytes, 256)));
}

ppf_status ppf_debug_device_math(int fn, const double* x, const double* y, double* out, int n) {
  if (!x || !out || n <= 0 || fn < 0 || fn > 5) return fail(PPF_ERR_INVALID, "ppf_debug_device_math: bad argument");
  if (!have_device()) return fail(PPF_ERR_HIP, "ppf_debug_device_math: no HIP device");
  DevBuf<double> dx, dy, dout;
  HIPCHK(dx.reserve(n));
  HIPCHK(dy.reserve(n));
  HIPCHK(dout.reserve(n));
  HIPCHK(hipMemcpy(dx.p, x, (size_t)n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dy.p, y ? y : x, (size_t)n * 8, hipMemcpyHostToDevice));
  k_debug_math<<<dim3((n + 255) / 256), dim3(256)>>>(fn, dx.p, dy.p, dout.p, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, dout.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  return PPF_OK;
}

ppf_status ppf_workspace_device_poses(ppf_workspace* ws, void** d_raw_poses, int* n_ref) {
  if (!ws || !ws->pending || !d_raw_poses) return fail(PPF_ERR_INVALID, "ppf_workspace_device_poses: bad argument");
  *d_raw_poses = ws->raw_poses.p;
  if (n_ref) *n_ref = ws->n_ref;
  return PPF_OK;
}

/* copy `cap` pose records to dst: the first min(n, cap) from src, zeros after them (num_votes == 0 marks an empty row);
 * n comes from the device (n_dev) when given.  Optionally also saves the count, the overflow flag of the call and its
 * four 64-bit totals (votes, pairs, LDS operations, hits) next to the block: what a batch needs per (crop, model). */
__global__ __launch_bounds__(256) void k_pose_block(const ppf_pose* __restrict__ src, const uint32_t* __restrict__ n_dev, int n_host,
                                                    ppf_pose* __restrict__ dst, int cap, uint32_t* __restrict__ meta_out,
                                                    const uint32_t* __restrict__ flag_in, unsigned long long* __restrict__ tot_out,
                                                    const unsigned long long* __restrict__ tot_in) {
  constexpr int W = (int)(sizeof(ppf_pose) / 8);
  const int n = n_dev ? (int)*n_dev : n_host;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap * W) {
    const int row = i / W;
    const unsigned long long* s64 = reinterpret_cast<const unsigned long long*>(src);
    reinterpret_cast<unsigned long long*>(dst)[i] = row < n ? s64[i] : 0ull;
  }
  if (i == 0 && meta_out) { meta_out[0] = (uint32_t)n; meta_out[1] = flag_in ? *flag_in : 0u; }
  if (i < 5 && tot_out && tot_in) tot_out[i] = tot_in[i]; /* votes, pairs, LDS operations, hits, runs */
}

ppf_status ppf_workspace_copy_top_poses(ppf_workspace* ws, void* d_dst, int k, void* stream) {
  if (!ws || !ws->pending || !d_dst || k <= 0) return fail(PPF_ERR_INVALID, "ppf_workspace_copy_top_poses: bad argument");
  if (!ws->clustered || ws->n_ref == 0) {
    HIPCHK(hipMemsetAsync(d_dst, 0, (size_t)k * sizeof(ppf_pose), (hipStream_t)stream));
    return PPF_OK;
  }
  const int words = k * (int)(sizeof(ppf_pose) / 8);
  k_pose_block<<<dim3((words + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(ws->d_final.p, ws->cl_u32.p, 0, (ppf_pose*)d_dst, k, nullptr,
                                                                                 nullptr, nullptr, nullptr);
  HIPCHK(hipGetLastError());
  return PPF_OK;
}

ppf_status ppf_workspace_copy_raw_poses(ppf_workspace* ws, void* d_dst, int cap, void* stream) {
  if (!ws || !ws->pending || !d_dst || cap <= 0) return fail(PPF_ERR_INVALID, "ppf_workspace_copy_raw_poses: bad argument");
  if (cap < ws->n_ref) return fail(PPF_ERR_CAPACITY, "ppf_workspace_copy_raw_poses: need room for %d reference points", ws->n_ref);
  if (ws->n_ref == 0) {
    HIPCHK(hipMemsetAsync(d_dst, 0, (size_t)cap * sizeof(ppf_pose), (hipStream_t)stream));
    return PPF_OK;
  }
  const int words = cap * (int)(sizeof(ppf_pose) / 8);
  k_pose_block<<<dim3((words + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(ws->raw_poses.p, nullptr, ws->n_ref, (ppf_pose*)d_dst, cap,
                                                                                 nullptr, nullptr, nullptr, nullptr);
  HIPCHK(hipGetLastError());
  return PPF_OK;
}

ppf_status ppf_cluster_poses_device(const ppf_model* m, ppf_workspace* ws, const void* d_in, int n, int num_poses,
                                    const ppf_match_params* params, void* stream) {
  if (!m) return fail(PPF_ERR_NOT_TRAINED, "ppf_cluster_poses_device: model is NULL");
  if (!ws || (!d_in && n > 0) || n < 0 || !params) return fail(PPF_ERR_INVALID, "ppf_cluster_poses_device: bad argument");
  if (!have_device()) return fail(PPF_ERR_HIP, "ppf_cluster_poses_device: no HIP device (this engine has no CPU fallback)");
  workspace_hold_model(ws, m);
  HIPCHK(hipGetDevice(&ws->device));
  ws->params = *params;
  ws->stream = (hipStream_t)stream;
  ws->final_poses.clear();
  memset(&ws->stats, 0, sizeof(ws->stats));
  ws->n_ref = 0; ws->n_ref_total = 0; ws->n_batches = 0;
  ws->pending = true; ws->checked = true; /* no hit pools involved */
  ws->clustered = false;
  if (n == 0) return PPF_OK;
  double pos, rot;
  resolve_thresholds(m, params, &pos, &rot);
  ppf_status s = enqueue_cluster(ws, (const ppf_pose*)d_in, n, num_poses, pos, rot, params->use_weighted_avg != 0, (hipStream_t)stream,
                                 params->rot_metric_relative != 0);
  if (s != PPF_OK) return s;
  ws->clustered = true;
  ws->n_ref = n; /* d_final / cl_u32 hold the clusters; results come back through ppf_workspace_results(poses) / _copy_top_poses */
  ws->stats.n_ref = n;
  return PPF_OK;
}

ppf_status ppf_cluster_poses(const ppf_model* m, const ppf_pose* in, int n, int num_poses,
                             const ppf_match_params* params, ppf_pose* out, int cap, int* n_out) {
  if (!m) return fail(PPF_ERR_NOT_TRAINED, "ppf_cluster_poses: model is NULL");
  if ((!in && n > 0) || n < 0 || !params || !n_out) return fail(PPF_ERR_INVALID, "ppf_cluster_poses: bad argument");
  *n_out = 0;
  if (n == 0) return PPF_OK;
  if (!have_device()) return fail(PPF_ERR_HIP, "ppf_cluster_poses: no HIP device (this engine has no CPU fallback)");
  ppf_workspace ws;
  DevBuf<ppf_pose> d_in;
  HIPCHK(d_in.reserve(n));
  HIPCHK(hipMemcpy(d_in.p, in, (size_t)n * sizeof(ppf_pose), hipMemcpyHostToDevice));
  double pos, rot;
  resolve_thresholds(m, params, &pos, &rot);
  ppf_status s = enqueue_cluster(&ws, d_in.p, n, num_poses, pos, rot, params->use_weighted_avg != 0, nullptr, params->rot_metric_relative != 0);
  if (s != PPF_OK) return s;
  HIPCHK(hipStreamSynchronize(nullptr));
  uint32_t nf = 0;
  HIPCHK(hipMemcpy(&nf, ws.cl_u32.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
  *n_out = (int)nf;
  if (out) {
    if (cap < (int)nf) return fail(PPF_ERR_CAPACITY, "ppf_cluster_poses: need room for %d poses", (int)nf);
    if (nf) HIPCHK(hipMemcpy(out, ws.d_final.p, (size_t)nf * sizeof(ppf_pose), hipMemcpyDeviceToHost));
  }
  return PPF_OK;
}

/* host-buffer conveniences: upload, run on the default stream, download */
static ppf_status run_host(const ppf_model* m, const float* scene, int ns, int sstride, const float* edge, int ne,
                           int estride, const ppf_match_params* params, ppf_workspace* ws) {
  if (!m) return fail(PPF_ERR_NOT_TRAINED, "The model is not trained. Cannot match without training");
  if (!have_device()) return fail(PPF_ERR_HIP, "match: no HIP device (this engine has no CPU fallback)");
  ppf_status s = check_match_args(m, scene, ns, sstride, edge, ne, estride, params);
  if (s != PPF_OK) return s;
  DevBuf<float> d_scene, d_edge;
  HIPCHK(d_scene.reserve((size_t)ns * sstride));
  HIPCHK(hipMemcpy(d_scene.p, scene, (size_t)ns * sstride * sizeof(float), hipMemcpyHostToDevice));
  if (edge) {
    HIPCHK(d_edge.reserve((size_t)ne * estride));
    HIPCHK(hipMemcpy(d_edge.p, edge, (size_t)ne * estride * sizeof(float), hipMemcpyHostToDevice));
  }
  s = ppf_match_device(m, ws, d_scene.p, ns, sstride, edge ? d_edge.p : nullptr, ne, estride, params, nullptr);
  if (s != PPF_OK) return s;
  HIPCHK(hipStreamSynchronize(nullptr));
  return PPF_OK;
}

ppf_status ppf_match(const ppf_model* m, const float* scene, int ns, int sstride, const float* edge, int ne,
                     int estride, const ppf_match_params* params, ppf_pose* out, int cap, int* n_out) {
  if (!n_out) return fail(PPF_ERR_INVALID, "ppf_match: n_out is NULL");
  *n_out = 0;
  ppf_workspace ws;
  ppf_status s = run_host(m, scene, ns, sstride, edge, ne, estride, params, &ws);
  if (s == PPF_OK) s = ppf_workspace_results(&ws, nullptr, nullptr, 0, nullptr, out, cap, n_out, nullptr);
  return s;
}

/* ---- many crops x many models (BASELINE config C5) ------------------------------------------------------------------
 * A batch context owns `lanes` (stream, workspace) pairs.  Crop c goes to lane c mod lanes: its rows are staged through
 * pinned memory (host scenes), uploaded and sampled once, then matched against every model back to back on the lane's
 * stream; after each match a small kernel saves the best `cap` clustered poses, their count, the hit-pool flag and the
 * counters into the batch's device block, so nothing waits for the host between matches.  One synchronisation per lane
 * at the end, one read-back of the block.  A match whose hit pools were too small (flag) is repeated afterwards. */
struct ppf_batch {
  int lanes = 0;
  int device = 0;
  std::vector<ppf_workspace*> ws;
  std::vector<hipStream_t> streams;
  std::vector<float*> pinned;        /* 2 staging buffers per lane */
  std::vector<size_t> pinned_cap;
  std::vector<hipEvent_t> pinned_ev; /* upload from that staging buffer finished */
  std::vector<DevBuf<float>*> d_scene;
  DevBuf<ppf_pose> d_out;
  DevBuf<uint32_t> d_meta;
  DevBuf<unsigned long long> d_tot;
  int last_records = 0;
};

ppf_status ppf_batch_create(int lanes, ppf_batch** out) {
  if (!out || lanes < 1 || lanes > 64) return fail(PPF_ERR_INVALID, "ppf_batch_create: bad argument");
  *out = nullptr;
  if (!have_device()) return fail(PPF_ERR_HIP, "ppf_batch_create: no HIP device (this engine has no CPU fallback)");
  std::unique_ptr<ppf_batch> b(new (std::nothrow) ppf_batch());
  if (!b) return fail(PPF_ERR_NOMEM, "ppf_batch_create: out of memory");
  HIPCHK(hipGetDevice(&b->device));
  b->lanes = lanes;
  for (int l = 0; l < lanes; l++) {
    hipStream_t st = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e != hipSuccess) { (void)ppf_batch_destroy(b.release()); return fail(PPF_ERR_HIP, "ppf_batch_create: %s", hipGetErrorString(e)); }
    b->streams.push_back(st);
    b->ws.push_back(new ppf_workspace());
    b->d_scene.push_back(new DevBuf<float>());
    for (int k = 0; k < 2; k++) {
      hipEvent_t ev = nullptr;
      e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
      b->pinned.push_back(nullptr); b->pinned_cap.push_back(0); b->pinned_ev.push_back(ev);
      if (e != hipSuccess) { (void)ppf_batch_destroy(b.release()); return fail(PPF_ERR_HIP, "ppf_batch_create: %s", hipGetErrorString(e)); }
    }
  }
  *out = b.release();
  return PPF_OK;
}

ppf_status ppf_batch_destroy(ppf_batch* b) {
  if (!b) return PPF_OK;
  sync_device(b->device);
  for (auto* w : b->ws) delete w;
  for (auto* d : b->d_scene) delete d;
  for (auto st : b->streams) if (st) (void)hipStreamDestroy(st);
  for (auto p : b->pinned) if (p) (void)hipHostFree(p);
  for (auto e : b->pinned_ev) if (e) (void)hipEventDestroy(e);
  delete b;
  return PPF_OK;
}

ppf_status ppf_batch_run(ppf_batch* b, const ppf_model* const* models, int n_models, const float* const* scenes, const int* ns,
                         int sstride, int n_scenes, int scenes_on_device, const ppf_match_params* params, ppf_pose* out, int cap,
                         int* n_out, ppf_batch_stats* stats) {
  if (!b || !models || n_models <= 0 || !scenes || !ns || n_scenes <= 0 || !params || cap <= 0)
    return fail(PPF_ERR_INVALID, "ppf_batch_run: bad argument");
  for (int k = 0; k < n_models; k++)
    if (!models[k]) return fail(PPF_ERR_NOT_TRAINED, "ppf_batch_run: model %d is not trained", k);
  ppf_match_params p = *params;
  if (p.skip_clustering) return fail(PPF_ERR_INVALID, "ppf_batch_run: a batch returns clustered poses");
  for (int c = 0; c < n_scenes; c++) {
    ppf_status s = check_match_args(models[0], scenes[c], ns[c], sstride, nullptr, 0, 6, &p);
    if (s != PPF_OK) return s;
  }
  const auto t_start = std::chrono::steady_clock::now();
  const size_t n_match = (size_t)n_scenes * n_models;
  HIPCHK(b->d_out.reserve(n_match * cap));
  HIPCHK(b->d_meta.reserve(n_match * 2));
  HIPCHK(b->d_tot.reserve(n_match * 8));
  b->last_records = (int)(n_match * cap);
  const int words = cap * (int)(sizeof(ppf_pose) / 8);

  /* one (crop, model) match on a lane, results saved to the block */
  auto enqueue_pair = [&](int lane, int c, int k) -> ppf_status {
    ppf_workspace* ws = b->ws[lane];
    hipStream_t st = b->streams[lane];
    ppf_status s = match_prepared(models[k], ws, &p, st);
    if (s != PPF_OK) return s;
    const size_t idx = (size_t)c * n_models + k;
    if (ws->n_ref == 0) {
      HIPCHK(hipMemsetAsync(b->d_out.p + idx * cap, 0, (size_t)cap * sizeof(ppf_pose), st));
      HIPCHK(hipMemsetAsync(b->d_meta.p + idx * 2, 0, 2 * sizeof(uint32_t), st));
      HIPCHK(hipMemsetAsync(b->d_tot.p + idx * 8, 0, 8 * sizeof(unsigned long long), st));
      return PPF_OK;
    }
    const unsigned long long* tot = ws->counters.p + (size_t)ws->n_ref * models[k]->info.n_tiles + ws->n_ref;
    k_pose_block<<<dim3((words + 255) / 256), dim3(256), 0, st>>>(ws->d_final.p, ws->cl_u32.p, 0, b->d_out.p + idx * cap, cap,
                                                                   b->d_meta.p + idx * 2, ws->cursors.p + CUR_OVERFLOW,
                                                                   b->d_tot.p + idx * 8, tot);
    HIPCHK(hipGetLastError());
    return PPF_OK;
  };
  /* bring crop c into the lane's workspace (upload if it is a host cloud, then A2) */
  auto stage_crop = [&](int lane, int c, int use) -> ppf_status {
    hipStream_t st = b->streams[lane];
    const float* d_src = scenes[c];
    if (!scenes_on_device) {
      const size_t floats = (size_t)ns[c] * sstride;
      const int slot = lane * 2 + (use & 1);
      HIPCHK(hipEventSynchronize(b->pinned_ev[slot])); /* the upload that last used this staging buffer is done */
      if (b->pinned_cap[slot] < floats) {
        if (b->pinned[slot]) HIPCHK(hipHostFree(b->pinned[slot]));
        b->pinned[slot] = nullptr; b->pinned_cap[slot] = 0;
        HIPCHK(hipHostMalloc((void**)&b->pinned[slot], floats * sizeof(float), hipHostMallocDefault));
        b->pinned_cap[slot] = floats;
      }
      memcpy(b->pinned[slot], scenes[c], floats * sizeof(float));
      HIPCHK(b->d_scene[lane]->reserve(floats));
      HIPCHK(hipMemcpyAsync(b->d_scene[lane]->p, b->pinned[slot], floats * sizeof(float), hipMemcpyHostToDevice, st));
      HIPCHK(hipEventRecord(b->pinned_ev[slot], st));
      d_src = b->d_scene[lane]->p;
    }
    return prepare_scene(b->ws[lane], d_src, ns[c], sstride, nullptr, 0, 6, &p, st);
  };

  std::vector<int> uses(b->lanes, 0);
  for (int c = 0; c < n_scenes; c++) {
    const int lane = c % b->lanes;
    ppf_status s = stage_crop(lane, c, uses[lane]++);
    if (s != PPF_OK) return s;
    for (int k = 0; k < n_models; k++) {
      s = enqueue_pair(lane, c, k);
      if (s != PPF_OK) return s;
    }
  }
  for (int l = 0; l < b->lanes; l++) HIPCHK(hipStreamSynchronize(b->streams[l]));
  std::vector<uint32_t> meta(n_match * 2);
  std::vector<unsigned long long> tot(n_match * 8);
  HIPCHK(hipMemcpy(meta.data(), b->d_meta.p, meta.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(tot.data(), b->d_tot.p, tot.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  /* learn the hit fraction per lane; repeat the matches whose pools overflowed, one at a time, with doubled pools */
  int retries = 0;
  for (int c = 0; c < n_scenes; c++) {
    const int lane = c % b->lanes;
    ppf_workspace* ws = b->ws[lane];
    bool staged = false;
    for (int k = 0; k < n_models; k++) {
      const size_t idx = (size_t)c * n_models + k;
      bool at_full = false; /* the match that raised the flag already ran with worst-case pools */
      while (meta[idx * 2 + 1]) {
        if (at_full) return fail(PPF_ERR_CAPACITY, "ppf_batch_run: hit pools overflowed at worst-case size");
        retries++;
        ppf_status s = PPF_OK;
        if (!staged) { s = stage_crop(lane, c, uses[lane]++); staged = true; }
        if (s != PPF_OK) return s;
        /* bigger pools than this (crop, model) match had (the flag says which one was short): match_prepared looks the
         * model's fractions up itself */
        workspace_hold_model(ws, nullptr);
        ppf_workspace::Learned* fm = nullptr;
        for (auto& e : ws->frac_by_model)
          if (e.model == models[k]) fm = &e;
        if (!fm) { ws->frac_by_model.push_back({models[k], 0.25, 0.4}); fm = &ws->frac_by_model.back(); }
        const uint32_t flags = meta[idx * 2 + 1];
        at_full = fm->hit >= 1.0;
        if (flags & 3u) fm->hit = std::min(1.0, 2.0 * fm->hit);
        if (flags & 4u) fm->run = std::min(1.0, 2.0 * fm->run);
        s = enqueue_pair(lane, c, k);
        if (s != PPF_OK) return s;
        HIPCHK(hipStreamSynchronize(b->streams[lane]));
        HIPCHK(hipMemcpy(&meta[idx * 2], b->d_meta.p + idx * 2, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&tot[idx * 8], b->d_tot.p + idx * 8, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      }
    }
  }
  /* what every lane remembers per model: the densest crop it saw */
  std::vector<double> lane_hit((size_t)b->lanes * n_models, 0.0), lane_run((size_t)b->lanes * n_models, 0.0);
  ppf_batch_stats st{};
  for (size_t idx = 0; idx < n_match; idx++) {
    const unsigned long long* t = &tot[idx * 8];
    st.n_votes += t[0]; st.n_pairs += t[1]; st.n_lds_atomics += t[2]; st.n_hits += t[3];
    const size_t slot = (size_t)((idx / n_models) % b->lanes) * n_models + idx % n_models;
    if (t[1]) lane_hit[slot] = std::max(lane_hit[slot], (double)t[3] / (double)t[1]);
    if (t[3]) lane_run[slot] = std::max(lane_run[slot], (double)t[4] / (double)t[3]);
  }
  for (int l = 0; l < b->lanes; l++) {
    ppf_workspace* ws = b->ws[l];
    workspace_hold_model(ws, nullptr); /* the next call looks its model up */
    for (int k = 0; k < n_models; k++) {
      const double fh = lane_hit[(size_t)l * n_models + k], fr = lane_run[(size_t)l * n_models + k];
      if (!(fh > 0)) continue;
      const double hit = std::min(1.0, std::max(1e-3, 1.06 * fh)), run = std::min(1.0, std::max(0.02, 1.10 * fr));
      bool found = false;
      for (auto& fm : ws->frac_by_model)
        if (fm.model == models[k]) { fm.hit = hit; fm.run = run; found = true; }
      if (!found) ws->frac_by_model.push_back({models[k], hit, run});
    }
  }
  if (out) HIPCHK(hipMemcpy(out, b->d_out.p, n_match * cap * sizeof(ppf_pose), hipMemcpyDeviceToHost));
  if (n_out)
    for (size_t idx = 0; idx < n_match; idx++) n_out[idx] = (int)std::min<uint32_t>(meta[idx * 2], (uint32_t)cap);
  st.n_matches = (int)n_match;
  st.n_retries = retries;
  st.lanes = b->lanes;
  st.ms_wall = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_start).count();
  if (stats) *stats = st;
  return PPF_OK;
}

ppf_status ppf_batch_device_block(ppf_batch* b, void** d_poses, int* n_records) {
  if (!b || !d_poses) return fail(PPF_ERR_INVALID, "ppf_batch_device_block: bad argument");
  *d_poses = b->d_out.p;
  if (n_records) *n_records = b->last_records;
  return PPF_OK;
}

ppf_status ppf_batch_copy_block(ppf_batch* b, void* d_dst, int n_records, void* stream) {
  if (!b || !d_dst || n_records < 0 || n_records > b->last_records) return fail(PPF_ERR_INVALID, "ppf_batch_copy_block: bad argument");
  if (n_records)
    HIPCHK(hipMemcpyAsync(d_dst, b->d_out.p, (size_t)n_records * sizeof(ppf_pose), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return PPF_OK;
}

ppf_status ppf_match_batch(const ppf_model* const* models, int n_models, const float* const* scenes, const int* ns,
                           int sstride, int n_scenes, const ppf_match_params* params, ppf_pose* out, int cap, int* n_out) {
  if (!models || n_models <= 0 || !scenes || !ns || n_scenes <= 0 || !params || !out || cap <= 0 || !n_out)
    return fail(PPF_ERR_INVALID, "ppf_match_batch: bad argument");
  for (int k = 0; k < n_models; k++)
    if (!models[k]) return fail(PPF_ERR_NOT_TRAINED, "ppf_match_batch: model %d is not trained", k);
  if (!have_device()) return fail(PPF_ERR_HIP, "ppf_match_batch: no HIP device (this engine has no CPU fallback)");
  ppf_batch* b = nullptr;
  ppf_status s = ppf_batch_create(std::min(4, n_scenes), &b);
  if (s != PPF_OK) return s;
  s = ppf_batch_run(b, models, n_models, scenes, ns, sstride, n_scenes, 0, params, out, cap, n_out, nullptr);
  (void)ppf_batch_destroy(b);
  return s;
}

ppf_status ppf_raw_votes(const ppf_model* m, const float* scene, int ns, int sstride, const float* edge, int ne,
                         int estride, const ppf_match_params* params, ppf_vote* votes, ppf_pose* raw_poses, int cap,
                         int* n_ref, ppf_match_stats* stats) {
  ppf_workspace ws;
  ppf_status s = ppf_workspace_enable_timing(&ws, 1);
  if (s == PPF_OK) s = run_host(m, scene, ns, sstride, edge, ne, estride, params, &ws);
  if (s == PPF_OK) s = ppf_workspace_results(&ws, votes, raw_poses, cap, n_ref, nullptr, 0, nullptr, stats);
  return s;
}

/* ---- model (de)serialisation: versioned binary CSR (the reference's XML format is defined by a
 * private OpenCV patch and unknown) ------------------------------------------------- */
static const char PPF_MAGIC[8] = {'P', 'P', 'F', 'H', 'I', 'P', '0', '3'}; /* 02: pair-record table; 03: ppf_train_params.feature */

ppf_status ppf_model_save(const ppf_model* m, const char* path) {
  if (!m || !path) return fail(PPF_ERR_INVALID, "ppf_model_save: NULL");
  const size_t words = ((size_t)m->info.slots + 63) / 64, nb = m->info.n_buckets, ne = m->n_records;
  std::vector<SlotWord> slotmap(words);
  std::vector<uint32_t> boff((size_t)m->info.n_tiles * (nb + 1)), bslot(nb);
  std::vector<uint4> ent(ne);
  HIPCHK(hipMemcpy(slotmap.data(), m->slotmap.p, words * sizeof(SlotWord), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(boff.data(), m->bucket_off.p, boff.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (nb) HIPCHK(hipMemcpy(bslot.data(), m->bucket_slot.p, nb * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (ne) HIPCHK(hipMemcpy(ent.data(), m->records.p, ne * sizeof(uint4), hipMemcpyDeviceToHost));
  FILE* f = fopen(path, "wb");
  if (!f) return fail(PPF_ERR_IO, "ppf_model_save: cannot open %s", path);
  bool ok = fwrite(PPF_MAGIC, 1, 8, f) == 8;
  ok = ok && fwrite(&m->params, sizeof(m->params), 1, f) == 1;
  ok = ok && fwrite(&m->info, sizeof(m->info), 1, f) == 1;
  ok = ok && fwrite(&m->n_records, sizeof(m->n_records), 1, f) == 1;
  ok = ok && fwrite(m->sampled.data(), sizeof(float), m->sampled.size(), f) == m->sampled.size();
  ok = ok && fwrite(slotmap.data(), sizeof(SlotWord), words, f) == words;
  ok = ok && fwrite(boff.data(), sizeof(uint32_t), boff.size(), f) == boff.size();
  ok = ok && fwrite(bslot.data(), sizeof(uint32_t), nb, f) == nb;
  ok = ok && fwrite(ent.data(), sizeof(uint4), ne, f) == ne;
  ok = (fclose(f) == 0) && ok;
  if (!ok) return fail(PPF_ERR_IO, "ppf_model_save: short write to %s", path);
  return PPF_OK;
}

/* Everything read from the file is checked before it reaches a kernel: header fields against each other and against
 * the file size, the CSR rows, the record rows (LDS byte offsets k_vote adds to) and alphas, the slot map's ranks.  A file
 * that fails any check is PPF_ERR_IO; no exception leaves this function. */
static ppf_status model_load_impl(const char* path, ppf_model** out, bool check_only) {
  FILE* f = fopen(path, "rb");
  if (!f) return fail(PPF_ERR_IO, "ppf_model_load: cannot open %s", path);
  struct Closer { FILE* f; ~Closer() { if (f) fclose(f); } } closer{f};
  auto bad = [&](const char* what) { return fail(PPF_ERR_IO, "ppf_model_load: %s is not a valid model file (%s)", path, what); };
  if (fseek(f, 0, SEEK_END) != 0) return bad("seek");
  const long fsize = ftell(f);
  if (fsize < 0 || fseek(f, 0, SEEK_SET) != 0) return bad("seek");
  char magic[8];
  std::unique_ptr<ppf_model> owner(new ppf_model()); /* released on every early return */
  ppf_model* m = owner.get();
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, PPF_MAGIC, 8) != 0) return bad("magic");
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
  const uint64_t expect = 8 + sizeof(m->params) + sizeof(m->info) + sizeof(m->n_records) + N * 6 * sizeof(float) +
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
        const uint32_t row = codes[sl] & ROW_CODE_MASK, cx = (codes[sl] >> ROW_X_SHIFT) & 31u, cq = (codes[sl] >> ROW_Q_SHIFT) & 63u;
        if ((row & 2u) || row / 4 + (uint32_t)A + 1 > limit_words) return bad("record row"); /* bin A of the last row: the word behind the rows */
        if ((codes[sl] >> 29) || cx > (uint32_t)A || cq > (uint32_t)AGG_Q) return bad("record cell");
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

#include "ppf_icp_host.h"  /* row N2: ICP refinement, host side */
#include "ppf_prep_host.h" /* row N4: cloud stages, host side */
