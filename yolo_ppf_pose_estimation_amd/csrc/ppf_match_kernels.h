/*
 * ppf_match_kernels.h — the matching hot path on gfx950, included by
 * ppf_hip.hip.  Reference call sites: /root/reference/include/CloudProcessing.h:442 (match), :495
 * (match_S2B).
 *
 * Per batch of scene reference points:
 *
 *   k_frames   one thread per reference point: the rotation/translation (Rsg, tsg) that takes the
 *              reference point to the origin with its normal on +x (fp64, 12 doubles per point).
 *
 *   k_pairs    one thread per scene pair (s_r, s_i): pair feature -> 4 x int32 key -> dense bucket id
 *              (key table, MurmurHash3 for keys outside it).  About 15 % of the pairs of a crop land in a
 *              non-empty slot; a workgroup (3,072 pairs of one reference point) appends its hits
 *              {bucket, j} to a striped pool with ONE atomic, so the pool holds exactly the hits that
 *              exist (capacity is an estimate; running out raises a flag and the host repeats the call
 *              with a bigger pool).  VALU(fp64)-bound.
 *
 *   k_group    one workgroup per reference point: alpha_s of every hit, counting sort of the hits by
 *              bucket (one LDS counter per bucket), the run table {bucket, first hit, m hits} (runs with
 *              many hits first) and the per-hit payload in sorted order: (float)alpha_s, alpha_s, and the
 *              hit's cell (Y, p) in units of alpha bins (see "aggregated votes").  Also gives every run of many
 *              hits its count tables (one per 191 hits) out of the batch's table pool and writes down what each covers.
 *
 *   k_tables   one wave per count table: histogram of the table's hits by cell, the counts an entry in each cell adds
 *              to each bin, the hits' offsets in cell order; built in LDS, copied to the pool (HBM / L2).  Once per run:
 *              k_vote meets the run again for every chunk of the bucket's records, accumulator tile and half.
 *
 *   k_vote     one workgroup per (reference point, accumulator tile); the tile's Hough accumulator
 *              lives in LDS as 16-BIT cells, two model rows per 32-bit word: row r < H (H = half the tile's
 *              rows) counts in the low halves, row r + H in the high halves of the same words.  A pair
 *              record carries the byte offset of its row's bin 0 and, in bit 0, which half it owns, so a
 *              vote stays one ds_add_u32 of 1 or 0x10000 at (row + 4*bin) and a tile holds twice the rows
 *              (the 2,000-point model of the headline case: one tile instead of two, i.e. half the work
 *              items and table traffic).  A cell that passes 65,535 would carry into its neighbour; every
 *              vote cast lands somewhere in LDS (a cell or a guard word), so the workgroup compares the
 *              sum of what it finds there with the number of votes it issued: any carry makes the two
 *              differ (each carry loses 65,535 or 65,536 from the sum); the workgroup then flags its
 *              (reference point, tile) and a second launch with 32-bit cells, one workgroup per half of
 *              the tile's rows, votes the flagged ones again (a scene that flags many of them goes straight
 *              to 32-bit cells from its next call on).
 *              A run (one bucket, m hits) meets the bucket's entries in one of two ways:
 *                direct      every entry votes once per hit: 1 fma + cvt + fract + address + ds_add_u32
 *                            per vote, entries held in registers while the hits go by;
 *                aggregated  (m >= PPF_AGG_MIN_HITS) the hits of the run were histogrammed by cell (k_tables);
 *                            every entry adds COUNTS: 17 ds_add_u32 per entry and <= 191 hits instead of
 *                            one per hit (see below).
 *              Work items (run x chunk of entries x group of hits) are claimed by waves from an LDS
 *              counter.
 *
 * Alpha bin, exactly: bin = (int)(A*(alpha_m - alpha_s + 2pi)/(4pi)) in fp64 is what the reference
 * computes.  The direct path evaluates q = (alpha_m - alpha_s)*A/(4pi) + A/2 in fp32 (|error| <= 1.1e-7*A,
 * DESIGN.md section 4) and takes trunc(q) whenever q is farther than G = 5e-7*A from an integer; otherwise
 * (about 3e-5 of the votes) the lane re-evaluates the fp64 chain.  Both give the same integer.
 *
 * Aggregated votes.  With x = alpha_m*A/(4pi) + A/2 = X + phi and y = alpha_s*A/(4pi) = Y + psi
 * (X, Y integers, 0 <= phi, psi < 1) the bin is floor(x - y) = X - Y - [phi < psi].  The fraction is cut
 * into Q = 64 cells: an entry in cell q = floor(64 phi) and a hit in cell p = floor(64 psi) with p != q
 * are ordered by their cells alone, so for one entry all hits of a run with p < q and the same Y land
 * in bin X - Y, those with p > q in bin X - Y - 1: a table T[q][j] (j = Y + 8 = 0..16, built per run
 * from the hits' cell histogram) holds the COUNT that bin X + 8 - j receives, and the entry casts 17
 * counted atomics whatever m is.  Hits in the entry's own cell (1/64 of them) are voted one by one
 * with the direct arithmetic.  Exactness: the table is only used for pairs separated by a cell boundary
 * from which the entry is provably away (fp32 cell index with a guard band, fp64 when inside the band,
 * one-by-one votes when fp64 is still within 1e-7 of a boundary), where the fp64 chain's rounding
 * (1.5e-14 bins) cannot change floor().  alpha_m < -pi (only (float)-pi) could make x - y negative, where
 * the reference truncates towards zero: such entries vote one by one.  Needs Y in [-8, 7]: A <= 31.
 */
#ifndef PPF_MATCH_KERNELS_H
#define PPF_MATCH_KERNELS_H

#ifndef PPF_GUARD_REL
#define PPF_GUARD_REL 5e-7f /* alpha-bin guard band relative to A; the fp32 error bound is 1.1e-7 */
#endif
#ifndef PPF_AGG_MIN_HITS
#define PPF_AGG_MIN_HITS 24 /* runs with at least this many hits vote through the count table */
#endif
#ifndef PPF_AGG_MIN_RECORDS
#define PPF_AGG_MIN_RECORDS 32 /* ... when the (tile, bucket) holds at least this many pair records */
#endif
/* Attribution builds (tools/build_attribution.sh, never the product): PPF_ABL_<class>=0 leaves one class of k_vote's work
 * out; the votes are then wrong and the 16-bit overflow check is off, only times and counters of such a build mean anything.  The difference to the
 * product build is what the class costs, with the latency it exposes included (profiles/r03_vote_classes.md). */
#ifndef PPF_ABL_COUNTED
#define PPF_ABL_COUNTED 1      /* count-table items: the 17 counted atomics per entry and the table-row reads behind them */
#endif
#ifndef PPF_ABL_OWNCELL
#define PPF_ABL_OWNCELL 1      /* count-table items: the one-by-one votes of an entry's own cell */
#endif
#ifndef PPF_ABL_DIRECT_SMALL
#define PPF_ABL_DIRECT_SMALL 1 /* direct items on at most 32 records (one entry per lane) */
#endif
#ifndef PPF_ABL_DIRECT_BIG
#define PPF_ABL_DIRECT_BIG 1   /* direct items on more than 32 records: 0 = neither loads nor votes, 2 = the record loads without the votes */
#endif
/* PPF_MOCK_PAIRBINS (never the product; wrong votes, timing only): what an accumulator with two adjacent alpha bins per 32-bit word
 * would cost -- a count-table entry casts 9 counted atomics instead of 17 (one v_perm each), a one-by-one vote forms its word
 * address from bin / 2 and its increment from bin % 2 (two or three more VALU instructions than now). */
#ifndef PPF_MOCK_PAIRBINS
#define PPF_MOCK_PAIRBINS 0
#endif
#define PPF_ABL_ANY (PPF_ABL_COUNTED != 1 || PPF_ABL_OWNCELL != 1 || PPF_ABL_DIRECT_SMALL != 1 || PPF_ABL_DIRECT_BIG != 1 || PPF_MOCK_PAIRBINS)
/* one vote for bin k of the row at LDS address p */
__device__ __forceinline__ void vote_one(const uint32_t p, const int k, const uint32_t inc);

/* Diagnostic build (-DPPF_PHASE_CLOCKS, never the product): every k_vote wave sums the shader clocks (s_memtime) it spends in
 * each phase; ppf_match_stats.phase_clocks = the sums over all waves of the call (tools/vote_phases.py prints them).
 * 0 staging (clear, run table, scans, barriers)  1 claim + look-up + prefetch of the next item  2 count-table items
 * 3 direct items of more than 32 records  4 direct items of at most 32 records  5 end of segment: waiting for the other waves
 * 6 scan, reductions, result  7 whole workgroup, in ticks of the constant 100 MHz clock (s_memrealtime): summed over the waves
 * and divided by 16 waves x 256 CUs x the kernel's time it is the share of the CUs' time a workgroup was resident.
 * -DPPF_PHASE_CLOCKS=2 records only that one. */
#ifdef PPF_PHASE_CLOCKS
#define PPF_PHASE_DECL unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long ph_t_ = __builtin_amdgcn_s_memtime(); const unsigned long long ph_rt0_ = __builtin_amdgcn_s_memrealtime()
#if PPF_PHASE_CLOCKS == 2 /* only phase 7, the workgroup's wall time: the stamps of the other phases slow the kernel by a third */
#define PPF_PHASE(k) do { } while (0)
#else
#define PPF_PHASE(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_[k] += t_ - ph_t_; ph_t_ = t_; } while (0)
#endif
#define PPF_PHASE_FLUSH(tally, lane) do { ph_[7] = __builtin_amdgcn_s_memrealtime() - ph_rt0_; (void)ph_t_; if ((lane) == 0) for (int k_ = (PPF_PHASE_CLOCKS == 2 ? 7 : 0); k_ < 8; k_++) atomicAdd(&(tally)[6 + k_], ph_[k_]); } while (0)
#else
#define PPF_PHASE_DECL do { } while (0)
#define PPF_PHASE(k) do { } while (0)
#define PPF_PHASE_FLUSH(tally, lane) do { } while (0)
#endif
#ifndef PPF_COST_TABLE
#define PPF_COST_TABLE 16   /* k_vote's launch order: what one count table costs a pair record, in direct hits */
#endif
#ifndef PPF_COST_MIN_HITS
#define PPF_COST_MIN_HITS 8 /* ... and the least a direct run costs it (loading and unpacking the record) */
#endif
#ifndef PPF_COST_ITEM
#define PPF_COST_ITEM 512     /* ... and a run's fixed cost (claim, look-up, first loads), in record-hits */
#endif
#ifndef PPF_ACC32_COST
#define PPF_ACC32_COST 1.02 /* what voting a (reference point, tile) with 32-bit cells (two passes, one per half of its rows) costs, relative to 16-bit cells
                               (C4, round 4: 342 ms with 32-bit cells for everything against 380 ms when the tenth of the votes that overflows is cast twice) */
#endif
#ifndef PPF_ACC32_SWITCH
#define PPF_ACC32_SWITCH 0.05 /* share of a call's votes cast twice (16-bit cells overflowed) beyond which a workspace goes to 32-bit cells for everything */
#endif
#ifndef PPF_TWO_QUEUES
#define PPF_TWO_QUEUES 0 /* 1: k_vote claims count-table items and direct items from two queues, half of the waves preferring each (measured: +2 %, profiles/r03_vote_variants.md) */
#endif
#ifndef PPF_PIPE_VALU
#define PPF_PIPE_VALU 4 /* VALU instructions scheduled between two LDS atomics of the pipelined direct loop */
#endif

constexpr int PAIR_BLOCK = 256;
#ifndef PPF_PAIRS_PER_THREAD
#define PPF_PAIRS_PER_THREAD 12
#endif
/* one k_pairs workgroup covers 3,072 paired points of one reference point.  Swept in round 4 (k_pairs on C2 / C4 / C5):
 * 4: 0.74 ms; 8: 0.585 / 10.7 / 18.6; 10: 0.555; 12: 0.533 / 9.4 / 17.0; 14: 0.515 (50,000 points are 13.95 chunks of 3,584:
 * a fit of that size, not taken); 16: 0.560; 24: 0.625; 32: 0.81 -- fewer workgroups pay the prologue, the compaction and the
 * pool append (and k_group walks fewer pieces), until the stash takes the LDS of a fifth workgroup per CU */
constexpr int PAIRS_PER_THREAD = PPF_PAIRS_PER_THREAD;
constexpr int VOTE_BLOCK = 1024;
constexpr int VOTE_WAVES = VOTE_BLOCK / 64;
constexpr int VOTE_UNROLL = 4;        /* pair records (2 entries each) loaded per lane per batch */
constexpr int VOTE_CHUNK = 64 * VOTE_UNROLL * 4; /* pair records per direct work item (1024 = 2048 entries) */
constexpr int VOTE_MAX_HITS = PPF_AGG_MIN_HITS; /* hits of one run voted per direct work item: a run that votes directly on a bucket of some size has fewer,
                                                   so its records are read once (16 per item: +0.9 % on C2) */
#ifndef PPF_AGG_CHUNK
#define PPF_AGG_CHUNK 1024
#endif
constexpr int AGG_CHUNK = PPF_AGG_CHUNK; /* pair records per aggregated work item (each item copies its count table into LDS; 512 / 2048: +1 %) */
constexpr int AGG_NY = 16;            /* integer parts Y + 8 of a hit */
constexpr int AGG_SUB = 191;          /* hits per count table (counts are bytes; 3 hits per lane) */
constexpr int AGG_MAX_ANGLES = 31;    /* Y = floor(alpha_s*A/(4pi)) must stay in [-8, 7] */
#ifndef PPF_RUN_SEG
#define PPF_RUN_SEG 704
#endif
constexpr int RUN_SEG = PPF_RUN_SEG;  /* runs staged in LDS per segment, AT LEAST (every segment costs the workgroup two dependent reads and three barriers: 256 -> 512
                                         was worth 3 % on C2, whose reference points have about 600 runs): the tile size of a model is chosen with this much staging ... */
constexpr int RUN_SEG_MAX = 1024;     /* ... and a call gives the staging whatever LDS the model's accumulator tile then leaves, in steps of 64 runs (MatchArgs::run_seg;
                                         C2's 2,000-row tile: 896, and 704 -> 896 is another 1 % -- a third of its reference points have more than 704 runs) */
static_assert(RUN_SEG % 64 == 0 && RUN_SEG <= RUN_SEG_MAX && RUN_SEG_MAX == 1024, "the staging loop gives one thread to a run and scans whole waves");
constexpr int GROUP_BLOCK = 512;       /* two k_group workgroups per CU when the bucket counters fit half the LDS (0.705 -> 0.665 ms on C2) */
#ifndef PPF_GROUP_MLP
#define PPF_GROUP_MLP 4
#endif
constexpr int GROUP_MLP = PPF_GROUP_MLP; /* hits a k_group thread has in flight per step of its two passes */
constexpr int GROUP_MAX_BUCKETS = 36000; /* LDS counters of k_group per round (144 KB) */
constexpr int POOL_STRIPES = 64;      /* the raw hit pool is cut into stripes with one cursor each */
constexpr int LDS_HEADER = 256;       /* bytes: reduction scratch (16 words) + claim counter */

/* cursor words (each on its own 128-byte line): stripe s at s*32, then sorted hits, runs, overflow flag */
constexpr int CUR_STRIDE = 32;
constexpr int CUR_SORTED = POOL_STRIPES * CUR_STRIDE;
constexpr int CUR_RUNS = CUR_SORTED + CUR_STRIDE;
constexpr int CUR_ODDVALUES = CUR_RUNS + CUR_STRIDE; /* != 0: the batch has a reference frame or a paired point that is not finite (or absurdly large): k_group checks every hit for an alpha_s */
constexpr int CUR_OVFCOUNT = CUR_ODDVALUES + CUR_STRIDE; /* (reference point, tile)s of this batch whose 16-bit cells overflowed: length of ovf_list */
constexpr int CUR_TABLES = CUR_OVFCOUNT + CUR_STRIDE; /* count tables given out by k_group */
constexpr int CUR_OVERFLOW = CUR_TABLES + CUR_STRIDE; /* bits 1, 2, 4, 8: raw pool, sorted pool, run table, count-table pool too small */
constexpr int CUR_WORDS = CUR_OVERFLOW + CUR_STRIDE;

/* A count table (one per run and range of <= AGG_SUB hits; built once per batch by k_tables, kept in HBM/L2):
 *   rows      (AGG_Q + 1) x AGG_ROW bytes   T[q][j], j = 0..16 (row AGG_Q all zero: entries that vote one by one), then two bytes:
 *                                           [first, end) of cell q's hits in the cell-sorted hit list (row AGG_Q: [0, hits))
 *   offsets   AGG_SUB x f32                 folded offsets Ohg of the hits in cell order (while building: 16 x AGG_Q byte counters)
 *   index     AGG_SUB x u8                  position of each sorted hit inside the table's hit range (the exact-bin fallback)
 * k_vote copies the rows of the table an item works with into its wave's LDS (AGG_SCRATCH bytes) and reads the
 * offsets of an entry's own cell straight from the table (four of them, a block of records ahead of their votes).
 * While a table is built (k_tables, TBL_BUILD_BYTES of LDS) one counter per cell follows the table (AGG_OFF_CE). */
constexpr int AGG_ROW = 20;                                        /* bytes per row: 17 counts + padding; 5 words: the same word of rows q and q' never shares a bank */
/* the range of a cell's hits in the cell-sorted hit list rides in the cell's row: bytes 17 and 18 (first hit, end; at most
 * AGG_SUB = 191), next to count 16 in the row's last word -- which k_vote reads a block of records ahead anyway (round 4: the
 * ranges as a table of their own behind the rows cost 272 bytes of LDS per wave and two more LDS reads per block, -1.7 %) */
constexpr int AGG_ROW_CE = 17;
constexpr int AGG_SCRATCH = ((AGG_Q + 1) * AGG_ROW + 15) / 16 * 16; /* per-wave LDS of k_vote: the rows */
constexpr int TBL_OFF_A32 = AGG_SCRATCH;
constexpr int TBL_A32_BYTES = (AGG_NY * AGG_Q > 768 ? AGG_NY * AGG_Q : 768);
constexpr int TBL_OFF_IDX = TBL_OFF_A32 + TBL_A32_BYTES;
constexpr int TBL_BYTES = TBL_OFF_IDX + 192;                        /* one table in HBM */
constexpr int AGG_OFF_CE = TBL_BYTES;                               /* the build's cell counters: LDS of k_tables only, not part of the table */
constexpr int TBL_BUILD_BYTES = AGG_OFF_CE + ((AGG_Q + 1) * 4 + 15) / 16 * 16; /* the LDS a k_tables wave builds a table in */
static_assert(AGG_SUB * 4 <= 768 && AGG_SUB <= 192 && AGG_SUB < 256, "table hit range too long");
static_assert(AGG_SCRATCH % 16 == 0 && TBL_BYTES % 16 == 0, "tables are copied 16 bytes at a time");
constexpr int TABLE_BLOCK = 256; /* k_tables: four tables per workgroup */
/* count tables per hit: a run of c >= PPF_AGG_MIN_HITS hits takes ceil(c / AGG_SUB) of them */
constexpr double TBL_FRAC_MAX = 1.0 / PPF_AGG_MIN_HITS + 1.0 / AGG_SUB;
constexpr double TBL_FRAC_START = 0.02; /* before a workspace has seen the model (learned from the first call on) */

/* fp32 acos for BIN SELECTION only: acos(|x|) = sqrt(1-|x|) * P(|x|), degree-7 least-squares/minimax fit,
 * measured max error 3.4e-7 rad including fp32 evaluation with a correctly rounded sqrt; the square root here is the bare
 * v_sqrt_f32 (1 ulp: at most 1.9e-7 rad more; the IEEE fix-up sequence the compiler wraps around sqrtf costs 13
 * instructions, three times per pair), so 5.3e-7 rad in all against a guard band of 4e-6 (tests/test_gpu_fastkeys.py
 * re-checks the keys against the exact path).  `t` = 1-|x| is formed in fp64 so the estimate stays relative-accurate
 * near |x| = 1 (t is 0 or a normal float: no denormal input). */
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
  const float a = __builtin_amdgcn_sqrtf(t) * p;
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
  /* |d| and 1/|d| from one v_rsq_f64 and two Newton steps (a few ulp: far inside the guards below) instead of the
   * correctly rounded sqrt + a reciprocal; |d| = 0, denormal or huge makes r infinite or NaN and sends the lane to the
   * exact chain through the checks on f3 and q3 */
  /* fused multiply-adds here (the build has -ffp-contract=off for the exact chain's sake): this path only estimates, a
   * rounding more or less is far inside its guards, and it is 10 fp64 instructions of 115 fewer per pair */
  const double d2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
  const double hd = 0.5 * d2;
  double rinv = __builtin_amdgcn_rsq(d2);
  rinv = rinv * __builtin_fma(-(hd * rinv), rinv, 1.5);
  rinv = rinv * __builtin_fma(-(hd * rinv), rinv, 1.5);
  const double f3 = d2 * rinv;
  const double x0 = __builtin_fma(n1.z, dz, __builtin_fma(n1.y, dy, n1.x * dx)) * rinv;
  const double x1 = __builtin_fma(n2.z, dz, __builtin_fma(n2.y, dy, n2.x * dx)) * rinv;
  const double x2 = __builtin_fma(n1.z, n2.z, __builtin_fma(n1.y, n2.y, n1.x * n2.x));
  bool slow = !(f3 > 2.0 * PPF_EPS) || !(ppf_fabs(x0) <= 1.0 - 1e-12) || !(ppf_fabs(x1) <= 1.0 - 1e-12) || !(ppf_fabs(x2) <= 1.0);
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
    slow |= !(fr3 > 1e-9 && fr3 < 1.0 - 1e-9) || !(q3 < 1.0e5); /* 1e5 bins x 1e-15 relative error of f3 stays below the 1e-9 guard */
  }
  if (slow) {
    double f[4] = {0, 0, 0, 0};
    ppf_pair_feature(p1, n1, p2, n2, f);
    k[0] = ppf_d2i(f[0] / angle_step); k[1] = ppf_d2i(f[1] / angle_step); k[2] = ppf_d2i(f[2] / angle_step);
    k[3] = ppf_d2i(f[3] / dist_step);
  }
}

/* PCL's pair feature (policy switch): the exact fp64 chain for every pair -- no fp32 estimate, this is not the path the
 * headline metric runs.  false: degenerate pair, left out like PCL does. */
__device__ __forceinline__ bool pair_key_darboux(const ppf_vec3& p1, const ppf_vec3& n1, const ppf_vec3& p2, const ppf_vec3& n2,
                                                 const double angle_step, const double dist_step, int32_t (&k)[4]) {
  double f[4];
  if (!ppf_pair_feature_darboux(p1, n1, p2, n2, f)) return false;
  k[0] = ppf_floor_key(f[0] / angle_step); k[1] = ppf_floor_key(f[1] / angle_step); k[2] = ppf_floor_key(f[2] / angle_step);
  k[3] = ppf_floor_key(f[3] / dist_step);
  return true;
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
  /* key -> dense bucket id (-1: empty slot), indexed by key_index(): the hash of every quantised key a scene can
   * produce, tabulated once per model; keys outside the table take the hash path */
  const int32_t* key_lut;
  KeyDims kd;
  const uint32_t* bucket_off;
  int n_buckets;
  const uint4* records;    /* pair records {row_a, row_b, alpha_a, alpha_b}; bucket_off counts records */
  int n_tiles, tile_refs, num_angles, n_model;
  double angle_step, dist_step;
  /* per-batch scratch */
  double* frames;          /* [n_ref][12] */
  uint2* raw;              /* hit pool {bucket, j}: POOL_STRIPES stripes of stripe_cap */
  uint32_t stripe_cap;
  int stripe_bits;         /* log2 of the stripes in use (<= POOL_STRIPES): few workgroups share few stripes, so that every
                              stripe averages over hundreds of them */
  uint32_t* cursors;       /* CUR_WORDS */
  uint2* chunk_desc;       /* [n_ref][pair_chunks] {first raw hit, count} of each k_pairs workgroup */
  int pair_chunks;
  unsigned long long* hit_count; /* [n_ref] raw hits (also the k_rank key of k_group's launch order) */
  double* s_a64;           /* sorted payload: alpha_s */
  uint16_t* s_cell;        /*                 (Y + 8) * AGG_Q + p */
  uint32_t sorted_cap;
  uint4* runs;             /* {bucket, first sorted hit, m, first count table of the run (m >= agg_min_hits)} */
  unsigned char* tables;   /* count tables, TBL_BYTES each: built by k_tables, read by k_vote */
  uint2* table_desc;       /* {first sorted hit, hits <= AGG_SUB} of every table, written by k_group (zeroed before: 0 hits = not given out) */
  uint32_t table_cap;
  uint32_t run_cap;
  uint2* run_blocks;       /* [n_ref][n_rounds] {first run, runs} */
  int n_rounds, round_buckets; /* k_group sorts round_buckets bucket ids per pass over the reference point's hits */
  const uint32_t* bucket_total; /* [n_buckets] entries of a bucket over all tiles */
  const uint32_t* bucket_mid;   /* [n_tiles][n_buckets] entries of low-half rows = the first dealing positions of the bucket (k_bucket_mid) */
  unsigned long long* work;     /* [n_ref] votes the reference point will cast (sum of its hits' bucket sizes) */
  uint32_t* perm;               /* [n_ref] reference points ordered by work, heaviest first (k_rank) */
  const uint32_t* perm_group;   /* [n_ref] reference points ordered by hit count, for k_group */
  int agg_min_hits;             /* 0: every run votes directly */
  int key_exact;                /* PPF_KEY_EXACT table: keys outside the key table (but for NaN angle bins, key_index_nan) match nothing */
  double pair_radius;           /* > 0: pairs farther apart than this are skipped (not counted) */
  int run_seg;                  /* runs k_vote stages per segment: RUN_SEG .. RUN_SEG_MAX, a multiple of 64 (vote_run_seg) */
  int acc32;                    /* k_vote<.., true> (32-bit cells, one workgroup per half of a tile's rows): 1 = every (reference point, tile),
                                   2 = only those the 16-bit launch flagged in ovf_items */
  uint32_t* ovf_items;          /* [n_ref_all * n_tiles] != 0: voted with 32-bit cells -- 1: a 16-bit cell of this (reference point, tile) overflowed;
                                   2: sent there without a 16-bit attempt (it will cast at least heavy_votes votes) */
  unsigned long long* item_votes; /* [n_ref_all * n_tiles] votes the (reference point, tile) will cast (NULL: not wanted), written by the 16-bit launch */
  unsigned long long heavy_votes; /* a (reference point, tile) that will cast at least this many goes straight to 32-bit cells (~0: none does) */
  uint32_t* ovf_list;           /* [n_ref * n_tiles] the same for this batch as a list: local reference point | tile << 16 (cursors[CUR_OVFCOUNT] entries) */
  int count_only;               /* k_pairs only counts its hits (cold workspace: sizes the pools of the real pass) */
  /* results, indexed by global r */
  uint2* partial;               /* [n_ref_all * n_tiles * 2] {max votes, local flat index}: slot 2*tile (16-bit cells) or 2*tile + pass (32-bit cells) */
  uint32_t* edge;               /* [n_ref_all * n_tiles * 2] 32-bit cells only: pass 0: bin A of the last low-half row; pass 1: bin 0 of the first high-half row */
  unsigned long long* cellsum;  /* [n_ref_all * n_tiles] sum of the tile's accumulator == votes cast */
  unsigned long long* pairs;    /* [n_ref_all] pairs hashed */
  unsigned long long* tally;    /* [14] (8 of them: -DPPF_PHASE_CLOCKS only) LDS atomic lane-operations issued by k_vote; hits grouped, runs written by k_group; (reference point, tile)s voted with 32-bit cells; votes the 16-bit launch cast for the ones it flagged; count tables handed out by k_group */
  uint32_t* acc_dump;           /* optional [n_ref_all][n_model*num_angles] full accumulators (debug/tests) */
};

__device__ __forceinline__ int ref_row(const MatchArgs& a, int r_local) {
  return (a.ref_offset + (a.ref_base + r_local) * a.ref_stride) * a.scene_step;
}

__global__ __launch_bounds__(64) void k_frames(MatchArgs a) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  /* on the side: are all paired points ordinary numbers?  Then (with ordinary frames) every hit has an alpha_s and
   * k_group's counting pass need not look at the points at all. */
  bool odd = false;
  for (int j = r; j < a.paired.n; j += gridDim.x * blockDim.x) {
    const float m = (fabsf(a.paired.x[j]) + fabsf(a.paired.y[j])) + fabsf(a.paired.z[j]); /* a sum, not fmaxf: that one drops a NaN operand */
    odd |= !(m < 1e30f); /* NaN, infinite or beyond anything a transformed coordinate could keep finite */
  }
  if (r < a.n_ref) {
    const int i = ref_row(a, r);
    double R[9], t[3];
    ppf_transform_rt(ld3(a.surf.x, a.surf.y, a.surf.z, i), ld3(a.surf.nx, a.surf.ny, a.surf.nz, i), R, t);
    double* f = a.frames + (size_t)r * 12;
#pragma unroll
    for (int k = 0; k < 9; k++) { f[k] = R[k]; odd |= !(ppf_fabs(R[k]) < 1e30); }
#pragma unroll
    for (int k = 0; k < 3; k++) { f[9 + k] = t[k]; odd |= !(ppf_fabs(t[k]) < 1e30); }
  }
  if (odd) atomicOr(&a.cursors[CUR_ODDVALUES], 1u);
}

/* stripe of the raw pool a k_pairs workgroup appends to: a multiplicative hash of its linear index, so that no stripe
 * collects a systematic share of the (shorter) last chunks of the paired cloud */
__device__ __forceinline__ uint32_t pool_stripe(const uint32_t wg, const int stripe_bits) {
  return stripe_bits ? (wg * 2654435761u) >> (32 - stripe_bits) : 0u;
}

/* grid: x = chunks of PAIR_BLOCK*PAIRS_PER_THREAD paired points, y = reference point of the batch */
template <bool DARBOUX, bool S2B>
__global__ __launch_bounds__(PAIR_BLOCK) void k_pairs(MatchArgs a) {
  __shared__ uint2 stash[PAIRS_PER_THREAD][PAIR_BLOCK]; /* {bucket, j} of this thread's hits, one slot per iteration */
  __shared__ uint32_t wtot[PAIR_BLOCK / 64];
  __shared__ uint32_t wg_base;
  const int r = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  /* a 32-bit BYTE offset next to the scalar base pointer: the load takes it as is (global_load ... v_off, s[base]); with an
   * index the compiler forms a 64-bit address per array and load (it cannot know that 4 * index stays below 2^32): 7 VALU
   * instructions of address arithmetic per point */
  auto ldf = [](const float* __restrict__ base, const uint32_t byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
  {
    const uint32_t jc = (uint32_t)min(j0, n - 1) * 4u;
    nx0 = ldf(a.paired.x, jc); nx1 = ldf(a.paired.y, jc); nx2 = ldf(a.paired.z, jc);
    nx3 = ldf(a.paired.nx, jc); nx4 = ldf(a.paired.ny, jc); nx5 = ldf(a.paired.nz, jc);
  }
#pragma unroll 2
  for (int it = 0; it < PAIRS_PER_THREAD; it++) {
    const int j = j0 + it * PAIR_BLOCK;
    const ppf_vec3 p2 = ppf_mk3((double)nx0, (double)nx1, (double)nx2), n2 = ppf_mk3((double)nx3, (double)nx4, (double)nx5);
    {
      const uint32_t jn = (uint32_t)min(j + PAIR_BLOCK, n - 1) * 4u;
      nx0 = ldf(a.paired.x, jn); nx1 = ldf(a.paired.y, jn); nx2 = ldf(a.paired.z, jn);
      nx3 = ldf(a.paired.nx, jn); nx4 = ldf(a.paired.ny, jn); nx5 = ldf(a.paired.nz, jn);
    }
    if (j < n && (S2B || j != i_ref)) {
      /* match_S2B: the reference point itself is never paired, even when the edge cloud contains it
       * (bit-identical row), so edge == scene reduces exactly to match().  Values came from floats, so
       * comparing the doubles compares the float bits (no NaN/-0 cases in finite clouds). */
      const bool self_pair = S2B && p2.x == p1.x && p2.y == p1.y && p2.z == p1.z && n2.x == n1.x && n2.y == n1.y && n2.z == n1.z;
      bool skip = self_pair;
      if (__builtin_expect(a.pair_radius > 0.0, 0)) { /* PCL policy: neighbours within a radius only; the same fp64 distance the pair feature uses */
        const double dx = p2.x - p1.x, dy = p2.y - p1.y, dz = p2.z - p1.z;
        skip |= ppf_sqrt(dx * dx + dy * dy + dz * dz) > a.pair_radius;
      }
      if (!skip) {
        int32_t key[4];
        bool valid = true;
        if (DARBOUX) valid = pair_key_darboux(p1, n1, p2, n2, a.angle_step, a.dist_step, key);
        else pair_key(p1, n1, p2, n2, a.angle_step, a.dist_step, fk, key);
        if (valid) {
          int b;
          size_t ki;
          if (key_index(a.kd, key[0], key[1], key[2], key[3], &ki)) {
            b = a.key_lut[ki];
          } else if (a.key_exact) { /* NaN angles have bins of their own (key_index_nan); no model pair has any other key outside the table */
            b = key_index_nan(a.kd, key[0], key[1], key[2], key[3], &ki) ? a.key_lut[ki] : -1;
          } else { /* NaN features (INT_MIN bins) or pairs farther apart than the table covers */
            b = slot_to_bucket(a.slotmap, ppf_murmur_key16(key[0], key[1], key[2], key[3]) & a.slot_mask);
          }
          /* The reference skips a pair whose alpha_s is NaN BEFORE it counts it; for finite clouds it never is.  alpha_s
           * itself is computed later (k_group), only for the pairs that found a bucket (and a hit without one is retired there). */
          my_pairs += 1u; /* k_pairs_odd takes the pairs without an alpha_s off again, should there be any */
          if (b >= 0) {
            stash[it][tid] = make_uint2((uint32_t)b, (uint32_t)j);
            hit_mask |= 1u << it;
          }
        }
      }
    }
  }
  /* one returned atomic per WORKGROUP: exclusive scan of the per-lane hit counts over the wave, wave totals in LDS */
  const uint32_t mine = (uint32_t)__popc(hit_mask);
  uint32_t incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(incl, o);
    if (lane >= o) incl += y;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  uint32_t woff = 0, total = 0;
#pragma unroll
  for (int k = 0; k < PAIR_BLOCK / 64; k++) {
    const uint32_t w = wtot[k];
    if (k < wave) woff += w;
    total += w;
  }
  if (a.count_only) { /* cold workspace: only the number of hits is wanted (one 64-bit counter per stripe) */
    if (tid == 0 && total) {
      const uint32_t stripe = pool_stripe(blockIdx.y * gridDim.x + blockIdx.x, a.stripe_bits);
      atomicAdd(reinterpret_cast<unsigned long long*>(&a.cursors[stripe * CUR_STRIDE]), (unsigned long long)total);
    }
    return;
  }
  if (tid == 0) {
    uint32_t base = 0xFFFFFFFFu;
    if (total) {
      const uint32_t stripe = pool_stripe(blockIdx.y * gridDim.x + blockIdx.x, a.stripe_bits);
      const uint32_t b = atomicAdd(&a.cursors[stripe * CUR_STRIDE], total);
      if (b <= a.stripe_cap && total <= a.stripe_cap - b) base = stripe * a.stripe_cap + b;
      else atomicOr(&a.cursors[CUR_OVERFLOW], 1u); /* pool estimate too small: the host repeats the call */
    }
    wg_base = base;
    a.chunk_desc[(size_t)r * a.pair_chunks + blockIdx.x] = base == 0xFFFFFFFFu ? make_uint2(0u, 0u) : make_uint2(base, total);
  }
  __syncthreads();
  const uint32_t base = wg_base;
  if (base != 0xFFFFFFFFu) {
    uint32_t pos = base + woff + incl - mine;
#pragma unroll 1
    for (int it = 0; it < PAIRS_PER_THREAD; it++) {
      if (hit_mask & (1u << it)) a.raw[pos++] = stash[it][tid];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) my_pairs += __shfl_down(my_pairs, o);
  if (lane == 0 && my_pairs) atomicAdd(&a.pairs[a.ref_base + r], my_pairs);
}

/* The reference counts a pair only after its alpha_s has turned out not to be NaN (`continue` before the count).  For ordinary
 * numbers every pair has one and k_pairs counts without looking; when k_frames has seen a frame or a paired point that is no
 * ordinary number, this kernel takes the pairs k_pairs counted and the reference would not off the totals again (the same skip
 * rules, then the alpha's existence as k_group decides it for a hit).  A fixed small grid that returns at once otherwise: in
 * k_pairs itself the branch cost eight registers and with them a wave per SIMD (0.58 -> 0.64 ms on C2). */
template <bool DARBOUX, bool S2B>
__global__ __launch_bounds__(256) void k_pairs_odd(MatchArgs a) {
  if (a.cursors[CUR_ODDVALUES] == 0u) return;
  const int lane = threadIdx.x & 63;
  for (int r = blockIdx.x; r < a.n_ref; r += gridDim.x) {
    const int i_ref = ref_row(a, r);
    const ppf_vec3 p1 = ld3(a.surf.x, a.surf.y, a.surf.z, i_ref), n1 = ld3(a.surf.nx, a.surf.ny, a.surf.nz, i_ref);
    const double* f = a.frames + (size_t)r * 12;
    unsigned long long gone = 0;
    for (int j = threadIdx.x; j < a.paired.n; j += blockDim.x) {
      if (!(S2B || j != i_ref)) continue;
      const ppf_vec3 p2 = ld3(a.paired.x, a.paired.y, a.paired.z, j), n2 = ld3(a.paired.nx, a.paired.ny, a.paired.nz, j);
      const bool self_pair = S2B && p2.x == p1.x && p2.y == p1.y && p2.z == p1.z && n2.x == n1.x && n2.y == n1.y && n2.z == n1.z;
      bool skip = self_pair;
      if (a.pair_radius > 0.0) {
        const double dx = p2.x - p1.x, dy = p2.y - p1.y, dz = p2.z - p1.z;
        skip |= ppf_sqrt(dx * dx + dy * dy + dz * dz) > a.pair_radius;
      }
      if (skip) continue;
      if (DARBOUX) {
        int32_t key[4];
        if (!pair_key_darboux(p1, n1, p2, n2, a.angle_step, a.dist_step, key)) continue;
      }
      const double qy = f[10] + (f[3] * p2.x + f[4] * p2.y + f[5] * p2.z);
      const double qz = f[11] + (f[6] * p2.x + f[7] * p2.y + f[8] * p2.z);
      if (!ppf_alpha_exists(qy, qz)) gone += 1ull;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gone += __shfl_down(gone, o);
    if (lane == 0 && gone) atomicAdd(&a.pairs[a.ref_base + r], 0ull - gone);
  }
}

/* raw hits per reference point = sum of its workgroups' counts */
__global__ __launch_bounds__(256) void k_ref_hits(MatchArgs a) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.n_ref) return;
  unsigned long long s = 0;
  for (int c = 0; c < a.pair_chunks; c++) s += a.chunk_desc[(size_t)r * a.pair_chunks + c].y;
  a.hit_count[r] = s;
}

/* cell of a hit: y = alpha_s * A/(4 pi) = Y + psi; cell = (Y + 8) * AGG_Q + floor(AGG_Q * psi) */
__device__ __forceinline__ uint32_t hit_cell(const double alpha_s, const double s64) {
  const double y = alpha_s * s64;
  const double Y = __builtin_floor(y);
  int p = (int)((y - Y) * (double)AGG_Q);
  p = min(max(p, 0), AGG_Q - 1);
  const int yy = min(max((int)Y + 8, 0), AGG_NY - 1);
  return (uint32_t)(yy * AGG_Q + p);
}

/*
 * k_group: one workgroup per reference point.  Per round (round_buckets bucket ids; one round unless the table
 * has more buckets than LDS counters): count the hits per bucket (alpha_s is evaluated here: a NaN alpha makes the
 * reference skip the pair, the hit is retired), scan, write the run table -- runs that will vote through the count
 * table first, so the long work items of k_vote are claimed early --, then scatter the payload of every hit to its
 * sorted position through the counters (any order inside a bucket: votes commute).
 * The reference point's raw hits sit in pair_chunks pieces of the pool; the passes walk them as ONE list (thread t
 * takes hits t, t + 512, ...); the counting pass only needs to know that a hit HAS an alpha_s (always, for finite clouds), the
 * scatter pass computes it.  Dynamic LDS: [round_buckets counters][pair_chunks + 1 prefix].
 */
__global__ __launch_bounds__(GROUP_BLOCK) void k_group(MatchArgs a) {
  extern __shared__ __align__(8) uint32_t gcnt[]; /* round_buckets counters, then cursors */
  __shared__ unsigned long long wsum[GROUP_BLOCK / 64];
  __shared__ uint32_t wtot[4][GROUP_BLOCK / 64];
  __shared__ uint32_t sh[5];
  const int r = a.perm_group ? (int)a.perm_group[blockIdx.x] : (int)blockIdx.x; /* most hits first: no long block at the tail */
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t* cpre = gcnt + ((a.round_buckets + 1) & ~1);                     /* pair_chunks + 1 (even offsets: the doubles behind stay 8-byte aligned) */
  uint32_t* cbase = cpre + ((a.pair_chunks + 2) & ~1);                      /* pair_chunks: where each piece starts in the raw pool (a copy of desc[].x: the
                                                                               walk below would otherwise wait for a global read before every pool read) */
  const uint32_t n_raw = (uint32_t)a.hit_count[r];
  const uint2* __restrict__ desc = a.chunk_desc + (size_t)r * a.pair_chunks;
  if (tid == 0) {
    uint32_t base = 0, ok = 1;
    if (n_raw) {
      base = atomicAdd(&a.cursors[CUR_SORTED], n_raw);
      if (!(base <= a.sorted_cap && n_raw <= a.sorted_cap - base)) { ok = 0; atomicOr(&a.cursors[CUR_OVERFLOW], 2u); }
    }
    sh[0] = base; sh[1] = ok;
  }
  if (wave == 1) { /* prefix of the pieces' hit counts, 64 pieces per step, while thread 0 waits for its atomic (one thread
                      summing them one by one was a chain of pair_chunks dependent global reads: a sixth of the kernel) */
    uint32_t run = 0;
    for (int c0 = 0; c0 < a.pair_chunks; c0 += 64) {
      const int c = c0 + lane;
      const uint32_t v = c < a.pair_chunks ? desc[c].y : 0u;
      uint32_t incl = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
      }
      if (c < a.pair_chunks) { cpre[c] = run + incl - v; cbase[c] = desc[c].x; }
      run += __shfl(incl, 63);
    }
    if (lane == 0) cpre[a.pair_chunks] = run;
  }
  __syncthreads();
  const uint32_t hit_base = sh[0];
  const bool ok = sh[1] != 0;
  const double* __restrict__ fr = a.frames + (size_t)r * 12;
  const double R10 = fr[3], R11 = fr[4], R12 = fr[5], R20 = fr[6], R21 = fr[7], R22 = fr[8], ty = fr[10], tz = fr[11];
  const double s64 = (double)a.num_angles / (4 * PPF_PI);
  const uint32_t agg_min = a.agg_min_hits > 0 ? (uint32_t)a.agg_min_hits : 0xFFFFFFFFu;
  const uint32_t n_list = ok ? n_raw : 0u;
  const bool check_alpha = a.cursors[CUR_ODDVALUES] != 0u;
  unsigned long long w = 0;
  uint32_t placed = 0, runs_written = 0, tables_given = 0;
  for (int round = 0; round < a.n_rounds; round++) {
    const uint32_t b0 = (uint32_t)round * (uint32_t)a.round_buckets;
    const uint32_t nb = min((uint32_t)a.round_buckets, (uint32_t)a.n_buckets - b0);
    for (uint32_t k = tid; k < nb; k += GROUP_BLOCK) gcnt[k] = 0;
    __syncthreads();
    { /* GROUP_MLP hits per thread and step: their pool reads, then their gathers, are all in flight before the first
       * alpha_s is computed (one hit at a time the loop is a chain of two memory latencies per hit) */
      int c = 0;
      for (uint32_t g0 = tid; g0 < n_list; g0 += GROUP_MLP * GROUP_BLOCK) {
        uint32_t at[GROUP_MLP];
        uint2 key[GROUP_MLP];
#pragma unroll
        for (int u = 0; u < GROUP_MLP; u++) {
          const uint32_t g = g0 + (uint32_t)u * GROUP_BLOCK;
          /* unconditional reads at a clamped position (n_list > 0 here): the GROUP_MLP pool reads go out back to back, where
           * reads under `if (g < n_list)` were each waited for before the next one was issued */
          const uint32_t gc = min(g, n_list - 1u);
          while (gc >= cpre[c + 1]) c++;
          at[u] = cbase[c] + (gc - cpre[c]);
          key[u] = a.raw[at[u]];
          if (g >= n_list) key[u].x = 0xFFFFFFFFu;
        }
        ppf_vec3 p2[GROUP_MLP];
        bool in[GROUP_MLP];
#pragma unroll
        for (int u = 0; u < GROUP_MLP; u++) {
          in[u] = key[u].x - b0 < nb; /* retired hits (0xFFFFFFFF), the clamped repeats and other rounds' buckets fall out here */
          p2[u] = ppf_mk3(0.0, 0.0, 0.0);
          if (in[u] && check_alpha) p2[u] = ld3(a.paired.x, a.paired.y, a.paired.z, (int)key[u].y);
        }
#pragma unroll
        for (int u = 0; u < GROUP_MLP; u++) {
          if (in[u]) {
            bool has_alpha = true; /* ordinary numbers everywhere (k_frames looked): every hit has one */
            if (check_alpha) {
              const double qy = ty + (R10 * p2[u].x + R11 * p2[u].y + R12 * p2[u].z);
              const double qz = tz + (R20 * p2[u].x + R21 * p2[u].y + R22 * p2[u].z);
              has_alpha = ppf_alpha_exists(qy, qz);
            }
            if (has_alpha) { /* alpha_s itself is computed once, by the pass that stores it */
              atomicAdd(&gcnt[key[u].x - b0], 1u);
            } else {
              a.raw[at[u]].x = 0xFFFFFFFFu; /* retired: matches no round */
            }
          }
        }
      }
    }
    __syncthreads();
    /* each thread owns a contiguous slice of the counters: hits, count-table runs, direct runs in it */
    const uint32_t per = (nb + GROUP_BLOCK - 1) / GROUP_BLOCK;
    const uint32_t k0 = min((uint32_t)tid * per, nb), k1 = min(k0 + per, nb);
    uint32_t th = 0, tH = 0, tL = 0, tT = 0; /* hits, many-hit runs, few-hit runs, count tables (one per AGG_SUB hits of a many-hit run) */
    for (uint32_t k = k0; k < k1; k++) {
      const uint32_t c = gcnt[k];
      th += c;
      if (c) { if (c >= agg_min) { tH++; tT += (c + AGG_SUB - 1) / AGG_SUB; } else tL++; }
    }
    uint32_t ih = th, iH = tH, iL = tL, iT = tT;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y0 = __shfl_up(ih, o), y1 = __shfl_up(iH, o), y2 = __shfl_up(iL, o), y3 = __shfl_up(iT, o);
      if (lane >= o) { ih += y0; iH += y1; iL += y2; iT += y3; }
    }
    if (lane == 63) { wtot[0][wave] = ih; wtot[1][wave] = iH; wtot[2][wave] = iL; wtot[3][wave] = iT; }
    __syncthreads();
    uint32_t oh = ih - th, oH = iH - tH, oL = iL - tL, oT = iT - tT, nh = 0, nH = 0, nL = 0, nT = 0;
#pragma unroll
    for (int k = 0; k < GROUP_BLOCK / 64; k++) {
      const uint32_t v0 = wtot[0][k], v1 = wtot[1][k], v2 = wtot[2][k], v3 = wtot[3][k];
      if (k < wave) { oh += v0; oH += v1; oL += v2; oT += v3; }
      nh += v0; nH += v1; nL += v2; nT += v3;
    }
    if (tid == 0) {
      const uint32_t R = nH + nL;
      uint32_t rb = 0, okr = 1;
      if (R) {
        rb = atomicAdd(&a.cursors[CUR_RUNS], R);
        if (!(rb <= a.run_cap && R <= a.run_cap - rb)) { okr = 0; atomicOr(&a.cursors[CUR_OVERFLOW], 4u); }
      }
      uint32_t tb = 0, okt = 1;
      if (nT) {
        tb = atomicAdd(&a.cursors[CUR_TABLES], nT);
        if (!(tb <= a.table_cap && nT <= a.table_cap - tb)) { okt = 0; atomicOr(&a.cursors[CUR_OVERFLOW], 8u); }
      }
      okr &= okt; /* a block of runs whose tables have no room is left out like one that has no room itself: the call is repeated */
      sh[4] = tb;
      sh[2] = rb; sh[3] = okr;
      a.run_blocks[(size_t)r * a.n_rounds + round] = okr ? make_uint2(rb, R) : make_uint2(0u, 0u);
    }
    __syncthreads();
    {
      const uint32_t rb = sh[2];
      const bool okr = sh[3] != 0;
      uint32_t pos = oh, rh = oH, rl = nH + oL, tt = sh[4] + oT;
      for (uint32_t k = k0; k < k1; k++) {
        const uint32_t c = gcnt[k];
        gcnt[k] = placed + pos; /* the bucket's cursor */
        if (c) {
          const bool many = c >= agg_min;
          const uint32_t idx = many ? rh++ : rl++;
          if (okr) a.runs[rb + idx] = make_uint4(b0 + k, hit_base + placed + pos, c, many ? tt : 0u);
          if (many) {
            for (uint32_t h0 = 0; h0 < c; h0 += AGG_SUB, tt++)
              if (okr) a.table_desc[tt] = make_uint2(hit_base + placed + pos + h0, min((uint32_t)AGG_SUB, c - h0));
          }
          /* what this run will cost k_vote (its launch order: longest first), in direct votes: a run that votes through count
           * tables pays per table what PPF_AGG_MIN_HITS direct hits would, whatever its hits */
          w += (unsigned long long)(many ? ((c + AGG_SUB - 1) / AGG_SUB) * (uint32_t)PPF_COST_TABLE : max(c, (uint32_t)PPF_COST_MIN_HITS)) * a.bucket_total[b0 + k] + (uint32_t)PPF_COST_ITEM;
        }
        pos += c;
      }
    }
    __syncthreads();
    {
      int c = 0;
      for (uint32_t g0 = tid; g0 < n_list; g0 += GROUP_MLP * GROUP_BLOCK) {
        uint2 key[GROUP_MLP];
#pragma unroll
        for (int u = 0; u < GROUP_MLP; u++) {
          const uint32_t g = g0 + (uint32_t)u * GROUP_BLOCK;
          const uint32_t gc = min(g, n_list - 1u);
          while (gc >= cpre[c + 1]) c++;
          key[u] = a.raw[cbase[c] + (gc - cpre[c])];
          if (g >= n_list) key[u].x = 0xFFFFFFFFu;
        }
        double as[GROUP_MLP];
        ppf_vec3 p2[GROUP_MLP];
        bool in[GROUP_MLP];
#pragma unroll
        for (int u = 0; u < GROUP_MLP; u++) {
          in[u] = key[u].x - b0 < nb;
          as[u] = 0.0;
          p2[u] = ld3(a.paired.x, a.paired.y, a.paired.z, (int)key[u].y); /* every hit names a paired point: read all, use those of this round */
        }
        uint32_t gi[GROUP_MLP];
#pragma unroll
        for (int u = 0; u < GROUP_MLP; u++) {
          gi[u] = 0u;
          if (in[u]) {
            const double qy = ty + (R10 * p2[u].x + R11 * p2[u].y + R12 * p2[u].z);
            const double qz = tz + (R20 * p2[u].x + R21 * p2[u].y + R22 * p2[u].z);
            (void)ppf_alpha_in_frame(qy, qz, &as[u]);
            gi[u] = hit_base + atomicAdd(&gcnt[key[u].x - b0], 1u);
          }
        }
#pragma unroll
        for (int u = 0; u < GROUP_MLP; u++) {
          if (in[u]) {
            a.s_a64[gi[u]] = as[u];
            a.s_cell[gi[u]] = (uint16_t)hit_cell(as[u], s64);
          }
        }
      }
    }
    placed += nh;
    runs_written += nH + nL;
    tables_given += nT;
    __syncthreads();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o);
  if (lane == 0) wsum[wave] = w;
  __syncthreads();
  if (tid == 0) {
    unsigned long long t = 0;
    for (int k = 0; k < GROUP_BLOCK / 64; k++) t += wsum[k];
    a.work[r] = t;
    atomicAdd(&a.tally[1], (unsigned long long)placed);
    atomicAdd(&a.tally[2], (unsigned long long)runs_written);
    atomicAdd(&a.tally[5], (unsigned long long)tables_given);
  }
}

/*
 * Accumulator layout in LDS (words):  [guard] [ceil(tile_refs / 2) rows x A cells] [1 word]
 *   A cell is 16 bits wide: row r < H = ceil(tile_refs / 2) counts in the low halves of its word row, row r + H in the
 *   high halves of the same words (32-bit cells, the repeat after an overflow: one half of the rows per workgroup).
 *   Rows follow each other without a gap, so the "bin == A" the reference can produce (it indexes corrI*A + alpha_index
 *   without a range check: alpha_index == A is the next model reference point's bin 0) lands where the reference
 *   puts it by itself; only the last low-half row has its successor in the high halves: its bin A is the word behind
 *   the rows, added to row H's bin 0 when the accumulator is scanned (the same word's high half takes the bin A of
 *   the tile's last row, which belongs to the next tile: see the mirrored entries of the table).
 *   The guard words below cell 0 take every vote that must not count: mirrored spill entries
 *   (word offset GW-A) with any bin other than A, and the lanes past the end of a bucket (word = lane).
 *   With it the vote needs no range check at all.  Entry offsets are bytes from the guard's start.
 *
 * Two direct votes (one pair record) = 2 x v_fma_f32 (q'), 2 x v_cvt_i32_f32 (k), 2 x v_fract_f32 + a shared v_min3
 * reduction (guard band), 2 x v_lshl_add_u32 (byte address), 2 x ds_add_u32.
 *   q' = alpha_m*S + Ohg,  Ohg = A/2 - alpha_s*S + G  (folded once per hit), k = trunc(q')
 *   k is the reference's integer whenever fract(q') >= 2G; otherwise the lane
 *   re-evaluates the fp64 chain.  That happens for ~3e-5 of the votes, so the re-evaluation is
 *   taken once per batch of U entries and only when some lane of the wave needs it.
 */
typedef __attribute__((address_space(3))) unsigned char lds_byte; /* explicit LDS pointers: 32-bit arithmetic, ds_* ops */
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4_t lds_u32x4; /* ds_read_b128 / ds_write_b128 */
typedef __attribute__((address_space(3))) uint32_t lds_u32;

__device__ __forceinline__ void lds_add(const uint32_t addr, const uint32_t v) {
  (void)__hip_atomic_fetch_add((lds_u32*)(uintptr_t)addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void vote_one(const uint32_t p, const int k, const uint32_t inc) {
#if PPF_MOCK_PAIRBINS == 1 /* 2: only the counted atomics change (what halving THEIR number is worth on its own) */
  (void)inc;
  const uint32_t word = p + (((uint32_t)k & ~1u) << 1);                /* v_and + v_lshl_add */
  const uint32_t add = __builtin_amdgcn_ubfe((uint32_t)k, 0u, 1u) * 0xFFFFu + 1u; /* v_bfe / v_and + v_mad: 1 or 0x10000 */
  lds_add(word, add);
#else
  lds_add(p + ((uint32_t)k << 2), inc);
#endif
}
__device__ __forceinline__ uint32_t lds_add_rtn(const uint32_t addr, const uint32_t v) {
  return __hip_atomic_fetch_add((lds_u32*)(uintptr_t)addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t lds_ld(const uint32_t addr) { return *(volatile lds_u32*)(uintptr_t)addr; }
/* two consecutive words at a 4-byte aligned address (ds_read2_b32; an 8-byte typed load would let the compiler assume
 * 8-byte alignment, which the 28-byte table rows and the cell table do not have) */
struct __attribute__((packed, aligned(4))) lds_pair_t { uint32_t x, y; };
typedef __attribute__((address_space(3))) lds_pair_t lds_pair;
__device__ __forceinline__ uint2 lds_ld2(const uint32_t addr) {
  const lds_pair* p = (const lds_pair*)(uintptr_t)addr;
  return make_uint2(p->x, p->y);
}
__device__ __forceinline__ void lds_st(const uint32_t addr, const uint32_t v) { *(volatile lds_u32*)(uintptr_t)addr = v; }
__device__ __forceinline__ void lds_st8(const uint32_t addr, const uint32_t v) { *(volatile lds_byte*)(uintptr_t)addr = (unsigned char)v; }
__device__ __forceinline__ uint32_t lds_ld8(const uint32_t addr) { return *(volatile lds_byte*)(uintptr_t)addr; }
/* order the LDS phases of one wave (LDS executes a wave's operations in order; this keeps the compiler from moving them) */
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

/* WRAP (policy alpha_range_2pi): q' = (alpha_m - alpha_s)*A/(2pi) + 3A/2 + G lies in (A/2, 5A/2); the bin of the wrapped
 * difference is trunc(q') - A reduced into [0, A).  The wrap thresholds sit on integers of q' - G, i.e. on bin edges: the
 * guard band that sends near-edge votes to the fp64 chain covers them as well. */
template <bool WRAP>
__device__ __forceinline__ int vote_wrap(int t, const int A) {
  if constexpr (WRAP) {
    t -= A;
    t = t < 0 ? t + A : t;
    t = t >= A ? t - A : t;
  }
  return t;
}
/* The exact fp64 bin of a vote inside the guard band: 3e-5 of the votes, but dozens of call sites (every record group of every
 * vote-loop instantiation).  Kept OUT OF LINE: inlined, the division sequences made up half of k_vote's code. */
__device__ __attribute__((noinline)) int vote_bin_exact_4pi(const float am, const double as, const int A) { return ppf_alpha_bin_exact(am, as, A); }
__device__ __attribute__((noinline)) int vote_bin_exact_2pi(const float am, const double as, const int A) { return ppf_alpha_bin_exact_2pi(am, as, A); }
template <bool WRAP>
__device__ __forceinline__ int vote_bin_exact(const float am, const double as, const int A) {
  if constexpr (WRAP) return vote_bin_exact_2pi(am, as, A);
  else return vote_bin_exact_4pi(am, as, A);
}

/* record `idx` behind a wave-uniform pointer: the index goes in as a 32-bit byte offset next to the scalar base (an item has at
 * most a few thousand records), not as a 64-bit address formed per load */
__device__ __forceinline__ uint4 rec_at(const uint4* __restrict__ base, const uint32_t idx) {
  return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + idx * 16u);
}
template <int U>
__device__ __forceinline__ void load_records(uint4* rec, const uint4* __restrict__ src, const uint32_t e0, const int lane) {
#pragma unroll
  for (int u = 0; u < U; u++) rec[u] = rec_at(src, e0 + (uint32_t)(u * 64 + lane));
}

/* the bins of 2U votes against one hit; frmin = smallest fractional part (guard-band check) */
template <int U, bool WRAP>
__device__ __forceinline__ void vote_bins(const uint4* rec, const float S, const float Ohg, const int A, int (&ka)[U], int (&kb)[U], float& frmin_out) {
  float fa[U], fb[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    /* two scalar v_fma_f32: measured faster than one v_pk_fma_f32 on gfx950 */
    const float qa = __builtin_fmaf(__uint_as_float(rec[u].z), S, Ohg);
    const float qb = __builtin_fmaf(__uint_as_float(rec[u].w), S, Ohg);
    ka[u] = vote_wrap<WRAP>((int)qa, A);
    kb[u] = vote_wrap<WRAP>((int)qb, A);
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
}
/* rare path: votes within the guard band of a bin edge get the exact fp64 bin */
template <int U, bool WRAP>
__device__ __forceinline__ void vote_fix(const uint4* rec, const float S, const float Ohg, const double* __restrict__ asd,
                                         const float G2, const int A, const float frmin, int (&ka)[U], int (&kb)[U]) {
  if (__builtin_expect(__any(frmin < G2), 0)) {
    const double as = *asd; /* exact alpha_s of this hit, only needed here */
#pragma unroll
    for (int u = 0; u < U; u++) {
      uint32_t za = rec[u].z, zb = rec[u].w;
      asm volatile("" : "+v"(za), "+v"(zb)); /* keep the fp64 conversions of the rare path out of the hot loop */
      const float aa = __uint_as_float(za), ab = __uint_as_float(zb);
      if (__builtin_amdgcn_fractf(__builtin_fmaf(aa, S, Ohg)) < G2) ka[u] = vote_bin_exact<WRAP>(aa, as, A);
      if (__builtin_amdgcn_fractf(__builtin_fmaf(ab, S, Ohg)) < G2) kb[u] = vote_bin_exact<WRAP>(ab, as, A);
    }
  }
}
/* the 2U atomics of one hit: LDS address = row's bin 0 + 4k (one v_lshl_add_u32), ds_add_u32 */
template <int U>
__device__ __forceinline__ void vote_issue(const uint32_t (&pa)[U], const uint32_t (&pb)[U], const uint32_t (&ia)[U], const uint32_t (&ib)[U],
                                           const int (&ka)[U], const int (&kb)[U], const int n_valid) {
#pragma unroll
  for (int u = 0; u < U; u++) {
    if (u < n_valid) {
      vote_one(pa[u], ka[u], ia[u]);
      vote_one(pb[u], kb[u], ib[u]);
    }
  }
}
/* one pipeline stage: the atomics of the previous hit (bins pka/pkb) under the bin arithmetic of hit hh (-> nka/nkb) */
template <int U, bool WRAP>
__device__ __forceinline__ void vote_stage(const uint4* rec, const int n_valid, const float S, const float ohg_v, const int hh,
                                           const double* __restrict__ asd, const float G2, const int A, const uint32_t (&pa)[U],
                                           const uint32_t (&pb)[U], const uint32_t (&ia)[U], const uint32_t (&ib)[U], const int (&pka)[U],
                                           const int (&pkb)[U], int (&nka)[U], int (&nkb)[U]) {
  float frmin;
  const float Ohg = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ohg_v), hh));
  vote_issue<U>(pa, pb, ia, ib, pka, pkb, n_valid);
  vote_bins<U, WRAP>(rec, S, Ohg, A, nka, nkb, frmin);
#pragma unroll
  for (int i = 0; i < 2 * U; i++) {
    __builtin_amdgcn_sched_group_barrier(0x002, PPF_PIPE_VALU, 0); /* VALU */
    __builtin_amdgcn_sched_group_barrier(0x080, 1, 0); /* DS */
  }
  vote_fix<U, WRAP>(rec, S, Ohg, asd + hh, G2, A, frmin, nka, nkb);
}
/* all hits of a direct work item against one register batch of records; two sets of bins alternate so that a set is only
 * overwritten a full stage after the atomics that used it were issued */
/* increment of a vote for the row a record names: bit 0 of the row offset says which half of the word the row owns */
struct VoteInc {
  uint32_t lo, hi; /* 16-bit cells: 1, 0x10000; 32-bit cells, pass h: 1 for the rows of half h, 0 for the others */
};
/* mask ? b : a per bit (mask 0 or all ones).  Used where a plain `cond ? s.f1 : s.f0` on two fields of one struct would be
 * turned into an indexed load from a stack copy of the struct, which drags the whole struct -- a prefetched record, say --
 * through scratch memory with a wait for everything in flight. */
__device__ __forceinline__ uint32_t bit_select(const uint32_t mask, const uint32_t a, const uint32_t b) { return (mask & b) | (~mask & a); }

/* bit select, not `bit ? vi.hi : vi.lo`: the compiler turns that select of two struct fields into an indexed load from the
 * struct's stack copy (a scratch_load + s_waitcnt vmcnt(0) per entry, in the middle of the record prefetches) */
__device__ __forceinline__ uint32_t vote_inc(const VoteInc& vi, const uint32_t row_code) {
  const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)row_code, 0u, 1u); /* 0 or all ones */
  return (m & vi.hi) | (~m & vi.lo);
}

template <int U, bool WRAP>
__device__ __forceinline__ void vote_hits(const uint32_t acc_base, const VoteInc& vi, const uint4* rec, const int n_valid, const float S,
                                          const float ohg_v, const int nh, const double* __restrict__ asd, const float G2,
                                          const int A) {
  uint32_t pa[U], pb[U], ia[U], ib[U];
  int ka0[U], kb0[U], ka1[U], kb1[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    pa[u] = acc_base + (rec[u].x & ROW_OFFSET_MASK);
    pb[u] = acc_base + (rec[u].y & ROW_OFFSET_MASK);
    ia[u] = vote_inc(vi, rec[u].x);
    ib[u] = vote_inc(vi, rec[u].y);
    asm volatile("" : "+v"(pa[u]), "+v"(pb[u]), "+v"(ia[u]), "+v"(ib[u])); /* computed once per batch, not rematerialised per vote */
  }
  {
    float frmin;
    const float Ohg = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ohg_v), 0));
    vote_bins<U, WRAP>(rec, S, Ohg, A, ka0, kb0, frmin);
    vote_fix<U, WRAP>(rec, S, Ohg, asd, G2, A, frmin, ka0, kb0);
  }
  int hh = 1;
  for (; hh + 1 < nh; hh += 2) {
    vote_stage<U, WRAP>(rec, n_valid, S, ohg_v, hh, asd, G2, A, pa, pb, ia, ib, ka0, kb0, ka1, kb1);
    vote_stage<U, WRAP>(rec, n_valid, S, ohg_v, hh + 1, asd, G2, A, pa, pb, ia, ib, ka1, kb1, ka0, kb0);
  }
  if (hh < nh) {
    vote_stage<U, WRAP>(rec, n_valid, S, ohg_v, hh, asd, G2, A, pa, pb, ia, ib, ka0, kb0, ka1, kb1);
    vote_issue<U>(pa, pb, ia, ib, ka1, kb1, n_valid);
  } else {
    vote_issue<U>(pa, pb, ia, ib, ka0, kb0, n_valid);
  }
}

/* Runs of at most 32 pair records in this tile (64 entries: more than half of all (tile, bucket) runs): one ENTRY per
 * lane instead of one pair record per lane, so the 64 lanes of the single group are filled twice as well and a hit costs
 * one fma/cvt/fract/lshl_add/ds_add instead of two of each.  Same bins, same guard band, same exact fallback. */
template <bool WRAP>
__device__ __forceinline__ void vote_hits_single(const uint32_t acc_base, const VoteInc& vi, const uint32_t row_code, const uint32_t alpha_bits,
                                                 const float S, const float ohg_v, const int nh, const double* __restrict__ asd,
                                                 const float G2, const int A) {
  uint32_t pr = acc_base + (row_code & ROW_OFFSET_MASK), inc = vote_inc(vi, row_code);
  asm volatile("" : "+v"(pr), "+v"(inc));
  const float am = __uint_as_float(alpha_bits);
  uint32_t adr_prev = 0;
  int k_prev = 0;
  for (int hh = 0; hh < nh; hh++) {
    const float Ohg = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ohg_v), hh));
#if PPF_MOCK_PAIRBINS == 1
    if (hh) vote_one(pr, k_prev, inc);
#else
    if (hh) lds_add(adr_prev, inc);
#endif
    const float q = __builtin_fmaf(am, S, Ohg);
    int k = vote_wrap<WRAP>((int)q, A);
    if (__builtin_expect(__any(__builtin_amdgcn_fractf(q) < G2), 0)) {
      const double as = asd[hh];
      uint32_t z = alpha_bits;
      asm volatile("" : "+v"(z));
      const float az = __uint_as_float(z);
      if (__builtin_amdgcn_fractf(__builtin_fmaf(az, S, Ohg)) < G2) k = vote_bin_exact<WRAP>(az, as, A);
    }
    adr_prev = pr + ((uint32_t)k << 2);
    k_prev = k;
    (void)k_prev;
  }
#if PPF_MOCK_PAIRBINS == 1
  if (nh > 0) vote_one(pr, k_prev, inc);
#else
  if (nh > 0) lds_add(adr_prev, inc);
#endif
}

/* ---- aggregated votes ------------------------------------------------------------------------------------------- */
struct AggConsts {
  uint32_t acc_base; /* LDS byte address of the guard region */
  uint32_t ws;       /* LDS byte address of this wave's scratch */
  float S, half_a, Og, G2;
  double s64, half_a64;
  int A;
  VoteInc vi;
  /* v_perm_b32 selectors that move count byte jj of a table word to where the row's half counts: selector of byte 0 plus
   * jj times a step (low half: byte 0; high half: byte 2; 32-bit cells, other half's pass: all zero) */
  uint32_t sel_lo, step_lo, sel_hi, step_hi;
};

/* Count table of one range of <= AGG_SUB hits (global indices g0 .. g0+ms of the sorted payload), built by one wave in
 * TBL_BUILD_BYTES of LDS at `ws` (layout above):
 *   T[q][j], q < AGG_Q, j = 0..16   hits with cell p < q and Y + 8 == j, plus hits with p > q and Y + 8 == j - 1: what an
 *                                   entry in cell q adds to its bin X + 8 - j (bytes; row AGG_Q stays zero)
 *   ce[p], ce[p+1]                  range of the hits of cell p in the cell-sorted copy (offsets / index): counters behind the
 *                                   table while it is built, two bytes of row p in the table itself */
__device__ __forceinline__ void table_build(const uint32_t ws, const float S, const float Og, const double* __restrict__ g_a64,
                                            const uint16_t* __restrict__ g_cell, const int ms, const int lane) {
  constexpr int PER = AGG_Q / 4; /* cells of one Y a lane owns in the histogram pass */
#pragma unroll
  for (int w = 0; w < PER / 4; w++) lds_st(ws + TBL_OFF_A32 + lane * PER + w * 4, 0u); /* byte counters cnt[Y][p]: 16 x AGG_Q bytes */
  for (int e = lane; e < AGG_Q + 1; e += 64) lds_st(ws + AGG_OFF_CE + e * 4, 0u);
  if (lane < AGG_ROW / 4) lds_st(ws + AGG_Q * AGG_ROW + lane * 4, 0u); /* the all-zero row */
  uint32_t cell[3], a32[3];
#pragma unroll
  for (int t = 0; t < 3; t++) {
    const int i = lane + 64 * t;
    const bool v = i < ms;
    cell[t] = v ? (uint32_t)g_cell[i] : 0xFFFFu;
    a32[t] = v ? __float_as_uint(Og - (float)g_a64[i] * S) : 0u; /* Ohg = A/2 + G - alpha_s*S of the direct arithmetic */
  }
  wave_lds_fence();
#pragma unroll
  for (int t = 0; t < 3; t++) {
    if (cell[t] != 0xFFFFu) {
      lds_add(ws + TBL_OFF_A32 + (cell[t] & ~3u), 1u << (8u * (cell[t] & 3u)));
      lds_add(ws + AGG_OFF_CE + (cell[t] % (uint32_t)AGG_Q) * 4, 1u);
    }
  }
  wave_lds_fence();
  /* lane L holds cnt[Y = L/4][p = PER*(L%4) .. +PER] */
  uint32_t c[PER], less[PER], more[PER];
#pragma unroll
  for (int w = 0; w < PER / 4; w++) {
    const uint32_t cw = lds_ld(ws + TBL_OFF_A32 + lane * PER + w * 4);
#pragma unroll
    for (int t = 0; t < 4; t++) c[w * 4 + t] = (cw >> (8 * t)) & 0xFFu;
  }
  uint32_t run = 0;
#pragma unroll
  for (int t = 0; t < PER; t++) { less[t] = run; run += c[t]; }
  const int quarter = lane & 3;
  const uint32_t t1 = __shfl_up(run, 1, 4), t2 = __shfl_up(run, 2, 4), t3 = __shfl_up(run, 3, 4);
  const uint32_t before = (quarter >= 1 ? t1 : 0u) + (quarter >= 2 ? t2 : 0u) + (quarter >= 3 ? t3 : 0u);
  const uint32_t tot = __shfl(before + run, 3, 4); /* hits with this Y */
#pragma unroll
  for (int t = 0; t < PER; t++) {
    less[t] += before;
    more[t] = tot - less[t] - c[t];
  }
  const int yy = lane >> 2;
#pragma unroll
  for (int t = 0; t < PER; t++) {
    const uint32_t mprev = __shfl_up(more[t], 4);
    const uint32_t v = less[t] + (lane >= 4 ? mprev : 0u);
    const uint32_t p = (uint32_t)(quarter * PER + t);
    lds_st8(ws + p * AGG_ROW + yy, v);
    if (yy == AGG_NY - 1) lds_st8(ws + p * AGG_ROW + AGG_NY, more[t]);
  }
  wave_lds_fence();
  /* cell ends: inclusive scan of the per-p counts (one cell per lane); the scatter below counts them down to the cell starts */
  uint32_t h = lane < AGG_Q ? lds_ld(ws + AGG_OFF_CE + lane * 4) : 0u;
#pragma unroll
  for (int o = 1; o < AGG_Q; o <<= 1) {
    const uint32_t y = __shfl_up(h, o);
    if (lane >= o) h += y;
  }
  wave_lds_fence();
  if (lane < AGG_Q) lds_st(ws + AGG_OFF_CE + lane * 4, h);
  if (lane == 0) lds_st(ws + AGG_OFF_CE + AGG_Q * 4, (uint32_t)ms);
  wave_lds_fence();
#pragma unroll
  for (int t = 0; t < 3; t++) {
    if (cell[t] != 0xFFFFu) {
      const uint32_t pos = lds_add_rtn(ws + AGG_OFF_CE + (cell[t] % (uint32_t)AGG_Q) * 4, 0xFFFFFFFFu) - 1u;
      lds_st(ws + TBL_OFF_A32 + pos * 4, a32[t]);
      lds_st8(ws + TBL_OFF_IDX + pos, (uint32_t)(lane + 64 * t));
    }
  }
  wave_lds_fence();
  /* the counters now hold the cell starts (entry AGG_Q: the number of hits): every row gets its cell's range; the all-zero row
   * of the entries that vote one by one gets all hits */
  if (lane < AGG_Q) {
    const uint2 se = lds_ld2(ws + AGG_OFF_CE + lane * 4);
    lds_st8(ws + lane * AGG_ROW + AGG_ROW_CE, se.x);
    lds_st8(ws + lane * AGG_ROW + AGG_ROW_CE + 1, se.y);
  }
  if (lane == 0) lds_st8(ws + AGG_Q * AGG_ROW + AGG_ROW_CE + 1, (uint32_t)ms);
  wave_lds_fence();
}

/* guard band of the direct bin arithmetic (see k_vote): the folded offsets a table stores carry it */
__device__ __forceinline__ float vote_guard_band(const int A) {
#ifdef PPF_FORCE_EXACT
  (void)A;
  return 1.0f; /* test build: every direct vote takes the fp64 chain */
#else
  return PPF_GUARD_REL * (float)A;
#endif
}

/* k_tables: the count tables of a batch's many-hit runs, one wave per table (k_group gave the tables out and wrote what
 * each one covers: a run's hits, AGG_SUB at a time).  A wave builds its table in LDS and copies it out.  Built ONCE here,
 * where k_vote used to rebuild it for every chunk of records, accumulator tile and half it met the run in. */
__global__ __launch_bounds__(TABLE_BLOCK) void k_tables(MatchArgs a) {
  __shared__ __align__(16) unsigned char tsm[(TABLE_BLOCK / 64) * TBL_BUILD_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t t = blockIdx.x * (TABLE_BLOCK / 64) + (uint32_t)wave;
  if (t >= a.table_cap) return;
  const uint2 d = a.table_desc[t];
  if (d.y == 0u) return; /* not given out (or its block of runs was left out: the call is repeated) */
  const uint32_t ws = (uint32_t)(uintptr_t)(lds_byte*)(tsm + wave * TBL_BUILD_BYTES);
  const int A = a.num_angles;
  const double s64 = (double)A / (4 * PPF_PI);
  const float S = (float)s64, Og = (float)(0.5 * (double)A + (double)vote_guard_band(A));
  table_build(ws, S, Og, a.s_a64 + d.x, a.s_cell + d.x, (int)d.y, lane);
  const lds_u32x4* src = (const lds_u32x4*)(uintptr_t)ws;
  uint4* __restrict__ dst = reinterpret_cast<uint4*>(a.tables + (size_t)t * TBL_BYTES);
  for (int q = lane; q < TBL_BYTES / 16; q += 64) {
    const u32x4_t v = src[q];
    dst[q] = make_uint4(v.x, v.y, v.z, v.w);
  }
}

/* cell of an entry: x = alpha_m*A/(4 pi) + A/2 = X + phi, q = floor(AGG_Q * phi) (row AGG_Q: the entry votes one by one) */
__device__ __forceinline__ void agg_cell(const AggConsts& k, const float am, int& X, int& q) {
  const float x = __builtin_fmaf(am, k.S, k.half_a);
  X = (int)x;
  const float qf = __builtin_amdgcn_fractf(x) * (float)AGG_Q;
  q = (int)qf;
  const float fq = __builtin_amdgcn_fractf(qf);
  /* fp32 error of x: <= 1.5e-6 bins (S rounding 6e-8*7.5, fma 9.5e-7) = 1.5e-6 * AGG_Q cells (4.8e-5 at 32, 9.6e-5 at 64);
   * band = three times that */
  constexpr float band = 4.7e-6f * (float)AGG_Q;
  const bool near = !(fq > band && fq < 1.0f - band);
  bool one_by_one = am < -3.1415925f; /* (float)-pi and below: x - y may be negative, where the reference truncates to 0 */
  if (__builtin_expect(__any(near), 0)) {
    if (near) {
      const double xd = (double)am * k.s64 + k.half_a64;
      const double Xd = __builtin_floor(xd);
      const double qd = (xd - Xd) * (double)AGG_Q;
      const double qfl = __builtin_floor(qd);
      const double fqd = qd - qfl;
      X = (int)Xd;
      q = (int)qfl;
      one_by_one |= !(fqd > 1e-7 && fqd < 1.0 - 1e-7);
    }
  }
  q = one_by_one ? AGG_Q : min(q, AGG_Q - 1);
}

/* agg_cell for the table build: the bits a row code carries (zero when numAngles is too fine for count tables) */
__device__ uint32_t agg_cell_bits(const float am, const int A) {
  if (A > AGG_MAX_ANGLES) return 0u;
  AggConsts k;
  k.s64 = (double)A / (4 * PPF_PI);
  k.S = (float)k.s64;
  k.half_a = 0.5f * (float)A;
  k.half_a64 = 0.5 * (double)A;
  int X, q;
  agg_cell(k, am, X, q);
  return ((uint32_t)X << ROW_X_SHIFT) | ((uint32_t)q << ROW_Q_SHIFT);
}

/* four consecutive 32-bit words at a 4-byte aligned global address (global_load_dwordx4 needs no more) */
struct __attribute__((packed, aligned(4))) gl_quad_t { uint32_t x, y, z, w; };
__device__ __forceinline__ uint4 gl_ld4(const uint32_t* __restrict__ p) {
  const gl_quad_t q = *reinterpret_cast<const gl_quad_t*>(p);
  return make_uint4(q.x, q.y, q.z, q.w);
}
/* ... word `w` behind a wave-uniform base, as a 32-bit byte offset */
__device__ __forceinline__ uint4 gl_ld4_at(const uint32_t* __restrict__ base, const uint32_t w) {
  return gl_ld4(reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(base) + w * 4u));
}

/* one own-cell vote of each entry of a pair record (hit k of the entries' cells), guard band and exact fallback included */
__device__ __forceinline__ void agg_own_vote(const AggConsts& k, const uint32_t rz, const uint32_t rw, const float am_a, const float am_b,
                                             const uint32_t oa, const uint32_t ob, const bool da, const bool db, const uint32_t pa,
                                             const uint32_t pb, const uint32_t inc_a, const uint32_t inc_b,
                                             const unsigned char* __restrict__ tbl, const double* __restrict__ g_a64, const uint32_t ha,
                                             const uint32_t hb) {
  const float xa = __builtin_fmaf(am_a, k.S, __uint_as_float(oa)), xb = __builtin_fmaf(am_b, k.S, __uint_as_float(ob));
  int ba = (int)xa, bb = (int)xb;
  const float fa = da ? __builtin_amdgcn_fractf(xa) : 1.0f, fb = db ? __builtin_amdgcn_fractf(xb) : 1.0f;
  if (__builtin_expect(__any(__builtin_fminf(fa, fb) < k.G2), 0)) { /* some lane may sit in the guard band of a bin edge */
    uint32_t za = rz, zb = rw;
    asm volatile("" : "+v"(za), "+v"(zb)); /* keep the fp64 conversions of the rare path out of the loop */
    if (fa < k.G2) ba = vote_bin_exact_4pi(__uint_as_float(za), g_a64[tbl[TBL_OFF_IDX + ha]], k.A);
    if (fb < k.G2) bb = vote_bin_exact_4pi(__uint_as_float(zb), g_a64[tbl[TBL_OFF_IDX + hb]], k.A);
  }
  if (da) vote_one(pa, ba, inc_a);
  if (db) vote_one(pb, bb, inc_b);
}

/* Own-cell state of the two entries of one pair record: the ranges of their cells in the table's cell-sorted hit list (an
 * entry that votes one by one: all hits) and the first four folded offsets of each, read straight from the table `tbl`.
 * Issued a block of records ahead of the votes that use it (the loads need the record's cells and the ranges in the wave's LDS). */
struct AggOwn {
  uint32_t wa, wb; /* last word of the entries' table rows: count 16, then [first, end) of the cell in the cell-sorted hit list */
  uint4 oa, ob;    /* offsets of hits first .. first+3 (may run past the cell, never past the table) */
};
__device__ __forceinline__ void agg_own_fetch(const AggConsts& k, const uint4 rec, const unsigned char* __restrict__ tbl, const int ms, AggOwn& o) {
  (void)ms; /* the row of the entries that vote one by one carries [0, hits) */
  const uint32_t qa = (rec.x >> ROW_Q_SHIFT) & ROW_Q_MASK, qb = (rec.y >> ROW_Q_SHIFT) & ROW_Q_MASK;
  o.wa = lds_ld(k.ws + qa * AGG_ROW + 16);
  o.wb = lds_ld(k.ws + qb * AGG_ROW + 16);
#if PPF_ABL_OWNCELL
  const uint32_t* __restrict__ toff = reinterpret_cast<const uint32_t*>(tbl + TBL_OFF_A32);
  o.oa = gl_ld4_at(toff, __builtin_amdgcn_ubfe(o.wa, 8u, 8u));
  o.ob = gl_ld4_at(toff, __builtin_amdgcn_ubfe(o.wb, 8u, 8u));
#else
  (void)tbl;
#endif
}

/* The two model entries of one pair record against the count table (rows and cell ranges in the wave's LDS, copied from the
 * table `tbl` the item works with): 17 counted atomics each, then one vote per hit of each entry's own cell (all hits for an
 * entry that votes one by one) with the direct arithmetic.  The offsets of those hits come straight from the table, four
 * per entry, fetched a block ahead (agg_own_fetch): at 64 cells an entry's cell holds three hits on average, so the LDS
 * sees no reads for them and most blocks need no second fetch.  `votes` counts the one-by-one votes. */
__device__ __forceinline__ void agg_pair(const AggConsts& k, const uint4 rec, const AggOwn& own, const unsigned char* __restrict__ tbl,
                                         const double* __restrict__ g_a64, uint32_t& votes) {
  const float am_a = __uint_as_float(rec.z), am_b = __uint_as_float(rec.w);
  /* X and cell of the two entries: evaluated by the table build (agg_cell_bits), carried in the row codes */
  const int Xa = (int)((rec.x >> ROW_X_SHIFT) & 31u), qa = (int)((rec.x >> ROW_Q_SHIFT) & ROW_Q_MASK);
  const int Xb = (int)((rec.y >> ROW_X_SHIFT) & 31u), qb = (int)((rec.y >> ROW_Q_SHIFT) & ROW_Q_MASK);
  const uint32_t ta = k.ws + (uint32_t)qa * AGG_ROW, tb = k.ws + (uint32_t)qb * AGG_ROW;
  const uint2 a01 = lds_ld2(ta), a23 = lds_ld2(ta + 8);
  const uint2 b01 = lds_ld2(tb), b23 = lds_ld2(tb + 8);
  const uint32_t a4 = own.wa, b4 = own.wb; /* read a block ahead, with the cell's range in bytes 1 and 2 */
  const uint2 ca = make_uint2(__builtin_amdgcn_ubfe(a4, 8u, 8u), __builtin_amdgcn_ubfe(a4, 16u, 8u));
  const uint2 cb = make_uint2(__builtin_amdgcn_ubfe(b4, 8u, 8u), __builtin_amdgcn_ubfe(b4, 16u, 8u));
  const uint32_t pa = k.acc_base + (rec.x & ROW_OFFSET_MASK), pb = k.acc_base + (rec.y & ROW_OFFSET_MASK);
  const bool ha = (rec.x & 1u) != 0, hb = (rec.y & 1u) != 0;
  const uint32_t inc_a = vote_inc(k.vi, rec.x), inc_b = vote_inc(k.vi, rec.y);
  uint32_t va = pa + (uint32_t)((Xa - 8) * 4), vb = pb + (uint32_t)((Xb - 8) * 4); /* bin X + 8 - j lives at v + (16 - j)*4 */
  asm volatile("" : "+v"(va), "+v"(vb)); /* keep these as the bases: every atomic below is base + immediate offset */
  const uint32_t wa[5] = {a01.x, a01.y, a23.x, a23.y, a4}, wb[5] = {b01.x, b01.y, b23.x, b23.y, b4};
  uint32_t sa[4], sb[4]; /* one v_perm_b32 per count: byte jj of the table word -> the half this entry's row owns */
  {
    const uint32_t sa0 = ha ? k.sel_hi : k.sel_lo, sta = ha ? k.step_hi : k.step_lo;
    const uint32_t sb0 = hb ? k.sel_hi : k.sel_lo, stb = hb ? k.step_hi : k.step_lo;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) { sa[jj] = sa0 + (uint32_t)jj * sta; sb[jj] = sb0 + (uint32_t)jj * stb; }
  }
#if PPF_ABL_COUNTED
#if PPF_MOCK_PAIRBINS == 3 /* nine 64-bit atomics (two adjacent words each), one v_perm per count as now: timing of the ds_add_u64 form */
  {
    const uint32_t va8 = va & ~7u, vb8 = vb & ~7u;
#pragma unroll
    for (int j = 0; j <= AGG_NY; j += 2) {
      const int j1 = j + 1 > AGG_NY ? j : j + 1;
      const unsigned long long da = ((unsigned long long)__builtin_amdgcn_perm(0u, wa[j1 >> 2], sa[j1 & 3]) << 32) | __builtin_amdgcn_perm(0u, wa[j >> 2], sa[j & 3]);
      const unsigned long long db = ((unsigned long long)__builtin_amdgcn_perm(0u, wb[j1 >> 2], sb[j1 & 3]) << 32) | __builtin_amdgcn_perm(0u, wb[j >> 2], sb[j & 3]);
      typedef __attribute__((address_space(3))) unsigned long long lds_u64;
      (void)__hip_atomic_fetch_add((lds_u64*)(uintptr_t)(va8 + (uint32_t)((AGG_NY - j) * 4)), da, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      (void)__hip_atomic_fetch_add((lds_u64*)(uintptr_t)(vb8 + (uint32_t)((AGG_NY - j) * 4)), db, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
#elif PPF_MOCK_PAIRBINS
#pragma unroll
  for (int j = 0; j <= AGG_NY; j += 2) { /* 9 atomics, one two-source v_perm each */
    lds_add(va + (uint32_t)((AGG_NY - j) * 2), __builtin_amdgcn_perm(wa[(j + 1 > AGG_NY ? j : j + 1) >> 2], wa[j >> 2], sa[(j >> 1) & 3]));
    lds_add(vb + (uint32_t)((AGG_NY - j) * 2), __builtin_amdgcn_perm(wb[(j + 1 > AGG_NY ? j : j + 1) >> 2], wb[j >> 2], sb[(j >> 1) & 3]));
  }
#else
#pragma unroll
  for (int j = 0; j <= AGG_NY; j++) {
    lds_add(va + (uint32_t)((AGG_NY - j) * 4), __builtin_amdgcn_perm(0u, wa[j >> 2], sa[j & 3]));
    lds_add(vb + (uint32_t)((AGG_NY - j) * 4), __builtin_amdgcn_perm(0u, wb[j >> 2], sb[j & 3]));
  }
#endif
#endif
  const uint32_t na = ca.y - ca.x, nb = cb.y - cb.x;
  votes += na + nb;
#if PPF_ABL_OWNCELL
  if (__any((na > 0u) | (nb > 0u))) {
    /* the first four hits of the cells, from the offsets fetched a block ahead: straight-line code, so that the wait for
     * them lets the loads issued since (the next blocks' records and offsets) stay in flight -- as the head of a loop that
     * also re-fetches, it was a wait for everything */
#define PPF_OWN_VOTE(OA, OB, J) \
    agg_own_vote(k, rec.z, rec.w, am_a, am_b, OA, OB, (J) < na, (J) < nb, pa, pb, inc_a, inc_b, tbl, g_a64, ca.x + (J), cb.x + (J))
    PPF_OWN_VOTE(own.oa.x, own.ob.x, 0u);
    if (__any((1u < na) | (1u < nb))) {
      PPF_OWN_VOTE(own.oa.y, own.ob.y, 1u);
      if (__any((2u < na) | (2u < nb))) {
        PPF_OWN_VOTE(own.oa.z, own.ob.z, 2u);
        if (__any((3u < na) | (3u < nb))) {
          PPF_OWN_VOTE(own.oa.w, own.ob.w, 3u);
          if (__builtin_expect(__any((4u < na) | (4u < nb)), 0)) {
            /* a cell with more than four hits (an entry that votes one by one: all of them): four more at a time; lanes that
             * are done re-read their last words */
            const uint32_t* __restrict__ toff = reinterpret_cast<const uint32_t*>(tbl + TBL_OFF_A32);
            for (uint32_t done = 4u;;) {
              const uint4 oa = gl_ld4_at(toff, min(ca.x + done, (uint32_t)AGG_SUB)), ob = gl_ld4_at(toff, min(cb.x + done, (uint32_t)AGG_SUB));
              PPF_OWN_VOTE(oa.x, ob.x, done + 0u);
              if (!__any((done + 1u < na) | (done + 1u < nb))) break;
              PPF_OWN_VOTE(oa.y, ob.y, done + 1u);
              if (!__any((done + 2u < na) | (done + 2u < nb))) break;
              PPF_OWN_VOTE(oa.z, ob.z, done + 2u);
              if (!__any((done + 3u < na) | (done + 3u < nb))) break;
              PPF_OWN_VOTE(oa.w, ob.w, done + 3u);
              done += 4u;
              if (!__any((done < na) | (done < nb))) break;
            }
          }
        }
      }
    }
#undef PPF_OWN_VOTE
  }
#endif
}

/* a staged run of k_vote's run table in LDS: its work items' exclusive prefix, this tile's record range of its bucket, its
 * hits, its first count table.  One record per run (not an array per field): a look-up reads the six words behind ONE address,
 * and the size of the staging area (MatchArgs::run_seg) moves no other array */
constexpr int SEG_WORDS = 6, SEG_PREFIX = 0, SEG_OFF = 1, SEG_CNT = 2, SEG_HIT = 3, SEG_M = 4, SEG_TBL = 5;
__host__ __device__ constexpr size_t vote_seg_bytes(int run_seg) { return ((size_t)(run_seg + 1) * SEG_WORDS * 4 + 15) / 16 * 16; } /* + the sentinel; what follows stays 16-byte aligned */
/* one work item of k_vote, located and with its first loads issued (wave-uniform fields live in scalar registers) */
struct VoteItem {
  const uint4* src; /* first record of the item */
  uint32_t c;       /* records */
  uint32_t g0;      /* first sorted hit */
  int nh;           /* hits */
  bool agg;         /* votes through the count table */
  double a64;       /* direct items: alpha_s of hit g0 + lane (lanes < nh) */
  uint4 rec0;       /* record min(lane, c-1) (items of <= 32 records: record min(lane/2, c-1), the one-entry-per-lane layout) */
  const unsigned char* tbl; /* count-table items: the table of the item's run and hit range (k_tables) */
};

#ifndef PPF_PREFETCH
#define PPF_PREFETCH 2 /* loads issued for the NEXT work item while the current one votes: 2 = hits' alpha_s and first records, 1 = alpha_s, 0 = none */
#endif
__device__ __forceinline__ void vote_fetch_hits(VoteItem& it, const int lane, const MatchArgs& a) {
  it.a64 = (!it.agg && lane < it.nh) ? a.s_a64[it.g0 + lane] : 0.0;
}
__device__ __forceinline__ void vote_fetch_records(VoteItem& it, const int lane) {
  const uint32_t idx = (!it.agg && it.c <= 32) ? ((uint32_t)lane >> 1) : (uint32_t)lane;
  it.rec0 = rec_at(it.src, min(idx, it.c - 1));
}
static_assert(AGG_SCRATCH / 16 <= 128, "the table copy gives a lane two 16-byte pieces");

/* The first item of a segment: wait for its loads before the item loop is entered.  Inside the loop an item's loads were
 * issued an item ago and are waited for where the item is handed over (cur = nxt); without this, the loop body would have
 * to allow for the first item's loads still being in flight on the way in, and its wait for them -- counted back from the
 * newest load -- would in every later round be a wait for the prefetch of the NEXT item, issued moments before. */
__device__ __forceinline__ void vote_item_settle(VoteItem& it) {
  uint32_t lo = (uint32_t)__double2loint(it.a64), hi = (uint32_t)__double2hiint(it.a64);
  asm volatile("" : "+v"(lo), "+v"(hi), "+v"(it.rec0.x), "+v"(it.rec0.y), "+v"(it.rec0.z), "+v"(it.rec0.w));
  it.a64 = __hiloint2double((int)hi, (int)lo);
}

__device__ __forceinline__ void vote_locate(VoteItem& it, const uint32_t item, int& h, const uint32_t* seg, const int lane, const MatchArgs& a,
                                            const uint4* __restrict__ records) {
  while (true) { /* advance h to the last position with prefix <= item (position run_seg: the sentinel) */
    const uint32_t pv = seg[min(h + 1 + lane, a.run_seg) * SEG_WORDS + SEG_PREFIX];
    const unsigned long long le = __ballot(pv <= item);
    const int adv = __popcll(le);
    h += adv;
    if (adv < 64) break;
  }
  /* the staged values are the same in every lane: move them to scalar registers so the item
   * runs on scalar control flow and scalar base addresses */
  const uint32_t* run = seg + h * SEG_WORDS; /* the staged run: six consecutive words behind one wave-uniform address */
  const uint32_t local = item - (uint32_t)__builtin_amdgcn_readfirstlane((int)run[SEG_PREFIX]);
  const uint32_t c_all = (uint32_t)__builtin_amdgcn_readfirstlane((int)run[SEG_CNT]);
  const uint32_t mm = (uint32_t)__builtin_amdgcn_readfirstlane((int)run[SEG_M]);
  const uint32_t m_all = mm & 0x7FFFFFFFu;
  const uint32_t hit0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)run[SEG_HIT]);
  const uint32_t off0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)run[SEG_OFF]);
  it.agg = (mm & 0x80000000u) != 0;
  const uint32_t chunk_sz = it.agg ? (uint32_t)AGG_CHUNK : (uint32_t)VOTE_CHUNK;
  const uint32_t group_sz = it.agg ? (uint32_t)AGG_SUB : (uint32_t)VOTE_MAX_HITS;
  const uint32_t nchunk = (c_all + chunk_sz - 1) / chunk_sz;
  const uint32_t sub = local / nchunk, chunk = local - sub * nchunk;
  it.c = min(chunk_sz, c_all - chunk * chunk_sz);
  it.src = records + off0 + chunk * chunk_sz;
  it.g0 = hit0 + sub * group_sz;
  it.nh = (int)min(group_sz, m_all - sub * group_sz);
  it.tbl = it.agg ? a.tables + (size_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)run[SEG_TBL]) + sub) * TBL_BYTES : nullptr;
#if PPF_PREFETCH >= 1
  vote_fetch_hits(it, lane, a);
#endif
#if PPF_PREFETCH >= 2
  vote_fetch_records(it, lane);
#endif
}

/* WRAP: PCL's alpha binning (2 pi range with wrap-around, policy switch alpha_range_2pi): direct votes only */
template <bool WRAP, bool ACC32>
__global__ __launch_bounds__(VOTE_BLOCK) void k_vote(MatchArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t* red = reinterpret_cast<uint32_t*>(smem);                               /* LDS_HEADER */
  const int RS = a.run_seg;                                                        /* runs staged per segment (RUN_SEG .. RUN_SEG_MAX) */
  uint32_t* seg = reinterpret_cast<uint32_t*>(smem + LDS_HEADER);                  /* RS staged runs of SEG_WORDS words + the sentinel's prefix */
  unsigned char* wave_scratch = smem + LDS_HEADER + vote_seg_bytes(RS);            /* VOTE_WAVES x AGG_SCRATCH */
  const int A = a.num_angles;
  const int P = vote_pitch(A);
  const int GW = vote_guard(A);
  uint32_t* lds_acc = reinterpret_cast<uint32_t*>(wave_scratch + VOTE_WAVES * AGG_SCRATCH); /* guard + cells */
  uint32_t* acc = lds_acc + GW;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); /* scalar: the work-item loop is wave-uniform */
  /* tile-major launch order: the workgroups in flight work on the same accumulator tile, i.e. the same slice of the
   * model table (a 10k-point model: 80 MB of 800 MB), which then stays in L2 / Infinity Cache while the reference
   * points go by (C4: 483 -> 471 ms; C2, whose table fits the cache anyway: no change) */
  /* 32-bit cells: one unit of work per (reference point, tile, half of the tile's rows), the two halves of a tile one after
   * the other so that the workgroups in flight still share one slice of the table.  After a 16-bit launch (a.acc32 == 2)
   * the units are the flagged (reference point, tile)s of ovf_list, shared out over a small fixed grid: a launch with
   * nothing to repeat costs a few microseconds. */
  const bool listed = ACC32 && a.acc32 == 2;
  const uint32_t n_work = listed ? 2u * a.cursors[CUR_OVFCOUNT] : gridDim.x;
  uint32_t wid = blockIdx.x;
  if (ACC32 && wid >= n_work) return;
  do { /* one unit per workgroup except for the listed units of the 32-bit launch */
  int vt, tile, r;
  if (listed) {
    const uint32_t e = a.ovf_list[wid >> 1];
    r = (int)(e & 0xFFFFu);
    tile = (int)(e >> 16);
    vt = 2 * tile + (int)(wid & 1u);
  } else {
    vt = (int)(wid / (uint32_t)a.n_ref);
    tile = ACC32 ? vt >> 1 : vt;
    r = (int)a.perm[wid - (uint32_t)vt * (uint32_t)a.n_ref]; /* heaviest reference points first */
  }
  const int rg = a.ref_base + r;
  const int tile_base = tile * a.tile_refs;
  const int refs_here = min(a.tile_refs, a.n_model - tile_base);
  const int H = (a.tile_refs + 1) >> 1;            /* rows per half: row r < H owns the low halves, row r + H the high halves */
  const int words = GW + min(H, refs_here) * P + 1; /* + the word that takes bin A of each half's last row */
  constexpr bool acc32 = ACC32; /* 32-bit cells, two passes per tile: the rare repeat after a 16-bit cell overflowed */
  if (tid == 0) { red[50] = red[51] = red[52] = red[53] = red[58] = red[59] = 0u; } /* votes issued / found / counted (64-bit each), see the overflow check */

  const uint32_t* __restrict__ boff = a.bucket_off + (size_t)tile * (a.n_buckets + 1);
  if (!ACC32 && a.item_votes) {
    /* How many votes will this (reference point, tile) cast?  hits x entries of its runs, known before a single vote: one pass
     * over the run table.  The host learns from it (k_finalize files every item's count under "needed 32-bit cells" or "did
     * not"); and an item at or above the learned limit is not tried with 16-bit cells at all -- it joins the list of the
     * 32-bit launch right away (an overflow that is found by voting costs the item twice). */
    unsigned long long v = 0;
    for (int blk = 0; blk < a.n_rounds; blk++) {
      const uint2 rb = a.run_blocks[(size_t)r * a.n_rounds + blk];
      for (uint32_t i = (uint32_t)tid; i < rb.y; i += VOTE_BLOCK) {
        const uint4 run = a.runs[rb.x + i];
        v += (unsigned long long)run.z * (unsigned long long)(boff[run.x + 1] - boff[run.x]);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    unsigned long long* red64 = reinterpret_cast<unsigned long long*>(red + 16); /* 16 x 8 bytes of the header, free until the staging */
    __syncthreads();
    if (lane == 0) red64[wave] = v;
    __syncthreads();
    unsigned long long V = 0;
#pragma unroll
    for (int k = 0; k < VOTE_WAVES; k++) V += red64[k];
    V *= 2ull; /* two entries per pair record */
    const size_t item = (size_t)rg * a.n_tiles + tile;
    if (tid == 0) a.item_votes[item] = V;
    if (V >= a.heavy_votes) {
      if (tid == 0) {
        a.ovf_items[item] = 2u;
        a.ovf_list[atomicAdd(&a.cursors[CUR_OVFCOUNT], 1u)] = (uint32_t)r | ((uint32_t)tile << 16);
      }
      return;
    }
    __syncthreads(); /* red64 is staging scratch again */
  }
  const uint4* __restrict__ records = a.records;
  const double s64 = WRAP ? (double)A / (2 * PPF_PI) : (double)A / (4 * PPF_PI);
  const float S = (float)s64;
  /* guard band: the fp32 error bound is 1.1e-7*A; with the 2 pi range q' reaches 2.5 A and its
   * rounding steps are four times coarser (bound 3.7e-7*A).  PPF_FORCE_EXACT (test build): 1, every direct vote takes the fp64 chain */
  const float G = WRAP ? fmaxf(1.6e-6f * (float)A, vote_guard_band(A) >= 1.0f ? 1.0f : 0.0f) : vote_guard_band(A);
  const float G2 = 2.0f * G;
  const double og64 = (WRAP ? 1.5 : 0.5) * (double)A + (double)G; /* folded offsets are formed in fp64 and rounded once */
  const uint32_t tail_bytes = (uint32_t)(lane * 4 + 8); /* per-lane guard word for lanes past the end of a bucket (words 2..65: the count
                                                          table path also touches the word below the row's bin 0) */
  AggConsts ak;
  ak.acc_base = (uint32_t)(uintptr_t)(lds_byte*)reinterpret_cast<unsigned char*>(lds_acc);
  ak.ws = (uint32_t)(uintptr_t)(lds_byte*)(wave_scratch + wave * AGG_SCRATCH);
  ak.S = S; ak.half_a = 0.5f * (float)A; ak.Og = (float)og64; ak.G2 = G2;
  ak.s64 = s64; ak.half_a64 = 0.5 * (double)A;
  ak.A = A;
  const uint32_t acc_base = ak.acc_base;
  const uint32_t agg_min = (!WRAP && a.agg_min_hits > 0 && A <= AGG_MAX_ANGLES) ? (uint32_t)a.agg_min_hits : 0xFFFFFFFFu;
  unsigned long long ops = 0; /* LDS atomic lane-operations issued by this wave (wave-uniform part) */
  uint32_t agg_votes = 0;     /* ... plus this lane's one-by-one votes on the count-table path */
  unsigned long long issued = 0; /* votes cast by this wave, wave-uniform: every lane slot of an item casts one vote per hit, into a cell or a guard word */

  const int pass = ACC32 ? (vt & 1) : 0;
  PPF_PHASE_DECL;
  { /* clear guard + cells with 16-byte LDS stores (the region starts 16-byte aligned) */
    uint4* z = reinterpret_cast<uint4*>(lds_acc);
    for (int c = tid; c < words / 4; c += VOTE_BLOCK) z[c] = make_uint4(0u, 0u, 0u, 0u);
    for (int c = (words & ~3) + tid; c < words; c += VOTE_BLOCK) lds_acc[c] = 0u;
  }
  VoteInc vi;
  vi.lo = acc32 ? (pass == 0 ? 1u : 0u) : 1u;
  vi.hi = acc32 ? (pass == 1 ? 1u : 0u) : 0x10000u;
  ak.vi = vi;
  ak.sel_lo = (acc32 && pass == 1) ? 0x0c0c0c0cu : 0x0c0c0c00u; ak.step_lo = (acc32 && pass == 1) ? 0u : 1u;
  ak.sel_hi = acc32 ? (pass == 1 ? 0x0c0c0c00u : 0x0c0c0c0cu) : 0x0c000c0cu;
  ak.step_hi = acc32 ? (pass == 1 ? 1u : 0u) : 0x10000u;

  for (int blk = 0; blk < a.n_rounds; blk++) {
    const uint2 rb = a.run_blocks[(size_t)r * a.n_rounds + blk];
    for (uint32_t seg0 = 0; seg0 < rb.y; seg0 += (uint32_t)RS) {
      /* Stage a segment of the run table: this tile's record range of every run and its work items, exclusive scan */
      __syncthreads(); /* previous segment fully consumed (and the accumulator clear, first time) */
      PPF_PHASE(seg0 ? 5 : 0);
      const uint32_t n_seg = min((uint32_t)RS, rb.y - seg0);
      uint32_t items = 0;
      bool heavy_run = false; /* a run k_group filed under "many hits": those come first in every round's run list */
      if (tid < RS) {
        uint32_t off = 0, cnt = 0, hs = 0, mm = 0, tb0 = 0;
        if ((uint32_t)tid < n_seg) {
          const uint4 run = a.runs[rb.x + seg0 + tid];
          off = boff[run.x];
          cnt = boff[run.x + 1] - off;
          if (acc32) { /* this pass counts one half of the tile's rows: only the records that can hold them */
            const uint32_t n0 = a.bucket_mid[(size_t)tile * a.n_buckets + run.x], blk = 32u * (n0 >> 6), d = n0 & 63u;
            if (pass == 0) cnt = min(cnt, blk + min(d, 32u));
            else { const uint32_t skip = min(cnt, blk + (d > 32u ? d - 32u : 0u)); off += skip; cnt -= skip; }
          }
          hs = run.y;
          mm = run.z;
          tb0 = run.w;
          heavy_run = run.z >= agg_min;
          (void)heavy_run; /* only the two-queue variant (PPF_TWO_QUEUES) counts them */
          if (cnt) {
            if (run.z >= agg_min && cnt >= PPF_AGG_MIN_RECORDS) {
              items = ((run.z + AGG_SUB - 1) / AGG_SUB) * ((cnt + AGG_CHUNK - 1) / AGG_CHUNK);
              mm |= 0x80000000u;
            } else {
              items = ((run.z + VOTE_MAX_HITS - 1) / VOTE_MAX_HITS) * ((cnt + VOTE_CHUNK - 1) / VOTE_CHUNK);
            }
          }
        }
        uint32_t* run_out = seg + tid * SEG_WORDS;
        run_out[SEG_OFF] = off; run_out[SEG_CNT] = cnt; run_out[SEG_HIT] = hs; run_out[SEG_M] = mm; run_out[SEG_TBL] = tb0;
      }
      uint32_t incl = items;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
      }
      if (lane == 63) red[wave] = incl;
#if PPF_TWO_QUEUES
      { const uint32_t nh = (uint32_t)__popcll(__ballot(heavy_run)); if (lane == 0) red[16 + wave] = nh; }
#endif
      __syncthreads();
      uint32_t woff = 0, total = 0;
#pragma unroll
      for (int kk = 0; kk < RUN_SEG_MAX / 64; kk++) { /* waves that staged nothing wrote a zero */
        const uint32_t w = red[kk];
        if (kk < wave) woff += w;
        total += w;
      }
      if (tid < RS) seg[tid * SEG_WORDS + SEG_PREFIX] = woff + incl - items; /* exclusive */
      if (tid == 0) seg[RS * SEG_WORDS + SEG_PREFIX] = total;                 /* the sentinel the 64-wide look-ahead is clamped to */
#if PPF_TWO_QUEUES
      /* Two queues: the items of the many-hit runs (count tables: LDS-bound) and those of the few-hit runs behind them
       * (direct votes over whole buckets: bound by the latency of the record loads).  `split` = first item of the second. */
      {
        uint32_t n_heavy = 0;
#pragma unroll
        for (int kk = 0; kk < RUN_SEG_MAX / 64; kk++) n_heavy += red[16 + kk];
        if (tid == (int)n_heavy && n_heavy < (uint32_t)RS) red[56] = woff + incl - items;
        if (tid == 0) { if (n_heavy >= (uint32_t)RS) red[56] = total; red[48] = 0u; }
      }
      __syncthreads();
      if (tid == 0) red[49] = red[56];
#else
      if (tid == 0) red[48] = VOTE_WAVES; /* next unclaimed work item (each wave starts with item == its id) */
#endif
      __syncthreads();
      PPF_PHASE(0);

      /* Work items are claimed from an LDS counter as waves become free (items differ by orders of magnitude in
       * size); a wave's items still come in increasing order, so the owning run is found with a 64-wide look-ahead
       * from the previous one (runs with no records in this tile have 0 items and are skipped).  A wave always holds
       * TWO items: while it votes one, the first loads of the next (its hits' alpha_s, its first records) are in
       * flight -- most items are small (a bucket's share of one tile: median 14 records) and would otherwise spend
       * their time waiting for those loads. */
      VoteItem cur, nxt;
#if PPF_TWO_QUEUES
      /* Half of the waves (two per SIMD) take from the first queue while it lasts, the other half from the second, so a CU
       * runs its LDS-bound and its load-bound work side by side instead of one after the other; a wave whose queue is
       * empty helps with the other one.  Inside a queue the order is k_group's: heaviest first.  A wave's items of ONE
       * queue still come in increasing order: one look-ahead cursor per queue. */
      const uint32_t split = red[56];
      int hq[2] = {0, 0};
      const int pref = (wave >> 2) & 1; /* waves w, w + 4, w + 8, w + 12 share a SIMD: every SIMD gets two of each kind */
      auto claim = [&](uint32_t& it_out, int& q_out) -> bool {
        int q = pref;
#pragma unroll
        for (int attempt = 0; attempt < 2; attempt++, q ^= 1) {
          const uint32_t lim = q == 0 ? split : total;
          /* an empty queue is left alone once its counter has passed the end (the counter may overshoot by one per wave) */
          const uint32_t it = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lane == 0 ? atomicAdd(&red[48 + q], 1u) : 0u));
          if (it < lim) { it_out = it; q_out = q; return true; }
        }
        return false;
      };
      uint32_t item = 0;
      int q_item = 0;
      bool have = claim(item, q_item);
      if (have) vote_locate(cur, item, hq[q_item], seg, lane, a, records);
      vote_item_settle(cur);
      while (have) {
        uint4 tbl0 = make_uint4(0u, 0u, 0u, 0u), tbl1 = tbl0;
        if (cur.agg) {
          const uint4* __restrict__ t = reinterpret_cast<const uint4*>(cur.tbl);
          tbl0 = rec_at(t, (uint32_t)min(lane, AGG_SCRATCH / 16 - 1));
          tbl1 = rec_at(t, (uint32_t)min(lane + 64, AGG_SCRATCH / 16 - 1));
        }
        const bool have_next = claim(item, q_item);
        if (have_next) vote_locate(nxt, item, hq[q_item], seg, lane, a, records);
#else
      int h = 0;
      uint32_t item = (uint32_t)wave;
      bool have = item < total;
      if (have) vote_locate(cur, item, h, seg, lane, a, records);
      vote_item_settle(cur);
      while (have) {
        /* a count-table item: its table's rows and cell ranges (two 16-byte pieces per lane) set out first, ahead of the next
         * item's prefetch, so that the wait for them is not also a wait for that */
        uint4 tbl0 = make_uint4(0u, 0u, 0u, 0u), tbl1 = tbl0;
        if (cur.agg) {
          const uint4* __restrict__ t = reinterpret_cast<const uint4*>(cur.tbl);
          tbl0 = rec_at(t, (uint32_t)min(lane, AGG_SCRATCH / 16 - 1));
          tbl1 = rec_at(t, (uint32_t)min(lane + 64, AGG_SCRATCH / 16 - 1));
        }
        item = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lane == 0 ? atomicAdd(&red[48], 1u) : 0u));
        const bool have_next = item < total;
        if (have_next) vote_locate(nxt, item, h, seg, lane, a, records);
#endif
#if PPF_PREFETCH < 1
        vote_fetch_hits(cur, lane, a);
#endif
#if PPF_PREFETCH < 2
        vote_fetch_records(cur, lane);
#endif
        PPF_PHASE(1);
        if (cur.agg) {
          /* ---- count-table item: <= AGG_SUB hits x <= AGG_CHUNK records ---- */
          const uint32_t c = cur.c;
          const uint4* __restrict__ src = cur.src;
          uint4 rec_cur = cur.rec0;
          { /* rows + cell ranges of the item's table (k_tables) into this wave's LDS.  Loaded at the top of the round, not with
             * the item's prefetch: a reference point has some 60 count-table items among 600, and eight registers of table in
             * every prefetched item cost the direct items more than this wait costs these */
            lds_u32x4* w = (lds_u32x4*)(uintptr_t)ak.ws;
            wave_lds_fence(); /* the previous item's reads of this scratch are done */
            if (lane < AGG_SCRATCH / 16) w[lane] = u32x4_t{tbl0.x, tbl0.y, tbl0.z, tbl0.w};
            if (lane + 64 < AGG_SCRATCH / 16) w[lane + 64] = u32x4_t{tbl1.x, tbl1.y, tbl1.z, tbl1.w};
            wave_lds_fence();
          }
          const unsigned char* __restrict__ tbl = cur.tbl;
          const double* __restrict__ g_a64 = a.s_a64 + cur.g0;
          /* two blocks of records in flight: block i+2's records and block i+1's own-cell offsets (which need that block's
           * records and the cell ranges in LDS) are loaded under block i's votes, so a block waits for nothing issued in
           * its own iteration */
          AggOwn own_cur;
          agg_own_fetch(ak, rec_cur, tbl, cur.nh, own_cur);
          uint4 rec_n1 = rec_at(src, min((uint32_t)lane + 64u, c - 1));
          for (uint32_t e0 = 0; e0 < c; e0 += 64) {
            const uint32_t e = e0 + (uint32_t)lane;
            const uint4 rec_n2 = rec_at(src, min(e + 128, c - 1));
            AggOwn own_n1;
            agg_own_fetch(ak, rec_n1, tbl, cur.nh, own_n1);
            uint4 rec = rec_cur;
            if (e >= c) { rec.x = tail_bytes | (rec.x & ~ROW_CODE_MASK); rec.y = tail_bytes | (rec.y & ~ROW_CODE_MASK); } /* the clamped record's cells go with its alphas */
            agg_pair(ak, rec, own_cur, tbl, g_a64, agg_votes);
            rec_cur = rec_n1;
            rec_n1 = rec_n2;
            own_cur = own_n1;
          }
          ops += 2u * (AGG_NY + 1) * 64u * ((c + 63u) / 64u);
          issued += 128ull * ((c + 63u) / 64u) * (uint32_t)cur.nh; /* counted + one-by-one votes of an entry = its hits */
          PPF_PHASE(2);
        } else {
          /* ---- direct item: <= VOTE_MAX_HITS hits x <= VOTE_CHUNK records ---- */
          const uint32_t c = cur.c;
          const int nh = cur.nh;
          /* the prefetched first records as plain register values: a `cond ? cur.rec0 : src[..]` further down would become a load
           * through a selected POINTER, which needs cur.rec0 in memory: the whole item then lives on the stack and its prefetch is
           * waited for and stored the moment it is issued */
          uint4 rec0_reg = cur.rec0;
          asm volatile("" : "+v"(rec0_reg.x), "+v"(rec0_reg.y), "+v"(rec0_reg.z), "+v"(rec0_reg.w));
          /* lane l holds the folded offset of hit g0+l: Ohg = A/2 + G - alpha_s*S */
          const float ohg_v = lane < nh ? (WRAP ? (float)(og64 - cur.a64 * s64) : (float)og64 - (float)cur.a64 * S) : 0.f;
          const double* __restrict__ asd = a.s_a64 + cur.g0; /* exact alpha_s, read on the guard path only */
          const uint4* __restrict__ src = cur.src;
          constexpr uint32_t B = 64 * VOTE_UNROLL; /* records per batch */
          const uint32_t nfull = c / B;
          ops += 2ull * 64u * ((c + 63u) / 64u) * (uint32_t)nh;
          issued += 128ull * VOTE_UNROLL * nfull * (uint32_t)nh;
          /* Three self-contained cases, each with its own record registers: nothing loaded for one item is live (or still in
           * flight) when the next one starts, so that the start of an item waits for its own prefetched record only */
#if PPF_ABL_DIRECT_BIG != 1
          if (c > 32) { /* attribution build: these items without their votes (2: the record loads stay) */
#if PPF_ABL_DIRECT_BIG == 2
            uint4 eb[VOTE_UNROLL];
            for (uint32_t e0 = 0; e0 < c; e0 += 64 * VOTE_UNROLL) {
              load_records<VOTE_UNROLL>(eb, src, min(e0, c > 64u * VOTE_UNROLL ? c - 64u * VOTE_UNROLL : 0u), lane);
#pragma unroll
              for (int u = 0; u < VOTE_UNROLL; u++) asm volatile("" ::"v"(eb[u].x), "v"(eb[u].y), "v"(eb[u].z), "v"(eb[u].w));
            }
#endif
            cur = nxt;
            have = have_next;
            continue;
          }
#endif
          if (c <= 32) { /* at most 64 entries: one entry per lane (rec0 was fetched for this layout) */
            const uint32_t e = (uint32_t)lane >> 1;
            const uint4 rr = rec0_reg;
            const uint32_t odd = 0u - ((uint32_t)lane & 1u); /* the record's second entry */
            const uint32_t row_bytes = e < c ? bit_select(odd, rr.x, rr.y) : tail_bytes;
            if (PPF_ABL_DIRECT_SMALL) vote_hits_single<WRAP>(acc_base, vi, row_bytes, bit_select(odd, rr.z, rr.w), S, ohg_v, nh, asd, G2, A);
            issued += 64ull * (uint32_t)nh;
          } else {
            uint32_t e0 = 0;
            if (nfull) {
              /* full batches, software-pipelined over two register sets: the loads of batch b+1 are in flight while batch b is
               * voted for every hit of the item.  The prefetch is unconditional (the last one re-reads the final batch): a
               * conditional one would merge two control-flow paths and force the compiler into a vmcnt that also waits for
               * the prefetch. */
              uint4 ea[VOTE_UNROLL], eb[VOTE_UNROLL];
              ea[0] = rec0_reg;
#pragma unroll
              for (int u = 1; u < VOTE_UNROLL; u++) ea[u] = rec_at(src, (uint32_t)(u * 64 + lane));
              uint32_t b = 0;
              while (true) {
                load_records<VOTE_UNROLL>(eb, src, min(b + 1, nfull - 1) * B, lane);
                vote_hits<VOTE_UNROLL, WRAP>(acc_base, vi, ea, VOTE_UNROLL, S, ohg_v, nh, asd, G2, A);
                if (++b >= nfull) {
#pragma unroll
                  for (int u = 0; u < VOTE_UNROLL; u++) asm volatile("" ::"v"(eb[u].x), "v"(eb[u].y), "v"(eb[u].z), "v"(eb[u].w)); /* the re-read lands HERE, not under the next item */
                  break;
                }
                load_records<VOTE_UNROLL>(ea, src, min(b + 1, nfull - 1) * B, lane);
                vote_hits<VOTE_UNROLL, WRAP>(acc_base, vi, eb, VOTE_UNROLL, S, ohg_v, nh, asd, G2, A);
                if (++b >= nfull) {
#pragma unroll
                  for (int u = 0; u < VOTE_UNROLL; u++) asm volatile("" ::"v"(ea[u].x), "v"(ea[u].y), "v"(ea[u].z), "v"(ea[u].w));
                  break;
                }
              }
              e0 = nfull * B;
            }
            if (e0 < c && c - e0 <= 32) { /* at most 64 entries left: one entry per lane */
              const uint32_t e = e0 + ((uint32_t)lane >> 1);
              const uint4 rr = rec_at(src, min(e, c - 1));
              const uint32_t odd = 0u - ((uint32_t)lane & 1u); /* the record's second entry */
              const uint32_t row_bytes = e < c ? bit_select(odd, rr.x, rr.y) : tail_bytes;
              vote_hits_single<WRAP>(acc_base, vi, row_bytes, bit_select(odd, rr.z, rr.w), S, ohg_v, nh, asd, G2, A);
              issued += 64ull * (uint32_t)nh;
            } else if (e0 < c) { /* tail: clamped addresses; lanes past the end vote into their guard word */
              uint4 et[VOTE_UNROLL];
#pragma unroll
              for (int u = 0; u < VOTE_UNROLL; u++) {
                const uint32_t e = e0 + u * 64 + lane;
                if (u == 0 && e0 == 0) et[u] = rec0_reg; else et[u] = rec_at(src, min(e, c - 1));
                if (e >= c) { et[u].x = tail_bytes; et[u].y = tail_bytes; }
              }
              const int n_valid = (int)((c - e0 + 63) / 64);
              issued += 128ull * (uint32_t)n_valid * (uint32_t)nh;
              /* most buckets are smaller than a batch: only the 64-record groups that hold data get
               * their bin arithmetic, through an instantiation per group count */
              static_assert(VOTE_UNROLL == 4, "tail dispatch below assumes 4 groups per batch");
              switch (n_valid) {
                case 1: vote_hits<1, WRAP>(acc_base, vi, et, 1, S, ohg_v, nh, asd, G2, A); break;
                case 2: vote_hits<2, WRAP>(acc_base, vi, et, 2, S, ohg_v, nh, asd, G2, A); break;
                case 3: vote_hits<3, WRAP>(acc_base, vi, et, 3, S, ohg_v, nh, asd, G2, A); break;
                default: vote_hits<4, WRAP>(acc_base, vi, et, 4, S, ohg_v, nh, asd, G2, A); break;
              }
            }
          }
          PPF_PHASE(c <= 32 ? 4 : 3);
        }
        cur = nxt;
        have = have_next;
      }
    }
  }
  __syncthreads();
  PPF_PHASE(5);

  /* Scan in the reference's order (model ref ascending, alpha bin ascending, strict >) == smallest
   * upstream flat index ref*A + bin among the maxima.  Also the exact vote total of the tile.  One thread per accumulator
   * row, no integer division; bins ascending with strict > keeps the row's first maximum.  32-bit cells: this workgroup
   * holds the rows of half `pass` only. */
  uint32_t* dump = a.acc_dump ? a.acc_dump + (size_t)rg * a.n_model * A + (size_t)tile_base * A : nullptr;
  uint32_t bv = 0, bi = 0xFFFFFFFFu;
  unsigned long long sum = 0, found = 0;
  const int row_lo = acc32 ? pass * H : 0, row_hi = acc32 ? min((pass + 1) * H, refs_here) : refs_here;
  for (int ref = row_lo + tid; ref < row_hi; ref += VOTE_BLOCK) {
    const int hf = ref >= H ? 1 : 0;
    const uint32_t* row = acc + (ref - hf * H) * P;
    const int sh = acc32 ? 0 : hf * 16;
    const uint32_t msk = acc32 ? 0xFFFFFFFFu : 0xFFFFu;
    for (int bin = 0; bin < A; bin++) {
      uint32_t v = (row[bin] >> sh) & msk;
      /* a row's bin A (the reference's spill into the next row) is the next row's bin 0 in memory; only the last low-half row
       * has its successor elsewhere: its bin A is the low half of the word behind the rows (32-bit cells: the other
       * half's workgroup holds it, k_finalize adds it) */
      if (bin == 0 && ref == H && !acc32) v += acc[H * P] & 0xFFFFu;
      if (dump) { if (acc32 && ref == H && bin == 0) atomicAdd(&dump[ref * A + bin], v); else dump[ref * A + bin] = v; }
      sum += v;
      if (v > bv) { bv = v; bi = (uint32_t)(ref * A + bin); }
    }
  }
  if (!acc32) /* everything the votes left in LDS: guard words are whole counters, cell words two 16-bit counters */
    for (int c = tid; c < words; c += VOTE_BLOCK) {
      const uint32_t w = lds_acc[c];
      found += c < GW ? (unsigned long long)w : (unsigned long long)((w & 0xFFFFu) + (w >> 16));
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t v2 = __shfl_down(bv, o), i2 = __shfl_down(bi, o);
    sum += __shfl_down(sum, o);
    found += __shfl_down(found, o);
    if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
  }
  uint32_t* red_v = seg; /* the staging area is free now */
  uint32_t* red_i = seg + VOTE_WAVES;
  __syncthreads();
  if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
  if (lane == 0) {
    if (!acc32) {
      atomicAdd(reinterpret_cast<unsigned long long*>(&red[50]), issued);
      atomicAdd(reinterpret_cast<unsigned long long*>(&red[52]), found);
    }
    if (sum) atomicAdd(reinterpret_cast<unsigned long long*>(&red[58]), sum);
  }
  __syncthreads();
  if (wave == 0) {
    uint32_t v = (lane < VOTE_WAVES) ? red_v[lane] : 0u;
    uint32_t ix = (lane < VOTE_WAVES) ? red_i[lane] : 0xFFFFFFFFu;
#pragma unroll
    for (int o = VOTE_WAVES / 2; o > 0; o >>= 1) {
      const uint32_t v2 = __shfl_down(v, o), i2 = __shfl_down(ix, o);
      if (v2 > v || (v2 == v && i2 < ix)) { v = v2; ix = i2; }
    }
    if (lane == 0) {
      const size_t item = (size_t)rg * a.n_tiles + tile;
      unsigned long long total = *reinterpret_cast<unsigned long long*>(&red[58]);
      /* 16-bit cells: a cell that wrapped or carried into its neighbour makes the votes found differ from the votes issued;
       * the (reference point, tile) is then voted again with 32-bit cells, and nothing of this attempt counts */
      const bool overflow = !PPF_ABL_ANY && !acc32 && (red[50] != red[52] || red[51] != red[53]);
      if (overflow) {
        a.ovf_items[item] = 1u;
        a.ovf_list[atomicAdd(&a.cursors[CUR_OVFCOUNT], 1u)] = (uint32_t)r | ((uint32_t)tile << 16);
        atomicAdd(&a.tally[4], *reinterpret_cast<unsigned long long*>(&red[50])); /* work that is done twice */
        if (dump && refs_here > H) dump[H * A] = 0u; /* the 32-bit launch adds its two parts of this cell */
      } else {
        if (acc32) { /* what the two halves owe each other across the row H-1 / row H boundary */
          const uint32_t e = refs_here > H ? (pass == 0 ? acc[H * P] : acc[0]) : 0u;
          a.edge[item * 2 + pass] = e;
          if (pass == 0) { total += e; if (dump && e) atomicAdd(&dump[H * A], e); } /* the scan of the high halves leaves this spill out */
          if (pass == 0) atomicAdd(&a.tally[3], 1ull);
        }
        a.partial[item * 2 + pass] = make_uint2(v, ix);
        if (total) atomicAdd(&a.cellsum[item], total);
      }
    }
  }

  unsigned long long wops = (unsigned long long)agg_votes; /* summed over the lanes below; the uniform part is added once */
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wops += __shfl_down(wops, o);
  wops += ops;
  if (lane == 0 && wops) atomicAdd(&a.tally[0], wops);
  PPF_PHASE(6);
  PPF_PHASE_FLUSH(a.tally, lane);
  if (!ACC32) break;
  wid += gridDim.x;
  if (wid < n_work) __syncthreads(); /* the next unit clears the accumulator this one's last readers are done with */
  } while (wid < n_work);
}

/* fixed LDS of k_vote: header + run staging + per-wave count tables (the guard and the cells are sized per model) */
constexpr size_t vote_lds_fixed(int run_seg) { return LDS_HEADER + vote_seg_bytes(run_seg) + (size_t)VOTE_WAVES * AGG_SCRATCH; }
constexpr size_t VOTE_LDS_FIXED = vote_lds_fixed(RUN_SEG); /* with the least staging: what a model's tile size is chosen against */
/* runs a call stages per segment: what the LDS holds next to the accumulator of `acc_words` words, RUN_SEG at least */
inline int vote_run_seg(size_t acc_words, size_t lds_bytes) {
  const size_t least = VOTE_LDS_FIXED + acc_words * 4;
  if (least >= lds_bytes) return RUN_SEG;
  const size_t more = (lds_bytes - least) / (64 * SEG_WORDS * 4); /* 64 more staged runs */
  return (int)std::min<size_t>((size_t)RUN_SEG_MAX, (size_t)RUN_SEG + 64 * more);
}

#endif /* PPF_MATCH_KERNELS_H */
