/*
 * ppf_match_host.h — C-ABI, matching side: workspaces, ppf_match_device and its result accessors, the host-buffer entries
 * ppf_match / ppf_raw_votes, pose-list clustering.  Reference call sites: /root/reference/include/CloudProcessing.h:442 (match), :495 (match_S2B).
 */
#ifndef PPF_MATCH_HOST_H
#define PPF_MATCH_HOST_H

extern "C" {

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
      if (!(value == 0 || value == 1 || value == 2 || value == 3)) return fail(PPF_ERR_INVALID, "ppf_workspace_set_option: PPF_OPT_ACC32 takes 0 .. 3");
      ws->force_acc32 = value == 1;
      ws->acc32_policy = value == 1 ? 0 : (int)value;
      ws->heavy_votes = ~0ull;
      if (value != 0) ws->acc32 = false;
      return PPF_OK;
    case PPF_OPT_BATCH_REFS:
      if (!(value >= 0 && value <= 1e9)) return fail(PPF_ERR_INVALID, "ppf_workspace_set_option: bad batch size");
      ws->batch_refs_cap = (int)value;
      return PPF_OK;
    case PPF_OPT_RUN_STAGING:
      if (!(value >= 0 && value <= 1e9)) return fail(PPF_ERR_INVALID, "ppf_workspace_set_option: bad run staging size");
      ws->run_seg_cap = (int)value;
      return PPF_OK;
    case PPF_OPT_TABLE_FRACTION:
      if (!(value > 0 && value <= 1)) return fail(PPF_ERR_INVALID, "ppf_workspace_set_option: table fraction must be in (0, 1]");
      ws->tbl_frac = std::min(TBL_FRAC_MAX, value);
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

static ppf_status check_match_args(const ppf_model* m, const void* scene, int ns, int sstride, int snoff, const void* edge, int ne,
                                   int estride, int enoff, const ppf_match_params* p) {
  if (!m) return fail(PPF_ERR_NOT_TRAINED, "The model is not trained. Cannot match without training");
  if (!scene || ns <= 0 || bad_layout(sstride, snoff) || !p) return fail(PPF_ERR_INVALID, "match: bad scene argument");
  if (edge && (ne <= 0 || bad_layout(estride, enoff))) return fail(PPF_ERR_INVALID, "match: bad edge argument");
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
static ppf_status prepare_scene(ppf_workspace* ws, const float* d_scene, int ns, int sstride, int snoff, const float* d_edge, int ne,
                                int estride, int enoff, const ppf_match_params* params, hipStream_t st) {
  auto load = [&](CloudDev& dst, const float* d_src, int rows, int stride, int noff) -> ppf_status {
    if (params->presampled) return dst.load_device(d_src, rows, stride, noff, st);
    return device_sample_cloud(d_src, rows, stride, noff, (float)params->relative_scene_distance, dst, nullptr, st);
  };
  ppf_status s = load(ws->surf, d_scene, ns, sstride, snoff);
  if (s != PPF_OK) return s;
  if (d_edge) s = load(ws->edge, d_edge, ne, estride, enoff);
  ws->has_edge = d_edge != nullptr;
  return s;
}

static ppf_status match_prepared(const ppf_model* m, ppf_workspace* ws, const ppf_match_params* params, hipStream_t st, bool retry = false);

ppf_status ppf_match_device(const ppf_model* m, ppf_workspace* ws, const float* d_scene, int ns, int sstride, int snoff,
                            const float* d_edge, int ne, int estride, int enoff, const ppf_match_params* params, void* stream) {
  if (!ws) return fail(PPF_ERR_INVALID, "ppf_match_device: workspace is NULL");
  if (!m) return fail(PPF_ERR_NOT_TRAINED, "The model is not trained. Cannot match without training");
  if (!have_device()) return fail(PPF_ERR_HIP, "ppf_match_device: no HIP device (this engine has no CPU fallback)");
  ppf_status s = check_match_args(m, d_scene, ns, sstride, snoff, d_edge, ne, estride, enoff, params);
  if (s != PPF_OK) return s;
  hipStream_t st = (hipStream_t)stream;
  s = prepare_scene(ws, d_scene, ns, sstride, snoff, d_edge, ne, estride, enoff, params, st);
  if (s != PPF_OK) return s;
  return match_prepared(m, ws, params, st);
}

/* The fractions a workspace has learned for a model (hits per scene pair, runs per hit), keyed by the model's serial
 * number; create = true makes room for a model it has not met (at most 16 are remembered, the oldest goes). */
static ppf_workspace::Learned* workspace_learned(ppf_workspace* ws, const ppf_model* m, bool create) {
  for (auto& fm : ws->frac_by_model)
    if (fm.model_serial == m->serial) return &fm;
  if (!create) return nullptr;
  if (ws->frac_by_model.size() >= 16) ws->frac_by_model.erase(ws->frac_by_model.begin());
  ws->frac_by_model.push_back({m->serial, 0.25, 0.4, TBL_FRAC_START});
  return &ws->frac_by_model.back();
}

/* store the current model's learned hit fraction */
static void workspace_remember_frac(ppf_workspace* ws) {
  if (!ws->model || !ws->frac_known) return;
  ppf_workspace::Learned* fm = workspace_learned(ws, ws->model, true);
  fm->hit = ws->hit_frac; fm->run = ws->run_frac; fm->tbl = ws->tbl_frac;
}

/* the workspace keeps the model alive until its next call (or its destruction): results are fetched later.  A context
 * workspace (model_owns_me) lives inside its model and takes no reference. */
static void workspace_hold_model(ppf_workspace* ws, const ppf_model* m) {
  if (ws->model == m) return;
  ppf_model* old = ws->model;
  ws->model = const_cast<ppf_model*>(m);
  if (ws->model_owns_me) return;
  if (ws->model) ws->model->refcount.fetch_add(1);
  if (old) (void)ppf_model_release(old);
}

ppf_workspace::~ppf_workspace() {
  sync_device(device); /* buffers return to the block cache: nothing may still be using them */
  for (auto& e : ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : batch_ev)
    if (e) (void)hipEventDestroy(e);
  if (model && !model_owns_me) (void)ppf_model_release(model);
  if (h_sum) (void)hipHostFree(h_sum);
}

/* bytes of hit scratch one hit costs: raw {bucket, j} + sorted payload (alpha_s, cell) + its share of the run table */
constexpr double HIT_SCRATCH_BYTES = 8.0 + 8.0 + 2.0 + 16.0 / 6.0;

/* k_pairs<pair feature, surface-to-boundary>: same_cloud == 0 is match_S2B (the paired points come from the edge cloud) */
static void launch_pairs(const MatchArgs& va, bool darboux, hipStream_t st) {
  const dim3 grid(va.pair_chunks, va.n_ref), block(PAIR_BLOCK);
  const dim3 ogrid((unsigned)std::min(va.n_ref, 1024)), oblock(256); /* k_pairs_odd: returns at once unless k_frames raised its flag */
  if (darboux) {
    if (va.same_cloud) k_pairs<true, false><<<grid, block, 0, st>>>(va);
    else k_pairs<true, true><<<grid, block, 0, st>>>(va);
    if (!va.count_only) { if (va.same_cloud) k_pairs_odd<true, false><<<ogrid, oblock, 0, st>>>(va); else k_pairs_odd<true, true><<<ogrid, oblock, 0, st>>>(va); }
  } else {
    if (va.same_cloud) k_pairs<false, false><<<grid, block, 0, st>>>(va);
    else k_pairs<false, true><<<grid, block, 0, st>>>(va);
    if (!va.count_only) { if (va.same_cloud) k_pairs_odd<false, false><<<ogrid, oblock, 0, st>>>(va); else k_pairs_odd<false, true><<<ogrid, oblock, 0, st>>>(va); }
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
    ws->heavy_votes = ~0ull; /* another table: its own limit, learned from its first call on */
    if (const ppf_workspace::Learned* fm = workspace_learned(ws, m, false)) {
      ws->hit_frac = fm->hit; ws->run_frac = fm->run; ws->tbl_frac = fm->tbl;
      ws->frac_known = true;
    } else if (!ws->frac_by_model.empty()) {
      ws->frac_known = false; /* a model this workspace has not met: count first */
    }
  }
  workspace_hold_model(ws, m);
  HIPCHK(hipGetDevice(&ws->device));
  ws->params = *params;
  ws->stream = st;
  ws->clustered = false;
  ws->checked = false;
  ws->sum_valid = false;
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
  if (ws->acc32_policy == 3) {
    HIPCHK(ws->item_votes.reserve((size_t)n_ref * T));
    HIPCHK(ws->need_hist.reserve(2 * ACC_HIST));
  }
  const size_t n_cnt = (size_t)n_ref * T + n_ref + 16; /* cellsum | pairs | totals[2] | tally[14]: LDS operations, hits, runs, 32-bit items, votes cast twice, count tables, 8 phase clocks (diagnostic build) */
  HIPCHK(ws->counters.reserve(n_cnt));
  HIPCHK(ws->votes.reserve(n_ref));
  HIPCHK(ws->raw_poses.reserve(n_ref));
  if (ws->timing) HIPCHK(hipEventRecord(ws->ev[0], st));
  HIPCHK(hipMemsetAsync(ws->counters.p, 0, n_cnt * sizeof(unsigned long long), st));
  HIPCHK(hipMemsetAsync(ws->ovf_items.p, 0, (size_t)n_ref * T * sizeof(uint32_t), st));
  if (ws->acc_dump) /* debug dump of the full accumulators: a repeat of the call starts from zeros again (one of its cells is accumulated, not assigned) */
    HIPCHK(hipMemsetAsync(ws->acc_dump, 0, (size_t)n_ref * (size_t)m->info.n_ref * (size_t)m->info.num_angles * sizeof(uint32_t), st));
  if (ws->acc32_policy == 3) {
    HIPCHK(hipMemsetAsync(ws->need_hist.p, 0, 2 * ACC_HIST * sizeof(unsigned long long), st));
    HIPCHK(hipMemsetAsync(ws->item_votes.p, 0, (size_t)n_ref * T * sizeof(unsigned long long), st));
  }

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
  /* 16-bit cells first, 32-bit cells for the (reference point, tile)s whose cells overflow -- until a call casts more than
   * PPF_ACC32_SWITCH of its votes twice that way: from then on 32-bit cells for everything (what they cost extra is less than
   * what the repeats cost: profiles/r04_c4_acc32_policies.md).  PPF_OPT_ACC32 = 3 instead sends the items that will cast at
   * least ws->heavy_votes votes straight to the 32-bit launch, decided on the device, per item, for THIS scene, before a vote
   * is cast (measured on C4: slower than either, the vote count of an item says little about its fullest cell). */
  const bool acc32_all = ws->force_acc32 || (ws->acc32 && ws->acc32_policy == 0);
  va.item_votes = (!acc32_all && ws->acc32_policy == 3) ? ws->item_votes.p : nullptr;
  va.heavy_votes = ws->heavy_votes;
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
  const double tbl_frac = !va.agg_min_hits ? 0.0 : frac >= 1.0 ? TBL_FRAC_MAX : std::min(TBL_FRAC_MAX, ws->tbl_frac);
  int batch = (int)std::min<double>((double)n_ref, std::max(1.0, (double)HIT_BYTES_BUDGET / (hits_per_ref * (HIT_SCRATCH_BYTES + tbl_frac * TBL_BYTES))));
  batch = (int)std::min<double>((double)batch, std::max(1.0, 2.0e9 / hits_per_ref));
  batch = std::min(batch, 32768); /* grid.y of k_pairs */
  if (ws->batch_refs_cap > 0) batch = std::min(batch, ws->batch_refs_cap);
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
  /* count tables: a run of c >= agg_min_hits hits takes ceil(c / AGG_SUB), so TBL_FRAC_MAX per hit is the ceiling */
  const uint32_t table_cap = !va.agg_min_hits ? 1u : (uint32_t)std::min(1.0e9, std::max(est * (worst_case ? TBL_FRAC_MAX : tbl_frac), 4.0 * batch) + 256.0);
  HIPCHK(ws->runs.fit(run_cap));
  HIPCHK(ws->tables.fit((size_t)table_cap * TBL_BYTES));
  HIPCHK(ws->table_desc.fit(table_cap));
  HIPCHK(ws->run_blocks.reserve((size_t)batch * va.n_rounds));
  HIPCHK(ws->work.reserve(batch));
  HIPCHK(ws->perm.reserve(batch));
  HIPCHK(ws->perm_group.reserve(batch));
  HIPCHK(ws->ovf_list.reserve((size_t)batch * T));
  va.ovf_list = ws->ovf_list.p;
  ws->stats.scratch_bytes = ws->frames.bytes() + ws->raw.bytes() + ws->cursors.bytes() + ws->chunk_desc.bytes() + ws->hit_count.bytes() +
                            ws->s_a64.bytes() + ws->s_cell.bytes() + ws->runs.bytes() + ws->run_blocks.bytes() + ws->tables.bytes() + ws->table_desc.bytes() +
                            ws->work.bytes() + ws->perm.bytes() + ws->perm_group.bytes();
  va.frames = ws->frames.p;
  va.raw = ws->raw.p; va.stripe_cap = stripe_cap; va.stripe_bits = stripe_bits;
  va.chunk_desc = ws->chunk_desc.p; va.hit_count = ws->hit_count.p;
  va.s_a64 = ws->s_a64.p; va.s_cell = ws->s_cell.p; va.sorted_cap = sorted_cap;
  va.runs = ws->runs.p; va.run_cap = run_cap; va.run_blocks = ws->run_blocks.p;
  va.tables = ws->tables.p; va.table_desc = ws->table_desc.p; va.table_cap = table_cap;
  va.work = ws->work.p; va.perm = ws->perm.p; va.perm_group = ws->perm_group.p;

  /* the run staging gets the LDS this model's accumulator tile leaves (its least size is what the tile was sized against) */
  const size_t acc_words = (size_t)vote_lds_words(m->info.tile_refs, m->info.num_angles);
  va.run_seg = vote_run_seg(acc_words, (size_t)LDS_BYTES);
  if (ws->run_seg_cap > 0) va.run_seg = std::max(64, std::min(va.run_seg, ws->run_seg_cap / 64 * 64)); /* test knob: PPF_OPT_RUN_STAGING */
  const size_t lds = vote_lds_fixed(va.run_seg) + acc_words * 4;
  if (lds > (size_t)LDS_BYTES) return fail(PPF_ERR_INVALID, "match: model tile of %d reference points does not fit the LDS accumulator", m->info.tile_refs);
  /* k_group's dynamic LDS: one counter per bucket of a round, the prefix of the pool pieces */
  const size_t group_lds = (size_t)((va.round_buckets + 1) & ~1) * sizeof(uint32_t) + (size_t)((va.pair_chunks + 2) & ~1) * sizeof(uint32_t) * 2;
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
    while (ws->batch_ev.size() < ws->ev_base + (size_t)n_batches * 4) {
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
    if (va.agg_min_hits) HIPCHK(hipMemsetAsync(ws->table_desc.p, 0, (size_t)table_cap * sizeof(uint2), st));
    /* a thread per reference point, and enough threads for its look at the paired points (40 waves walking 50,000 points: 10 us) */
    k_frames<<<dim3(std::max((va.n_ref + 63) / 64, std::min((va.paired.n + 63) / 64, 1024))), dim3(64), 0, st>>>(va);
    HIPCHK(hipGetLastError());
    if (ws->timing) HIPCHK(hipEventRecord(ws->batch_ev[ws->ev_base + bi * 4 + 0], st));
    launch_pairs(va, darboux, st);
    HIPCHK(hipGetLastError());
    if (ws->timing) HIPCHK(hipEventRecord(ws->batch_ev[ws->ev_base + bi * 4 + 1], st));
    k_ref_hits<<<dim3((va.n_ref + 255) / 256), dim3(256), 0, st>>>(va);
    /* k_group takes the reference points with the most hits first */
    k_rank<<<dim3((va.n_ref + RANK_KEYS - 1) / RANK_KEYS), dim3(256), 0, st>>>(va.hit_count, va.n_ref, nullptr, ws->perm_group.p, nullptr);
    k_group<<<dim3(va.n_ref), dim3(GROUP_BLOCK), group_lds, st>>>(va);
    HIPCHK(hipGetLastError());
    if (va.agg_min_hits) { /* the count tables of the batch's many-hit runs, once per run */
      k_tables<<<dim3((table_cap + TABLE_BLOCK / 64 - 1) / (TABLE_BLOCK / 64)), dim3(TABLE_BLOCK), 0, st>>>(va);
      HIPCHK(hipGetLastError());
    }
    /* k_vote takes the reference points that will cast the most votes first */
    k_rank<<<dim3((va.n_ref + RANK_KEYS - 1) / RANK_KEYS), dim3(256), 0, st>>>(va.work, va.n_ref, nullptr, va.perm, nullptr);
    HIPCHK(hipGetLastError());
    if (ws->timing) HIPCHK(hipEventRecord(ws->batch_ev[ws->ev_base + bi * 4 + 2], st));
    const dim3 grid16((unsigned)((size_t)va.n_ref * T)), grid32((unsigned)((size_t)va.n_ref * T * 2));
    if (!acc32_all) {
      va.acc32 = 0;
      if (params->alpha_range_2pi) k_vote<true, false><<<grid16, dim3(VOTE_BLOCK), lds, st>>>(va);
      else k_vote<false, false><<<grid16, dim3(VOTE_BLOCK), lds, st>>>(va);
      HIPCHK(hipGetLastError());
    }
    va.acc32 = acc32_all ? 1 : 2; /* 2: the (reference point, tile)s the 16-bit launch listed, over a grid that does not depend on their number */
    const dim3 g32 = acc32_all ? grid32 : dim3(std::min(grid32.x, 256u)); /* one workgroup per CU walks the list (a launch of 1,024 with nothing to do: 10 us) */
    if (params->alpha_range_2pi) k_vote<true, true><<<g32, dim3(VOTE_BLOCK), lds, st>>>(va);
    else k_vote<false, true><<<g32, dim3(VOTE_BLOCK), lds, st>>>(va);
    va.acc32 = 0;
    HIPCHK(hipGetLastError());
    if (ws->timing) HIPCHK(hipEventRecord(ws->batch_ev[ws->ev_base + bi * 4 + 3], st));
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
  fa.item_votes = va.item_votes; fa.need_hist = va.item_votes ? ws->need_hist.p : nullptr;
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
  if (!ws->h_sum) HIPCHK(hipHostMalloc((void**)&ws->h_sum, 32 * sizeof(unsigned long long), hipHostMallocDefault));
  k_summary<<<dim3(1), dim3(64), 0, st>>>(fa.totals, ws->cursors.p + CUR_OVERFLOW, ws->clustered ? ws->cl_u32.p : nullptr, ws->h_sum);
  HIPCHK(hipGetLastError());
  ws->sum_valid = true;
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
    unsigned long long tot[16];
    uint32_t ovf = 0;
    if (ws->sum_valid) { /* k_summary wrote them into pinned memory as the call's last act */
      memcpy(tot, ws->h_sum, sizeof(tot));
      ovf = (uint32_t)ws->h_sum[16];
    } else {
      HIPCHK(hipMemcpy(tot, ws->counters.p + (size_t)ws->n_ref * T + ws->n_ref, sizeof(tot), hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(&ovf, ws->cursors.p + CUR_OVERFLOW, sizeof(ovf), hipMemcpyDeviceToHost));
    }
    if (!ovf) {
      ws->stats.n_votes = tot[0];
      ws->stats.n_pairs = tot[1];
      ws->stats.n_lds_atomics = tot[2];
      ws->stats.n_hits = tot[3];
      ws->stats.n_acc32_items = tot[5];
      /* The limit of the next call: k_finalize filed the votes of every (reference point, tile) under its size class, as
       * "needed 32-bit cells" (a cell beyond 65,535) or "did not".  An item sent to 32-bit cells costs about PPF_ACC32_COST of
       * what 16-bit cells cost it; one that overflows 16-bit cells costs 1 + PPF_ACC32_COST.  The class boundary that makes
       * the sum smallest becomes the limit (none above every class: the scene needs no 32-bit cells). */
      if (!ws->acc32 && ws->acc32_policy == 0 && (double)tot[6] > PPF_ACC32_SWITCH * (double)tot[0]) ws->acc32 = true;
      if (!ws->force_acc32 && ws->acc32_policy == 3) {
        unsigned long long hist[2 * ACC_HIST];
        HIPCHK(hipMemcpy(hist, ws->need_hist.p, sizeof(hist), hipMemcpyDeviceToHost));
        double best = -1;
        int best_k = ACC_HIST;
        double above = 0, below = 0; /* cost of the classes >= k (sent to 32-bit cells) / < k (tried with 16-bit cells) */
        for (int b = 0; b < ACC_HIST; b++) below += (double)hist[ACC_HIST + b] + (1.0 + PPF_ACC32_COST) * (double)hist[b];
        for (int k = ACC_HIST; k >= 0; k--) {
          if (k < ACC_HIST) {
            const double need = (double)hist[k], rest = (double)hist[ACC_HIST + k];
            above += PPF_ACC32_COST * (need + rest);
            below -= rest + (1.0 + PPF_ACC32_COST) * need;
          }
          const double cost = above + below;
          if (best < 0 || cost < best) { best = cost; best_k = k; }
        }
        ws->heavy_votes = best_k >= ACC_HIST ? ~0ull : acc_hist_lower(best_k);
        if (getenv("PPF_DEBUG_ACC32")) {
          for (int b = 0; b < ACC_HIST; b++)
            if (hist[b] || hist[ACC_HIST + b])
              fprintf(stderr, "acc32 class %3d (>= %llu votes): needed %llu  not needed %llu\n", b, acc_hist_lower(b), hist[b], hist[ACC_HIST + b]);
          fprintf(stderr, "acc32 limit -> class %d (%llu votes)\n", best_k, ws->heavy_votes);
        }
      }
      if (tot[1]) ws->hit_frac = std::min(1.0, std::max(1e-3, 1.06 * (double)tot[3] / (double)tot[1]));
      if (tot[3]) ws->run_frac = std::min(1.0, std::max(0.02, 1.10 * (double)tot[4] / (double)tot[3]));
      if (tot[3]) ws->tbl_frac = std::min(TBL_FRAC_MAX, std::max(1e-4, 1.15 * (double)tot[7] / (double)tot[3]));
      ws->stats.n_tables = tot[7];
      for (int k = 0; k < 8; k++) ws->stats.phase_clocks[k] = tot[8 + k];
      ws->frac_known = true;
      if (ws->clustered) {
        uint32_t nf = 0;
        if (ws->sum_valid) nf = (uint32_t)ws->h_sum[17];
        else HIPCHK(hipMemcpy(&nf, ws->cl_u32.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
        ws->stats.n_poses = (int)nf;
      }
      if (ws->timing) {
        float pr = 0, gr = 0, vo = 0;
        for (int b = 0; b < ws->n_batches; b++) {
          float t0 = 0, t1 = 0, t2 = 0;
          const size_t e0 = ws->ev_base + (size_t)b * 4;
          HIPCHK(hipEventElapsedTime(&t0, ws->batch_ev[e0 + 0], ws->batch_ev[e0 + 1]));
          HIPCHK(hipEventElapsedTime(&t1, ws->batch_ev[e0 + 1], ws->batch_ev[e0 + 2]));
          HIPCHK(hipEventElapsedTime(&t2, ws->batch_ev[e0 + 2], ws->batch_ev[e0 + 3]));
          pr += t0; gr += t1; vo += t2;
        }
        ws->stats.ms_pair_kernel = pr; ws->stats.ms_group_kernel = gr; ws->stats.ms_vote_kernel = vo;
        HIPCHK(hipEventElapsedTime(&ws->stats.ms_total_device, ws->ev[0], ws->ev[1]));
      }
      ws->checked = true;
      return PPF_OK;
    }
    if (ws->hit_frac >= 1.0) {
      ws->pools_failed = true; /* not the caller's output buffer: a context in this state is not kept (HostLoan) */
      return fail(PPF_ERR_CAPACITY, "match: hit pools overflowed at worst-case size (flags %u)", ovf);
    }
    if (ovf & 3u) ws->hit_frac = std::min(1.0, ws->hit_frac * 2.0); /* raw or sorted hit pool */
    if (ovf & 4u) ws->run_frac = std::min(1.0, ws->run_frac * 2.0); /* run table */
    if (ovf & 8u) { /* count-table pool */
      if ((ovf & 3u) == 0 && ws->tbl_frac >= TBL_FRAC_MAX) ws->hit_frac = std::min(1.0, ws->hit_frac * 2.0); /* more hits than expected, then */
      ws->tbl_frac = std::min(TBL_FRAC_MAX, ws->tbl_frac * 2.0);
    }
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
      if (ws->sum_valid) nf = (uint32_t)ws->h_sum[17];
      else HIPCHK(hipMemcpy(&nf, ws->cl_u32.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
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

static ppf_status run_host(const ppf_model* m, const float* scene, int ns, int sstride, int snoff, const float* edge, int ne,
                           int estride, int enoff, const ppf_match_params* params, ppf_workspace* ws);

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

ppf_status ppf_debug_accumulators(const ppf_model* m, const float* scene, int ns, int sstride, int snoff, const float* edge, int ne,
                                  int estride, int enoff, const ppf_match_params* params, uint32_t* acc, size_t cap_words,
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
  ppf_status s = run_host(m, scene, ns, sstride, snoff, edge, ne, estride, enoff, &p, &ws);
  if (s != PPF_OK) return s;
  /* a cold workspace sizes its pools from estimates: when one ran out, runs were left out of the vote and the dump is incomplete.
   * workspace_finish reads the flag and repeats the call with bigger pools (match_prepared clears the dump first) -- without it
   * this entry returned a partial accumulator whenever the table pool's first guess was too small, depending on which
   * reference points reached the pool first */
  if ((s = workspace_finish(&ws)) != PPF_OK) return s;
  HIPCHK(hipStreamSynchronize(nullptr));
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
  if (i < 8 && tot_out && tot_in) tot_out[i] = tot_in[i]; /* votes, pairs, LDS operations, hits, runs, (two of k_vote's own), count tables */
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
  ws->sum_valid = false;
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

/* ---- host-buffer entries (what detector.match / match_S2B bind to) --------------------------------------------------
 * The reference calls match() once per detected object, again and again on the same detector
 * (/root/reference/include/CloudProcessing.h:441-446, :494-499 -- its own "PPF Elapsed Time" bracket).  A call therefore
 * borrows a WARM context from its model: a workspace that has already sized its hit pools and learned this scene
 * family's hit density (no counting pass, no scratch allocation), a non-blocking stream, pinned staging for the upload.
 * Contexts are handed out under the model's mutex, one per call in flight, so concurrent calls on one model from several
 * host threads stay independent.  A model keeps as many idle ones as it has had calls in flight at once (at least
 * HOST_CTX_KEEP, at most HOST_CTX_MAX): N threads matching on one detector all stay warm; ppf_model_trim_contexts releases
 * them (each holds the scratch of its last call: 0.45 GB for a C2-sized crop). */
constexpr size_t HOST_CTX_KEEP = 2;
constexpr size_t HOST_CTX_MAX = 16;

struct HostCtx {
  ppf_workspace ws;
  hipStream_t stream = nullptr;
  float* pinned[2] = {nullptr, nullptr}; /* scene, edge */
  size_t pinned_cap[2] = {0, 0};
  DevBuf<float> d_rows[2];
  int device = -1;
  ~HostCtx() {
    /* everything this context ever enqueued went to its own stream: wait for that, not for the whole device (other threads'
     * calls on other contexts keep running) */
    if (stream) (void)hipStreamSynchronize(stream); else sync_device(device);
    for (float* p : pinned)
      if (p) (void)hipHostFree(p);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

ppf_model::~ppf_model() {
  for (HostCtx* c : ctx_idle) delete c;
}

extern "C" ppf_status ppf_model_trim_contexts(const ppf_model* m, int keep, int* released) {
  if (!m || keep < 0) return fail(PPF_ERR_INVALID, "ppf_model_trim_contexts: bad argument");
  std::vector<HostCtx*> drop;
  {
    std::lock_guard<std::mutex> g(m->ctx_mu);
    while (m->ctx_idle.size() > (size_t)keep) { drop.push_back(m->ctx_idle.back()); m->ctx_idle.pop_back(); }
    m->ctx_peak = m->ctx_out; /* the high-water mark starts again: later calls keep what THEY need */
  }
  for (HostCtx* c : drop) delete c; /* waits for the context's stream, returns its scratch to the block cache */
  if (released) *released = (int)drop.size();
  return PPF_OK;
}

namespace {

/* RAII loan of a context: returned to the model's idle list when the call succeeded, destroyed otherwise (a failed call
 * leaves pools and flags in an unknown state; fail() has already drained the device) */
struct HostLoan {
  const ppf_model* m;
  HostCtx* c = nullptr;
  bool ok = false;
  bool counted = false;
  explicit HostLoan(const ppf_model* model) : m(model) {}
  ppf_status open() {
    {
      std::lock_guard<std::mutex> g(m->ctx_mu);
      if (!m->ctx_idle.empty()) { c = m->ctx_idle.back(); m->ctx_idle.pop_back(); }
      m->ctx_out++;
      m->ctx_peak = std::max(m->ctx_peak, m->ctx_out);
      counted = true;
    }
    if (c) return PPF_OK;
    std::unique_ptr<HostCtx> n(new (std::nothrow) HostCtx());
    if (!n) return fail(PPF_ERR_NOMEM, "match: out of memory");
    n->ws.model_owns_me = true;
    HIPCHK(hipGetDevice(&n->device));
    HIPCHK(hipStreamCreateWithFlags(&n->stream, hipStreamNonBlocking));
    c = n.release();
    return PPF_OK;
  }
  ~HostLoan() {
    {
      std::lock_guard<std::mutex> g(m->ctx_mu);
      if (counted) m->ctx_out--;
      if (c && ok && !c->ws.pools_failed && m->ctx_idle.size() < std::min(HOST_CTX_MAX, std::max(HOST_CTX_KEEP, m->ctx_peak))) {
        m->ctx_idle.push_back(c);
        c = nullptr;
      }
    }
    delete c; /* outside the lock: it waits for its stream */
  }
  /* host rows -> pinned staging -> device, all on the context's stream */
  ppf_status upload(int which, const float* rows, int n, int stride, const float** d_out) {
    const size_t floats = (size_t)n * stride;
    if (c->pinned_cap[which] < floats) {
      if (c->pinned[which]) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipHostFree(c->pinned[which])); }
      c->pinned[which] = nullptr; c->pinned_cap[which] = 0;
      const size_t want = floats + floats / 4; /* some slack: crops of one camera differ by a few percent */
      HIPCHK(hipHostMalloc((void**)&c->pinned[which], want * sizeof(float), hipHostMallocDefault));
      c->pinned_cap[which] = want;
    }
    memcpy(c->pinned[which], rows, floats * sizeof(float));
    HIPCHK(c->d_rows[which].reserve(floats));
    HIPCHK(hipMemcpyAsync(c->d_rows[which].p, c->pinned[which], floats * sizeof(float), hipMemcpyHostToDevice, c->stream));
    *d_out = c->d_rows[which].p;
    return PPF_OK;
  }
};

/* upload + enqueue on a borrowed context; the caller fetches through ppf_workspace_results(&loan.c->ws, ...) */
ppf_status run_host_warm(HostLoan& loan, const ppf_model* m, const float* scene, int ns, int sstride, int snoff, const float* edge, int ne,
                         int estride, int enoff, const ppf_match_params* params, bool timing) {
  if (!m) return fail(PPF_ERR_NOT_TRAINED, "The model is not trained. Cannot match without training");
  if (!have_device()) return fail(PPF_ERR_HIP, "match: no HIP device (this engine has no CPU fallback)");
  ppf_status s = check_match_args(m, scene, ns, sstride, snoff, edge, ne, estride, enoff, params);
  if (s != PPF_OK) return s;
  if ((s = loan.open()) != PPF_OK) return s;
  if ((s = ppf_workspace_enable_timing(&loan.c->ws, timing ? 1 : 0)) != PPF_OK) return s;
  const float *d_scene = nullptr, *d_edge = nullptr;
  if ((s = loan.upload(0, scene, ns, sstride, &d_scene)) != PPF_OK) return s;
  if (edge && (s = loan.upload(1, edge, ne, estride, &d_edge)) != PPF_OK) return s;
  return ppf_match_device(m, &loan.c->ws, d_scene, ns, sstride, snoff, d_edge, ne, estride, enoff, params, loan.c->stream);
}

}  // namespace

/* cold variant on the caller's own workspace and the default stream (debug entry only) */
static ppf_status run_host(const ppf_model* m, const float* scene, int ns, int sstride, int snoff, const float* edge, int ne,
                           int estride, int enoff, const ppf_match_params* params, ppf_workspace* ws) {
  if (!m) return fail(PPF_ERR_NOT_TRAINED, "The model is not trained. Cannot match without training");
  if (!have_device()) return fail(PPF_ERR_HIP, "match: no HIP device (this engine has no CPU fallback)");
  ppf_status s = check_match_args(m, scene, ns, sstride, snoff, edge, ne, estride, enoff, params);
  if (s != PPF_OK) return s;
  DevBuf<float> d_scene, d_edge;
  HIPCHK(d_scene.reserve((size_t)ns * sstride));
  HIPCHK(hipMemcpy(d_scene.p, scene, (size_t)ns * sstride * sizeof(float), hipMemcpyHostToDevice));
  if (edge) {
    HIPCHK(d_edge.reserve((size_t)ne * estride));
    HIPCHK(hipMemcpy(d_edge.p, edge, (size_t)ne * estride * sizeof(float), hipMemcpyHostToDevice));
  }
  s = ppf_match_device(m, ws, d_scene.p, ns, sstride, snoff, edge ? d_edge.p : nullptr, ne, estride, enoff, params, nullptr);
  if (s != PPF_OK) return s;
  HIPCHK(hipStreamSynchronize(nullptr));
  return PPF_OK;
}

ppf_status ppf_match(const ppf_model* m, const float* scene, int ns, int sstride, int snoff, const float* edge, int ne,
                     int estride, int enoff, const ppf_match_params* params, ppf_pose* out, int cap, int* n_out) {
  if (!n_out) return fail(PPF_ERR_INVALID, "ppf_match: n_out is NULL");
  *n_out = 0;
  if (!m) return fail(PPF_ERR_NOT_TRAINED, "The model is not trained. Cannot match without training");
  HostLoan loan(m);
  ppf_status s = run_host_warm(loan, m, scene, ns, sstride, snoff, edge, ne, estride, enoff, params, false);
  if (s == PPF_OK) s = ppf_workspace_results(&loan.c->ws, nullptr, nullptr, 0, nullptr, out, cap, n_out, nullptr);
  loan.ok = s == PPF_OK || s == PPF_ERR_CAPACITY; /* a too small output buffer leaves the context in order */
  return s;
}

ppf_status ppf_raw_votes(const ppf_model* m, const float* scene, int ns, int sstride, int snoff, const float* edge, int ne,
                         int estride, int enoff, const ppf_match_params* params, ppf_vote* votes, ppf_pose* raw_poses, int cap,
                         int* n_ref, ppf_match_stats* stats) {
  if (!m) return fail(PPF_ERR_NOT_TRAINED, "The model is not trained. Cannot match without training");
  HostLoan loan(m);
  ppf_status s = run_host_warm(loan, m, scene, ns, sstride, snoff, edge, ne, estride, enoff, params, true);
  if (s == PPF_OK) s = ppf_workspace_results(&loan.c->ws, votes, raw_poses, cap, n_ref, nullptr, 0, nullptr, stats);
  loan.ok = s == PPF_OK || s == PPF_ERR_CAPACITY;
  return s;
}

}  // extern "C"

#endif /* PPF_MATCH_HOST_H */
