/*
 * ppf_batch_host.h — C-ABI, many crops x many models in one call (BASELINE config C5; the per-object loop around Matching,
 * /root/reference/include/CloudProcessing.h:41-45,58).
 */
#ifndef PPF_BATCH_HOST_H
#define PPF_BATCH_HOST_H

extern "C" {

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
  bool timing = false; /* ppf_batch_enable_timing: HIP events around the kernels of every match */
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

ppf_status ppf_batch_enable_timing(ppf_batch* b, int on) {
  if (!b) return fail(PPF_ERR_INVALID, "ppf_batch_enable_timing: NULL");
  for (auto* w : b->ws) {
    ppf_status s = ppf_workspace_enable_timing(w, on);
    if (s != PPF_OK) return s;
  }
  b->timing = on != 0;
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
                         int sstride, int snoff, int n_scenes, int scenes_on_device, const ppf_match_params* params, ppf_pose* out,
                         int cap, int* n_out, ppf_batch_stats* stats) {
  if (!b || !models || n_models <= 0 || !scenes || !ns || n_scenes <= 0 || !params || cap <= 0)
    return fail(PPF_ERR_INVALID, "ppf_batch_run: bad argument");
  for (int k = 0; k < n_models; k++)
    if (!models[k]) return fail(PPF_ERR_NOT_TRAINED, "ppf_batch_run: model %d is not trained", k);
  ppf_match_params p = *params;
  if (p.skip_clustering) return fail(PPF_ERR_INVALID, "ppf_batch_run: a batch returns clustered poses");
  for (int c = 0; c < n_scenes; c++) {
    ppf_status s = check_match_args(models[0], scenes[c], ns[c], sstride, snoff, nullptr, 0, 6, 3, &p);
    if (s != PPF_OK) return s;
  }
  for (int k = 1; k < n_models; k++) /* every table must live on the device the batch runs on */
    if (models[k]->device != models[0]->device)
      return fail(PPF_ERR_INVALID, "ppf_batch_run: model %d lives on device %d, model 0 on device %d", k, models[k]->device, models[0]->device);
  const auto t_start = std::chrono::steady_clock::now();
  const size_t n_match = (size_t)n_scenes * n_models;
  HIPCHK(b->d_out.reserve(n_match * cap));
  HIPCHK(b->d_meta.reserve(n_match * 2));
  HIPCHK(b->d_tot.reserve(n_match * 8));
  b->last_records = (int)(n_match * cap);
  const int words = cap * (int)(sizeof(ppf_pose) / 8);

  struct TimedCall { int lane; size_t ev_base; int n_batches; };
  std::vector<TimedCall> timed; /* the event sets of this run's matches (timing on): read after the lanes have drained */
  for (auto* w : b->ws) w->ev_base = 0;
  /* one (crop, model) match on a lane, results saved to the block */
  auto enqueue_pair = [&](int lane, int c, int k) -> ppf_status {
    ppf_workspace* ws = b->ws[lane];
    hipStream_t st = b->streams[lane];
    ppf_status s = match_prepared(models[k], ws, &p, st);
    if (s != PPF_OK) return s;
    if (b->timing && ws->n_ref > 0) {
      timed.push_back({lane, ws->ev_base, ws->n_batches});
      ws->ev_base += (size_t)ws->n_batches * 4; /* the next match on this lane records into its own events */
    }
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
    return prepare_scene(b->ws[lane], d_src, ns[c], sstride, snoff, nullptr, 0, 6, 3, &p, st);
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
        ppf_workspace::Learned* fm = workspace_learned(ws, models[k], true);
        const uint32_t flags = meta[idx * 2 + 1];
        at_full = fm->hit >= 1.0;
        if (flags & 3u) fm->hit = std::min(1.0, 2.0 * fm->hit);
        if (flags & 4u) fm->run = std::min(1.0, 2.0 * fm->run);
        if (flags & 8u) {
          if ((flags & 3u) == 0 && fm->tbl >= TBL_FRAC_MAX) fm->hit = std::min(1.0, 2.0 * fm->hit);
          fm->tbl = std::min(TBL_FRAC_MAX, 2.0 * fm->tbl);
        }
        s = enqueue_pair(lane, c, k);
        if (s != PPF_OK) return s;
        HIPCHK(hipStreamSynchronize(b->streams[lane]));
        HIPCHK(hipMemcpy(&meta[idx * 2], b->d_meta.p + idx * 2, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&tot[idx * 8], b->d_tot.p + idx * 8, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      }
    }
  }
  /* what every lane remembers per model: the densest crop it saw */
  std::vector<double> lane_hit((size_t)b->lanes * n_models, 0.0), lane_run((size_t)b->lanes * n_models, 0.0), lane_tbl((size_t)b->lanes * n_models, 0.0);
  ppf_batch_stats st{};
  for (size_t idx = 0; idx < n_match; idx++) {
    const unsigned long long* t = &tot[idx * 8];
    st.n_votes += t[0]; st.n_pairs += t[1]; st.n_lds_atomics += t[2]; st.n_hits += t[3];
    const size_t slot = (size_t)((idx / n_models) % b->lanes) * n_models + idx % n_models;
    if (t[1]) lane_hit[slot] = std::max(lane_hit[slot], (double)t[3] / (double)t[1]);
    if (t[3]) lane_run[slot] = std::max(lane_run[slot], (double)t[4] / (double)t[3]);
    if (t[3]) lane_tbl[slot] = std::max(lane_tbl[slot], (double)t[7] / (double)t[3]);
  }
  for (int l = 0; l < b->lanes; l++) {
    ppf_workspace* ws = b->ws[l];
    workspace_hold_model(ws, nullptr); /* the next call looks its model up */
    for (int k = 0; k < n_models; k++) {
      const double fh = lane_hit[(size_t)l * n_models + k], fr = lane_run[(size_t)l * n_models + k], ft = lane_tbl[(size_t)l * n_models + k];
      if (!(fh > 0)) continue;
      const double hit = std::min(1.0, std::max(1e-3, 1.06 * fh)), run = std::min(1.0, std::max(0.02, 1.10 * fr));
      ppf_workspace::Learned* fm = workspace_learned(ws, models[k], true);
      fm->hit = hit; fm->run = run; fm->tbl = std::min(TBL_FRAC_MAX, std::max(1e-4, 1.15 * ft));
    }
  }
  if (out) HIPCHK(hipMemcpy(out, b->d_out.p, n_match * cap * sizeof(ppf_pose), hipMemcpyDeviceToHost));
  if (n_out)
    for (size_t idx = 0; idx < n_match; idx++) n_out[idx] = (int)std::min<uint32_t>(meta[idx * 2], (uint32_t)cap);
  for (const TimedCall& tc : timed) { /* kernel times of every match (retries after an overflow included) */
    ppf_workspace* ws = b->ws[tc.lane];
    for (int bi = 0; bi < tc.n_batches; bi++) {
      const size_t e0 = tc.ev_base + (size_t)bi * 4;
      float t0 = 0, t1 = 0, t2 = 0;
      HIPCHK(hipEventElapsedTime(&t0, ws->batch_ev[e0 + 0], ws->batch_ev[e0 + 1]));
      HIPCHK(hipEventElapsedTime(&t1, ws->batch_ev[e0 + 1], ws->batch_ev[e0 + 2]));
      HIPCHK(hipEventElapsedTime(&t2, ws->batch_ev[e0 + 2], ws->batch_ev[e0 + 3]));
      st.ms_pair_kernel += t0; st.ms_group_kernel += t1; st.ms_vote_kernel += t2;
    }
  }
  for (auto* w : b->ws) w->ev_base = 0;
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
                           int sstride, int snoff, int n_scenes, const ppf_match_params* params, ppf_pose* out, int cap, int* n_out) {
  if (!models || n_models <= 0 || !scenes || !ns || n_scenes <= 0 || !params || !out || cap <= 0 || !n_out)
    return fail(PPF_ERR_INVALID, "ppf_match_batch: bad argument");
  for (int k = 0; k < n_models; k++)
    if (!models[k]) return fail(PPF_ERR_NOT_TRAINED, "ppf_match_batch: model %d is not trained", k);
  if (!have_device()) return fail(PPF_ERR_HIP, "ppf_match_batch: no HIP device (this engine has no CPU fallback)");
  ppf_batch* b = nullptr;
  ppf_status s = ppf_batch_create(std::min(4, n_scenes), &b);
  if (s != PPF_OK) return s;
  s = ppf_batch_run(b, models, n_models, scenes, ns, sstride, snoff, n_scenes, 0, params, out, cap, n_out, nullptr);
  (void)ppf_batch_destroy(b);
  return s;
}

}  // extern "C"

#endif /* PPF_BATCH_HOST_H */
