/*
 * ppf_icp_host.h — host side of the ICP refinement (row N2): job pool, the lock-step level loop over concurrent
 * registrations, and the C-ABI entry points ppf_icp_refine / ppf_icp_refine_device / ppf_icp_register (+ the edge
 * helpers ppf_sample_cloud / ppf_transform_pc_pose, which run on the device too).  Kernels: ppf_icp_kernels.h.
 * Included by ppf_hip.hip (one translation unit: shares DevBuf, fail(), HIPCHK and the kernels above).
 */
/* ============================================================================================ */
/* ICP refinement (row N2; kernels in ppf_icp_kernels.h)                                          */
/* ============================================================================================ */
namespace {

struct IcpScratch {
  DevBuf<float> src0, dst0, src_pct, moved, dst_pcs;
  DevBuf<float4> q4;
  DevBuf<unsigned long long> best, owner;
  DevBuf<int2> sel;
  DevBuf<double> parts, sum_src, sum_dst;
  DevBuf<IcpState> state;
};

constexpr int ICP_BATCH = 2;     /* iterations enqueued between two reads of the done flag (measured best of 1..6) */
/* ICP_MAX_JOBS (ppf_icp_kernels.h): poses refined per batch of launches (legacy path: concurrently, one HIP stream each) */
#ifndef PPF_ICP_NN_WAVES
#define PPF_ICP_NN_WAVES 65536   /* batched path: waves of the neighbour search beyond which a wave takes several rows */
#endif
#ifndef PPF_ICP_BATCH2
#define PPF_ICP_BATCH2 2         /* batched path: passes (two launches each) kept in the stream ahead of the device (1: 3.30 ms, 2: 3.16 - 3.20, 3: 3.25 ms of ICP on C1) */
#endif

inline long icp_round(double v) { return std::lrint(v); } /* cvRound */

/* one registration in flight: its scratch, its stream, a pinned mirror of the device loop state */
struct IcpJob {
  IcpScratch sc;
  hipStream_t st = nullptr; /* borrowed: the caller's stream or one of the pool's */
  IcpState* h_st = nullptr;
  double pose[16];
  double fval_min = 9999999999.0;
  int total = 0, launched = 0;
  bool active = false;
  ~IcpJob() {
    if (h_st) (void)hipHostFree(h_st);
  }
};

/* ICP::registerModelToScene(srcPC, dstPC, residual, pose) on device-resident clouds for jobs.size() initial poses at
 * once (init_poses[j] may be NULL: register from the identity).  The registrations are independent chains of small
 * kernels, so each runs on its own stream and the host walks them in lock-step: same level, one batch of iterations
 * enqueued on every stream, then one read of every done flag. */
ppf_status icp_register_many(const float* d_src, int n, int sstride, int snoff, const float* d_dst, int nd_all, int dstride, int dnoff,
                             const ppf_icp_params& prm, const double* const* init_poses, std::vector<IcpJob*>& jobs,
                             double* poses_out /* jobs x 16 */, double* residuals, int* iters_total) {
  const size_t chunks_src = ((size_t)n + ICP_CHUNK - 1) / ICP_CHUNK, chunks_dst = ((size_t)nd_all + ICP_CHUNK - 1) / ICP_CHUNK;
  auto grid = [](size_t items, int block) { return dim3((unsigned)((items + block - 1) / block)); };
  const int robust = prm.rejection_scale > 0 ? 1 : 0;
  const int icp_batch = ICP_BATCH;
  for (size_t j = 0; j < jobs.size(); j++) {
    IcpJob& J = *jobs[j];
    IcpScratch& sc = J.sc;
    HIPCHK(sc.src0.reserve((size_t)n * 6));
    HIPCHK(sc.src_pct.reserve((size_t)n * 6));
    HIPCHK(sc.moved.reserve((size_t)n * 6));
    HIPCHK(sc.dst0.reserve((size_t)nd_all * 6));
    HIPCHK(sc.dst_pcs.reserve((size_t)nd_all * 6));
    HIPCHK(sc.q4.reserve((size_t)nd_all));
    HIPCHK(sc.best.reserve((size_t)n));
    HIPCHK(sc.owner.reserve((size_t)nd_all));
    HIPCHK(sc.sel.reserve((size_t)std::min(n, nd_all)));
    HIPCHK(sc.parts.reserve(std::max(chunks_src, chunks_dst) * ICP_ENTRIES));
    HIPCHK(sc.sum_src.reserve(chunks_src * 3));
    HIPCHK(sc.sum_dst.reserve(chunks_dst * 3));
    HIPCHK(sc.state.reserve(1));
    if (!J.h_st) HIPCHK(hipHostMalloc((void**)&J.h_st, sizeof(IcpState), hipHostMallocDefault));
    IcpState* d_st = sc.state.p;
    hipStream_t st = J.st;
    /* the two clouds, packed; the source moved by the initial pose */
    if (init_poses && init_poses[j]) {
      IcpMat44 T0;
      memcpy(T0.m, init_poses[j], sizeof(T0.m));
      k_icp_set_pose<<<dim3(1), dim3(1), 0, st>>>(d_st, T0);
      k_icp_transform<<<grid(n, 256), dim3(256), 0, st>>>(d_src, sstride, snoff, 1, n, d_st->T, sc.src0.p, nullptr, nullptr, nullptr);
    } else {
      k_icp_sample<<<grid(n, 256), dim3(256), 0, st>>>(d_src, sstride, snoff, 1, n, sc.src0.p, nullptr);
    }
    k_icp_sample<<<grid(nd_all, 256), dim3(256), 0, st>>>(d_dst, dstride, dnoff, 1, nd_all, sc.dst0.p, nullptr);
    /* centre on the average of the two means, scale to unit average distance from the origin */
    for (int mode = 0; mode < 2; mode++) {
      k_icp_chunk_sums<<<grid(chunks_src, 64), dim3(64), 0, st>>>(sc.src0.p, n, mode, sc.sum_src.p);
      k_icp_chunk_sums<<<grid(chunks_dst, 64), dim3(64), 0, st>>>(sc.dst0.p, nd_all, mode, sc.sum_dst.p);
      k_icp_reduce<<<dim3(1), dim3(64), 0, st>>>(sc.sum_src.p, n, sc.sum_dst.p, nd_all, mode, d_st);
      k_icp_center_scale<<<grid(n, 256), dim3(256), 0, st>>>(sc.src0.p, n, mode, d_st);
      k_icp_center_scale<<<grid(nd_all, 256), dim3(256), 0, st>>>(sc.dst0.p, nd_all, mode, d_st);
    }
    HIPCHK(hipGetLastError());
    for (int k = 0; k < 16; k++) J.pose[k] = (k % 5 == 0) ? 1.0 : 0.0;
    J.fval_min = 9999999999.0;
    J.total = 0;
  }
  for (int level = prm.num_levels - 1; level >= 0; level--) {
    const double div = std::pow(2.0, (double)level);
    const int num_samples = (int)icp_round((double)n / div);
    const double tol_p = (double)prm.tolerance * (double)(level + 1) * (level + 1);
    const int max_iter = (int)icp_round((double)prm.iterations / (level + 1));
    const int step = std::max(1, (int)icp_round((double)n / (double)std::max(num_samples, 1)));
    const int ns = (n + step - 1) / step, nd = (nd_all + step - 1) / step;
    /* NN launch shape: model points x scene slices, enough workgroups to fill 256 CUs */
    const unsigned gx = (unsigned)((ns + 255) / 256);
    const int max_splits = (nd + 63) / 64;
    const int splits = std::max(1, std::min(max_splits, (int)(2048 / gx)));
    const int slice = (nd + splits - 1) / splits;
    const unsigned gy = (unsigned)((nd + slice - 1) / slice);
    const unsigned n_chunks = (unsigned)((std::min(ns, nd) + ICP_CHUNK - 1) / ICP_CHUNK);
    for (auto& jp : jobs) {
      IcpJob& J = *jp;
      IcpScratch& sc = J.sc;
      IcpState* d_st = sc.state.p;
      IcpMat44 T;
      memcpy(T.m, J.pose, sizeof(T.m));
      k_icp_set_pose<<<dim3(1), dim3(1), 0, J.st>>>(d_st, T);
      k_icp_transform<<<grid(ns, 256), dim3(256), 0, J.st>>>(sc.src0.p, 6, 3, step, ns, d_st->T, sc.src_pct.p, sc.moved.p, sc.best.p, nullptr);
      k_icp_sample<<<grid(nd, 256), dim3(256), 0, J.st>>>(sc.dst0.p, 6, 3, step, nd, sc.dst_pcs.p, sc.q4.p);
      k_icp_level_init<<<dim3(1), dim3(1), 0, J.st>>>(d_st, tol_p, max_iter, robust);
      J.launched = 0;
      J.active = true;
    }
    static std::once_flag once_thr;
    static hipError_t attr_thr = hipSuccess;
    std::call_once(once_thr, [] {
      attr_thr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_icp_threshold), hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024);
    });
    HIPCHK(attr_thr);
    const int staged = ns <= 32768 ? 1 : 0; /* the level's distances fit LDS (4 bytes each): the selection passes read them there */
    const bool small = ns <= ICP_SMALL_NS && !(prm.flags & PPF_ICP_NO_SMALL_LEVELS); /* the whole level in one workgroup, one launch */
    for (bool any = true; any;) {
      for (auto& jp : jobs) {
        IcpJob& J = *jp;
        if (!J.active) continue;
        IcpScratch& sc = J.sc;
        IcpState* d_st = sc.state.p;
        hipStream_t st = J.st;
        if (small) {
          k_icp_level_small<<<dim3(1), dim3(1024), 0, st>>>(sc.src_pct.p, ns, sc.q4.p, sc.dst_pcs.p, nd, sc.owner.p, prm.rejection_scale, d_st);
          J.launched = max_iter;
        } else {
          const int batch = std::min(icp_batch, max_iter - J.launched);
          for (int b = 0; b < batch; b++) {
            k_icp_nn<<<dim3(gx, gy), dim3(256), 0, st>>>(sc.moved.p, ns, sc.q4.p, nd, slice, sc.best.p, d_st);
            k_icp_threshold<<<dim3(1), dim3(1024), staged ? (size_t)ns * 4 : 0, st>>>(sc.best.p, ns, prm.rejection_scale, sc.owner.p, nd,
                                                                                     staged, d_st);
            k_icp_owner<<<grid(ns, 256), dim3(256), 0, st>>>(sc.best.p, ns, sc.owner.p, d_st);
            k_icp_compact<<<dim3(1), dim3(1024), 0, st>>>(sc.owner.p, nd, sc.sel.p, d_st);
            k_icp_chunks<<<dim3(n_chunks), dim3(64), 0, st>>>(sc.sel.p, sc.src_pct.p, sc.dst_pcs.p, sc.parts.p, d_st);
            k_icp_solve<<<dim3(1), dim3(64), 0, st>>>(sc.parts.p, ns, d_st);
            k_icp_transform<<<grid(ns, 256), dim3(256), 0, st>>>(sc.src_pct.p, 6, 3, 1, ns, d_st->PoseX, sc.moved.p, nullptr, sc.best.p, d_st);
          }
          J.launched += std::max(batch, 0);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(J.h_st, d_st, sizeof(IcpState), hipMemcpyDeviceToHost, st));
      }
      any = false;
      for (auto& jp : jobs) {
        IcpJob& J = *jp;
        if (!J.active) continue;
        HIPCHK(hipStreamSynchronize(J.st));
        if (J.h_st->done || J.launched >= max_iter) J.active = false;
        else any = true;
      }
    }
    for (auto& jp : jobs) {
      IcpJob& J = *jp;
      J.total += J.h_st->iter;
      J.fval_min = J.h_st->fval_min;
      double tmp[16];
      ppf_mat44_mul(J.h_st->PoseX, J.pose, tmp);
      memcpy(J.pose, tmp, sizeof(tmp));
    }
  }
  for (size_t j = 0; j < jobs.size(); j++) {
    IcpJob& J = *jobs[j];
    if (prm.num_levels <= 0) {
      HIPCHK(hipMemcpyAsync(J.h_st, J.sc.state.p, sizeof(IcpState), hipMemcpyDeviceToHost, J.st));
      HIPCHK(hipStreamSynchronize(J.st));
    }
    /* undo centring and scaling: t = t/scale + meanAvg - R*meanAvg */
    const IcpState& h = *J.h_st;
    double* pose = J.pose;
    double Rm[3];
    for (int r = 0; r < 3; r++) Rm[r] = pose[r * 4] * h.mean_avg[0] + pose[r * 4 + 1] * h.mean_avg[1] + pose[r * 4 + 2] * h.mean_avg[2];
    for (int r = 0; r < 3; r++) pose[r * 4 + 3] = pose[r * 4 + 3] / h.scale + h.mean_avg[r] - Rm[r];
    memcpy(poses_out + j * 16, pose, 16 * sizeof(double));
    if (residuals) residuals[j] = J.fval_min;
    if (iters_total) iters_total[j] = J.total;
  }
  return PPF_OK;
}

/* Streams (and pinned state mirrors) for concurrent jobs come from a process-wide pool: creating a HIP stream costs
 * milliseconds, far more than a registration.  One caller at a time owns the pool (others fall back to one stream). */
struct IcpPool {
  std::mutex mu;
  IcpJob jobs[ICP_MAX_JOBS]; /* scratch buffers and pinned state mirrors persist across calls; they only grow */
  hipStream_t st[ICP_MAX_JOBS] = {};
  int device = -1;
  bool ok = false;
};
/* never destroyed: its buffers must not be freed after the HIP runtime has shut down at process exit */
IcpPool& g_icp_pool = *new IcpPool();

/* Jobs for `count` concurrent registrations.  With the pool (one caller at a time; others get private jobs on the
 * caller's stream) nothing is allocated after the first call: streams, scratch and pinned mirrors are reused.  Pool
 * streams are ordered after whatever the caller's stream has enqueued so far. */
ppf_status icp_make_jobs(int count, hipStream_t user, std::vector<IcpJob*>& jobs, std::vector<std::unique_ptr<IcpJob>>& owned,
                         std::unique_lock<std::mutex>& pool_lock, bool one_stream = false) {
  jobs.clear();
  owned.clear();
  bool pooled = false;
  pool_lock = std::unique_lock<std::mutex>(g_icp_pool.mu, std::try_to_lock);
  if (pool_lock.owns_lock()) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (!g_icp_pool.ok || g_icp_pool.device != dev) {
      for (int j = 0; j < ICP_MAX_JOBS; j++) {
        if (g_icp_pool.st[j]) (void)hipStreamDestroy(g_icp_pool.st[j]);
        g_icp_pool.st[j] = nullptr;
        HIPCHK(hipStreamCreateWithFlags(&g_icp_pool.st[j], hipStreamNonBlocking));
        g_icp_pool.jobs[j].~IcpJob();
        new (&g_icp_pool.jobs[j]) IcpJob(); /* buffers of another device are dropped */
      }
      g_icp_pool.device = dev;
      g_icp_pool.ok = true;
    }
    pooled = true;
  }
  const bool own_streams = pooled && count > 1 && !one_stream;
  hipEvent_t ready = nullptr;
  if (own_streams) {
    HIPCHK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ready, user));
  }
  for (int j = 0; j < count; j++) {
    IcpJob* J;
    if (pooled) {
      J = &g_icp_pool.jobs[j];
    } else {
      owned.emplace_back(new IcpJob());
      J = owned.back().get();
    }
    J->st = own_streams ? g_icp_pool.st[j] : user;
    if (own_streams) {
      const hipError_t e = hipStreamWaitEvent(J->st, ready, 0);
      if (e != hipSuccess) { (void)hipEventDestroy(ready); return fail(PPF_ERR_HIP, "ICP: hipStreamWaitEvent failed: %s", hipGetErrorString(e)); }
    }
    jobs.push_back(J);
  }
  if (ready) (void)hipEventDestroy(ready);
  return PPF_OK;
}

/* ---- batched path (default; kernels k_icp2_*) ------------------------------------------------------------------- */
struct IcpBatchScratch {
  DevBuf<float> src0, dst0, src_pct;
  DevBuf<unsigned long long> best, owner;
  DevBuf<int2> sel;
  DevBuf<double> parts, sum_src, sum_dst;
  DevBuf<float4> g_pts, g_box2;
  DevBuf<uint32_t> g_start, g_cur, g_box1u, own_a;
  DevBuf<float> bb_parts;
  DevBuf<IcpState2> state;
  int* h_done = nullptr;        /* pinned: one flag per job, written by the kernels */
  IcpState2* h_state = nullptr; /* pinned: the jobs' loop states, written by the kernels when a level ends */
  unsigned long long* h_ticks = nullptr; /* pinned: finished k_icp2_tail workgroups of the running call */
  ~IcpBatchScratch() {
    if (h_done) (void)hipHostFree(h_done);
    if (h_state) (void)hipHostFree(h_state);
    if (h_ticks) (void)hipHostFree(h_ticks);
  }
};

/* registerModelToScene for `jobs` initial poses at once (init_poses NULL: one registration from the identity): every
 * launch covers all jobs, an iteration is two launches (k_icp2_nn, k_icp2_tail), the host keeps PPF_ICP_BATCH2 of them in
 * the stream ahead of the device and reads the jobs' done flags whenever one reports.  Everything runs on `st`. */
ppf_status icp_register_batch(const float* d_src, int n, int sstride, int snoff, const float* d_dst, int nd_all, int dstride, int dnoff,
                              const ppf_icp_params& prm, const double* const* init_poses, int jobs, IcpBatchScratch& sc, hipStream_t st,
                              double* poses_out /* jobs x 16 */, double* residuals, int* iters_total) {
  const size_t chunks_src = ((size_t)n + ICP_CHUNK - 1) / ICP_CHUNK, chunks_dst = ((size_t)nd_all + ICP_CHUNK - 1) / ICP_CHUNK;
  const size_t J = (size_t)jobs;
  IcpBatch B;
  memset(&B, 0, sizeof(B));
  B.p_sel = (size_t)std::min(n, nd_all);
  B.p_parts = ((size_t)std::min(n, nd_all) + ICP_CHUNK - 1) / ICP_CHUNK * ICP_ENTRIES;
  B.p_sums = chunks_src * 3;
  B.p_sumd = chunks_dst * 3;
  HIPCHK(sc.src0.reserve(J * n * 6));
  HIPCHK(sc.src_pct.reserve(J * n * 6));
  HIPCHK(sc.dst0.reserve(J * nd_all * 6));
  HIPCHK(sc.best.reserve(J * n));
  HIPCHK(sc.owner.reserve(J * nd_all));
  HIPCHK(sc.sel.reserve(J * B.p_sel));
  HIPCHK(sc.parts.reserve(J * B.p_parts));
  HIPCHK(sc.sum_src.reserve(J * B.p_sums));
  HIPCHK(sc.sum_dst.reserve(J * B.p_sumd));
  HIPCHK(sc.g_pts.reserve(J * nd_all));
  HIPCHK(sc.g_start.reserve(J * (ICP_LEAVES + 64)));
  HIPCHK(sc.g_cur.reserve(J * ICP_LEAVES));
  HIPCHK(sc.g_box2.reserve(J * ICP_LEAVES * 2));
  HIPCHK(sc.g_box1u.reserve(J * 64 * 8));
  HIPCHK(sc.own_a.reserve(J * nd_all));
  HIPCHK(sc.bb_parts.reserve(J * B.p_sumd * 2));
  HIPCHK(sc.state.reserve(ICP_MAX_JOBS));
  if (!sc.h_done) HIPCHK(hipHostMalloc((void**)&sc.h_done, ICP_MAX_JOBS * sizeof(int), hipHostMallocDefault));
  if (!sc.h_state) HIPCHK(hipHostMalloc((void**)&sc.h_state, ICP_MAX_JOBS * sizeof(IcpState2), hipHostMallocDefault));
  if (!sc.h_ticks) HIPCHK(hipHostMalloc((void**)&sc.h_ticks, 64, hipHostMallocDefault));
  *sc.h_ticks = 0ull; /* no kernel of an earlier call is running: every call ends with all of its launches accounted for */
  B.src = d_src; B.dst = d_dst;
  B.n = n; B.sstride = sstride; B.snoff = snoff; B.nd_all = nd_all; B.dstride = dstride; B.dnoff = dnoff;
  B.src0 = sc.src0.p; B.dst0 = sc.dst0.p; B.src_pct = sc.src_pct.p;
  B.best = sc.best.p; B.owner = sc.owner.p; B.sel = sc.sel.p;
  B.parts = sc.parts.p; B.sum_src = sc.sum_src.p; B.sum_dst = sc.sum_dst.p;
  B.g_pts = sc.g_pts.p; B.g_start = sc.g_start.p; B.g_cur = sc.g_cur.p; B.g_box2 = sc.g_box2.p; B.g_box1u = sc.g_box1u.p; B.own_a = sc.own_a.p; B.bb_parts = sc.bb_parts.p;
  B.state = sc.state.p;
  B.h_done = sc.h_done;
  B.h_ticks = sc.h_ticks;
  B.h_state = sc.h_state;
  B.has_init = init_poses ? 1 : 0;
  for (int j = 0; j < jobs; j++)
    for (int k = 0; k < 16; k++) B.T0[j][k] = init_poses && init_poses[j] ? init_poses[j][k] : ((k % 5 == 0) ? 1.0 : 0.0);
  static std::once_flag once_tail;
  static hipError_t attr_tail = hipSuccess;
  std::call_once(once_tail, [] {
    attr_tail = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_icp2_tail), hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
  });
  HIPCHK(attr_tail);
  const unsigned uj = (unsigned)jobs;
  /* the two clouds packed (the source moved by its job's initial pose), centred on the average of the two means, scaled to
   * unit average distance from the origin; the search grid over each job's scene */
  k_icp2_reset<<<dim3(uj, 32), dim3(256), 0, st>>>(B);
  k_icp2_pack_sums<<<dim3((unsigned)(chunks_src + chunks_dst), uj), dim3(64), 0, st>>>(B);
  k_icp2_mean<<<dim3(uj), dim3(256), 0, st>>>(B);
  k_icp2_dist_sums<<<dim3((unsigned)(chunks_src + chunks_dst), uj), dim3(64), 0, st>>>(B);
  k_icp2_scale<<<dim3(uj), dim3(256), 0, st>>>(B);
  const unsigned nb_dst = (unsigned)((nd_all + ICP_ROWS_BLOCK - 1) / ICP_ROWS_BLOCK), nb_src = (unsigned)((n + ICP_ROWS_BLOCK - 1) / ICP_ROWS_BLOCK);
  k_icp2_rows_count<<<dim3(nb_dst + nb_src, uj), dim3(1024), 0, st>>>(B);
  k_icp2_grid_scan<<<dim3(uj), dim3(1024), 0, st>>>(B);
  k_icp2_grid_scatter<<<dim3(nb_dst, uj), dim3(1024), 0, st>>>(B);
  k_icp2_grid_boxes<<<dim3(ICP_LEAVES / 4, uj), dim3(256), 0, st>>>(B);
  HIPCHK(hipGetLastError());
  const int robust = prm.rejection_scale > 0 ? 1 : 0;
  unsigned long long tails_launched = 0;
  bool last_level_waited = false;
  auto wait_ticks = [&](const unsigned long long want) -> ppf_status {
    volatile unsigned long long* ticks = sc.h_ticks;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (*ticks < want) {
      __builtin_ia32_pause();
      if ((++spins & 0xFFFFu) == 0) {
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited > 0.05 && hipStreamQuery(st) != hipErrorNotReady) { /* the stream is idle (or broken): nothing more will come */
          HIPCHK(hipStreamSynchronize(st));
          if (*ticks < want) return fail(PPF_ERR_HIP, "ICP: %llu of %llu workgroups reported", (unsigned long long)*ticks, want);
        }
        if (waited > 30.0) return fail(PPF_ERR_HIP, "ICP: timed out waiting for the device");
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return PPF_OK;
  };
  for (int level = prm.num_levels - 1; level >= 0; level--) {
    const double div = std::pow(2.0, (double)level);
    const int num_samples = (int)icp_round((double)n / div);
    const double tol_p = (double)prm.tolerance * (double)(level + 1) * (level + 1);
    const int max_iter = (int)icp_round((double)prm.iterations / (level + 1));
    const int step = std::max(1, (int)icp_round((double)n / (double)std::max(num_samples, 1)));
    const int ns = (n + step - 1) / step, nd = (nd_all + step - 1) / step;
    int step_shift = -1;
    if ((step & (step - 1)) == 0) { step_shift = 0; while ((1 << step_shift) < step) step_shift++; }
    k_icp2_level_begin<<<dim3((unsigned)((ns + 255) / 256), uj), dim3(256), 0, st>>>(B, step, ns, tol_p, max_iter, robust, level == 0 ? 1 : 0);
    const int staged = ns <= 32768 ? 1 : 0; /* the level's distances fit LDS (4 bytes each): the selection passes read them there */
    const size_t tail_lds = std::max<size_t>(staged ? (size_t)ns * 4 : 0, (size_t)ICP_TAIL_VAL_BYTES);
    /* rows per wave of the neighbour search: one, unless that makes more waves than the chip holds several times over */
    const int nn_rows = (int)std::min<long long>(ICP_NN_ROWS, std::max<long long>(1, ((long long)ns * jobs) / PPF_ICP_NN_WAVES));
    const unsigned nn_blocks = (unsigned)((ns + 4 * nn_rows - 1) / (4 * nn_rows));
    /* The passes of a level are launched ahead of the device: PPF_ICP_BATCH2 (neighbour search, tail) pairs are in the stream,
     * and every time the oldest of them reports, the next one is launched -- the device never waits for the host between
     * passes (with whole batches it idled ~11 us after every second pass).  Every k_icp2_tail workgroup, whatever it did,
     * adds one to a counter in pinned memory as its last act (after its done flag and, at the end of a level, its state);
     * polling that is a few microseconds quicker than going through the stream (which is only asked when the counter has
     * not moved for a long while).  A pass launched after the level's last one finds every job done and returns at once. */
    int launched = 0, reported = 0;
    unsigned long long due[PPF_ICP_BATCH2 + 1]; /* the counter's value when pass k has reported, k modulo the passes in the stream */
    bool first_of_level = true;
    while (true) {
      while (launched < max_iter && launched - reported < (int)PPF_ICP_BATCH2) {
        /* a pass is launched for the jobs that had not finished the level when the host last looked (all of them at its start) */
        IcpLive live;
        unsigned nl = 0;
        for (int j = 0; j < jobs; j++)
          if (first_of_level || reinterpret_cast<volatile int*>(sc.h_done)[j] == 0) live.job[nl++] = j;
        if (nl == 0) live.job[nl++] = 0; /* cannot happen: the loop ends when every job is done */
        for (unsigned k = nl; k < (unsigned)ICP_MAX_JOBS; k++) live.job[k] = live.job[0];
        k_icp2_nn<<<dim3(nn_blocks, nl), dim3(256), 0, st>>>(B, live, ns, nd, step, step_shift, nn_rows, (prm.flags & PPF_ICP_GRID_ALWAYS) ? 0 : ICP_BRUTE_ND);
        k_icp2_tail<<<dim3(nl), dim3(1024), tail_lds, st>>>(B, live, ns, nd, step, prm.rejection_scale, staged, level == 0 ? 1 : 0);
        tails_launched += (unsigned long long)nl;
        due[launched % (PPF_ICP_BATCH2 + 1)] = tails_launched;
        launched++;
      }
      first_of_level = false;
      HIPCHK(hipGetLastError());
      if (reported >= launched) break; /* max_iter == 0, or every launched pass has reported */
      if (ppf_status rc = wait_ticks(due[reported % (PPF_ICP_BATCH2 + 1)])) return rc;
      reported++;
      bool all = true;
      for (int j = 0; j < jobs; j++) all &= reinterpret_cast<volatile int*>(sc.h_done)[j] != 0;
      if (all) break; /* the passes still in the stream return at once; the next level's launches queue up behind them */
    }
    last_level_waited = launched > 0;
  }
  /* the passes launched ahead of the last level's end are still in the stream and will report too: the scratch (and its
   * counter) goes back to the pool only when they have */
  if (ppf_status rc = wait_ticks(tails_launched)) return rc;
  if (!last_level_waited) HIPCHK(hipStreamSynchronize(st)); /* nothing was polled after the last state went out */
  for (int j = 0; j < jobs; j++) {
    /* undo centring and scaling: t = t/scale + meanAvg - R*meanAvg */
    const IcpState2& h = sc.h_state[j];
    double pose[16];
    memcpy(pose, h.pose, sizeof(pose));
    double Rm[3];
    for (int r = 0; r < 3; r++) Rm[r] = pose[r * 4] * h.mean_avg[0] + pose[r * 4 + 1] * h.mean_avg[1] + pose[r * 4 + 2] * h.mean_avg[2];
    for (int r = 0; r < 3; r++) pose[r * 4 + 3] = pose[r * 4 + 3] / h.scale + h.mean_avg[r] - Rm[r];
    memcpy(poses_out + (size_t)j * 16, pose, sizeof(pose));
    if (residuals) residuals[j] = h.fval_min;
    if (iters_total) iters_total[j] = h.total;
#ifdef PPF_ICP_CLOCKS
    fprintf(stderr, "icp job %d: start + staging %.1f  median %.1f  MAD %.1f  ownership %.1f  compaction %.1f  chunks %.1f  sums %.1f  solve %.1f us (iterations %d)\n", j,
            h.ph[6] * 0.01, h.ph[7] * 0.01, h.ph[0] * 0.01, h.ph[1] * 0.01, h.ph[2] * 0.01, h.ph[3] * 0.01, h.ph[4] * 0.01, h.ph[5] * 0.01, h.total);
#endif
  }
  return PPF_OK;
}

/* one process-wide scratch for the batched path (buffers only grow); a second concurrent caller works on a private one */
struct IcpBatchPool {
  std::mutex mu;
  IcpBatchScratch sc;
  int device = -1;
};
IcpBatchPool& g_icp_batch_pool = *new IcpBatchPool(); /* never destroyed: nothing is freed after the HIP runtime has shut down */

ppf_status icp_batch_run(const float* d_src, int n, int sstride, int snoff, const float* d_dst, int nd_all, int dstride, int dnoff,
                         const ppf_icp_params& prm, const double* const* init_poses, int jobs, hipStream_t st, double* poses_out,
                         double* residuals, int* iters_total) {
  std::unique_lock<std::mutex> lock(g_icp_batch_pool.mu, std::try_to_lock);
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (lock.owns_lock() && (g_icp_batch_pool.device == dev || g_icp_batch_pool.device < 0)) {
    g_icp_batch_pool.device = dev;
    return icp_register_batch(d_src, n, sstride, snoff, d_dst, nd_all, dstride, dnoff, prm, init_poses, jobs, g_icp_batch_pool.sc, st,
                              poses_out, residuals, iters_total);
  }
  IcpBatchScratch priv;
  const ppf_status s = icp_register_batch(d_src, n, sstride, snoff, d_dst, nd_all, dstride, dnoff, prm, init_poses, jobs, priv, st, poses_out,
                                          residuals, iters_total);
  (void)hipStreamSynchronize(st); /* the private scratch goes back to the block cache */
  return s;
}

ppf_status icp_check(const char* who, const void* src, int n, int sstride, int snoff, const void* dst, int nd, int dstride, int dnoff,
                     const ppf_icp_params* prm) {
  if (!src || !dst || !prm || n <= 0 || nd <= 0 || bad_layout(sstride, snoff) || bad_layout(dstride, dnoff))
    return fail(PPF_ERR_INVALID, "%s: bad argument", who);
  if (prm->iterations < 0 || prm->num_levels < 0 || prm->num_levels > 30 || !(prm->tolerance >= 0))
    return fail(PPF_ERR_INVALID, "%s: bad ICP parameters", who);
  if (!have_device()) return fail(PPF_ERR_HIP, "%s: no HIP device (this engine has no CPU fallback)", who);
  return PPF_OK;
}

/* Pose3D::appendPose: pose = incremental * pose, then q / t / angle from the new matrix */
void icp_append_pose(ppf_pose* p, const double* inc, double residual) {
  double out[16];
  ppf_mat44_mul(inc, p->pose, out);
  memcpy(p->pose, out, sizeof(out));
  const double R[9] = {out[0], out[1], out[2], out[4], out[5], out[6], out[8], out[9], out[10]};
  p->t[0] = out[3]; p->t[1] = out[7]; p->t[2] = out[11];
  ppf_dcm_to_quat(R, p->q);
  p->angle = ppf_angle_from_trace(R[0] + R[4] + R[8]);
  p->residual = residual;
}

ppf_status icp_refine_device(const float* d_model, int n, int mstride, int mnoff, const float* d_scene, int nd, int sstride, int snoff,
                             const ppf_icp_params* prm, ppf_pose* poses, int n_poses, int* iters, hipStream_t st) {
  for (int k0 = 0; k0 < n_poses; k0 += ICP_MAX_JOBS) {
    const int cnt = std::min(ICP_MAX_JOBS, n_poses - k0);
    if (!(prm->flags & PPF_ICP_LEGACY)) {
      const double* init[ICP_MAX_JOBS];
      double inc[ICP_MAX_JOBS * 16], res[ICP_MAX_JOBS];
      int it[ICP_MAX_JOBS];
      for (int j = 0; j < cnt; j++) init[j] = poses[k0 + j].pose;
      const ppf_status s = icp_batch_run(d_model, n, mstride, mnoff, d_scene, nd, sstride, snoff, *prm, init, cnt, st, inc, res, it);
      if (s != PPF_OK) return s;
      for (int j = 0; j < cnt; j++) {
        icp_append_pose(&poses[k0 + j], inc + j * 16, res[j]);
        if (iters) iters[k0 + j] = it[j];
      }
      continue;
    }
    std::vector<IcpJob*> jobs;
    std::vector<std::unique_ptr<IcpJob>> owned;
    std::unique_lock<std::mutex> pool_lock;
    ppf_status s = icp_make_jobs(cnt, st, jobs, owned, pool_lock, (prm->flags & PPF_ICP_ONE_STREAM) != 0);
    if (s != PPF_OK) return s;
    const double* init[ICP_MAX_JOBS];
    double inc[ICP_MAX_JOBS * 16], res[ICP_MAX_JOBS];
    int it[ICP_MAX_JOBS];
    for (int j = 0; j < cnt; j++) init[j] = poses[k0 + j].pose;
    s = icp_register_many(d_model, n, mstride, mnoff, d_scene, nd, sstride, snoff, *prm, init, jobs, inc, res, it);
    if (s != PPF_OK) return s;
    for (int j = 0; j < cnt; j++) {
      icp_append_pose(&poses[k0 + j], inc + j * 16, res[j]);
      if (iters) iters[k0 + j] = it[j];
    }
  }
  return PPF_OK;
}

/* host rows (x y z at 0, normal at noff) -> packed device rows of 6 */
ppf_status icp_upload(const float* h, int n, int stride, int noff, DevBuf<float>& d) {
  HIPCHK(d.reserve((size_t)n * 6));
  if (noff == 3) {
    HIPCHK(hipMemcpy2D(d.p, 6 * sizeof(float), h, (size_t)stride * sizeof(float), 6 * sizeof(float), (size_t)n, hipMemcpyHostToDevice));
  } else {
    HIPCHK(hipMemcpy2D(d.p, 6 * sizeof(float), h, (size_t)stride * sizeof(float), 3 * sizeof(float), (size_t)n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy2D(d.p + 3, 6 * sizeof(float), h + noff, (size_t)stride * sizeof(float), 3 * sizeof(float), (size_t)n, hipMemcpyHostToDevice));
  }
  return PPF_OK;
}

}  // namespace

extern "C" {

void ppf_default_icp_params(ppf_icp_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->iterations = 100; /* ICP icp(100, 0.005f, 2.5f, 8), CloudProcessing.h:465,518 */
  p->tolerance = 0.005f;
  p->rejection_scale = 2.5f;
  p->num_levels = 8;
}

ppf_status ppf_icp_refine(const float* model, int n_model, int mstride, int mnoff, const float* scene, int n_scene, int sstride,
                          int snoff, const ppf_icp_params* params, ppf_pose* poses_io, int n_poses, int* iterations_out) {
  ppf_status s = icp_check("ppf_icp_refine", model, n_model, mstride, mnoff, scene, n_scene, sstride, snoff, params);
  if (s != PPF_OK) return s;
  if (n_poses < 0 || (n_poses > 0 && !poses_io)) return fail(PPF_ERR_INVALID, "ppf_icp_refine: bad pose list");
  DevBuf<float> dm, ds;
  if ((s = icp_upload(model, n_model, mstride, mnoff, dm)) != PPF_OK) return s;
  if ((s = icp_upload(scene, n_scene, sstride, snoff, ds)) != PPF_OK) return s;
  return icp_refine_device(dm.p, n_model, 6, 3, ds.p, n_scene, 6, 3, params, poses_io, n_poses, iterations_out, nullptr);
}

ppf_status ppf_icp_refine_device(const float* d_model, int n_model, int mstride, int mnoff, const float* d_scene, int n_scene,
                                 int sstride, int snoff, const ppf_icp_params* params, ppf_pose* poses_io, int n_poses,
                                 int* iterations_out, void* stream) {
  ppf_status s = icp_check("ppf_icp_refine_device", d_model, n_model, mstride, mnoff, d_scene, n_scene, sstride, snoff, params);
  if (s != PPF_OK) return s;
  if (n_poses < 0 || (n_poses > 0 && !poses_io)) return fail(PPF_ERR_INVALID, "ppf_icp_refine_device: bad pose list");
  return icp_refine_device(d_model, n_model, mstride, mnoff, d_scene, n_scene, sstride, snoff, params, poses_io, n_poses,
                           iterations_out, (hipStream_t)stream);
}

ppf_status ppf_icp_register(const float* src, int n_src, int sstride, int snoff, const float* dst, int n_dst, int dstride, int dnoff,
                            const ppf_icp_params* params, double* pose16_out, double* residual_out, int* iterations_out) {
  ppf_status s = icp_check("ppf_icp_register", src, n_src, sstride, snoff, dst, n_dst, dstride, dnoff, params);
  if (s != PPF_OK) return s;
  if (!pose16_out) return fail(PPF_ERR_INVALID, "ppf_icp_register: pose16_out is NULL");
  DevBuf<float> dsrc, ddst;
  if ((s = icp_upload(src, n_src, sstride, snoff, dsrc)) != PPF_OK) return s;
  if ((s = icp_upload(dst, n_dst, dstride, dnoff, ddst)) != PPF_OK) return s;
  if (!(params->flags & PPF_ICP_LEGACY))
    return icp_batch_run(dsrc.p, n_src, 6, 3, ddst.p, n_dst, 6, 3, *params, nullptr, 1, nullptr, pose16_out, residual_out, iterations_out);
  std::vector<IcpJob*> jobs;
  std::vector<std::unique_ptr<IcpJob>> owned;
  std::unique_lock<std::mutex> pool_lock;
  if ((s = icp_make_jobs(1, nullptr, jobs, owned, pool_lock)) != PPF_OK) return s;
  return icp_register_many(dsrc.p, n_src, 6, 3, ddst.p, n_dst, 6, 3, *params, nullptr, jobs, pose16_out, residual_out, iterations_out);
}

/* ---- helpers on the path's edges, on the device like everything else ---------------------------------------- */
ppf_status ppf_sample_cloud(const float* xyzn, int n, int stride, int noff, double relative_step, float* out, int cap_rows,
                            int* n_out) {
  if (!xyzn || n <= 0 || bad_layout(stride, noff) || !(relative_step > 0)) return fail(PPF_ERR_INVALID, "ppf_sample_cloud: bad argument");
  if (!have_device()) return fail(PPF_ERR_HIP, "ppf_sample_cloud: no HIP device (this engine has no CPU fallback)");
  DevBuf<float> d_raw;
  HIPCHK(d_raw.reserve((size_t)n * stride));
  HIPCHK(hipMemcpy(d_raw.p, xyzn, (size_t)n * stride * sizeof(float), hipMemcpyHostToDevice));
  CloudDev sampled;
  std::vector<float> rows_host;
  ppf_status s = device_sample_cloud(d_raw.p, n, stride, noff, (float)relative_step, sampled, &rows_host, nullptr);
  if (s != PPF_OK) return s;
  const int rows = (int)(rows_host.size() / 6);
  if (n_out) *n_out = rows;
  if (out) {
    if (cap_rows < rows) return fail(PPF_ERR_CAPACITY, "ppf_sample_cloud: need %d rows, have %d", rows, cap_rows);
    memcpy(out, rows_host.data(), rows_host.size() * sizeof(float));
  }
  return PPF_OK;
}

ppf_status ppf_transform_pc_pose(const float* xyzn, int n, int stride, int noff, const double* T, float* out) {
  if (!xyzn || !T || !out || n < 0 || bad_layout(stride, noff)) return fail(PPF_ERR_INVALID, "ppf_transform_pc_pose: bad argument");
  if (!have_device()) return fail(PPF_ERR_HIP, "ppf_transform_pc_pose: no HIP device (this engine has no CPU fallback)");
  if (n == 0) return PPF_OK;
  DevBuf<float> d_in, d_out;
  DevBuf<double> d_T;
  ppf_status s = icp_upload(xyzn, n, stride, noff, d_in);
  if (s != PPF_OK) return s;
  HIPCHK(d_out.reserve((size_t)n * 6));
  HIPCHK(d_T.reserve(16));
  HIPCHK(hipMemcpy(d_T.p, T, 16 * sizeof(double), hipMemcpyHostToDevice));
  k_icp_transform<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(d_in.p, 6, 3, 1, n, d_T.p, d_out.p, nullptr, nullptr, nullptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, d_out.p, (size_t)n * 6 * sizeof(float), hipMemcpyDeviceToHost));
  return PPF_OK;
}

}  // extern "C"
