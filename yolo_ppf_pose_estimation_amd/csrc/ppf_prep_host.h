/*
 * ppf_prep_host.h — host side of the stages that produce the matcher's input (row N4): the device-resident ppf_cloud,
 * ordered compaction, the uniform-grid kNN driver, the C-ABI entry points ppf_cloud_* / ppf_prep_* and the resident
 * ppf_match_clouds / ppf_icp_refine_clouds.  Kernels: ppf_prep_kernels.h.  Included by ppf_hip.hip (one translation
 * unit: shares DevBuf, fail(), HIPCHK, sort_segments and the kernels above).
 */
/* ============================================================================================ */
/* Pre-processing stages (row N4; kernels in ppf_prep_kernels.h)                                  */
/* ============================================================================================ */
struct ppf_cloud {
  DevBuf<float> rows; /* n x 6: x y z nx ny nz */
  DevBuf<float> curv; /* n */
  int n = 0;
};

/* cells per axis of the neighbour-search grid = sqrt(n) / PPF_KNN_GDIV.  Swept in round 4 (normals(30) / SOR(50) on the reference's
 * 10,395 / 11,369-point clouds and on 50,000 points): 6.0: 0.280 / 0.359 / 1.27 / 1.39 ms; 4.5: 0.255 / 0.350 / 1.09 / 1.22;
 * 4.0: 0.252 / 0.353 / 1.07 / 1.19; 3.0: 0.253 / 0.382 / 1.01 / 1.17; 2.0: 0.330 / 0.530 -- finer cells, fewer candidates per
 * step of the merge network, until the cube has to grow a second time for most queries.  The result does not depend on it. */
#ifndef PPF_KNN_GDIV
#define PPF_KNN_GDIV 4.0
#endif
namespace {

inline dim3 grid_for(size_t items, int block) { return dim3((unsigned)std::max<size_t>((items + block - 1) / block, 1)); }

ppf_status cloud_alloc(std::unique_ptr<ppf_cloud>& c, int n) {
  c.reset(new ppf_cloud());
  c->n = n;
  HIPCHK(c->rows.reserve((size_t)std::max(n, 1) * 6));
  HIPCHK(c->curv.reserve((size_t)std::max(n, 1)));
  return PPF_OK;
}

/* ordered compaction of the rows whose flag is set */
ppf_status cloud_compact(const ppf_cloud* in, const DevBuf<uint32_t>& flags, ppf_cloud** out) {
  const int n = in->n;
  DevBuf<uint32_t> pos;
  HIPCHK(pos.reserve((size_t)n + 1));
  ppf_status s = device_exclusive_scan(flags.p, pos.p, (size_t)n + 1, nullptr);
  if (s != PPF_OK) return s;
  uint32_t kept = 0;
  HIPCHK(hipMemcpy(&kept, pos.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost));
  std::unique_ptr<ppf_cloud> c;
  if ((s = cloud_alloc(c, (int)kept)) != PPF_OK) return s;
  if (kept) {
    k_prep_gather<<<grid_for(n, 256), dim3(256)>>>(in->rows.p, in->curv.p, n, flags.p, pos.p, c->rows.p, c->curv.p);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipDeviceSynchronize());
  *out = c.release();
  return PPF_OK;
}

ppf_status prep_check(const char* who, const ppf_cloud* in, ppf_cloud** out) {
  if (!out) return fail(PPF_ERR_INVALID, "%s: out is NULL", who);
  *out = nullptr;
  if (!in) return fail(PPF_ERR_INVALID, "%s: cloud is NULL", who);
  if (!have_device()) return fail(PPF_ERR_HIP, "%s: no HIP device (this engine has no CPU fallback)", who);
  return PPF_OK;
}

/* exact kNN lists of every point of the cloud: idx/d2 are [n][k], k <= min(n, KNN_MAX_K); q4 = xyz by original row */
ppf_status cloud_knn(const ppf_cloud* in, int k, DevBuf<float4>& q4, DevBuf<int>& idx, DevBuf<float>& d2) {
  const int n = in->n;
  HIPCHK(q4.reserve((size_t)n));
  HIPCHK(idx.reserve((size_t)n * k));
  HIPCHK(d2.reserve((size_t)n * k));
  DevBuf<float> scratch;
  HIPCHK(scratch.reserve((size_t)n * 6));
  k_icp_sample<<<grid_for(n, 256), dim3(256)>>>(in->rows.p, 6, 3, 1, n, scratch.p, q4.p);
  /* grid over the bounding box: about sqrt(n)/6 cells along the longest side (a 3x3x3 cube of a surface-like cloud
   * then holds a few hundred points), at most 128 */
  DevBuf<uint32_t> mm;
  HIPCHK(mm.reserve(6));
  const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
  HIPCHK(hipMemcpy(mm.p, init, sizeof(init), hipMemcpyHostToDevice));
  k_prep_minmax<<<dim3(std::max(1u, std::min<unsigned>((unsigned)((n + 2047) / 2048), 256u))), dim3(256)>>>(in->rows.p, n, mm.p);
  HIPCHK(hipGetLastError());
  uint32_t h_mm[6];
  HIPCHK(hipMemcpy(h_mm, mm.p, sizeof(h_mm), hipMemcpyDeviceToHost));
  KnnGrid g;
  float ext_max = 0.f;
  for (int a = 0; a < 3; a++) {
    const float lo = ordered_to_float(h_mm[a]), hi = ordered_to_float(h_mm[3 + a]);
    if (!std::isfinite(lo) || !std::isfinite(hi))
      return fail(PPF_ERR_INVALID, "neighbour search: the cloud holds non-finite points (crop or voxel-grid it first)");
    g.lo[a] = lo;
    ext_max = std::max(ext_max, hi - lo);
  }
  const int G = std::max(1, std::min(128, (int)(std::sqrt((double)n) / PPF_KNN_GDIV)));
  g.h = ext_max > 0.f ? ext_max / (float)G : 1.0f;
  g.inv_h = 1.0f / g.h;
  size_t cells = 1;
  for (int a = 0; a < 3; a++) {
    const float hi = ordered_to_float(h_mm[3 + a]);
    g.dim[a] = std::max(1, std::min(G + 1, (int)std::floor((hi - g.lo[a]) * g.inv_h) + 1));
    cells *= (size_t)g.dim[a];
  }
  DevBuf<uint32_t> keys, vals, keys2, vals2, starts, cell_count, cell_begin;
  HIPCHK(keys.reserve(n)); HIPCHK(vals.reserve(n)); HIPCHK(keys2.reserve(n)); HIPCHK(vals2.reserve(n));
  HIPCHK(cell_count.reserve(cells + 1)); HIPCHK(cell_begin.reserve(cells + 1));
  HIPCHK(hipMemset(cell_count.p, 0, (cells + 1) * sizeof(uint32_t)));
  k_prep_knn_keys<<<grid_for(n, 256), dim3(256)>>>(in->rows.p, n, g, keys.p, vals.p, cell_count.p);
  HIPCHK(hipGetLastError());
  ppf_status s = device_exclusive_scan(cell_count.p, cell_begin.p, cells + 1, nullptr);
  if (s != PPF_OK) return s;
  uint32_t n_runs = 0;
  uint32_t* order = nullptr;
  if ((s = sort_segments(keys, vals, keys2, vals2, n, (unsigned long long)cells, starts, &order, &n_runs, nullptr)) != PPF_OK) return s;
  DevBuf<float4> pts;
  HIPCHK(pts.reserve((size_t)n));
  k_prep_knn_pack<<<grid_for(n, 256), dim3(256)>>>(in->rows.p, order, n, pts.p);
  k_prep_knn<<<grid_for(n, KNN_WAVES), dim3(KNN_WAVES * 64)>>>(pts.p, cell_begin.p, g, n, k, idx.p, d2.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  return PPF_OK;
}

}  // namespace

extern "C" {

ppf_status ppf_cloud_upload(const float* rows, int n, int stride, int noff, int cols, ppf_cloud** out) {
  if (!out) return fail(PPF_ERR_INVALID, "ppf_cloud_upload: out is NULL");
  *out = nullptr;
  if (!rows || n < 0 || (cols != 3 && cols != 6) || stride < cols || (cols == 6 && bad_layout(stride, noff)))
    return fail(PPF_ERR_INVALID, "ppf_cloud_upload: bad argument");
  if (!have_device()) return fail(PPF_ERR_HIP, "ppf_cloud_upload: no HIP device (this engine has no CPU fallback)");
  std::unique_ptr<ppf_cloud> c;
  ppf_status s = cloud_alloc(c, n);
  if (s != PPF_OK) return s;
  if (n) {
    DevBuf<float> raw;
    HIPCHK(raw.reserve((size_t)n * stride));
    HIPCHK(hipMemcpy(raw.p, rows, (size_t)n * stride * sizeof(float), hipMemcpyHostToDevice));
    k_prep_pack<<<grid_for(n, 256), dim3(256)>>>(raw.p, n, stride, noff, cols, c->rows.p, c->curv.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
  }
  *out = c.release();
  return PPF_OK;
}
ppf_status ppf_cloud_release(ppf_cloud* c) {
  if (c) (void)hipDeviceSynchronize(); /* its rows return to the block cache: no kernel may still be reading them */
  delete c;
  return PPF_OK;
}
ppf_status ppf_cloud_size(const ppf_cloud* c, int* n) {
  if (!c || !n) return fail(PPF_ERR_INVALID, "ppf_cloud_size: NULL");
  *n = c->n;
  return PPF_OK;
}
ppf_status ppf_cloud_download(const ppf_cloud* c, float* rows6, float* curvature, int cap_rows) {
  if (!c) return fail(PPF_ERR_INVALID, "ppf_cloud_download: NULL");
  if (cap_rows < c->n) return fail(PPF_ERR_CAPACITY, "ppf_cloud_download: need %d rows, have %d", c->n, cap_rows);
  if (rows6 && c->n) HIPCHK(hipMemcpy(rows6, c->rows.p, (size_t)c->n * 6 * sizeof(float), hipMemcpyDeviceToHost));
  if (curvature && c->n) HIPCHK(hipMemcpy(curvature, c->curv.p, (size_t)c->n * sizeof(float), hipMemcpyDeviceToHost));
  return PPF_OK;
}
ppf_status ppf_cloud_device_rows(const ppf_cloud* c, const float** d_rows6, int* n) {
  if (!c || !d_rows6 || !n) return fail(PPF_ERR_INVALID, "ppf_cloud_device_rows: NULL");
  *d_rows6 = c->rows.p;
  *n = c->n;
  return PPF_OK;
}

/* SceneCropping (CloudProcessing.h:263-339) for one bounding box */
ppf_status ppf_prep_crop(const ppf_cloud* in, const int* box_xywh, const float* depth, int depth_rows, int depth_cols,
                         const double* intr, ppf_cloud** out) {
  ppf_status s = prep_check("ppf_prep_crop", in, out);
  if (s != PPF_OK) return s;
  if (!box_xywh || !depth || !intr || depth_rows <= 0 || depth_cols <= 0) return fail(PPF_ERR_INVALID, "ppf_prep_crop: bad argument");
  double left = box_xywh[0] - 30; if (left < 0) left = 0;
  double top = box_xywh[1] - 30; if (top < 0) top = 0;
  double right = box_xywh[0] + box_xywh[2] + 30; if (right >= depth_cols) right = depth_cols - 1;
  double bottom = box_xywh[1] + box_xywh[3] + 30; if (bottom >= depth_rows) bottom = depth_rows - 1;
  const int il = (int)left, it = (int)top, ir = (int)right, ib = (int)bottom;
  if (il < 0 || it < 0 || ir >= depth_cols || ib >= depth_rows || il > ir || it > ib) return fail(PPF_ERR_INVALID, "ppf_prep_crop: box outside the depth image");
  const float d1 = depth[(size_t)it * depth_cols + il], d2 = depth[(size_t)it * depth_cols + ir],
              d3 = depth[(size_t)ib * depth_cols + il], d4 = depth[(size_t)ib * depth_cols + ir];
  const float davg = (d1 + d2 + d3 + d4) / 4;
  const double fx = intr[0], fy = intr[1], ppx = intr[2], ppy = intr[3];
  auto back_project = [&](int u, int v, float* o) { /* Camera::back_projection_bbox, Camera.h:50-61 */
    o[0] = (float)((double)((float)((double)u - ppx) * davg) / fx);
    o[1] = (float)((double)((float)((double)v - ppy) * davg) / fy);
    o[2] = (float)((double)davg + 0.15); /* corners pushed 0.15 m back, :292-295 */
  };
  float c[4][3];
  back_project(il, it, c[0]); back_project(il, ib, c[1]); back_project(ir, it, c[2]); back_project(ir, ib, c[3]);
  CropPlanes pl;
  pl.z_base = c[0][2];
  const double ctr[3] = {((double)c[0][0] + c[1][0] + c[2][0] + c[3][0]) / 4, ((double)c[0][1] + c[1][1] + c[2][1] + c[3][1]) / 4, (double)pl.z_base};
  const int face[4][2] = {{0, 1}, {1, 3}, {3, 2}, {2, 0}};
  for (int f = 0; f < 4; f++) {
    const float* a = c[face[f][0]]; const float* b = c[face[f][1]];
    pl.n[f][0] = (double)a[1] * b[2] - (double)a[2] * b[1];
    pl.n[f][1] = (double)a[2] * b[0] - (double)a[0] * b[2];
    pl.n[f][2] = (double)a[0] * b[1] - (double)a[1] * b[0];
    const double sgn = pl.n[f][0] * ctr[0] + pl.n[f][1] * ctr[1] + pl.n[f][2] * ctr[2];
    if (sgn < 0) { pl.n[f][0] = -pl.n[f][0]; pl.n[f][1] = -pl.n[f][1]; pl.n[f][2] = -pl.n[f][2]; }
  }
  DevBuf<uint32_t> flags;
  HIPCHK(flags.reserve((size_t)in->n + 1));
  HIPCHK(hipMemset(flags.p + in->n, 0, sizeof(uint32_t)));
  if (in->n) k_prep_crop_flags<<<grid_for(in->n, 256), dim3(256)>>>(in->rows.p, in->n, pl, flags.p);
  HIPCHK(hipGetLastError());
  return cloud_compact(in, flags, out);
}

/* Subsampling (:361-380): pcl::VoxelGrid with a cubic leaf */
ppf_status ppf_prep_voxel_grid(const ppf_cloud* in, double leaf, ppf_cloud** out) {
  ppf_status s = prep_check("ppf_prep_voxel_grid", in, out);
  if (s != PPF_OK) return s;
  if (!((float)leaf > 0.f)) return fail(PPF_ERR_INVALID, "ppf_prep_voxel_grid: leaf size must be positive");
  /* non-finite points do not take part */
  DevBuf<uint32_t> fin;
  HIPCHK(fin.reserve((size_t)in->n + 1));
  HIPCHK(hipMemset(fin.p + in->n, 0, sizeof(uint32_t)));
  if (in->n) k_prep_finite_flags<<<grid_for(in->n, 256), dim3(256)>>>(in->rows.p, in->n, fin.p);
  ppf_cloud* dense_raw = nullptr;
  if ((s = cloud_compact(in, fin, &dense_raw)) != PPF_OK) return s;
  std::unique_ptr<ppf_cloud> dense(dense_raw);
  const int n = dense->n;
  std::unique_ptr<ppf_cloud> c;
  if (n == 0) {
    if ((s = cloud_alloc(c, 0)) != PPF_OK) return s;
    *out = c.release();
    return PPF_OK;
  }
  DevBuf<uint32_t> mm, keys, vals, keys2, vals2, starts;
  HIPCHK(mm.reserve(6));
  const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
  HIPCHK(hipMemcpy(mm.p, init, sizeof(init), hipMemcpyHostToDevice));
  k_prep_minmax<<<dim3(std::max(1u, std::min<unsigned>((unsigned)((n + 2047) / 2048), 256u))), dim3(256)>>>(dense->rows.p, n, mm.p);
  HIPCHK(hipGetLastError());
  uint32_t h_mm[6];
  HIPCHK(hipMemcpy(h_mm, mm.p, sizeof(h_mm), hipMemcpyDeviceToHost));
  VoxelGridDims g;
  g.inv_leaf = 1.0f / (float)leaf;
  long long cells = 1;
  for (int k = 0; k < 3; k++) {
    const float lo = ordered_to_float(h_mm[k]), hi = ordered_to_float(h_mm[3 + k]);
    g.min_b[k] = (int)std::floor(lo * g.inv_leaf);
    const int max_b = (int)std::floor(hi * g.inv_leaf);
    g.div_b[k] = max_b - g.min_b[k] + 1;
    cells *= g.div_b[k];
    if (cells > 0x7fffffffLL) return fail(PPF_ERR_INVALID, "ppf_prep_voxel_grid: leaf size is too small for the cloud (index overflow)");
  }
  HIPCHK(keys.reserve(n)); HIPCHK(vals.reserve(n)); HIPCHK(keys2.reserve(n)); HIPCHK(vals2.reserve(n));
  k_prep_voxel_keys<<<grid_for(n, 256), dim3(256)>>>(dense->rows.p, n, g, keys.p, vals.p);
  HIPCHK(hipGetLastError());
  uint32_t n_cells = 0;
  uint32_t* va = nullptr;
  if ((s = sort_segments(keys, vals, keys2, vals2, n, (unsigned long long)cells, starts, &va, &n_cells, nullptr)) != PPF_OK) return s;
  if ((s = cloud_alloc(c, (int)n_cells)) != PPF_OK) return s;
  k_prep_voxel_sum<<<grid_for(n_cells, 64), dim3(64)>>>(dense->rows.p, va, starts.p, (int)n_cells, n, c->rows.p, c->curv.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  *out = c.release();
  return PPF_OK;
}

/* exact k nearest neighbours of every point (debug / parity surface): idx and d2 are [n][k]; missing = -1 / 0 */
ppf_status ppf_prep_knn(const ppf_cloud* in, int k, int* idx, float* d2) {
  if (!in || !idx || !d2 || k < 1 || k > KNN_MAX_K) return fail(PPF_ERR_INVALID, "ppf_prep_knn: bad argument (k <= %d)", KNN_MAX_K);
  if (!have_device()) return fail(PPF_ERR_HIP, "ppf_prep_knn: no HIP device (this engine has no CPU fallback)");
  const int n = in->n, ke = std::min(k, n);
  if (n == 0) return PPF_OK;
  DevBuf<float4> q4; DevBuf<int> d_idx; DevBuf<float> d_d2;
  ppf_status s = cloud_knn(in, ke, q4, d_idx, d_d2);
  if (s != PPF_OK) return s;
  std::vector<int> hi((size_t)n * ke);
  std::vector<float> hd((size_t)n * ke);
  HIPCHK(hipMemcpy(hi.data(), d_idx.p, hi.size() * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hd.data(), d_d2.p, hd.size() * sizeof(float), hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++)
    for (int m = 0; m < k; m++) {
      idx[(size_t)i * k + m] = m < ke ? hi[(size_t)i * ke + m] : -1;
      d2[(size_t)i * k + m] = m < ke ? hd[(size_t)i * ke + m] : 0.f;
    }
  return PPF_OK;
}

/* OutlierProcessing (:341-360): pcl::StatisticalOutlierRemoval(meanK, stddevMul) */
ppf_status ppf_prep_outlier_removal(const ppf_cloud* in, int mean_k, double stddev_mul, ppf_cloud** out) {
  ppf_status s = prep_check("ppf_prep_outlier_removal", in, out);
  if (s != PPF_OK) return s;
  if (mean_k < 1 || mean_k + 1 > KNN_MAX_K) return fail(PPF_ERR_INVALID, "ppf_prep_outlier_removal: meanK must be in [1, %d]", KNN_MAX_K - 1);
  const int n = in->n;
  DevBuf<uint32_t> flags;
  HIPCHK(flags.reserve((size_t)n + 1));
  HIPCHK(hipMemset(flags.p + n, 0, sizeof(uint32_t)));
  if (n) {
    DevBuf<float4> q4; DevBuf<int> idx; DevBuf<float> d2, dist;
    DevBuf<double> parts, thr;
    const int valid = n > mean_k ? 1 : 0;
    if (valid && (s = cloud_knn(in, mean_k + 1, q4, idx, d2)) != PPF_OK) return s;
    HIPCHK(dist.reserve(n));
    HIPCHK(parts.reserve((size_t)((n + 63) / 64) * 2));
    HIPCHK(thr.reserve(1));
    k_prep_sor_dist<<<grid_for(n, 256), dim3(256)>>>(d2.p, n, mean_k, valid, dist.p);
    k_prep_sor_chunks<<<grid_for((n + 63) / 64, 64), dim3(64)>>>(dist.p, n, parts.p);
    k_prep_sor_threshold<<<dim3(1), dim3(64)>>>(parts.p, n, stddev_mul, thr.p);
    k_prep_sor_flags<<<grid_for(n, 256), dim3(256)>>>(dist.p, n, thr.p, flags.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
  }
  return cloud_compact(in, flags, out);
}

/* NormalEstimation (:381-405): k nearest neighbours, plane fit, normal towards the camera, curvature */
ppf_status ppf_prep_normals(const ppf_cloud* in, int k, ppf_cloud** out) {
  ppf_status s = prep_check("ppf_prep_normals", in, out);
  if (s != PPF_OK) return s;
  if (k < 1 || k > KNN_MAX_K) return fail(PPF_ERR_INVALID, "ppf_prep_normals: k must be in [1, %d]", KNN_MAX_K);
  const int n = in->n;
  std::unique_ptr<ppf_cloud> c;
  if ((s = cloud_alloc(c, n)) != PPF_OK) return s;
  if (n) {
    HIPCHK(hipMemcpy(c->rows.p, in->rows.p, (size_t)n * 6 * sizeof(float), hipMemcpyDeviceToDevice));
    DevBuf<float4> q4; DevBuf<int> idx; DevBuf<float> d2;
    const int ke = std::min(k, n);
    if ((s = cloud_knn(in, ke, q4, idx, d2)) != PPF_OK) return s;
    k_prep_normals<<<grid_for(n, 64), dim3(64)>>>(c->rows.p, c->curv.p, n, idx.p, ke, q4.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
  }
  *out = c.release();
  return PPF_OK;
}

/* EdgeExtraction (:406-427): points whose curvature exceeds the threshold */
ppf_status ppf_prep_edges(const ppf_cloud* in, float curvature_threshold, ppf_cloud** out) {
  ppf_status s = prep_check("ppf_prep_edges", in, out);
  if (s != PPF_OK) return s;
  DevBuf<uint32_t> flags;
  HIPCHK(flags.reserve((size_t)in->n + 1));
  HIPCHK(hipMemset(flags.p + in->n, 0, sizeof(uint32_t)));
  if (in->n) k_prep_curv_flags<<<grid_for(in->n, 256), dim3(256)>>>(in->curv.p, in->n, curvature_threshold, flags.p);
  HIPCHK(hipGetLastError());
  return cloud_compact(in, flags, out);
}

/* PointCloudXYZNormalToMat (:163-190): the N x 6 rows the detector consumes, normals re-normalised */
ppf_status ppf_prep_to_mat(const ppf_cloud* in, ppf_cloud** out) {
  ppf_status s = prep_check("ppf_prep_to_mat", in, out);
  if (s != PPF_OK) return s;
  std::unique_ptr<ppf_cloud> c;
  if ((s = cloud_alloc(c, in->n)) != PPF_OK) return s;
  if (in->n) {
    k_prep_to_mat<<<grid_for(in->n, 256), dim3(256)>>>(in->rows.p, in->n, c->rows.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(c->curv.p, in->curv.p, (size_t)in->n * sizeof(float), hipMemcpyDeviceToDevice));
    HIPCHK(hipDeviceSynchronize());
  }
  *out = c.release();
  return PPF_OK;
}

/* ---- the PPF calls on device-resident clouds: the whole chain after the detector's boxes without host copies --- */
ppf_status ppf_match_clouds(const ppf_model* m, const ppf_cloud* scene, const ppf_cloud* edge, const ppf_match_params* params,
                            ppf_pose* out, int cap, int* n_out) {
  if (!n_out) return fail(PPF_ERR_INVALID, "ppf_match_clouds: n_out is NULL");
  *n_out = 0;
  if (!scene) return fail(PPF_ERR_INVALID, "ppf_match_clouds: scene is NULL");
  if (!m) return fail(PPF_ERR_NOT_TRAINED, "The model is not trained. Cannot match without training");
  if (!have_device()) return fail(PPF_ERR_HIP, "ppf_match_clouds: no HIP device (this engine has no CPU fallback)");
  ppf_status s = check_match_args(m, scene->rows.p, scene->n, 6, 3, edge ? edge->rows.p : nullptr, edge ? edge->n : 0, 6, 3, params);
  if (s != PPF_OK) return s;
  /* a warm context of the model (ppf_match_host.h); the clouds are resident, so nothing is staged.  The stages that made
   * them ran on the default stream and waited for it before returning. */
  HostLoan loan(m);
  if ((s = loan.open()) != PPF_OK) return s;
  if ((s = ppf_workspace_enable_timing(&loan.c->ws, 0)) != PPF_OK) return s;
  s = ppf_match_device(m, &loan.c->ws, scene->rows.p, scene->n, 6, 3, edge ? edge->rows.p : nullptr, edge ? edge->n : 0, 6, 3, params,
                       loan.c->stream);
  if (s == PPF_OK) s = ppf_workspace_results(&loan.c->ws, nullptr, nullptr, 0, nullptr, out, cap, n_out, nullptr);
  loan.ok = s == PPF_OK || s == PPF_ERR_CAPACITY;
  return s;
}

ppf_status ppf_icp_refine_clouds(const ppf_cloud* model, const ppf_cloud* scene, const ppf_icp_params* params, ppf_pose* poses_io,
                                 int n_poses, int* iterations_out) {
  if (!model || !scene) return fail(PPF_ERR_INVALID, "ppf_icp_refine_clouds: cloud is NULL");
  return ppf_icp_refine_device(model->rows.p, model->n, 6, 3, scene->rows.p, scene->n, 6, 3, params, poses_io, n_poses, iterations_out, nullptr);
}

}  // extern "C"
