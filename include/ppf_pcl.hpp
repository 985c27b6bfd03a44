/*
 * ppf_pcl.hpp — header-only facade with the shape of PCL's PPF pipeline
 * (pcl::PPFEstimation / pcl::PPFHashMapSearch / pcl::PPFRegistration), which BASELINE.json's north_star names,
 * over the same C-ABI (ppf_hip.h).  The reference itself never calls these classes (it uses PCL only
 * for I/O, cropping, voxel grid, outlier removal and normals; its PPF calls go to OpenCV's PPF3DDetector, see
 * ppf_match_3d.hpp), so this header exists for code written against the PCL names:
 *
 *     PPFEstimation<PointNormal, PointNormal, PPFSignature> est;
 *     est.setInputCloud(model); est.setInputNormals(model); est.compute(*model_ppf);
 *     PPFHashMapSearch::Ptr search(new PPFHashMapSearch(12.0f / 180.0f * M_PI, 0.05f));
 *     search->setInputFeatureCloud(model_ppf);
 *     PPFRegistration<PointNormal, PointNormal> reg;
 *     reg.setSceneReferencePointSamplingRate(10);
 *     reg.setPositionClusteringThreshold(0.2f); reg.setRotationClusteringThreshold(30.0f / 180.0f * M_PI);
 *     reg.setSearchMethod(search); reg.setInputSource(model); reg.setInputTarget(scene);
 *     reg.align(output);  reg.getFinalTransformation();
 *
 * SEMANTICS (stated, not hidden): these classes switch the engine's PCL policy flags on -- the pair feature is
 * pcl::computePairFeatures' (ppf_train_params.feature = PPF_FEATURE_DARBOUX, keys floor(f / step)), the model table is
 * keyed on the exact quantised feature (key_equality = PPF_KEY_EXACT, PPFHashMapSearch's hash map with key equality),
 * a reference point is paired with the scene points within model_diameter / 2
 * (ppf_match_params.pair_radius, PPFRegistration's kd-tree radius search), alpha differences are wrapped into
 * [-pi, pi] and binned over 2*pi (alpha_range_2pi), and poses cluster on the angle of their relative rotation
 * (rot_metric_relative).  What remains different from PCL proper: fp64 arithmetic where PCL computes in float (pairs within
 * rounding of a bin edge can land in the neighbouring bin), and the engine's reference frame / alpha convention (the
 * reference's library's, applied to model and scene alike, so alpha differences agree).  Point types only need members
 * x, y, z, normal_x, normal_y, normal_z (pcl::PointNormal qualifies); clouds only need `.points` or to be a
 * std::vector of such points, and are handed to the engine WHERE THEY ARE (stride 12 floats, normals at float 4 for
 * pcl::PointNormal): no host-side repack.  Normals are used as given, as PCL does (the reference's Mat adaptor
 * re-normalises them; ppf_prep_to_mat is that step on the device).  Compiles without PCL and without Eigen.
 */
#ifndef PPF_PCL_HPP
#define PPF_PCL_HPP

#include <array>
#include <cmath>
#include <cstddef>
#include <functional>
#include <memory>
#include <utility>
#include <type_traits>
#include <vector>

#include "ppf_match_3d.hpp"

namespace ppfhip {
namespace pcl_shaped {

/* pcl::PointNormal's storage for builds without PCL: 48 bytes, x y z 1 | normal_x normal_y normal_z 0 | curvature pad pad pad
 * (PCL aligns the struct to 16 bytes; the float positions are what matters here) */
struct PointNormal {
  float x, y, z, pad0;
  float normal_x, normal_y, normal_z, pad1;
  float curvature, pad2[3];
  PointNormal() : x(0), y(0), z(0), pad0(1.f), normal_x(0), normal_y(0), normal_z(0), pad1(0), curvature(0), pad2{0, 0, 0} {}
  PointNormal(float x_, float y_, float z_, float nx, float ny, float nz)
      : x(x_), y(y_), z(z_), pad0(1.f), normal_x(nx), normal_y(ny), normal_z(nz), pad1(0), curvature(0), pad2{0, 0, 0} {}
};

struct PPFSignature {
  float f1, f2, f3, f4, alpha_m;
};

template <class PointT>
struct PointCloud {
  std::vector<PointT> points;
  std::size_t size() const { return points.size(); }
  typedef std::shared_ptr<PointCloud<PointT>> Ptr;
  typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
};

namespace detail {
template <class C> auto pts(const C& c) -> decltype(c.points) const& { return c.points; }
template <class P> const std::vector<P>& pts(const std::vector<P>& v) { return v; }
template <class C> auto pts_mut(C& c) -> decltype(c.points)& { return c.points; }
template <class P> std::vector<P>& pts_mut(std::vector<P>& v) { return v; }

/* A cloud of PointNormal-like points as the C-ABI's (pointer, rows, stride, normal_offset): the caller's storage itself
 * when the point type keeps x y z and normal_x normal_y normal_z as two runs of three consecutive floats (pcl::PointNormal
 * does: stride 12, normal offset 4), so nothing is repacked on the host -- unlike the reference's
 * PointCloudXYZNormalToMat (CloudProcessing.h:163-190), which copies every point into an N x 6 Mat first.  Any other
 * point type is copied into N x 6 rows.  A view holds the caller's smart pointer, not an address: `refresh()` takes pointer
 * and row count from the cloud AS IT IS THEN, so points that were resized or reallocated between setInputCloud() and
 * compute() / align() -- legal in PCL, which reads through the cloud pointer at compute time -- are read where they are now. */
struct RowsView {
  const float* p = nullptr;
  int n = 0, stride = 6, noff = PPF_NOFF_MAT;
  std::shared_ptr<void> keep;             /* the caller's cloud pointer, or the copied rows */
  std::function<void(RowsView&)> resolve; /* fills p / n / stride / noff from the cloud */
  void refresh() { if (resolve) resolve(*this); }
};

template <class CloudPtr>
RowsView view_of(const CloudPtr& cloud) {
  RowsView v;
  v.resolve = [cloud](RowsView& r) {
    const auto& p = pts(*cloud);
    typedef typename std::decay<decltype(p[0])>::type PointT;
    r.n = (int)p.size();
    r.p = nullptr;
    if (p.empty()) return;
    /* member offsets in bytes from the point's first byte (no pointer arithmetic across members) */
    const char* b0 = reinterpret_cast<const char*>(&p[0]);
    const std::ptrdiff_t ox = reinterpret_cast<const char*>(&p[0].x) - b0, oy = reinterpret_cast<const char*>(&p[0].y) - b0,
                         oz = reinterpret_cast<const char*>(&p[0].z) - b0, nx = reinterpret_cast<const char*>(&p[0].normal_x) - b0,
                         ny = reinterpret_cast<const char*>(&p[0].normal_y) - b0, nz = reinterpret_cast<const char*>(&p[0].normal_z) - b0;
    const std::ptrdiff_t F = (std::ptrdiff_t)sizeof(float);
    if (sizeof(PointT) % sizeof(float) == 0 && oy == ox + F && oz == ox + 2 * F && nx >= ox + 3 * F && (nx - ox) % F == 0 && ny == nx + F &&
        nz == nx + 2 * F && (std::size_t)(nz + F) <= sizeof(PointT)) {
      r.p = reinterpret_cast<const float*>(b0 + ox);
      r.stride = (int)(sizeof(PointT) / sizeof(float));
      r.noff = (int)((nx - ox) / F);
      r.keep = std::make_shared<CloudPtr>(cloud); /* a copy of the caller's smart pointer: the points stay where they are */
      return;
    }
    auto rows = std::make_shared<std::vector<float>>(p.size() * 6);
    for (std::size_t i = 0; i < p.size(); i++) {
      float* d = &(*rows)[i * 6];
      d[0] = p[i].x; d[1] = p[i].y; d[2] = p[i].z;
      d[3] = p[i].normal_x; d[4] = p[i].normal_y; d[5] = p[i].normal_z;
    }
    r.p = rows->data();
    r.stride = 6;
    r.noff = PPF_NOFF_MAT;
    r.keep = rows;
  };
  v.refresh();
  return v;
}
}  // namespace detail

/* The feature cloud of the PCL pipeline: PPFEstimation::compute fills `points` with the N x N PPFSignature rows PCL
 * materialises (row i*N + j = pair (i, j): pcl::computePairFeatures' f1..f4 and alpha_m; rows i == j are NaN), computed on
 * the device (ppf_pair_features).  PPFHashMapSearch trains from the model rows, which travel along as a view. */
struct PPFFeatureCloud {
  std::vector<PPFSignature> points;
  detail::RowsView model;
  std::size_t size() const { return points.size(); }
  typedef std::shared_ptr<PPFFeatureCloud> Ptr;
};

template <class PointInT, class PointNT, class PointOutT = PPFSignature>
class PPFEstimation {
 public:
  template <class CloudPtr> void setInputCloud(const CloudPtr& cloud) { model_ = detail::view_of(cloud); }
  template <class CloudPtr> void setInputNormals(const CloudPtr&) {} /* normals travel with the points */
  void compute(PPFFeatureCloud& out) {
    model_.refresh(); /* the cloud as it is now */
    out.model = model_;
    const std::size_t n = (std::size_t)model_.n;
    out.points.assign(n * n, PPFSignature());
    if (n) {
      static_assert(sizeof(PPFSignature) == 5 * sizeof(float), "PPFSignature must be five packed floats");
      ppf_match_3d::check(ppf_pair_features(model_.p, model_.n, model_.stride, model_.noff, PPF_FEATURE_DARBOUX,
                                            reinterpret_cast<float*>(out.points.data()), n * n));
    }
  }

 private:
  detail::RowsView model_;
};

class PPFHashMapSearch {
 public:
  typedef std::shared_ptr<PPFHashMapSearch> Ptr;
  PPFHashMapSearch(float angle_discretization_step = 12.0f / 180.0f * 3.14159265358979f,
                   float distance_discretization_step = 0.01f)
      : angle_step_(angle_discretization_step), dist_step_(distance_discretization_step) {}

  /* trains the device table: the model rows are used as they are (PCL does not resample the model), the distance
   * step is PCL's absolute step expressed relative to the model's bbox diagonal */
  void setInputFeatureCloud(const PPFFeatureCloud::Ptr& features) {
    features->model.refresh();
    const detail::RowsView& mv = features->model;
    if (!mv.p || mv.n < 2) throw ppf_match_3d::Error(PPF_ERR_INVALID, "PPFHashMapSearch: empty feature cloud");
    const float* r = mv.p;
    const int n = mv.n;
    const std::size_t st = (std::size_t)mv.stride;
    float lo[3] = {r[0], r[1], r[2]}, hi[3] = {r[0], r[1], r[2]};
    for (int i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], r[(size_t)i * st + k]); hi[k] = std::max(hi[k], r[(size_t)i * st + k]); }
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    diameter_ = std::sqrt(dx * dx + dy * dy + dz * dz);
    ppf_train_params tp;
    ppf_default_train_params(&tp);
    tp.presampled = 1;
    tp.feature = PPF_FEATURE_DARBOUX; /* PPFEstimation's four values */
    tp.key_equality = PPF_KEY_EXACT; /* nearestNeighborSearch() returns the model pairs with the SAME quantised feature */
    tp.distance_from_distance_step = 1;
    tp.relative_distance_step = dist_step_ / diameter_;
    tp.relative_sampling_step = tp.relative_distance_step; /* only feeds the default clustering threshold */
    tp.num_angles = 2.0 * 3.14159265358979311600 / angle_step_;
    ppf_model* m = nullptr;
    ppf_match_3d::check(ppf_model_train(r, n, mv.stride, mv.noff, &tp, &m)); /* pcl::PointNormal storage as it is: stride 12, normals at 4 */
    model_.reset(m, [](ppf_model* p) { ppf_model_release(p); });
  }
  /* the model pairs (i, j) filed under the quantised (f1, f2, f3, f4): PCL's hash-map lookup with key equality */
  void nearestNeighborSearch(float& f1, float& f2, float& f3, float& f4, std::vector<std::pair<std::size_t, std::size_t>>& indices) {
    if (!model_) throw ppf_match_3d::Error(PPF_ERR_NOT_TRAINED, "PPFHashMapSearch: no feature cloud set");
    const float f[4] = {f1, f2, f3, f4};
    int n = 0;
    ppf_match_3d::check(ppf_model_nearest_pairs(model_.get(), f, nullptr, 0, &n));
    std::vector<uint32_t> ij((std::size_t)std::max(n, 1) * 2);
    ppf_match_3d::check(ppf_model_nearest_pairs(model_.get(), f, ij.data(), std::max(n, 1), &n));
    indices.clear();
    indices.reserve((std::size_t)n);
    for (int q = 0; q < n; q++) indices.push_back(std::make_pair((std::size_t)ij[2 * q], (std::size_t)ij[2 * q + 1]));
  }
  float getAngleDiscretizationStep() const { return angle_step_; }
  float getDistanceDiscretizationStep() const { return dist_step_; }
  float getModelDiameter() const { return diameter_; }
  const ppf_model* handle() const { return model_.get(); }

 private:
  float angle_step_, dist_step_, diameter_ = 0.f;
  std::shared_ptr<ppf_model> model_;
};

template <class PointSource, class PointTarget>
class PPFRegistration {
 public:
  typedef std::array<float, 16> Matrix4f; /* row-major 4x4, model -> scene */
  struct PoseWithVotes {
    Matrix4f pose;
    unsigned votes;
  };

  void setSearchMethod(const PPFHashMapSearch::Ptr& search) { search_ = search; }
  void setSceneReferencePointSamplingRate(unsigned rate) { rate_ = rate ? rate : 1; }
  void setPositionClusteringThreshold(float t) { pos_thr_ = t; }
  void setRotationClusteringThreshold(float t) { rot_thr_ = t; }
  template <class CloudPtr> void setInputSource(const CloudPtr& model) { source_ = detail::view_of(model); } /* voted from the search method's table; align() moves THIS cloud */
  template <class CloudPtr> void setInputTarget(const CloudPtr& scene) { scene_ = detail::view_of(scene); }

  /* computeTransformation(): votes, clusters; keeps every clustered pose, best first */
  template <class CloudT> void align(CloudT& output) {
    if (!search_ || !search_->handle()) throw ppf_match_3d::Error(PPF_ERR_NOT_TRAINED, "PPFRegistration: no trained search method");
    ppf_match_params mp;
    ppf_default_match_params(&mp);
    mp.presampled = 1; /* PCL votes on the target cloud as given */
    mp.relative_scene_sample_step = 1.0 / (double)rate_;
    mp.position_threshold = pos_thr_;
    mp.rotation_threshold = rot_thr_;
    mp.pair_radius = 0.5 * (double)search_->getModelDiameter(); /* the radius search of computeTransformation() */
    mp.rot_metric_relative = 1;                                 /* posesWithinErrorBounds(): angle of the relative rotation */
    mp.alpha_range_2pi = 1;                                     /* alpha wrapped into [-pi, pi], bins of the angle discretisation step */
    scene_.refresh(); /* the target as it is now */
    const int n = scene_.n;
    if (!scene_.p || n <= 0) throw ppf_match_3d::Error(PPF_ERR_INVALID, "PPFRegistration: no target cloud");
    std::vector<ppf_pose> out((size_t)n / rate_ + 8);
    int n_out = 0;
    ppf_match_3d::check(ppf_match(search_->handle(), scene_.p, n, scene_.stride, scene_.noff, nullptr, 0, 6, PPF_NOFF_MAT, &mp, out.data(),
                                  (int)out.size(), &n_out));
    results_.clear();
    for (int i = 0; i < n_out; i++) {
      PoseWithVotes p;
      for (int k = 0; k < 16; k++) p.pose[(size_t)k] = (float)out[(size_t)i].pose[k];
      p.votes = out[(size_t)i].num_votes;
      results_.push_back(p);
    }
    converged_ = n_out > 0;
    /* PCL: `output` = the input source transformed by the final transformation (the source itself when nothing was found) */
    source_.refresh();
    auto& op = detail::pts_mut(output);
    op.resize((std::size_t)std::max(source_.n, 0));
    if (source_.p && source_.n > 0) {
      const double ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      std::vector<float> rows((std::size_t)source_.n * 6);
      ppf_match_3d::check(ppf_transform_pc_pose(source_.p, source_.n, source_.stride, source_.noff, n_out > 0 ? out[0].pose : ident, rows.data()));
      for (int i = 0; i < source_.n; i++) {
        const float* r = &rows[(std::size_t)i * 6];
        op[(std::size_t)i].x = r[0]; op[(std::size_t)i].y = r[1]; op[(std::size_t)i].z = r[2];
        op[(std::size_t)i].normal_x = r[3]; op[(std::size_t)i].normal_y = r[4]; op[(std::size_t)i].normal_z = r[5];
      }
    }
  }
  bool hasConverged() const { return converged_; }
  Matrix4f getFinalTransformation() const {
    if (results_.empty()) return Matrix4f{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    return results_[0].pose;
  }
  const std::vector<PoseWithVotes>& getBestPoseCandidates() const { return results_; }

 private:
  PPFHashMapSearch::Ptr search_;
  unsigned rate_ = 5;
  float pos_thr_ = -1.f, rot_thr_ = -1.f;
  detail::RowsView scene_, source_;
  std::vector<PoseWithVotes> results_;
  bool converged_ = false;
};

}  // namespace pcl_shaped
}  // namespace ppfhip

#endif /* PPF_PCL_HPP */
