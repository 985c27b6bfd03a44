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
 * std::vector of such points.  Compiles without PCL and without Eigen.
 */
#ifndef PPF_PCL_HPP
#define PPF_PCL_HPP

#include <array>
#include <cmath>
#include <memory>
#include <vector>

#include "ppf_match_3d.hpp"

namespace ppfhip {
namespace pcl_shaped {

struct PointNormal { /* layout-compatible subset of pcl::PointNormal for builds without PCL */
  float x, y, z, normal_x, normal_y, normal_z;
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

/* N x 6 row-major float rows from any cloud of PointNormal-like points (what PointCloudXYZNormalToMat does,
 * CloudProcessing.h:163-190, including the re-normalisation of the normals) */
template <class CloudT>
std::vector<float> to_rows(const CloudT& cloud) {
  const auto& p = pts(cloud);
  std::vector<float> rows(p.size() * 6);
  for (std::size_t i = 0; i < p.size(); i++) {
    float* d = &rows[i * 6];
    d[0] = p[i].x; d[1] = p[i].y; d[2] = p[i].z;
    d[3] = p[i].normal_x; d[4] = p[i].normal_y; d[5] = p[i].normal_z;
    const double a = std::sqrt((double)d[3] * d[3] + (double)d[4] * d[4] + (double)d[5] * d[5]);
    if (a > 0.00001) { d[3] /= (float)a; d[4] /= (float)a; d[5] /= (float)a; }
  }
  return rows;
}
}  // namespace detail

/* The "feature cloud" of the PCL pipeline.  PCL materialises N^2 PPFSignature rows here; the engine builds the
 * pair features on the device while training, so this object only carries the model rows to PPFHashMapSearch. */
struct PPFFeatureCloud {
  std::vector<float> model_rows; /* N x 6 */
  typedef std::shared_ptr<PPFFeatureCloud> Ptr;
};

template <class PointInT, class PointNT, class PointOutT = PPFSignature>
class PPFEstimation {
 public:
  template <class CloudPtr> void setInputCloud(const CloudPtr& cloud) { rows_ = detail::to_rows(*cloud); }
  template <class CloudPtr> void setInputNormals(const CloudPtr&) {} /* normals travel with the points */
  void compute(PPFFeatureCloud& out) { out.model_rows = rows_; }

 private:
  std::vector<float> rows_;
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
    const std::vector<float>& r = features->model_rows;
    const int n = (int)(r.size() / 6);
    float lo[3] = {r[0], r[1], r[2]}, hi[3] = {r[0], r[1], r[2]};
    for (int i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], r[(size_t)i * 6 + k]); hi[k] = std::max(hi[k], r[(size_t)i * 6 + k]); }
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
    ppf_match_3d::check(ppf_model_train(r.data(), n, 6, &tp, &m));
    model_.reset(m, [](ppf_model* p) { ppf_model_release(p); });
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
  template <class CloudPtr> void setInputSource(const CloudPtr&) {} /* the model lives in the search method */
  template <class CloudPtr> void setInputTarget(const CloudPtr& scene) { scene_rows_ = detail::to_rows(*scene); }

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
    const int n = (int)(scene_rows_.size() / 6);
    std::vector<ppf_pose> out((size_t)n / rate_ + 8);
    int n_out = 0;
    ppf_match_3d::check(ppf_match(search_->handle(), scene_rows_.data(), n, 6, nullptr, 0, 6, &mp, out.data(), (int)out.size(), &n_out));
    results_.clear();
    for (int i = 0; i < n_out; i++) {
      PoseWithVotes p;
      for (int k = 0; k < 16; k++) p.pose[(size_t)k] = (float)out[(size_t)i].pose[k];
      p.votes = out[(size_t)i].num_votes;
      results_.push_back(p);
    }
    converged_ = n_out > 0;
    (void)output; /* PCL fills `output` with the transformed source; left to the caller (ppf_transform_pc_pose) */
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
  std::vector<float> scene_rows_;
  std::vector<PoseWithVotes> results_;
  bool converged_ = false;
};

}  // namespace pcl_shaped
}  // namespace ppfhip

#endif /* PPF_PCL_HPP */
