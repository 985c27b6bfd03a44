/*
 * ppf_cloud_stages.hpp — header-only C++ wrapper of the device-resident cloud stages (ppf_cloud_* / ppf_prep_* in
 * ppf_hip.h): the PCL half of the reference's ppf::CloudProcessor (/root/reference/include/CloudProcessing.h)
 *
 *   SceneCropping      :263-339  ->  Cloud::crop(box, depth, rows, cols, fx, fy, ppx, ppy)
 *   Subsampling        :361-380  ->  Cloud::voxelGrid(leaf)
 *   OutlierProcessing  :341-360  ->  Cloud::outlierRemoval(meanK, stddevMul)
 *   NormalEstimation   :381-405  ->  Cloud::normals(k)
 *   EdgeExtraction     :406-427  ->  Cloud::edges(curvatureThreshold)
 *   PointCloudXYZNormalToMat :163-190 -> Cloud::toMat()   (an N x 6 ppf_match_3d::Mat for PPF3DDetector / ICP)
 *
 * Every stage returns a new Cloud that stays in HBM; only toMat()/download() copy to the host.  A maintainer replaces
 * the bodies of those CloudProcessor methods by these one-liners; pcl::PointCloud<PointXYZ> goes
 * in through Cloud::fromXYZ(&cloud.points[0].x, cloud.size(), 4) (PointXYZ is 4 floats wide).
 */
#ifndef PPF_CLOUD_STAGES_HPP
#define PPF_CLOUD_STAGES_HPP

#include <memory>
#include <vector>

#include "ppf_match_3d.hpp"

namespace ppfhip {
namespace prep {

class Cloud {
 public:
  Cloud() {}
  /* rows of `cols` (3 or 6) floats, `strideFloats` apart */
  static Cloud fromRows(const float* rows, int n, int strideFloats, int cols) {
    ppf_cloud* c = nullptr;
    ppf_match_3d::check(ppf_cloud_upload(rows, n, strideFloats, PPF_NOFF_MAT, cols, &c));
    return Cloud(c);
  }
  static Cloud fromXYZ(const float* xyz, int n, int strideFloats = 3) { return fromRows(xyz, n, strideFloats, 3); }
  /* an N x 3 or N x 6 float32 Mat (cv::Mat when OpenCV is present: rows read through step1()) */
  static Cloud fromMat(const ppf_match_3d::Mat& m) {
    return fromRows(m.ptr<float>(0), m.rows, ppf_match_3d::detail::stride_of(m), m.cols >= 6 ? 6 : 3);
  }

  bool empty() const { return size() == 0; }
  int size() const {
    if (!h_) return 0;
    int n = 0;
    ppf_match_3d::check(ppf_cloud_size(h_.get(), &n));
    return n;
  }
  const ppf_cloud* handle() const { return h_.get(); }

  /* box = {x, y, width, height} of the detection; depth: host image rows x cols, metres */
  Cloud crop(const int box[4], const float* depth, int depthRows, int depthCols, double fx, double fy, double ppx, double ppy) const {
    const double intr[4] = {fx, fy, ppx, ppy};
    ppf_cloud* o = nullptr;
    ppf_match_3d::check(ppf_prep_crop(need(), box, depth, depthRows, depthCols, intr, &o));
    return Cloud(o);
  }
  Cloud voxelGrid(double leaf) const { ppf_cloud* o = nullptr; ppf_match_3d::check(ppf_prep_voxel_grid(need(), leaf, &o)); return Cloud(o); }
  Cloud outlierRemoval(int meanK = 50, double stddevMul = 1.5) const {
    ppf_cloud* o = nullptr;
    ppf_match_3d::check(ppf_prep_outlier_removal(need(), meanK, stddevMul, &o));
    return Cloud(o);
  }
  Cloud normals(int k = 30) const { ppf_cloud* o = nullptr; ppf_match_3d::check(ppf_prep_normals(need(), k, &o)); return Cloud(o); }
  Cloud edges(float curvatureThreshold) const { ppf_cloud* o = nullptr; ppf_match_3d::check(ppf_prep_edges(need(), curvatureThreshold, &o)); return Cloud(o); }

  /* the N x 6 CV_32FC1-shaped Mat of PointCloudXYZNormalToMat (normals re-normalised) */
  ppf_match_3d::Mat toMat() const {
    ppf_cloud* o = nullptr;
    ppf_match_3d::check(ppf_prep_to_mat(need(), &o));
    Cloud tmp(o);
    const int n = tmp.size();
    ppf_match_3d::Mat m = ppf_match_3d::detail::new_cloud(n, 6);
    if (n) ppf_match_3d::check(ppf_cloud_download(tmp.handle(), m.ptr<float>(0), nullptr, n));
    return m;
  }
  /* rows (n x 6) and curvature (n) as they are on the device */
  void download(std::vector<float>& rows6, std::vector<float>& curvature) const {
    const int n = size();
    rows6.assign((size_t)n * 6, 0.f);
    curvature.assign((size_t)n, 0.f);
    if (n) ppf_match_3d::check(ppf_cloud_download(h_.get(), rows6.data(), curvature.data(), n));
  }

 private:
  explicit Cloud(ppf_cloud* c) : h_(c, [](ppf_cloud* p) { ppf_cloud_release(p); }) {}
  const ppf_cloud* need() const {
    if (!h_) throw ppf_match_3d::Error(PPF_ERR_INVALID, "prep::Cloud: empty handle");
    return h_.get();
  }
  std::shared_ptr<ppf_cloud> h_;
};

}  // namespace prep
}  // namespace ppfhip

#endif /* PPF_CLOUD_STAGES_HPP */
