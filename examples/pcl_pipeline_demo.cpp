/*
 * pcl_pipeline_demo.cpp — PCL's ppf_object_recognition call sequence on the HIP engine through
 * include/ppf_pcl.hpp (no PCL needed to build).  usage: pcl_pipeline_demo model.ply scene.ply
 */
#include <cmath>
#include <iostream>
#include <utility>
#include <vector>

#include "ppf_pcl.hpp"

using namespace ppfhip;
using namespace ppfhip::pcl_shaped;

static PointCloud<PointNormal>::Ptr load(const char* path) {
  ppf_match_3d::Mat m = ppf_match_3d::loadPLYSimple(path, 1);
  PointCloud<PointNormal>::Ptr c(new PointCloud<PointNormal>());
  c->points.resize((size_t)m.rows);
  for (int i = 0; i < m.rows; i++) {
    const float* r = m.ptr<float>(i);
    c->points[(size_t)i] = PointNormal{r[0], r[1], r[2], r[3], r[4], r[5]};
  }
  return c;
}

int main(int argc, char** argv) {
  if (argc < 3) { std::cerr << "usage: " << argv[0] << " model.ply scene.ply" << std::endl; return 1; }
  try {
    PointCloud<PointNormal>::Ptr cloud_model = load(argv[1]), cloud_scene = load(argv[2]);
    PPFFeatureCloud::Ptr cloud_model_ppf(new PPFFeatureCloud());
    PPFEstimation<PointNormal, PointNormal, PPFSignature> ppf_estimator;
    ppf_estimator.setInputCloud(cloud_model);
    ppf_estimator.setInputNormals(cloud_model);
    cloud_model->points.reserve(cloud_model->points.capacity() * 2 + 8); /* the points move: compute() must read them where they are NOW */
    ppf_estimator.compute(*cloud_model_ppf); /* N x N PPFSignature rows, as PCL materialises them */
    {
      const size_t N = cloud_model->size();
      const PPFSignature& s01 = cloud_model_ppf->points[1]; /* pair (0, 1) */
      std::cout << "FEATURES rows=" << cloud_model_ppf->size() << " n=" << N << " pair01=" << s01.f1 << "," << s01.f2 << "," << s01.f3 << ","
                << s01.f4 << "," << s01.alpha_m << " diag_nan=" << (cloud_model_ppf->points[0].f1 != cloud_model_ppf->points[0].f1) << std::endl;
    }
    PPFHashMapSearch::Ptr hashmap_search(new PPFHashMapSearch(12.0f / 180.0f * 3.14159265f, 0.012f));
    hashmap_search->setInputFeatureCloud(cloud_model_ppf);
    {
      /* the hash map's lookup: the model pairs filed under the quantised feature of pair (0, 1) -- (0, 1) must be among them */
      PPFSignature s01 = cloud_model_ppf->points[1];
      std::vector<std::pair<std::size_t, std::size_t>> pairs;
      hashmap_search->nearestNeighborSearch(s01.f1, s01.f2, s01.f3, s01.f4, pairs);
      bool has01 = false, sorted = true;
      for (std::size_t q = 0; q < pairs.size(); q++) {
        has01 |= pairs[q].first == 0 && pairs[q].second == 1;
        if (q && !(pairs[q - 1] < pairs[q])) sorted = false;
      }
      std::cout << "SEARCH pairs=" << pairs.size() << " has01=" << has01 << " sorted=" << sorted << std::endl;
    }
    PPFRegistration<PointNormal, PointNormal> ppf_registration;
    ppf_registration.setSceneReferencePointSamplingRate(20);
    ppf_registration.setPositionClusteringThreshold(0.05f);
    ppf_registration.setRotationClusteringThreshold(30.0f / 180.0f * 3.14159265f);
    ppf_registration.setSearchMethod(hashmap_search);
    ppf_registration.setInputSource(cloud_model);
    ppf_registration.setInputTarget(cloud_scene);
    PointCloud<PointNormal> cloud_output;
    ppf_registration.align(cloud_output);
    auto T = ppf_registration.getFinalTransformation();
    {
      /* align(output): the source moved by the final transformation */
      const PointNormal& a = cloud_model->points[0];
      const PointNormal& b = cloud_output.points[0];
      const float ex = T[0] * a.x + T[1] * a.y + T[2] * a.z + T[3], ey = T[4] * a.x + T[5] * a.y + T[6] * a.z + T[7],
                  ez = T[8] * a.x + T[9] * a.y + T[10] * a.z + T[11];
      const float err = std::fabs(ex - b.x) + std::fabs(ey - b.y) + std::fabs(ez - b.z);
      std::cout << "OUTPUT rows=" << cloud_output.size() << " moved_ok=" << (err < 1e-5f) << std::endl;
    }
    std::cout << "RESULT converged=" << ppf_registration.hasConverged() << " votes="
              << (ppf_registration.getBestPoseCandidates().empty() ? 0u : ppf_registration.getBestPoseCandidates()[0].votes)
              << " t=" << T[3] << "," << T[7] << "," << T[11] << std::endl;
  } catch (const ppf_match_3d::Error& e) {
    std::cerr << "ppf error " << (int)e.status << ": " << e.what() << std::endl;
    return 10 + (int)e.status;
  }
  return 0;
}
