/*
 * cloud_stages_demo.cpp — the PCL half of the reference's driver (/root/reference/src/YOLO_cropping_ppf_test.cpp:84-103)
 * on the device stages: crop -> subsample -> outlier removal -> normals -> edges -> the two N x 6 Mats that
 * Matching_S2B consumes, then the match + ICP of examples/cloud_processor_demo.cpp.
 *
 *   usage: cloud_stages_demo model.ply scene_xyz.ply depth.f32 rows cols x y w h fx fy ppx ppy leaf sor_thresh
 *          (depth.f32: raw little-endian float32 image, metres)
 *   build: g++ -std=c++17 -Iinclude examples/cloud_stages_demo.cpp -Lyolo_ppf_pose_estimation_amd/csrc -lppf_hip
 */
#include <cstdio>
#include <fstream>
#include <iostream>

#include "ppf_cloud_stages.hpp"

using namespace std;
using namespace ppfhip;
using namespace ppfhip::ppf_match_3d;

int main(int argc, char** argv) {
  if (argc < 16) {
    cerr << "usage: " << argv[0] << " model.ply scene_xyz.ply depth.f32 rows cols x y w h fx fy ppx ppy leaf sor_thresh" << endl;
    return 1;
  }
  try {
    Mat bottle = loadPLYSimple(argv[1], 1);
    Mat scene = loadPLYSimple(argv[2], 0);
    const int rows = atoi(argv[4]), cols = atoi(argv[5]);
    vector<float> depth((size_t)rows * cols);
    ifstream df(argv[3], ios::binary);
    if (!df.read(reinterpret_cast<char*>(depth.data()), (streamsize)(depth.size() * sizeof(float)))) throw Error(PPF_ERR_IO, "cannot read the depth image");
    const int box[4] = {atoi(argv[6]), atoi(argv[7]), atoi(argv[8]), atoi(argv[9])};
    const double fx = atof(argv[10]), fy = atof(argv[11]), ppx = atof(argv[12]), ppy = atof(argv[13]);
    const double leafsize = atof(argv[14]), outlierremoval_thresh = atof(argv[15]);

    prep::Cloud scene_cloud = prep::Cloud::fromMat(scene);
    prep::Cloud object = scene_cloud.crop(box, depth.data(), rows, cols, fx, fy, ppx, ppy); /* SceneCropping, src:91 */
    printf("Cropping : %d / %d \n", scene_cloud.size(), object.size());
    object = object.voxelGrid(leafsize);                                                   /* Subsampling, src:96 */
    object = object.outlierRemoval(50, outlierremoval_thresh);                             /* OutlierProcessing, src:98 */
    prep::Cloud object_with_normals = object.normals(30);                                  /* NormalEstimation, src:101 */
    prep::Cloud object_edges = object_with_normals.edges(0.03f);                           /* EdgeExtraction, src:103 */
    cout << "object " << object_with_normals.size() << " edges size: " << object_edges.size() << endl;
    Mat object_wn_mat = object_with_normals.toMat(), edges_mat = object_edges.toMat();     /* src:119-120 */

    PPF3DDetector detector(0.05, 0.05);
    detector.trainModel(bottle);
    vector<Pose3DPtr> results;
    detector.match_S2B(object_wn_mat, edges_mat, results, 0.05, 0.05);
    if (results.empty()) { cout << "No matching Poses found. Exiting." << endl; return 0; }
    vector<Pose3DPtr> resultsSub(results.begin(), results.begin() + min<size_t>(5, results.size()));
    ICP icp(100, 0.005f, 2.5f, 8);
    icp.registerModelToScene(bottle, object_wn_mat, resultsSub);
    resultsSub[0]->printPose();
    cout.precision(17);
    cout << "RESULT object=" << object_wn_mat.rows << " edges=" << edges_mat.rows << " votes=" << resultsSub[0]->numVotes
         << " residual=" << resultsSub[0]->residual << endl;
  } catch (const Error& e) {
    cerr << "ppf error " << (int)e.status << ": " << e.what() << endl;
    return 10 + (int)e.status;
  }
  return 0;
}
