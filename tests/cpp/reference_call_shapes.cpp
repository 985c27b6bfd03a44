/*
 * reference_call_shapes.cpp — TEST translation unit (builder-written; not the reference's file, not a copy of it).
 *
 * Proves that the call shapes of the reference's PPF half compile and run against include/ppf_match_3d.hpp in the
 * reference's situation: OpenCV present, file-scope using directives, and ONLY UNQUALIFIED names for
 * Mat / vector<Mat> / Pose3D / Pose3DPtr / PPF3DDetector / ICP / FileStorage and the three PLY / transform helpers.
 * Shapes covered, with the reference lines that have them:
 *   using directives at file scope                         Camera.h:10, CloudProcessing.h:28-32, src/YOLO_cropping_ppf_test.cpp:6-9
 *   vector<Mat> members, Mat used for an image too         CloudProcessing.h:44-48
 *   vector<PPF3DDetector>, explicit ~PPF3DDetector()       :58, :79-83
 *   FileStorage READ + detector.read(fs.root())            :106-121
 *   Mat(rows, 6, CV_32FC1) + ptr<float>(i) row fill        :163-190
 *   ppf_match_3d::PPF3DDetector d(a, b); push_back(d)      :205-206, :217-218
 *   trainModel, detectors[i] = d, FileStorage WRITE        :222-261
 *   by-value detector copy, CV_Error when untrained,
 *   match / match_S2B, top-5, ICP(100, 0.005f, 2.5f, 8),
 *   registerModelToScene, return *resultsSub[0]            :428-533
 *   loadPLYSimple, printPose, transformPCPose(m, p.pose),
 *   writePLY                                               src/YOLO_cropping_ppf_test.cpp:113-127
 * Built by tests/test_cpp_facade.py with g++ -std=c++11 -Wall -Wextra -Werror against tests/mock_opencv, in two
 * configurations:
 *   -DSHAPES_RECIPE=1  include swap only: PPF_MATCH_3D_AS_CV makes the namespace visible as cv::ppf_match_3d, the using
 *                      directives are the reference's own, untouched
 *   -DSHAPES_RECIPE=2  include swap + `using namespace ppfhip; using namespace ppfhip::ppf_match_3d;`
 *
 *   usage: reference_call_shapes model.ply scene.ply edge.ply|- detector_dir out.ply [train_step]
 */
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include <opencv2/core.hpp>
using namespace cv; /* in effect before the PPF header's own directives, as Camera.h:10 leaves it */

#if SHAPES_RECIPE == 1
#define PPF_MATCH_3D_AS_CV
#include "ppf_match_3d.hpp"
using namespace ppf_match_3d; /* = cv::ppf_match_3d, found through the directive above */
using namespace cv;
using namespace std;
#elif SHAPES_RECIPE == 2
#include "ppf_match_3d.hpp"
using namespace ppfhip;
using namespace ppfhip::ppf_match_3d;
using namespace cv;
using namespace std;
#else
#error "build with -DSHAPES_RECIPE=1 or 2"
#endif

/* a stand-in for pcl::PointNormal storage: what the Mat adaptor reads its rows from */
struct SurfacePoint {
  float x, y, z, normal_x, normal_y, normal_z;
};

class ObjectMatcher {
 private:
  vector<Mat> final_objects;
  Mat depth; /* an IMAGE under the same unqualified name */
  vector<Mat> models;
  map<string, int> label_to_id;
  map<int, string> id_to_label;
  vector<bool> if_trained;
  double relativeSamplingStep, relativeDistanceStep;

 public:
  vector<PPF3DDetector> detectors;

  ObjectMatcher(Mat depth_input, double sampling = 0.025, double distance = 0.05)
      : depth(depth_input), relativeSamplingStep(sampling), relativeDistanceStep(distance) {}
  ~ObjectMatcher() {
    for (uint32_t i = 0; i < detectors.size(); i++) detectors[i].~PPF3DDetector(); /* then once more by the vector */
  }

  void AddModel(Mat model_input, string label) {
    models.push_back(model_input);
    int id = (int)models.size() - 1;
    if_trained.push_back(false);
    label_to_id[label] = id;
    id_to_label[id] = label;
    ppf_match_3d::PPF3DDetector detector(relativeSamplingStep, relativeDistanceStep);
    detectors.push_back(detector);
  }

  void TrainAll(const string dir, bool saveflag, const double sampling_train, const double distance_train) {
    if (models.size() == 0) {
      CV_Error(1, "No 3D model loaded");
      return;
    }
    for (uint32_t i = 0; i < models.size(); i++) {
      ppf_match_3d::PPF3DDetector detector(sampling_train, distance_train);
      int64 tick1 = getTickCount();
      detector.trainModel(models[i]);
      int64 tick2 = getTickCount();
      cout << "trained model " << (i + 1) << " in " << (double)(tick2 - tick1) / getTickFrequency() << " sec" << endl;
      detectors[i] = detector;
      if_trained[i] = true;
      if (saveflag) {
        string FileName = dir + "detector_" + id_to_label[i] + ".xml";
        FileStorage fsOut(FileName, FileStorage::WRITE);
        detector.write(fsOut);
        fsOut.release();
      }
    }
  }

  void LoadTrained(const string name, const string file) {
    int id = label_to_id[name];
    FileStorage fsload(file, FileStorage::READ);
    detectors[id].read(fsload.root());
    fsload.release();
    if_trained[id] = true;
  }

  void RowsToMat(Mat& cv_cloud, const vector<SurfacePoint>& pts) {
    uint32_t rows = (uint32_t)pts.size();
    uint32_t cols = 6;
    cv_cloud = Mat(rows, cols, CV_32FC1);
    for (uint32_t i = 0; i < rows; i++) {
      float* data = cv_cloud.ptr<float>(i);
      data[0] = pts[i].x; data[1] = pts[i].y; data[2] = pts[i].z;
      data[3] = pts[i].normal_x; data[4] = pts[i].normal_y; data[5] = pts[i].normal_z;
    }
    final_objects.push_back(cv_cloud);
  }

  Pose3D Match(const string name, Mat scene, Mat edge, double relativeSceneSampleStep = 0.05, double relativeSceneDistance = 0.05) {
    int id = label_to_id[name];
    PPF3DDetector detector = detectors[id]; /* by value */
    vector<Pose3DPtr> results;
    if (!if_trained[id]) {
      CV_Error(2, "Model [" + name + "] not trained yet.");
      exit(1);
    }
    int64 tick1 = cv::getTickCount();
    if (edge.empty()) detector.match(scene, results, relativeSceneSampleStep, relativeSceneDistance);
    else detector.match_S2B(scene, edge, results, relativeSceneSampleStep, relativeSceneDistance);
    int64 tick2 = cv::getTickCount();
    cout << "PPF Elapsed Time " << (tick2 - tick1) / cv::getTickFrequency() << " sec" << endl;
    size_t results_size = results.size();
    if (results_size == 0) {
      cout << "No matching Poses found. Exiting." << endl;
      exit(0);
    }
    size_t N = 5;
    if (results_size < N) N = results_size;
    vector<Pose3DPtr> resultsSub(results.begin(), results.begin() + N);
    ICP icp(100, 0.005f, 2.5f, 8);
    icp.registerModelToScene(models[id], scene, resultsSub);
    return *resultsSub[0];
  }

  float DepthAt(int r, int c) const { return depth.at<float>(r, c); }
};

static vector<SurfacePoint> rows_of(const Mat& m) {
  vector<SurfacePoint> pts((size_t)m.rows);
  for (int i = 0; i < m.rows; i++) {
    const float* r = m.ptr<float>(i);
    SurfacePoint p = {r[0], r[1], r[2], r[3], r[4], r[5]};
    pts[(size_t)i] = p;
  }
  return pts;
}

int main(int argc, char** argv) {
  if (argc < 6) {
    cerr << "usage: " << argv[0] << " model.ply scene.ply edge.ply|- detector_dir out.ply [train_step]" << endl;
    return 2;
  }
  try {
    const double train_step = argc > 6 ? atof(argv[6]) : 0.05;
    Mat depth(4, 4, CV_32FC1);
    depth.at<float>(1, 2) = 0.635f;
    string bottle_file = argv[1];
    Mat bottle = ppf_match_3d::loadPLYSimple(bottle_file.c_str(), 1);
    Mat scene_rows = loadPLYSimple(argv[2], 1);

    ObjectMatcher matcher(depth, train_step, 0.05);
    matcher.AddModel(bottle, "bottle");
    matcher.AddModel(bottle, "never_trained");
    int refused = 0;
    try {
      matcher.Match("never_trained", scene_rows, Mat());
    } catch (const cv::Exception& e) {
      refused = e.code == 2;
    }
    /* train both slots (the reference trains every model it holds), keep slot 0's file, then forget slot 0's table and
     * bring it back from the file */
    matcher.TrainAll(string(argv[4]) + "/", true, train_step, 0.05);
    const string file = string(argv[4]) + "/detector_bottle.xml";
    matcher.detectors[0] = PPF3DDetector(train_step, 0.05);
    matcher.LoadTrained("bottle", file);

    Mat object_wn_mat, edges_mat;
    matcher.RowsToMat(object_wn_mat, rows_of(scene_rows));
    if (string(argv[3]) != "-") matcher.RowsToMat(edges_mat, rows_of(loadPLYSimple(argv[3], 1)));
    Pose3D result_pose = matcher.Match("bottle", object_wn_mat, edges_mat);
    Pose3D trained_pose = matcher.Match("never_trained", object_wn_mat, edges_mat); /* slot 1: the table as trained, never saved */
    cout << "Result Pose: " << endl;
    result_pose.printPose();
    Mat object_trans = transformPCPose(bottle, result_pose.pose);
    writePLY(object_trans, argv[5]);
    const bool same = result_pose.numVotes == trained_pose.numVotes && result_pose.pose == trained_pose.pose &&
                      result_pose.residual == trained_pose.residual;
    cout.precision(17);
    cout << "RESULT votes=" << result_pose.numVotes << " residual=" << result_pose.residual << " t0=" << result_pose.t[0]
         << " q0=" << result_pose.q[0] << " p03=" << result_pose.pose(0, 3) << " loaded_equals_trained=" << (int)same
         << " untrained_refused=" << refused << " depth=" << matcher.DepthAt(1, 2) << " moved_rows=" << object_trans.rows << endl;
    return same && refused ? 0 : 1;
  } catch (const ppfhip::ppf_match_3d::Error& e) {
    cerr << "ppf error " << (int)e.status << ": " << e.what() << endl;
    return 10 + (int)e.status;
  }
}
