/*
 * ppf_match_3d.hpp — header-only C++ facade over the C-ABI (ppf_hip.h), shaped like the part of
 * OpenCV's cv::ppf_match_3d that /root/reference/include/CloudProcessing.h and
 * /root/reference/src/YOLO_cropping_ppf_test.cpp use:
 *
 *     PPF3DDetector(relSampling, relDistance)          CloudProcessing.h:205,217,234
 *     detector.trainModel(Mat)                         :236
 *     detector.match(scene, results, step, dist)       :442
 *     detector.match_S2B(scene, edge, results, ...)    :495
 *     detector.read(...) / detector.write(...)         :112 / :250
 *     by-value copies, explicit ~PPF3DDetector()       :81,206,218,240,432,485
 *     Pose3D (.pose, printPose), Pose3DPtr             src:122-125
 *     loadPLYSimple / transformPCPose / writePLY       src:114,125,127
 *
 * With <opencv2/core.hpp> available (the reference's situation) the facade introduces NO second matrix type:
 * ppf_match_3d::Mat, Matx44d, Vec3d, Vec4d ARE cv::Mat, cv::Matx44d, cv::Vec3d, cv::Vec4d (using-declarations) and
 * Pose3DPtr is cv::Ptr<Pose3D>, so the reference's file-scope `using namespace cv; using namespace ppf_match_3d;`
 * (Camera.h:10, CloudProcessing.h:28-29) and its unqualified `Mat` (clouds AND images) stay unambiguous and
 * loadPLYSimple / transformPCPose / writePLY / samplePCByQuantization take and return cv::Mat (N x 6 or N x 3 CV_32FC1,
 * rows read through step1()).  -DPPF_MATCH_3D_AS_CV additionally makes the namespace visible as cv::ppf_match_3d, the
 * name the reference spells (tests/cpp/reference_call_shapes.cpp compiles both ways, C++11, -Werror).
 * WITHOUT OpenCV (this build container) the same names are small stand-alone types with the same members
 * (rows / cols / ptr<float>(i) / at<float>(i, j) / empty / clone; val[16] / operator()(i, j) / eye()).
 *
 * Failures of the C-ABI become exceptions (ppf_match_3d::Error carries the status and the
 * library's message), which is how CV_Error / CV_Assert surface in the reference's library.
 * The detector is a ref-counted handle: copies share the device table, destruction (even the
 * reference's explicit double destruction) is safe.
 */
#ifndef PPF_MATCH_3D_HPP
#define PPF_MATCH_3D_HPP

#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <climits>

#include "ppf_hip.h"

#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>) && !defined(PPF_MATCH_3D_NO_OPENCV)
#include <opencv2/core.hpp>
#define PPF_MATCH_3D_HAVE_OPENCV 1
#endif
#endif

namespace ppfhip {
namespace ppf_match_3d {

class Error : public std::runtime_error {
 public:
  Error(ppf_status st, const std::string& msg) : std::runtime_error(msg), status(st) {}
  ppf_status status;
};

inline void check(ppf_status st) {
  if (st == PPF_OK) return;
  char buf[512];
  ppf_last_error(buf, (int)sizeof(buf));
  throw Error(st, buf);
}

#ifdef PPF_MATCH_3D_HAVE_OPENCV
/* the reference's types themselves: a using-declaration names the same entity, so `using namespace cv;` next to
 * `using namespace ppf_match_3d;` never makes `Mat` ambiguous */
using cv::Mat;
using cv::Matx44d;
using cv::Vec3d;
using cv::Vec4d;
#else
/* Minimal float32 matrix with the accessors the reference uses on cv::Mat for point clouds. */
class Mat {
 public:
  Mat() : rows(0), cols(0) {}
  Mat(int r, int c) : rows(r), cols(c), data_(new std::vector<float>((size_t)r * c, 0.f)) {}
  Mat(int r, int c, const float* src) : rows(r), cols(c), data_(new std::vector<float>(src, src + (size_t)r * c)) {}
  int rows, cols;
  bool empty() const { return rows == 0 || !data_; }
  template <class T> T* ptr(int i = 0) { return reinterpret_cast<T*>(data_->data() + (size_t)i * cols); }
  template <class T> const T* ptr(int i = 0) const { return reinterpret_cast<const T*>(data_->data() + (size_t)i * cols); }
  template <class T> T& at(int i, int j) { return reinterpret_cast<T*>(data_->data() + (size_t)i * cols)[j]; }
  template <class T> const T& at(int i, int j) const { return reinterpret_cast<const T*>(data_->data() + (size_t)i * cols)[j]; }
  Mat clone() const { return empty() ? Mat() : Mat(rows, cols, data_->data()); }

 private:
  std::shared_ptr<std::vector<float>> data_;
};

/* row-major fixed-size matrices with cv::Matx's members: val[], operator()(i, j), eye(), == */
template <int M, int N> struct MatxD {
  double val[M * N];
  MatxD() { for (int k = 0; k < M * N; k++) val[k] = 0; }
  static MatxD eye() { MatxD m; for (int k = 0; k < (M < N ? M : N); k++) m.val[k * N + k] = 1; return m; }
  double& operator()(int i, int j) { return val[i * N + j]; }
  const double& operator()(int i, int j) const { return val[i * N + j]; }
  double& operator[](int i) { return val[i]; } /* cv::Vec */
  const double& operator[](int i) const { return val[i]; }
  bool operator==(const MatxD& o) const { for (int k = 0; k < M * N; k++) if (val[k] != o.val[k]) return false; return true; }
  bool operator!=(const MatxD& o) const { return !(*this == o); }
};
typedef MatxD<4, 4> Matx44d;
typedef MatxD<3, 1> Vec3d;
typedef MatxD<4, 1> Vec4d;
#endif

namespace detail {
/* row pitch in floats: cols for the stand-alone Mat; step1() (and a CV_32FC1 check) for cv::Mat, so that a
 * non-continuous view (a column range, a row stride) is read correctly and a double Mat is refused */
#ifdef PPF_MATCH_3D_HAVE_OPENCV
inline int stride_of(const Mat& m) {
  if (m.depth() != CV_32F || m.channels() != 1) throw Error(PPF_ERR_INVALID, "expected a CV_32FC1 cloud (N x 6 float32)");
  return (int)m.step1();
}
inline Mat new_cloud(int rows, int cols) { return Mat(rows, cols, CV_32FC1); }
#else
inline int stride_of(const Mat& m) { return m.cols; }
inline Mat new_cloud(int rows, int cols) { return Mat(rows, cols); }
#endif
}  // namespace detail

/* cv::ppf_match_3d::Pose3D: the fields and methods of the library's class (pose_3d.hpp), filled from the engine's ppf_pose */
class Pose3D;
#ifdef PPF_MATCH_3D_HAVE_OPENCV
typedef cv::Ptr<Pose3D> Pose3DPtr;
#else
typedef std::shared_ptr<Pose3D> Pose3DPtr; /* cv::Ptr<T> derives from std::shared_ptr<T> in OpenCV 4 */
#endif

class Pose3D {
 public:
  Pose3D() : alpha(0), residual(0), modelIndex(0), numVotes(0), pose(Matx44d::eye()), angle(0) { q[0] = 1; }
  Pose3D(double Alpha, size_t ModelIndex = 0, size_t NumVotes = 0)
      : alpha(Alpha), residual(0), modelIndex(ModelIndex), numVotes(NumVotes), pose(Matx44d::eye()), angle(0) { q[0] = 1; }
  explicit Pose3D(const ppf_pose& p) : alpha(p.alpha), residual(p.residual), modelIndex(p.model_index),
                                       numVotes(p.num_votes), angle(p.angle) {
    std::memcpy(pose.val, p.pose, sizeof(p.pose));
    std::memcpy(t.val, p.t, sizeof(p.t));
    std::memcpy(q.val, p.q, sizeof(p.q));
  }
  /* the engine's record of this pose (what ppf_icp_refine and ppf_cluster_poses take) */
  ppf_pose record() const {
    ppf_pose r;
    std::memset(&r, 0, sizeof(r));
    std::memcpy(r.pose, pose.val, sizeof(r.pose));
    std::memcpy(r.q, q.val, sizeof(r.q));
    std::memcpy(r.t, t.val, sizeof(r.t));
    r.angle = angle; r.alpha = alpha; r.residual = residual;
    r.model_index = (uint32_t)modelIndex; r.num_votes = (uint32_t)numVotes;
    return r;
  }
  void printPose() const {
    std::printf("\n-- Pose to Model Index %u: NumVotes = %u, Residual = %f\n", (unsigned)modelIndex, (unsigned)numVotes, residual);
    for (int i = 0; i < 4; i++) std::printf("[%g, %g, %g, %g]\n", pose(i, 0), pose(i, 1), pose(i, 2), pose(i, 3));
  }
  Pose3DPtr clone() const { return Pose3DPtr(new Pose3D(*this)); }

  double alpha, residual;
  size_t modelIndex, numVotes;
  Matx44d pose;
  double angle;
  Vec3d t;
  Vec4d q;
};

class PPF3DDetector {
 public:
  PPF3DDetector() : PPF3DDetector(0.05, 0.05, 30) {}
  PPF3DDetector(double relativeSamplingStep, double relativeDistanceStep = 0.05, double numAngles = 30) : model_(nullptr) {
    ppf_default_train_params(&tp_);
    tp_.relative_sampling_step = relativeSamplingStep;
    tp_.relative_distance_step = relativeDistanceStep;
    tp_.num_angles = numAngles;
    ppf_default_match_params(&mp_);
  }
  PPF3DDetector(const PPF3DDetector& o) : tp_(o.tp_), mp_(o.mp_), model_(o.model_) {
    if (model_) ppf_model_retain(model_);
  }
  PPF3DDetector& operator=(const PPF3DDetector& o) {
    if (this != &o) {
      if (o.model_) ppf_model_retain(o.model_);
      if (model_) ppf_model_release(model_);
      tp_ = o.tp_; mp_ = o.mp_; model_ = o.model_;
    }
    return *this;
  }
  /* idempotent: the reference destroys its detectors explicitly and then again through the vector */
  virtual ~PPF3DDetector() {
    if (model_) { ppf_model_release(model_); model_ = nullptr; }
  }

  void setSearchParams(double positionThreshold = -1, double rotationThreshold = -1, bool useWeightedClustering = false) {
    mp_.position_threshold = positionThreshold;
    mp_.rotation_threshold = rotationThreshold;
    mp_.use_weighted_avg = useWeightedClustering ? 1 : 0;
  }
  bool isTrained() const { return model_ != nullptr; }

  /* raw rows: x y z first, the normal `normalOffset` floats into each row of `strideFloats` floats (PPF_NOFF_MAT for the
   * N x 6 Mat, PPF_NOFF_PCL with stride 12 for pcl::PointNormal storage) */
  void trainModel(const float* xyzn, int rows, int strideFloats, int normalOffset = PPF_NOFF_MAT) {
    ppf_model* m = nullptr;
    check(ppf_model_train(xyzn, rows, strideFloats, normalOffset, &tp_, &m));
    if (model_) ppf_model_release(model_);
    model_ = m;
  }
  void match(const float* scene, int rows, int stride, std::vector<Pose3DPtr>& results, double relativeSceneSampleStep = 1.0 / 5.0,
             double relativeSceneDistance = 0.03, int normalOffset = PPF_NOFF_MAT) {
    run(scene, rows, stride, normalOffset, nullptr, 0, 6, PPF_NOFF_MAT, results, relativeSceneSampleStep, relativeSceneDistance);
  }
  void match_S2B(const float* scene, int rows, int stride, const float* edge, int erows, int estride,
                 std::vector<Pose3DPtr>& results, double relativeSceneSampleStep = 0.05, double relativeSceneDistance = 0.05,
                 int normalOffset = PPF_NOFF_MAT, int edgeNormalOffset = PPF_NOFF_MAT) {
    run(scene, rows, stride, normalOffset, edge, erows, estride, edgeNormalOffset, results, relativeSceneSampleStep, relativeSceneDistance);
  }

  /* the calls the reference makes (CloudProcessing.h:236, 442, 495): N x 6 CV_32FC1 clouds, rows read through step1() */
  void trainModel(const Mat& Model) { require_cloud(Model, "trainModel"); trainModel(Model.ptr<float>(0), Model.rows, detail::stride_of(Model)); }
  void match(const Mat& scene, std::vector<Pose3DPtr>& results, const double relativeSceneSampleStep = 1.0 / 5.0,
             const double relativeSceneDistance = 0.03) {
    require_cloud(scene, "match");
    match(scene.ptr<float>(0), scene.rows, detail::stride_of(scene), results, relativeSceneSampleStep, relativeSceneDistance);
  }
  void match_S2B(const Mat& scene, const Mat& edge, std::vector<Pose3DPtr>& results, const double relativeSceneSampleStep = 0.05,
                 const double relativeSceneDistance = 0.05) {
    require_cloud(scene, "match_S2B"); require_cloud(edge, "match_S2B");
    match_S2B(scene.ptr<float>(0), scene.rows, detail::stride_of(scene), edge.ptr<float>(0), edge.rows, detail::stride_of(edge), results,
              relativeSceneSampleStep, relativeSceneDistance);
  }

  /* (de)serialisation: the reference's FileStorage XML format comes from a private OpenCV patch and is
   * unknown; the engine keeps a versioned binary table file instead */
  void write(const std::string& path) const { require_trained(); check(ppf_model_save(model_, path.c_str())); }
  void read(const std::string& path) {
    ppf_model* m = nullptr;
    check(ppf_model_load(path.c_str(), &m));
    if (model_) ppf_model_release(model_);
    model_ = m;
    ppf_model_info info;
    check(ppf_model_get_info(model_, &info));
  }
#ifdef PPF_MATCH_3D_HAVE_OPENCV
  /* The reference's own calls (CloudProcessing.h:111-113, 249-251):
   *     cv::FileStorage fsOut(file, cv::FileStorage::WRITE); detector.write(fsOut); fsOut.release();
   *     cv::FileStorage fsLoad(file, cv::FileStorage::READ); cv::FileNode fn = fsLoad.root(); detector.read(fn);
   * The table travels inside the storage as one CV_8U row under the key "ppf_hip_model" (the bytes of the binary table
   * file); read() validates them like any other model file. */
  void write(cv::FileStorage& fs) const {
    require_trained();
    size_t size = 0;
    check(ppf_model_save_mem(model_, nullptr, 0, &size));
    if (size == 0 || size > (size_t)INT_MAX) throw Error(PPF_ERR_IO, "write(FileStorage): the table does not fit one cv::Mat row (> 2 GiB); use write(path)");
    std::vector<unsigned char> bytes(size);
    check(ppf_model_save_mem(model_, bytes.data(), bytes.size(), &size));
    cv::Mat blob(1, (int)size, CV_8U, bytes.data());
    fs << "ppf_hip_model" << blob;
  }
  void read(const cv::FileNode& fn) {
    cv::Mat blob;
    fn["ppf_hip_model"] >> blob;
    if (blob.empty() || blob.depth() != CV_8U) throw Error(PPF_ERR_IO, "read(FileNode): no ppf_hip_model entry in this storage");
    std::vector<unsigned char> bytes;
    bytes.reserve((size_t)blob.rows * (size_t)blob.cols);
    for (int r = 0; r < blob.rows; r++) bytes.insert(bytes.end(), blob.ptr<unsigned char>(r), blob.ptr<unsigned char>(r) + blob.cols);
    ppf_model* m = nullptr;
    check(ppf_model_load_mem(bytes.data(), bytes.size(), &m)); /* validated field by field like any model file */
    if (model_) ppf_model_release(model_);
    model_ = m;
  }
#endif
  ppf_model_info info() const { require_trained(); ppf_model_info i; check(ppf_model_get_info(model_, &i)); return i; }
  const ppf_model* handle() const { return model_; }

 private:
  static void require_cloud(const Mat& m, const char* who) {
    if (m.rows <= 0 || m.cols < 6) throw Error(PPF_ERR_INVALID, std::string(who) + ": expected an N x 6 float32 cloud (x y z nx ny nz)");
  }
  void require_trained() const {
    if (!model_) throw Error(PPF_ERR_NOT_TRAINED, "The model is not trained. Cannot match without training");
  }
  void run(const float* scene, int rows, int stride, int noff, const float* edge, int erows, int estride, int enoff,
           std::vector<Pose3DPtr>& results, double step, double dist) {
    require_trained();
    ppf_match_params p = mp_;
    p.relative_scene_sample_step = step;
    p.relative_scene_distance = dist;
    /* one clustered pose per voted reference point at most: every (int)(1/step)-th (sampled) scene row */
    const int every = step > 0 && step <= 1 ? (int)(1.0 / step) : 1;
    int cap = rows / (every > 0 ? every : 1) + 8, n = 0;
    std::vector<ppf_pose> out((size_t)cap);
    check(ppf_match(model_, scene, rows, stride, noff, edge, erows, estride, enoff, &p, out.data(), cap, &n));
    results.clear();
    results.reserve((size_t)n);
    for (int i = 0; i < n; i++) results.push_back(Pose3DPtr(new Pose3D(out[(size_t)i])));
  }

  ppf_train_params tp_;
  ppf_match_params mp_;
  ppf_model* model_;
};

/* cv::ppf_match_3d::ICP as the reference constructs and calls it (CloudProcessing.h:465-470, :518-523):
 *   ICP icp(100, 0.005f, 2.5f, 8);
 *   icp.registerModelToScene(models[id], scene, resultsSub);      // refines every pose in place
 * Uniform sampling and one correspondence per point (the library defaults, the only mode the reference uses). */
class ICP {
 public:
  enum { ICP_SAMPLING_TYPE_UNIFORM = 0, ICP_SAMPLING_TYPE_GELFAND = 1 };
  ICP() : ICP(250, 0.05f, 2.5f, 6) {} /* the library's default constructor values */
  ICP(const int iterations, const float tolerence = 0.05f, const float rejectionScale = 2.5f, const int numLevels = 6,
      const int sampleType = ICP_SAMPLING_TYPE_UNIFORM, const int numMaxCorr = 1) {
    if (sampleType != ICP_SAMPLING_TYPE_UNIFORM || numMaxCorr != 1)
      throw Error(PPF_ERR_INVALID, "ICP: only uniform sampling with one correspondence per point is implemented");
    ppf_default_icp_params(&prm_);
    prm_.iterations = iterations;
    prm_.tolerance = tolerence;
    prm_.rejection_scale = rejectionScale;
    prm_.num_levels = numLevels;
  }
  virtual ~ICP() {}

  /* one registration from the identity: returns 0, fills residual and the 4x4 src -> dst */
  int registerModelToScene(const Mat& srcPC, const Mat& dstPC, double& residual, Matx44d& pose) {
    require_cloud(srcPC); require_cloud(dstPC);
    check(ppf_icp_register(srcPC.ptr<float>(0), srcPC.rows, detail::stride_of(srcPC), PPF_NOFF_MAT, dstPC.ptr<float>(0), dstPC.rows,
                           detail::stride_of(dstPC), PPF_NOFF_MAT, &prm_, pose.val, &residual, nullptr));
    return 0;
  }
  /* every pose: move the model by it, register to the scene, pose <- poseICP * pose, residual set */
  int registerModelToScene(const Mat& srcPC, const Mat& dstPC, std::vector<Pose3DPtr>& poses) {
    require_cloud(srcPC); require_cloud(dstPC);
    std::vector<ppf_pose> recs(poses.size());
    for (size_t i = 0; i < poses.size(); i++) recs[i] = poses[i]->record();
    check(ppf_icp_refine(srcPC.ptr<float>(0), srcPC.rows, detail::stride_of(srcPC), PPF_NOFF_MAT, dstPC.ptr<float>(0), dstPC.rows,
                         detail::stride_of(dstPC), PPF_NOFF_MAT, &prm_, recs.data(), (int)recs.size(), nullptr));
    for (size_t i = 0; i < poses.size(); i++) *poses[i] = Pose3D(recs[i]);
    return 0;
  }

 private:
  static void require_cloud(const Mat& m) {
    if (m.rows <= 0 || m.cols < 6) throw Error(PPF_ERR_INVALID, "ICP: expected an N x 6 float32 cloud (x y z nx ny nz)");
  }
  ppf_icp_params prm_;
};

/* ---- helpers the reference's driver takes from the same namespace ---------------------------------- */
inline Mat loadPLYSimple(const char* fileName, int withNormals = 0) {
  std::ifstream ifs(fileName);
  if (!ifs.is_open()) throw Error(PPF_ERR_IO, std::string("Error opening input file: ") + fileName);
  std::string line;
  int numVertices = 0, numProps = 0;
  bool inVertex = false;
  while (std::getline(ifs, line)) {
    std::istringstream ss(line);
    std::string tok;
    ss >> tok;
    if (tok == "element") { std::string what; ss >> what; inVertex = (what == "vertex"); if (inVertex) ss >> numVertices; }
    else if (tok == "property" && inVertex) numProps++;
    else if (tok == "format") { std::string f; ss >> f; if (f != "ascii") throw Error(PPF_ERR_IO, "loadPLYSimple: only ascii PLY is supported"); }
    else if (tok == "end_header") break;
  }
  const int cols = withNormals ? 6 : 3;
  if (numProps < cols) throw Error(PPF_ERR_IO, "loadPLYSimple: not enough vertex properties");
  Mat cloud = detail::new_cloud(numVertices, cols);
  for (int i = 0; i < numVertices; i++) {
    float* row = cloud.ptr<float>(i);
    std::getline(ifs, line);
    std::istringstream ss(line);
    for (int c = 0; c < numProps; c++) { float v; ss >> v; if (c < cols) row[c] = v; }
    if (withNormals) {
      const double nrm = std::sqrt((double)row[3] * row[3] + (double)row[4] * row[4] + (double)row[5] * row[5]);
      if (nrm > 0.00001) { row[3] = (float)(row[3] / nrm); row[4] = (float)(row[4] / nrm); row[5] = (float)(row[5] / nrm); }
    }
  }
  return cloud;
}

inline void writePLY(Mat PC, const char* fileName) {
  const int stride = detail::stride_of(PC);
  (void)stride; /* the CV_32FC1 check */
  std::ofstream out(fileName);
  if (!out) throw Error(PPF_ERR_IO, std::string("Error opening output file: ") + fileName);
  out << "ply\nformat ascii 1.0\nelement vertex " << PC.rows << "\nproperty float x\nproperty float y\nproperty float z\n";
  if (PC.cols >= 6) out << "property float nx\nproperty float ny\nproperty float nz\n";
  out << "end_header\n";
  for (int i = 0; i < PC.rows; i++) {
    const float* r = PC.ptr<float>(i);
    out << r[0] << " " << r[1] << " " << r[2];
    if (PC.cols >= 6) out << " " << r[3] << " " << r[4] << " " << r[5];
    out << "\n";
  }
}

/* src/YOLO_cropping_ppf_test.cpp:125: Mat object_trans = transformPCPose(bottle, result_pose.pose); */
inline Mat transformPCPose(Mat pc, const Matx44d& Pose) {
  if (pc.cols < 6) throw Error(PPF_ERR_INVALID, "transformPCPose: expected an N x 6 cloud");
  Mat out = detail::new_cloud(pc.rows, 6);
  if (pc.rows > 0) check(ppf_transform_pc_pose(pc.ptr<float>(0), pc.rows, detail::stride_of(pc), PPF_NOFF_MAT, Pose.val, out.ptr<float>(0)));
  return out;
}

inline Mat samplePCByQuantization(Mat pc, float sampleStep) {
  if (pc.rows <= 0 || pc.cols < 6) throw Error(PPF_ERR_INVALID, "samplePCByQuantization: expected an N x 6 cloud");
  int n = 0;
  const int stride = detail::stride_of(pc);
  check(ppf_sample_cloud(pc.ptr<float>(0), pc.rows, stride, PPF_NOFF_MAT, sampleStep, nullptr, 0, &n));
  Mat out = detail::new_cloud(n, 6);
  if (n > 0) check(ppf_sample_cloud(pc.ptr<float>(0), pc.rows, stride, PPF_NOFF_MAT, sampleStep, out.ptr<float>(0), n, &n));
  return out;
}

}  // namespace ppf_match_3d
}  // namespace ppfhip

#if defined(PPF_MATCH_3D_HAVE_OPENCV) && defined(PPF_MATCH_3D_AS_CV)
/* the name the reference spells: cv::ppf_match_3d (`using namespace cv; using namespace ppf_match_3d;`, CloudProcessing.h:28-29,
 * and `ppf_match_3d::PPF3DDetector`, :205).  Only for builds that no longer include <opencv2/surface_matching.hpp>. */
namespace cv { namespace ppf_match_3d = ::ppfhip::ppf_match_3d; }
#endif

#endif /* PPF_MATCH_3D_HPP */
