/*
 * TEST SCAFFOLDING, not OpenCV: just enough of cv::Mat / cv::Matx / cv::Vec / cv::Ptr / cv::FileStorage / cv::FileNode /
 * CV_Error / getTickCount for tests/test_cpp_facade.py to COMPILE AND RUN the OpenCV-present configuration of
 * include/ppf_match_3d.hpp in a container without OpenCV (reference call sites: CloudProcessing.h:28-32 file-scope using
 * directives, :46-48 vector<Mat>, :111-113, :249-251 FileStorage, :167-188 Mat(rows, cols, CV_32FC1) + ptr<float>(i),
 * :435-439 CV_Error, :441-446 tick counters; cv::Mat views with a row step larger than cols).  Member names and
 * signatures follow OpenCV 4's core.hpp; the storage keeps its nodes in memory and in a flat text file.
 */
#ifndef MOCK_OPENCV_CORE_HPP
#define MOCK_OPENCV_CORE_HPP
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_32FC1 CV_32F
#define CV_8UC1 CV_8U

#define CV_Error(code, msg) throw ::cv::Exception((code), (msg), __func__, __FILE__, __LINE__)
#define CV_Assert(expr) do { if (!(expr)) CV_Error(-215, #expr); } while (0)

namespace cv {
typedef int64_t int64;
typedef std::string String;

class Exception : public std::runtime_error {
 public:
  Exception(int c, const std::string& m, const char* fn, const char* file, int ln)
      : std::runtime_error(std::string(file) + ":" + std::to_string(ln) + ": error: (" + std::to_string(c) + ") " + m + " in function '" + fn + "'"),
        code(c), err(m) {}
  int code;
  std::string err;
};

inline int64 getTickCount() { return (int64)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline double getTickFrequency() { return 1e9; }

/* cv::Ptr<T> of OpenCV 4: a std::shared_ptr<T> with its own constructors */
template <class T> struct Ptr : public std::shared_ptr<T> {
  Ptr() {}
  Ptr(T* p) : std::shared_ptr<T>(p) {} /* implicit in OpenCV too */
  Ptr(const std::shared_ptr<T>& o) : std::shared_ptr<T>(o) {}
  bool empty() const { return !this->get(); }
};
template <class T, class... A> Ptr<T> makePtr(A&&... a) { return Ptr<T>(std::make_shared<T>(std::forward<A>(a)...)); }

template <class T, int m, int n> class Matx {
 public:
  Matx() { for (int k = 0; k < m * n; k++) val[k] = T(0); }
  static Matx all(T v) { Matx r; for (int k = 0; k < m * n; k++) r.val[k] = v; return r; }
  static Matx eye() { Matx r; for (int k = 0; k < (m < n ? m : n); k++) r.val[k * n + k] = T(1); return r; }
  T& operator()(int i, int j) { return val[i * n + j]; }
  const T& operator()(int i, int j) const { return val[i * n + j]; }
  T val[m * n];
};
template <class T, int m, int n> bool operator==(const Matx<T, m, n>& a, const Matx<T, m, n>& b) {
  for (int k = 0; k < m * n; k++) if (a.val[k] != b.val[k]) return false;
  return true;
}
template <class T, int m, int n> bool operator!=(const Matx<T, m, n>& a, const Matx<T, m, n>& b) { return !(a == b); }
template <class T, int cn> class Vec : public Matx<T, cn, 1> {
 public:
  Vec() {}
  T& operator[](int i) { return this->val[i]; }
  const T& operator[](int i) const { return this->val[i]; }
};
typedef Matx<double, 4, 4> Matx44d;
typedef Vec<double, 3> Vec3d;
typedef Vec<double, 4> Vec4d;

template <class T> struct Rect_ { T x, y, width, height; Rect_() : x(0), y(0), width(0), height(0) {} Rect_(T a, T b, T c, T d) : x(a), y(b), width(c), height(d) {} };
typedef Rect_<int> Rect;

class Mat {
 public:
  Mat() : rows(0), cols(0), data(nullptr), type_(CV_8U), step_(0) {}
  Mat(int r, int c, int type) : rows(r), cols(c), type_(type), step_((size_t)c * esz(type)) {
    own_ = std::make_shared<std::vector<unsigned char>>((size_t)r * step_);
    data = own_->data();
  }
  Mat(int r, int c, int type, void* ext, size_t step_bytes = 0) : rows(r), cols(c), data((unsigned char*)ext), type_(type),
                                                               step_(step_bytes ? step_bytes : (size_t)c * esz(type)) {}
  int rows, cols;
  unsigned char* data;
  bool empty() const { return rows == 0 || cols == 0 || !data; }
  int depth() const { return type_; }
  int channels() const { return 1; }
  size_t step1() const { return step_ / esz(type_); }
  bool isContinuous() const { return step_ == (size_t)cols * esz(type_); }
  template <class T> T* ptr(int i = 0) { return reinterpret_cast<T*>(data + (size_t)i * step_); }
  template <class T> const T* ptr(int i = 0) const { return reinterpret_cast<const T*>(data + (size_t)i * step_); }
  template <class T> T& at(int i, int j) { return ptr<T>(i)[j]; }
  template <class T> const T& at(int i, int j) const { return ptr<T>(i)[j]; }
  int type() const { return type_; }
  Mat clone() const {
    Mat m(rows, cols, type_);
    for (int i = 0; i < rows; i++) std::memcpy(m.data + (size_t)i * m.step_, data + (size_t)i * step_, (size_t)cols * esz(type_));
    return m;
  }
  /* columns [c0, c1) of every row: a non-continuous view on the same memory */
  Mat colRange(int c0, int c1) const {
    Mat v(rows, c1 - c0, type_, data + (size_t)c0 * esz(type_), step_);
    v.own_ = own_;
    return v;
  }

 private:
  static size_t esz(int t) { return t == CV_8U ? 1 : (t == CV_32F ? 4 : 8); }
  int type_;
  size_t step_;
  std::shared_ptr<std::vector<unsigned char>> own_;
};

class FileNode {
 public:
  FileNode() : nodes_(nullptr) {}
  explicit FileNode(const std::map<std::string, Mat>* nodes, std::string key = std::string()) : nodes_(nodes), key_(std::move(key)) {}
  FileNode operator[](const char* k) const { return FileNode(nodes_, k); }
  bool empty() const { return !nodes_ || nodes_->find(key_) == nodes_->end(); }
  const Mat* mat() const { return empty() ? nullptr : &nodes_->find(key_)->second; }

 private:
  const std::map<std::string, Mat>* nodes_;
  std::string key_;
};
inline void operator>>(const FileNode& n, Mat& m) { m = n.mat() ? n.mat()->clone() : Mat(); }

class FileStorage {
 public:
  enum Mode { READ = 0, WRITE = 1 };
  FileStorage() : mode_(READ), released_(true) {}
  FileStorage(const std::string& file, int mode) : file_(file), mode_(mode) {
    if (mode == READ) {
      std::ifstream in(file.c_str(), std::ios::binary);
      std::string key;
      int r, c, t;
      while (in >> key >> r >> c >> t) {
        in.get();
        Mat m(r, c, t);
        in.read(reinterpret_cast<char*>(m.data), (std::streamsize)((size_t)r * m.step1() * (t == CV_8U ? 1 : (t == CV_32F ? 4 : 8))));
        nodes_[key] = m;
      }
    }
  }
  ~FileStorage() { release(); }
  bool isOpened() const { return true; }
  FileNode root() const { return FileNode(&nodes_); }
  FileNode operator[](const char* k) const { return FileNode(&nodes_, k); }
  void release() {
    if (mode_ == WRITE && !released_) {
      std::ofstream out(file_.c_str(), std::ios::binary);
      for (auto& kv : nodes_) {
        out << kv.first << " " << kv.second.rows << " " << kv.second.cols << " " << kv.second.depth() << "\n";
        const Mat c = kv.second.clone();
        out.write(reinterpret_cast<const char*>(c.data), (std::streamsize)((size_t)c.rows * c.step1() * (c.depth() == CV_8U ? 1 : (c.depth() == CV_32F ? 4 : 8))));
      }
    }
    released_ = true;
  }
  /* fs << "key" << mat; */
  FileStorage& operator<<(const char* key) { pending_ = key; return *this; }
  FileStorage& operator<<(const std::string& key) { pending_ = key; return *this; }
  FileStorage& operator<<(const Mat& m) { nodes_[pending_] = m.clone(); return *this; }

 private:
  std::string file_, pending_;
  int mode_;
  bool released_ = false;
  std::map<std::string, Mat> nodes_;
};
}  // namespace cv
#endif
