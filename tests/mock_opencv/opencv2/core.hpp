/*
 * TEST SCAFFOLDING, not OpenCV: just enough of cv::Mat / cv::FileStorage / cv::FileNode for
 * tests/test_cpp_facade.py to COMPILE AND RUN the `#ifdef PPF_MATCH_3D_HAVE_OPENCV` overloads of
 * include/ppf_match_3d.hpp in a container without OpenCV (reference call sites: CloudProcessing.h:111-113, 249-251;
 * cv::Mat views with a row step larger than cols).  The storage keeps its nodes in memory and in a flat text file.
 */
#ifndef MOCK_OPENCV_CORE_HPP
#define MOCK_OPENCV_CORE_HPP
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_32FC1 CV_32F
#define CV_8UC1 CV_8U

namespace cv {
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
