// tests/mini_ecto/mini_ecto.hpp -- TEST DOUBLE, not part of the product.
// The smallest ecto::tendrils / cv::Mat / cv::DMatch / cv::KeyPoint surface that adapter/ecto_cells.hpp touches, so
// that the adapter's declare/configure/process protocol can be compiled and run in an image that has neither
// ecto nor OpenCV. It contains no detection logic. (It is NOT used to build any reference source.)
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <typeinfo>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_8UC4 24
#define CV_16U 2
#define CV_32F 5
#define CV_64F 6
#define CV_32FC3 21

namespace cv {
struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
struct DMatch { int queryIdx, trainIdx, imgIdx; float distance; };
class Mat {
 public:
  int rows, cols;
  Mat() : rows(0), cols(0), type_(CV_8U) {}
  Mat(int r, int c, int type) : rows(r), cols(c), type_(type), buf_(new std::vector<uint8_t>((size_t)r * c * esz(type))) {}
  static size_t esz(int type) {
    return type == CV_8U ? 1 : type == CV_8UC3 ? 3 : type == CV_8UC4 ? 4 : type == CV_16U ? 2 : type == CV_32F ? 4 : type == CV_64F ? 8 : 12;
  }
  bool empty() const { return rows == 0 || cols == 0; }
  bool isContinuous() const { return true; }
  int type() const { return type_; }
  Mat clone() const { Mat m(rows, cols, type_); if (buf_) *m.buf_ = *buf_; return m; }
  Mat t() const {                       // only vector shapes are transposed by the adapter (n x 1 <-> 1 x n)
    if (rows != 1 && cols != 1) throw std::runtime_error("mini cv::Mat::t(): vector shapes only");
    Mat m = clone(); m.rows = cols; m.cols = rows; return m;
  }
  template <typename T> T* ptr(int r) { return reinterpret_cast<T*>(buf_ ? buf_->data() + (size_t)r * cols * esz(type_) : nullptr); }
  template <typename T> const T* ptr(int r) const { return reinterpret_cast<const T*>(buf_ ? buf_->data() + (size_t)r * cols * esz(type_) : nullptr); }
 private:
  int type_;
  std::shared_ptr<std::vector<uint8_t> > buf_;
};
}  // namespace cv

namespace ecto {
enum ReturnCode { OK = 0 };
class tendril {
 public:
  tendril() : required_(false) {}
  tendril& required(bool r) { required_ = r; return *this; }
  template <typename T> void set(const T& v) { holder_.reset(new T(v), [](void* p) { delete static_cast<T*>(p); }); type_ = typeid(T).name(); }
  template <typename T> const T& get() const {
    if (!holder_) throw std::runtime_error("tendril has no value");
    if (type_ != typeid(T).name()) throw std::runtime_error("tendril type mismatch");
    return *static_cast<const T*>(holder_.get());
  }
  template <typename T> const tendril& operator<<(const T& v) const { const_cast<tendril*>(this)->set<T>(v); return *this; }
  bool required_;
  std::string doc_, type_;
 private:
  std::shared_ptr<void> holder_;
};
class tendrils {
 public:
  template <typename T> tendril& declare(const std::string& name, const std::string& doc) {
    tendril& t = map_[name]; t.doc_ = doc; t.type_ = typeid(T).name(); return t;
  }
  template <typename T> tendril& declare(const std::string& name, const std::string& doc, const T& dflt) {
    tendril& t = declare<T>(name, doc); t.set<T>(dflt); return t;
  }
  template <typename T> const T& get(const std::string& name) const { return at(name).get<T>(); }
  const tendril& operator[](const std::string& name) const { return at(name); }
  tendril& operator[](const std::string& name) { return map_[name]; }
  bool has(const std::string& name) const { return map_.count(name) != 0; }
  std::vector<std::string> names() const { std::vector<std::string> n; for (auto& kv : map_) n.push_back(kv.first); return n; }
 private:
  const tendril& at(const std::string& name) const {
    std::map<std::string, tendril>::const_iterator it = map_.find(name);
    if (it == map_.end()) throw std::runtime_error("no tendril named " + name);
    return it->second;
  }
  mutable std::map<std::string, tendril> map_;
};
}  // namespace ecto
