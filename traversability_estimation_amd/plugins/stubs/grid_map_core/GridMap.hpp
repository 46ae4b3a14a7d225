// stand-in for grid_map_core's GridMap: float32 column-major layers + geometry, only what the adapters use
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace grid_map {

struct Vec2d {
  double v[2];
  double x() const { return v[0]; }
  double y() const { return v[1]; }
  double operator()(int k) const { return v[k]; }
};
struct Arr2i {
  int v[2];
  int operator()(int k) const { return v[k]; }
};

class Matrix {  // Eigen::MatrixXf look-alike (column-major)
 public:
  Matrix() : r_(0), c_(0) {}
  Matrix(int r, int c) : r_(r), c_(c), d_((size_t)r * c) {}
  int rows() const { return r_; }
  int cols() const { return c_; }
  float* data() { return d_.data(); }
  const float* data() const { return d_.data(); }
  float& operator()(int i, int j) { return d_[(size_t)j * r_ + i]; }
  float operator()(int i, int j) const { return d_[(size_t)j * r_ + i]; }
  void setConstant(float v) { d_.assign(d_.size(), v); }

 private:
  int r_, c_;
  std::vector<float> d_;
};

class GridMap {
 public:
  GridMap() : res_(0), stamp_(0) { len_ = {{0, 0}}; pos_ = {{0, 0}}; size_ = {{0, 0}}; start_ = {{0, 0}}; }
  void setTimestamp(uint64_t t) { stamp_ = t; }  // grid_map::Time, nanoseconds
  uint64_t getTimestamp() const { return stamp_; }
  void setGeometry(const Vec2d& length, double resolution, const Vec2d& position) {
    size_.v[0] = (int)std::lround(length(0) / resolution);
    size_.v[1] = (int)std::lround(length(1) / resolution);
    res_ = resolution;
    len_.v[0] = size_(0) * resolution;
    len_.v[1] = size_(1) * resolution;
    pos_ = position;
  }
  void add(const std::string& layer, float value = std::numeric_limits<float>::quiet_NaN()) {
    Matrix m(size_(0), size_(1));
    m.setConstant(value);
    layers_[layer] = m;
  }
  bool exists(const std::string& layer) const { return layers_.count(layer) != 0; }
  bool erase(const std::string& layer) { return layers_.erase(layer) != 0; }
  Matrix& get(const std::string& layer) {
    std::map<std::string, Matrix>::iterator it = layers_.find(layer);
    if (it == layers_.end()) throw std::out_of_range("GridMap::get(...) : No map layer '" + layer + "' available.");
    return it->second;
  }
  const Matrix& get(const std::string& layer) const { return const_cast<GridMap*>(this)->get(layer); }
  Matrix& operator[](const std::string& layer) { return get(layer); }
  const Arr2i& getSize() const { return size_; }
  double getResolution() const { return res_; }
  const Vec2d& getLength() const { return len_; }
  const Vec2d& getPosition() const { return pos_; }
  const Arr2i& getStartIndex() const { return start_; }
  bool isDefaultStartIndex() const { return start_(0) == 0 && start_(1) == 0; }
  void setStartIndex(const Arr2i& start) { start_ = start; }  // the layers are circular buffers (GridMap::move)
  void convertToDefaultStartIndex() {}
  std::vector<std::string> getLayers() const {
    std::vector<std::string> v;
    for (std::map<std::string, Matrix>::const_iterator it = layers_.begin(); it != layers_.end(); ++it) v.push_back(it->first);
    return v;
  }

 private:
  std::map<std::string, Matrix> layers_;
  Vec2d len_, pos_;
  Arr2i size_, start_;
  double res_;
  uint64_t stamp_;
};

}  // namespace grid_map
