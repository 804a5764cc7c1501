// Drop-in for include/stereo_binary_feature.h + src/stereo_binary_feature.cpp (the descriptor carrier).
#ifndef GHICP_DROPIN_SBF_H_
#define GHICP_DROPIN_SBF_H_
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "ghicp_shim_types.h"

namespace ghicp {
class StereoBinaryFeature {
 public:
  char* feature_;      // new char[byte_], deep-copied (stereo_binary_feature.h:27, 93-128)
  unsigned int size_;  // number of bits
  unsigned int byte_;  // number of bytes
  int bscVisualWordsIndex_ = 0;
  std::vector<int> bscVisualWordsIndexV_;
  size_t keypointIndex_ = 0;
  struct CoordinateSystem { Eigen::Vector3f xAxis, yAxis, zAxis, origin; };  // stereo_binary_feature.h:38-45
  CoordinateSystem localSystem_;
  StereoBinaryFeature(unsigned int size = 0) : feature_(nullptr), size_(size), byte_(0) {  // stereo_binary_feature.h:48-83
    if (size != 0) {
      byte_ = static_cast<unsigned int>(std::ceil(float(size_) / 8.f));
      feature_ = new char[byte_];
      for (unsigned int i = 0; i < byte_; i++) feature_[i] = 0;
    }
  }
  ~StereoBinaryFeature() { delete[] feature_; }
  StereoBinaryFeature(const StereoBinaryFeature& o)
      : feature_(nullptr), size_(o.size_), byte_(o.byte_), bscVisualWordsIndex_(o.bscVisualWordsIndex_), bscVisualWordsIndexV_(o.bscVisualWordsIndexV_),
        keypointIndex_(o.keypointIndex_), localSystem_(o.localSystem_) {
    if (byte_) { feature_ = new char[byte_]; std::memcpy(feature_, o.feature_, byte_); }
  }
  StereoBinaryFeature& operator=(const StereoBinaryFeature& o) {
    if (this == &o) return *this;
    delete[] feature_;
    feature_ = nullptr;
    size_ = o.size_; byte_ = o.byte_; bscVisualWordsIndex_ = o.bscVisualWordsIndex_; bscVisualWordsIndexV_ = o.bscVisualWordsIndexV_;
    keypointIndex_ = o.keypointIndex_; localSystem_ = o.localSystem_;
    if (byte_) { feature_ = new char[byte_]; std::memcpy(feature_, o.feature_, byte_); }
    return *this;
  }
  // stereo_binary_feature.cpp:87-104 (byte LUT popcount of XOR; -1 on size mismatch)
  int hammingDistance(const StereoBinaryFeature& a, const StereoBinaryFeature& b) const {
    if (a.size_ != b.size_) { std::cout << "Different size of binary feature\n"; return -1; }
    int c = 0;
    for (unsigned i = 0; i < a.byte_; i++) c += __builtin_popcount((unsigned char)(a.feature_[i] ^ b.feature_[i]));
    return c;
  }
  // stereo_binary_feature.cpp:107-124: u32 bits, u32 bytes, i32 count, raw bytes
  void writeFeatures(const std::vector<StereoBinaryFeature>& f, const std::string& path) const {
    std::ofstream o(path, std::ios::binary | std::ios::out);
    if (f.empty()) return;
    o.write((const char*)&f[0].size_, 4); o.write((const char*)&f[0].byte_, 4);
    int n = (int)f.size(); o.write((const char*)&n, 4);
    for (auto& x : f) o.write(x.feature_, f[0].byte_);
  }
  void readFeatures(std::vector<StereoBinaryFeature>& f, const std::string& path) const {
    f.clear();
    unsigned bits = 0, bytes = 0; int n = 0;
    std::ifstream i(path, std::ios::binary | std::ios::in);
    i.read((char*)&bits, 4); i.read((char*)&bytes, 4); i.read((char*)&n, 4);
    f.assign(n > 0 ? n : 0, StereoBinaryFeature(bits));
    for (auto& x : f) i.read(x.feature_, bytes);
  }
};
typedef StereoBinaryFeature SBF;
typedef std::vector<StereoBinaryFeature> vectorSBF;
typedef std::vector<vectorSBF> doubleVectorSBF;
}  // namespace ghicp
#endif
