// TEST INFRASTRUCTURE: stand-in for boost::shared_array (only what include/stereo_binary_feature.h of the reference needs).
#pragma once
#include <memory>
namespace boost {
template <typename T> class shared_array {
 public:
  shared_array() {}
  explicit shared_array(T* p) : p_(p, std::default_delete<T[]>()) {}
  T& operator[](long i) const { return p_.get()[i]; }
  T* get() const { return p_.get(); }
 private:
  std::shared_ptr<T> p_;
};
}  // namespace boost
