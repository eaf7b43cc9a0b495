#pragma once
// Stand-ins for the PDELab ISTL backend containers: a vector / matrix wrapper around a shared native dune-istl object.
#include <cstddef>
#include <memory>
namespace Dune::PDELab {
namespace mock {
template <class GFS, class NativeVec>
class Vector {
public:
  using ElementType = double;
  using Container = NativeVec;
  explicit Vector(const GFS& gfs) : c(std::make_shared<NativeVec>(gfs.size())) { *c = 0; }
  Vector(const Vector& o) : c(std::make_shared<NativeVec>(*o.c)) {}
  Vector& operator=(const Vector& o) { *c = *o.c; return *this; }
  std::size_t N() const { return c->N(); }
  NativeVec& native() { return *c; }
  const NativeVec& native() const { return *c; }
private:
  std::shared_ptr<NativeVec> c;
};
template <class NativeMat>
class Matrix {
public:
  using Container = NativeMat;
  explicit Matrix(std::shared_ptr<NativeMat> m) : c(std::move(m)) {}
  std::shared_ptr<NativeMat> storage() const { return c; }
  NativeMat& native() { return *c; }
  const NativeMat& native() const { return *c; }
private:
  std::shared_ptr<NativeMat> c;
};
}  // namespace mock
namespace Backend {
template <class T>
using Native = typename T::Container;
template <class T>
auto& native(T& t) { return t.native(); }
template <class T>
const auto& native(const T& t) { return t.native(); }
}  // namespace Backend
}  // namespace Dune::PDELab
