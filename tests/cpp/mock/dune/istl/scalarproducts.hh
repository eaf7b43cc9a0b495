#pragma once
#include "solvercategory.hh"
namespace Dune {
template <class X>
class ScalarProduct {
public:
  using field_type = double;
  using real_type = double;
  virtual field_type dot(const X& x, const X& y) const = 0;
  virtual real_type norm(const X& x) const = 0;
  virtual SolverCategory::Category category() const = 0;
  virtual ~ScalarProduct() = default;
};
}  // namespace Dune
