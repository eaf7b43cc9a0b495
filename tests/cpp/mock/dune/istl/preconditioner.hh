#pragma once
#include "solvercategory.hh"
namespace Dune {
template <class X, class Y>
class Preconditioner {
public:
  using domain_type = X;
  using range_type = Y;
  virtual void pre(X& x, Y& b) = 0;
  virtual void apply(X& v, const Y& d) = 0;
  virtual void post(X& x) = 0;
  virtual SolverCategory::Category category() const = 0;
  virtual ~Preconditioner() = default;
};
}  // namespace Dune
