#pragma once
#include "solvercategory.hh"
namespace Dune {
template <class X, class Y>
class LinearOperator {
public:
  using domain_type = X;
  using range_type = Y;
  using field_type = double;
  virtual void apply(const X& x, Y& y) const = 0;
  virtual void applyscaleadd(field_type alpha, const X& x, Y& y) const = 0;
  virtual SolverCategory::Category category() const = 0;
  virtual ~LinearOperator() = default;
};
template <class M, class X, class Y>
class AssembledLinearOperator : public LinearOperator<X, Y> {
public:
  using matrix_type = M;
  virtual const M& getmat() const = 0;
};
}  // namespace Dune
