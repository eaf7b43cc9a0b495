#pragma once
#include "solvercategory.hh"
namespace Dune {
struct InverseOperatorResult {
  int iterations = 0;
  double reduction = 0, conv_rate = 0, elapsed = 0, condition_estimate = -1;
  bool converged = false;
  void clear() { *this = InverseOperatorResult(); }
};
template <class X, class Y>
class InverseOperator {
public:
  using domain_type = X;
  using range_type = Y;
  virtual void apply(X& x, Y& b, InverseOperatorResult& res) = 0;
  virtual void apply(X& x, Y& b, double reduction, InverseOperatorResult& res) = 0;
  virtual SolverCategory::Category category() const = 0;
  virtual ~InverseOperator() = default;
};
}  // namespace Dune
