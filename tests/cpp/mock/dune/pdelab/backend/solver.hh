#pragma once
namespace Dune::PDELab {
template <class RFType>
struct LinearSolverResult {
  bool converged = false;
  unsigned int iterations = 0;
  double elapsed = 0;
  RFType reduction = 0, conv_rate = 0;
};
class LinearResultStorage {
public:
  const LinearSolverResult<double>& result() const { return res; }
protected:
  LinearSolverResult<double> res;
};
}  // namespace Dune::PDELab
