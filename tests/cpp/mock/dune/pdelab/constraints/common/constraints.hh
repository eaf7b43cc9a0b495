#pragma once
#include <cstddef>
namespace Dune::PDELab {
// cc: the constrained local indices
template <class CC, class V>
void set_constrained_dofs(const CC& cc, double value, V& v)
{
  for (std::size_t i : cc) v.native()[i] = value;
}
}  // namespace Dune::PDELab
