#pragma once
#include <cstddef>
namespace Dune::PDELab {
// nodal interpolation on the stand-in function space: DoF i sits at gfs.position(i)
template <class F, class GFS, class V>
void interpolate(const F& f, const GFS& gfs, V& v)
{
  for (std::size_t i = 0; i < gfs.size(); ++i) v.native()[i] = f(gfs.position(i));
}
}  // namespace Dune::PDELab
