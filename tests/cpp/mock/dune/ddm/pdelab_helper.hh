#pragma once
// Stand-ins for make_communication (dune/ddm/pdelab_helper.hh:16) and make_additive (:108-149) on one process.
#include <cstddef>
#include <memory>
#include <dune/istl/owneroverlapcopy.hh>
template <class GFS>
auto make_communication(const GFS& gfs)
{
  using Comm = Dune::OwnerOverlapCopyCommunication<std::size_t, int>;
  auto c = std::make_shared<Comm>();
  for (std::size_t i = 0; i < gfs.size(); ++i) c->indexSet().v.push_back({i, {i, Dune::OwnerOverlapCopyAttributeSet::owner}});
  return c;
}
template <class Mat, class Communication>
void make_additive(Mat&, const Communication&) {}   // (every row is an owner row on one process)
