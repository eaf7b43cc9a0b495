#pragma once
#include <cstddef>
#include <map>
#include <utility>
#include <vector>
namespace Dune {
class InterfaceInformation {
public:
  std::size_t size() const { return idx.size(); }
  std::size_t operator[](std::size_t i) const { return idx[i]; }
  std::vector<std::size_t> idx;
};
// build(remoteIndices, sourceFlags, destFlags): for every neighbour the local indices whose OWN attribute is
// in sourceFlags (send list) resp. in destFlags with the REMOTE attribute in sourceFlags (receive list),
// both in the neighbour list's order (ascending global index).
class Interface {
public:
  using InformationMap = std::map<int, std::pair<InterfaceInformation, InterfaceInformation>>;
  template <class RemoteIndices, class S, class D>
  void build(const RemoteIndices& ri, const S& src, const D& dst)
  {
    m.clear();
    for (const auto& [nbr, list] : ri.lists) {
      std::pair<InterfaceInformation, InterfaceInformation> p;
      for (const auto& e : list) {
        if (src.contains(e.mine) && dst.contains(e.remote)) p.first.idx.push_back(e.local);
        if (dst.contains(e.mine) && src.contains(e.remote)) p.second.idx.push_back(e.local);
      }
      if (p.first.size() || p.second.size()) m.emplace(nbr, std::move(p));
    }
  }
  const InformationMap& interfaces() const { return m; }
  void free() { m.clear(); }
private:
  InformationMap m;
};
}  // namespace Dune
