#pragma once
// stand-in for the reference's dune/ddm/pou.hh: only the accessors (pou.hh:190-208)
#include <cstddef>
#include <vector>
class PartitionOfUnity {
public:
  explicit PartitionOfUnity(std::vector<double> v, int shrink = 0) : v(std::move(v)), shrink_(shrink) {}
  int get_shrink() const { return shrink_; }
  std::size_t size() const { return v.size(); }
  double operator[](std::size_t i) const { return v[i]; }
private:
  std::vector<double> v;
  int shrink_ = 0;
};
