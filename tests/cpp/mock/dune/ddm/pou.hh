#pragma once
// stand-in for the reference's dune/ddm/pou.hh: the accessors (pou.hh:190-208) and, for the PDELab-facing adaptor, the constructor
// from matrix + communication + parameter tree (:161) on ONE process (no neighbours: weight 1 everywhere)
#include <cstddef>
#include <string>
#include <vector>
#include <dune/common/parametertree.hh>
class PartitionOfUnity {
public:
  explicit PartitionOfUnity(std::vector<double> v, int shrink = 0) : v(std::move(v)), shrink_(shrink) {}
  template <class Mat, class Communication>
  PartitionOfUnity(const Mat& A, const Communication&, const Dune::ParameterTree& ptree, int /*overlap*/, const std::string& subtree_name = "pou")
      : v(A.N(), 1.0), shrink_(ptree.sub(subtree_name).get("shrink", 0))
  {
  }
  int get_shrink() const { return shrink_; }
  std::size_t size() const { return v.size(); }
  double operator[](std::size_t i) const { return v[i]; }
private:
  std::vector<double> v;
  int shrink_ = 0;
};
