#pragma once
#include <cstddef>
#include <vector>
namespace Dune {
template <class K, int n>
struct FieldVector {
  K v[n];
  K& operator[](int i) { return v[i]; }
  const K& operator[](int i) const { return v[i]; }
  operator K() const { return v[0]; }
  FieldVector& operator=(K x) { for (int i = 0; i < n; ++i) v[i] = x; return *this; }
};
template <class B>
class BlockVector {
public:
  using field_type = double;
  using block_type = B;
  BlockVector() = default;
  explicit BlockVector(std::size_t n) : d(n) {}
  std::size_t N() const { return d.size(); }
  std::size_t size() const { return d.size(); }
  void resize(std::size_t n) { d.resize(n); }
  B& operator[](std::size_t i) { return d[i]; }
  const B& operator[](std::size_t i) const { return d[i]; }
  BlockVector& operator=(double x) { for (auto& b : d) b = x; return *this; }
  BlockVector& operator+=(const BlockVector& o) { for (std::size_t i = 0; i < d.size(); ++i) d[i][0] += o.d[i][0]; return *this; }
  BlockVector& operator-=(const BlockVector& o) { for (std::size_t i = 0; i < d.size(); ++i) d[i][0] -= o.d[i][0]; return *this; }
  BlockVector& operator*=(double a) { for (auto& b : d) b[0] *= a; return *this; }
  void axpy(double a, const BlockVector& o) { for (std::size_t i = 0; i < d.size(); ++i) d[i][0] += a * o.d[i][0]; }
private:
  std::vector<B> d;
};
}  // namespace Dune
