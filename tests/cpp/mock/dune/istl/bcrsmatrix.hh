#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
namespace Dune {
template <class K, int r, int c>
struct FieldMatrix {
  K v[r][c];
  K* operator[](int i) { return v[i]; }
  const K* operator[](int i) const { return v[i]; }
};
// CSR-backed stand-in exposing the iterator interface the adaptors use
// (A.begin()/end(), ri->begin()/end(), cit.index(), *cit, N(), M(), nonzeroes()).
template <class B>
class BCRSMatrix {
public:
  class ColIterator {
  public:
    ColIterator(const BCRSMatrix* A, std::size_t k) : A(A), k(k) {}
    std::size_t index() const { return A->ci[k]; }
    const B& operator*() const { return A->va[k]; }
    ColIterator& operator++() { ++k; return *this; }
    bool operator!=(const ColIterator& o) const { return k != o.k; }
  private:
    const BCRSMatrix* A;
    std::size_t k;
  };
  class Row {
  public:
    Row(const BCRSMatrix* A, std::size_t i) : A(A), i(i) {}
    ColIterator begin() const { return ColIterator(A, A->rp[i]); }
    ColIterator end() const { return ColIterator(A, A->rp[i + 1]); }
  private:
    const BCRSMatrix* A;
    std::size_t i;
  };
  class RowIterator {
  public:
    RowIterator(const BCRSMatrix* A, std::size_t i) : row(A, i), A(A), i(i) {}
    std::size_t index() const { return i; }
    const Row* operator->() const { return &row; }
    RowIterator& operator++() { ++i; row = Row(A, i); return *this; }
    bool operator!=(const RowIterator& o) const { return i != o.i; }
  private:
    Row row;
    const BCRSMatrix* A;
    std::size_t i;
  };
  BCRSMatrix() = default;
  BCRSMatrix(std::size_t n, std::size_t m, std::vector<std::size_t> rp_, std::vector<std::size_t> ci_, const std::vector<double>& v)
      : n(n), m(m), rp(std::move(rp_)), ci(std::move(ci_)), va(v.size())
  {
    for (std::size_t k = 0; k < v.size(); ++k) va[k][0][0] = v[k];
  }
  BCRSMatrix& operator=(double x)   // scalar assignment: every stored entry
  {
    for (auto& b : va) b[0][0] = x;
    return *this;
  }
  void copy_values_from(const BCRSMatrix& o) { va = o.va; }   // (stand-in only: same pattern assumed)
  std::size_t N() const { return n; }
  std::size_t M() const { return m; }
  std::size_t nonzeroes() const { return ci.size(); }
  Row operator[](std::size_t i) const { return Row(this, i); }
  RowIterator begin() const { return RowIterator(this, 0); }
  RowIterator end() const { return RowIterator(this, n); }
private:
  std::size_t n = 0, m = 0;
  std::vector<std::size_t> rp, ci;
  std::vector<B> va;
};
}  // namespace Dune
