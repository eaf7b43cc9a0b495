#pragma once
// Single-process stand-in of OwnerOverlapCopyCommunication: index set with attributes, per-neighbour
// remote index lists (filled by hand in tests), a trivial communicator.
#include <cmath>
#include <cstddef>
#include <map>
#include <vector>
namespace Dune {
struct OwnerOverlapCopyAttributeSet {
  enum AttributeSet { owner = 1, overlap = 2, copy = 3 };
};
namespace mock {
template <int... A>
struct Flags {
  bool contains(int a) const { return ((a == A) || ...); }
};
struct LocalIndex {
  std::size_t l;
  int attr;
  std::size_t local() const { return l; }
  int attribute() const { return attr; }
};
struct IndexPair {
  std::size_t g;
  LocalIndex li;
  std::size_t global() const { return g; }
  const LocalIndex& local() const { return li; }
};
struct IndexSet {
  std::vector<IndexPair> v;
  std::size_t size() const { return v.size(); }
  auto begin() const { return v.begin(); }
  auto end() const { return v.end(); }
};
struct RemoteEntry {
  std::size_t local;
  int mine, remote;
};
struct RemoteIndices {
  std::map<int, std::vector<RemoteEntry>> lists;   // ascending global index per neighbour
};
struct Communicator {
  int rank() const { return 0; }
  int size() const { return 1; }
  template <class T>
  void allgather(const T* in, int len, T* out) const { for (int i = 0; i < len; ++i) out[i] = in[i]; }
  template <class T>
  void sum(T*, int) const {}
};
}  // namespace mock
template <class G, class L>
class OwnerOverlapCopyCommunication {
public:
  using OwnerSet = mock::Flags<OwnerOverlapCopyAttributeSet::owner>;
  using CopySet = mock::Flags<OwnerOverlapCopyAttributeSet::copy>;
  using OwnerCopySet = mock::Flags<OwnerOverlapCopyAttributeSet::owner, OwnerOverlapCopyAttributeSet::copy>;
  using AllSet = mock::Flags<OwnerOverlapCopyAttributeSet::owner, OwnerOverlapCopyAttributeSet::overlap, OwnerOverlapCopyAttributeSet::copy>;
  mock::IndexSet& indexSet() { return is; }
  const mock::IndexSet& indexSet() const { return is; }
  mock::RemoteIndices& remoteIndices() { return ri; }
  const mock::RemoteIndices& remoteIndices() const { return ri; }
  const mock::Communicator& communicator() const { return cc; }
  // single process: every index has one holder, the exchanges move nothing
  template <class T1, class T2> void copyOwnerToAll(const T1&, T2&) const {}
  template <class T1, class T2> void addOwnerCopyToAll(const T1&, T2&) const {}
  template <class T1, class T2> void addOwnerCopyToOwnerCopy(const T1&, T2&) const {}
  template <class T> double norm(const T& x) const
  {
    double s = 0;
    for (const auto& idx : is)
      if (idx.local().attribute() == OwnerOverlapCopyAttributeSet::owner) s += x[idx.local().local()][0] * x[idx.local().local()][0];
    return std::sqrt(s);
  }
private:
  mock::IndexSet is;
  mock::RemoteIndices ri;
  mock::Communicator cc;
};
}  // namespace Dune
