// Host arrays of matrix size: std::vector with an allocator that leaves new elements uninitialised (shared by the library's host code).
#pragma once
#include <algorithm>
#include <cstring>
#include <memory>
#include <thread>
#include <utility>
#include <vector>

// Host arrays of matrix size (10^8..10^9 entries): a std::vector whose resize() leaves the new elements UNINITIALISED -- the value
// initialisation of std::vector is a single-threaded pass over fresh pages (0.5 s per 3 GB), the threads that fill the array then
// touch the pages themselves.
template <class T>
struct noinit_alloc : std::allocator<T> {
  template <class U>
  struct rebind {
    using other = noinit_alloc<U>;
  };
  template <class U, class... Args>
  void construct(U *p, Args &&...args)
  {
    if constexpr (sizeof...(Args) == 0) ::new ((void *)p) U;
    else ::new ((void *)p) U(std::forward<Args>(args)...);
  }
};
template <class T>
using hvec = std::vector<T, noinit_alloc<T>>;
// zero-fill on `nthreads` workers (fresh pages: a single-threaded fill of several GB is page-fault bound)
inline void parallel_zero(void *p, size_t bytes, unsigned nthreads)
{
  const size_t nth = std::min<size_t>(std::max(1u, nthreads), std::max<size_t>(1, bytes >> 24));
  if (nth <= 1) {
    if (bytes) std::memset(p, 0, bytes);
    return;
  }
  std::vector<std::thread> th;
  for (size_t t = 0; t < nth; ++t)
    th.emplace_back([=]() {
      const size_t a = bytes * t / nth, b = bytes * (t + 1) / nth;
      std::memset(static_cast<unsigned char *>(p) + a, 0, b - a);
    });
  for (auto &t : th) t.join();
}
