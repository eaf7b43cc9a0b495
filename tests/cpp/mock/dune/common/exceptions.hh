// Minimal stand-in for dune-common's exception machinery: ONLY to compile and run the adaptors of
// dune-ddm_amd/dune/ddm/hip/ in a container without DUNE (tests/cpp).  Not part of the product.
#pragma once
#include <sstream>
#include <stdexcept>
#include <string>
namespace Dune {
class Exception : public std::runtime_error {
public:
  Exception() : std::runtime_error("") {}
  void message(const std::string& m) { msg_ = m; }
  const char* what() const noexcept override { return msg_.c_str(); }
private:
  std::string msg_;
};
class NotImplemented : public Exception {};
class InvalidStateException : public Exception {};
class RangeError : public Exception {};
}  // namespace Dune
#define DUNE_THROW(E, m)               \
  do {                                 \
    E th__ex;                          \
    std::ostringstream th__out;        \
    th__out << m;                      \
    th__ex.message(th__out.str());     \
    throw th__ex;                      \
  } while (0)
