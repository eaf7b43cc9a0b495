#pragma once
namespace Dune {
template <class T>
struct FieldTraits {
  using field_type = T;
  using real_type = T;
};
}  // namespace Dune
