#pragma once
namespace Dune {
struct SolverCategory {
  enum Category { sequential, nonoverlapping, overlapping };
};
}  // namespace Dune
