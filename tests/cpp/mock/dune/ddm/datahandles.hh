#pragma once
// Single-process stand-ins for the matrix data handles of the reference (dune/ddm/datahandles.hh:436-591): the overlapping matrix of a
// lone rank is its own matrix.  Same constructor arguments and accessor as the originals.
template <class Mat, class ParallelIndexSet>
class CreateMatrixDataHandle {
public:
  CreateMatrixDataHandle(const Mat& A, const ParallelIndexSet&) : A(A) {}
  CreateMatrixDataHandle(const CreateMatrixDataHandle&) = delete;
  void single_rank_pass() { made = true; }
  Mat getOverlappingMatrix() { return Mat(A); }
private:
  const Mat& A;
  bool made = false;
};
template <class Mat, class ParallelIndexSet>
class AddMatrixDataHandle {
public:
  AddMatrixDataHandle(const Mat& A, Mat& Aovlp, const ParallelIndexSet&) : A(A), Aovlp(Aovlp) {}
  AddMatrixDataHandle(const AddMatrixDataHandle&) = delete;
  void single_rank_pass() { Aovlp.copy_values_from(A); }
private:
  const Mat& A;
  Mat& Aovlp;
};
