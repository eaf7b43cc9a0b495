// MI355X drop-in for dune/ddm/nonoverlapping_operator.hh: same class names, template parameters and
// constructor signatures; the work is done by libddm_hip.so.  Vectors handed in by dune-istl are host
// BlockVectors, so every call pays one H2D + one D2H copy of n_o doubles -- this adaptor is the
// compatibility layer; the fast path keeps the Krylov vectors on the device (ddm_cg_solve).
#pragma once

#include <memory>

#include <dune/common/parallel/interface.hh>
#include <dune/istl/operators.hh>
#include <dune/istl/scalarproducts.hh>
#include <dune/istl/solvercategory.hh>

#include "backend.hh"

template <class Mat, class X, class Y, class Communication>
class NonOverlappingOperator : public Dune::AssembledLinearOperator<Mat, X, Y>, public ddm_hip::DeviceOperator {
public:
  using domain_type = X;
  using range_type = Y;
  using matrix_type = Mat;
  using communication_type = Communication;
  using field_type = typename X::field_type;

  // reference: nonoverlapping_operator.hh:20
  NonOverlappingOperator(std::shared_ptr<Mat> A, std::shared_ptr<Communication> comm)
      : A(std::move(A)), comm(std::move(comm)), ctx(ddm_hip::Context::get())
  {
    ctx->require(this->comm->communicator());
    dA = std::make_unique<ddm_hip::DeviceCsr>(ctx, *this->A);
    // addOwnerCopyToOwnerCopy: (owner|copy) -> (owner|copy) on the non-overlapping index set
    typename Communication::OwnerCopySet oc;   // Combine<OwnerSet, CopySet>
    Dune::Interface iface;
    iface.build(this->comm->remoteIndices(), oc, oc);
    halo = std::make_unique<ddm_hip::Halo>(ctx, /*tag*/ 1, /*add*/ 1, iface);
    std::vector<std::uint8_t> owner(this->A->N(), 0);
    for (const auto& idx : this->comm->indexSet())
      owner[idx.local().local()] = idx.local().attribute() == Dune::OwnerOverlapCopyAttributeSet::owner;
    ddm_hip::check(ctx->handle(), ddm_op_create(ctx->handle(), dA->handle(), halo->handle(), owner.data(), &op), "ddm_op_create");
    dx = std::make_unique<ddm_hip::DeviceVector>(ctx, this->A->N());
    dy = std::make_unique<ddm_hip::DeviceVector>(ctx, this->A->N());
  }
  NonOverlappingOperator(const NonOverlappingOperator&) = delete;
  NonOverlappingOperator& operator=(const NonOverlappingOperator&) = delete;
  ~NonOverlappingOperator() { ddm_op_destroy(op); }

  Dune::SolverCategory::Category category() const override { return Dune::SolverCategory::nonoverlapping; }

  void apply(const X& x, Y& y) const override   // :34-39
  {
    dx->upload(x);
    ddm_hip::check(ctx->handle(), ddm_op_apply(ctx->handle(), op, dx->data(), dy->data()), "ddm_op_apply");
    dy->download(y);
  }
  void applyscaleadd(field_type alpha, const X& x, Y& y) const override   // :41-50
  {
    dx->upload(x);
    dy->upload(y);
    ddm_hip::check(ctx->handle(), ddm_op_applyscaleadd(ctx->handle(), op, alpha, dx->data(), dy->data()), "ddm_op_applyscaleadd");
    dy->download(y);
  }
  const Mat& getmat() const override { return *A; }
  const communication_type& getCommunication() const { return *comm; }
  std::shared_ptr<communication_type> getCommunicationPtr() const { return comm; }

  // device-side handles for the device-resident solver path
  ddm_op* handle() const { return op; }
  ddm_op* op_handle() const override { return op; }
  std::shared_ptr<ddm_hip::Context> context() const { return ctx; }

private:
  std::shared_ptr<Mat> A;
  std::shared_ptr<communication_type> comm;
  std::shared_ptr<ddm_hip::Context> ctx;
  std::unique_ptr<ddm_hip::DeviceCsr> dA;
  std::unique_ptr<ddm_hip::Halo> halo;
  std::unique_ptr<ddm_hip::DeviceVector> dx, dy;
  ddm_op* op = nullptr;
};

namespace Dune {
template <class X, class C>
class NonOverlappingScalarProduct : public Dune::ScalarProduct<X> {
public:
  using communication_type = C;
  using field_type = typename X::field_type;
  using real_type = field_type;
  // the scalar product shares the operator's device objects (owner mask, context)
  template <class Op>
  explicit NonOverlappingScalarProduct(std::shared_ptr<Op> op_)
      : ctx(op_->context()), op(op_->handle()), keep(op_), n(op_->getmat().N()), dx(ctx, n), dy(ctx, n)
  {
  }
  field_type dot(const X& x, const X& y) const override   // :76-81
  {
    double r = 0;
    dx.upload(x);
    dy.upload(y);
    ddm_hip::check(ctx->handle(), ddm_dot(ctx->handle(), op, dx.data(), dy.data(), &r), "ddm_dot");
    return r;
  }
  real_type norm(const X& x) const override   // :83
  {
    double r = 0;
    dx.upload(x);
    ddm_hip::check(ctx->handle(), ddm_norm(ctx->handle(), op, dx.data(), &r), "ddm_norm");
    return r;
  }
  SolverCategory::Category category() const override { return SolverCategory::nonoverlapping; }

private:
  std::shared_ptr<ddm_hip::Context> ctx;
  ddm_op* op;
  std::shared_ptr<void> keep;
  std::size_t n;
  mutable ddm_hip::DeviceVector dx, dy;
};

// found by ADL from Dune::getSolverFromFactory, as in the reference (:91-95)
template <class M, class X, class Y, class C>
std::shared_ptr<NonOverlappingScalarProduct<X, C>> createScalarProduct(const std::shared_ptr<NonOverlappingOperator<M, X, Y, C>>& op)
{
  return std::make_shared<NonOverlappingScalarProduct<X, C>>(op);
}
}  // namespace Dune
