// Drives the DUNE-facing adaptors (dune-ddm_amd/dune/ddm/hip/*.hh) the way examples/poisson.cc:229-321
// drives the reference classes: SchwarzPreconditioner + (POU) GalerkinPreconditioner +
// CombinedPreconditioner + NonOverlappingOperator + its scalar product inside a CG loop written
// against the abstract dune-istl interfaces.  Single rank (mock communication, see mock/).
//   usage: poisson_adaptor <dir with n.txt rowptr.bin col.bin val.bin b.bin dirichlet.bin pou.bin> <mode>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include <dune/istl/bcrsmatrix.hh>
#include <dune/istl/bvector.hh>
#include <dune/istl/owneroverlapcopy.hh>

#include <dune/ddm/hip/combined_preconditioner.hh>
#include <dune/ddm/hip/galerkin_preconditioner.hh>
#include <dune/ddm/hip/nonoverlapping_operator.hh>
#include <dune/ddm/hip/schwarz.hh>

template <class T>
static std::vector<T> slurp(const std::string& f)
{
  std::ifstream in(f, std::ios::binary | std::ios::ate);
  if (!in) { std::cerr << "cannot open " << f << "\n"; std::exit(2); }
  const std::size_t bytes = in.tellg();
  in.seekg(0);
  std::vector<T> v(bytes / sizeof(T));
  in.read(reinterpret_cast<char*>(v.data()), bytes);
  return v;
}

int main(int argc, char** argv)
{
  if (argc < 3) return 2;
  const std::string dir = argv[1], mode = argv[2];
  using Vec = Dune::BlockVector<Dune::FieldVector<double, 1>>;
  using Mat = Dune::BCRSMatrix<Dune::FieldMatrix<double, 1, 1>>;
  using Comm = Dune::OwnerOverlapCopyCommunication<std::size_t, int>;
  try {
    auto rp64 = slurp<int64_t>(dir + "/rowptr.bin");
    auto ci32 = slurp<int32_t>(dir + "/col.bin");
    auto va = slurp<double>(dir + "/val.bin");
    auto bb = slurp<double>(dir + "/b.bin");
    auto dm = slurp<unsigned char>(dir + "/dirichlet.bin");
    auto pw = slurp<double>(dir + "/pou.bin");
    const std::size_t n = rp64.size() - 1;
    auto A = std::make_shared<Mat>(n, n, std::vector<std::size_t>(rp64.begin(), rp64.end()), std::vector<std::size_t>(ci32.begin(), ci32.end()), va);
    auto comm = std::make_shared<Comm>();
    for (std::size_t i = 0; i < n; ++i) comm->indexSet().v.push_back({i, {i, Dune::OwnerOverlapCopyAttributeSet::owner}});

    Dune::ParameterTree ptree;
    ptree.sub("schwarz")["type"] = "standard";
    ptree.sub("schwarz").sub("subdomain_solver")["type"] = "ilu0";
    ptree.sub("combined_preconditioner")["mode"] = mode;

    auto pou = std::make_shared<PartitionOfUnity>(pw);
    auto schwarz = std::make_shared<SchwarzPreconditioner<Mat, Vec, Comm>>(A, comm, pou, ptree);
    // POUCoarseSpace + zero_at_dirichlet (coarse_spaces.hh:1211-1224, poisson.cc:235-238)
    std::vector<Vec> basis(1, Vec(n));
    double nrm = 0;
    for (std::size_t i = 0; i < n; ++i) nrm += pw[i] * pw[i];
    for (std::size_t i = 0; i < n; ++i) basis[0][i] = dm[i] ? 0.0 : pw[i] / std::sqrt(nrm);
    auto coarse = std::make_shared<GalerkinPreconditioner<Vec, Comm>>(*A, basis, comm, ptree, "coarse_solver");
    auto op = std::make_shared<NonOverlappingOperator<Mat, Vec, Vec, Comm>>(A, comm);
    auto prec = std::make_shared<CombinedPreconditioner<Vec>>(ptree);
    prec->set_op(op);
    prec->add(schwarz);
    prec->add(coarse);
    auto sp = Dune::createScalarProduct(op);

    // dune-istl CGSolver::apply on the abstract interfaces
    Dune::LinearOperator<Vec, Vec>& L = *op;
    Dune::Preconditioner<Vec, Vec>& P = *prec;
    Vec x(n), b(n), p(n), q(n);
    x = 0;
    for (std::size_t i = 0; i < n; ++i) b[i] = bb[i];
    P.pre(x, b);
    L.applyscaleadd(-1.0, x, b);
    const double def0 = sp->norm(b);
    std::printf("it 0 %.17g\n", def0);
    p = 0;
    P.apply(p, b);
    double rholast = sp->dot(p, b);
    for (int i = 1; i <= 500; ++i) {
      L.apply(p, q);
      const double lambda = rholast / sp->dot(p, q);
      x.axpy(lambda, p);
      b.axpy(-lambda, q);
      const double def = sp->norm(b);
      std::printf("it %d %.17g\n", i, def);
      if (def < def0 * 1e-10) break;
      q = 0;
      P.apply(q, b);
      const double rho = sp->dot(q, b);
      p *= rho / rholast;
      p += q;
      rholast = rho;
    }
    P.post(x);
    std::printf("coarse %d %.17g\n", coarse->coarse_size(), coarse->coarse_matrix()[0]);
    // error conventions: unknown Schwarz type / missing solver key throw like the reference
    int caught = 0;
    try {
      Dune::ParameterTree bad;
      bad.sub("schwarz")["type"] = "bogus";
      bad.sub("schwarz").sub("subdomain_solver")["type"] = "ilu0";
      SchwarzPreconditioner<Mat, Vec, Comm> s(A, comm, pou, bad);
    } catch (Dune::NotImplemented&) { ++caught; }
    try {
      Dune::ParameterTree bad;
      SchwarzPreconditioner<Mat, Vec, Comm> s(A, comm, pou, bad);
    } catch (Dune::Exception&) { ++caught; }
    try {
      Dune::ParameterTree bad;
      bad.sub("combined_preconditioner")["mode"] = "bogus";
      CombinedPreconditioner<Vec> c(bad);
    } catch (Dune::NotImplemented&) { ++caught; }
    std::printf("errors_caught %d\n", caught);
  } catch (Dune::Exception& e) {
    std::cerr << "Dune exception: " << e.what() << "\n";
    return 1;
  }
  return 0;
}
