// The PDELab-facing class TwoLevelSchwarzSolver of dune-ddm_amd/dune/ddm/hip/twolevel_schwarz.hh compiled with HAVE_DUNE_PDELAB=1
// against stand-in PDELab containers and single-process stand-ins of the reference's setup layer (tests/cpp/mock/dune/{pdelab,ddm}),
// and run on one rank: constructor (template vectors by interpolation, constrained DoFs zeroed), two apply() calls (the first creates
// the overlapping objects, the second only refreshes the matrix values), norm(), the result storage.  Same input files and output
// format as twolevel_adaptor.cc, whose results it must reproduce bit for bit.
//   usage: twolevel_pdelab <dir> <mode> <subdomain solver> <krylov>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include <dune/istl/bcrsmatrix.hh>
#include <dune/istl/bvector.hh>
#include <dune/istl/owneroverlapcopy.hh>

#include <dune/ddm/hip/twolevel_schwarz.hh>

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

struct Space {   // what the stand-in interpolate / make_communication ask of a function space
  std::vector<double> xy;
  std::size_t size() const { return xy.size() / 2; }
  std::array<double, 2> position(std::size_t i) const { return {xy[2 * i], xy[2 * i + 1]}; }
};

int main(int argc, char** argv)
{
  if (argc < 5) return 2;
  const std::string dir = argv[1], mode = argv[2], local = argv[3], krylov = argv[4];
  using NVec = Dune::BlockVector<Dune::FieldVector<double, 1>>;
  using NMat = Dune::BCRSMatrix<Dune::FieldMatrix<double, 1, 1>>;
  using Vec = Dune::PDELab::mock::Vector<Space, NVec>;
  using Mat = Dune::PDELab::mock::Matrix<NMat>;
  try {
    auto rp64 = slurp<int64_t>(dir + "/rowptr.bin");
    auto ci32 = slurp<int32_t>(dir + "/col.bin");
    auto va = slurp<double>(dir + "/val.bin");
    auto bb = slurp<double>(dir + "/b.bin");
    Space gfs{slurp<double>(dir + "/coords.bin")};
    const std::size_t n = rp64.size() - 1;
    Mat A(std::make_shared<NMat>(n, n, std::vector<std::size_t>(rp64.begin(), rp64.end()), std::vector<std::size_t>(ci32.begin(), ci32.end()), va));

    Dune::ParameterTree ptree;   // examples/convectiondiffusiondg.ini:5-24
    auto& sub = ptree.sub("twolevelschwarz");
    sub["overlap"] = "1";
    sub["mode"] = mode;
    sub.sub("fine")["type"] = "restricted";
    sub.sub("fine").sub("subdomain_solver")["type"] = local;
    sub.sub("coarse")["type"] = "umfpack";
    sub.sub("solver")["type"] = krylov;
    sub.sub("solver")["maxit"] = "300";
    sub.sub("solver")["restart"] = "50";
    sub.sub("solver")["verbose"] = "0";

    const std::vector<std::size_t> no_constraints;
    TwoLevelSchwarzSolver<Mat, Vec> backend(gfs, no_constraints, ptree);                       // default sub-tree name, additive input
    TwoLevelSchwarzSolver<Mat, Vec> backend2(gfs, no_constraints, ptree, "twolevelschwarz", false);   // (make_additive path: compile + run)
    for (int call = 0; call < 2; ++call) {
      Vec z(gfs), r(gfs);
      for (std::size_t i = 0; i < n; ++i) r.native()[i] = bb[i];
      backend.apply(A, z, r, 1e-8);
      const auto& st = backend.result();
      std::printf("solve %d iterations %u converged %d reduction %.17g norm_z %.17g\n", call, st.iterations, (int)st.converged, (double)st.reduction, (double)backend.norm(z));
      std::ofstream out(dir + "/zp" + std::to_string(call) + ".bin", std::ios::binary);
      for (std::size_t i = 0; i < n; ++i) { const double v = z.native()[i][0]; out.write(reinterpret_cast<const char*>(&v), 8); }
    }
    {
      Vec z(gfs), r(gfs);
      for (std::size_t i = 0; i < n; ++i) r.native()[i] = bb[i];
      backend2.apply(A, z, r, 1e-8);
      std::printf("nonadditive_input iterations %u converged %d\n", backend2.result().iterations, (int)backend2.result().converged);
    }
    // constrained DoFs: the template vectors are zero there (checked through the coarse basis being built without error and the
    // solver still converging with the first row constrained)
    const std::vector<std::size_t> first_row = {0};
    TwoLevelSchwarzSolver<Mat, Vec> backend3(gfs, first_row, ptree);
    Vec z(gfs), r(gfs);
    for (std::size_t i = 0; i < n; ++i) r.native()[i] = bb[i];
    backend3.apply(A, z, r, 1e-8);
    std::printf("constrained iterations %u converged %d\n", backend3.result().iterations, (int)backend3.result().converged);
  } catch (Dune::Exception& e) {
    std::cerr << "Dune exception: " << e.what() << "\n";
    return 1;
  }
  return 0;
}
