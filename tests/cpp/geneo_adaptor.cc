// GenEOCoarseSpace adaptor (dune-ddm_amd/dune/ddm/hip/coarse_spaces.hh) driven like examples/poisson.cc:270-295 drives the
// reference's: constructed from (A_neu, B_neu, pou, ptree, taskflow), executed through the taskflow, get_basis() consumed by a
// follow-up task.  One rank = one subdomain: the matrices of one subdomain of a 2 x 2 x 2 decomposition are read from files.
//   usage: geneo_adaptor <dir with A_{rowptr,col,val}.bin B_{rowptr,col,val}.bin pou.bin> <nev>   -> basis.bin, prints eigenvalues
#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include <dune/istl/bcrsmatrix.hh>
#include <dune/istl/bvector.hh>

#include <dune/ddm/hip/coarse_spaces.hh>

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
  const std::string dir = argv[1];
  using Vec = Dune::BlockVector<Dune::FieldVector<double, 1>>;
  using Mat = Dune::BCRSMatrix<Dune::FieldMatrix<double, 1, 1>>;
  try {
    auto load = [&](const std::string& pre) {
      auto rp = slurp<int64_t>(dir + "/" + pre + "_rowptr.bin");
      auto ci = slurp<int32_t>(dir + "/" + pre + "_col.bin");
      auto va = slurp<double>(dir + "/" + pre + "_val.bin");
      const std::size_t n = rp.size() - 1;
      return std::make_shared<const Mat>(n, n, std::vector<std::size_t>(rp.begin(), rp.end()), std::vector<std::size_t>(ci.begin(), ci.end()), va);
    };
    auto A = load("A"), B = load("B");
    auto pou = std::make_shared<const PartitionOfUnity>(slurp<double>(dir + "/pou.bin"));
    Dune::ParameterTree ptree;
    ptree.sub("geneo").sub("eigensolver")["nev"] = argv[2];
    tf::Taskflow taskflow("Main taskflow");
    std::unique_ptr<CoarseSpaceBuilder<Vec>> coarse_space = std::make_unique<GenEOCoarseSpace<Mat, Vec>>(A, B, pou, ptree, taskflow);
    std::size_t got = 0;
    auto consume = taskflow.emplace([&]() { got = coarse_space->get_basis().size(); });
    consume.succeed(coarse_space->get_setup_task());
    tf::Executor executor(1);
    executor.run(taskflow).get();
    auto* g = dynamic_cast<GenEOCoarseSpace<Mat, Vec>*>(coarse_space.get());
    std::printf("size %zu consumed %zu iterations %d direct %d\n", coarse_space->size(), got, g->info().iterations, g->info().used_direct);
    for (double l : g->eigenvalues()) std::printf("lambda %.17g\n", l);
    std::ofstream out(dir + "/basis.bin", std::ios::binary);
    for (const auto& v : coarse_space->get_basis())
      for (std::size_t i = 0; i < v.N(); ++i) { const double x = v[i][0]; out.write(reinterpret_cast<const char*>(&x), 8); }
    // error conventions (coarse_spaces.hh:323, eigensolver_params.hh:35)
    int caught = 0;
    try {
      GenEOCoarseSpace<Mat, Vec> bad;
      bad.setup_geneo_impl(A, B, std::make_shared<const PartitionOfUnity>(std::vector<double>(3, 1.0)), ptree.sub("geneo").sub("eigensolver"));
    } catch (Dune::Exception&) { ++caught; }
    try {
      Dune::ParameterTree e;
      e["type"] = "arpack";
      GenEOCoarseSpace<Mat, Vec> bad;
      bad.setup_geneo_impl(A, B, pou, e);
    } catch (Dune::NotImplemented&) { ++caught; }
    std::printf("errors_caught %d\n", caught);
  } catch (Dune::Exception& e) {
    std::cerr << "Dune exception: " << e.what() << "\n";
    return 1;
  }
  return 0;
}
