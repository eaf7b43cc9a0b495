// Compile check of the MPI exchange callbacks a DUNE build installs (dune-ddm_amd/dune/ddm/hip/mpi_exchange.hh) against the MPI
// headers of this image; instantiates install_mpi_exchange so that every line is type-checked.  Not linked, not run.
#include <dune/ddm/hip/mpi_exchange.hh>

std::shared_ptr<ddm_hip::Context> ddm_mpi_exchange_check(MPI_Comm comm) { return ddm_hip::install_mpi_exchange(comm); }
