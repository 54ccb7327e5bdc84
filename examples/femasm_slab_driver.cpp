// femasm_slab_driver.cpp -- the multi-GPU slab path from compiled code on the C ABI alone, as an MPI driver with one
// rank per GPU would use it, rehearsed on ONE GPU: two z-slabs of the mesh, one plan each, the interface rows moved
// by femasm_interface_exchange over RCCL (a one-rank communicator: the upper slab's send and the lower slab's receive
// are a send to and a receive from this same rank, grouped), overlapped with the bulk of the assembly through the
// split call, then femasm_interface_apply.  The two slabs' rows are compared with a one-plan assembly of the mesh.
// This is what replaces MatAssemblyBegin/End's stash and DMLocalToGlobal(ADD_VALUES) (S3:710-711, 794-795;
// P3:345-346, 638-639) in a patched driver.
//   usage: femasm_slab_driver poisson3d|elastic3d nel        exit code 0 = slabs equal the global system (<= 1e-13)
#include "femasm.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(call)                                                                  \
  do {                                                                             \
    int rc_ = (call);                                                              \
    if (rc_) {                                                                     \
      fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #call, rc_, femasm_last_error()); \
      return 2;                                                                    \
    }                                                                              \
  } while (0)
#define HCHK(call)                                                                 \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return 2;                                                                    \
    }                                                                              \
  } while (0)
#define NCHK(call)                                                                 \
  do {                                                                             \
    ncclResult_t r_ = (call);                                                      \
    if (r_ != ncclSuccess) {                                                       \
      fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, ncclGetErrorString(r_)); \
      return 2;                                                                    \
    }                                                                              \
  } while (0)

int main(int argc, char **argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s poisson3d|elastic3d nel\n", argv[0]);
    return 2;
  }
  const bool elastic = strcmp(argv[1], "elastic3d") == 0;
  const int nel = atoi(argv[2]);
  const int64_t M = nel + 1, P = M;
  const int kind = elastic ? FEMASM_ELASTIC3D_Q1 : FEMASM_POISSON3D_Q1;
  const int dof = elastic ? 3 : 1;
  const int bckind = elastic ? FEMASM_BC_S3_ALLFACES_EXACT : FEMASM_BC_POISSON_INLOOP;

  // DMDASetUniformCoordinates, interior nodes moved by a smooth offset: the general kernels run
  std::vector<double> xyz((size_t)3 * M * M * P);
  const double h = 1.0 / (M - 1);
  for (int64_t k = 0; k < P; k++)
    for (int64_t j = 0; j < M; j++)
      for (int64_t i = 0; i < M; i++) {
        double *c = &xyz[3 * (i + M * (j + M * k))];
        c[0] = h * i; c[1] = h * j; c[2] = h * k;
        if (i > 0 && i < M - 1 && j > 0 && j < M - 1 && k > 0 && k < P - 1) {
          c[0] += 0.15 * h * sin(7 * c[1] + 5 * c[2]);
          c[1] += 0.15 * h * sin(7 * c[2] + 5 * h * i + 1);
          c[2] += 0.15 * h * sin(7 * h * i + 5 * h * j + 2);
        }
      }
  femasm_material mat;
  memset(&mat, 0, sizeof mat);
  mat.layout = FEMASM_MAT_UNIFORM;
  mat.E = 90.0;
  mat.nu = 0.28;
  femasm_bc bc = {bckind, 0};
  femasm_quad quad = {8, 0};

  // ---- the whole mesh as one plan (host arrays, staged by the library) ----
  femasm_mesh mesh;
  memset(&mesh, 0, sizeof mesh);
  mesh.problem = kind;
  mesh.M = mesh.N = M; mesh.P = P;
  mesh.ks = 0; mesh.ke = P; mesh.cks = 0; mesh.cke = P;
  mesh.coords = xyz.data();
  femasm_plan *g = nullptr;
  CHK(femasm_plan_create(&mesh, &quad, 0, &g));
  int64_t nrows = 0, nnz = 0;
  CHK(femasm_pattern(g, &nrows, &nnz, nullptr, nullptr));
  std::vector<double> K(nnz), b(nrows);
  CHK(femasm_assemble_system(g, &mat, &bc, K.data(), b.data(), nullptr));
  CHK(femasm_plan_destroy(g));

  // ---- two slabs by the DMDA rule (P/p + (P%p > r) node layers), device-resident outputs ----
  const int64_t n0 = P / 2 + (P % 2 > 0 ? 1 : 0);
  const int64_t ks[2] = {0, n0}, ke[2] = {n0, P};
  femasm_plan *pl[2] = {nullptr, nullptr};
  double *dK[2] = {nullptr, nullptr}, *db[2] = {nullptr, nullptr};
  int64_t sn[2], sz[2];
  for (int r = 0; r < 2; r++) {
    femasm_mesh m = mesh;
    m.ks = ks[r]; m.ke = ke[r];
    m.cks = ks[r] > 0 ? ks[r] - 1 : 0;
    m.cke = ke[r] < P ? ke[r] + 1 : P;
    m.coords = xyz.data() + 3 * M * M * m.cks;  // the ghosted local coordinate vector of this rank
    CHK(femasm_plan_create(&m, &quad, 0, &pl[r]));
    CHK(femasm_pattern(pl[r], &sn[r], &sz[r], nullptr, nullptr));
    HCHK(hipMalloc((void **)&dK[r], sz[r] * sizeof(double)));
    HCHK(hipMalloc((void **)&db[r], sn[r] * sizeof(double)));
  }
  int dev0 = 0;
  ncclComm_t comm;
  NCHK(ncclCommInitAll(&comm, 1, &dev0));

  // upper slab: the rows owed to the rank below first, their transfer queued, then the bulk of the assembly
  CHK(femasm_assemble_interface(pl[1], &mat, &bc));
  NCHK(ncclGroupStart());  // (one rank plays both ends here: its send and its receive have to be in one group)
  CHK(femasm_interface_exchange(pl[1], comm, /*rank_below=*/0, /*rank_above=*/-1));
  CHK(femasm_interface_exchange(pl[0], comm, /*rank_below=*/-1, /*rank_above=*/0));
  NCHK(ncclGroupEnd());
  CHK(femasm_assemble_system(pl[1], &mat, &bc, dK[1], db[1], nullptr));
  CHK(femasm_assemble_system(pl[0], &mat, &bc, dK[0], db[0], nullptr));
  for (int r = 0; r < 2; r++) CHK(femasm_interface_apply(pl[r], &bc, dK[r], db[r]));
  for (int r = 0; r < 2; r++) CHK(femasm_plan_sync(pl[r]));

  if (sz[0] + sz[1] != nnz || sn[0] + sn[1] != nrows) {
    fprintf(stderr, "slab sizes do not add up\n");
    return 1;
  }
  std::vector<double> Ks(nnz), bs(nrows);
  HCHK(hipMemcpy(Ks.data(), dK[0], sz[0] * sizeof(double), hipMemcpyDeviceToHost));
  HCHK(hipMemcpy(Ks.data() + sz[0], dK[1], sz[1] * sizeof(double), hipMemcpyDeviceToHost));
  HCHK(hipMemcpy(bs.data(), db[0], sn[0] * sizeof(double), hipMemcpyDeviceToHost));
  HCHK(hipMemcpy(bs.data() + sn[0], db[1], sn[1] * sizeof(double), hipMemcpyDeviceToHost));
  double nK = 0, dKn = 0, nb = 0, dbn = 0;
  for (int64_t t = 0; t < nnz; t++) { nK += (Ks[t] - K[t]) * (Ks[t] - K[t]); dKn += K[t] * K[t]; }
  for (int64_t t = 0; t < nrows; t++) { nb += (bs[t] - b[t]) * (bs[t] - b[t]); dbn += b[t] * b[t]; }
  const double relK = sqrt(nK / dKn), relb = dbn > 0 ? sqrt(nb / dbn) : sqrt(nb);
  int64_t sc = 0, rc = 0;
  CHK(femasm_interface_buffers(pl[1], nullptr, &sc, nullptr, nullptr));
  CHK(femasm_interface_buffers(pl[0], nullptr, nullptr, nullptr, &rc));
  printf("%s nel=%d dof=%d: 2 slabs over RCCL, interface message %lld doubles (%.2f MB): rel K %.3e  rel b %.3e\n",
         argv[1], nel, dof, (long long)sc, 8.0 * sc / 1e6, relK, relb);
  for (int r = 0; r < 2; r++) {
    CHK(femasm_plan_destroy(pl[r]));
    (void)hipFree(dK[r]);
    (void)hipFree(db[r]);
  }
  ncclCommDestroy(comm);
  return (sc == rc && relK <= 1e-13 && relb <= 1e-13) ? 0 : 1;
}
