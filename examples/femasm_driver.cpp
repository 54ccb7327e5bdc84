// The reference drivers' flow in compiled code on top of the C ABI alone (no Python, no torch, no PETSc):
//
//   mesh (DMDACreate3d + DMDASetUniformCoordinates, P3:658-663 / S3:1185-1198)
//   -> properties fill (S3:1243-1363)                       femasm_fill_properties
//   -> DMCreateMatrix pattern (P3:667 / S3:1373)            femasm_pattern_copy
//   -> AssembleA_* + AssembleF_* + Dirichlet rows           femasm_assemble_system   (P3:675-678 / S3:1384-1392)
//   -> KSPSolve (P3:688-705 / S3:1399-1416)                 a plain Jacobi-CG below: the solve is NOT part of femasm
//   -> DMDAIntegrateErrors3D (P3:416-532 / S3:1041-1151)    femasm_error_integrals
//
// and the reference's only known-answer check: the L2 error of the manufactured solution falls ~4x per halving of h.
// Every array handed to femasm here is a HOST array, as in the unmodified drivers.
//
//   femasm_driver [poisson3d|elastic3d] [nel ...]      exit code 0 = every consecutive L2 ratio in [3,5]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "femasm.h"

#define CHK(call)                                                                     \
  do {                                                                                \
    if ((call) != FEMASM_OK) {                                                        \
      std::fprintf(stderr, "%s failed: %s\n", #call, femasm_last_error());            \
      std::exit(2);                                                                   \
    }                                                                                 \
  } while (0)

struct Csr {
  int64_t nrows = 0, nnz = 0;
  std::vector<int64_t> rowptr;
  std::vector<int32_t> colidx;
  std::vector<double> a;
};

// y = A x restricted to the free rows; x carries the Dirichlet values (or zeros) in the other entries
static void spmv_free(const Csr &A, const std::vector<char> &fixed, const std::vector<double> &x, std::vector<double> &y) {
  for (int64_t r = 0; r < A.nrows; r++) {
    double s = 0.0;
    if (!fixed[r])
      for (int64_t q = A.rowptr[r]; q < A.rowptr[r + 1]; q++) s += A.a[q] * x[A.colidx[q]];
    y[r] = s;
  }
}

// The drivers zero ROWS only (MatZeroRows, P3:286-290): x_D = b_D / diag is known and the free unknowns solve the
// symmetric positive definite system K_FF x_F = b_F - K_FD x_D.  Jacobi-preconditioned CG on it.
static int solve(const Csr &A, const std::vector<char> &fixed, const std::vector<double> &b, std::vector<double> &x) {
  const int64_t n = A.nrows;
  std::vector<double> r(n), z(n), p(n), q(n), dinv(n, 0.0);
  for (int64_t i = 0; i < n; i++) {
    double d = 0.0;
    for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; k++)
      if (A.colidx[k] == i) d = A.a[k];
    x[i] = fixed[i] ? b[i] / d : 0.0;
    dinv[i] = fixed[i] ? 0.0 : 1.0 / d;
  }
  spmv_free(A, fixed, x, q);
  double rz = 0.0, r0 = 0.0;
  for (int64_t i = 0; i < n; i++) {
    r[i] = fixed[i] ? 0.0 : b[i] - q[i];
    z[i] = dinv[i] * r[i];
    p[i] = z[i];
    rz += r[i] * z[i];
    r0 += r[i] * r[i];
  }
  int it = 0;
  for (; it < 20000 && rz > 0.0; it++) {
    spmv_free(A, fixed, p, q);
    double pq = 0.0;
    for (int64_t i = 0; i < n; i++) pq += p[i] * q[i];
    const double alpha = rz / pq;
    double rr = 0.0, rz1 = 0.0;
    for (int64_t i = 0; i < n; i++) {
      x[i] += alpha * p[i];
      r[i] -= alpha * q[i];
      z[i] = dinv[i] * r[i];
      rr += r[i] * r[i];
      rz1 += r[i] * z[i];
    }
    if (rr <= 1e-24 * r0) { it++; break; }
    const double beta = rz1 / rz;
    rz = rz1;
    for (int64_t i = 0; i < n; i++) p[i] = z[i] + beta * p[i];
  }
  return it;
}

static double run(int kind, int64_t nel) {
  const int dof = kind == FEMASM_ELASTIC3D_Q1 ? 3 : 1;
  const int64_t M = nel + 1, nn = M * M * M;
  std::vector<double> xyz(3 * nn);
  const double h = (1.0 - 0.0) / (double)(M - 1);  // DMDASetUniformCoordinates: min + h*i
  for (int64_t k = 0, c = 0; k < M; k++)
    for (int64_t j = 0; j < M; j++)
      for (int64_t i = 0; i < M; i++) { xyz[c++] = 0.0 + h * i; xyz[c++] = 0.0 + h * j; xyz[c++] = 0.0 + h * k; }

  femasm_mesh mesh{kind, 0, M, M, M, 0, M, 0, M, xyz.data()};
  femasm_quad quad{8, 0};
  femasm_plan *plan = nullptr;
  CHK(femasm_plan_create(&mesh, &quad, 0, &plan));

  Csr A;
  CHK(femasm_pattern(plan, &A.nrows, &A.nnz, nullptr, nullptr));
  A.rowptr.resize(A.nrows + 1);
  A.colidx.resize(A.nnz);
  A.a.assign(A.nnz, 0.0);
  CHK(femasm_pattern_copy(plan, A.rowptr.data(), A.colidx.data()));

  std::vector<double> props, f(A.nrows, 0.0);
  femasm_material mat{};
  femasm_bc bc{kind == FEMASM_ELASTIC3D_Q1 ? FEMASM_BC_S3_ALLFACES_EXACT : FEMASM_BC_POISSON_INLOOP, 0};
  if (kind == FEMASM_ELASTIC3D_Q1) {
    props.resize((size_t)nel * nel * nel * 400);
    CHK(femasm_fill_properties(plan, 90.0, 0.28, 400, props.data()));
    mat.layout = FEMASM_MAT_AOS;
    mat.stride = 400;
    mat.e_ks = 0;
    mat.gp = props.data();
  }
  femasm_stats st{};
  CHK(femasm_assemble_system(plan, kind == FEMASM_ELASTIC3D_Q1 ? &mat : nullptr, &bc, A.a.data(), f.data(), &st));

  std::vector<char> fixed(A.nrows, 0);
  for (int64_t k = 0, n = 0; k < M; k++)
    for (int64_t j = 0; j < M; j++)
      for (int64_t i = 0; i < M; i++, n++) {
        const bool bnd = i == 0 || i == M - 1 || j == 0 || j == M - 1 || k == 0 || k == M - 1;
        for (int c = 0; c < dof; c++) fixed[n * dof + c] = bnd;
      }
  std::vector<double> u(A.nrows);
  const int its = solve(A, fixed, f, u);

  double sums[2] = {0.0, 0.0};
  CHK(femasm_error_integrals(plan, u.data(), sums));
  CHK(femasm_plan_destroy(plan));
  std::printf("%s nel=%lld rows=%lld nnz=%lld | assembly kernels %.3f ms (K) + %.3f ms (b), staging %.1f ms | CG its %d | "
              "L2 error %.6e  H1 error %.6e\n",
              kind == FEMASM_ELASTIC3D_Q1 ? "elastic3d" : "poisson3d", (long long)nel, (long long)A.nrows,
              (long long)A.nnz, st.ms_kernel_K, st.ms_kernel_rhs, st.ms_staging, its, std::sqrt(sums[0]), std::sqrt(sums[1]));
  return std::sqrt(sums[0]);
}

int main(int argc, char **argv) {
  int kind = FEMASM_ELASTIC3D_Q1;
  std::vector<int64_t> nels;
  for (int i = 1; i < argc; i++) {
    if (!std::strcmp(argv[i], "poisson3d")) kind = FEMASM_POISSON3D_Q1;
    else if (!std::strcmp(argv[i], "elastic3d")) kind = FEMASM_ELASTIC3D_Q1;
    else nels.push_back(std::atoll(argv[i]));
  }
  if (nels.empty()) nels = {4, 8, 16};
  std::printf("%s\n", femasm_version());
  int bad = 0;
  double prev = 0.0;
  for (size_t t = 0; t < nels.size(); t++) {
    const double e = run(kind, nels[t]);
    if (t > 0 && nels[t] == 2 * nels[t - 1]) {
      const double ratio = prev / e;
      std::printf("  L2 error ratio %lld -> %lld: %.3f (O(h^2): 4)\n", (long long)nels[t - 1], (long long)nels[t], ratio);
      if (!(ratio >= 3.0 && ratio <= 5.0)) bad = 1;
    }
    prev = e;
  }
  return bad;
}
