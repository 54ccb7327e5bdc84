// femasm_kernels.h -- hand-written gfx950 (CDNA4, wave64) kernels of the assembly engine.
//
// Design (DESIGN.md has the long form):
//  * K kernel = deterministic GATHER.  A workgroup owns a brick of 64 nodes and the
//    (halo'd) brick of elements around it.  Per Gauss point, the first NE threads build
//    one geometry record per element (physical shape-function gradients + w*detJ-scaled
//    material coefficients) into LDS, double-buffered; then every thread contracts the
//    records of the elements that touch ITS node into register accumulators.
//    A thread is (node, type): the 4 waves of the workgroup are the 4 combinations
//    (element layer below/above the node) x (lower/upper face of that element) in the
//    slowest direction, so each thread owns one third of its node's stencil
//    (3^(nsd-1) neighbour blocks of dof x dof), accumulates 2^(nsd-1) elements x
//    2^(nsd-1) local nodes, and no CSR entry is ever touched by two threads except the
//    middle third, whose two halves are summed in a fixed order through LDS.
//    Every CSR value is written exactly once and never read: traffic = algorithmic bytes.
//  * The element->nonzero scatter map is the closed form of femasm_device.h.
//  * RHS = per-element Fe/Ge into a small scratch array, then a per-node gather in the
//    reference's element order (k->j->i).
#pragma once
#include "femasm_device.h"
#include <type_traits>
// Profiling builds (never shipped: the build of __graft_entry__ does not define it).  -DFEMASM_EXP=n removes one
// phase of a kernel so that its cost can be read off a timing (DESIGN.md section 3 quotes these):
//   1  the global stores of the stiffness kernels (values are consumed by an empty asm)
//   2  the contraction of the general kernels / direct lane-wise stores instead of the LDS transposition
//   3  the geometry phase of the general kernels; in the row-sweep and RHS kernels: per-phase cycle counters
//      (PH_* macros, dumped by femasm_plan_destroy)
//   5  the b stores of the one-pass k_assemble_e2_rect<true>
//   7  the record stores of k_rhs_elem_elastic_runs;  8  also its arithmetic (input stream + LDS commit only)
// femasm_hex2.h has H2_EXP and femasm_p3hex.h PH_EXP (listed there), under the same rule.
// femasm_enriched.h: 1 removes the ring stores of k_enr_elem (the VALU-versus-MFMA contraction-only timing);
//   k_enr_node: 11 no global stores, 12 no staging/rounds, 13 no contraction, 14 all three (phase 1 + block zero/read)
#ifndef FEMASM_EXP
#define FEMASM_EXP 0
#endif
#if (FEMASM_EXP != 0 || defined(H2_EXP) || defined(PH_EXP)) && !defined(FEMASM_DEV)
#error "profiling branches (-DFEMASM_EXP=n, -DH2_EXP=n, -DPH_EXP=n) are compiled only into development builds: add -DFEMASM_DEV"
#endif

namespace femasm {

enum { KIND_POISSON = 0, KIND_PLANE = 1, KIND_ELASTIC = 2 };
enum { BC_NONE = 0, BC_POISSON = 1, BC_E2 = 2, BC_S3 = 3 };

template <int KIND> struct Traits;
template <> struct Traits<KIND_POISSON> {
  static constexpr int NSD = 3, DOF = 1, NEN = 8, NGP = 8, NCOEF = 1;
  static constexpr int BX = 4, BY = 4, BZ = 4;
  static constexpr int NFE = 8;  // doubles of per-element RHS scratch
};
template <> struct Traits<KIND_PLANE> {
  static constexpr int NSD = 2, DOF = 2, NEN = 4, NGP = 4, NCOEF = 3;
  static constexpr int BX = 16, BY = 4, BZ = 1;
  static constexpr int NFE = 8;
};
template <> struct Traits<KIND_ELASTIC> {
  static constexpr int NSD = 3, DOF = 3, NEN = 8, NGP = 8, NCOEF = 3;
  static constexpr int BX = 4, BY = 4, BZ = 4;
  static constexpr int NFE = 48;  // Fe[24] then Ge[24]
};

// ---------------------------------------------------------------------------
// Reference-element tables, evaluated at compile time in the reference's operation
// order (A1, A2, A5): literal abscissa 0.57735026919, unit weights.
// ---------------------------------------------------------------------------
constexpr double GAUSS_G = 0.57735026919;
constexpr double HSX[8] = {-1, -1, 1, 1, -1, -1, 1, 1};
constexpr double HSY[8] = {-1, 1, 1, -1, -1, 1, 1, -1};
constexpr double HSZ[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
constexpr double QSX[4] = {-1, -1, 1, 1};
constexpr double QSY[4] = {-1, 1, 1, -1};

struct Tab3 {
  double G[8][3][8];   // dN_n/dxi_k at Gauss point p : G[p][k][n]
  double N[8][8];      // N_n at Gauss point p
  double NF[24][8];    // N_n at face Gauss point p (S3:253-289)
};
struct Tab2 {
  double G[4][2][4];
  double N[4][4];
};

constexpr Tab3 make_tab3() {
  Tab3 t{};
  for (int p = 0; p < 8; p++) {
    const double xi = HSX[p] * GAUSS_G, eta = HSY[p] * GAUSS_G, zeta = HSZ[p] * GAUSS_G;
    for (int n = 0; n < 8; n++) {
      const double ax = 1.0 + HSX[n] * xi, ay = 1.0 + HSY[n] * eta, az = 1.0 + HSZ[n] * zeta;
      t.G[p][0][n] = (HSX[n] * 0.125) * ay * az;
      t.G[p][1][n] = (HSY[n] * 0.125) * ax * az;
      t.G[p][2][n] = (HSZ[n] * 0.125) * ax * ay;
      t.N[p][n] = 0.125 * ax * ay * az;
    }
  }
  const double s0[4] = {-1, 1, -1, 1}, s1[4] = {-1, -1, 1, 1};
  for (int f = 0; f < 6; f++) {
    const int axis = f / 2;
    const double side = (f % 2 == 0) ? 1.0 : -1.0;
    const int a0 = (axis == 0) ? 1 : 0, a1 = (axis == 2) ? 1 : 2;
    for (int q = 0; q < 4; q++) {
      double x[3] = {0, 0, 0};
      x[axis] = side;
      x[a0] = s0[q] * GAUSS_G;
      x[a1] = s1[q] * GAUSS_G;
      for (int n = 0; n < 8; n++)
        t.NF[4 * f + q][n] = 0.125 * (1.0 + HSX[n] * x[0]) * (1.0 + HSY[n] * x[1]) * (1.0 + HSZ[n] * x[2]);
    }
  }
  return t;
}
constexpr Tab2 make_tab2() {
  Tab2 t{};
  for (int p = 0; p < 4; p++) {
    const double xi = QSX[p] * GAUSS_G, eta = QSY[p] * GAUSS_G;
    for (int n = 0; n < 4; n++) {
      t.G[p][0][n] = (QSX[n] * 0.25) * (1.0 + QSY[n] * eta);
      t.G[p][1][n] = (QSY[n] * 0.25) * (1.0 + QSX[n] * xi);
      t.N[p][n] = 0.25 * (1.0 + QSX[n] * xi) * (1.0 + QSY[n] * eta);
    }
  }
  return t;
}

__constant__ Tab3 c_tab3 = make_tab3();
__constant__ Tab2 c_tab2 = make_tab2();

// ---------------------------------------------------------------------------
struct KArgs {
  const double *coords;   // node layers [cks, cke)
  const double *gp;       // material AoS (may be null for uniform / Poisson)
  double *out;            // CSR values of the processed rows
  unsigned long long *bad;  // det(J) <= 0 counter
  int64_t base_off;       // row_start of the first processed row (subtracted)
  int64_t cke;            // one past the last coords layer
  int64_t s0, s1;         // processed node layers of the slowest direction
  int64_t x0, x1, y0, y1; // processed node box in the fast directions (y unused in 2-D)
  double E, nu;           // uniform material
  int32_t mat_layout, mat_stride;
  int32_t bc;             // BC_* kind
  int32_t iface;          // 1: interface pass (elastic BC rows become zero rows)
  int32_t zc;             // row-sweep kernels: node layers (rows in 2-D) per workgroup
  int32_t rect;           // row-sweep kernels: the mesh was verified rectilinear (k_check_rectilinear): geometry from 1-D lines
  int32_t ends;           // the interior kernel writes the Dirichlet rows of the x-end nodes of its lines; k_bc_rows skips them
  double *bout;           // k_assemble_e2_rect: also write b of its node rows (E2 one-pass form), or null
  const double *ecoef;    // k_assemble_s3_hex2: (c11, c12, c44, -) per element of layers [els, ele) (k_s3_elem_coefs)
  int32_t cond;           // general kernels: run only if bad[1] != 0, i.e. the fast kernel launched before on the same rows
                          // found its premise broken (femasm_assemble_interface: no host round trip)
};

// isotropic coefficient triple scaled by w*detJ, in the reference's operation order
template <int KIND>
__device__ inline void scaled_coefs(double E, double nu, double wJ, double &d11, double &d12, double &d44) {
  if constexpr (KIND == KIND_PLANE) {  // E2:297-305
    const double factor = E / ((1.0 + nu) * (1.0 - 2.0 * nu));
    d11 = factor * (1.0 - nu) * 1.0 * wJ;
    d12 = factor * nu * 1.0 * wJ;
    d44 = factor * (0.5 * (1.0 - 2.0 * nu)) * 1.0 * wJ;
  } else {                             // S3:487-500
    const double factor = E * (1 - nu) / ((1.0 + nu) * (1.0 - 2.0 * nu));
    d11 = factor * 1.0 * 1.0 * wJ;
    d12 = factor * (nu / (1 - nu)) * 1.0 * wJ;
    d44 = factor * (0.5 * (1 - 2 * nu) / (1 - nu)) * 1.0 * wJ;
  }
}

// 3x3 Jacobian, its inverse and determinant in the reference's order (A3, A4)
struct Jac3 { double iJ[3][3]; double det; };
// closed-form inverse (shared products t4 ... t14, t17 = 1/det: P3:86-107, S3:162-183) and the determinant by its
// own six-term formula (P3:143, S3:220) -- the ONE place these are written out; every kernel calls this
__device__ inline void inv3_ref(const double (&J)[3][3], double (&iJ)[3][3], double &det) {
  const double t4 = J[2][0] * J[0][1], t6 = J[2][0] * J[0][2], t8 = J[1][0] * J[0][1];
  const double t10 = J[1][0] * J[0][2], t12 = J[0][0] * J[1][1], t14 = J[0][0] * J[1][2];
  const double t17 = 1.0 / (t4 * J[1][2] - t6 * J[1][1] - t8 * J[2][2] + t10 * J[2][1] + t12 * J[2][2] - t14 * J[2][1]);
  iJ[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) * t17;
  iJ[0][1] = -(J[0][1] * J[2][2] - J[0][2] * J[2][1]) * t17;
  iJ[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * t17;
  iJ[1][0] = -(-J[2][0] * J[1][2] + J[1][0] * J[2][2]) * t17;
  iJ[1][1] = (-t6 + J[0][0] * J[2][2]) * t17;
  iJ[1][2] = -(-t10 + t14) * t17;
  iJ[2][0] = (-J[2][0] * J[1][1] + J[1][0] * J[2][1]) * t17;
  iJ[2][1] = -(-t4 + J[0][0] * J[2][1]) * t17;
  iJ[2][2] = (-t8 + t12) * t17;
  det = J[0][0] * J[1][1] * J[2][2] - J[0][0] * J[1][2] * J[2][1] - J[1][0] * J[0][1] * J[2][2] +
        J[1][0] * J[0][2] * J[2][1] + J[2][0] * J[0][1] * J[1][2] - J[2][0] * J[0][2] * J[1][1];
}
__device__ inline Jac3 jac3(const double (*G)[8], const double *xc /*[8][3]*/) {
  double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
  for (int n = 0; n < 8; n++)
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) J[r][c] = J[r][c] + G[r][n] * xc[3 * n + c];
  Jac3 o;
  inv3_ref(J, o.iJ, o.det);
  return o;
}
__device__ inline double det3_only(const double (*G)[8], const double *xc) {
  double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
  for (int n = 0; n < 8; n++)
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) J[r][c] = J[r][c] + G[r][n] * xc[3 * n + c];
  return J[0][0] * J[1][1] * J[2][2] - J[0][0] * J[1][2] * J[2][1] - J[1][0] * J[0][1] * J[2][2] +
         J[1][0] * J[0][2] * J[2][1] + J[2][0] * J[0][1] * J[1][2] - J[2][0] * J[0][2] * J[1][1];
}
struct Jac2 { double i00, i01, i10, i11, det; };
__device__ inline Jac2 jac2(const double (*G)[4], const double *xc /*[4][2]*/) {
  double J00 = 0, J01 = 0, J10 = 0, J11 = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    J00 = J00 + G[0][i] * xc[2 * i];
    J01 = J01 + G[0][i] * xc[2 * i + 1];
    J10 = J10 + G[1][i] * xc[2 * i];
    J11 = J11 + G[1][i] * xc[2 * i + 1];
  }
  Jac2 o;
  o.det = (J00 * J11) - (J01 * J10);
  o.i00 = J11 / o.det;
  o.i01 = -J01 / o.det;
  o.i10 = -J10 / o.det;
  o.i11 = J00 / o.det;
  return o;
}

// ---------------------------------------------------------------------------
// contraction of one (a,b) node pair at one Gauss point into a dof x dof block
// ---------------------------------------------------------------------------
template <int KIND, int D>
__device__ inline void add_block(double *acc, const double *l, const double *m, const double *q, const double *gb) {
  if constexpr (KIND == KIND_POISSON) {
    // l = w*detJ * grad N_a   (P3:279: Ke += g.g * w * J)
    double v = acc[0];
#pragma unroll
    for (int k = 0; k < 3; k++) v = fma(l[k], gb[k], v);
    acc[0] = v;
  } else {
    // B^T D B with isotropic D (E2:285-320, S3:470-515), zeros of B and D skipped:
    //   K[i][j] += D12 ga_i gb_j + D44 ga_j gb_i            (i != j)
    //   K[i][i] += D11 ga_i gb_i + D44 sum_{k!=i} ga_k gb_k
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
      for (int j = 0; j < D; j++) {
        double v = acc[i * D + j];
        if (i != j) {
          v = fma(l[i], gb[j], v);
          v = fma(m[j], gb[i], v);
        } else {
          v = fma(q[i], gb[i], v);
#pragma unroll
          for (int k = 0; k < D; k++)
            if (k != i) v = fma(m[k], gb[k], v);
        }
        acc[i * D + j] = v;
      }
  }
}

// ---------------------------------------------------------------------------
// small index helpers
// ---------------------------------------------------------------------------
constexpr int HDI[8] = {0, 0, 1, 1, 0, 0, 1, 1};
constexpr int HDJ[8] = {0, 1, 1, 0, 0, 1, 1, 0};
constexpr int HDK[8] = {0, 0, 0, 0, 1, 1, 1, 1};
constexpr int QDI[4] = {0, 0, 1, 1};
constexpr int QDJ[4] = {0, 1, 1, 0};

__device__ inline bool on_boundary(const Slab &s, int64_t i, int64_t j, int64_t k) {
  bool b = (i == 0) || (i == s.dim[0] - 1) || (j == 0) || (j == s.dim[1] - 1);
  if (s.nsd == 3) b = b || (k == 0) || (k == s.dim[2] - 1);
  return b;
}
// is (node, any dof) a Dirichlet row of this BC kind
__device__ inline bool is_bc_node(const Slab &s, int bc, int64_t i, int64_t j, int64_t k) {
  if (bc == BC_POISSON || bc == BC_S3) return on_boundary(s, i, j, k);
  if (bc == BC_E2) return (i == 0) || (i == s.dim[0] - 1);
  return false;
}

// ---------------------------------------------------------------------------
// K kernel
// ---------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256, 2) void k_assemble_rows(const Slab s, const KArgs a) {
  using T = Traits<KIND>;
  if (a.cond && a.bad[1] == 0) return;
  constexpr int NSD = T::NSD, DOF = T::DOF, NEN = T::NEN, NGP = T::NGP, NCOEF = T::NCOEF;
  constexpr int BX = T::BX, BY = T::BY, BZ = T::BZ;
  constexpr int EX = BX + 1, EY = BY + 1, EZ = (NSD == 3) ? BZ + 1 : 1;
  constexpr int CX = BX + 2, CY = BY + 2, CZ = (NSD == 3) ? BZ + 2 : 1;
  constexpr int NE = EX * EY * EZ, NC = CX * CY * CZ;
  constexpr int REC = NSD * NEN + NCOEF;
  constexpr int DD = DOF * DOF;
  constexpr int NBLK = (NSD == 3) ? 9 : 3;  // neighbour blocks per thread
  constexpr int NACC = NBLK * DD;
  constexpr int XSTR = NACC | 1;            // odd exchange stride: no LDS bank conflicts
  static_assert(BX * BY * BZ == 64, "a brick is one wave of nodes");
  static_assert(NE <= 256, "one geometry thread per element");
  static_assert(64 * XSTR <= 2 * NE * REC, "exchange area aliases the record buffers");

  __shared__ double sh_x[NC * NSD];
  __shared__ double sh_rec[2 * NE * REC];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int type = tid >> 6;            // wave-uniform: 0..3
  const int oes = (type < 2) ? -1 : 0;  // element layer relative to the node (slowest direction)
  const int obs = type & 1;             // which face of that element (slowest direction)
  const int as = -oes;                  // this node's position in that element (slowest direction)

  // ---- brick -> global ----------------------------------------------------
  const int64_t M = s.dim[0], N = s.dim[1], P = s.dim[2];
  int64_t g0x, g0y, g0z;  // global index of the brick's first node; grid = bricks per direction
  int lx, ly, lz;
  if constexpr (NSD == 3) {
    g0x = a.x0 + (int64_t)blockIdx.x * BX;
    g0y = a.y0 + (int64_t)blockIdx.y * BY;
    g0z = a.s0 + (int64_t)blockIdx.z * BZ;
    lx = lane & 3; ly = (lane >> 2) & 3; lz = lane >> 4;
  } else {
    g0x = a.x0 + (int64_t)blockIdx.x * BX;
    g0y = a.s0 + (int64_t)blockIdx.y * BY;
    g0z = 0;
    lx = lane & 15; ly = lane >> 4; lz = 0;
  }

  // ---- stage the brick's node coordinates in LDS (clamped reads: elements that would
  //      use a clamped node are masked out below) -------------------------------------
  for (int q = tid; q < NC; q += 256) {
    const int cx = q % CX, cy = (q / CX) % CY, cz = q / (CX * CY);
    int64_t gi = g0x - 1 + cx, gj = g0y - 1 + cy, gk = g0z - 1 + cz;
    gi = gi < 0 ? 0 : (gi > M - 1 ? M - 1 : gi);
    int64_t node;
    if constexpr (NSD == 3) {
      gj = gj < 0 ? 0 : (gj > N - 1 ? N - 1 : gj);
      gk = gk < s.cks ? s.cks : (gk > a.cke - 1 ? a.cke - 1 : gk);
      node = gi + M * (gj + N * (gk - s.cks));
    } else {
      gj = gj < s.cks ? s.cks : (gj > a.cke - 1 ? a.cke - 1 : gj);
      node = gi + M * (gj - s.cks);
    }
#pragma unroll
    for (int d = 0; d < NSD; d++) sh_x[q * NSD + d] = a.coords[node * NSD + d];
  }

  // ---- geometry-thread set-up: thread tid < NE owns element tid of the halo'd brick ----
  const bool gthread = tid < NE;
  const int ex = tid % EX, ey = (tid / EX) % EY, ez = tid / (EX * EY);
  bool eexists = false;
  int64_t emat = 0;
  if (gthread) {
    const int64_t gex = g0x - 1 + ex, gey = g0y - 1 + ey, gez = g0z - 1 + ez;
    if constexpr (NSD == 3) {
      eexists = gex >= 0 && gex < M - 1 && gey >= 0 && gey < N - 1 && gez >= s.els && gez < s.ele;
      emat = gex + (M - 1) * (gey + (N - 1) * (gez - s.mat_e_ks));
    } else {
      eexists = gex >= 0 && gex < M - 1 && gey >= s.els && gey < s.ele;
      emat = gex + (M - 1) * (gey - s.mat_e_ks);
    }
  }
  __syncthreads();

  // one geometry record: grad N_b (physical) for the NEN nodes, then the w*detJ-scaled coefficients
  auto geometry = [&](int p, double *recbuf) {
    if (!gthread) return;
    double *r = recbuf + tid * REC;
    if (!eexists) {
#pragma unroll
      for (int q = 0; q < REC; q++) r[q] = 0.0;
      return;
    }
    double xc[NEN * NSD];
    double wJ;
    if constexpr (NSD == 3) {
#pragma unroll
      for (int n = 0; n < 8; n++) {
        const int cn = (ex + HDI[n]) + CX * ((ey + HDJ[n]) + CY * (ez + HDK[n]));
#pragma unroll
        for (int d = 0; d < 3; d++) xc[3 * n + d] = sh_x[3 * cn + d];
      }
      const double(*G)[8] = c_tab3.G[p];
      const Jac3 jj = jac3(G, xc);
#pragma unroll
      for (int n = 0; n < 8; n++)
#pragma unroll
        for (int k = 0; k < 3; k++)
          r[3 * n + k] = G[0][n] * jj.iJ[k][0] + G[1][n] * jj.iJ[k][1] + G[2][n] * jj.iJ[k][2];
      wJ = 1.0 * jj.det;
    } else {
#pragma unroll
      for (int n = 0; n < 4; n++) {
        const int cn = (ex + QDI[n]) + CX * (ey + QDJ[n]);
        xc[2 * n] = sh_x[2 * cn];
        xc[2 * n + 1] = sh_x[2 * cn + 1];
      }
      const double(*G)[4] = c_tab2.G[p];
      const Jac2 jj = jac2(G, xc);
#pragma unroll
      for (int n = 0; n < 4; n++) {
        r[2 * n] = G[0][n] * jj.i00 + G[1][n] * jj.i01;
        r[2 * n + 1] = G[0][n] * jj.i10 + G[1][n] * jj.i11;
      }
      wJ = 1.0 * jj.det;
    }
    if (!(wJ > 0.0)) atomicAdd(a.bad, 1ULL);
    if constexpr (KIND == KIND_POISSON) {
      r[NSD * NEN] = wJ;
    } else {
      double E = a.E, nu = a.nu;
      if (a.mat_layout == 0 || a.mat_layout == 3) {
        // the element's record (AoS: E[p], nu[p] at offE, offNu) or its entry in the E plane of FEMASM_MAT_ELEMENT_SOA
        // (nu one plane distance further); one address computation for both, no extra live registers
        constexpr int offE = (KIND == KIND_PLANE) ? 8 : 96, offNu = (KIND == KIND_PLANE) ? 12 : 104;
        const bool rec = a.mat_layout == 0;
        const double *q = a.gp + emat * (rec ? a.mat_stride : 1);
        E = q[rec ? offE + p : 0];
        nu = q[rec ? offNu + p : a.mat_stride];
      }
      scaled_coefs<KIND>(E, nu, wJ, r[NSD * NEN], r[NSD * NEN + 1], r[NSD * NEN + 2]);
    }
  };

  double acc[NACC];
#pragma unroll
  for (int q = 0; q < NACC; q++) acc[q] = 0.0;

  // contraction of the records that touch this thread's node, for its third of the stencil
  auto contract = [&](const double *recbuf) {
    if constexpr (NSD == 3) {
      const int ezl = lz + 1 + oes;
#pragma unroll
      for (int oy = -1; oy <= 0; oy++)
#pragma unroll
        for (int ox = -1; ox <= 0; ox++) {
          const double *r = recbuf + ((lx + 1 + ox) + EX * ((ly + 1 + oy) + EY * ezl)) * REC;
          const double *ra = r + (4 * as + quad_idx(-ox, -oy)) * 3;
          double l[3], m[3], q[3];
          if constexpr (KIND == KIND_POISSON) {
            const double wJ = r[24];
#pragma unroll
            for (int k = 0; k < 3; k++) l[k] = ra[k] * wJ;
          } else {
            const double d11 = r[24], d12 = r[25], d44 = r[26];
#pragma unroll
            for (int k = 0; k < 3; k++) {
              l[k] = d12 * ra[k];
              m[k] = d44 * ra[k];
              q[k] = d11 * ra[k];
            }
          }
#pragma unroll
          for (int by = 0; by <= 1; by++)
#pragma unroll
            for (int bx = 0; bx <= 1; bx++) {
              const double *rb = r + (4 * obs + quad_idx(bx, by)) * 3;
              const double gb[3] = {rb[0], rb[1], rb[2]};
              add_block<KIND, DOF>(acc + ((oy + by + 1) * 3 + (ox + bx + 1)) * DD, l, m, q, gb);
            }
        }
    } else {
      const int eyl = ly + 1 + oes;
#pragma unroll
      for (int ox = -1; ox <= 0; ox++) {
        const double *r = recbuf + ((lx + 1 + ox) + EX * eyl) * REC;
        // local index of (bx, by): bx == 0 ? by : 3 - by
        const double *ra = r + ((ox == 0) ? as : 3 - as) * 2;
        const double d11 = r[8], d12 = r[9], d44 = r[10];
        double l[2], m[2], q[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
          l[k] = d12 * ra[k];
          m[k] = d44 * ra[k];
          q[k] = d11 * ra[k];
        }
#pragma unroll
        for (int bx = 0; bx <= 1; bx++) {
          const double *rb = r + ((bx == 0) ? obs : 3 - obs) * 2;
          const double gb[2] = {rb[0], rb[1]};
          add_block<KIND, DOF>(acc + (ox + bx + 1) * DD, l, m, q, gb);
        }
      }
    }
  };

  // ---- Gauss-point pipeline: records of point p+1 are built while point p is contracted ----
  geometry(0, sh_rec);
  __syncthreads();
  for (int p = 0; p < NGP; p++) {
#if FEMASM_EXP != 3
    if (p + 1 < NGP) geometry(p + 1, sh_rec + ((p + 1) & 1) * (NE * REC));
#endif
#if FEMASM_EXP != 2
    contract(sh_rec + (p & 1) * (NE * REC));
#endif
    __syncthreads();
  }

  // ---- middle third: lower-layer half (type 1) + upper-layer half (type 2), fixed order ----
  double *exch = sh_rec;
  if (type == 1) {
#pragma unroll
    for (int q = 0; q < NACC; q++) exch[lane * XSTR + q] = acc[q];
  }
  __syncthreads();
  if (type == 1) return;
  if (type == 2) {
#pragma unroll
    for (int q = 0; q < NACC; q++) acc[q] = exch[lane * XSTR + q] + acc[q];
  }

  // ---- write this thread's third of its node's rows: each CSR value exactly once ----
  const int64_t i = g0x + lx, j = g0y + ly, k = g0z + lz;
  const int ds = oes + obs;  // -1, 0, +1 : neighbour layer in the slowest direction
  bool valid;
  int64_t sl, SL;            // index / extent of the slowest direction
  if constexpr (NSD == 3) { valid = i < a.x1 && j < a.y1 && k < a.s1; sl = k; SL = P; }
  else                    { valid = i < a.x1 && j < a.s1;            sl = j; SL = N; }
  if (!valid || sl + ds < 0 || sl + ds > SL - 1) return;

  const bool isbc = is_bc_node(s, a.bc, i, j, k);
  double bcdiag = a.iface ? 0.0 : 1.0;
  if (KIND == KIND_POISSON && isbc) {
    // _ZERO_ROW_i then ADD: the diagonal becomes the number of (assembled) elements sharing the node
    const int vx = (i > 0) + (i < M - 1), vy = (j > 0) + (j < N - 1);
    const int vz = (k - 1 >= s.els && k - 1 < s.ele) + (k >= s.els && k < s.ele);
    bcdiag = (double)(vx * vy * vz);
  }
  const int cntx = span1(i, M), xi0 = (i == 0);
  const int si = ds + 1 - (sl == 0);
  const int64_t rs0 = row_start(s, i, j, k, 0) - a.base_off;
  if constexpr (NSD == 3) {
    const int cnty = span1(j, N), yi0 = (j == 0);
    const int cnt = cntx * cnty * span1(k, P);
#pragma unroll
    for (int c = 0; c < DOF; c++) {
      double *row = a.out + rs0 + (int64_t)c * DOF * cnt;
#pragma unroll
      for (int dy = -1; dy <= 1; dy++) {
        if (j + dy < 0 || j + dy > N - 1) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
          if (i + dx < 0 || i + dx > M - 1) continue;
          const int idx = (si * cnty + (dy + 1 - yi0)) * cntx + (dx + 1 - xi0);
#pragma unroll
          for (int cc = 0; cc < DOF; cc++) {
            double v = acc[((dy + 1) * 3 + (dx + 1)) * DD + c * DOF + cc];
            if (isbc) v = (ds == 0 && dy == 0 && dx == 0 && cc == c) ? bcdiag : 0.0;
#if FEMASM_EXP == 1
            asm volatile("" :: "v"(v));
#else
            row[DOF * idx + cc] = v;
#endif
          }
        }
      }
    }
  } else {
    const int cnt = cntx * span1(j, N);
#pragma unroll
    for (int c = 0; c < DOF; c++) {
      double *row = a.out + rs0 + (int64_t)c * DOF * cnt;
#pragma unroll
      for (int dx = -1; dx <= 1; dx++) {
        if (i + dx < 0 || i + dx > M - 1) continue;
        const int idx = si * cntx + (dx + 1 - xi0);
#pragma unroll
        for (int cc = 0; cc < DOF; cc++) {
          double v = acc[(dx + 1) * DD + c * DOF + cc];
          if (isbc) v = (ds == 0 && dx == 0 && cc == c) ? bcdiag : 0.0;
          row[DOF * idx + cc] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// RHS, pass 1: one thread per assembled element -> Fe (and Ge) in the scratch array
// ---------------------------------------------------------------------------
// sin(pi x) for the Poisson load on general hexahedra (P3:556 evaluates sin(PI * x) at every Gauss point: 24 sines per
// element).  r = x - rint(x), |r| <= 1/2, sin(pi x) = (-1)^rint(x) sin(pi r) with ONE odd Taylor polynomial of degree 21 in
// r (truncation 1e-18 at |r| = 1/2, coefficients rounded from 60 digits; relative error <= 4e-16 over the interval,
// scripts/ check in DESIGN 4.5): 17 instructions where the library's sinpi (a sine and a cosine polynomial and a select,
// any argument class) takes 40.  Inf/NaN arguments give NaN, like the library's.
__device__ inline double sinpi_poly(double x) {
  const double n = rint(x), r = x - n, t = r * r;
  double p = 5.392664662608129e-10;
  p = fma(p, t, -2.2948428997269873e-08);
  p = fma(p, t, 7.952054001475513e-07);
  p = fma(p, t, -2.1915353447830217e-05);
  p = fma(p, t, 0.00046630280576761255);
  p = fma(p, t, -0.0073704309457143504);
  p = fma(p, t, 0.08214588661112823);
  p = fma(p, t, -0.5992645293207921);
  p = fma(p, t, 2.5501640398773455);
  p = fma(p, t, -5.16771278004997);
  p = fma(p, t, 3.141592653589793);
  // parity of n: n is an integer-valued double; n * 0.5 has a fractional part iff n is odd (exact for |n| < 2^52; larger
  // arguments are even integers times ulp: r = 0 and the sign does not matter)
  const double h = n * 0.5;
  const double sgn = (h != rint(h)) ? -r : r;
  return p * sgn;
}

#define PH_BUF a.bad
#if FEMASM_EXP == 3  // phase clocks (profiling build only): PH_ACC(p, id) adds the time since the last mark to phase p
#define PH_DECL unsigned long long ph_[4] = {0, 0, 0, 0}, ph_n = 0, ph_t = __builtin_readcyclecounter()
#define PH_T(id) ph_t = __builtin_readcyclecounter(), ph_n++
#define PH_ACC(p, id)                                            \
  do {                                                           \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    ph_[p] += now_ - ph_t;                                       \
    ph_t = now_;                                                 \
  } while (0)
#define PH_PIN(x) asm volatile("" ::"v"(x))
#define PH_FLUSH(base)                                                    \
  if (lane == 0) {                                                        \
    atomicAdd(PH_BUF + (base), ph_n);                                      \
    for (int q_ = 0; q_ < 4; q_++) atomicAdd(PH_BUF + (base) + 1 + q_, ph_[q_]); \
  }
#else
#define PH_DECL
#define PH_T(id)
#define PH_ACC(p, id)
#define PH_PIN(x)
#define PH_FLUSH(base)
#endif

// Element-vector array (pass 1 -> pass 2 of the RHS and of the operator apply): blocks of 16 consecutive
// elements, NOUT = nen*dof lines of 16 doubles per block, so a block is NOUT*128 contiguous bytes.  (Component-major
// rows over the whole mesh made every wave write NOUT single lines megabytes apart: 0.26 ms of 1.13 ms at 128^3.)
template <int NOUT>
__device__ inline int64_t ef_idx(int64_t e, int t) { return (e >> 4) * (NOUT * 16) + t * 16 + (e & 15); }

struct RArgs {
  const double *coords;
  const double *gp;
  double *elemF;          // element vectors, blocks of 16 elements x (nen*dof) lines (ef_idx)
  double *out;            // b of the processed rows
  const double *recv;     // interface_apply
  int64_t cke;
  int64_t s0, s1;
  int64_t nel;            // assembled elements
  int32_t mat_layout, mat_stride;
  int32_t bc, iface;
  unsigned long long *bad;  // plan flags / profiling counters
  const double *x;          // N4 operator apply: input vector on the node layers of coords (else null)
  int32_t rect;             // the rectilinearity pre-pass ran in this call: bad[3] == 0 <=> every element is a box
  int32_t runs;             // elemF holds the records of k_rhs_elem_elastic_runs, not element vectors
  double E, nu;             // FEMASM_MAT_MANUFACTURED: the drivers' -iso_E / -iso_nu
  int32_t cond;             // k_rhs_e2_fused: run only if bad[1] != 0 (the one-pass kernel's premise failed: b again)
  int32_t simple;           // option rhs_simple: the Poisson load on general hexahedra as the reference states it
};

template <int KIND>
__device__ inline void rhs_elem_one(const Slab &s, const RArgs &a, int64_t e) {
  using T = Traits<KIND>;
  constexpr int NFE = T::NFE;
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t mx = M - 1, my = N - 1;
  double F[NFE];
#pragma unroll
  for (int q = 0; q < NFE; q++) F[q] = 0.0;

  if constexpr (KIND == KIND_PLANE) {
    const int64_t ex = e % mx, ey = s.els + e / mx;
    double xc[8];
#pragma unroll
    for (int n = 0; n < 4; n++) {
      const int64_t node = (ex + QDI[n]) + M * (ey + QDJ[n] - s.cks);
      xc[2 * n] = a.coords[2 * node];
      xc[2 * n + 1] = a.coords[2 * node + 1];
    }
    if (a.mat_layout == 0) {
      const double *q = a.gp + (ex + mx * (ey - s.mat_e_ks)) * a.mat_stride;
      for (int p = 0; p < 4; p++) {  // E2:339-349
        const Jac2 jj = jac2(c_tab2.G[p], xc);
        const double fac = 1.0 * jj.det;
        const double fx = q[16 + p], fy = q[20 + p];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          F[2 * i] += fac * c_tab2.N[p][i] * fx;
          F[2 * i + 1] += fac * c_tab2.N[p][i] * fy;
        }
      }
    }
  } else {
    const int64_t ex = e % mx, ey = (e / mx) % my, ez = s.els + e / (mx * my);
    double xc[24];
#pragma unroll
    for (int n = 0; n < 8; n++) {
      const int64_t node = (ex + HDI[n]) + M * ((ey + HDJ[n]) + N * (ez + HDK[n] - s.cks));
#pragma unroll
      for (int d = 0; d < 3; d++) xc[3 * n + d] = a.coords[3 * node + d];
    }
    if constexpr (KIND == KIND_POISSON) {
      constexpr double PI = 3.14159265358979323846264338327950288;
      // axis-aligned box (every element of a DMDASetUniformCoordinates mesh): the Gauss points are a tensor grid,
      // so 6 sines instead of 24 and one det(J); tested exactly per element, other elements take the general loop
      bool box = true;
#pragma unroll
      for (int n = 0; n < 8; n++) {
        box = box && xc[3 * n] == xc[3 * (HDI[n] ? 3 : 0)] && xc[3 * n + 1] == xc[3 * (HDJ[n] ? 1 : 0) + 1] &&
              xc[3 * n + 2] == xc[3 * (HDK[n] ? 4 : 0) + 2];
      }
      if (box) {
        const double h[3] = {xc[9] - xc[0], xc[4] - xc[1], xc[14] - xc[2]};
        const double fac = 1.0 * ((0.5 * h[0]) * (0.5 * h[1]) * (0.5 * h[2]));
        double sn[3][2];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          const double lo = xc[d], hi = xc[(d == 0 ? 9 : (d == 1 ? 3 : 12)) + d];
          const double a0 = 0.5 * (1.0 + GAUSS_G), a1 = 0.5 * (1.0 - GAUSS_G);
          sn[d][0] = sinpi(a0 * lo + a1 * hi);  // Gauss abscissa -g
          sn[d][1] = sinpi(a1 * lo + a0 * hi);  // Gauss abscissa +g
        }
        for (int p = 0; p < 8; p++) {
          const double fv = 3.0 * PI * PI * sn[0][HDI[p]] * sn[1][HDJ[p]] * sn[2][HDK[p]];
#pragma unroll
          for (int i = 0; i < 8; i++) F[i] += fac * fv * c_tab3.N[p][i];
        }
      } else if (a.simple) {
        for (int p = 0; p < 8; p++) {  // P3:549-566, as the reference states it (option rhs_simple: for A/B and tests)
          const double fac = 1.0 * det3_only(c_tab3.G[p], xc);
          double pos[3] = {0.0, 0.0, 0.0};
#pragma unroll
          for (int n = 0; n < 8; n++)
#pragma unroll
            for (int d = 0; d < 3; d++) pos[d] += c_tab3.N[p][n] * xc[3 * n + d];
          // sinpi: the reference's sin(pi*x) without the large-argument reduction (differs by ~1 ulp of the argument)
          const double fv = 3.0 * PI * PI * sinpi(pos[0]) * sinpi(pos[1]) * sinpi(pos[2]);
#pragma unroll
          for (int i = 0; i < 8; i++) F[i] += fac * fv * c_tab3.N[p][i];
        }
      } else {
        // P3:549-566 on a general hexahedron in the tensor-product form of the trilinear map (round 3; as the geometry of
        // k_assemble_p3_hex): x(xi) = sum c_ijk xi^i eta^j zeta^k / 8 with c the Hadamard transform of the node
        // coordinates; every row of 8 J takes four values over the Gauss points; det J = r0 . (r1 x r2) / 512; the Gauss
        // points' positions by three butterfly stages.  810 instead of 2 200 instructions per element (24 of the
        // library's sinpi were 960 of them).  Same sums in a different order: rounding only.
        double c[2][2][2][3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          double v[2][2][2];
#pragma unroll
          for (int n = 0; n < 8; n++) v[HDI[n]][HDJ[n]][HDK[n]] = xc[3 * n + d];
#pragma unroll
          for (int q = 0; q < 2; q++)
#pragma unroll
            for (int r = 0; r < 2; r++) {
              const double a0 = v[0][q][r], a1 = v[1][q][r];
              v[0][q][r] = a1 + a0;
              v[1][q][r] = a1 - a0;
            }
#pragma unroll
          for (int q = 0; q < 2; q++)
#pragma unroll
            for (int r = 0; r < 2; r++) {
              const double a0 = v[q][0][r], a1 = v[q][1][r];
              v[q][0][r] = a1 + a0;
              v[q][1][r] = a1 - a0;
            }
#pragma unroll
          for (int q = 0; q < 2; q++)
#pragma unroll
            for (int r = 0; r < 2; r++) {
              const double a0 = v[q][r][0], a1 = v[q][r][1];
              v[q][r][0] = a1 + a0;
              v[q][r][1] = a1 - a0;
            }
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
              for (int k = 0; k < 2; k++) c[i][j][k][d] = v[i][j][k];
        }
        double R0[2][2][3], R1[2][2][3], R2[2][2][3], X[2][2][2][3];
#pragma unroll
        for (int d = 0; d < 3; d++)
#pragma unroll
          for (int q = 0; q < 2; q++) {
            const double gq = q ? GAUSS_G : -GAUSS_G;
            const double u0 = fma(gq, c[1][1][0][d], c[1][0][0][d]), v0 = fma(gq, c[1][1][1][d], c[1][0][1][d]);  // eta = gq
            const double u1 = fma(gq, c[1][1][0][d], c[0][1][0][d]), v1 = fma(gq, c[1][1][1][d], c[0][1][1][d]);  // xi = gq
            const double u2 = fma(gq, c[1][0][1][d], c[0][0][1][d]);                                              // xi = gq
            // position: e[j][k] = c0jk + xi c1jk at xi = gq
            const double e00 = fma(gq, c[1][0][0][d], c[0][0][0][d]), e10 = u1, e01 = u2, e11 = v1;
#pragma unroll
            for (int r = 0; r < 2; r++) {
              const double gr = r ? GAUSS_G : -GAUSS_G;
              R0[q][r][d] = fma(gr, v0, u0);  // dx/dxi   at (eta_q, zeta_r)
              R1[q][r][d] = fma(gr, v1, u1);  // dx/deta  at (xi_q, zeta_r)
              R2[q][r][d] = fma(gr, v1, u2);  // dx/dzeta at (xi_q, eta_r)
              const double f0 = fma(gr, e10, e00), f1 = fma(gr, e11, e01);  // eta = gr: f[k] = e[0][k] + eta e[1][k]
              X[q][r][0][d] = 0.125 * fma(-GAUSS_G, f1, f0);
              X[q][r][1][d] = 0.125 * fma(GAUSS_G, f1, f0);
            }
          }
#pragma unroll
        for (int ia = 0; ia < 2; ia++)
#pragma unroll
          for (int ib = 0; ib < 2; ib++)
#pragma unroll
            for (int ic = 0; ic < 2; ic++) {
              const double *r0 = R0[ib][ic], *r1 = R1[ia][ic], *r2 = R2[ia][ib];
              const double a0 = fma(r1[1], r2[2], -(r1[2] * r2[1])), a1 = fma(r1[2], r2[0], -(r1[0] * r2[2])),
                           a2 = fma(r1[0], r2[1], -(r1[1] * r2[0]));
              const double fac = fma(r0[2], a2, fma(r0[1], a1, r0[0] * a0)) * (1.0 / 512.0);  // 1.0 * det J
              const double *x = X[ia][ib][ic];
              const double fv = 3.0 * PI * PI * sinpi_poly(x[0]) * sinpi_poly(x[1]) * sinpi_poly(x[2]);
              const int p = hex_idx(ia, ib, ic);
#pragma unroll
              for (int i = 0; i < 8; i++) F[i] += fac * fv * c_tab3.N[p][i];
            }
      }
    } else if (a.mat_layout == 0) {
      const double *q = a.gp + (ex + mx * (ey + my * (ez - s.mat_e_ks))) * a.mat_stride;
      for (int p = 0; p < 8; p++) {  // S3:534-545
        const double fac = 1.0 * det3_only(c_tab3.G[p], xc);
        const double fx = q[160 + p], fy = q[168 + p], fz = q[176 + p];
#pragma unroll
        for (int i = 0; i < 8; i++) {
          F[3 * i] += fac * c_tab3.N[p][i] * fx;
          F[3 * i + 1] += fac * c_tab3.N[p][i] * fy;
          F[3 * i + 2] += fac * c_tab3.N[p][i] * fz;
        }
      }
      // natural-BC face integral over all six faces of every element, S3:548-656
      const double dx = fabs(xc[0] - xc[9]), dy = fabs(xc[1] - xc[4]), dz = fabs(xc[2] - xc[14]);
      const double *gx = q + 184, *gy = q + 256, *gz = q + 328, *EB = q + 112, *nuB = q + 136;
      for (int p = 0; p < 24; p++) {
        const double pe = EB[p], pn = nuB[p];
        const double factor = pe * (1 - pn) / ((1.0 + pn) * (1.0 - 2.0 * pn));
        const double d11 = factor * 1.0, d12 = factor * (pn / (1 - pn)), d44 = factor * (0.5 * (1 - 2 * pn) / (1 - pn));
        const double e0 = gx[3 * p], e1 = gy[3 * p + 1], e2 = gz[3 * p + 2];
        const double e3 = gx[3 * p + 1] + gy[3 * p], e4 = gy[3 * p + 2] + gz[3 * p + 1], e5 = gx[3 * p + 2] + gz[3 * p];
        const double z0 = d11 * e0 + d12 * e1 + d12 * e2;
        const double z1 = d12 * e0 + d11 * e1 + d12 * e2;
        const double z2 = d12 * e0 + d12 * e1 + d11 * e2;
        const double z3 = d44 * e3, z4 = d44 * e4, z5 = d44 * e5;
        const int axis = p / 8;
        const double side = ((p / 4) % 2 == 0) ? 1.0 : -1.0;
        double sn[3], Jp;
        if (axis == 0)      { sn[0] = z0 * side; sn[1] = z3 * side; sn[2] = z5 * side; Jp = (0.5 * dy) * (0.5 * dz); }
        else if (axis == 1) { sn[0] = z3 * side; sn[1] = z1 * side; sn[2] = z4 * side; Jp = (0.5 * dx) * (0.5 * dz); }
        else                { sn[0] = z5 * side; sn[1] = z4 * side; sn[2] = z2 * side; Jp = (0.5 * dx) * (0.5 * dy); }
        const double fac = 1.0 * Jp;
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const double w = fac * c_tab3.NF[p][i];
          F[24 + 3 * i] += w * sn[0];
          F[24 + 3 * i + 1] += w * sn[1];
          F[24 + 3 * i + 2] += w * sn[2];
        }
      }
    }
  }
  // component-major: consecutive lanes (elements) write consecutive doubles, whole 128-byte lines per instruction
  constexpr int NOUT = T::NEN * T::DOF;
#pragma unroll
  for (int q = 0; q < NOUT; q++) {
    if constexpr (KIND == KIND_ELASTIC) a.elemF[ef_idx<NOUT>(e, q)] = F[q] + F[NOUT + q];
    else a.elemF[ef_idx<NOUT>(e, q)] = F[q];
  }
}
// ---------------------------------------------------------------------------
// FEMASM_MAT_MANUFACTURED (S3/R3): the loads of the drivers' manufactured solution evaluated HERE instead of being
// streamed from the 3 200-byte records the drivers fill with them (S3:1243-1363 fill -> S3:720-803 use): per element
// the positions of the 8 + 24 Gauss points (S3:1271-1303), body force and displacement gradient of
// u = (0, 0, sin pi x sin pi y sin pi z) (evaluate_Elastic, S3:958-985) with the drivers' constant E, nu, then Fe + Ge
// exactly as from a record (S3:520-656).  An axis-aligned box (tested exactly per element) needs the sines and cosines
// of only 4 abscissae per axis (2 Gauss, 2 ends): 12 sincospi instead of 96.  sincospi(x) for sin(pi x): rounding only.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_rhs_elem_s3_manufactured(const Slab s, const RArgs a) {
  constexpr double PI = 3.14159265358979323846264338327950288;
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t mx = M - 1, my = N - 1;
  const double E = a.E, nu = a.nu;
  const double den = 2 * nu * nu + nu - 1;
  const double factor = E * (1 - nu) / ((1.0 + nu) * (1.0 - 2.0 * nu));
  const double d11 = factor * 1.0, d12 = factor * (nu / (1 - nu)), d44 = factor * (0.5 * (1 - 2 * nu) / (1 - nu));
  for (int64_t e = (int64_t)blockIdx.x * 128 + threadIdx.x; e < a.nel; e += (int64_t)gridDim.x * 128) {
    const int64_t ex = e % mx, ey = (e / mx) % my, ez = s.els + e / (mx * my);
    double xc[24];
#pragma unroll
    for (int n = 0; n < 8; n++) {
      const int64_t node = (ex + HDI[n]) + M * ((ey + HDJ[n]) + N * (ez + HDK[n] - s.cks));
#pragma unroll
      for (int d = 0; d < 3; d++) xc[3 * n + d] = a.coords[3 * node + d];
    }
    bool box = true;
#pragma unroll
    for (int n = 0; n < 8; n++)
      box = box && xc[3 * n] == xc[3 * (HDI[n] ? 3 : 0)] && xc[3 * n + 1] == xc[3 * (HDJ[n] ? 1 : 0) + 1] &&
            xc[3 * n + 2] == xc[3 * (HDK[n] ? 4 : 0) + 2];
    double F[24];
#pragma unroll
    for (int q = 0; q < 24; q++) F[q] = 0.0;
    const double dx = fabs(xc[0] - xc[9]), dy = fabs(xc[1] - xc[4]), dz = fabs(xc[2] - xc[14]);  // S3:560-562
    // sin / cos of pi * coordinate at the abscissae of one axis: index 0: -g, 1: +g, 2: lower end, 3: upper end
    double sn[3][4], cs[3][4];
    if (box) {
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const double lo = xc[d], hi = xc[(d == 0 ? 9 : (d == 1 ? 3 : 12)) + d];
        const double a0 = 0.5 * (1.0 + GAUSS_G), a1 = 0.5 * (1.0 - GAUSS_G);
        sincospi(a0 * lo + a1 * hi, &sn[d][0], &cs[d][0]);
        sincospi(a1 * lo + a0 * hi, &sn[d][1], &cs[d][1]);
        sincospi(lo, &sn[d][2], &cs[d][2]);
        sincospi(hi, &sn[d][3], &cs[d][3]);
      }
    }
    // body force at the volume Gauss points (S3:1304-1330 fill, S3:534-545 use)
    for (int p = 0; p < 8; p++) {
      double sx, cx, sy, cy, sz, cz, fac;
      if (box) {
        sx = sn[0][HDI[p]]; cx = cs[0][HDI[p]]; sy = sn[1][HDJ[p]]; cy = cs[1][HDJ[p]]; sz = sn[2][HDK[p]]; cz = cs[2][HDK[p]];
        fac = 1.0 * ((0.5 * (xc[9] - xc[0])) * (0.5 * (xc[4] - xc[1])) * (0.5 * (xc[14] - xc[2])));
      } else {
        double pos[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int n = 0; n < 8; n++)
#pragma unroll
          for (int d = 0; d < 3; d++) pos[d] = pos[d] + c_tab3.N[p][n] * xc[3 * n + d];
        sincospi(pos[0], &sx, &cx); sincospi(pos[1], &sy, &cy); sincospi(pos[2], &sz, &cz);
        fac = 1.0 * det3_only(c_tab3.G[p], xc);
      }
      const double fx = (E * PI * PI * cx * cz * sy) / (2 * den);            // S3:972-977
      const double fy = (E * PI * PI * cy * sx * cz) / (2 * den);
      const double fz = (E * PI * PI * sx * sy * sz * (3 * nu - 2)) / den;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        F[3 * i] += fac * c_tab3.N[p][i] * fx;
        F[3 * i + 1] += fac * c_tab3.N[p][i] * fy;
        F[3 * i + 2] += fac * c_tab3.N[p][i] * fz;
      }
    }
    // natural-BC term: grad u_z at the 24 face Gauss points (S3:1331-1363 fill, S3:568-648 use); u_x = u_y = 0
    for (int p = 0; p < 24; p++) {
      const int axis = p / 8, q = p & 3;
      const double side = ((p / 4) % 2 == 0) ? 1.0 : -1.0;
      double sx, cx, sy, cy, sz, cz;
      if (box) {
        // face point: the face's axis sits at an end (+1 face: upper), the two free axes at -g/+g in the order of S3:256-284
        const int a0 = (axis == 0) ? 1 : 0, a1 = (axis == 2) ? 1 : 2;
        int idx[3];
        idx[axis] = (side > 0.0) ? 3 : 2;
        idx[a0] = q & 1;
        idx[a1] = q >> 1;
        sx = sn[0][idx[0]]; cx = cs[0][idx[0]]; sy = sn[1][idx[1]]; cy = cs[1][idx[1]]; sz = sn[2][idx[2]]; cz = cs[2][idx[2]];
      } else {
        double pos[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int n = 0; n < 8; n++)
#pragma unroll
          for (int d = 0; d < 3; d++) pos[d] = pos[d] + c_tab3.NF[p][n] * xc[3 * n + d];
        sincospi(pos[0], &sx, &cx); sincospi(pos[1], &sy, &cy); sincospi(pos[2], &sz, &cz);
      }
      const double g0 = PI * cx * sy * sz, g1 = PI * sx * cy * sz, g2 = PI * sx * sy * cz;  // grad u_z, S3:978-983
      const double z0 = d12 * g2, z1 = d12 * g2, z2 = d11 * g2;   // strains: e2 = g2, e4 = g1, e5 = g0, the others 0
      const double z4 = d44 * g1, z5 = d44 * g0;
      double t[3], Jp;
      if (axis == 0)      { t[0] = z0 * side; t[1] = 0.0 * side; t[2] = z5 * side; Jp = (0.5 * dy) * (0.5 * dz); }
      else if (axis == 1) { t[0] = 0.0 * side; t[1] = z1 * side; t[2] = z4 * side; Jp = (0.5 * dx) * (0.5 * dz); }
      else                { t[0] = z5 * side; t[1] = z4 * side; t[2] = z2 * side; Jp = (0.5 * dx) * (0.5 * dy); }
      const double fac = 1.0 * Jp;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const double w = fac * c_tab3.NF[p][i];
        F[3 * i] += w * t[0];
        F[3 * i + 1] += w * t[1];
        F[3 * i + 2] += w * t[2];
      }
    }
#pragma unroll
    for (int q = 0; q < 24; q++) a.elemF[ef_idx<24>(e, q)] = F[q];
  }
}

// grid-stride over the elements (a capped grid: the kernel is also launched when it only has to find out, on the
// device, that the rectilinear P3 path has already written b)
template <int KIND>
__global__ __launch_bounds__(256) void k_rhs_elem(const Slab s, const RArgs a) {
  if constexpr (KIND == KIND_POISSON) {
    if (a.rect != 0 && a.bad[3] == 0ULL) return;  // rectilinear mesh: k_poisson_rhs_rect has written b
  }
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < a.nel; e += (int64_t)gridDim.x * 256)
    rhs_elem_one<KIND>(s, a, e);
}

// ---------------------------------------------------------------------------
// RHS, pass 1 for S3/R3, "runs" form.  One wave = one workgroup, autonomous.  Element pass, 8 lanes per element,
// 8 elements per step: the 8 x 2.3 KB of AoS inputs (doubles [112,400) = E_B[24] nu_B[24] fx[8] fy[8] fz[8] gx[72]
// gy[72] gz[72]) are read as 18 contiguous 16-byte-per-lane non-temporal runs, one step ahead through registers
// (indices clamped, never masked: a load under an exec mask or a uniform branch makes the compiler merge old and
// new register contents through copies behind s_waitcnt vmcnt(0), which serialises the prefetch), and laid out per
// element in LDS (element stride 296 doubles: conflict-free for the strides 1 and 3 of the per-Gauss-point reads).
// Lane j of an element owns the face Gauss points j, j+8, j+16 -- one per face pair, so the three evaluations are
// straight-line code with the axis a compile-time constant and no lane idles -- and the volume Gauss point j; then
// lane j = node j sums the 12 face points of its three faces and the 8 volume points (Fe + Ge).
// Then the part of the node sum that is local to a wave.  Writes that share the memory system with a read stream are expensive on this part (the
// 0.4 GB of element vectors cost 0.25 ms of the 1.1 ms element pass, whatever their layout or cache policy;
// scripts/ubench/read_pattern.hip), so a wave keeps a run of 16 x-consecutive elements and walks it through
// RUNS_ZC element layers: it adds the two elements either side of each of the run's 17 node columns and the two
// element layers either side of each node layer, and writes, per node layer, 17 x 2 (y side) x 3 partial sums as
// one 1 KB record -- a third of the bytes.  k_rhs_gather<ELASTIC, true> adds the <= 8 records' entries that meet
// in a node (2 y sides x run boundary x chunk boundary), in a fixed order.
// Record of (task, layer l): 128 doubles, entry (c*2 + dy)*20 + X; task = (chunk*my + ey)*nr + run.
// ---------------------------------------------------------------------------
constexpr int RUNS_ZC = 32;
template <bool RECT>
__global__ __launch_bounds__(64) void k_rhs_elem_elastic_runs(const Slab s, const RArgs a) {
  if ((a.rect != 0 && a.bad[3] == 0ULL) != RECT) return;
  constexpr int S = 296, W01S = 40, W2S = 40, ZC = RUNS_ZC;
  __shared__ __attribute__((aligned(16))) double sh_in[8 * S];  // inputs; reused for the tractions once consumed
  __shared__ __attribute__((aligned(16))) double sh_xc[8][RECT ? 8 : 24];
  __shared__ __attribute__((aligned(16))) double sh_o[16 * 8 * 3];  // element vectors of the run; then the record
  __shared__ double sh_G[RECT ? 1 : 8][3][8];
  const int lane = threadIdx.x, x = lane >> 3, j = lane & 7;
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t mx = M - 1, my = N - 1;
  const int nlay = (int)(s.ele - s.els);
  const uint32_t nr = (uint32_t)((mx + 15) >> 4), nzc = (uint32_t)((nlay + ZC - 1) / ZC);
  const int64_t ntasks = (int64_t)nzc * my * nr;
  if ((int64_t)blockIdx.x >= ntasks) return;
  if constexpr (!RECT) {
    for (int t = lane; t < 192; t += 64) (&sh_G[0][0][0])[t] = (&c_tab3.G[0][0][0])[t];
  }
  const int hd[3] = {HDI[j], HDJ[j], HDK[j]};
  double cff[3][4], cfv[8];
  int pb[3];
#pragma unroll
  for (int ax = 0; ax < 3; ax++) {
    pb[ax] = 8 * ax + (hd[ax] ? 0 : 4);
#pragma unroll
    for (int m = 0; m < 4; m++) cff[ax][m] = c_tab3.NF[pb[ax] + m][j];
  }
#pragma unroll
  for (int m = 0; m < 8; m++) cfv[m] = c_tab3.N[m][j];
  int64_t cofs;
  if constexpr (RECT) {
    const int64_t nofs = (j == 1) ? 1 : (j == 3) ? M : (j == 5) ? M * N : 0;
    cofs = 3 * nofs + ((j < 6) ? (j >> 1) : 0);
  } else {
    cofs = 3 * (HDI[j] + M * (HDJ[j] + N * (int64_t)HDK[j]));
  }
  const int64_t eoff = mx * my * (s.els - s.mat_e_ks);
  const int st2 = a.mat_stride >> 1;
  const double side = (j & 4) ? -1.0 : 1.0;
  auto wave_sync = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  // node-sum roles: lane = (column X = lane>>2, y side dy, z side dz); lanes 0..3 also serve column 16
  const int X = lane >> 2, dy = (lane >> 1) & 1, dz = lane & 1;
  const int nL = hex_idx(1, dy, dz), nR = hex_idx(0, dy, dz);  // the node on the column: right nodes of X-1, left of X

  double2 pin[18];
  double pxc[RECT ? 1 : 3];
#pragma unroll
  for (int k = 0; k < 18; k++) pin[k] = make_double2(0.0, 0.0);
#pragma unroll
  for (int d = 0; d < (RECT ? 1 : 3); d++) pxc[d] = 0.0;
  // step (task t, layer l of its chunk, half h): elements 8h .. 8h+7 of the run -> registers (clamped, never masked).
  // COMMIT: each register is first written to LDS (the step that was requested before) and re-requested at once, so
  // the wave keeps its 19 loads in flight instead of draining them all, committing, and only then asking again
  // (measured: the input stream alone 0.82 -> see DESIGN 4.5).
  auto request = [&](auto commit, int64_t t, int l, int h, double2 *pin, double *pxc) {
    constexpr bool COMMIT = decltype(commit)::value;
    double2 *dstq = reinterpret_cast<double2 *>(sh_in);
    if (t >= ntasks) { t = ntasks - 1; l = 0; h = 0; }
    const uint32_t t32 = (uint32_t)t, tt = t32 / nr, r = t32 - tt * nr, ch = tt / (uint32_t)my, ey = tt - ch * (uint32_t)my;
    const int64_t ex0 = 16 * (int64_t)r, ezr = (int64_t)ch * ZC + l;  // element layer relative to els
    const int nx = (int)((mx - ex0 < 16) ? mx - ex0 : 16);
    const double2 *blk =
        reinterpret_cast<const double2 *>(a.gp + (ex0 + mx * (ey + my * ezr) + eoff) * a.mat_stride + 112);
#pragma unroll
    for (int k = 0; k < 18; k++) {
      const int c = 64 * k + lane, xe = c / 144, rr = c - 144 * xe;
      const int xo = (8 * h + xe < nx) ? 8 * h + xe : nx - 1;
      if constexpr (COMMIT) dstq[xe * (S / 2) + rr] = pin[k];
      // streamed once: non-temporal (6.2 -> 7.0 TB/s on the bare read pattern, scripts/ubench/read_pattern.hip)
      typedef double d2v __attribute__((ext_vector_type(2)));
      const d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(blk) + (xo * st2 + rr));
      pin[k] = make_double2(v.x, v.y);
      // keep the issue order identical at every call site: the waits are counted in issue order, and a different
      // order in the prologue makes the compiler drain all loads before the first commit
      asm volatile("" ::: "memory");
    }
    const int xo = (8 * h + x < nx) ? 8 * h + x : nx - 1;
    const double *cn = a.coords + 3 * (ex0 + xo + M * (ey + N * (s.els + ezr - s.cks))) + cofs;
    if constexpr (COMMIT) {
      if constexpr (RECT) {
        sh_xc[x][j] = pxc[0];
      } else {
#pragma unroll
        for (int d = 0; d < 3; d++) sh_xc[x][3 * j + d] = pxc[d];
      }
    }
    pxc[0] = cn[0];
    if constexpr (!RECT) {
      pxc[1] = cn[1];
      pxc[2] = cn[2];
    }
  };

  int64_t t = blockIdx.x;
  request(std::false_type{}, t, 0, 0, pin, pxc);
  for (; t < ntasks; t += gridDim.x) {
    const uint32_t t32 = (uint32_t)t, tt = t32 / nr, r = t32 - tt * nr, ch = tt / (uint32_t)my;
    const int nx = (int)((mx - 16 * (int64_t)r < 16) ? mx - 16 * (int64_t)r : 16);
    const int nlz = (nlay - (int)ch * ZC < ZC) ? nlay - (int)ch * ZC : ZC;
    double *rec = a.elemF + t * (int64_t)((ZC + 1) * 128);
    double carry[3] = {0.0, 0.0, 0.0}, carry16[3] = {0.0, 0.0, 0.0};
    auto emit = [&](const double *o, const double *o16, int l) {  // record of node layer l of the chunk
      if (dz == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) sh_o[(c * 2 + dy) * 20 + X] = o[c];
        if (lane < 4)
#pragma unroll
          for (int c = 0; c < 3; c++) sh_o[(c * 2 + dy) * 20 + 16] = o16[c];
      }
      wave_sync();
#if FEMASM_EXP == 7 || FEMASM_EXP == 8
      if (sh_o[0] == 123.456)
#endif
      {
        typedef double d2v __attribute__((ext_vector_type(2)));
        const double2 v = *reinterpret_cast<const double2 *>(&sh_o[2 * lane]);
        d2v vv; vv.x = v.x; vv.y = v.y;
        __builtin_nontemporal_store(vv, reinterpret_cast<d2v *>(rec + l * 128 + 2 * lane));
      }
      wave_sync();
    };
#pragma unroll 1
    for (int l = 0; l < nlz; l++) {
#pragma unroll 1
      for (int h = 0; h < 2; h++) {
        {  // commit this step's inputs while asking for those of the step after it
          int64_t tn = t;
          int ln = l, hn = h + 1;
          if (hn == 2) { hn = 0; ln = l + 1; }
          if (ln == nlz) { ln = 0; tn = t + gridDim.x; }
          request(std::true_type{}, tn, ln, hn, pin, pxc);
        }
        wave_sync();
        const double *in = sh_in + x * S;
#if FEMASM_EXP == 8  // skeleton: input stream + commit only
        {
          const double vv = in[j] + sh_xc[x][j & 3];
          wave_sync();
#pragma unroll
          for (int c = 0; c < 3; c++) sh_o[((8 * h + x) * 8 + j) * 3 + c] = vv;
          wave_sync();
          continue;
        }
#endif
        double dx, dyy, dzz;
        if constexpr (RECT) {
          const double2 cx = *reinterpret_cast<const double2 *>(&sh_xc[x][0]);
          const double2 cy = *reinterpret_cast<const double2 *>(&sh_xc[x][2]);
          const double2 cz = *reinterpret_cast<const double2 *>(&sh_xc[x][4]);
          dx = fabs(cx.x - cx.y); dyy = fabs(cy.x - cy.y); dzz = fabs(cz.x - cz.y);
        } else {
          const double *xc = sh_xc[x];
          dx = fabs(xc[0] - xc[9]); dyy = fabs(xc[1] - xc[4]); dzz = fabs(xc[2] - xc[14]);  // S3:560-562
        }
        double sn[3][3], u[3];
#pragma unroll
        for (int ax = 0; ax < 3; ax++) {  // face Gauss point p = j + 8 ax: traction sigma.n * face Jacobian, S3:568-648
          const int p = j + 8 * ax;
          const double pe = in[p], pn = in[24 + p];
          const double tq = pe / ((1.0 + pn) * (1.0 - 2.0 * pn));  // S3:585-593 with one division
          const double d11 = tq * (1 - pn), d12 = tq * pn, d44 = tq * (0.5 * (1 - 2 * pn));
          const double *gx = in + 72 + 3 * p, *gy = in + 144 + 3 * p, *gz = in + 216 + 3 * p;
          const double e0 = gx[0], e1 = gy[1], e2 = gz[2];
          const double e3 = gx[1] + gy[0], e4 = gy[2] + gz[1], e5 = gx[2] + gz[0];
          const double z0 = d11 * e0 + d12 * e1 + d12 * e2;
          const double z1 = d12 * e0 + d11 * e1 + d12 * e2;
          const double z2 = d12 * e0 + d12 * e1 + d11 * e2;
          const double z3 = d44 * e3, z4 = d44 * e4, z5 = d44 * e5;
          double Jp;
          if (ax == 0)      { sn[ax][0] = z0 * side; sn[ax][1] = z3 * side; sn[ax][2] = z5 * side; Jp = (0.5 * dyy) * (0.5 * dzz); }
          else if (ax == 1) { sn[ax][0] = z3 * side; sn[ax][1] = z1 * side; sn[ax][2] = z4 * side; Jp = (0.5 * dx) * (0.5 * dzz); }
          else              { sn[ax][0] = z5 * side; sn[ax][1] = z4 * side; sn[ax][2] = z2 * side; Jp = (0.5 * dx) * (0.5 * dyy); }
          const double fac = 1.0 * Jp;
#pragma unroll
          for (int c = 0; c < 3; c++) sn[ax][c] *= fac;
        }
        {  // volume Gauss point j: w detJ f, S3:534-545
          double fac;
          if constexpr (RECT) fac = 1.0 * ((0.5 * dx) * (0.5 * dyy) * (0.5 * dzz));
          else fac = 1.0 * det3_only(sh_G[j], sh_xc[x]);
          u[0] = fac * in[48 + j];
          u[1] = fac * in[56 + j];
          u[2] = fac * in[64 + j];
        }
        wave_sync();  // every lane has consumed its inputs: the region is reused for the tractions
        double2 *w01 = reinterpret_cast<double2 *>(sh_in) + x * W01S;
        double *w2 = sh_in + 2 * 8 * W01S + x * W2S;
#pragma unroll
        for (int ax = 0; ax < 3; ax++) {
          w01[j + 8 * ax] = make_double2(sn[ax][0], sn[ax][1]);
          w2[j + 8 * ax] = sn[ax][2];
        }
        w01[24 + j] = make_double2(u[0], u[1]);
        w2[24 + j] = u[2];
        wave_sync();
        double o[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int ax = 0; ax < 3; ax++)
#pragma unroll
          for (int m = 0; m < 4; m++) {
            const double2 v = w01[pb[ax] + m];
            const double v2 = w2[pb[ax] + m];
            o[0] = fma(cff[ax][m], v.x, o[0]);
            o[1] = fma(cff[ax][m], v.y, o[1]);
            o[2] = fma(cff[ax][m], v2, o[2]);
          }
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const double2 v = w01[24 + m];
          const double v2 = w2[24 + m];
          o[0] = fma(cfv[m], v.x, o[0]);
          o[1] = fma(cfv[m], v.y, o[1]);
          o[2] = fma(cfv[m], v2, o[2]);
        }
#pragma unroll
        for (int c = 0; c < 3; c++) sh_o[((8 * h + x) * 8 + j) * 3 + c] = o[c];  // node j of element 8h + x of the run
        wave_sync();
      }
      // node columns of the run on this element layer: right nodes of element X-1 + left nodes of element X
      double v[3], v16[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double left = (X > 0 && X - 1 < nx) ? sh_o[((X - 1) * 8 + nL) * 3 + c] : 0.0;
        const double right = (X < nx) ? sh_o[(X * 8 + nR) * 3 + c] : 0.0;
        v[c] = left + right;
        v16[c] = (lane < 4 && nx == 16) ? sh_o[(15 * 8 + nL) * 3 + c] : 0.0;
      }
      wave_sync();  // sh_o is reused for the record
      // node layer l of the chunk: upper nodes of element layer l-1 (carried) + lower nodes of layer l
      double o[3], o16[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        o[c] = carry[c] + v[c];
        o16[c] = carry16[c] + v16[c];
      }
      emit(o, o16, l);
#pragma unroll
      for (int c = 0; c < 3; c++) {  // the z-lower lanes receive their partner's upper-node sums
        carry[c] = __shfl_xor(v[c], 1);
        carry16[c] = __shfl_xor(v16[c], 1);
      }
    }
    emit(carry, carry16, nlz);  // the node layer above the chunk's last element layer
  }
}

// ---------------------------------------------------------------------------
// N4: matrix-free operator apply, pass 1: one thread per element computes f_e = Ke x_e without forming Ke
// (per Gauss point: displacement gradient H = sum_b x_b grad N_b^T, stress from the w*detJ-scaled D11/D12/D44,
// f_a = sigma grad N_a; Poisson: f_a = w detJ grad N_a . sum_b x_b grad N_b) into the element-vector array;
// pass 2 is k_rhs_gather (same order of summation as the assembled rows' element contributions).
// ---------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void k_op_elem(const Slab s, const RArgs a, double Eu, double nuu) {
  using T = Traits<KIND>;
  constexpr int NSD = T::NSD, DOF = T::DOF, NEN = T::NEN, NGP = T::NGP;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= a.nel) return;
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t mx = M - 1, my = N - 1;
  double xc[NEN * NSD], xe[NEN * DOF], F[NEN * DOF];
  int64_t emat;
  if constexpr (NSD == 3) {
    const int64_t ex = e % mx, ey = (e / mx) % my, ez = s.els + e / (mx * my);
    emat = ex + mx * (ey + my * (ez - s.mat_e_ks));
#pragma unroll
    for (int n = 0; n < 8; n++) {
      const int64_t node = (ex + HDI[n]) + M * ((ey + HDJ[n]) + N * (ez + HDK[n] - s.cks));
#pragma unroll
      for (int d = 0; d < 3; d++) xc[3 * n + d] = a.coords[3 * node + d];
#pragma unroll
      for (int c = 0; c < DOF; c++) xe[n * DOF + c] = a.x[node * DOF + c];
    }
  } else {
    const int64_t ex = e % mx, ey = s.els + e / mx;
    emat = ex + mx * (ey - s.mat_e_ks);
#pragma unroll
    for (int n = 0; n < 4; n++) {
      const int64_t node = (ex + QDI[n]) + M * (ey + QDJ[n] - s.cks);
#pragma unroll
      for (int d = 0; d < 2; d++) xc[2 * n + d] = a.coords[2 * node + d];
#pragma unroll
      for (int c = 0; c < DOF; c++) xe[n * DOF + c] = a.x[node * DOF + c];
    }
  }
#pragma unroll
  for (int q = 0; q < NEN * DOF; q++) F[q] = 0.0;
  for (int p = 0; p < NGP; p++) {
    double g[NEN][NSD], wJ;
    if constexpr (NSD == 3) {
      const double(*G)[8] = c_tab3.G[p];
      const Jac3 jj = jac3(G, xc);
#pragma unroll
      for (int n = 0; n < 8; n++)
#pragma unroll
        for (int k = 0; k < 3; k++) g[n][k] = G[0][n] * jj.iJ[k][0] + G[1][n] * jj.iJ[k][1] + G[2][n] * jj.iJ[k][2];
      wJ = 1.0 * jj.det;
    } else {
      const double(*G)[4] = c_tab2.G[p];
      const Jac2 jj = jac2(G, xc);
#pragma unroll
      for (int n = 0; n < 4; n++) {
        g[n][0] = G[0][n] * jj.i00 + G[1][n] * jj.i01;
        g[n][1] = G[0][n] * jj.i10 + G[1][n] * jj.i11;
      }
      wJ = 1.0 * jj.det;
    }
    if constexpr (KIND == KIND_POISSON) {
      double gu[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int n = 0; n < 8; n++)
#pragma unroll
        for (int k = 0; k < 3; k++) gu[k] += g[n][k] * xe[n];
#pragma unroll
      for (int n = 0; n < 8; n++) F[n] += wJ * (g[n][0] * gu[0] + g[n][1] * gu[1] + g[n][2] * gu[2]);
    } else {
      double E = Eu, nu = nuu;
      if (a.mat_layout == 0 || a.mat_layout == 3) {  // AoS record, or the planes of FEMASM_MAT_ELEMENT_SOA
        constexpr int offE = (KIND == KIND_PLANE) ? 8 : 96, offNu = (KIND == KIND_PLANE) ? 12 : 104;
        const bool rec = a.mat_layout == 0;
        const double *q = a.gp + emat * (rec ? a.mat_stride : 1);
        E = q[rec ? offE + p : 0];
        nu = q[rec ? offNu + p : a.mat_stride];
      }
      double d11, d12, d44;
      scaled_coefs<KIND>(E, nu, wJ, d11, d12, d44);
      double H[NSD][NSD];  // H[i][j] = d u_i / d x_j
#pragma unroll
      for (int i = 0; i < NSD; i++)
#pragma unroll
        for (int j = 0; j < NSD; j++) {
          double v = 0.0;
#pragma unroll
          for (int n = 0; n < NEN; n++) v += xe[n * DOF + i] * g[n][j];
          H[i][j] = v;
        }
      double sg[NSD][NSD];
      double tr = 0.0;
#pragma unroll
      for (int i = 0; i < NSD; i++) tr += H[i][i];
#pragma unroll
      for (int i = 0; i < NSD; i++)
#pragma unroll
        for (int j = 0; j < NSD; j++)
          sg[i][j] = (i == j) ? d11 * H[i][i] + d12 * (tr - H[i][i]) : d44 * (H[i][j] + H[j][i]);
#pragma unroll
      for (int n = 0; n < NEN; n++)
#pragma unroll
        for (int i = 0; i < NSD; i++) {
          double v = 0.0;
#pragma unroll
          for (int j = 0; j < NSD; j++) v += sg[i][j] * g[n][j];
          F[n * DOF + i] += v;
        }
    }
  }
#pragma unroll
  for (int q = 0; q < NEN * DOF; q++) a.elemF[ef_idx<NEN * DOF>(e, q)] = F[q];
}

// Dirichlet rows of b (and of the operator apply): the value the drivers set after the element loop
template <int KIND>
__device__ inline void rhs_dirichlet(const Slab &s, const RArgs &a, int64_t i, int64_t j, int64_t k, double *b) {
  using T = Traits<KIND>;
  constexpr int NSD = T::NSD, DOF = T::DOF;
  const int64_t M = s.dim[0], N = s.dim[1];
  if (is_bc_node(s, a.bc, i, j, k)) {
    if (a.iface) {
      // partial sums owed to the rank below: Dirichlet rows are ignored by the receiver
#pragma unroll
      for (int c = 0; c < DOF; c++) b[c] = 0.0;
    } else if (a.x) {
      // operator apply: a Dirichlet row is diag * e_row (diag 1; Poisson: valence of the assembled elements)
      double diag = 1.0;
      if (a.bc == BC_POISSON) {
        const int vx = (i > 0) + (i < M - 1), vy = (j > 0) + (j < N - 1);
        const int vz = (k - 1 >= s.els && k - 1 < s.ele) + (k >= s.els && k < s.ele);
        diag = (double)(vx * vy * vz);
      }
      const int64_t node = (NSD == 3) ? i + M * (j + N * (k - s.cks)) : i + M * (j - s.cks);
#pragma unroll
      for (int c = 0; c < DOF; c++) b[c] = diag * a.x[node * DOF + c];
    } else if (a.bc == BC_POISSON) {
      b[0] = 0.0;                                   // P3:624-630
    } else if (a.bc == BC_E2) {
      b[0] = (i == M - 1) ? -1.0 : 1.0;             // E2:882-885
      if constexpr (DOF > 1) b[1] = 0.0;
    } else if (a.bc == BC_S3) {
      constexpr double PI = 3.14159265358979323846264338327950288;
      const int64_t node = i + M * (j + N * (k - s.cks));
      const double x = a.coords[3 * node], y = a.coords[3 * node + 1], z = a.coords[3 * node + 2];
      b[0] = 0.0;                                   // S3:964-968 via S3:1501-1505
      if constexpr (DOF > 2) { b[1] = 0.0; b[2] = sin(PI * x) * sin(PI * y) * sin(PI * z); }
    }
  }
}

// ---------------------------------------------------------------------------
// RHS, pass 2: one thread per node; contributions added in the reference's element
// order (k -> j -> i) (S3: Fe + Ge of an element enter as one term); then the Dirichlet value.
// ---------------------------------------------------------------------------
template <int KIND, bool RUNS>
__device__ inline void rhs_gather_node(const Slab &s, const RArgs &a, int64_t i, int64_t by, int64_t bz) {
  using T = Traits<KIND>;
  constexpr int NSD = T::NSD, DOF = T::DOF;
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t mx = M - 1, my = N - 1;
  int64_t j, k, q;
  if constexpr (NSD == 3) { j = by; k = a.s0 + bz; q = i + M * (j + N * bz); }
  else                    { j = a.s0 + by; k = 0; q = i + M * by; }

  double b[DOF];
#pragma unroll
  for (int c = 0; c < DOF; c++) b[c] = 0.0;

  if constexpr (RUNS) {
    // records of k_rhs_elem_elastic_runs: element layer k-1 before k, row j-1 before j, run r-1 before r
    constexpr int ZC = RUNS_ZC;
    const int64_t nlay = s.ele - s.els, nr = (mx + 15) >> 4, nzc = (nlay + ZC - 1) / ZC;
    const int64_t qz = k - s.els, c0 = qz / ZC, l0 = qz - c0 * ZC;
    const int64_t r0 = i >> 4, X0 = i & 15;
#pragma unroll
    for (int pz = 0; pz < 2; pz++) {
      const int64_t ch = pz == 0 ? c0 - 1 : c0, l = pz == 0 ? ZC : l0;
      if (pz == 0 ? !(l0 == 0 && c0 > 0) : !(c0 < nzc)) continue;
#pragma unroll
      for (int py = 0; py < 2; py++) {
        const int64_t ey = j - 1 + py;  // py = 0: the row below the node (its dy = 1 side)
        if (ey < 0 || ey >= my) continue;
#pragma unroll
        for (int px = 0; px < 2; px++) {
          const int64_t r = px == 0 ? r0 - 1 : r0, Xv = px == 0 ? 16 : X0;
          if (px == 0 ? !(X0 == 0 && r0 > 0) : !(r0 < nr)) continue;
          const double *R = a.elemF + (((ch * my + ey) * nr + r) * (ZC + 1) + l) * 128 + (1 - py) * 20 + Xv;
#pragma unroll
          for (int c = 0; c < DOF; c++) b[c] = b[c] + R[c * 40];
        }
      }
    }
  } else if constexpr (NSD == 3) {
    const int64_t e111 = i + mx * (j + my * (k - s.els));  // element (i,j,k); the others by constant offsets
#pragma unroll
    for (int oz = -1; oz <= 0; oz++)
#pragma unroll
      for (int oy = -1; oy <= 0; oy++)
#pragma unroll
        for (int ox = -1; ox <= 0; ox++) {
          const int64_t ex = i + ox, ey = j + oy, ez = k + oz;
          if (ex < 0 || ex >= mx || ey < 0 || ey >= my || ez < s.els || ez >= s.ele) continue;
          const double *F = a.elemF + ef_idx<T::NEN * DOF>(e111 + ox + mx * (oy + my * oz), 0);
          const int al = hex_idx(-ox, -oy, -oz);
#pragma unroll
          for (int c = 0; c < DOF; c++) b[c] = b[c] + F[(al * DOF + c) * 16];
        }
  } else {
    for (int oy = -1; oy <= 0; oy++)
      for (int ox = -1; ox <= 0; ox++) {
        const int64_t ex = i + ox, ey = j + oy;
        if (ex < 0 || ex >= mx || ey < s.els || ey >= s.ele) continue;
        const double *F = a.elemF + ef_idx<T::NEN * DOF>(ex + mx * (ey - s.els), 0);
        const int al = quad_idx(-ox, -oy);
#pragma unroll
        for (int c = 0; c < DOF; c++) b[c] = b[c] + F[(al * DOF + c) * 16];
      }
  }

  rhs_dirichlet<KIND>(s, a, i, j, k, b);
#pragma unroll
  for (int c = 0; c < DOF; c++) a.out[q * DOF + c] = b[c];
}
// one thread per node of a node layer taken as a flat (i, j) plane (every lane busy whatever M is); grid (x, 1, z)
// walked with strides (the launch caps it)
template <int KIND, bool RUNS = false>
__global__ __launch_bounds__(256) void k_rhs_gather(const Slab s, const RArgs a) {
  if constexpr (KIND == KIND_POISSON) {
    if (a.rect != 0 && a.bad[3] == 0ULL) return;  // rectilinear mesh: k_poisson_rhs_rect has written b
  }
  const int64_t M = s.dim[0];
  const int64_t ny = (Traits<KIND>::NSD == 3) ? s.dim[1] : a.s1 - a.s0, nz = (Traits<KIND>::NSD == 3) ? a.s1 - a.s0 : 1;
  const int64_t plane = M * ny;
  for (int64_t bz = blockIdx.z; bz < nz; bz += gridDim.z)
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < plane; n += (int64_t)gridDim.x * 256) {
      const int64_t by = n / M;
      rhs_gather_node<KIND, RUNS>(s, a, n - by * M, by, bz);
    }
}

// ---------------------------------------------------------------------------
// E2 right-hand side in one pass (no element-vector array).  A wave owns 64 x-consecutive elements and the 63 nodes
// between them (lane l: element column ex = n0 - 1 + l and, for l >= 1, node column i = ex) and walks E2R_YC node
// rows upwards: per element row every lane integrates its element's Fe (E2:325-350: 4 corner nodes, fx/fy at the 4
// Gauss points = 64 of the record's 192 bytes, read one row ahead), a node's four terms are added in the
// reference's element order (j-1 before j, i-1 before i) -- the two of the left-hand element come from lane l-1, the
// two of the row below from the previous step.  Every input is read once (+1/63 in x, +1/E2R_YC in y), b written once.
// 2048^2: 0.20 ms (two passes: 0.25 ms); chunk heights 8/16/32 and a second row of prefetch (FEMASM_E2R_PF2) land
// within 5 % of each other: the 64-of-192-byte reads, not latency, set the time.
// ---------------------------------------------------------------------------
#ifndef FEMASM_E2R_YC
#define FEMASM_E2R_YC 8
#endif
constexpr int E2R_YC = FEMASM_E2R_YC;
struct E2RIn { double xc[8]; double f[8]; };
template <bool ALIGNED>
__device__ inline void e2r_load(const Slab &s, const RArgs &a, int64_t exc, int64_t ey, E2RIn &o) {
  const int64_t M = s.dim[0], mx = M - 1;
  // element row clamped to the assembled ones (the caller discards the result of a clamped row)
  const int64_t eyc = ey < s.els ? s.els : (ey >= s.ele ? s.ele - 1 : ey);
  const double *c0 = a.coords + 2 * (exc + M * (eyc - s.cks)), *c1 = c0 + 2 * M;
  const double *q = a.gp + (exc + mx * (eyc - s.mat_e_ks)) * (int64_t)a.mat_stride + 16;
  if constexpr (ALIGNED) {
    const double2 n0 = *(const double2 *)c0, n3 = *(const double2 *)(c0 + 2);
    const double2 n1 = *(const double2 *)c1, n2 = *(const double2 *)(c1 + 2);
    o.xc[0] = n0.x; o.xc[1] = n0.y; o.xc[2] = n1.x; o.xc[3] = n1.y;
    o.xc[4] = n2.x; o.xc[5] = n2.y; o.xc[6] = n3.x; o.xc[7] = n3.y;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const double2 v = *(const double2 *)(q + 2 * t);
      o.f[2 * t] = v.x; o.f[2 * t + 1] = v.y;
    }
  } else {
    o.xc[0] = c0[0]; o.xc[1] = c0[1]; o.xc[2] = c1[0]; o.xc[3] = c1[1];
    o.xc[4] = c1[2]; o.xc[5] = c1[3]; o.xc[6] = c0[2]; o.xc[7] = c0[3];
#pragma unroll
    for (int t = 0; t < 8; t++) o.f[t] = q[t];
  }
}

template <bool ALIGNED, bool LIVE /* AoS material and at least one element row: else every Fe is zero */>
__global__ __launch_bounds__(256) void k_rhs_e2_fused(const Slab s, const RArgs a) {
  if (a.cond && a.bad[1] == 0) return;
  const int64_t M = s.dim[0], mx = M - 1;
  const int lane = threadIdx.x & 63;
  const int64_t n0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 63;
  if (n0 >= M) return;
  const int64_t ex = n0 - 1 + lane;
  const bool ex_ok = ex >= 0 && ex < mx;
  const int64_t exc = ex < 0 ? 0 : (ex >= mx ? mx - 1 : ex);
  const bool node_ok = lane >= 1 && ex < M;
  const int64_t ny = a.s1 - a.s0;
  const int64_t by0 = (int64_t)blockIdx.y * E2R_YC, by1 = by0 + E2R_YC < ny ? by0 + E2R_YC : ny;
  constexpr bool live = LIVE;
  E2RIn nxt;
#pragma unroll
  for (int t = 0; t < 8; t++) { nxt.xc[t] = 0.0; nxt.f[t] = 0.0; }
  if (live) e2r_load<ALIGNED>(s, a, exc, a.s0 + by0 - 1, nxt);
#ifdef FEMASM_E2R_PF2
  E2RIn nx2 = nxt;
  if (live) e2r_load<ALIGNED>(s, a, exc, a.s0 + by0, nx2);
#endif
  double lo1[2] = {0.0, 0.0}, lo2[2] = {0.0, 0.0};  // Fe of the element row below: local nodes 1 (0,1) and 2 (1,1)
  for (int64_t by = by0 - 1; by < by1; by++) {
    const int64_t ey = a.s0 + by;  // element row; also the node row written in this step (if by >= by0)
    const E2RIn cur = nxt;
#ifdef FEMASM_E2R_PF2
    nxt = nx2;
    if (live) e2r_load<ALIGNED>(s, a, exc, ey + 2, nx2);
#else
    if (live) e2r_load<ALIGNED>(s, a, exc, ey + 1, nxt);  // unconditional (clamped): stays in flight over the step
#endif
    double F[8];
#pragma unroll
    for (int t = 0; t < 8; t++) F[t] = 0.0;
    if (live) {
      for (int p = 0; p < 4; p++) {  // E2:339-349
        const Jac2 jj = jac2(c_tab2.G[p], cur.xc);
        const double fac = 1.0 * jj.det;
        const double fx = cur.f[p], fy = cur.f[4 + p];
#pragma unroll
        for (int n = 0; n < 4; n++) {
          F[2 * n] += fac * c_tab2.N[p][n] * fx;
          F[2 * n + 1] += fac * c_tab2.N[p][n] * fy;
        }
      }
    }
    const bool el_ok = ex_ok && ey >= s.els && ey < s.ele;
#pragma unroll
    for (int t = 0; t < 8; t++) F[t] = el_ok ? F[t] : 0.0;
    if (by >= by0) {
      double b[2];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const double l2 = __shfl_up(lo2[c], 1), l3 = __shfl_up(F[6 + c], 1);
        b[c] = 0.0 + l2;          // element (i-1, j-1), its node (1,1)
        b[c] = b[c] + lo1[c];     // element (i,   j-1), its node (0,1)
        b[c] = b[c] + l3;         // element (i-1, j  ), its node (1,0)
        b[c] = b[c] + F[c];       // element (i,   j  ), its node (0,0)
      }
      if (node_ok) {
        rhs_dirichlet<KIND_PLANE>(s, a, ex, ey, 0, b);
        double *o = a.out + 2 * (ex + M * by);
        o[0] = b[0];
        o[1] = b[1];
      }
    }
#pragma unroll
    for (int c = 0; c < 2; c++) { lo1[c] = F[2 + c]; lo2[c] = F[4 + c]; }
  }
}

// ---------------------------------------------------------------------------
// P3 right-hand side on a rectilinear mesh (verified by k_check_rectilinear earlier on this stream, verdict in
// bad[3]; otherwise both kernels return at once and k_rhs_elem + k_rhs_gather do the work).  Every element is a box,
// so det J = (hx/2)(hy/2)(hz/2), a Gauss point's position and the trilinear N_a factor per axis, and so does
// f = 3 pi^2 sin(pi x) sin(pi y) sin(pi z) (P3:352-359): the assembled entry of node (i,j,k) is
// 3 pi^2 Sx(i) Sy(j) Sz(k) with S(n) = sum over the two elements either side of n and their two Gauss abscissae
// of (h/2) N_side(xi) sin(pi x(xi)) -- the same terms as P3:534-567 + the node sum, in another order (rounding only).
// Pass 1: the three lines of S (z restricted to the element layers this rank assembles); pass 2: the outer product,
// Dirichlet nodes 0 (P3:624-630).  b is written once, no element vectors: 0.34 + 0.40 ms -> 0.05 ms at 256^3.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_poisson_rhs_lines(const Slab s, const RArgs a, double *lines) {
  if (!(a.rect != 0 && a.bad[3] == 0ULL)) return;
  const int64_t M = s.dim[0], N = s.dim[1], L = a.cke - s.cks;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= M + N + L) return;
  int64_t n, cnt, stride, e0, e1;  // node index on its line, nodes on the line, coordinate stride, element range
  int comp;
  if (t < M)          { n = t;         cnt = M; stride = 3;         comp = 0; e0 = 0; e1 = M - 1; }
  else if (t < M + N) { n = t - M;     cnt = N; stride = 3 * M;     comp = 1; e0 = 0; e1 = N - 1; }
  else                { n = t - M - N; cnt = L; stride = 3 * M * N; comp = 2; e0 = s.els - s.cks; e1 = s.ele - s.cks; }
  (void)cnt;
  double S = 0.0;
#pragma unroll
  for (int side = 1; side >= 0; side--) {  // element n-1 (the node is its upper end: side 1), then element n
    const int64_t e = n - side;
    if (e < e0 || e >= e1) continue;
    const double lo = a.coords[stride * e + comp], hi = a.coords[stride * (e + 1) + comp];
    const double hh = 0.5 * (hi - lo);
#pragma unroll
    for (int p = 0; p < 2; p++) {
      const double xi = (p == 0) ? -GAUSS_G : GAUSS_G;
      const double Nlo = 0.5 * (1.0 - xi), Nhi = 0.5 * (1.0 + xi);
      const double pos = Nlo * lo + Nhi * hi;
      S += hh * (side ? Nhi : Nlo) * sinpi(pos);
    }
  }
  lines[t] = S;
}
__global__ __launch_bounds__(256) void k_poisson_rhs_rect(const Slab s, const RArgs a, const double *lines) {
  if (!(a.rect != 0 && a.bad[3] == 0ULL)) return;
  constexpr double PI = 3.14159265358979323846264338327950288;
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t pq = (int64_t)blockIdx.x * 256 + threadIdx.x;  // node of the (i,j) plane: whole waves, coalesced
  if (pq >= M * N) return;
  const uint32_t j32 = (uint32_t)pq / (uint32_t)M;
  const int64_t j = j32, i = pq - j * M;
  for (int64_t bz = blockIdx.y; bz < a.s1 - a.s0; bz += gridDim.y) {
    const int64_t k = a.s0 + bz;
    double v = 0.0;
    if (!is_bc_node(s, a.bc, i, j, k)) v = 3.0 * PI * PI * lines[i] * lines[M + j] * lines[M + N + (k - s.cks)];
    a.out[pq + M * N * bz] = v;
  }
}

// ---------------------------------------------------------------------------
// Slab interface.  The rows of node layer L = ks-1 that a rank owes to the rank below come from ONE element layer
// (L), so only the columns in node layers L and L+1 can be non-zero: 18 dof of an interior row's 27 dof.  The wire
// buffer holds those two thirds only ("packed": per node, dof rows of 2 * cnt_rest * dof values; SURVEY section 5:
// 21.6 MB instead of 32 MB at 128^3 S3), then the b slice.
//   k_iface_pack   full CSR rows of the layer (what the stiffness kernels write) -> packed
//   k_iface_apply  own + received on the top owned node layer (the same layer seen from below), Dirichlet nodes skipped
// One wave per node; a node's rows are contiguous in both forms.
// ---------------------------------------------------------------------------
__device__ inline int64_t iface_inplane_pairs(const Slab &s, int64_t i, int64_t j) {
  return (s.nsd == 3) ? pre1(j, s.dim[1]) * tot1(s.dim[0]) + (int64_t)span1(j, s.dim[1]) * pre1(i, s.dim[0])
                      : pre1(i, s.dim[0]);
}
__global__ __launch_bounds__(256) void k_iface_pack(const Slab s, int64_t layer, const double *full, double *packed) {
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t nnode = (s.nsd == 3) ? M * N : M;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (q >= nnode) return;
  int64_t i = q % M, j, k;
  if (s.nsd == 3) { j = q / M; k = layer; } else { j = layer; k = 0; }
  const int crest = (s.nsd == 3) ? span1(i, M) * span1(j, N) : span1(i, M);
  const int cs = span1(layer, (s.nsd == 3) ? s.dim[2] : N);
  const int rl = cs * crest * s.dof, prl = 2 * crest * s.dof, skip = (layer > 0) ? crest * s.dof : 0;
  const int64_t i0 = 0, j0 = (s.nsd == 3) ? 0 : layer, k0 = (s.nsd == 3) ? layer : 0;
  const double *src = full + (row_start(s, i, j, k, 0) - row_start(s, i0, j0, k0, 0));
  double *dst = packed + (int64_t)s.dof * s.dof * 2 * iface_inplane_pairs(s, i, j);
  for (int t = lane; t < s.dof * prl; t += 64) {
    const int c = t / prl, r = t - c * prl;
    dst[t] = src[c * rl + skip + r];
  }
}
__global__ __launch_bounds__(256) void k_iface_apply(const Slab s, int bc, double *vals, double *b,
                                                     const double *recvK, const double *recvB) {
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t layer = s.ke - 1;
  const int64_t nnode = (s.nsd == 3) ? M * N : M;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (q >= nnode) return;
  int64_t i = q % M, j, k;
  if (s.nsd == 3) { j = q / M; k = layer; } else { j = layer; k = 0; }
  if (bc != BC_POISSON && is_bc_node(s, bc, i, j, k)) return;
  const int crest = (s.nsd == 3) ? span1(i, M) * span1(j, N) : span1(i, M);
  const int cs = span1(layer, (s.nsd == 3) ? s.dim[2] : N);
  const int rl = cs * crest * s.dof, prl = 2 * crest * s.dof, skip = (layer > 0) ? crest * s.dof : 0;
  if (vals && recvK) {
    double *own = vals + row_start(s, i, j, k, 0);
    const double *rcv = recvK + (int64_t)s.dof * s.dof * 2 * iface_inplane_pairs(s, i, j);
    for (int t = lane; t < s.dof * prl; t += 64) {
      const int c = t / prl, r = t - c * prl;
      own[c * rl + skip + r] = own[c * rl + skip + r] + rcv[t];
    }
  }
  if (b && recvB && lane < s.dof) {
    const int64_t lr = (q + nnode * (layer - s.ks)) * s.dof + lane;
    b[lr] = b[lr] + recvB[q * s.dof + lane];
  }
}

// ---------------------------------------------------------------------------
// Rectilinear-mesh test (3-D): x depends on i only, y on j only, z on k only -- bitwise, for every node of the
// slab's coordinate layers.  Then every element is an axis-aligned box and its Jacobian follows from three
// 1-D coordinate lines.  One flag increment per wave that sees a violation.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_check_rectilinear(const Slab s, const double *coords, int64_t nlayers,
                                                           unsigned long long *flag) {
  // the coordinate array as a flat stream of doubles: one x-row of nodes (3M doubles) per pass of the workgroup,
  // 8 B per lane, coalesced; the j- and k-line references are uniform per row, the i-line is 3M hot doubles
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t nrows = N * nlayers;
  const int n3 = (int)(3 * M);
  bool bad = false;
  // CR_H x-rows per pass (rows r + h * gridDim.x): 4 * CR_H 8-byte loads per lane in flight
  constexpr int CR_H = 4;
  for (int64_t r = blockIdx.x; r < nrows; r += CR_H * (int64_t)gridDim.x) {
    const double *row[CR_H];
    double yref[CR_H], zref[CR_H];
#pragma unroll
    for (int h = 0; h < CR_H; h++) {
      const int64_t rh = r + h * (int64_t)gridDim.x < nrows ? r + h * (int64_t)gridDim.x : r;  // (past the end: row r again)
      const int64_t kl = rh / N, j = rh - kl * N;
      row[h] = coords + 3 * M * rh;
      yref[h] = coords[3 * (M * j) + 1];
      zref[h] = coords[3 * (M * N * kl) + 2];
    }
    for (int t0 = threadIdx.x; t0 < n3; t0 += 4 * 256) {  // (clamped, streamed once)
      double v[CR_H][4];
#pragma unroll
      for (int h = 0; h < CR_H; h++)
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int t = t0 + 256 * u;
          v[h][u] = __builtin_nontemporal_load(&row[h][t < n3 ? t : n3 - 1]);
        }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int t = (t0 + 256 * u < n3) ? t0 + 256 * u : n3 - 1, c = t % 3;
        const double xref = coords[t];
#pragma unroll
        for (int h = 0; h < CR_H; h++) bad |= v[h][u] != (c == 0 ? xref : (c == 1 ? yref[h] : zref[h]));
      }
    }
  }
  if (__ballot(bad) != 0ULL && (threadIdx.x & 63) == 0) {
    atomicAdd(flag, 1ULL);      // bad[1]: premise of the level-2 stiffness kernels
    atomicAdd(flag + 2, 1ULL);  // bad[3]: the mesh is not rectilinear (read by the RHS element pass)
  }
}

// the same test for the 2-D drivers: x depends on i only, y on j only
__global__ __launch_bounds__(256) void k_check_rectilinear2d(const Slab s, const double *coords, int64_t nrows,
                                                             unsigned long long *flag) {
  const int64_t M = s.dim[0];
  const int n2 = (int)(2 * M);
  bool bad = false;
  for (int64_t r = blockIdx.x; r < nrows; r += gridDim.x) {
    const double *row = coords + 2 * M * r;
    const double yref = row[1];
    for (int t = threadIdx.x; t < n2; t += 256) {
      const double v = __builtin_nontemporal_load(&row[t]);
      bad |= v != ((t & 1) ? yref : coords[t]);
    }
  }
  if (__ballot(bad) != 0ULL && (threadIdx.x & 63) == 0) {
    atomicAdd(flag, 1ULL);
    atomicAdd(flag + 2, 1ULL);
  }
}

// ---------------------------------------------------------------------------
// N1: fill of the GaussPointCoefficients array on device (the step before the path; S3:1243-1363, E2:618-670).
// S3: 32 lanes per element, lane q < 24 = face Gauss point q (position, E_B, nu_B, grad u of the manufactured
// solution), lane q >= 24 = volume Gauss point q-24 (position, E, nu, body force); the 400 doubles of a record
// are gathered in LDS and written as whole 128-byte lines.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fill_props_s3(const Slab s, const double *coords, double E, double nu,
                                                       int stride, int64_t nel, double *gp) {
  __shared__ __attribute__((aligned(16))) double sh_q[8][416];
  __shared__ double sh_xc[8][24];
  __shared__ double sh_sc[8][3][4][2];  // box elements: sin, cos of pi * abscissa [axis][-g, +g, lower end, upper end]
  constexpr double PI = 3.14159265358979323846264338327950288;
  const int tid = threadIdx.x, el = tid >> 5, q = tid & 31;
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t mx = M - 1, my = N - 1;
  const int64_t e = (int64_t)blockIdx.x * 8 + el;
  const bool live = e < nel;
  if (live && q < 24) {
    const int64_t ex = e % mx, ey = (e / mx) % my, ez = s.els + e / (mx * my);
    const int n = q / 3, d = q % 3;
    const int64_t node = (ex + HDI[n]) + M * ((ey + HDJ[n]) + N * (ez + HDK[n] - s.cks));
    sh_xc[el][q] = coords[3 * node + d];
  }
  __syncthreads();
  // An axis-aligned box (every element of DMDASetUniformCoordinates, S3:1198; tested exactly): the 32 points share 4
  // abscissae per axis, so 12 sincospi per element instead of 6 FP64 sin / cos calls per point (the fill was bound by
  // them: 2.5 ms at 128^3).  sincospi(x) for sin(pi x), cos(pi x): rounding only.
  bool box = live;
  if (live) {
    const double *xc = sh_xc[el];
#pragma unroll
    for (int n = 0; n < 8; n++)
      box = box && xc[3 * n] == xc[3 * (HDI[n] ? 3 : 0)] && xc[3 * n + 1] == xc[3 * (HDJ[n] ? 1 : 0) + 1] &&
            xc[3 * n + 2] == xc[3 * (HDK[n] ? 4 : 0) + 2];
    if (box && q < 12) {
      const int d = q >> 2, t = q & 3;
      const double lo = xc[d], hi = xc[(d == 0 ? 9 : (d == 1 ? 3 : 12)) + d];
      const double a0 = 0.5 * (1.0 + GAUSS_G), a1 = 0.5 * (1.0 - GAUSS_G);
      const double x = (t == 0) ? a0 * lo + a1 * hi : (t == 1) ? a1 * lo + a0 * hi : (t == 2) ? lo : hi;
      double sv, cv;
      sincospi(x, &sv, &cv);
      sh_sc[el][d][t][0] = sv;
      sh_sc[el][d][t][1] = cv;
    }
  }
  __syncthreads();
  if (live) {
    const double *xc = sh_xc[el];
    double *o = sh_q[el];
    const double *Nn = (q < 24) ? c_tab3.NF[q] : c_tab3.N[q - 24];
    double x = 0.0, y = 0.0, z = 0.0;
#pragma unroll
    for (int n = 0; n < 8; n++) {
      x = x + Nn[n] * xc[3 * n];
      y = y + Nn[n] * xc[3 * n + 1];
      z = z + Nn[n] * xc[3 * n + 2];
    }
    double sx, cx, sy, cy, sz, cz;
    if (box) {
      int idx[3];
      if (q < 24) {  // face point: its face's axis at an end, the free axes at -g / +g in the order of S3:256-284
        const int axis = q / 8, fq = q & 3;
        const int a0 = (axis == 0) ? 1 : 0, a1 = (axis == 2) ? 1 : 2;
        idx[axis] = ((q / 4) % 2 == 0) ? 3 : 2;
        idx[a0] = fq & 1;
        idx[a1] = fq >> 1;
      } else {
        const int p = q - 24;
        idx[0] = HDI[p]; idx[1] = HDJ[p]; idx[2] = HDK[p];
      }
      sx = sh_sc[el][0][idx[0]][0]; cx = sh_sc[el][0][idx[0]][1];
      sy = sh_sc[el][1][idx[1]][0]; cy = sh_sc[el][1][idx[1]][1];
      sz = sh_sc[el][2][idx[2]][0]; cz = sh_sc[el][2][idx[2]][1];
    } else {
      sincospi(x, &sx, &cx);
      sincospi(y, &sy, &cy);
      sincospi(z, &sz, &cz);
    }
    if (q < 24) {
      o[24 + 3 * q] = x; o[24 + 3 * q + 1] = y; o[24 + 3 * q + 2] = z;
      o[112 + q] = E;
      o[136 + q] = nu;
#pragma unroll
      for (int c = 0; c < 3; c++) { o[184 + 3 * q + c] = 0.0; o[256 + 3 * q + c] = 0.0; }
      o[328 + 3 * q] = PI * cx * sy * sz;
      o[328 + 3 * q + 1] = PI * sx * cy * sz;
      o[328 + 3 * q + 2] = PI * sx * sy * cz;
    } else {
      const int p = q - 24;
      o[3 * p] = x; o[3 * p + 1] = y; o[3 * p + 2] = z;
      o[96 + p] = E;
      o[104 + p] = nu;
      const double den = 2 * nu * nu + nu - 1;
      o[160 + p] = (E * PI * PI * cx * cz * sy) / (2 * den);
      o[168 + p] = (E * PI * PI * cy * sx * cz) / (2 * den);
      o[176 + p] = (E * PI * PI * sx * sy * sz * (3 * nu - 2)) / den;
    }
    for (int t = 400 + q; t < stride; t += 32) o[t] = 0.0;  // R3: enrich_type / n_enrich = 0
  }
  __syncthreads();
  if (live) {
    double *dst = gp + e * (int64_t)stride;
    if ((stride & 1) == 0) {
      for (int k = q; k < stride / 2; k += 32) {  // whole records, streamed: non-temporal (nothing here reads them back)
        typedef double d2v __attribute__((ext_vector_type(2)));
        const double2 v = reinterpret_cast<const double2 *>(sh_q[el])[k];
        d2v vv;
        vv.x = v.x;
        vv.y = v.y;
        __builtin_nontemporal_store(vv, reinterpret_cast<d2v *>(dst) + k);
      }
    } else {
      for (int k = q; k < stride; k += 32) dst[k] = sh_q[el][k];
    }
  }
}

// E2: one thread per element (24 doubles: Gauss point positions, E, nu, zero body force)
__global__ __launch_bounds__(256) void k_fill_props_e2(const Slab s, const double *coords, double E, double nu,
                                                       int stride, int64_t nel, double *gp) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= nel) return;
  const int64_t M = s.dim[0], mx = M - 1;
  const int64_t ex = e % mx, ey = s.els + e / mx;
  double xc[8];
#pragma unroll
  for (int n = 0; n < 4; n++) {
    const int64_t node = (ex + QDI[n]) + M * (ey + QDJ[n] - s.cks);
    xc[2 * n] = coords[2 * node];
    xc[2 * n + 1] = coords[2 * node + 1];
  }
  double *o = gp + e * (int64_t)stride;
#pragma unroll
  for (int p = 0; p < 4; p++) {
    double gx = 0.0, gy = 0.0;
#pragma unroll
    for (int n = 0; n < 4; n++) {
      gx = gx + c_tab2.N[p][n] * xc[2 * n];
      gy = gy + c_tab2.N[p][n] * xc[2 * n + 1];
    }
    o[2 * p] = gx;
    o[2 * p + 1] = gy;
    o[8 + p] = E;
    o[12 + p] = nu;
    o[16 + p] = 0.0;
    o[20 + p] = 0.0;
  }
  for (int t = 24; t < stride; t++) o[t] = 0.0;
}

// ---------------------------------------------------------------------------
// N3: error integrals against the manufactured solutions (DMDAIntegrateErrors3D, P3:416-532, S3:1041-1151):
// per element  sum_p w detJ sum_i N_i e_i^2  and  sum_p w detJ sum_i |grad N_i|^2 e_i^2  with e = nodal error
// (the reference's nodal form).  One thread per element, fixed-order tree per workgroup -> partial[2*block],
// a second one-workgroup launch sums the partials: deterministic, no atomics.
// ---------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void k_error_integrals(const Slab s, const double *coords, const double *u,
                                                         int64_t nel, double *partial) {
  constexpr int DOF = Traits<KIND>::DOF;
  constexpr double PI = 3.14159265358979323846264338327950288;
  __shared__ double red[2][256];
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t mx = M - 1, my = N - 1;
  double eL2 = 0.0, eH1 = 0.0;
  if (e < nel) {
    const int64_t ex = e % mx, ey = (e / mx) % my, ez = s.els + e / (mx * my);
    double xc[24], err2[8];
#pragma unroll
    for (int n = 0; n < 8; n++) {
      const int64_t node = (ex + HDI[n]) + M * ((ey + HDJ[n]) + N * (ez + HDK[n] - s.cks));
#pragma unroll
      for (int d = 0; d < 3; d++) xc[3 * n + d] = coords[3 * node + d];
      const double ua = sin(PI * xc[3 * n]) * sin(PI * xc[3 * n + 1]) * sin(PI * xc[3 * n + 2]);
      if constexpr (DOF == 1) {
        const double d0 = u[node] - ua;
        err2[n] = d0 * d0;
      } else {  // exact displacement (0, 0, sin sin sin), S3:964-968
        const double d0 = u[3 * node] - 0.0, d1 = u[3 * node + 1] - 0.0, d2 = u[3 * node + 2] - ua;
        err2[n] = d0 * d0 + d1 * d1 + d2 * d2;
      }
    }
    for (int p = 0; p < 8; p++) {
      const double(*G)[8] = c_tab3.G[p];
      const Jac3 jj = jac3(G, xc);
      const double fac = 1.0 * jj.det;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        double g2 = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const double gk = G[0][i] * jj.iJ[k][0] + G[1][i] * jj.iJ[k][1] + G[2][i] * jj.iJ[k][2];
          g2 += gk * gk;
        }
        eL2 += fac * c_tab3.N[p][i] * err2[i];
        eH1 += fac * g2 * err2[i];
      }
    }
  }
  red[0][threadIdx.x] = eL2;
  red[1][threadIdx.x] = eH1;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      red[0][threadIdx.x] += red[0][threadIdx.x + st];
      red[1][threadIdx.x] += red[1][threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = red[0][0];
    partial[2 * blockIdx.x + 1] = red[1][0];
  }
}

__global__ __launch_bounds__(256) void k_sum_partials(const double *partial, int64_t n, double *out) {
  __shared__ double red[2][256];
  double a0 = 0.0, a1 = 0.0;
  for (int64_t q = threadIdx.x; q < n; q += 256) {
    a0 += partial[2 * q];
    a1 += partial[2 * q + 1];
  }
  red[0][threadIdx.x] = a0;
  red[1][threadIdx.x] = a1;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      red[0][threadIdx.x] += red[0][threadIdx.x + st];
      red[1][threadIdx.x] += red[1][threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = red[0][0];
    out[1] = red[1][0];
  }
}

// ---------------------------------------------------------------------------
// pattern (A12) and explicit scatter map (A13)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pattern(const Slab s, int64_t nrows, int64_t *rowptr, int32_t *colidx) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r > nrows) return;
  const int64_t M = s.dim[0], N = s.dim[1], P = s.dim[2];
  const int64_t first = (s.nsd == 3) ? M * N * s.ks : M * s.ks;  // first owned node
  if (r == nrows) {
    // one past the last owned layer: pre1() clamps, so this is the slab's stored-entry count
    const int64_t j = (s.nsd == 3) ? 0 : s.ke, k = (s.nsd == 3) ? s.ke : 0;
    rowptr[r] = (int64_t)s.dof * s.dof * (pairs_before(s, 0, j, k) - s.before0);
    return;
  }
  const int c = (int)(r % s.dof);
  const int64_t n = first + r / s.dof;
  const int64_t i = n % M, j = (n / M) % N, k = n / (M * N);
  int64_t pos = row_start(s, i, j, k, c);
  rowptr[r] = pos;
  for (int64_t kk = (k > 0 ? k - 1 : 0); kk <= (k < P - 1 ? k + 1 : P - 1); kk++)
    for (int64_t jj = (j > 0 ? j - 1 : 0); jj <= (j < N - 1 ? j + 1 : N - 1); jj++)
      for (int64_t ii = (i > 0 ? i - 1 : 0); ii <= (i < M - 1 ? i + 1 : M - 1); ii++)
        for (int cc = 0; cc < s.dof; cc++) colidx[pos++] = (int32_t)(s.dof * (ii + M * (jj + N * kk)) + cc);
}

// one thread per (element, local row); writes the nen*dof positions of that row
__global__ __launch_bounds__(256) void k_scatter_map(const Slab s, int64_t nel, int nen, int64_t *map) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int nr = nen * s.dof;
  if (t >= nel * nr) return;
  const int64_t e = t / nr;
  const int r = (int)(t % nr), ar = r / s.dof, cr = r % s.dof;
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t mx = M - 1, my = N - 1;
  int64_t ex, ey, ez;
  if (s.nsd == 3) { ex = e % mx; ey = (e / mx) % my; ez = s.els + e / (mx * my); }
  else            { ex = e % mx; ey = s.els + e / mx; ez = 0; }
  const int di = (s.nsd == 3) ? HDI[ar] : QDI[ar], dj = (s.nsd == 3) ? HDJ[ar] : QDJ[ar];
  const int dk = (s.nsd == 3) ? HDK[ar] : 0;
  const int64_t i = ex + di, j = ey + dj, k = ez + dk;
  const int64_t sl = (s.nsd == 3) ? k : j;
  int64_t *o = map + t * nr;
  if (sl < s.ks || sl >= s.ke) {
    for (int q = 0; q < nr; q++) o[q] = -1;
    return;
  }
  const int64_t rs = row_start(s, i, j, k, cr);
  const int cntx = span1(i, s.dim[0]), cnty = span1(j, s.dim[1]);
  for (int q = 0; q < nr; q++) {
    const int ac = q / s.dof, cc = q % s.dof;
    const int ci = (s.nsd == 3) ? HDI[ac] : QDI[ac], cj = (s.nsd == 3) ? HDJ[ac] : QDJ[ac];
    const int ck = (s.nsd == 3) ? HDK[ac] : 0;
    const int dx = ci - di, dy = cj - dj, dz = ck - dk;
    const int xi = dx + 1 - (i == 0), yi = dy + 1 - (j == 0), zi = (s.nsd == 3) ? dz + 1 - (k == 0) : 0;
    o[q] = rs + (int64_t)s.dof * ((zi * cnty + yi) * cntx + xi) + cc;
  }
}

// ---------------------------------------------------------------------------
// PETSc's truncated MatZeroRows storage
// ---------------------------------------------------------------------------
__device__ inline bool is_bc_row(const Slab &s, int bc, int64_t r /*local row*/) {
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t first = (s.nsd == 3) ? M * N * s.ks : M * s.ks;
  const int64_t n = first + r / s.dof;
  return is_bc_node(s, bc, n % M, (n / M) % N, n / (M * N));
}
__global__ __launch_bounds__(256) void k_bc_rowlen(const Slab s, int bc, int64_t nrows, const int64_t *rowptr,
                                                   int64_t *len) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r > nrows) return;
  len[r] = (r == nrows) ? 0 : (is_bc_row(s, bc, r) ? 1 : rowptr[r + 1] - rowptr[r]);
}
__global__ __launch_bounds__(256) void k_bc_compress(const Slab s, int bc, int64_t nrows, const int64_t *rowptr,
                                                     const int32_t *colidx, const double *vals,
                                                     const int64_t *rowptr_new, int32_t *colidx_new,
                                                     double *vals_new) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= nrows) return;
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t first = (s.nsd == 3) ? M * N * s.ks : M * s.ks;
  int64_t o = rowptr_new[r];
  if (is_bc_row(s, bc, r)) {
    colidx_new[o] = (int32_t)(first * s.dof + r);
    vals_new[o] = 1.0;
    return;
  }
  for (int64_t q = rowptr[r]; q < rowptr[r + 1]; q++, o++) {
    colidx_new[o] = colidx[q];
    vals_new[o] = vals[q];
  }
}

// ---------------------------------------------------------------------------
// N2: coordinate (COO) form of the pattern in the order of the CSR value array, for MatSetPreallocationCOO /
// MatSetValuesCOO: one wave per row.  I = PetscInt (32 or 64 bit).
// ---------------------------------------------------------------------------
template <typename I>
__global__ __launch_bounds__(256) void k_coo_indices(int64_t nrows, int64_t row0, const int64_t *rowptr,
                                                     const int32_t *colidx, I *coo_i, I *coo_j) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nrows) return;
  const int64_t e = rowptr[r + 1];
  for (int64_t t = rowptr[r] + (threadIdx.x & 63); t < e; t += 64) {
    coo_i[t] = (I)(row0 + r);
    coo_j[t] = (I)colidx[t];
  }
}

// ---------------------------------------------------------------------------
// N4 (second half): rigid-body near-nullspace vectors, the input of MatNullSpaceCreateRigidBody (E2:682-685,
// S3:1374-1377).  Pass 1: first and second moments of the owned nodes' coordinates, one row of 10 partial sums
// per workgroup (fixed order: deterministic).  Pass 2: the modes as a linear combination C of the raw vectors
// (translations sN e_k; rotations (y,-x,0), (0,-z,y), (z,0,-x); 2-D: (-y,x)).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_coord_moments(const Slab s, const double *coords, int64_t nown,
                                                       double *partial /*[gridDim.x][10]*/) {
  __shared__ double red[10][256];
  const int64_t plane = (s.nsd == 3) ? s.dim[0] * s.dim[1] : s.dim[0];
  const double *c0 = coords + (s.ks - s.cks) * plane * s.nsd;
  double m[10];
#pragma unroll
  for (int q = 0; q < 10; q++) m[q] = 0.0;
  for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < nown; n += (int64_t)gridDim.x * 256) {
    const double x = c0[s.nsd * n], y = c0[s.nsd * n + 1], z = (s.nsd == 3) ? c0[3 * n + 2] : 0.0;
    m[0] += 1.0; m[1] += x; m[2] += y; m[3] += z;
    m[4] += x * x; m[5] += y * y; m[6] += z * z;
    m[7] += x * y; m[8] += y * z; m[9] += z * x;
  }
#pragma unroll
  for (int q = 0; q < 10; q++) red[q][threadIdx.x] = m[q];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st)
#pragma unroll
      for (int q = 0; q < 10; q++) red[q][threadIdx.x] += red[q][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x < 10) partial[10 * (int64_t)blockIdx.x + threadIdx.x] = red[threadIdx.x][0];
}
__global__ __launch_bounds__(64) void k_sum_rows10(const double *partial, int64_t n, double *out) {
  if (threadIdx.x >= 10) return;
  double a = 0.0;
  for (int64_t q = 0; q < n; q++) a += partial[10 * q + threadIdx.x];
  out[threadIdx.x] = a;
}
struct RigidArgs {
  double C[6][6];  // mode m = sum_k C[m][k] raw_k
  double sN;
  int32_t nmodes;
};
__global__ __launch_bounds__(256) void k_rigid_modes(const Slab s, const double *coords, int64_t nown,
                                                     const RigidArgs a, double *modes /*[nmodes][nown*nsd]*/) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= nown) return;
  const int nsd = s.nsd;
  const int64_t plane = (nsd == 3) ? s.dim[0] * s.dim[1] : s.dim[0];
  const double *c0 = coords + (s.ks - s.cks) * plane * nsd;
  const double x = c0[nsd * n], y = c0[nsd * n + 1], z = (nsd == 3) ? c0[3 * n + 2] : 0.0;
  double raw[6][3];
#pragma unroll
  for (int k = 0; k < 6; k++)
#pragma unroll
    for (int c = 0; c < 3; c++) raw[k][c] = 0.0;
  if (nsd == 3) {
    raw[0][0] = a.sN; raw[1][1] = a.sN; raw[2][2] = a.sN;
    raw[3][0] = y;  raw[3][1] = -x;
    raw[4][1] = -z; raw[4][2] = y;
    raw[5][0] = z;  raw[5][2] = -x;
  } else {
    raw[0][0] = a.sN; raw[1][1] = a.sN;
    raw[2][0] = -y; raw[2][1] = x;
  }
  const int64_t nr = nown * nsd;
  for (int m = 0; m < a.nmodes; m++)
    for (int c = 0; c < nsd; c++) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < 6; k++) v += a.C[m][k] * raw[k][c];
      modes[m * nr + nsd * n + c] = v;
    }
}

// ---------------------------------------------------------------------------
// A11 in its general form: MatZeroRows(A, n, rows, 1.0, NULL, NULL) + VecSetValues(b, n, rows, vals, INSERT_VALUES)
// for a caller-supplied list of GLOBAL rows (rows owned by other slabs are skipped).  One wave per listed row.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_zero_rows(int64_t n, const int64_t *rows, const double *vals, int64_t row0,
                                                   int64_t nrows, const int64_t *rowptr, const int32_t *colidx,
                                                   double *csr, double *b) {
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= n) return;
  const int64_t g = rows[q], r = g - row0;
  if (r < 0 || r >= nrows) return;
  const int lane = threadIdx.x & 63;
  if (csr)
    for (int64_t t = rowptr[r] + lane; t < rowptr[r + 1]; t += 64) csr[t] = (colidx[t] == g) ? 1.0 : 0.0;
  if (b && lane == 0) b[r] = vals ? vals[q] : 0.0;
}

// ---------------------------------------------------------------------------
// Opt-in symmetric elimination (north_star (5) "row/column elimination"; NOT what the reference does: it zeroes rows
// only, S3:1518 MatZeroRows(A,n,ids,1.0,0,0), P3:286-290 -- see include/femasm.h).  After the rows-only system is
// complete: for every non-Dirichlet row r and every stored column c that is a Dirichlet dof with value g_c:
// b[r] -= K[r][c] g_c, K[r][c] = 0.  One wave per row; the lanes' products are summed in a fixed tree (deterministic).
// ---------------------------------------------------------------------------
__device__ inline double dirichlet_value(const Slab &s, int bc, const double *coords, int64_t i, int64_t j, int64_t k,
                                         int c) {
  if (bc == BC_E2) return (c == 0) ? ((i == s.dim[0] - 1) ? -1.0 : 1.0) : 0.0;   // E2:882-885
  if (bc == BC_S3 && c == 2) {                                                    // S3:964-968 via S3:1501-1505
    constexpr double PI = 3.14159265358979323846264338327950288;
    const int64_t node = i + s.dim[0] * (j + s.dim[1] * (k - s.cks));
    return sin(PI * coords[3 * node]) * sin(PI * coords[3 * node + 1]) * sin(PI * coords[3 * node + 2]);
  }
  return 0.0;
}
__global__ __launch_bounds__(256) void k_eliminate_columns(const Slab s, int bc, const double *coords, int64_t nrows,
                                                           const int64_t *rowptr, const int32_t *colidx, double *vals,
                                                           double *b) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= nrows) return;
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t first = (s.nsd == 3) ? M * N * s.ks : M * s.ks;
  {
    const int64_t n = first + r / s.dof;
    if (is_bc_node(s, bc, n % M, (n / M) % N, n / (M * N))) return;  // Dirichlet rows stay unit rows
  }
  double lift = 0.0;
  for (int64_t t = rowptr[r] + lane; t < rowptr[r + 1]; t += 64) {
    const int64_t col = colidx[t], n = col / s.dof;
    const int c = (int)(col - n * s.dof);
    const int64_t i = n % M, j = (n / M) % N, k = n / (M * N);
    if (is_bc_node(s, bc, i, j, k)) {
      if (b) lift += vals[t] * dirichlet_value(s, bc, coords, i, j, k, c);
      vals[t] = 0.0;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) lift += __shfl_down(lift, off);
  if (b && lane == 0) b[r] = b[r] - lift;
}

}  // namespace femasm
