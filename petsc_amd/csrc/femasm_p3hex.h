// femasm_p3hex.h -- level 0 of the Poisson kind (any hexahedra) since round 3: k_assemble_p3_hex.
//
// Reference: FEM_poisson3D/FEM.cpp:185-243 (FormStiffMatrix: Ke = sum_p gradN^T gradN detJ over the 2x2x2 Gauss points
// of each element, ADD_VALUES into the 27-point rows) on a mesh whose elements are general hexahedra.
//
// k_assemble_ke_p3 (rounds 1-2, femasm_sweep.h) forms each of the 36 unique entries of an element matrix as
// sum_p ghat_a^T C_p ghat_b with literal reference gradients: 8 x 6 FMAs per entry, 1728 per element, on top of eight
// Jacobians from the 24 coordinates (72 FMAs each) -- 2900 FP64 instructions per element, and its 15 x 3 node tile forms
// 64 element matrices for 45 rows.  The kernel was bound by FP64 issue (3.6 ms at 256^3, 71 % of the issue slots).
// This kernel does the same sums in the tensor-product form the trilinear element allows:
//
//   geometry    x(xi) = sum_{ijk} c_ijk xi^i eta^j zeta^k, c = the 8-point Hadamard transform of the node coordinates
//               (24 additions per component).  dx/dxi = c100 + eta c110 + zeta (c101 + eta c111) does not depend on xi:
//               each row of J takes FOUR values over the eight Gauss points (8 FMAs per component and row, 72 in all
//               instead of 576).
//   C_p         = detJ iJ^T iJ = (A_k . A_l) / det with A_0 = r1 x r2, A_1 = r2 x r0, A_2 = r0 x r1 the cofactor rows and
//               det = r0 . A_0: 50 instructions per Gauss point, no 3x3 inverse (one reciprocal: hardware estimate + two
//               Newton steps, relative error 1e-16; results are compared with the reference's at 1e-13).  The factors 1/8
//               of the gradients and of J are powers of two and are applied once, exactly, as 1/512.
//   sums over   ghat_a[0](p) = sx_a/8 (1 + sy_a eta_p)(1 + sz_a zeta_p) etc., so ghat_a[k] ghat_b[l] is a product of one
//   Gauss       factor per axis that depends on (a, b) only through the signs of a and b on that axis:
//   points        k = l:  sum_p C_kk(p) YY(eta_p) ZZ(zeta_p), YY in {(1+-eta)^2, 1-eta^2}: 9 sums per k, 34 instructions
//                 k < l:  sum_p C_kl(p) X_b(xi_p) Y_a(eta_p) ZZ(zeta_p): 12 sums per (k, l), 56 instructions
//               270 instructions for the 63 table values, then every entry K_ab is 9 signed additions of table values
//               (324 for the 36 entries): 594 instead of 1728.
//
// 1140 FP64 instructions per element instead of 2900, and the tile is 15 x 7 nodes (128 element matrices for 105 rows:
// 1.22 instead of 1.42).  The additions are in a different order than the reference's (per Gauss point, then per element):
// the results differ from k_assemble_ke_p3's in the last bits, like those of every level-1 and level-2 kernel.
//
// Workgroup = 2 waves, lane = element in the element phase, lane = node (lx = lane & 15 < 15, ly = lane >> 4 < 7) in the
// gather.  Per element layer L of a z-march:
//   1  node layer L+2 of the halo'd tile requested (registers; into the LDS ring of two layers after phase 3)  barrier
//   2  element matrices of layer L -> sh_ke[element][36] (stride 37)                                barrier
//   3  node lanes: finish the rows of node layer L (middle third += layer L, upper third) into registers; start node layer
//      L+1 (lower third, middle third) -- the carry, 18 doubles per lane                            barrier
//   4  the finished rows -> LDS in HBM order (the area of sh_ke), each tile row from the parity of its HBM address
//                                                                                                   barrier
//   5  tile rows -> the CSR array, 16 B per lane from the 128-byte boundary below each run (non-temporal)
// 45 KB of LDS: three workgroups per CU.  Every CSR value has one writer and a fixed order of additions.
#pragma once
#include "femasm_sweep.h"

#ifndef PH_EXP
#define PH_EXP 0  // profiling builds only (-DPH_EXP=n): 1 no global stores, 2 no element arithmetic, 4 no gather,
#endif            // 8 wave placement of k_assemble_p3_hex4, 9 phase clocks (printed when the plan is destroyed)
#if PH_EXP == 9
#define PHX(q) do { const unsigned long long n_ = __builtin_readcyclecounter(); phc[q] += n_ - pht; pht = n_; } while (0)
#else
#define PHX(q)
#endif

namespace femasm {

constexpr int PH_TX = 15, PH_TY = 7;             // nodes of a tile (KArgs::zc: node layers per workgroup)  // nodes of a tile, node layers per workgroup
constexpr int PH_HX = PH_TX + 2, PH_HY = PH_TY + 2;  // halo'd node layer
constexpr int PH_RS = 408;                           // doubles of a staged tile row (15 x 27 = 405, + parity, even)
constexpr int PH_NT = 128;

__device__ constexpr int ph_bx(int n) { return (n >> 1) & 1; }
__device__ constexpr int ph_by(int n) { return ((n & 3) == 1 || (n & 3) == 2) ? 1 : 0; }
__device__ constexpr int ph_bz(int n) { return n >> 2; }
// type of the pair of signs on one axis: 0 both +, 1 both -, 2 different
__device__ constexpr int ph_type(int ba, int bb) { return (ba && bb) ? 0 : ((!ba && !bb) ? 1 : 2); }

__device__ inline double ph_recip(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  return fma(fma(-x, r, 1.0), r, r);
}

// typed sums along one axis: in v[2] (index 0: -g, 1: +g) -> out[3] (both +, both -, different)
__device__ inline void ph_typed(double v0, double v1, double &pp, double &mm, double &pm) {
  constexpr double al = (1.0 + GAUSS_G) * (1.0 + GAUSS_G), be = (1.0 - GAUSS_G) * (1.0 - GAUSS_G), ga = 1.0 - GAUSS_G * GAUSS_G;
  pp = fma(al, v1, be * v0);
  mm = fma(be, v1, al * v0);
  pm = ga * (v0 + v1);
}
// linear sums along one axis: out[0] for a node on the - side (1 - xi), out[1] for a node on the + side (1 + xi)
__device__ inline void ph_lin(double v0, double v1, double &m, double &p) {
  constexpr double a = 1.0 + GAUSS_G, b = 1.0 - GAUSS_G;
  m = fma(b, v1, a * v0);
  p = fma(a, v1, b * v0);
}

// k = l: F[t1][t2] = sum over the axis k plainly, typed over the two other axes (in axis order); D[i1][i2] is that plain sum
__device__ inline void ph_diag_table(const double (&D)[2][2], double (&F)[3][3]) {
  double E[3][2];
#pragma unroll
  for (int c = 0; c < 2; c++) ph_typed(D[0][c], D[1][c], E[0][c], E[1][c], E[2][c]);
#pragma unroll
  for (int t = 0; t < 3; t++) ph_typed(E[t][0], E[t][1], F[t][0], F[t][1], F[t][2]);
}
// k < l: T[sk][sl][tm] from P[sk][il][im], the linear sums over the axis of k (formed Gauss point by Gauss point);
// il: the axis of l, im: the third axis
__device__ inline void ph_off_table(const double (&P)[2][2][2], double (&T)[2][2][3]) {
  double Q[2][2][2];
#pragma unroll
  for (int sk = 0; sk < 2; sk++)
#pragma unroll
    for (int im = 0; im < 2; im++) ph_lin(P[sk][0][im], P[sk][1][im], Q[sk][0][im], Q[sk][1][im]);
#pragma unroll
  for (int sk = 0; sk < 2; sk++)
#pragma unroll
    for (int sl = 0; sl < 2; sl++) ph_typed(Q[sk][sl][0], Q[sk][sl][1], T[sk][sl][0], T[sk][sl][1], T[sk][sl][2]);
}

// the 36 unique entries of the element matrix of the hexahedron with nodes xs (LDS: 3 doubles per node, node (cx, cy, cz)
// of the halo'd tile at 3 (cx + PH_HX (cy + PH_HY cz'))), cz' the ring slot) -> ke[36].  Returns the number of Gauss points
// with det J <= 0 (or NaN).  scale: 1 for an element that is assembled here, 0 otherwise (its matrix is zero).
__device__ __forceinline__ int ph_element_matrix(const double *__restrict__ xlo, const double *__restrict__ xhi, double scale,
                                        double *__restrict__ ke) {
  // ---- Hadamard coefficients c[i][j][k][component] (unnormalised: 8 x the polynomial coefficients) ----
  double c[2][2][2][3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    double v[2][2][2];
#pragma unroll
    for (int bz = 0; bz < 2; bz++)
#pragma unroll
      for (int by = 0; by < 2; by++)
#pragma unroll
        for (int bx = 0; bx < 2; bx++) v[bx][by][bz] = (bz ? xhi : xlo)[3 * (bx + PH_HX * by) + d];
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
  // ---- the rows of 8 J: four values each ----
  double R0[2][2][3], R1[2][2][3], R2[2][2][3];
#pragma unroll
  for (int d = 0; d < 3; d++)
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const double gq = q ? GAUSS_G : -GAUSS_G;
      const double u0 = fma(gq, c[1][1][0][d], c[1][0][0][d]), v0 = fma(gq, c[1][1][1][d], c[1][0][1][d]);  // eta = gq
      const double u1 = fma(gq, c[1][1][0][d], c[0][1][0][d]), v1 = fma(gq, c[1][1][1][d], c[0][1][1][d]);  // xi = gq
      const double u2 = fma(gq, c[1][0][1][d], c[0][0][1][d]);                                              // xi = gq
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const double gr = r ? GAUSS_G : -GAUSS_G;
        R0[q][r][d] = fma(gr, v0, u0);  // dx/dxi   at (eta_q, zeta_r)
        R1[q][r][d] = fma(gr, v1, u1);  // dx/deta  at (xi_q, zeta_r)
        R2[q][r][d] = fma(gr, v1, u2);  // dx/dzeta at (xi_q, eta_r)
      }
    }
  // ---- C_p / 512 at the eight Gauss points, summed over the first axis on the way (the eight values of the six fields
  // never exist together: registers).  D: plain sums (k = l), P: linear sums over the axis of k (k < l) ----
  double D00[2][2], D11[2][2], D22[2][2], P01[2][2][2], P02[2][2][2], P12[2][2][2];
  int nbad = 0;
  constexpr double wa = 1.0 + GAUSS_G, wb = 1.0 - GAUSS_G;
#pragma unroll
  for (int ia = 0; ia < 2; ia++)
#pragma unroll
    for (int ib = 0; ib < 2; ib++)
#pragma unroll
      for (int ic = 0; ic < 2; ic++) {
        const double *r0 = R0[ib][ic], *r1 = R1[ia][ic], *r2 = R2[ia][ib];
        double A[3][3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          const int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
          A[0][d] = fma(r1[d1], r2[d2], -(r1[d2] * r2[d1]));
          A[1][d] = fma(r2[d1], r0[d2], -(r2[d2] * r0[d1]));
          A[2][d] = fma(r0[d1], r1[d2], -(r0[d2] * r1[d1]));
        }
        const double det = fma(r0[2], A[0][2], fma(r0[1], A[0][1], r0[0] * A[0][0]));  // 512 det J
        nbad += !(det > 0.0);
        // scale = 1/512, or 0 for an element that is not assembled here (clamped coordinates: det may be 0)
        const double rd = (scale != 0.0) ? ph_recip(det) * scale : 0.0;
        const double c00 = fma(A[0][2], A[0][2], fma(A[0][1], A[0][1], A[0][0] * A[0][0])) * rd;
        const double c11 = fma(A[1][2], A[1][2], fma(A[1][1], A[1][1], A[1][0] * A[1][0])) * rd;
        const double c22 = fma(A[2][2], A[2][2], fma(A[2][1], A[2][1], A[2][0] * A[2][0])) * rd;
        const double c01 = fma(A[0][2], A[1][2], fma(A[0][1], A[1][1], A[0][0] * A[1][0])) * rd;
        const double c02 = fma(A[0][2], A[2][2], fma(A[0][1], A[2][1], A[0][0] * A[2][0])) * rd;
        const double c12 = fma(A[1][2], A[2][2], fma(A[1][1], A[2][1], A[1][0] * A[2][0])) * rd;
        D00[ib][ic] = ia ? D00[ib][ic] + c00 : c00;  // over xi;   [eta][zeta]
        D11[ia][ic] = ib ? D11[ia][ic] + c11 : c11;  // over eta;  [xi][zeta]
        D22[ia][ib] = ic ? D22[ia][ib] + c22 : c22;  // over zeta; [xi][eta]
        // (k, l, m) = (xi, eta, zeta), (xi, zeta, eta), (eta, zeta, xi); side 0 = a node on the - side of the axis of k
        P01[0][ib][ic] = ia ? fma(wb, c01, P01[0][ib][ic]) : wa * c01;
        P01[1][ib][ic] = ia ? fma(wa, c01, P01[1][ib][ic]) : wb * c01;
        P02[0][ic][ib] = ia ? fma(wb, c02, P02[0][ic][ib]) : wa * c02;
        P02[1][ic][ib] = ia ? fma(wa, c02, P02[1][ic][ib]) : wb * c02;
        P12[0][ic][ia] = ib ? fma(wb, c12, P12[0][ic][ia]) : wa * c12;
        P12[1][ic][ia] = ib ? fma(wa, c12, P12[1][ic][ia]) : wb * c12;
      }
  // ---- tables ----
  double F00[3][3], F11[3][3], F22[3][3], T01[2][2][3], T02[2][2][3], T12[2][2][3];
  ph_diag_table(D00, F00);
  ph_diag_table(D11, F11);
  ph_diag_table(D22, F22);
  ph_off_table(P01, T01);
  ph_off_table(P02, T02);
  ph_off_table(P12, T12);
  // ---- the 36 entries: nine signed table values each ----
#pragma unroll
  for (int q = 0; q < 36; q++) {
    const int na = KE_PAIRS.a[q], nb = KE_PAIRS.b[q];
    const int ax = ph_bx(na), ay = ph_by(na), az = ph_bz(na), bx = ph_bx(nb), by = ph_by(nb), bz = ph_bz(nb);
    const int tx = ph_type(ax, bx), ty = ph_type(ay, by), tz = ph_type(az, bz);
    auto sg = [](int p, int q2, double t) { return (p == q2) ? t : -t; };  // the product of two signs
    // four partial sums, each a function of a few sign bits only: entries that share one share its instructions
    // (common subexpressions after unrolling: 27 + 3 x ~30 distinct partial sums for the 36 entries)
    const double d3 = (sg(ax, bx, F00[ty][tz]) + sg(ay, by, F11[tx][tz])) + sg(az, bz, F22[tx][ty]);
    const double p01 = sg(ax, by, T01[bx][ay][tz]) + sg(ay, bx, T01[ax][by][tz]);
    const double p02 = sg(ax, bz, T02[bx][az][ty]) + sg(az, bx, T02[ax][bz][ty]);
    const double p12 = sg(ay, bz, T12[by][az][tx]) + sg(az, by, T12[ay][bz][tx]);
    ke[q] = (d3 + p01) + (p02 + p12);
  }
  return nbad;
}

__global__ __launch_bounds__(PH_NT) void k_assemble_p3_hex(const Slab s, const KArgs a) {
  if (a.cond && a.bad[1] == 0) return;
  constexpr int NXL = PH_HX * PH_HY, NXD = 3 * NXL;      // nodes / doubles of a halo'd layer
  constexpr int NLD = (NXD + PH_NT - 1) / PH_NT;         // coordinate doubles per lane and layer (4)
  __shared__ __attribute__((aligned(16))) double sh_ke[128 * KE_KS];  // element matrices; then the staged rows
  __shared__ double sh_x[2 * NXD];                       // ring of two node layers, [cy][cx][component]
  static_assert(16 + PH_TY * PH_RS + 4 * PH_NT <= 128 * KE_KS, "the staged rows alias the element matrices");
  const int tid = threadIdx.x;
  const int lx = tid & 15, ly = tid >> 4;                // element (ex, ey) of the layer / node of the tile
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t i0 = a.x0 + (int64_t)blockIdx.x * PH_TX, j0 = a.y0 + (int64_t)blockIdx.y * PH_TY;
  const int64_t zc0 = a.s0 + (int64_t)blockIdx.z * a.zc;
  const int64_t zc1 = (zc0 + a.zc < a.s1) ? zc0 + a.zc : a.s1;
  const int nsteps = (int)(zc1 - zc0) + 1;
  const int nvx = (a.x1 - i0 < PH_TX) ? (int)(a.x1 - i0) : PH_TX;
  const int nvy = (a.y1 - j0 < PH_TY) ? (int)(a.y1 - j0) : PH_TY;
  const int n = nvx * 27;                                // doubles of a tile row's run in the CSR array
  const bool mynode = lx < nvx && ly < nvy;
  const int glx = (lx < PH_TX) ? lx : 0, gly = (ly < PH_TY) ? ly : 0;  // idle lanes gather node (0, 0) again: reads in range
  const int64_t gex = i0 - 1 + lx, gey = j0 - 1 + ly;
  const bool exy_ok = gex >= 0 && gex < M - 1 && gey >= 0 && gey < N - 1;
  double lower[9], mid[9];
#pragma unroll
  for (int q = 0; q < 9; q++) lower[q] = mid[q] = 0.0;

  // a halo'd node layer is 9 runs of 51 doubles (clamped at the mesh's edge): lane -> doubles tid + 128 q of the layer,
  // whose place in LDS is the same index; the offsets inside a layer are formed once
  unsigned xoff[NLD];  // (the launch checks 3 M N < 2^31)
#pragma unroll
  for (int q = 0; q < NLD; q++) {
    int d = tid + PH_NT * q;
    d = d < NXD ? d : NXD - 1;
    const int cy = d / (3 * PH_HX), col = d - 3 * PH_HX * cy, cx = col / 3, comp = col - 3 * cx;
    int64_t gi = i0 - 1 + cx, gj = j0 - 1 + cy;
    gi = gi < 0 ? 0 : (gi > M - 1 ? M - 1 : gi);
    gj = gj < 0 ? 0 : (gj > N - 1 ? N - 1 : gj);
    xoff[q] = (unsigned)(3 * (gi + M * gj) + comp);
  }
  double xn[NLD];  // the node layer requested one step ahead
  auto request_layer = [&](int64_t gk) {
    gk = gk < s.cks ? s.cks : (gk > a.cke - 1 ? a.cke - 1 : gk);
    const double *base = a.coords + 3 * M * N * (gk - s.cks);
#pragma unroll
    for (int q = 0; q < NLD; q++) xn[q] = base[xoff[q]];
  };
  auto commit_layer = [&](int slot) {
#pragma unroll
    for (int q = 0; q < NLD; q++)
      if (q < NLD - 1 || tid < NXD - PH_NT * (NLD - 1)) sh_x[slot * NXD + tid + PH_NT * q] = xn[q];
  };
  // the runs of the tile's rows in the CSR array: interior nodes, so the run of (row r, node layer k) starts at
  // o00 + r sj + (k - zc0) sk
  double *const o00 = a.out + (row_start(s, i0, j0, zc0, 0) - a.base_off);
  const int64_t sj = row_start(s, i0, j0 + 1, zc0, 0) - row_start(s, i0, j0, zc0, 0);
  const int64_t sk = row_start(s, i0, j0, zc0 + 1, 0) - row_start(s, i0, j0, zc0, 0);
  const unsigned sji = __builtin_amdgcn_readfirstlane((unsigned)sj);
  double *oL = o00;  // the run of tile row 0 in the node layer that the step finishes

  request_layer(zc0 - 1);
  commit_layer(0);
  request_layer(zc0);
  commit_layer(1);
#if PH_EXP == 9
  unsigned long long phc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pht = __builtin_readcyclecounter();
#endif
  for (int st = 0; st < nsteps; st++) {
    const int64_t L = zc0 - 1 + st;  // element layer of this step; node layers L (slot st & 1) and L + 1
    PHX(7);
    // node layer L + 2, in flight during this step's arithmetic and committed after it, in front of this step's stores:
    // gfx9 retires a wave's loads and stores through one in-order counter, so a load waited for behind stores waits for
    // their acknowledgement as well
    if (st + 1 < nsteps) request_layer(L + 2);
    PHX(0);
    __syncthreads();  // every lane is through the previous step's copy-out (sh_ke is free); node layer L + 1 is staged
    PHX(1);

    {  // ---- element (lx, ly) of layer L ----
      const bool eok = exy_ok && L >= s.els && L < s.ele;
      const double *xlo = sh_x + (st & 1) * NXD + 3 * (lx + PH_HX * ly);
      const double *xhi = sh_x + ((st + 1) & 1) * NXD + 3 * (lx + PH_HX * ly);
#if PH_EXP == 2
      int nbad = 0;
      for (int q = 0; q < 36; q++) sh_ke[tid * KE_KS + q] = xlo[q % 3] + xhi[q % 3];
#else
      const int nbad = ph_element_matrix(xlo, xhi, eok ? 1.0 / 512.0 : 0.0, sh_ke + tid * KE_KS);
#endif
      if (eok && nbad) atomicAdd(a.bad, (unsigned long long)nbad);
    }
    PHX(2);
    __syncthreads();  // element matrices complete
    PHX(3);

    double row[27];
    const bool finish = st >= 1, start = st < nsteps - 1;
    if (finish) {  // node layer L: the elements above it
#pragma unroll
      for (int q = 0; q < 9; q++) {
        row[q] = lower[q];
        row[9 + q] = mid[q];
        row[18 + q] = 0.0;
      }
#if PH_EXP != 4
      ke_gather_p3<0, 0>(sh_ke, glx, gly, row + 9);
      ke_gather_p3<0, 1>(sh_ke, glx, gly, row + 18);
#endif
    }
    if (start) {  // node layer L + 1: the elements below it
#pragma unroll
      for (int q = 0; q < 9; q++) lower[q] = mid[q] = 0.0;
#if PH_EXP != 4
      ke_gather_p3<1, 0>(sh_ke, glx, gly, lower);
      ke_gather_p3<1, 1>(sh_ke, glx, gly, mid);
#else
      lower[0] = sh_ke[tid];
#endif
    }
    if (st + 1 < nsteps) commit_layer(st & 1);  // over node layer L, which the element phase is through (barrier above)
    PHX(4);
    if (!finish) continue;  // (uniform: the next step's first barrier orders this step's reads before its writes)
    __syncthreads();        // every lane has read what it needs from sh_ke
    PHX(5);

    // ---- the rows of node layer L -> LDS in HBM order; tile row r starts at the parity of its HBM address ----
    double *stg = sh_ke + 16;  // (16: the copy-out reads whole chunks from the 128-byte boundary below a run, unconditionally)
    const unsigned oLi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(oL) >> 3));  // (low bits: parities, heads)
    if (mynode) {
      double *dst = stg + ly * PH_RS + ((oLi + ly * sji) & 1) + lx * 27;
#pragma unroll
      for (int q = 0; q < 27; q++) dst[q] = row[q];
    }
    __syncthreads();
    PHX(6);
    // ---- tile rows -> the CSR array: 16 B per lane from the 128-byte boundary below each run; all LDS reads first.
    // Chunk e (doubles 2 tid + 256 it above that boundary) is the same lane offset for every row: one address register ----
    double2 v[PH_TY][2];
    const unsigned e0 = 2u * (unsigned)tid;
#pragma unroll
    for (int rr = 0; rr < PH_TY; rr++) {
      const unsigned head = (oLi + rr * sji) & 15u;
      const double *src = stg + rr * PH_RS - (int)(head & ~1u);  // (uniform; chunk e of the run is src[e], 16-byte aligned)
#pragma unroll
      for (int it = 0; it < 2; it++) v[rr][it] = *reinterpret_cast<const double2 *>(src + e0 + 2u * PH_NT * it);
    }
#pragma unroll
    for (int rr = 0; rr < PH_TY; rr++) {
      if (rr < nvy) {
        const unsigned head = (oLi + rr * sji) & 15u;
        double *bal = oL + rr * sj - head;  // the 128-byte boundary below the run (uniform)
#pragma unroll
        for (int it = 0; it < 2; it++) {
          const unsigned e = e0 + 2u * PH_NT * it;
          if (e >= head && e + 1 < head + n) {
#if PH_EXP == 1
            asm volatile("" ::"v"(v[rr][it].x), "v"(v[rr][it].y));
#else
            store_pair(bal + e, v[rr][it]);
#endif
          }
        }
      }
    }
    if (tid < 16) {  // the odd double at either end of a run: lane r the head of row r, lane 8 + r its tail
      const int rr = tid & 7, tail = tid >> 3;
      const int head = (int)((oLi + rr * sji) & 15);
      const bool odd = tail ? ((head + n) & 1) : (head & 1);
      if (rr < nvy && odd) {
        const int t = tail ? n - 1 : 0;
        (oL + rr * sj)[t] = stg[rr * PH_RS + (head & 1) + t];
      }
    }
    oL += sk;
  }
#if PH_EXP == 9
  PHX(7);
  if ((tid & 63) == 0) {
    for (int q = 0; q < 8; q++) atomicAdd(a.bad + 8 + q, phc[q]);
    atomicAdd(a.bad + 16, (unsigned long long)nsteps);
  }
#endif
}

// ---------------------------------------------------------------------------
// k_assemble_p3_hex4: the same arithmetic with the element phase and the rows in DIFFERENT waves.  k_assemble_p3_hex
// needs ~330 registers (the element phase ~190 of them, the carry, the rows, the copy-out) and so holds one wave per
// SIMD: nothing overlaps a wave's element phase, the SIMDs are half idle.  Here four waves work on a 16 x 8 element tile:
//   two element waves  lane = element: coordinates (requested a step ahead), element matrices of layer L -> sh_ke
//   two row waves      lane = node: gather from sh_ke (the carry stays in THEIR registers), rows -> their own stage ->
//                      CSR array; a row wave owns whole tile rows (0-3, 4-6), so staging and copy-out need no
//                      workgroup barrier
// and two barriers per step: X (sh_ke complete) and Y (the gather is through: sh_ke may be overwritten).  The row waves'
// staging and stores of layer L run beside the element waves' layer L + 1.  Neither role needs more than 256 registers
// (207): two waves per SIMD.  A workgroup is EIGHT waves on two x-neighbouring tiles (145 KB of LDS, one per CU), so that
// which SIMD holds what is fixed: see the wave roles below (they decide speed, not results).
// ---------------------------------------------------------------------------
constexpr int PH4_STG = 16 + PH_TY * PH_RS + 512;  // the stage: 16 doubles of head-room, 7 rows, read slack of the copy-out

__global__ __launch_bounds__(512, 2) void k_assemble_p3_hex4(const Slab s, const KArgs a) {
  if (a.cond && a.bad[1] == 0) return;
  constexpr int NXL = PH_HX * PH_HY, NXD = 3 * NXL;
  constexpr int NLD = (NXD + 127) / 128;
  __shared__ __attribute__((aligned(16))) double sh_ke2[2][128 * KE_KS];
  __shared__ __attribute__((aligned(16))) double sh_stg2[2][PH4_STG];
  __shared__ double sh_x2[2][2 * NXD];
  const int w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // two tiles per workgroup, x-neighbours: waves 0-3 the first, 4-7 the second.  The hardware places the waves of a
  // workgroup on the SIMDs s, s+2, s+1, s+3, s, s+2, s+1, s+3 (measured, PH_EXP = 8): with the element waves 0, 1 of the
  // first tile and 6, 7 of the second every SIMD holds one element wave and one row wave
  const int half = w8 >> 2, w = w8 & 3;
  const bool element_wave = half ? (w >= 2) : (w < 2);
#if PH_EXP == 8  // where the waves sit: histogram of (wave of the workgroup, SIMD) and of the element waves per SIMD
  if (lane == 0) {
    const unsigned simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);  // HW_REG_HW_ID, bits 5:4
    if (blockIdx.x == 0 && blockIdx.y < 4 && blockIdx.z == 0) atomicAdd(a.bad + 8 + w8, (unsigned long long)(simd + 1) << (8 * blockIdx.y));
    if (element_wave) atomicAdd(a.bad + 24 + simd, 1ULL);
  }
#endif
  double *sh_ke = sh_ke2[half], *sh_stg = sh_stg2[half], *sh_x = sh_x2[half];
  const int tid = ((w & 1) << 6) | lane;                  // 0 .. 127 within the role
  const int lx = tid & 15, ly = tid >> 4;
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t i0 = a.x0 + ((int64_t)blockIdx.x * 2 + half) * PH_TX, j0 = a.y0 + (int64_t)blockIdx.y * PH_TY;
  const int64_t zc0 = a.s0 + (int64_t)blockIdx.z * a.zc;
  const int64_t zc1 = (zc0 + a.zc < a.s1) ? zc0 + a.zc : a.s1;
  const int nsteps = (int)(zc1 - zc0) + 1;
  // (an odd number of tiles in x: the last workgroup's second tile is empty -- nvx = 0, its waves keep the barriers)
  const int nvx = (a.x1 - i0 < PH_TX) ? (a.x1 > i0 ? (int)(a.x1 - i0) : 0) : PH_TX;
  const int nvy = (a.y1 - j0 < PH_TY) ? (int)(a.y1 - j0) : PH_TY;

  if (element_wave) {
    const int64_t gex = i0 - 1 + lx, gey = j0 - 1 + ly;
    const bool exy_ok = gex >= 0 && gex < M - 1 && gey >= 0 && gey < N - 1;
    unsigned xoff[NLD];  // (the launch checks 3 M N < 2^31)
#pragma unroll
    for (int q = 0; q < NLD; q++) {
      int d = tid + 128 * q;
      d = d < NXD ? d : NXD - 1;
      const int cy = d / (3 * PH_HX), col = d - 3 * PH_HX * cy, cx = col / 3, comp = col - 3 * cx;
      int64_t gi = i0 - 1 + cx, gj = j0 - 1 + cy;
      gi = gi < 0 ? 0 : (gi > M - 1 ? M - 1 : gi);
      gj = gj < 0 ? 0 : (gj > N - 1 ? N - 1 : gj);
      xoff[q] = (unsigned)(3 * (gi + M * gj) + comp);
    }
    double xn[NLD];
    auto request_layer = [&](int64_t gk) {
      gk = gk < s.cks ? s.cks : (gk > a.cke - 1 ? a.cke - 1 : gk);
      const double *base = a.coords + 3 * M * N * (gk - s.cks);
#pragma unroll
      for (int q = 0; q < NLD; q++) xn[q] = base[xoff[q]];
    };
    auto commit_layer = [&](int slot) {
#pragma unroll
      for (int q = 0; q < NLD; q++)
        if (q < NLD - 1 || tid < NXD - 128 * (NLD - 1)) sh_x[slot * NXD + tid + 128 * q] = xn[q];
    };
    request_layer(zc0 - 1);
    commit_layer(0);
    request_layer(zc0);
    commit_layer(1);
    for (int st = 0; st < nsteps; st++) {
      const int64_t L = zc0 - 1 + st;
      if (st + 1 < nsteps) request_layer(L + 2);
      // Y of the previous step (at st = 0: the barrier that makes the two staged layers visible to both element waves)
      __syncthreads();
      const bool eok = exy_ok && L >= s.els && L < s.ele;
      const double *xlo = sh_x + (st & 1) * NXD + 3 * (lx + PH_HX * ly);
      const double *xhi = sh_x + ((st + 1) & 1) * NXD + 3 * (lx + PH_HX * ly);
      const int nbad = ph_element_matrix(xlo, xhi, eok ? 1.0 / 512.0 : 0.0, sh_ke + tid * KE_KS);
      if (eok && nbad) atomicAdd(a.bad, (unsigned long long)nbad);
      // node layer L + 2 over node layer L.  Both element waves must be through layer L first: they are not until X, so
      // the commit waits for it (the row waves gather meanwhile)
      __syncthreads();  // X: sh_ke complete
      if (st + 1 < nsteps) commit_layer(st & 1);
    }
    __syncthreads();  // Y of the last step
  } else {
    const int n = nvx * 27;
    const bool mynode = lx < nvx && ly < nvy;
    const int glx = (lx < PH_TX) ? lx : 0, gly = (ly < PH_TY) ? ly : 0;
    double lower[9], mid[9];
#pragma unroll
    for (int q = 0; q < 9; q++) lower[q] = mid[q] = 0.0;
    double *const o00 = a.out + (row_start(s, i0, j0, zc0, 0) - a.base_off);
    const int64_t sj = row_start(s, i0, j0 + 1, zc0, 0) - row_start(s, i0, j0, zc0, 0);
    const int64_t sk = row_start(s, i0, j0, zc0 + 1, 0) - row_start(s, i0, j0, zc0, 0);
    const unsigned sji = __builtin_amdgcn_readfirstlane((unsigned)sj);
    double *oL = o00;
    double *stg = sh_stg + 16;
    const int r0 = 4 * (w & 1);  // this wave's tile rows: r0 .. r0 + 3 (row 7 does not exist)
    auto wave_sync = [&]() {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    };
    __syncthreads();  // (the element waves' first barrier)
    for (int st = 0; st < nsteps; st++) {
      const int64_t L = zc0 - 1 + st;
      __syncthreads();  // X: the element matrices of layer L
      double row[27];
      const bool finish = st >= 1, start = st < nsteps - 1;
      if (finish) {
#pragma unroll
        for (int q = 0; q < 9; q++) {
          row[q] = lower[q];
          row[9 + q] = mid[q];
        }
        ke_gather_p3<0, 0>(sh_ke, glx, gly, row + 9);
        ke_gather_p3<0, 1, true>(sh_ke, glx, gly, row + 18);
      }
      if (start) {
        ke_gather_p3<1, 0, true>(sh_ke, glx, gly, lower);
        ke_gather_p3<1, 1, true>(sh_ke, glx, gly, mid);
      }
      __syncthreads();  // Y: the gather is through (the element waves' next step, or their final barrier)
      if (!finish) continue;
      const unsigned oLi = __builtin_amdgcn_readfirstlane((unsigned)(reinterpret_cast<uintptr_t>(oL) >> 3));
      wave_sync();  // this wave's copy-out of the previous step has read its rows of the stage
      if (mynode) {
        double *dst = stg + ly * PH_RS + ((oLi + ly * sji) & 1) + lx * 27;
#pragma unroll
        for (int q = 0; q < 27; q++) dst[q] = row[q];
      }
      wave_sync();
      double2 v[4][4];
      const unsigned e0 = 2u * (unsigned)lane;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int rr = r0 + u;
        const unsigned head = (oLi + rr * sji) & 15u;
        const double *src = stg + rr * PH_RS - (int)(head & ~1u);
#pragma unroll
        for (int it = 0; it < 4; it++) v[u][it] = *reinterpret_cast<const double2 *>(src + e0 + 128u * it);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int rr = r0 + u;
        if (rr < nvy) {
          const unsigned head = (oLi + rr * sji) & 15u;
          double *bal = oL + rr * sj - head;
#pragma unroll
          for (int it = 0; it < 4; it++) {
            const unsigned e = e0 + 128u * it;
            if (e >= head && e + 1 < head + n) {
#if PH_EXP == 1
              asm volatile("" ::"v"(v[u][it].x), "v"(v[u][it].y));
#else
              store_pair(bal + e, v[u][it]);
#endif
            }
          }
        }
      }
      if (lane < 8) {  // the odd double at either end of a run: lane u the head of this wave's row u, lane 4 + u its tail
        const int rr = r0 + (lane & 3), tail = lane >> 2;
        const int head = (int)((oLi + rr * sji) & 15);
        const bool odd = tail ? ((head + n) & 1) : (head & 1);
        if (rr < nvy && odd && n > 0) {
          const int t = tail ? n - 1 : 0;
          (oL + rr * sj)[t] = stg[rr * PH_RS + (head & 1) + t];
        }
      }
      oL += sk;
    }
  }
}

}  // namespace femasm
