// femasm_sweep.h -- the 3-D interior K kernel: a z-sweep over element layers.
//
// Why a second K kernel: in the 4x4x4 brick kernel (femasm_kernels.h) the lanes of a wave sit in
// four z-layers, and no linearisation of the halo'd 5x5x5 element brick keeps the 32 (or 16) lanes
// of one LDS access on distinct banks: rocprof shows 43 % of its LDS cycles are bank conflicts and the
// LDS busy for half the kernel time.  Here every lane of a wave works in the SAME element layer:
//
//  * workgroup = 16x4 node tile, 4 waves, sweeping ZC node layers along z.  One step = one element
//    layer L (17x5 elements with the halo): its 8 Gauss-point records are built once and consumed by
//    all four waves, so the quadrature redundancy is 85/64 (was 125/64).
//  * wave roles per step (wave-uniform):  A  = lower third (dz=-1) of node layer L+1,
//                                         D  = upper third (dz=+1) of node layer L,
//                                         M1/M2 alternate: "start" the middle third of layer L+1 from the
//    element layer below it, and one step later "finish" it from the layer above, in the same registers --
//    no cross-wave sum at all.
//  * records are pair-SoA: rec[pair][slot] is a 16-byte (2-double) cell, slot = ex + 24*ey.  With the lane
//    map (lane>>4) -> y row {0,2,1,3}, every ds_read_b128 of a wave hits 16 distinct 16-byte bank groups
//    (24*2 = 48 = 0 mod 16, 24 = 8 mod 16): conflict-free at the full 256 B/clk LDS rate.
//  * each third of a node's rows (27*dof contiguous doubles per row) is transposed through a per-wave LDS
//    area and written as contiguous runs, 64 lanes x 8 bytes per store instruction.
//  * only nodes with the full 27-neighbour stencil are handled here (no existence tests on the hot path);
//    Dirichlet rows are written by k_bc_rows, other boundary rows by the brick kernel.
#pragma once
#include "femasm_kernels.h"

namespace femasm {

constexpr int SW_TX = 16, SW_TY = 4, SW_SY = 24, SW_NS = 120, SW_ZC = 16;

// one quadrature record in registers: physical gradients of the 8 nodes + scaled coefficients
template <int KIND>
__device__ inline void sweep_record(const double *xc, int p, double E, double nu, double *r /*[28]*/, bool &bad) {
  const double(*G)[8] = c_tab3.G[p];
  const Jac3 jj = jac3(G, xc);
#pragma unroll
  for (int n = 0; n < 8; n++)
#pragma unroll
    for (int k = 0; k < 3; k++) r[3 * n + k] = G[0][n] * jj.iJ[k][0] + G[1][n] * jj.iJ[k][1] + G[2][n] * jj.iJ[k][2];
  const double wJ = 1.0 * jj.det;
  bad = !(wJ > 0.0);
  if constexpr (KIND == KIND_POISSON) {
    r[24] = wJ; r[25] = 0.0; r[26] = 0.0; r[27] = 0.0;
  } else {
    scaled_coefs<KIND>(E, nu, wJ, r[24], r[25], r[26]);
    r[27] = 0.0;
  }
}

template <int KIND, int AS, int OBS>
__device__ inline void sweep_contract(const double2 *__restrict__ recbuf, int myslot, double *acc) {
  constexpr int DOF = Traits<KIND>::DOF, DD = DOF * DOF;
#pragma unroll
  for (int oy = -1; oy <= 0; oy++)
#pragma unroll
    for (int ox = -1; ox <= 0; ox++) {
      const double2 *rp = recbuf + (myslot + ox + SW_SY * oy);
      double gbf[12];
#pragma unroll
      for (int q = 0; q < 6; q++) {
        const double2 v = rp[(6 * OBS + q) * SW_NS];
        gbf[2 * q] = v.x;
        gbf[2 * q + 1] = v.y;
      }
      const int ia = quad_idx(-ox, -oy);  // compile time after unrolling
      double ga[3];
      if constexpr (AS == OBS) {
        ga[0] = gbf[3 * ia]; ga[1] = gbf[3 * ia + 1]; ga[2] = gbf[3 * ia + 2];
      } else {
        const int q0 = (3 * ia) / 2;
        const double2 u = rp[(6 * AS + q0) * SW_NS], v = rp[(6 * AS + q0 + 1) * SW_NS];
        if ((3 * ia) % 2 == 0) { ga[0] = u.x; ga[1] = u.y; ga[2] = v.x; }
        else                   { ga[0] = u.y; ga[1] = v.x; ga[2] = v.y; }
      }
      double l[3], m[3], q[3];
      if constexpr (KIND == KIND_POISSON) {
        const double wJ = rp[12 * SW_NS].x;
#pragma unroll
        for (int k = 0; k < 3; k++) l[k] = ga[k] * wJ;
      } else {
        const double2 c0 = rp[12 * SW_NS], c1 = rp[13 * SW_NS];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          q[k] = c0.x * ga[k];
          l[k] = c0.y * ga[k];
          m[k] = c1.x * ga[k];
        }
      }
#pragma unroll
      for (int by = 0; by <= 1; by++)
#pragma unroll
        for (int bx = 0; bx <= 1; bx++) {
          const int ib = quad_idx(bx, by);
          const double gb[3] = {gbf[3 * ib], gbf[3 * ib + 1], gbf[3 * ib + 2]};
          add_block<KIND, DOF>(acc + ((oy + by + 1) * 3 + (ox + bx + 1)) * DD, l, m, q, gb);
        }
    }
}

// transposed, coalesced store of one third (dz) of the rows of the wave's 64 nodes (node layer kn):
// registers -> per-wave LDS area -> contiguous runs in HBM
template <int KIND>
__device__ inline void sweep_store_third(const Slab &s, const KArgs &a, const double *acc, double *stage, int lane,
                                         int64_t i0, int64_t j0, int64_t kn, int dz, int nvx) {
  constexpr int DOF = Traits<KIND>::DOF, DD = DOF * DOF, CH = 9 * DOF, NODESTRIDE = 27 * DD;
  constexpr int NIT = (SW_TX * CH + 63) / 64;
  int goff[NIT];  // position of staged value idx in the node-major CSR run of a tile row
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int idx = lane + 64 * it;
    goff[it] = idx + (NODESTRIDE - CH) * (idx / CH);
  }
#if FEMASM_EXP == 2
  {  // experiment: direct lane-wise stores, no LDS transposition
    const int lx = lane & 15, r = lane >> 4, ly = ((r & 1) << 1) | (r >> 1);
    if (lx < nvx && j0 + ly < a.y1) {
      double *base = a.out + (row_start(s, i0 + lx, j0 + ly, kn, 0) - a.base_off) + (dz + 1) * CH;
#pragma unroll
      for (int c = 0; c < DOF; c++)
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
          for (int cc = 0; cc < DOF; cc++) base[c * (27 * DOF) + t * DOF + cc] = acc[t * DD + c * DOF + cc];
    }
    return;
  }
#endif
#pragma unroll
  for (int c = 0; c < DOF; c++) {
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
      for (int cc = 0; cc < DOF; cc++) stage[lane * CH + t * DOF + cc] = acc[t * DD + c * DOF + cc];
    __builtin_amdgcn_wave_barrier();
    for (int rr = 0; rr < 4; rr++) {
      const int64_t jr = j0 + (((rr & 1) << 1) | (rr >> 1));
      if (jr >= a.y1) continue;
      double *base = a.out + (row_start(s, i0, jr, kn, 0) - a.base_off) + (int64_t)c * (27 * DOF) + (dz + 1) * CH;
      const double *src = stage + rr * (SW_TX * CH);
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int idx = lane + 64 * it;
#if FEMASM_EXP == 1
        if (idx < nvx * CH) asm volatile("" :: "v"(src[idx]));
#else
        if (idx < nvx * CH) base[goff[it]] = src[idx];
#endif
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------
// general path: Jacobian, inverse and det per Gauss point (any hexahedron, per-Gauss-point material)
// ---------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256, 2) void k_assemble_sweep(const Slab s, const KArgs a) {
  using T = Traits<KIND>;
  if (a.cond && a.bad[1] == 0) return;
  constexpr int DOF = T::DOF, DD = DOF * DOF, NACC = 9 * DD;
  constexpr int NPAIR = 14;
  constexpr int CH = 9 * DOF;              // doubles of one (node,row) chunk of a third
  static_assert(3 * 64 * CH * 8 <= 2 * NPAIR * SW_NS * 16, "staging aliases the record buffers");

  __shared__ double2 sh_rec[2 * NPAIR * SW_NS];   // 53 760 B
  __shared__ double sh_x[2 * 108 * 3];            //  5 184 B: node layers L, L+1 of the 18x6 halo'd tile

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lx = lane & 15, r = lane >> 4, ly = ((r & 1) << 1) | (r >> 1);
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t i0 = a.x0 + (int64_t)blockIdx.x * SW_TX, j0 = a.y0 + (int64_t)blockIdx.y * SW_TY;
  const int64_t zc0 = a.s0 + (int64_t)blockIdx.z * SW_ZC;
  const int64_t zc1 = (zc0 + SW_ZC < a.s1) ? zc0 + SW_ZC : a.s1;
  const int nsteps = (int)(zc1 - zc0) + 1;
  const int myslot = (lx + 1) + SW_SY * (ly + 1);

  // geometry thread g = tid < 120: element (ex, ey) of the halo'd 17x5 tile, record slot g
  const int ex = tid % SW_SY, ey = tid / SW_SY;
  const int64_t gex = i0 - 1 + ex, gey = j0 - 1 + ey;
  const bool gthread = tid < SW_NS;
  const bool exy_ok = gthread && ex < SW_TX + 1 && gex >= 0 && gex < M - 1 && gey >= 0 && gey < N - 1;

  const int nvx = (a.x1 - i0 < SW_TX) ? (int)(a.x1 - i0) : SW_TX;  // valid nodes of the tile in x
  // three waves write per step: A (wave 0), D (wave 3) and whichever of waves 1/2 finishes a middle third
  double *stage = reinterpret_cast<double *>(sh_rec) + ((w == 0) ? 0 : (w == 3 ? 1 : 2)) * (64 * CH);

  double acc[NACC];

  for (int st = 0; st < nsteps; st++) {
    const int64_t L = zc0 - 1 + st;  // element layer of this step
    // ---- node coordinates of layers L, L+1 (clamped reads; masked elements never use them) ----
    for (int q = tid; q < 216; q += 256) {
      const int cz = q / 108, rem = q - 108 * cz, cy = rem / 18, cx = rem - 18 * cy;
      int64_t gi = i0 - 1 + cx, gj = j0 - 1 + cy, gk = L + cz;
      gi = gi < 0 ? 0 : (gi > M - 1 ? M - 1 : gi);
      gj = gj < 0 ? 0 : (gj > N - 1 ? N - 1 : gj);
      gk = gk < s.cks ? s.cks : (gk > a.cke - 1 ? a.cke - 1 : gk);
      const int64_t node = gi + M * (gj + N * (gk - s.cks));
#pragma unroll
      for (int d = 0; d < 3; d++) sh_x[3 * q + d] = a.coords[3 * node + d];
    }
    const bool eok = exy_ok && L >= s.els && L < s.ele;
    const double *matq = nullptr;  // the element's record (AoS) or its entry in the E plane (FEMASM_MAT_ELEMENT_SOA)
    if (eok && KIND != KIND_POISSON && (a.mat_layout == 0 || a.mat_layout == 3))
      matq = a.gp + (gex + (M - 1) * (gey + (N - 1) * (L - s.mat_e_ks))) * (a.mat_layout == 0 ? a.mat_stride : 1);
    __syncthreads();  // coordinates staged; the previous step's staged output has been read

    // One record, streamed: coordinates are read from LDS inside the J loop and every gradient pair is
    // stored as soon as it exists, so no 24-double coordinate or 28-double record array stays live next
    // to the 81 accumulators.
    auto geometry = [&](int p, double2 *recbuf) {
      if (!gthread) return;
      double2 *dst = recbuf + tid;
      if (!eok) {
#pragma unroll
        for (int q = 0; q < NPAIR; q++) dst[q * SW_NS] = make_double2(0.0, 0.0);
        return;
      }
      const double(*G)[8] = c_tab3.G[p];
      double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
      for (int n = 0; n < 8; n++) {
        const double *xn = sh_x + 3 * ((ex + HDI[n]) + 18 * ((ey + HDJ[n]) + 6 * HDK[n]));
        const double x0 = xn[0], x1 = xn[1], x2 = xn[2];
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
          J[rr][0] = J[rr][0] + G[rr][n] * x0;
          J[rr][1] = J[rr][1] + G[rr][n] * x1;
          J[rr][2] = J[rr][2] + G[rr][n] * x2;
        }
      }
      double iJ[3][3], det;
      inv3_ref(J, iJ, det);
      const double wJ = 1.0 * det;
      if (!(wJ > 0.0)) atomicAdd(a.bad, 1ULL);
      // gradient component c = 3*n + k of node n, direction k
      auto grad = [&](int c) {
        const int n = c / 3, k = c % 3;
        return G[0][n] * iJ[k][0] + G[1][n] * iJ[k][1] + G[2][n] * iJ[k][2];
      };
#pragma unroll
      for (int q = 0; q < 12; q++) dst[q * SW_NS] = make_double2(grad(2 * q), grad(2 * q + 1));
      if constexpr (KIND == KIND_POISSON) {
        dst[12 * SW_NS] = make_double2(wJ, 0.0);
        dst[13 * SW_NS] = make_double2(0.0, 0.0);
      } else {
        double E = a.E, nu = a.nu;
        if (matq) {
          const bool rec = a.mat_layout == 0;
          E = matq[rec ? 96 + p : 0];
          nu = matq[rec ? 104 + p : a.mat_stride];
        }
        double d11, d12, d44;
        scaled_coefs<KIND>(E, nu, wJ, d11, d12, d44);
        dst[12 * SW_NS] = make_double2(d11, d12);
        dst[13 * SW_NS] = make_double2(d44, 0.0);
      }
    };

    geometry(0, sh_rec);
    __syncthreads();

    // ---- wave roles of this step (wave-uniform) ----
    //   role 0 = A (node layer L+1, element below, lower face)   role 1 = start (L+1, below, upper face)
    //   role 2 = finish (L, above, lower face)                    role 3 = D (L, above, upper face)
    int role;
    if (w == 0) role = 0;
    else if (w == 3) role = 3;
    else role = ((st + w) & 1) ? 1 : 2;
    const int64_t kn = (role < 2) ? L + 1 : L;         // node layer served
    const bool active = (role < 2) ? (kn < zc1) : (kn >= zc0);
    if (role != 2) {
#pragma unroll
      for (int q = 0; q < NACC; q++) acc[q] = 0.0;
    }

    for (int p = 0; p < 8; p++) {
#if FEMASM_EXP != 3
      if (p + 1 < 8) geometry(p + 1, sh_rec + ((p + 1) & 1) * (NPAIR * SW_NS));
#endif
      const double2 *rb = sh_rec + (p & 1) * (NPAIR * SW_NS);
#if FEMASM_EXP == 2
      if (false) {
#else
      if (active) {
#endif
        if (role == 0)      sweep_contract<KIND, 1, 0>(rb, myslot, acc);
        else if (role == 1) sweep_contract<KIND, 1, 1>(rb, myslot, acc);
        else if (role == 2) sweep_contract<KIND, 0, 0>(rb, myslot, acc);
        else                sweep_contract<KIND, 0, 1>(rb, myslot, acc);
      }
      __syncthreads();
    }

    // ---- output: roles A (dz=-1), finish (dz=0), D (dz=+1) write one third of their node layer ----
    if (active && role != 1)
      sweep_store_third<KIND>(s, a, acc, stage, lane, i0, j0, kn, (role == 0) ? -1 : (role == 2 ? 0 : 1), nvx);
    // the next step's first __syncthreads orders these LDS reads before the records are rebuilt
  }
}

// ---------------------------------------------------------------------------
// General path for P3 (any hexahedra).  With g_a = iJ^T ghat_a the Gauss-point term w detJ g_a.g_b is
// ghat_a^T C_p ghat_b, C_p = detJ_p iJ_p^T iJ_p: the reference gradients ghat are compile-time numbers, so a
// record is the six entries of C_p (48 B instead of the 200 B of eight physical gradients) and the products
// ghat_a[k] ghat_b[l] fold into literals once the Gauss-point loop is unrolled.  Same tile, roles and stores as
// k_assemble_sweep; all 8 x 85 records of an element layer are built in one go by all four waves (wave w: Gauss
// points 2w, 2w+1), so a step has three barriers instead of nine.  LDS reads per node 12.8 KB -> 3 KB.
// ---------------------------------------------------------------------------
struct RefGP { double c[8][8][8][6]; };  // [p][a][b]: coefficients of C00, C11, C22, C01, C02, C12
constexpr RefGP make_refGP() {
  RefGP o{};
  const Tab3 t = make_tab3();
  for (int p = 0; p < 8; p++)
    for (int a = 0; a < 8; a++)
      for (int b = 0; b < 8; b++) {
        const double ga[3] = {t.G[p][0][a], t.G[p][1][a], t.G[p][2][a]};
        const double gb[3] = {t.G[p][0][b], t.G[p][1][b], t.G[p][2][b]};
        o.c[p][a][b][0] = ga[0] * gb[0];
        o.c[p][a][b][1] = ga[1] * gb[1];
        o.c[p][a][b][2] = ga[2] * gb[2];
        o.c[p][a][b][3] = ga[0] * gb[1] + ga[1] * gb[0];
        o.c[p][a][b][4] = ga[0] * gb[2] + ga[2] * gb[0];
        o.c[p][a][b][5] = ga[1] * gb[2] + ga[2] * gb[1];
      }
  return o;
}
__device__ const RefGP d_refGP = make_refGP();

template <int AS, int OBS>
__device__ inline void sweep_contract_p3(const double2 *__restrict__ rec, int myslot, double *acc) {
#pragma unroll
  for (int oy = -1; oy <= 0; oy++)
#pragma unroll
    for (int ox = -1; ox <= 0; ox++) {
      const double2 *rp = rec + (myslot + ox + SW_SY * oy);
      const int ia = hex_idx(-ox, -oy, AS);
#pragma unroll
      for (int p = 0; p < 8; p++) {
        const double2 c0 = rp[(3 * p) * SW_NS], c1 = rp[(3 * p + 1) * SW_NS], c2 = rp[(3 * p + 2) * SW_NS];
#pragma unroll
        for (int by = 0; by <= 1; by++)
#pragma unroll
          for (int bx = 0; bx <= 1; bx++) {
            const double *k = d_refGP.c[p][ia][hex_idx(bx, by, OBS)];
            const int t = (oy + by + 1) * 3 + (ox + bx + 1);
            double v = acc[t];
            v = fma(k[0], c0.x, v);
            v = fma(k[1], c0.y, v);
            v = fma(k[2], c1.x, v);
            v = fma(k[3], c1.y, v);
            v = fma(k[4], c2.x, v);
            v = fma(k[5], c2.y, v);
            acc[t] = v;
          }
      }
    }
}

__global__ __launch_bounds__(256, 2) void k_assemble_sweep_p3(const Slab s, const KArgs a) {
  if (a.cond && a.bad[1] == 0) return;
  constexpr int KIND = KIND_POISSON, NACC = 9, NPAIR = 24, CH = 9;
  static_assert(3 * 64 * CH * 8 <= NPAIR * SW_NS * 16, "staging aliases the record buffer");
  __shared__ double2 sh_rec[NPAIR * SW_NS];  // 46 080 B: C_p of the 8 Gauss points, pair-SoA rec[3p + q][slot]
  __shared__ double sh_x[2 * 108 * 3];       // node layers L, L+1 of the 18x6 halo'd tile

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lx = lane & 15, r = lane >> 4, ly = ((r & 1) << 1) | (r >> 1);
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t i0 = a.x0 + (int64_t)blockIdx.x * SW_TX, j0 = a.y0 + (int64_t)blockIdx.y * SW_TY;
  const int64_t zc0 = a.s0 + (int64_t)blockIdx.z * SW_ZC;
  const int64_t zc1 = (zc0 + SW_ZC < a.s1) ? zc0 + SW_ZC : a.s1;
  const int nsteps = (int)(zc1 - zc0) + 1;
  const int myslot = (lx + 1) + SW_SY * (ly + 1);
  const int nvx = (a.x1 - i0 < SW_TX) ? (int)(a.x1 - i0) : SW_TX;
  double *stage = reinterpret_cast<double *>(sh_rec) + ((w == 0) ? 0 : (w == 3 ? 1 : 2)) * (64 * CH);
  double acc[NACC];

  for (int st = 0; st < nsteps; st++) {
    const int64_t L = zc0 - 1 + st;  // element layer of this step
    for (int q = tid; q < 216; q += 256) {  // node coordinates of layers L, L+1 (clamped reads)
      const int cz = q / 108, rem = q - 108 * cz, cy = rem / 18, cx = rem - 18 * cy;
      int64_t gi = i0 - 1 + cx, gj = j0 - 1 + cy, gk = L + cz;
      gi = gi < 0 ? 0 : (gi > M - 1 ? M - 1 : gi);
      gj = gj < 0 ? 0 : (gj > N - 1 ? N - 1 : gj);
      gk = gk < s.cks ? s.cks : (gk > a.cke - 1 ? a.cke - 1 : gk);
      const int64_t node = gi + M * (gj + N * (gk - s.cks));
#pragma unroll
      for (int d = 0; d < 3; d++) sh_x[3 * q + d] = a.coords[3 * node + d];
    }
    __syncthreads();  // coordinates staged; the previous step's staged output has been read

    // records: wave w builds Gauss points 2w and 2w+1 of every element slot
#pragma unroll 1
    for (int pp = 0; pp < 2; pp++) {
      const int p = 2 * w + pp;
      const double(*G)[8] = c_tab3.G[p];
#pragma unroll 1
      for (int e = lane; e < SW_NS; e += 64) {
        const int ex = e % SW_SY, ey = e / SW_SY;
        const int64_t gex = i0 - 1 + ex, gey = j0 - 1 + ey;
        const bool eok = ex < SW_TX + 1 && gex >= 0 && gex < M - 1 && gey >= 0 && gey < N - 1 && L >= s.els && L < s.ele;
        double2 *dst = sh_rec + (3 * p) * SW_NS + e;
        if (!eok) {
          dst[0] = dst[SW_NS] = dst[2 * SW_NS] = make_double2(0.0, 0.0);
          continue;
        }
        double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
        for (int n = 0; n < 8; n++) {
          const double *xn = sh_x + 3 * ((ex + HDI[n]) + 18 * ((ey + HDJ[n]) + 6 * HDK[n]));
          const double x0 = xn[0], x1 = xn[1], x2 = xn[2];
#pragma unroll
          for (int rr = 0; rr < 3; rr++) {
            J[rr][0] = J[rr][0] + G[rr][n] * x0;
            J[rr][1] = J[rr][1] + G[rr][n] * x1;
            J[rr][2] = J[rr][2] + G[rr][n] * x2;
          }
        }
        double iJ[3][3], det;
        inv3_ref(J, iJ, det);
        const double wJ = 1.0 * det;
        if (!(wJ > 0.0)) atomicAdd(a.bad, 1ULL);
        double Cm[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
          for (int l = k; l < 3; l++) Cm[k][l] = (iJ[0][k] * iJ[0][l] + iJ[1][k] * iJ[1][l] + iJ[2][k] * iJ[2][l]) * wJ;
        dst[0] = make_double2(Cm[0][0], Cm[1][1]);
        dst[SW_NS] = make_double2(Cm[2][2], Cm[0][1]);
        dst[2 * SW_NS] = make_double2(Cm[0][2], Cm[1][2]);
      }
    }
    __syncthreads();

    int role;
    if (w == 0) role = 0;
    else if (w == 3) role = 3;
    else role = ((st + w) & 1) ? 1 : 2;
    const int64_t kn = (role < 2) ? L + 1 : L;
    const bool active = (role < 2) ? (kn < zc1) : (kn >= zc0);
    if (role != 2) {
#pragma unroll
      for (int q = 0; q < NACC; q++) acc[q] = 0.0;
    }
    if (active) {
      if (role == 0)      sweep_contract_p3<1, 0>(sh_rec, myslot, acc);
      else if (role == 1) sweep_contract_p3<1, 1>(sh_rec, myslot, acc);
      else if (role == 2) sweep_contract_p3<0, 0>(sh_rec, myslot, acc);
      else                sweep_contract_p3<0, 1>(sh_rec, myslot, acc);
    }
    __syncthreads();  // every wave has read its records: the staging areas alias them
    if (active && role != 1)
      sweep_store_third<KIND>(s, a, acc, stage, lane, i0, j0, kn, (role == 0) ? -1 : (role == 2 ? 0 : 1), nvx);
  }
}

// ---------------------------------------------------------------------------
// General path for P3, element matrices through LDS: k_assemble_ke_p3.
// k_assemble_sweep_p3 is bound by FP64 issue: every node row forms its 27 entries from the Gauss-point records itself,
// 8 elements x 8 nodes x 8 Gauss points x 6 FMAs = 3072 FMAs per node, and each entry K_ab of an element is formed
// twice (in the row of a and in the row of b).  Here the element matrix is formed ONCE, its 36 unique entries
// (a <= b) only, and the node rows pick their entries out of LDS:
//   tile      15 x 3 interior nodes, i.e. 16 x 4 = 64 elements per element layer: lane = element in the two element
//             phases, lane = node (45 of 64 busy) in the cheap gather
//   records   wave w: C_p = detJ iJ^T iJ of Gauss points 2w, 2w+1 of element `lane` (the arithmetic of
//             k_assemble_sweep_p3), pair-SoA in LDS, slot = lane: conflict-free 16-byte reads
//   Ke        wave w: entries 9w .. 9w+8 of the 36, each sum_p ghat_a^T C_p ghat_b with literal coefficients
//             (48 FMAs), written to sh_ke[element * 37 + entry] (37: conflict-free for the gather's stride)
//   gather    the wave roles of k_assemble_sweep (lower third of layer L+1, start / finish of the middle third, upper
//             third of layer L): 16 LDS reads and adds per node and role, in a fixed order
//   stores    per-wave LDS transposition as in k_assemble_sweep (216-byte runs)
// Per element layer a wave issues 2 x ~140 + 432 FMAs for 45 nodes where k_assemble_sweep_p3 issues 4 x ~140 + 1536
// for 64.  Sums are formed element by element and then over the elements -- the reference's own order (Ke, then
// ADD_VALUES), rounding differs from k_assemble_sweep_p3 in the last bit.
// ---------------------------------------------------------------------------
constexpr int KE_TX = 15, KE_TY = 3, KE_ZC = 16, KE_KS = 37;
struct KePairs { int a[36], b[36], idx[8][8]; };
constexpr KePairs make_ke_pairs() {
  KePairs o{};
  int n = 0;
  for (int a = 0; a < 8; a++)
    for (int b = a; b < 8; b++) {
      o.a[n] = a;
      o.b[n] = b;
      o.idx[a][b] = o.idx[b][a] = n;
      n++;
    }
  return o;
}
__device__ const KePairs KE_PAIRS = make_ke_pairs();  // const + constant initialiser: folded after unrolling

template <int W>
__device__ inline void ke_entries_p3(const double2 *__restrict__ rec /* sh_rec + lane */, double *__restrict__ ke) {
  double v[9];
#pragma unroll
  for (int q = 0; q < 9; q++) v[q] = 0.0;
#pragma unroll
  for (int p = 0; p < 8; p++) {
    const double2 c0 = rec[(3 * p) * 64], c1 = rec[(3 * p + 1) * 64], c2 = rec[(3 * p + 2) * 64];
#pragma unroll
    for (int q = 0; q < 9; q++) {
      const double *k = d_refGP.c[p][KE_PAIRS.a[9 * W + q]][KE_PAIRS.b[9 * W + q]];
      double t = v[q];
      t = fma(k[0], c0.x, t);
      t = fma(k[1], c0.y, t);
      t = fma(k[2], c1.x, t);
      t = fma(k[3], c1.y, t);
      t = fma(k[4], c2.x, t);
      t = fma(k[5], c2.y, t);
      v[q] = t;
    }
  }
#pragma unroll
  for (int q = 0; q < 9; q++) ke[9 * W + q] = v[q];
}
// the entries of the four elements of one element layer that belong to the node's row: AS = 1: the node is on the
// top face of these elements; OBS: the neighbours on their bottom (0) or top (1) face
// FRESH: acc holds nothing yet -- the first contribution of an entry is assigned (the same sums as from zero: 0 + x = x
// for every x but -0, which a stiffness entry's sign does not distinguish), nine additions fewer per third
template <int AS, int OBS, bool FRESH = false>
__device__ inline void ke_gather_p3(const double *__restrict__ ke, int lx, int ly, double *acc) {
  unsigned seen = FRESH ? 0u : 0x1FFu;  // (folds away: the loops are unrolled)
#pragma unroll
  for (int oy = -1; oy <= 0; oy++)
#pragma unroll
    for (int ox = -1; ox <= 0; ox++) {
      const double *kp = ke + ((lx + 1 + ox) + 16 * (ly + 1 + oy)) * KE_KS;
      const int ia = hex_idx(-ox, -oy, AS);
#pragma unroll
      for (int by = 0; by <= 1; by++)
#pragma unroll
        for (int bx = 0; bx <= 1; bx++) {
          const int t = (oy + by + 1) * 3 + (ox + bx + 1);
          const double v = kp[KE_PAIRS.idx[ia][hex_idx(bx, by, OBS)]];
          acc[t] = ((seen >> t) & 1u) ? acc[t] + v : v;
          seen |= 1u << t;
        }
    }
}
// one third (dz) of the rows of the wave's nodes (lane = lx + 16 ly, lx < nvx, ly < nvy) -> per-wave LDS area ->
// 72-byte pieces of the rows, consecutive lanes on consecutive values
__device__ inline void ke_store_third(const Slab &s, const KArgs &a, const double *acc, double *stage, int lane, int64_t i0,
                                      int64_t j0, int64_t kn, int dz, int nvx, int nvy) {
  constexpr int CH = 9, NODESTRIDE = 27;
  constexpr int NIT = (16 * CH + 63) / 64;
  int goff[NIT];  // position of staged value idx in the node-major CSR run of a tile row
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int idx = lane + 64 * it;
    goff[it] = idx + (NODESTRIDE - CH) * (idx / CH);
  }
#pragma unroll
  for (int t = 0; t < 9; t++) stage[lane * CH + t] = acc[t];
  __builtin_amdgcn_wave_barrier();
  for (int rr = 0; rr < nvy; rr++) {
    double *base = a.out + (row_start(s, i0, j0 + rr, kn, 0) - a.base_off) + (dz + 1) * CH;
    const double *src = stage + rr * (16 * CH);
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int idx = lane + 64 * it;
#if FEMASM_EXP == 1
      if (idx < nvx * CH) asm volatile("" ::"v"(src[idx]), "v"(base));
#else
      if (idx < nvx * CH) base[goff[it]] = src[idx];
#endif
    }
  }
  __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256, 3) void k_assemble_ke_p3(const Slab s, const KArgs a) {
  if (a.cond && a.bad[1] == 0) return;
  constexpr int NACC = 9, NPAIR = 24, CH = 9, XS = 17, YS = 5;
  __shared__ double2 sh_rec[NPAIR * 64];          // C_p of the 8 Gauss points, pair-SoA rec[3p + q][element]
  __shared__ double sh_ke[64 * KE_KS];             // the 36 unique entries of the 64 element matrices
  __shared__ double sh_x[2 * XS * YS * 3];         // node layers L, L+1 of the halo'd tile
  static_assert(3 * 64 * CH * 8 <= NPAIR * 64 * 16, "staging aliases the record buffer");
  double *sh_stage = reinterpret_cast<double *>(sh_rec);  // per storing wave; the records are dead once Ke is formed
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lx = lane & 15, ly = lane >> 4;        // node of the tile (lx < 15, ly < 3) / element of the layer
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t i0 = a.x0 + (int64_t)blockIdx.x * KE_TX, j0 = a.y0 + (int64_t)blockIdx.y * KE_TY;
  const int64_t zc0 = a.s0 + (int64_t)blockIdx.z * KE_ZC;
  const int64_t zc1 = (zc0 + KE_ZC < a.s1) ? zc0 + KE_ZC : a.s1;
  const int nsteps = (int)(zc1 - zc0) + 1;
  const int nvx = (a.x1 - i0 < KE_TX) ? (int)(a.x1 - i0) : KE_TX;
  const int nvy = (a.y1 - j0 < KE_TY) ? (int)(a.y1 - j0) : KE_TY;
  const bool mynode = lx < KE_TX && ly < KE_TY;
  const int glx = mynode ? lx : 0, gly = mynode ? ly : 0;  // idle lanes gather node (0, 0) again: no LDS read out of range
  double *stage = sh_stage + ((w == 0) ? 0 : (w == 3 ? 1 : 2)) * (64 * CH);
  double acc[NACC];
#pragma unroll
  for (int q = 0; q < NACC; q++) acc[q] = 0.0;

  for (int st = 0; st < nsteps; st++) {
    const int64_t L = zc0 - 1 + st;  // element layer of this step
    // node coordinates of layers L, L+1 (clamped reads).  (Requesting the new layer a step ahead through registers was
    // measured 3 % slower: the other workgroups of the CU cover the wait, the extra registers and code do not pay.)
    for (int q = tid; q < 2 * XS * YS; q += 256) {
      const int cz = q / (XS * YS), rem = q - (XS * YS) * cz, cy = rem / XS, cx = rem - XS * cy;
      int64_t gi = i0 - 1 + cx, gj = j0 - 1 + cy, gk = L + cz;
      gi = gi < 0 ? 0 : (gi > M - 1 ? M - 1 : gi);
      gj = gj < 0 ? 0 : (gj > N - 1 ? N - 1 : gj);
      gk = gk < s.cks ? s.cks : (gk > a.cke - 1 ? a.cke - 1 : gk);
      const int64_t node = gi + M * (gj + N * (gk - s.cks));
#pragma unroll
      for (int d = 0; d < 3; d++) sh_x[3 * q + d] = a.coords[3 * node + d];
    }
    __syncthreads();  // coordinates staged; every wave is through the previous step's gather and stores

    {  // records: wave w builds Gauss points 2w and 2w+1 of element `lane` = (ex, ey) = (lx, ly)
      const int64_t gex = i0 - 1 + lx, gey = j0 - 1 + ly;
      const bool eok = gex >= 0 && gex < M - 1 && gey >= 0 && gey < N - 1 && L >= s.els && L < s.ele;
#pragma unroll 1
      for (int pp = 0; pp < 2; pp++) {
        const int p = 2 * w + pp;
        const double(*G)[8] = c_tab3.G[p];
        double2 *dst = sh_rec + (3 * p) * 64 + lane;
        double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
        for (int n = 0; n < 8; n++) {
          const double *xn = sh_x + 3 * ((lx + HDI[n]) + XS * ((ly + HDJ[n]) + YS * HDK[n]));
          const double x0 = xn[0], x1 = xn[1], x2 = xn[2];
#pragma unroll
          for (int rr = 0; rr < 3; rr++) {
            J[rr][0] = J[rr][0] + G[rr][n] * x0;
            J[rr][1] = J[rr][1] + G[rr][n] * x1;
            J[rr][2] = J[rr][2] + G[rr][n] * x2;
          }
        }
        double iJ[3][3], det;
        inv3_ref(J, iJ, det);
        const double wJ = 1.0 * det;
        if (eok && !(wJ > 0.0)) atomicAdd(a.bad, 1ULL);
        double Cm[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
          for (int l = k; l < 3; l++) Cm[k][l] = (iJ[0][k] * iJ[0][l] + iJ[1][k] * iJ[1][l] + iJ[2][k] * iJ[2][l]) * wJ;
        dst[0] = eok ? make_double2(Cm[0][0], Cm[1][1]) : make_double2(0.0, 0.0);
        dst[64] = eok ? make_double2(Cm[2][2], Cm[0][1]) : make_double2(0.0, 0.0);
        dst[2 * 64] = eok ? make_double2(Cm[0][2], Cm[1][2]) : make_double2(0.0, 0.0);
      }
    }
    __syncthreads();  // records complete
    if (w == 0)      ke_entries_p3<0>(sh_rec + lane, sh_ke + lane * KE_KS);
    else if (w == 1) ke_entries_p3<1>(sh_rec + lane, sh_ke + lane * KE_KS);
    else if (w == 2) ke_entries_p3<2>(sh_rec + lane, sh_ke + lane * KE_KS);
    else             ke_entries_p3<3>(sh_rec + lane, sh_ke + lane * KE_KS);
    __syncthreads();  // element matrices complete

    int role;
    if (w == 0) role = 0;
    else if (w == 3) role = 3;
    else role = ((st + w) & 1) ? 1 : 2;
    const int64_t kn = (role < 2) ? L + 1 : L;
    const bool active = (role < 2) ? (kn < zc1) : (kn >= zc0);
    if (role != 2) {
#pragma unroll
      for (int q = 0; q < NACC; q++) acc[q] = 0.0;
    }
    if (active) {
      if (role == 0)      ke_gather_p3<1, 0>(sh_ke, glx, gly, acc);
      else if (role == 1) ke_gather_p3<1, 1>(sh_ke, glx, gly, acc);
      else if (role == 2) ke_gather_p3<0, 0>(sh_ke, glx, gly, acc);
      else                ke_gather_p3<0, 1>(sh_ke, glx, gly, acc);
      if (role != 1) ke_store_third(s, a, acc, stage, lane, i0, j0, kn, (role == 0) ? -1 : (role == 2 ? 0 : 1), nvx, nvy);
    }
  }
}

// ---------------------------------------------------------------------------
// Affine fast path.  When an element is a parallelepiped (its twelve edges are three exactly equal
// quadruples -- true for every element of DMDASetUniformCoordinates meshes, P3:663, S3:1198) the Jacobian
// is the same at all Gauss points, so with Gauss-point-constant material
//     sum_p w_p detJ_p g_a(p) g_b(p)^T  =  detJ * iJ R_ab iJ^T ,   R_ab = sum_p dN_a/dxi(p) dN_b/dxi(p)^T ,
// with R_ab a compile-time reference tensor.  The Gauss-point pipeline of k_assemble_sweep collapses to one
// record per element (iJ + 3 scaled coefficients) and the per-pair work from 8 x 21 to 75 FMAs.  Results differ
// from the per-Gauss-point path only by rounding.  The kernel VERIFIES its assumption on every element it
// touches (exact edge equalities, bitwise-equal E/nu over the Gauss points) and raises a flag otherwise; the
// host then reruns the general kernel, so this is an optimisation, never an approximation.
// ---------------------------------------------------------------------------
struct RefR { double R[8][8][3][3]; };
// Closed form of the 2x2x2 Gauss sum: with s_a = (+-1,+-1,+-1) the corner signs of node a and
// f_m = 1 + s_a[m] s_b[m] g^2,  R_ab[k][k] = s_a[k] s_b[k] f_m1 f_m2 / 8  and  R_ab[k][l] = s_a[k] s_b[l] f_m3 / 8.
// Only five magnitudes occur, so the unrolled contraction keeps them in a handful of scalar registers.
constexpr RefR make_refR() {
  RefR o{};
  const double g2 = GAUSS_G * GAUSS_G;
  for (int a = 0; a < 8; a++)
    for (int b = 0; b < 8; b++) {
      const double sa[3] = {HSX[a], HSY[a], HSZ[a]}, sb[3] = {HSX[b], HSY[b], HSZ[b]};
      double f[3] = {0, 0, 0};
      for (int m = 0; m < 3; m++) f[m] = 1.0 + sa[m] * sb[m] * g2;
      for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++) {
          if (k == l) o.R[a][b][k][l] = sa[k] * sb[k] * 0.125 * (f[(k + 1) % 3] * f[(k + 2) % 3]);
          else        o.R[a][b][k][l] = sa[k] * sb[l] * 0.125 * f[3 - k - l];
        }
    }
  return o;
}
constexpr bool refR_matches_gauss_sum() {  // the closed form against the literal sum over Gauss points
  const RefR o = make_refR();
  const Tab3 t = make_tab3();
  for (int a = 0; a < 8; a++)
    for (int b = 0; b < 8; b++)
      for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++) {
          double v = 0.0;
          for (int p = 0; p < 8; p++) v += t.G[p][k][a] * t.G[p][l][b];
          const double d = v - o.R[a][b][k][l];
          if (d > 1e-16 || d < -1e-16) return false;
        }
  return true;
}
static_assert(refR_matches_gauss_sum(), "reference tensor R_ab");
__device__ const RefR d_refR = make_refR();  // const + constant initialiser: folded into literals after unrolling

// ---------------------------------------------------------------------------
// Row-sweep kernel of the fast path.
//
// HBM write rule measured on MI355X (scripts/ubench/store_pattern.hip): a store instruction that leaves a
// 128-byte line partly written -- to be completed by another instruction, even the next one of the same wave --
// costs about four full-line writes; the same 4 GB lands at 5.6 TB/s when every instruction covers whole
// aligned lines and at 2.0-3.0 TB/s when runs of 216 B / 648 B / 1944 B, or a run shifted by 8 B, are written.
// So the values of a tile row (TX x-consecutive interior nodes = one contiguous run of the CSR value array)
// are first gathered in LDS in HBM order and then written as 16 B per lane from the 128-byte boundary below
// the run's start: only the first and last line of a run are partial.
//
// Work split: 4 waves.  Waves 0-2 ("row waves") own the accumulators of one CSR row per lane:
//   elastic  lane = node of the 16x4 tile, wave = dof row c: 81 values = 27 neighbours x 3 column dofs
//   Poisson  lane = node of one 16x4 third of a 48x4 tile, wave = third: 27 values
// and contract the records of the 8 elements around the node (element layers kn-1 and kn, both resident in a
// ring of three record layers), so nothing is carried between steps.  Wave 3 ("record wave") is the only one
// that reads global memory: it stages node coordinates and (E,nu) one step ahead through registers, tests the
// premise of the fast path on every element and writes the records of layer kn+1.  gfx9 counts loads and
// stores in one in-order counter, so a wave that loads waits for its own older stores; here the storing waves
// never load and the loading wave never stores.
//
// Records (per element, pair-SoA rec[pair][slot], slot = ex + SY*ey, 16-byte cells, SY = 8 mod 16 and lane rows
// mapped {0,2,1,3}: conflict-free ds_read_b128):
//   elastic (axis-aligned box, 1/J_ii = a_i):  d11 a_i^2 (3), d44 a_i^2 (3), d12 a_i a_j (3), d44 a_i a_j (3)
//   Poisson (parallelepiped):  C = detJ iJ^T iJ (6)
// ---------------------------------------------------------------------------
template <int KIND>
struct RowSweep {
  static constexpr int DOF = Traits<KIND>::DOF, DD = DOF * DOF;
  static constexpr int NSUB = (DOF == 1) ? 3 : 1;  // 16x4 node sub-tiles per tile (row waves = NSUB * DOF = 3)
  static constexpr int TX = 16 * NSUB, TY = 4;
  static constexpr int SY = (DOF == 1) ? 56 : 24;
  static constexpr int NS = SY * (TY + 1);
  static constexpr int NPA = (DOF == 1) ? 3 : 6;
  static constexpr int NXX = TX + 2, NXL = NXX * (TY + 2);  // halo'd node layer
  static constexpr int NEX = TX + 1, NEL = NEX * (TY + 1);  // elements of a tile layer
  static constexpr int ROW = 27 * DOF;                      // values of an interior CSR row
  static constexpr int NB = ROW * DOF;                      // values of a node block
  static constexpr int RUN = TX * NB;                       // doubles of a tile-row run
  static constexpr int CPASS = (NXL + 63) / 64;             // coordinate passes of the record wave
  static constexpr int EPASS = (NEL + 63) / 64;             // element passes of the record wave
  static constexpr int NRW = 1;  // record waves; 2 for Poisson was measured slower (2.17 vs 1.62 ms): the polling hand-shake costs more than it saves
};
constexpr int RS_ZC = 16;  // default node layers per workgroup (KArgs::zc overrides)


// contributions of the four elements of one element layer to row C of the lane's node.
// AS = 1: the node is on the top face of these elements (layer kn-1), AS = 0: on their bottom face (layer kn).
template <int KIND, int C, int AS>
__device__ inline void row_contract(const double2 *__restrict__ recbuf, int myslot, double *acc) {
  using R_ = RowSweep<KIND>;
  constexpr int DOF = R_::DOF, NS = R_::NS, SY = R_::SY;
#pragma unroll
  for (int oy = -1; oy <= 0; oy++)
#pragma unroll
    for (int ox = -1; ox <= 0; ox++) {
      const double2 *rp = recbuf + (myslot + ox + SY * oy);
      const int ia = hex_idx(-ox, -oy, AS);
      if constexpr (KIND == KIND_POISSON) {
        const double2 c0 = rp[0], c1 = rp[NS], c2 = rp[2 * NS];
        const double C00 = c0.x, C01 = c0.y, C02 = c1.x, C11 = c1.y, C12 = c2.x, C22 = c2.y;
#pragma unroll
        for (int bz = 0; bz <= 1; bz++)
#pragma unroll
          for (int by = 0; by <= 1; by++)
#pragma unroll
            for (int bx = 0; bx <= 1; bx++) {
              const double(*R)[3] = d_refR.R[ia][hex_idx(bx, by, bz)];
              const int t = ((bz - AS + 1) * 3 + (oy + by + 1)) * 3 + (ox + bx + 1);
              double v = acc[t];
              v = fma(R[0][0], C00, v);
              v = fma(R[1][1], C11, v);
              v = fma(R[2][2], C22, v);
              v = fma(R[0][1] + R[1][0], C01, v);
              v = fma(R[0][2] + R[2][0], C02, v);
              v = fma(R[1][2] + R[2][1], C12, v);
              acc[t] = v;
            }
      } else {
        const double2 r0 = rp[0], r1 = rp[NS], r2 = rp[2 * NS], r3 = rp[3 * NS], r4 = rp[4 * NS], r5 = rp[5 * NS];
        const double pd[3] = {r0.x, r0.y, r1.x};  // d11 a_i^2
        const double pe[3] = {r1.y, r2.x, r2.y};  // d44 a_i^2
        const double po[3] = {r3.x, r3.y, r4.x};  // d12 a_i a_j for (i,j) = (0,1), (0,2), (1,2)
        const double pq[3] = {r4.y, r5.x, r5.y};  // d44 a_i a_j
        constexpr int C1 = (C + 1) % 3, C2 = (C + 2) % 3;
#pragma unroll
        for (int bz = 0; bz <= 1; bz++)
#pragma unroll
          for (int by = 0; by <= 1; by++)
#pragma unroll
            for (int bx = 0; bx <= 1; bx++) {
              const double(*R)[3] = d_refR.R[ia][hex_idx(bx, by, bz)];
              const int t = ((bz - AS + 1) * 3 + (oy + by + 1)) * 3 + (ox + bx + 1);
#pragma unroll
              for (int cc = 0; cc < DOF; cc++) {
                double v = acc[t * DOF + cc];
                if (cc == C) {
                  v = fma(R[C][C], pd[C], v);
                  v = fma(R[C1][C1], pe[C1], v);
                  v = fma(R[C2][C2], pe[C2], v);
                } else {
                  v = fma(R[C][cc], po[C + cc - 1], v);
                  v = fma(R[cc][C], pq[C + cc - 1], v);
                }
                acc[t * DOF + cc] = v;
              }
            }
      }
    }
}

// steps of one row wave: wave index wv (0..2), row C (elastic) or third wv of the tile (Poisson)
// Barrier among the three row waves only (the record wave keeps working): an LDS arrival counter that every
// row wave bumps once per call and then polls.  All waves of a workgroup are resident, so the poll always
// makes progress; the guard turns a logic error into a reported failure (FEMASM_ERR_HIP) instead of a hung queue.
template <int NW = 3>
__device__ inline void row_waves_sync(unsigned *cnt, unsigned &epoch, int lane, unsigned long long *bad) {
  epoch += NW;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's LDS traffic has completed
  if (lane == 0) atomicAdd(cnt, 1u);
  int guard = 0;
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < epoch) {
    if (++guard > (1 << 20)) {  // ~0.1 s: cannot happen by design; reported as an internal error (flag 2)
      if (lane == 0) atomicAdd(bad + 2, 1ULL);
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// 16-byte store of a CSR value pair in the run pattern of the level-2 kernels (consecutive lanes, whole 128-byte
// lines), with the non-temporal hint: the values are never read back, and without the hint the kernel's dirty lines
// are still draining from L2 when the read-bound RHS kernels start.  Measured (plain -> hint): P3 256^3 interior kernel
// 0.82 -> 0.76 ms; S3 128^3 interior kernel unchanged, the RHS pass after it 1.04 -> 0.97 ms; E2 2048^2 interior
// kernel 0.34 -> 0.36 ms but the RHS after it 0.19 -> 0.15 ms.  (-DFEMASM_PLAIN_K_STORES: plain stores.  The hint on a
// per-lane row pattern, DESIGN section 3, is 8x slower: it only pays on whole lines.)
__device__ inline void store_pair(double *p, double2 v) {
#ifndef FEMASM_PLAIN_K_STORES
  typedef double d2v __attribute__((ext_vector_type(2)));
  d2v vv;
  vv.x = v.x;
  vv.y = v.y;
  __builtin_nontemporal_store(vv, reinterpret_cast<d2v *>(p));
#else
  *reinterpret_cast<double2 *>(p) = v;
#endif
}

template <int KIND, int C>
__device__ inline void rowsweep_row_wave(const Slab &s, const KArgs &a, const double2 *sh_rec, double *sh_stage,
                                         unsigned *sh_cnt, int wv, int lane, int64_t i0, int64_t j0, int64_t zc0,
                                         int nz) {
  using R_ = RowSweep<KIND>;
  constexpr int DOF = R_::DOF, ROW = R_::ROW, NB = R_::NB, NS = R_::NS, NPA = R_::NPA, SY = R_::SY;
  const int sub = (DOF == 1) ? wv : 0;
  const int nvx = (a.x1 - i0 < R_::TX) ? (int)(a.x1 - i0) : R_::TX;  // valid nodes of the tile in x
  int q3 = 0;                                                         // st % 3
  unsigned epoch = 0;
  PH_DECL;
  for (int st = 0; st < nz; st++) {
    const int64_t kn = zc0 + st;
    int ln = lane;  // opaque per step: lane-derived addressing is recomputed, not hoisted and spilled
    asm volatile("" : "+v"(ln));
    const int lx = ln & 15, r = ln >> 4, ly = ((r & 1) << 1) | (r >> 1);
    const int myslot = (16 * sub + lx + 1) + SY * (ly + 1);
    const int q3a = (q3 == 2) ? 0 : q3 + 1;
    PH_T(0);
    __syncthreads();  // records of element layers kn-1 (ring slot q3) and kn (slot q3a) are complete
    PH_ACC(0, 1);
    double acc[ROW];
#pragma unroll
    for (int q = 0; q < ROW; q++) acc[q] = 0.0;
    row_contract<KIND, C, 1>(sh_rec + q3 * (NPA * NS), myslot, acc);
    row_contract<KIND, C, 0>(sh_rec + q3a * (NPA * NS), myslot, acc);
    const int ct = 64 * wv + ln;  // 0..191 over the row waves
    PH_ACC(1, 2);
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      row_waves_sync(sh_cnt, epoch, ln, a.bad);  // the previous run has left the staging buffer
      PH_ACC(0, 3);
      if (ly == rr) {
        double *dst = sh_stage + (16 * sub + lx) * NB + C * ROW;
#pragma unroll
        for (int q = 0; q < ROW; q++) dst[q] = acc[q];
      }
      PH_ACC(2, 4);
      row_waves_sync(sh_cnt, epoch, ln, a.bad);  // run complete in LDS, in HBM order
      PH_ACC(0, 5);
      const int64_t jr = j0 + rr;
      if (jr < a.y1) {
        const int n = nvx * NB;
        double *runp = a.out + (row_start(s, i0, jr, kn, 0) - a.base_off);
        const int head = (int)((reinterpret_cast<uintptr_t>(runp) & 127) >> 3);  // doubles above a line boundary
        const int nchunk = (head + n + 1) >> 1;
        constexpr int NIT = (R_::RUN / 2 + 8 + 191) / 192;
        double2 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {  // chunk q = the q-th 16 bytes from that boundary; LDS reads first
          const int s0 = 2 * (ct + 192 * it) - head;
          const int sa = s0 < 0 ? 0 : s0, sb = s0 + 1 < n ? s0 + 1 : n - 1;
          v[it] = make_double2(sh_stage[sa], sh_stage[sb]);
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
          const int q = ct + 192 * it, s0 = 2 * q - head;
          if (q < nchunk) {
            if (s0 >= 0 && s0 + 1 < n) {
#if FEMASM_EXP == 1
              asm volatile("" ::"v"(v[it].x), "v"(v[it].y));
#else
              store_pair(runp + s0, v[it]);
#endif
            } else if (s0 == -1) {
              runp[0] = v[it].y;
            } else if (s0 == n - 1) {
              runp[s0] = v[it].x;
            }
          }
        }
      }
      PH_ACC(3, 6);
    }
    q3 = q3a;
  }
  PH_FLUSH(8);
}

template <int KIND>
__global__ __launch_bounds__(64 * (3 + RowSweep<KIND>::NRW), 2) void k_assemble_rowsweep(const Slab s, const KArgs a) {
  using R_ = RowSweep<KIND>;
  constexpr int DOF = R_::DOF, NS = R_::NS, NPA = R_::NPA, SY = R_::SY, NXX = R_::NXX, NXL = R_::NXL;
  constexpr int NEX = R_::NEX, NEL = R_::NEL, CPASS = R_::CPASS, EPASS = R_::EPASS, NRW = R_::NRW;
  constexpr bool HASMAT = (KIND != KIND_POISSON);

  __shared__ double2 sh_rec[3 * NPA * NS];  // ring: element layer L0+q in slot q % 3
  __shared__ double sh_stage[R_::RUN];      // one tile-row run in HBM order
  __shared__ double sh_x[3][NXL * 3];       // ring of node layers (record wave only): layer L0+q in slot q % 3
  __shared__ unsigned sh_cnt;               // arrival counter of row_waves_sync
  __shared__ unsigned sh_cnt2;              // the same among the record waves

  const int tid = threadIdx.x, lane = tid & 63;
  const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t i0 = a.x0 + (int64_t)blockIdx.x * R_::TX, j0 = a.y0 + (int64_t)blockIdx.y * R_::TY;
  const int64_t zc0 = a.s0 + (int64_t)blockIdx.z * a.zc;
  const int64_t zc1 = (zc0 + a.zc < a.s1) ? zc0 + a.zc : a.s1;
  const int nz = (int)(zc1 - zc0);
  const int64_t L0 = zc0 - 1;

  if (tid == 0) sh_cnt = sh_cnt2 = 0;
  if (wu < 3) {
    __syncthreads();  // counters initialised
    __syncthreads();  // matches the record waves' prologue
    if constexpr (DOF == 1) {
      rowsweep_row_wave<KIND, 0>(s, a, sh_rec, sh_stage, &sh_cnt, wu, lane, i0, j0, zc0, nz);
    } else {
      if (wu == 0) rowsweep_row_wave<KIND, 0>(s, a, sh_rec, sh_stage, &sh_cnt, 0, lane, i0, j0, zc0, nz);
      else if (wu == 1) rowsweep_row_wave<KIND, 1>(s, a, sh_rec, sh_stage, &sh_cnt, 1, lane, i0, j0, zc0, nz);
      else rowsweep_row_wave<KIND, 2>(s, a, sh_rec, sh_stage, &sh_cnt, 2, lane, i0, j0, zc0, nz);
    }
    return;
  }

  // ------------------------------- record wave -------------------------------
  __builtin_amdgcn_s_setprio(3);  // its few memory instructions should not queue behind the row waves' stores
  const bool soa = HASMAT && a.mat_layout == 3;  // FEMASM_MAT_ELEMENT_SOA: one (E, nu) per element, two planes
  const bool aos = HASMAT && (a.mat_layout == 0 || soa);
  const int rw = wu - 3;  // record wave index: it takes the passes ps = rw (mod NRW)
  unsigned epoch2 = 0;
  auto record_waves_sync = [&](int ln) {
    if constexpr (NRW > 1) row_waves_sync<NRW>(&sh_cnt2, epoch2, ln, a.bad);
  };
  __syncthreads();  // counters initialised
  auto load_layer = [&](int64_t gk, double (*cr)[3], int ln) {  // node layer gk of the halo'd tile -> registers
    if (a.rect) return;
    gk = gk < s.cks ? s.cks : (gk > a.cke - 1 ? a.cke - 1 : gk);
#pragma unroll
    for (int ps = 0; ps < CPASS; ps++) {
      if (ps % NRW != rw) continue;
      const int q = ln + 64 * ps;
      if (q < NXL) {
        const int cy = q / NXX, cx = q - NXX * cy;
        int64_t gi = i0 - 1 + cx, gj = j0 - 1 + cy;
        gi = gi < 0 ? 0 : (gi > M - 1 ? M - 1 : gi);
        gj = gj < 0 ? 0 : (gj > N - 1 ? N - 1 : gj);
        const double *src = a.coords + 3 * (gi + M * (gj + N * (gk - s.cks)));
        struct __attribute__((packed, aligned(8))) P2 { double x, y; };  // 16-byte load at 8-byte alignment
        const P2 xy = *reinterpret_cast<const P2 *>(src);
        cr[ps][0] = xy.x;
        cr[ps][1] = xy.y;
        cr[ps][2] = src[2];
      }
    }
  };
  auto commit_layer = [&](int slot, const double (*cr)[3], int ln) {
    if (a.rect) return;
#pragma unroll
    for (int ps = 0; ps < CPASS; ps++) {
      if (ps % NRW != rw) continue;
      const int q = ln + 64 * ps;
      if (q < NXL) {
#pragma unroll
        for (int d = 0; d < 3; d++) sh_x[slot][3 * q + d] = cr[ps][d];
      }
    }
  };
  // E[8], nu[8] of the elements of layer L: one element per lane and pass (its 128 contiguous bytes)
  auto load_mat = [&](int64_t L, double (*mraw)[16], int ln) {
    if (!aos) return;
#pragma unroll
    for (int ps = 0; ps < EPASS; ps++) {
      const int e = ln + 64 * ps, my = e / NEX, mx = e - NEX * my;
      const int64_t mgx = i0 - 1 + mx, mgy = j0 - 1 + my;
      const bool ok = e < NEL && mgx >= 0 && mgx < M - 1 && mgy >= 0 && mgy < N - 1 && L >= s.els && L < s.ele;
      if (ok && soa) {
        const int64_t eidx = mgx + (M - 1) * (mgy + (N - 1) * (L - s.mat_e_ks));
        const double vE = a.gp[eidx], vNu = a.gp[a.mat_stride + eidx];
#pragma unroll
        for (int q = 0; q < 8; q++) mraw[ps][q] = vE, mraw[ps][8 + q] = vNu;
      } else if (ok) {
        const double *mq = a.gp + (mgx + (M - 1) * (mgy + (N - 1) * (L - s.mat_e_ks))) * a.mat_stride + 96;
#pragma unroll
        for (int q = 0; q < 16; q++) mraw[ps][q] = mq[q];
      }
    }
  };
  // all eight E and all eight nu of an element must be bitwise equal, else the fast path does not apply
  auto reduce_mat = [&](const double (*mraw)[16], double *mE, double *mNu) {
    if (!aos) return;
#pragma unroll
    for (int ps = 0; ps < EPASS; ps++) {
      bool same = true;
#pragma unroll
      for (int q = 1; q < 8; q++) same = same && mraw[ps][q] == mraw[ps][0] && mraw[ps][8 + q] == mraw[ps][8];
      mE[ps] = mraw[ps][0];
      mNu[ps] = same ? mraw[ps][8] : __builtin_nan("");  // NaN marks "varies inside the element"
    }
  };
  // records of element layer L from node layers xb (bottom) and xt (top); verifies the premise of the fast path
  auto build_records = [&](int64_t L, const double *xb, const double *xt, double2 *recbuf, int ln, int ps, double mE,
                           double mNu) {
    {
      const int e = ln + 64 * ps;
      if (e >= NEL) return;
      const int ey = e / NEX, ex = e - NEX * ey;
      const int64_t gex = i0 - 1 + ex, gey = j0 - 1 + ey;
      double2 *dst = recbuf + (ex + SY * ey);
      const bool eok = gex >= 0 && gex < M - 1 && gey >= 0 && gey < N - 1 && L >= s.els && L < s.ele;
      if (!eok) {
#pragma unroll
        for (int q = 0; q < NPA; q++) dst[q * NS] = make_double2(0.0, 0.0);
        return;
      }
      bool affine = true;
      double J[3][3];
      if (a.rect) {
        // verified rectilinear mesh: the element is the box X[ex..ex+1] x Y[ey..ey+1] x Z[l..l+1]; the same
        // differences of the same coordinates as below, so the records are bitwise those of the other branch
        const int l = (int)(L - L0);
#pragma unroll
        for (int r_ = 0; r_ < 3; r_++)
#pragma unroll
          for (int d = 0; d < 3; d++) J[r_][d] = 0.0;
        J[0][0] = 0.5 * (sh_x[0][ex + 1] - sh_x[0][ex]);
        J[1][1] = 0.5 * (sh_x[1][ey + 1] - sh_x[1][ey]);
        J[2][2] = 0.5 * (sh_x[2][l + 1] - sh_x[2][l]);
      } else {
        double x[8][3];
#pragma unroll
        for (int n = 0; n < 8; n++) {
          const double *xn = (HDK[n] ? xt : xb) + 3 * ((ex + HDI[n]) + NXX * (ey + HDJ[n]));
#pragma unroll
          for (int d = 0; d < 3; d++) x[n][d] = xn[d];
        }
        // edges: xi-direction 0->3, 1->2, 4->7, 5->6 ; eta 0->1, 3->2, 4->5, 7->6 ; zeta 0->4, 1->5, 2->6, 3->7
#pragma unroll
        for (int d = 0; d < 3; d++) {
          const double e1 = x[3][d] - x[0][d], e2 = x[1][d] - x[0][d], e3 = x[4][d] - x[0][d];
          affine = affine && (x[2][d] - x[1][d] == e1) && (x[7][d] - x[4][d] == e1) && (x[6][d] - x[5][d] == e1);
          affine = affine && (x[2][d] - x[3][d] == e2) && (x[5][d] - x[4][d] == e2) && (x[6][d] - x[7][d] == e2);
          affine = affine && (x[5][d] - x[1][d] == e3) && (x[6][d] - x[2][d] == e3) && (x[7][d] - x[3][d] == e3);
          J[0][d] = 0.5 * e1;  // J_rc = d x_c / d xi_r
          J[1][d] = 0.5 * e2;
          J[2][d] = 0.5 * e3;
        }
      }
      if constexpr (HASMAT) {  // elastic: axis-aligned boxes only
        double E = a.E, nu = a.nu;
        if (aos) {
          E = mE;
          nu = mNu;
          affine = affine && (mNu == mNu);
        }
        affine = affine && J[0][1] == 0.0 && J[0][2] == 0.0 && J[1][0] == 0.0 && J[1][2] == 0.0 && J[2][0] == 0.0 &&
                 J[2][1] == 0.0;
        if (!affine) atomicAdd(a.bad + 1, 1ULL);
        const double wJ = 1.0 * (J[0][0] * J[1][1] * J[2][2]);
        if (!(wJ > 0.0)) atomicAdd(a.bad, 1ULL);
        const double ai[3] = {1.0 / J[0][0], 1.0 / J[1][1], 1.0 / J[2][2]};
        double d11, d12, d44;
        scaled_coefs<KIND>(E, nu, wJ, d11, d12, d44);
        const double q0 = ai[0] * ai[0], q1 = ai[1] * ai[1], q2 = ai[2] * ai[2];
        const double q01 = ai[0] * ai[1], q02 = ai[0] * ai[2], q12 = ai[1] * ai[2];
        dst[0] = make_double2(d11 * q0, d11 * q1);
        if constexpr (NPA == 6) {
          dst[NS] = make_double2(d11 * q2, d44 * q0);
          dst[2 * NS] = make_double2(d44 * q1, d44 * q2);
          dst[3 * NS] = make_double2(d12 * q01, d12 * q02);
          dst[4 * NS] = make_double2(d12 * q12, d44 * q01);
          dst[5 * NS] = make_double2(d44 * q02, d44 * q12);
        }
      } else {  // Poisson: any parallelepiped
        if (!affine) atomicAdd(a.bad + 1, 1ULL);
        if (J[0][1] == 0.0 && J[0][2] == 0.0 && J[1][0] == 0.0 && J[1][2] == 0.0 && J[2][0] == 0.0 && J[2][1] == 0.0) {
          // axis-aligned box: iJ = diag(1/J_ii), C = detJ diag(1/J_ii^2); same operations as the general branch
          // below with the zero terms dropped, so both give the same bits
          const double wJ = 1.0 * (J[0][0] * J[1][1] * J[2][2]);
          if (!(wJ > 0.0)) atomicAdd(a.bad, 1ULL);
          const double t17 = 1.0 / (J[0][0] * J[1][1] * J[2][2]);
          const double i0 = (J[1][1] * J[2][2]) * t17, i1 = (J[0][0] * J[2][2]) * t17, i2 = (J[0][0] * J[1][1]) * t17;
          dst[0] = make_double2((i0 * i0) * wJ, 0.0);
          dst[NS] = make_double2(0.0, (i1 * i1) * wJ);
          dst[2 * NS] = make_double2(0.0, (i2 * i2) * wJ);
          return;
        }
        double iJ[3][3], det;
        inv3_ref(J, iJ, det);
        const double wJ = 1.0 * det;
        if (!(wJ > 0.0)) atomicAdd(a.bad, 1ULL);
        // physical gradient g_i = sum_k iJ[i][k] ghat_k  ->  g_a.g_b = ghat_a^T (iJ^T iJ) ghat_b
        double Cm[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
          for (int l = k; l < 3; l++)
            Cm[k][l] = (iJ[0][k] * iJ[0][l] + iJ[1][k] * iJ[1][l] + iJ[2][k] * iJ[2][l]) * wJ;
        dst[0] = make_double2(Cm[0][0], Cm[0][1]);
        dst[NS] = make_double2(Cm[0][2], Cm[1][1]);
        dst[2 * NS] = make_double2(Cm[1][2], Cm[2][2]);
      }
    }
  };

  // Two register sets: what step st commits was requested during step st-2.  The compute unit's memory queue
  // is a FIFO shared with the row waves' stores, so under a store-bound load a read waits about one step.
  double cr[2][CPASS][3], mraw[2][EPASS][16], mE[EPASS], mNu[EPASS];
#pragma unroll
  for (int p = 0; p < 2; p++) {
#pragma unroll
    for (int ps = 0; ps < CPASS; ps++)
#pragma unroll
      for (int d = 0; d < 3; d++) cr[p][ps][d] = 0.0;
#pragma unroll
    for (int ps = 0; ps < EPASS; ps++)
#pragma unroll
      for (int q = 0; q < 16; q++) mraw[p][ps][q] = 0.0;
  }
#pragma unroll
  for (int ps = 0; ps < EPASS; ps++) mE[ps] = mNu[ps] = 0.0;
  auto lds_fence = [&]() {  // LDS written by this wave is read by other lanes of this wave: the LDS queue of a
    asm volatile("" ::: "memory");    // wave is in order, only the compiler has to keep the order
    __builtin_amdgcn_wave_barrier();
  };

  // prologue: node layers L0..L0+2, records of element layers L0 and L0+1; requests for steps 0 and 1
  if (a.rect) {  // the three coordinate lines of this workgroup: X[i0-1 ..], Y[j0-1 ..], Z[L0 ..], clamped like the layers
    if (lane < NXX) {
      int64_t gi = i0 - 1 + lane;
      gi = gi < 0 ? 0 : (gi > M - 1 ? M - 1 : gi);
      sh_x[0][lane] = a.coords[3 * gi];
    }
    if (lane < R_::TY + 2) {
      int64_t gj = j0 - 1 + lane;
      gj = gj < 0 ? 0 : (gj > N - 1 ? N - 1 : gj);
      sh_x[1][lane] = a.coords[3 * (M * gj) + 1];
    }
    for (int q = lane; q < nz + 3; q += 64) {
      int64_t gk = L0 + q;
      gk = gk < s.cks ? s.cks : (gk > a.cke - 1 ? a.cke - 1 : gk);
      sh_x[2][q] = a.coords[3 * (M * N * (gk - s.cks)) + 2];
    }
  }
#pragma unroll 1
  for (int q = 0; q < 3; q++) {
    load_layer(L0 + q, cr[0], lane);
    commit_layer(q, cr[0], lane);
  }
  record_waves_sync(lane);  // every record wave has staged its share of the three node layers
#pragma unroll 1
  for (int q = 0; q < 2; q++) {
    load_mat(L0 + q, mraw[0], lane);
    reduce_mat(mraw[0], mE, mNu);
    lds_fence();
#pragma unroll
    for (int ps = 0; ps < EPASS; ps++)
      if (ps % NRW == rw)
        build_records(L0 + q, sh_x[q], sh_x[q + 1], sh_rec + q * (NPA * NS), lane, ps, mE[ps], mNu[ps]);
  }
  load_layer(L0 + 3, cr[0], lane);
  load_mat(L0 + 2, mraw[0], lane);
  load_layer(L0 + 4, cr[1], lane);
  load_mat(L0 + 3, mraw[1], lane);
  __syncthreads();

  int q3 = 0;  // st % 3
  PH_DECL;
  auto step = [&](int st, double (*crp)[3], double (*mrawp)[16]) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int q3a = (q3 == 2) ? 0 : q3 + 1, q3b = (q3a == 2) ? 0 : q3a + 1;
    PH_T(0);
    __syncthreads();
    PH_ACC(0, 1);
    if (st + 1 < nz) {  // records of element layer L0+st+2, read by the row waves from the next step on
      commit_layer(q3, crp, ln);  // node layer L0+st+3 replaces layer L0+st
      reduce_mat(mrawp, mE, mNu);
      if (st + 3 < nz) {
        load_layer(L0 + st + 5, crp, ln);
        load_mat(L0 + st + 4, mrawp, ln);
      }
      lds_fence();
      record_waves_sync(ln);  // the new node layer is complete
      PH_ACC(1, 2);
#pragma unroll
      for (int ps = 0; ps < EPASS; ps++)
        if (ps % NRW == rw)
          build_records(L0 + st + 2, sh_x[q3b], sh_x[q3], sh_rec + q3b * (NPA * NS), ln, ps, mE[ps], mNu[ps]);
      PH_ACC(2, 3);
    }
    q3 = q3a;
  };
  for (int st = 0; st < nz; st += 2) {
    step(st, cr[0], mraw[0]);
    if (st + 1 < nz) step(st + 1, cr[1], mraw[1]);
  }
  PH_FLUSH(16);
}

// ---------------------------------------------------------------------------
// S3 / R3 on a verified rectilinear mesh (level 2), ONE-SHOT UNITS: k_assemble_s3_unit.
// What the memory system of this part takes fastest is not a set of persistent workgroups that each walk the mesh
// (4.5-4.8 TB/s for the elastic K kernel's 4.25 GB, which is where k_assemble_rowsweep sits) but short-lived workgroups
// that each write ONE contiguous block and end, dispatched in memory order (5.7-6.9 TB/s; scripts/ubench/fill_roof*.hip).
// So: one workgroup = ONE WAVE; a unit = 8 x-consecutive interior nodes of one (j, k) line = one 15.5 KB run of the CSR
// array; a wave walks KArgs::zc x-consecutive units (one contiguous piece of the array) and ends; the grid is
// (x pieces, lines, layers) = memory order; no barrier, no polling, nothing carried between units but a prefetch.
//   lane      (layer AS, node lx, element e = (oy, ox)): ONE of the eight box elements around ONE node.  The lane
//             builds that element's record itself (two x, two y, two z coordinates and the element's (E, nu), all
//             eight Gauss points bitwise equal, else bad[1]) with the arithmetic of k_assemble_rowsweep's record wave.
//   uniform   every lane runs the code of the element whose (+,+,+) corner is the node: with s = the node's corner
//             signs in the lane's element, R_ab[k][l] = s_k s_l R_a'b'[k][l] for the mirrored pair (a', b'), so the
//             lane multiplies its six off-diagonal record numbers by +-1 (exact) and indexes its neighbours by
//             "same side / far side" per axis: 8 neighbours x 3 column dofs = 24 accumulators per dof row.
//   sum       a neighbour is shared by 1, 2, 4 or 8 of the node's elements: quad DPP adds over (oy, ox), then the two
//             layers add in place in the stage; a fixed tree, so deterministic (the last bit can differ from the chain
//             order of k_assemble_rowsweep, which is why the two kernels never share a K).
//   run       deposit in HBM order in the 15.5 KB stage, then 16 B per lane from the 128-byte boundary below the run
//   prefetch  the next unit's x edge and (E, nu) are requested right after this unit's record is built and before its
//             stores (loads and stores retire through one in-order counter: a load issued before a store does not
//             wait for it).
// Measured (S3 128^3, 254 016 units): one unit per wave 1.09 ms (phase clocks per unit: loads + record
// 10.4k, rows 5.2k, stage reads 1.3k, stores until acknowledged 4.5k cycles, ten waves per CU), whole line per wave
// 0.91 ms against 0.93 ms for k_assemble_rowsweep on the same box; without its stores 0.61 ms (row-sweep: 0.89 ms), so
// neither kernel is at the store roof: this one is bound by its ~1 060 instructions per unit (620 vector: records rebuilt
// by the eight lanes that share an element, 216 for the cross-lane sums; VALU 29 % and LDS 46 % busy, nothing saturated;
// a third wave per SIMD, a deeper prefetch and conflict-free deposits were each measured and change nothing).  Where it clearly wins
// is below the headline size: 96^3 0.35 against 0.46 ms, 64^3 0.10 against 0.18 ms, 32^3 0.022 against 0.17 ms (the
// row-sweep grid has too few workgroups there and a two-layer prologue each).
// ---------------------------------------------------------------------------
constexpr int SU_TX = 8;
#ifndef SU_OCC
#define SU_OCC 2  // waves per SIMD the register allocation aims at (3: measured slower, the prefetched unit spills)
#endif
template <int CTRL>
__device__ inline double quad_perm(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);  // quad_perm reads a lane of the same quad: always valid
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
#if SU_OCC > 2
#define SU_SCHED __builtin_amdgcn_sched_barrier(0)  // one piece at a time: interleaved, the pieces' accumulators add up
#else
#define SU_SCHED
#endif
// row C of the lane's node from the lane's element, the neighbours on the far side (MZ = 1) or on the node's side
// (MZ = 0) of the element in z; stg = stage + node block + self neighbour
template <int C, int MZ>
__device__ inline void unit_half(const double *pd, const double *pe, const double *po, const double *pq, double *stg,
                                 int pz, int py, int px, int e, bool lower, bool live) {
  constexpr int ia = hex_idx(1, 1, 1);
  constexpr int C1 = (C + 1) % 3, C2 = (C + 2) % 3;
  double acc[12];  // [my][mx][cc], m = 0: the neighbour on the node's side of the element, 1: the far side
#pragma unroll
  for (int my = 0; my <= 1; my++)
#pragma unroll
    for (int mx = 0; mx <= 1; mx++) {
      const double(*R)[3] = d_refR.R[ia][hex_idx(1 - mx, 1 - my, 1 - MZ)];
      const int t = my * 2 + mx;
#pragma unroll
      for (int cc = 0; cc < 3; cc++) {
        double v = 0.0;
        if (cc == C) {
          v = fma(R[C][C], pd[C], v);
          v = fma(R[C1][C1], pe[C1], v);
          v = fma(R[C2][C2], pe[C2], v);
        } else {
          v = fma(R[C][cc], po[C + cc - 1], v);
          v = fma(R[cc][C], pq[C + cc - 1], v);
        }
        acc[t * 3 + cc] = v;
      }
    }
  // the elements of this layer that share a neighbour: (oy, ox) partners in the quad
#pragma unroll
  for (int cc = 0; cc < 3; cc++) {
    double &a00 = acc[0 * 3 + cc], &a01 = acc[1 * 3 + cc], &a10 = acc[2 * 3 + cc];
    a00 += quad_perm<0xB1>(a00);  // ox partner
    a00 += quad_perm<0x4E>(a00);  // oy partners
    a01 += quad_perm<0x4E>(a01);  // far in x, same y: the element across y
    a10 += quad_perm<0xB1>(a10);  // far in y, same x: the element across x
  }
#if SU_OCC > 2
  // (pinned: the compiler otherwise sinks the last add of every sum into the predicated deposit below and keeps both of
  // its operands alive until then, which doubles the accumulators' registers)
#pragma unroll
  for (int q = 0; q < 12; q++) asm volatile("" : "+v"(acc[q]));
#endif
  // deposit.  A shared neighbour's sum is the same bits in every lane that holds it (the adds above commute), so all of
  // them write it: no exec mask per group of owners (measured: the 36 predicated blocks per unit cost more issue slots
  // than the duplicate writes cost LDS cycles).  The middle third (MZ = 0) is the sum of both element layers: the lanes
  // of layer kn-1 write, then the lanes of layer kn add theirs in place -- LDS executes a wave's accesses in order, so
  // the second sees the first.
  auto put = [&](double *d, int t, bool add) {
#pragma unroll
    for (int cc = 0; cc < 3; cc++) d[cc] = add ? d[cc] + acc[t * 3 + cc] : acc[t * 3 + cc];
  };
  double *d = stg + C * 81 + (MZ ? pz : 0);
  (void)e;
#pragma unroll
  for (int pass = (MZ ? 0 : 1); pass < (MZ ? 1 : 3); pass++) {  // 0: far third, 1: middle third from below, 2: from above
    const bool wz = live && (pass == 0 || (pass == 1) == lower);
    const bool add = pass == 2;
    if (wz) {
      put(d + py + px, 3, add);
      put(d + py, 2, add);
      put(d + px, 1, add);
      put(d, 0, add);
    }
    if (pass == 1) asm volatile("" ::: "memory");
  }
}
template <int C>
__device__ inline void unit_row(const double *pd, const double *pe, const double *po, const double *pq, double *stg,
                                int pz, int py, int px, int e, bool lower, bool live) {
  SU_SCHED;
  unit_half<C, 1>(pd, pe, po, pq, stg, pz, py, px, e, lower, live);
  SU_SCHED;
  unit_half<C, 0>(pd, pe, po, pq, stg, pz, py, px, e, lower, live);
  SU_SCHED;
}
__global__ __launch_bounds__(64, SU_OCC) void k_assemble_s3_unit(const Slab s, const KArgs a) {
  using R_ = RowSweep<KIND_ELASTIC>;
  constexpr int NB = R_::NB, RUN = SU_TX * NB;
  constexpr int PAD = 16;  // doubles either side of the run: the store loop reads whole 16-byte chunks from a line boundary
  __shared__ __attribute__((aligned(16))) double sh_all[PAD + RUN + PAD + 2];
  const int lane = threadIdx.x;
  PH_DECL;
  const int e = lane & 3, lx = (lane >> 2) & 7, AS = lane >> 5;  // AS = 1: element layer kn-1 (the node on its top face)
  const int ox = (e & 1) - 1, oy = (e >> 1) - 1;
  const int64_t M = s.dim[0], N = s.dim[1];
  const int NU = a.zc;  // x-consecutive units of this wave: their runs are one contiguous piece of the CSR array
  const int64_t ifirst = a.x0 + (int64_t)blockIdx.x * NU * SU_TX, j = a.y0 + blockIdx.y, kn = a.s0 + blockIdx.z;
  const int64_t gey = j + oy, L = kn - AS;
  const bool eok = L >= s.els && L < s.ele;
  const bool soa = a.mat_layout == 3 && eok;  // FEMASM_MAT_ELEMENT_SOA: one (E, nu) per element, two planes
  const bool aos = (a.mat_layout == 0 || a.mat_layout == 3) && eok;
  auto zline = [&](int64_t k) {
    const int64_t kc = k < s.cks ? s.cks : (k > a.cke - 1 ? a.cke - 1 : k);
    return a.coords[3 * (M * N * (kc - s.cks)) + 2];
  };
  // what a unit reads: the x edge of the lane's element and its (E, nu) at the eight Gauss points; requested one unit
  // ahead, right after the previous unit's record is built (the registers are free again) and before its stores:
  // gfx9 retires loads and stores in one in-order counter, so a load issued BEFORE a store does not wait for it.
  // (Two units ahead, with a second register set: measured the same 0.905 ms, so the load latency is covered.)
  struct UnitIn { double xa, xb, m[16]; };
  UnitIn inA;
  auto load_unit = [&](int64_t i0, UnitIn &in) {
    const int nv = (a.x1 - i0 < SU_TX) ? (int)(a.x1 - i0) : SU_TX;
    const int64_t gex = i0 + (lx < nv ? lx : nv - 1) + ox;
    in.xa = a.coords[3 * gex], in.xb = a.coords[3 * (gex + 1)];
    if (soa) {
      const int64_t eidx = gex + (M - 1) * (gey + (N - 1) * (L - s.mat_e_ks));
      const double vE = a.gp[eidx], vNu = a.gp[a.mat_stride + eidx];
#pragma unroll
      for (int q = 0; q < 8; q++) in.m[q] = vE, in.m[8 + q] = vNu;
    } else if (aos) {
      const double *mq = a.gp + (gex + (M - 1) * (gey + (N - 1) * (L - s.mat_e_ks))) * a.mat_stride + 96;
#pragma unroll
      for (int q = 0; q < 16; q++) in.m[q] = mq[q];
    }
  };
  if (ifirst < a.x1) load_unit(ifirst, inA);
  // the y and z edges are those of the whole line (k_check_rectilinear has verified that the three coordinate lines
  // describe the mesh)
  const double ya = a.coords[3 * (M * gey) + 1], yb = a.coords[3 * (M * (gey + 1)) + 1];
  const double za = zline(L), zb = zline(L + 1);
  const double J1 = 0.5 * (yb - ya), J2 = 0.5 * (zb - za);
  const double ai1 = 1.0 / J1, ai2 = 1.0 / J2;
  // corner signs of the node in this element: the mirror that makes it the (+,+,+) corner flips the mixed terms
  const double sx = ox ? 1.0 : -1.0, sy = oy ? 1.0 : -1.0, sz = AS ? 1.0 : -1.0;
  const double s01 = sx * sy, s02 = sx * sz, s12 = sy * sz;
  // far-side neighbour steps in the node block: (dz, dy, dx) -> ((dz+1)*9 + (dy+1)*3 + (dx+1)) * 3 + column dof
  const int pz = AS ? -27 : 27, py = oy ? -9 : 9, px = ox ? -3 : 3;
  const bool lower = AS == 1;
  auto unit = [&](int u, UnitIn &in) -> bool {
    const int64_t i0 = ifirst + (int64_t)u * SU_TX;
    if (i0 >= a.x1) return false;
    // the run in the CSR array and in the stage: the stage copy starts (head & 1) doubles above a 16-byte boundary of LDS,
    // head = the run's distance from a 128-byte line of HBM, so that a 16-byte chunk of HBM is a 16-byte chunk of LDS
    // (one ds_read_b128 per chunk in the copy-out instead of a ds_read2_b64 at a quarter of its rate)
    double *runp = a.out + (row_start(s, i0, j, kn, 0) - a.base_off);
    const int head = (int)((reinterpret_cast<uintptr_t>(runp) & 127) >> 3);  // doubles above a line boundary
    double *sh_stage = sh_all + PAD + (head & 1);
    double *stg = sh_stage + lx * NB + 39;
    const double xa = in.xa, xb = in.xb;
    const double(&m)[16] = in.m;
    const int nvx = (a.x1 - i0 < SU_TX) ? (int)(a.x1 - i0) : SU_TX;
    const bool live = lx < nvx;
    double pd[3], pe[3], po[3], pq[3];
    if (!eok) {
#pragma unroll
      for (int q = 0; q < 3; q++) pd[q] = pe[q] = po[q] = pq[q] = 0.0;
    } else {
      double mE = a.E, mNu = a.nu;
      if (aos) {
        bool same = true;
#pragma unroll
        for (int q = 1; q < 8; q++) same = same && m[q] == m[0] && m[8 + q] == m[8];
        mE = m[0];
        mNu = same ? m[8] : __builtin_nan("");  // NaN marks "varies inside the element"
      }
      // the operations of k_assemble_rowsweep's record wave on a box (same bits per element)
      const double J0 = 0.5 * (xb - xa);
      if (live && !(mNu == mNu)) atomicAdd(a.bad + 1, 1ULL);
      const double wJ = 1.0 * (J0 * J1 * J2);
      if (live && !(wJ > 0.0)) atomicAdd(a.bad, 1ULL);
      const double ai[3] = {1.0 / J0, ai1, ai2};
      double d11, d12, d44;
      scaled_coefs<KIND_ELASTIC>(mE, mNu, wJ, d11, d12, d44);
      const double q0 = ai[0] * ai[0], q1 = ai[1] * ai[1], q2 = ai[2] * ai[2];
      const double q01 = ai[0] * ai[1], q02 = ai[0] * ai[2], q12 = ai[1] * ai[2];
      pd[0] = d11 * q0, pd[1] = d11 * q1, pd[2] = d11 * q2;
      pe[0] = d44 * q0, pe[1] = d44 * q1, pe[2] = d44 * q2;
      po[0] = s01 * (d12 * q01), po[1] = s02 * (d12 * q02), po[2] = s12 * (d12 * q12);
      pq[0] = s01 * (d44 * q01), pq[1] = s02 * (d44 * q02), pq[2] = s12 * (d44 * q12);
    }
    PH_PIN(pd[0] + pq[2]);
    PH_ACC(0, 1);
    if (u + 1 < NU && i0 + SU_TX < a.x1) load_unit(i0 + SU_TX, in);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();  // the previous unit's reads of the stage are done
    unit_row<0>(pd, pe, po, pq, stg, pz, py, px, e, lower, live);
    unit_row<1>(pd, pe, po, pq, stg, pz, py, px, e, lower, live);
    unit_row<2>(pd, pe, po, pq, stg, pz, py, px, e, lower, live);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();  // the run is in the stage, in HBM order
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    PH_ACC(1, 2);
    const int n = nvx * NB;
    constexpr int NIT = (RUN / 2 + 8 + 63) / 64;
    // chunk q = the q-th 16 bytes from that boundary = stage[2q - head, 2q - head + 1]; whole chunks here, the two
    // possible half chunks (an odd head, an odd end) after the loop
    const int s0 = 2 * lane - head;
    const double *src = sh_stage + s0;
    double *dstp = runp + s0;
    constexpr int NPART = SU_OCC > 2 ? 4 : 2, NH = NIT / NPART;  // in parts: 16 chunks in flight would cost 64 registers
    static_assert(NIT % NPART == 0, "store rounds");
    // full = a whole unit (wave-uniform): only the first round can start below the run and only the last rounds can end
    // beyond it; the rounds between need no exec mask (an exec mask per round costs four scalar issue slots; measured
    // 0.898 against 0.913 ms with every round predicated.  The same change in k_assemble_p3_rect / k_assemble_e2_rect
    // measured SLOWER there, 0.74 against 0.68 ms and 0.423 against 0.414 ms, and was not kept)
    auto copy_out = [&](auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
      for (int h = 0; h < NPART; h++) {
        double2 v[NH];
#pragma unroll
        for (int q = 0; q < NH; q++) {
          const int it = h * NH + q;
          int o = 128 * it;
          if (128 * (it + 1) > RUN + PAD) o = (s0 + o + 2 <= RUN + PAD) ? o : 0;  // last round: stay inside the array
          v[q] = *reinterpret_cast<const double2 *>(src + o);  // 16-byte aligned: s0 + (head & 1) and o are even
        }
        if (h == NPART - 1) {
          PH_PIN(v[0].x + v[NH - 1].y);
          PH_ACC(2, 3);
        }
#pragma unroll
        for (int q = 0; q < NH; q++) {
          const int it = h * NH + q;
          const bool always = FULL && it > 0 && 2 * 63 + 128 * it + 2 <= RUN;
          if (always || (s0 + 128 * it >= 0 && s0 + 128 * it + 2 <= n)) {
#if FEMASM_EXP == 1
            asm volatile("" ::"v"(v[q].x), "v"(v[q].y));
#else
            store_pair(dstp + 128 * it, v[q]);
#endif
          }
        }
        asm volatile("" ::: "memory");
      }
    };
    if (nvx == SU_TX) copy_out(std::true_type{});
    else copy_out(std::false_type{});
    if ((head & 1) && lane == 0) runp[0] = sh_stage[0];
    if (((head + n) & 1) && lane == 1) runp[n - 1] = sh_stage[n - 1];
    PH_ACC(3, 4);
    PH_T(0);
    return true;
  };
  for (int u = 0; u < NU; u++)
    if (!unit(u, inA)) break;
  if (blockIdx.x == 1 && (blockIdx.y & 15) == 5) {  // profiling build: a sample of the waves, or the flush itself is what is timed
    PH_FLUSH(8);
  }
}

// ---------------------------------------------------------------------------
// P3 on a verified rectilinear mesh (level 2): one autonomous wave per workgroup, no record wave, no barrier.
// For dof 1 a tile row of 64 x-consecutive interior nodes is a contiguous 13.8 KB run of the CSR array that ONE
// wave can own: lane = node, 27 accumulators = its whole row; deposit into a wave-private stage in HBM order (all
// 64 lanes at once), then 16 B per lane from the 128-byte boundary below the run, as in k_assemble_rowsweep --
// but nothing to hand-shake.  The element records are three numbers (C = detJ diag(1/J_ii^2)) that follow from the
// three coordinate lines, so every lane derives those of its eight elements itself: per step one new element
// layer (4 elements, 4 divisions), the previous layer's kept in registers.  Same operations in the same order as
// the record path of k_assemble_rowsweep<POISSON> (the zero off-diagonal terms dropped): same bits.
// The premise is the rectilinearity pre-pass (bad[1]); on failure the host reassembles at level 1.
// ---------------------------------------------------------------------------
constexpr int P3W_TX = 64;
__global__ __launch_bounds__(64) void k_assemble_p3_rect(const Slab s, const KArgs a) {
  constexpr int ROW = 27, RUN = P3W_TX * ROW, END = 18;  // END: values of the row of an x-end node (2 x 3 x 3 neighbours)
  __shared__ double sh_stage[END + RUN + END];
  const int lane = threadIdx.x;
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t i0 = a.x0 + (int64_t)blockIdx.x * P3W_TX, j = a.y0 + blockIdx.y;
  const int64_t zc0 = a.s0 + (int64_t)blockIdx.z * a.zc;
  const int64_t zc1 = (zc0 + a.zc < a.s1) ? zc0 + a.zc : a.s1;
  const int nz = (int)(zc1 - zc0);
  const int nvx = (a.x1 - i0 < P3W_TX) ? (int)(a.x1 - i0) : P3W_TX;  // valid nodes of the run
  const int64_t i = (lane < nvx) ? i0 + lane : i0 + nvx - 1;          // idle lanes repeat the last node
  // half edges of the elements either side of the node, per axis: [0] = element index-1, [1] = element index
  const double xm = a.coords[3 * (i - 1)], xc = a.coords[3 * i], xp = a.coords[3 * (i + 1)];
  const double ym = a.coords[3 * (M * (j - 1)) + 1], yc = a.coords[3 * (M * j) + 1], yp = a.coords[3 * (M * (j + 1)) + 1];
  const double Jx[2] = {0.5 * (xc - xm), 0.5 * (xp - xc)}, Jy[2] = {0.5 * (yc - ym), 0.5 * (yp - yc)};
  // a.ends (Dirichlet shell, P3:286-290): the rows of the x-end nodes 0 and M-1 of this (j, k) line sit right before
  // and after the interior rows in the value array; the first and last wave of the line write them with their run
  // (diag * e_row, diag = valence of the node in the assembled elements), so no line of K is written in two goes
  const int pre = (a.ends && i0 == 1) ? END : 0, post = (a.ends && i0 + nvx == M - 1) ? END : 0;
  auto zline = [&](int64_t k) { return a.coords[3 * (M * N * (k - s.cks)) + 2]; };
  // C of the four elements (oy, ox) of element layer L; zero outside the layers this rank assembles
  auto layer = [&](int64_t L, double (*C)[3]) {
    const bool ok = L >= s.els && L < s.ele;
    const double Jz = ok ? 0.5 * (zline(L + 1) - zline(L)) : 1.0;
#pragma unroll
    for (int oy = 0; oy < 2; oy++)
#pragma unroll
      for (int ox = 0; ox < 2; ox++) {
        const double J0 = Jx[ox], J1 = Jy[oy], J2 = Jz;
        const double wJ = 1.0 * (J0 * J1 * J2);
        if (ok && !(wJ > 0.0)) atomicAdd(a.bad, 1ULL);
        const double t17 = 1.0 / (J0 * J1 * J2);
        const double q0 = (J1 * J2) * t17, q1 = (J0 * J2) * t17, q2 = (J0 * J1) * t17;
        C[2 * oy + ox][0] = ok ? (q0 * q0) * wJ : 0.0;
        C[2 * oy + ox][1] = ok ? (q1 * q1) * wJ : 0.0;
        C[2 * oy + ox][2] = ok ? (q2 * q2) * wJ : 0.0;
      }
  };
  double Clo[4][3], Chi[4][3];
  layer(zc0 - 1, Clo);
  for (int st = 0; st < nz; st++) {
    const int64_t kn = zc0 + st;
    layer(kn, Chi);
    double acc[ROW];
#pragma unroll
    for (int q = 0; q < ROW; q++) acc[q] = 0.0;
#pragma unroll
    for (int as = 1; as >= 0; as--)  // element layer kn-1 (the node is on its top face), then kn
#pragma unroll
      for (int oy = -1; oy <= 0; oy++)
#pragma unroll
        for (int ox = -1; ox <= 0; ox++) {
          const double *C = as ? Clo[2 * (oy + 1) + (ox + 1)] : Chi[2 * (oy + 1) + (ox + 1)];
          const int ia = hex_idx(-ox, -oy, as);
#pragma unroll
          for (int bz = 0; bz <= 1; bz++)
#pragma unroll
            for (int by = 0; by <= 1; by++)
#pragma unroll
              for (int bx = 0; bx <= 1; bx++) {
                const double(*R)[3] = d_refR.R[ia][hex_idx(bx, by, bz)];
                const int t = ((bz - as + 1) * 3 + (oy + by + 1)) * 3 + (ox + bx + 1);
                double v = acc[t];
                v = fma(R[0][0], C[0], v);
                v = fma(R[1][1], C[1], v);
                v = fma(R[2][2], C[2], v);
                acc[t] = v;
              }
        }
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int d = 0; d < 3; d++) Clo[q][d] = Chi[q][d];
    // deposit the row in HBM order, then write the run from the 128-byte boundary below its start
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();  // the previous step's reads of the stage are done
    {
      double *dst = sh_stage + pre + lane * ROW;
#pragma unroll
      for (int q = 0; q < ROW; q++) dst[q] = acc[q];
    }
    if (pre + post) {
      const int vz = (kn - 1 >= s.els && kn - 1 < s.ele) + (kn >= s.els && kn < s.ele);
      const double diag = (double)(1 * 2 * vz);  // x-end node of an interior (j, k) line
      // self position among the (dk, dj, di) neighbours: di in {0,1} at node 0 -> 8, di in {-1,0} at node M-1 -> 9
      if (pre && lane < END) sh_stage[lane] = (lane == 8) ? diag : 0.0;
      if (post && lane < END) sh_stage[pre + nvx * ROW + lane] = (lane == 9) ? diag : 0.0;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int n = pre + nvx * ROW + post;
    double *runp = a.out + (row_start(s, i0, j, kn, 0) - a.base_off) - pre;
    const int head = (int)((reinterpret_cast<uintptr_t>(runp) & 127) >> 3);  // doubles above a line boundary
    const int nchunk = (head + n + 1) >> 1;
    constexpr int NIT = ((RUN + 2 * END) / 2 + 8 + 63) / 64;
    double2 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {  // chunk q = the q-th 16 bytes from that boundary; LDS reads first
      const int s0 = 2 * (lane + 64 * it) - head;
      const int sa = s0 < 0 ? 0 : (s0 < n ? s0 : n - 1), sb = s0 + 1 < 0 ? 0 : (s0 + 1 < n ? s0 + 1 : n - 1);
      v[it] = make_double2(sh_stage[sa], sh_stage[sb]);
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int q = lane + 64 * it, s0 = 2 * q - head;
      if (q < nchunk) {
        if (s0 >= 0 && s0 + 1 < n) store_pair(runp + s0, v[it]);
        else if (s0 == -1) runp[0] = v[it].y;
        else if (s0 == n - 1) runp[s0] = v[it].x;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// 2-D fast path (plane strain, E2).  Premise: rectilinear mesh (k_check_rectilinear2d) with one (E,nu) over the four
// Gauss points of an element.  Record of an element: d11 a_i^2 (2), d44 a_i^2 (2), d12 a_x a_y, d44 a_x a_y with
// a_i = 1/J_ii; R_ab = sum_p dN_a/dxi dN_b/dxi^T a compile-time tensor (closed form checked against the Gauss sum).
// ---------------------------------------------------------------------------
struct RefR2 { double R[4][4][2][2]; };
constexpr RefR2 make_refR2() {
  RefR2 o{};
  const Tab2 t = make_tab2();
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++)
      for (int k = 0; k < 2; k++)
        for (int l = 0; l < 2; l++) {
          // closed form of the 2x2 Gauss sum (s = corner signs): k == l: s_a[k] s_b[k] (1 + s_a[m] s_b[m] g^2) / 4,
          // k != l: s_a[k] s_b[l] / 4
          const double sa[2] = {QSX[a], QSY[a]}, sb[2] = {QSX[b], QSY[b]};
          o.R[a][b][k][l] = (k == l) ? sa[k] * sb[k] * 0.25 * (1.0 + sa[1 - k] * sb[1 - k] * (GAUSS_G * GAUSS_G))
                                     : sa[k] * sb[l] * 0.25;
          double v = 0.0;
          for (int p = 0; p < 4; p++) v += t.G[p][k][a] * t.G[p][l][b];
          const double d = v - o.R[a][b][k][l];
          if (d > 1e-16 || d < -1e-16) o.R[a][b][k][l] = 1.0 / 0.0;  // trips the static_assert below
        }
  return o;
}
constexpr bool refR2_finite() {
  const RefR2 o = make_refR2();
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++)
      for (int k = 0; k < 2; k++)
        for (int l = 0; l < 2; l++)
          if (!(o.R[a][b][k][l] < 10.0 && o.R[a][b][k][l] > -10.0)) return false;
  return true;
}
static_assert(refR2_finite(), "2-D reference tensor R_ab: closed form disagrees with the Gauss sum");
__device__ const RefR2 d_refR2 = make_refR2();

constexpr int RS2_TX = 64;              // x-consecutive interior nodes of a run
constexpr int RS2_ROW = 18, RS2_NB = 36;  // values of a CSR row / of a node block (2 dof rows)

// ---------------------------------------------------------------------------
// E2 on a verified rectilinear mesh (level 2), autonomous waves as in k_assemble_p3_rect: a row of 64 x-consecutive
// interior nodes is an 18 KB run (2 dof rows x 18 values per node) that one wave owns.  Lane = node, 36
// accumulators; the six record numbers of an element follow from the two coordinate lines and its (E,nu) (read by
// the lanes either side of it one step ahead, all four Gauss points compared bitwise: bad[1] on a mismatch), with
// the operations of k_assemble_rowsweep2d's record wave in the same order.  No record wave, no hand-shake.
// FUSE = true, the one-pass form (KArgs::bout): the wave also integrates the right-hand side of its nodes (E2:325-350,
// 461-529).  The body force fx[4], fy[4] sits in the 64 bytes after E[4], nu[4] of the same 192-byte record, so the
// separate RHS pass and its second sweep over the records disappear: per element row every lane forms, for the
// elements left and right of its node, the Fe entries of the element's lower node on its column (used now) and upper
// node (kept for the next node row), and adds the node's four terms in the reference's element order.
// Measured at 2048^2: the fused kernel is 0.10 ms slower than the plain one and the pass it replaces
// takes 0.15 ms, so the step gains 7 %; selectable (FEMASM_E2_TWO_PASS=1 keeps the two-pass form).
// ---------------------------------------------------------------------------
// COAL (round 3): the records of an element row arrive as ONE contiguous stream -- the 65 records of the wave's elements
// are 12.5 KB of the AoS array, read as 13 non-temporal 16-B-per-lane loads a step ahead, laid out in LDS (the stage area,
// idle at that point of the step; record stride 26 doubles: conflict-free for the lanes' 16-byte reads) and picked up by
// the two lanes either side of each element.  Before, every lane read its two elements' 128 bytes itself: 32 loads per
// lane and step whose 64 lanes touch 64 different records each.  2048^2, one-pass form: 0.414 -> 0.400 ms.  Needs stride
// 24 and a 16-byte aligned array (the launch checks; otherwise COAL = false, the per-lane loads).  Not for K alone, which
// needs a third of a record while the stream reads all of it (0.414 against 0.359 ms).  Either way the kernel takes
// 262-287 registers where its phases need ~170: one wave per SIMD; capped at 256 it spills 20-26 values and runs 0.416 ms.
constexpr int E2W_YC = 32;
constexpr int E2W_RST = 26, E2W_NRAW = (65 * 12 + 63) / 64;  // LDS stride of a record; 16-byte chunks per lane and element row
template <bool FUSE, bool COAL>
__global__ __launch_bounds__(64) void k_assemble_e2_rect(const Slab s, const KArgs a) {
  constexpr int NM = FUSE ? 16 : 8;
  constexpr int NB = RS2_NB, RUN = RS2_TX * NB, END = 24;  // END: values of the two rows of an x-end node (2 x 3 neighbours x 2 x 2)
  __shared__ __attribute__((aligned(16))) double sh_stage[END + RUN + END];
  static_assert(66 * E2W_RST <= END + RUN + END, "the staged records alias the stage");
  const int lane = threadIdx.x;
  const int64_t M = s.dim[0];
  const int64_t i0 = a.x0 + (int64_t)blockIdx.x * RS2_TX;
  const int64_t yc0 = a.s0 + (int64_t)blockIdx.y * a.zc;  // a.zc node rows per wave (E2W_YC; 2 ... 64 measure the same)
  const int64_t yc1 = (yc0 + a.zc < a.s1) ? yc0 + a.zc : a.s1;
  const int ny = (int)(yc1 - yc0);
  const int nvx = (a.x1 - i0 < RS2_TX) ? (int)(a.x1 - i0) : RS2_TX;
  const int64_t i = (lane < nvx) ? i0 + lane : i0 + nvx - 1;  // idle lanes repeat the last node
  // a.ends (E2:877-887: the nodes i = 0 and M-1 are Dirichlet): their rows sit right before and after the interior rows
  // of this node row; the first and last wave of the row write them with their run (unit rows), as k_assemble_p3_rect does
  const int pre = (a.ends && i0 == 1) ? END : 0, post = (a.ends && i0 + nvx == s.dim[0] - 1) ? END : 0;
  const bool soa = !FUSE && a.mat_layout == 3;  // FEMASM_MAT_ELEMENT_SOA (never with FUSE: it carries no loads)
  const bool aos = a.mat_layout == 0 || soa;
  const double xm = a.coords[2 * (i - 1)], xc = a.coords[2 * i], xp = a.coords[2 * (i + 1)];
  const double Jx[2] = {0.5 * (xc - xm), 0.5 * (xp - xc)};
  const double ax[2] = {1.0 / Jx[0], 1.0 / Jx[1]};
  auto yline = [&](int64_t j) {
    j = j < s.cks ? s.cks : (j > a.cke - 1 ? a.cke - 1 : j);
    return a.coords[2 * (M * (j - s.cks)) + 1];
  };
  // COAL: element row L as a stream (request: global -> registers, clamped, never masked; commit: registers -> LDS;
  // fetch: the lane's two elements, doubles 8 .. 8 + NM of their records)
  typedef double e2_d2 __attribute__((ext_vector_type(2)));
  e2_d2 raw[COAL ? E2W_NRAW : 1];
  const int nch = (nvx + 1) * 12;  // 16-byte chunks of the wave's nvx + 1 records
  auto request = [&](int64_t L) {
    const int64_t Lc = L < s.els ? s.els : (L > s.ele - 1 ? s.ele - 1 : L);
    const e2_d2 *rowp = reinterpret_cast<const e2_d2 *>(a.gp + ((i0 - 1) + (M - 1) * (Lc - s.mat_e_ks)) * 24);
#pragma unroll
    for (int q = 0; q < E2W_NRAW; q++) {
      const int c = lane + 64 * q;
      raw[q] = __builtin_nontemporal_load(rowp + (c < nch ? c : nch - 1));
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int q = 0; q < E2W_NRAW; q++) {
      const int c = lane + 64 * q, r = (c * 5462) >> 16, o = c - 12 * r;  // r = c / 12 (c < 832)
      if (c < nch) *reinterpret_cast<e2_d2 *>(sh_stage + r * E2W_RST + 2 * o) = raw[q];
    }
  };
  auto fetch = [&](double (*m)[NM]) {
    const int li = (lane < nvx) ? lane : nvx - 1;
#pragma unroll
    for (int ox = 0; ox < 2; ox++) {
      const e2_d2 *rp = reinterpret_cast<const e2_d2 *>(sh_stage + (li + ox) * E2W_RST + 8);
#pragma unroll
      for (int q = 0; q < NM / 2; q++) {
        const e2_d2 v = rp[q];
        m[ox][2 * q] = v.x;
        m[ox][2 * q + 1] = v.y;
      }
    }
  };
  auto lds_sync = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  // E[4], nu[4] of the elements i-1, i of element row L (clamped to the rows this rank holds: never masked)
  auto load_mat = [&](int64_t L, double (*m)[NM]) {
    if (!aos) return;
    const int64_t Lc = L < s.els ? s.els : (L > s.ele - 1 ? s.ele - 1 : L);
#pragma unroll
    for (int ox = 0; ox < 2; ox++) {
      if (soa) {
        const int64_t eidx = (i - 1 + ox) + (M - 1) * (Lc - s.mat_e_ks);
        const double vE = a.gp[eidx], vNu = a.gp[a.mat_stride + eidx];
#pragma unroll
        for (int q = 0; q < 4; q++) m[ox][q] = vE, m[ox][4 + q] = vNu;
        continue;
      }
      const double *mq = a.gp + ((i - 1 + ox) + (M - 1) * (Lc - s.mat_e_ks)) * a.mat_stride + 8;
#pragma unroll
      for (int q = 0; q < NM; q++) m[ox][q] = mq[q];
    }
  };
  // Fe entries (E2:339-349: Fe[2n+c] += (fac N_n) f_c, Gauss points in order) of the element ox of row L at its lower
  // node on this lane's column (ox = 0: local node 3, ox = 1: node 0) and at its upper node (ox = 0: node 2, ox = 1: node 1)
  auto fe_terms = [&](int64_t L, double ylo, double yhi, const double (*m)[NM], double (*lo)[2], double (*up)[2]) {
    const bool rowok = aos && L >= s.els && L < s.ele;
    const double J11 = rowok ? 0.5 * (yhi - ylo) : 0.0;
#pragma unroll
    for (int ox = 0; ox < 2; ox++) {
      const double fac = 1.0 * (Jx[ox] * J11);
      const int nlo = ox ? 0 : 3, nup = ox ? 1 : 2;
#pragma unroll
      for (int c = 0; c < 2; c++) {
        double vl = 0.0, vu = 0.0;
#pragma unroll
        for (int pp = 0; pp < 4; pp++) {
          vl += fac * c_tab2.N[pp][nlo] * m[ox][(NM - 8) + 4 * c + pp];
          vu += fac * c_tab2.N[pp][nup] * m[ox][(NM - 8) + 4 * c + pp];
        }
        lo[ox][c] = rowok ? vl : 0.0;
        up[ox][c] = rowok ? vu : 0.0;
      }
    }
  };
  // records (pd0, pd1, pe0, pe1, po, pq) of the elements i-1, i of element row L
  // (ylo, yhi = the y line at node rows L and L+1: requested a step ahead like the material, so that no load of the
  // step's own sits on its critical path)
  auto records = [&](int64_t L, double ylo, double yhi, const double (*m)[NM], double (*rec)[6]) {
    const bool rowok = L >= s.els && L < s.ele;
    const double J11 = rowok ? 0.5 * (yhi - ylo) : 1.0;
    const double ay = 1.0 / J11;
#pragma unroll
    for (int ox = 0; ox < 2; ox++) {
      double E = a.E, nu = a.nu;
      bool ok = true;
      if (aos) {
        E = m[ox][0];
        nu = m[ox][4];
#pragma unroll
        for (int q = 1; q < 4; q++) ok = ok && m[ox][q] == E && m[ox][4 + q] == nu;
      }
      if (rowok && !ok) atomicAdd(a.bad + 1, 1ULL);
      const double wJ = 1.0 * (Jx[ox] * J11);
      if (rowok && !(wJ > 0.0)) atomicAdd(a.bad, 1ULL);
      double d11, d12, d44;
      scaled_coefs<KIND_PLANE>(E, nu, wJ, d11, d12, d44);
      rec[ox][0] = rowok ? d11 * (ax[ox] * ax[ox]) : 0.0;
      rec[ox][1] = rowok ? d11 * (ay * ay) : 0.0;
      rec[ox][2] = rowok ? d44 * (ax[ox] * ax[ox]) : 0.0;
      rec[ox][3] = rowok ? d44 * (ay * ay) : 0.0;
      rec[ox][4] = rowok ? d12 * (ax[ox] * ay) : 0.0;
      rec[ox][5] = rowok ? d44 * (ax[ox] * ay) : 0.0;
    }
  };
  double mcur[2][NM], mnext[2][NM], rlo[2][6], rhi[2][6];
  double fup[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, flo[2][2];  // Fe of the element row below at its upper nodes; scratch
#pragma unroll
  for (int ox = 0; ox < 2; ox++)
#pragma unroll
    for (int q = 0; q < NM; q++) mcur[ox][q] = mnext[ox][q] = 0.0;
  if constexpr (COAL) {
    request(yc0 - 1);
    commit();
    lds_sync();
    fetch(mcur);
    lds_sync();
    request(yc0);
  } else {
    load_mat(yc0 - 1, mcur);
  }
  double y0 = yline(yc0 - 1), y1 = yline(yc0), y2 = yline(yc0 + 1);  // rolling window of the y line: rows L, L+1, L+2
  records(yc0 - 1, y0, y1, mcur, rlo);
  if constexpr (FUSE) fe_terms(yc0 - 1, y0, y1, mcur, flo, fup);
  if constexpr (!COAL) load_mat(yc0, mnext);
  for (int st = 0; st < ny; st++) {
    const int64_t jn = yc0 + st;
    if constexpr (COAL) {
      lds_sync();  // the previous step's copy-out has read the stage
      commit();    // element row jn (requested a step ago)
      lds_sync();
      fetch(mcur);
      lds_sync();
      request(jn + 1);  // one step ahead: issued before this step's stores
    } else {
#pragma unroll
      for (int ox = 0; ox < 2; ox++)
#pragma unroll
        for (int q = 0; q < NM; q++) mcur[ox][q] = mnext[ox][q];
      load_mat(jn + 1, mnext);  // one step ahead: issued before this step's stores
    }
    y0 = y1;
    y1 = y2;
    y2 = yline(jn + 2);
    records(jn, y0, y1, mcur, rhi);
    if constexpr (FUSE) {
      double nup[2][2];
      fe_terms(jn, y0, y1, mcur, flo, nup);
      double bv[2];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        double v = 0.0 + fup[0][c];   // element (i-1, j-1), its node (1,1)
        v = v + fup[1][c];            // element (i,   j-1), its node (0,1)
        v = v + flo[0][c];            // element (i-1, j  ), its node (1,0)
        v = v + flo[1][c];            // element (i,   j  ), its node (0,0)
        bv[c] = v;
        fup[0][c] = nup[0][c];
        fup[1][c] = nup[1][c];
      }
      double *bo = a.bout + 2 * (M * (jn - s.ks));
#if FEMASM_EXP == 5  // profiling build: the one-pass kernel without its b stores
      asm volatile("" ::"v"(bv[0]), "v"(bv[1]));
#else
      if (lane < nvx) *reinterpret_cast<double2 *>(bo + 2 * i) = make_double2(bv[0], bv[1]);
#endif
      // the Dirichlet columns i = 0, M-1 of this node row (E2:882-885: ux = +1 / -1, uy = 0), with the first / last run
      if (pre && lane == 0) *reinterpret_cast<double2 *>(bo) = make_double2(1.0, 0.0);
      if (post && lane == 0) *reinterpret_cast<double2 *>(bo + 2 * (M - 1)) = make_double2(-1.0, 0.0);
    }
    double acc[2][RS2_ROW];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int q = 0; q < RS2_ROW; q++) acc[c][q] = 0.0;
#pragma unroll
    for (int oy = -1; oy <= 0; oy++)
#pragma unroll
      for (int ox = -1; ox <= 0; ox++) {
        const double *r = oy ? rlo[ox + 1] : rhi[ox + 1];
        const double pd[2] = {r[0], r[1]}, pe[2] = {r[2], r[3]};
        const double po = r[4], pq = r[5];
        const int ia = quad_idx(-ox, -oy);
#pragma unroll
        for (int by = 0; by <= 1; by++)
#pragma unroll
          for (int bx = 0; bx <= 1; bx++) {
            const double(*R)[2] = d_refR2.R[ia][quad_idx(bx, by)];
            const int t = (oy + by + 1) * 3 + (ox + bx + 1);
#pragma unroll
            for (int C = 0; C < 2; C++) {
              acc[C][2 * t + C] = fma(R[C][C], pd[C], fma(R[1 - C][1 - C], pe[1 - C], acc[C][2 * t + C]));
              acc[C][2 * t + 1 - C] = fma(R[C][1 - C], po, fma(R[1 - C][C], pq, acc[C][2 * t + 1 - C]));
            }
          }
      }
#pragma unroll
    for (int ox = 0; ox < 2; ox++)
#pragma unroll
      for (int q = 0; q < 6; q++) rlo[ox][q] = rhi[ox][q];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();  // the previous step's reads of the stage are done
    {
      double2 *dst = reinterpret_cast<double2 *>(sh_stage + pre + lane * NB);
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int q = 0; q < RS2_ROW / 2; q++) dst[c * (RS2_ROW / 2) + q] = make_double2(acc[c][2 * q], acc[c][2 * q + 1]);
    }
    if (pre + post) {
      // row c of an end node: 12 values, 1.0 at its own position (neighbour (dj, di) = (0, 0): index 2 resp. 3 of 6, times dof)
      const int c = lane / 12, r = lane - 12 * c;
      if (pre && lane < END) sh_stage[lane] = (r == 4 + c) ? 1.0 : 0.0;
      if (post && lane < END) sh_stage[pre + nvx * NB + lane] = (r == 6 + c) ? 1.0 : 0.0;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int n = pre + nvx * NB + post;
    double *runp = a.out + (row_start(s, i0, jn, 0, 0) - a.base_off) - pre;
    const int head = (int)((reinterpret_cast<uintptr_t>(runp) & 127) >> 3);
    const int nchunk = (head + n + 1) >> 1;
    constexpr int NIT = ((RUN + 2 * END) / 2 + 8 + 63) / 64;
    double2 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int s0 = 2 * (lane + 64 * it) - head;
      const int sa = s0 < 0 ? 0 : (s0 < n ? s0 : n - 1), sb = s0 + 1 < 0 ? 0 : (s0 + 1 < n ? s0 + 1 : n - 1);
      v[it] = make_double2(sh_stage[sa], sh_stage[sb]);
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int q = lane + 64 * it, s0 = 2 * q - head;
      if (q < nchunk) {
        if (s0 >= 0 && s0 + 1 < n) store_pair(runp + s0, v[it]);
        else if (s0 == -1) runp[0] = v[it].y;
        else if (s0 == n - 1) runp[s0] = v[it].x;
      }
    }
  }
}


// ---------------------------------------------------------------------------
// E2 on general quadrilaterals (level 0), element matrices through LDS: k_assemble_ke_e2 -- the 2-D sibling of
// k_assemble_ke_p3.  The brick kernel forms every 2 x 2 block of an element in the rows of both of its nodes and
// rebuilds the Gauss-point records of an element for each of its four nodes (~2 400 instructions per node row); here a
// thread forms ONE element: its four Gauss-point records and the ten unique node-pair blocks (a <= b) of its element
// matrix -- 36 numbers in LDS: a diagonal block (a, a) is symmetric, its K10 is stored as K01 (the two differ by the
// rounding of (d12 g0) g1 against (d12 g1) g0 where each is formed on its own) -- and the node rows pick their blocks
// (transposed for a > b) out of LDS.
//   tile      15 x 7 interior nodes = 16 x 8 = 128 elements, 128 threads; thread = element, then thread = node
//   stores    the tile's rows are 15-node runs of the CSR array (36 values per node): staged in HBM order in the
//             element-matrix area (dead by then) and written as 16 B per thread from the 128-byte boundary below
// Same arithmetic per Gauss point as the brick kernel (jac2, scaled_coefs, add_block); sums are formed per element
// and then over the elements -- the reference's own order -- so the last bit can differ from the brick kernel.
// ---------------------------------------------------------------------------
constexpr int KE2_TX = 15, KE2_TY = 7, KE2_NK = 36, KE2_KS = 37;  // 37: odd stride, and four workgroups fit a CU's LDS
struct Ke2Pairs { int off[4][4]; };  // first of the 4 numbers of block (a, b), a < b; of the 3 (K00, K01, K11) of block (a, a)
constexpr Ke2Pairs make_ke2_pairs() {
  Ke2Pairs o{};
  int n = 0;
  for (int a = 0; a < 4; a++)
    for (int b = a; b < 4; b++) {
      o.off[a][b] = o.off[b][a] = n;
      n += (a == b) ? 3 : 4;
    }
  return o;
}
static_assert(make_ke2_pairs().off[3][3] + 3 == KE2_NK, "element matrix storage");
__device__ const Ke2Pairs KE2_PAIRS = make_ke2_pairs();

__global__ __launch_bounds__(128) void k_assemble_ke_e2(const Slab s, const KArgs a) {
  constexpr int XS = KE2_TX + 2, YS = KE2_TY + 2, NB = RS2_NB, RUN = KE2_TX * NB;
  __shared__ double sh_x[XS * YS * 2];
  __shared__ __attribute__((aligned(16))) double sh_ke[128 * KE2_KS];  // element matrices; then the tile's rows in HBM order
  static_assert(KE2_TY * RUN <= 128 * KE2_KS, "the staged rows alias the element matrices");
  const int tid = threadIdx.x;
  const int lx = tid & 15, ly = tid >> 4;  // element (ex, ey) of the halo'd tile / node of the tile (lx < 15, ly < 7)
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t i0 = a.x0 + (int64_t)blockIdx.x * KE2_TX, j0 = a.s0 + (int64_t)blockIdx.y * KE2_TY;
  const int nvx = (a.x1 - i0 < KE2_TX) ? (int)(a.x1 - i0) : KE2_TX;
  const int nvy = (a.s1 - j0 < KE2_TY) ? (int)(a.s1 - j0) : KE2_TY;
  for (int q = tid; q < XS * YS; q += 128) {  // node coordinates of the halo'd tile (clamped reads)
    const int cy = q / XS, cx = q - XS * cy;
    int64_t gi = i0 - 1 + cx, gj = j0 - 1 + cy;
    gi = gi < 0 ? 0 : (gi > M - 1 ? M - 1 : gi);
    gj = gj < s.cks ? s.cks : (gj > a.cke - 1 ? a.cke - 1 : gj);
    const int64_t node = gi + M * (gj - s.cks);
    sh_x[2 * q] = a.coords[2 * node];
    sh_x[2 * q + 1] = a.coords[2 * node + 1];
  }
  __syncthreads();
  {  // thread = element: records of the four Gauss points, the ten blocks (a <= b) of the element matrix
    const int64_t gex = i0 - 1 + lx, gey = j0 - 1 + ly;
    const bool eok = gex >= 0 && gex < M - 1 && gey >= 0 && gey < N - 1 && gey >= s.els && gey < s.ele;
    double ke[KE2_NK];
#pragma unroll
    for (int q = 0; q < KE2_NK; q++) ke[q] = 0.0;
    if (eok) {
      double xc[8];
#pragma unroll
      for (int n = 0; n < 4; n++) {
        const int cn = (lx + QDI[n]) + XS * (ly + QDJ[n]);
        xc[2 * n] = sh_x[2 * cn];
        xc[2 * n + 1] = sh_x[2 * cn + 1];
      }
      const int64_t emat = gex + (M - 1) * (gey - s.mat_e_ks);
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const double(*G)[4] = c_tab2.G[p];
        const Jac2 jj = jac2(G, xc);
        double r[8];
#pragma unroll
        for (int n = 0; n < 4; n++) {
          r[2 * n] = G[0][n] * jj.i00 + G[1][n] * jj.i01;
          r[2 * n + 1] = G[0][n] * jj.i10 + G[1][n] * jj.i11;
        }
        const double wJ = 1.0 * jj.det;
        if (!(wJ > 0.0)) atomicAdd(a.bad, 1ULL);
        double E = a.E, nu = a.nu;
        if (a.mat_layout == 0 || a.mat_layout == 3) {  // AoS record, or the planes of FEMASM_MAT_ELEMENT_SOA
          const bool rec = a.mat_layout == 0;
          const double *q = a.gp + emat * (rec ? a.mat_stride : 1);
          E = q[rec ? 8 + p : 0];
          nu = q[rec ? 12 + p : a.mat_stride];
        }
        double d11, d12, d44;
        scaled_coefs<KIND_PLANE>(E, nu, wJ, d11, d12, d44);
#pragma unroll
        for (int na = 0; na < 4; na++) {
          const double ga[2] = {r[2 * na], r[2 * na + 1]};
          const double l[2] = {d12 * ga[0], d12 * ga[1]}, m[2] = {d44 * ga[0], d44 * ga[1]}, q[2] = {d11 * ga[0], d11 * ga[1]};
          {  // block (a, a): the entries (0,0), (0,1), (1,1) of add_block
            double *kd = ke + KE2_PAIRS.off[na][na];
            kd[0] = fma(m[1], ga[1], fma(q[0], ga[0], kd[0]));
            kd[1] = fma(m[1], ga[0], fma(l[0], ga[1], kd[1]));
            kd[2] = fma(m[0], ga[0], fma(q[1], ga[1], kd[2]));
          }
#pragma unroll
          for (int nb = na + 1; nb < 4; nb++) {
            const double gb[2] = {r[2 * nb], r[2 * nb + 1]};
            add_block<KIND_PLANE, 2>(ke + KE2_PAIRS.off[na][nb], l, m, q, gb);
          }
        }
      }
    }
    double *dst = sh_ke + tid * KE2_KS;
#pragma unroll
    for (int q = 0; q < KE2_NK; q++) dst[q] = ke[q];
  }
  __syncthreads();  // element matrices complete
  double acc[2][RS2_ROW];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int q = 0; q < RS2_ROW; q++) acc[c][q] = 0.0;
  const bool mynode = lx < nvx && ly < nvy;
  {
    const int glx = mynode ? lx : 0, gly = mynode ? ly : 0;
#pragma unroll
    for (int oy = -1; oy <= 0; oy++)
#pragma unroll
      for (int ox = -1; ox <= 0; ox++) {
        const double *kp = sh_ke + ((glx + 1 + ox) + 16 * (gly + 1 + oy)) * KE2_KS;
        const int ia = quad_idx(-ox, -oy);
#pragma unroll
        for (int by = 0; by <= 1; by++)
#pragma unroll
          for (int bx = 0; bx <= 1; bx++) {
            const int ib = quad_idx(bx, by);
            const int t = (oy + by + 1) * 3 + (ox + bx + 1);
            const double *blk = kp + KE2_PAIRS.off[ia][ib];
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
              for (int cc = 0; cc < 2; cc++) {  // block (a, b) is stored for a < b, (b, a) is its transpose; (a, a): K00 K01 K11
                const double v = (ia == ib) ? blk[c + cc] : (ia < ib ? blk[2 * c + cc] : blk[2 * cc + c]);
                acc[c][2 * t + cc] = acc[c][2 * t + cc] + v;
              }
          }
      }
  }
  __syncthreads();  // every thread has read its element matrices: the staged rows alias them
  if (mynode) {
    double2 *dst = reinterpret_cast<double2 *>(sh_ke + ly * RUN + lx * NB);
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int q = 0; q < RS2_ROW / 2; q++) dst[c * (RS2_ROW / 2) + q] = make_double2(acc[c][2 * q], acc[c][2 * q + 1]);
  }
  __syncthreads();
  const int n = nvx * NB;
  for (int rr = 0; rr < nvy; rr++) {  // one 15-node run per tile row, 16 B per thread from the 128-byte boundary below it
    const double *stage = sh_ke + rr * RUN;
    double *runp = a.out + (row_start(s, i0, j0 + rr, 0, 0) - a.base_off);
    const int head = (int)((reinterpret_cast<uintptr_t>(runp) & 127) >> 3);
    const int nchunk = (head + n + 1) >> 1;
    constexpr int NIT = (RUN / 2 + 8 + 127) / 128;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int q = tid + 128 * it, s0 = 2 * q - head;
      if (q < nchunk) {
        const int sa = s0 < 0 ? 0 : s0, sb = s0 + 1 < n ? s0 + 1 : n - 1;
        const double2 v = make_double2(stage[sa], stage[sb]);
        if (s0 >= 0 && s0 + 1 < n) store_pair(runp + s0, v);
        else if (s0 == -1) runp[0] = v.y;
        else if (s0 == n - 1) runp[s0] = v.x;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Dirichlet rows (A10/A11): one workgroup per x-row of nodes (j,k), one wave per Dirichlet node at a time; the
// lanes write the node's dof rows (contiguous in the CSR array) as consecutive values.  Rows that are not on a
// y/z face only have their two end nodes to do.
// Row = zeros with the diagonal 1 (Poisson: number of assembled elements sharing the node);
// interface pass: zero rows (the receiver ignores them).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bc_rows(const Slab s, const KArgs a) {
  if (a.cond && a.bad[1] == 0) return;
  const int64_t M = s.dim[0], N = s.dim[1];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // x-rows of nodes (j, k) are walked with a stride (most rows have two Dirichlet nodes only); S3 128^3: 45 us =
  // 4 TB/s of stores; P3 256^3: instruction-bound (splitting the face rows into 32-node segments over more
  // workgroups made it 5x slower: the per-unit index arithmetic dominates)
  const int64_t nrow = (s.nsd == 3) ? N * (a.s1 - a.s0) : (a.s1 - a.s0);
  for (uint32_t row = blockIdx.x; row < (uint32_t)nrow; row += gridDim.x) {  // (nrow < 2^31: checked by the host)
  int64_t j, k;
  if (s.nsd == 3) { const uint32_t kq = row / (uint32_t)N; j = row - kq * (uint32_t)N; k = a.s0 + kq; } else { j = a.s0 + row; k = 0; }
  const bool full = (M < 3) || is_bc_node(s, a.bc, 1, j, k);  // is an x-interior node of this row Dirichlet?
  if (!full && a.ends) continue;  // the two end nodes of this line were written by the interior kernel (k_assemble_p3_rect)
  // per line: neighbour counts, the node's own position among them and the Poisson valence in y and z, start of the line
  const int cy = span1(j, N), cz = span1(k, s.dim[2]), cyz = cy * cz;
  const int selfyz = (s.nsd == 3 ? (k > 0) : 0) * cy + (j > 0);
  const int vyz = ((j > 0) + (j < N - 1)) * ((k - 1 >= s.els && k - 1 < s.ele) + (k >= s.els && k < s.ele));
  double *line = a.out + (row_start(s, 0, j, k, 0) - a.base_off);
  // dof 1 (27 values per node): each half-wave takes a node of its own; otherwise one wave per node
  const int sub = (s.dof == 1) ? (lane >> 5) : 0, lw = (s.dof == 1) ? 32 : 64, ll = lane & (lw - 1);
  const int slot = (s.dof == 1) ? 2 * w + sub : w, nslot = (s.dof == 1) ? 8 : 4;
  const int64_t istep = full ? nslot : M;
  for (int64_t i = full ? slot : (slot == 0 ? 0 : ((slot == 1 && M > 1) ? M - 1 : M)); i < M; i += istep) {
    if (!is_bc_node(s, a.bc, i, j, k)) continue;
    double diag = a.iface ? 0.0 : 1.0;
    if (a.bc == BC_POISSON) diag = (double)(((i > 0) + (i < M - 1)) * vyz);
    const int cx = span1(i, M);
    const int cnt = cx * cyz * s.dof;
    // position of the node itself among its neighbours
    const int self = (selfyz * cx + (i > 0)) * s.dof;
    // the blocks of a line's nodes follow each other: dof^2 * cy * cz values per x-neighbour pair before node i
    double *blk = line + (int64_t)(s.dof * s.dof * cyz) * pre1(i, M);
    int c = 0, r = ll;  // value t = c * cnt + r of the node's block
    for (int t = ll; t < s.dof * cnt; t += lw) {
      while (r >= cnt) { r -= cnt; c++; }
      blk[t] = (r == self + c) ? diag : 0.0;
      r += lw;
    }
  }
  }
}

}  // namespace femasm
