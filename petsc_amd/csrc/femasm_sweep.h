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

template <int KIND>
__global__ __launch_bounds__(256, 2) void k_assemble_sweep(const Slab s, const KArgs a) {
  using T = Traits<KIND>;
  constexpr int DOF = T::DOF, DD = DOF * DOF, NACC = 9 * DD;
  constexpr int NPAIR = 14;
  constexpr int CH = 9 * DOF;              // doubles of one (node,row) chunk of a third
  constexpr int NODESTRIDE = 27 * DD;      // doubles between the rows of x-neighbouring nodes
  constexpr int NIT = (SW_TX * CH + 63) / 64;
  static_assert(3 * 64 * CH * 8 <= 2 * NPAIR * SW_NS * 16, "staging aliases the record buffers");

  __shared__ double2 sh_rec[2 * NPAIR * SW_NS];   // 53 760 B
  __shared__ double sh_x[3 * 108 * 3];             //  7 776 B: ring of three node layers (18x6 nodes each)

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

  // output: lane/iteration-invariant scatter offsets of the transposed store
  int goff[NIT];
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int idx = lane + 64 * it;
    goff[it] = idx + (NODESTRIDE - CH) * (idx / CH);
  }
  const int nvx = (a.x1 - i0 < SW_TX) ? (int)(a.x1 - i0) : SW_TX;  // valid nodes of the tile in x
  // three waves write per step: A (wave 0), D (wave 3) and whichever of waves 1/2 finishes a middle third
  double *stage = reinterpret_cast<double *>(sh_rec) + ((w == 0) ? 0 : (w == 3 ? 1 : 2)) * (64 * CH);

  double acc[NACC];

  // ---- coordinate ring: node layer zc0-1+m lives in ring slot m % 3.  Threads 128..235 (waves 2,3: the
  //      waves without quadrature duty) own one node of the 18x6 halo'd tile each. ----
  const int ctid = tid - 128;
  const bool cthread = ctid >= 0 && ctid < 108;
  int64_t cnode_xy = 0;
  if (cthread) {
    const int cy = ctid / 18, cx = ctid - 18 * cy;
    int64_t gi = i0 - 1 + cx, gj = j0 - 1 + cy;
    gi = gi < 0 ? 0 : (gi > M - 1 ? M - 1 : gi);
    gj = gj < 0 ? 0 : (gj > N - 1 ? N - 1 : gj);
    cnode_xy = gi + M * gj;
  }
  auto coord_ptr = [&](int64_t gk) {  // clamped: elements that would use a clamped layer are masked out
    gk = gk < s.cks ? s.cks : (gk > a.cke - 1 ? a.cke - 1 : gk);
    return a.coords + 3 * (cnode_xy + M * N * (gk - s.cks));
  };
  if (cthread) {
#pragma unroll
    for (int m = 0; m < 2; m++) {
      const double *src = coord_ptr(zc0 - 1 + m);
#pragma unroll
      for (int d = 0; d < 3; d++) sh_x[m * 324 + 3 * ctid + d] = src[d];
    }
  }

  // ---- material prefetch: E, nu of the NEXT quadrature call are loaded one call ahead, so no global
  //      load latency sits on the per-Gauss-point critical path ----
  const bool mat_aos = (KIND != KIND_POISSON) && a.mat_layout == 0;
  const int64_t mat_layer = (M - 1) * (N - 1) * (int64_t)a.mat_stride;
  const double *matq = a.gp + (gex + (M - 1) * (gey + (N - 1) * (zc0 - 1 - s.mat_e_ks))) * a.mat_stride;
  auto layer_ok = [&](int64_t L) { return exy_ok && L >= s.els && L < s.ele; };
  double preE = a.E, preNu = a.nu;
  if (mat_aos && layer_ok(zc0 - 1)) { preE = matq[96]; preNu = matq[104]; }

  for (int st = 0; st < nsteps; st++) {
    const int64_t L = zc0 - 1 + st;  // element layer of this step
    const bool eok = layer_ok(L);
    const int xb0 = (st % 3) * 324, xb1 = ((st + 1) % 3) * 324;  // ring slots of node layers L, L+1
    __syncthreads();  // ring slots of this step are filled; the previous step's staged output has been read

    // next step's new node layer: global loads issued now, parked in registers, stored to the ring late
    double cnext[3] = {0.0, 0.0, 0.0};
    const bool cfetch = cthread && st + 1 < nsteps;
    if (cfetch) {
      const double *src = coord_ptr(L + 2);
#pragma unroll
      for (int d = 0; d < 3; d++) cnext[d] = src[d];
    }

    // One record, streamed: coordinates are read from LDS inside the J loop and every gradient pair is
    // stored as soon as it exists, so no 24-double coordinate or 28-double record array stays live next
    // to the 81 accumulators.
    auto prefetch_material = [&](int p) {  // for the quadrature call that follows call p
      if (!mat_aos) return;
      if (p < 7) {
        if (eok) { preE = matq[96 + p + 1]; preNu = matq[104 + p + 1]; }
      } else if (st + 1 < nsteps && layer_ok(L + 1)) {
        preE = matq[mat_layer + 96];
        preNu = matq[mat_layer + 104];
      }
    };
    auto geometry = [&](int p, double2 *recbuf) {
      if (!gthread) return;
      double2 *dst = recbuf + tid;
      if (!eok) {
#pragma unroll
        for (int q = 0; q < NPAIR; q++) dst[q * SW_NS] = make_double2(0.0, 0.0);
        prefetch_material(p);
        return;
      }
      const double(*G)[8] = c_tab3.G[p];
      double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
      for (int n = 0; n < 8; n++) {
        const double *xn = sh_x + (HDK[n] ? xb1 : xb0) + 3 * ((ex + HDI[n]) + 18 * (ey + HDJ[n]));
        const double x0 = xn[0], x1 = xn[1], x2 = xn[2];
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
          J[rr][0] = J[rr][0] + G[rr][n] * x0;
          J[rr][1] = J[rr][1] + G[rr][n] * x1;
          J[rr][2] = J[rr][2] + G[rr][n] * x2;
        }
      }
      double iJ[3][3];
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
      const double det = J[0][0] * J[1][1] * J[2][2] - J[0][0] * J[1][2] * J[2][1] - J[1][0] * J[0][1] * J[2][2] +
                         J[1][0] * J[0][2] * J[2][1] + J[2][0] * J[0][1] * J[1][2] - J[2][0] * J[0][2] * J[1][1];
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
        double d11, d12, d44;
        scaled_coefs<KIND>(preE, preNu, wJ, d11, d12, d44);
        dst[12 * SW_NS] = make_double2(d11, d12);
        dst[13 * SW_NS] = make_double2(d44, 0.0);
      }
      prefetch_material(p);
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
      if (p == 5 && cfetch) {  // ring slot of layer L+2 = slot of layer L-1: last read in the previous step
#pragma unroll
        for (int d = 0; d < 3; d++) sh_x[((st + 2) % 3) * 324 + 3 * ctid + d] = cnext[d];
      }
      __syncthreads();
    }
    matq += mat_layer;

    // ---- output: roles A (dz=-1), finish (dz=0), D (dz=+1) write one third of their node layer ----
    if (active && role != 1) {
      const int dz = (role == 0) ? -1 : (role == 2 ? 0 : 1);
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
#elif FEMASM_EXP == 4
            if (idx < nvx * CH && lane == 77) base[goff[it]] = src[idx];
#else
            if (idx < nvx * CH) base[goff[it]] = src[idx];
#endif
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    // the next step's first __syncthreads orders these LDS reads before the records are rebuilt
  }
}

// ---------------------------------------------------------------------------
// Dirichlet rows (A10/A11): one thread per node of the processed layers; non-Dirichlet nodes exit.
// Row = zeros with the diagonal 1 (Poisson: number of assembled elements sharing the node);
// interface pass: zero rows (the receiver ignores them).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bc_rows(const Slab s, const KArgs a) {
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  int64_t j, k;
  if (s.nsd == 3) { j = blockIdx.y; k = a.s0 + blockIdx.z; } else { j = a.s0 + blockIdx.y; k = 0; }
  if (!is_bc_node(s, a.bc, i, j, k)) return;
  double diag = a.iface ? 0.0 : 1.0;
  if (a.bc == BC_POISSON) {
    const int vx = (i > 0) + (i < M - 1), vy = (j > 0) + (j < N - 1);
    const int vz = (k - 1 >= s.els && k - 1 < s.ele) + (k >= s.els && k < s.ele);
    diag = (double)(vx * vy * vz);
  }
  const int cx = span1(i, M), cy = span1(j, N), cz = span1(k, s.dim[2]);
  const int cnt = cx * cy * cz * s.dof;
  // position of the node itself among its neighbours
  const int self = (((s.nsd == 3 ? (k > 0) : 0) * cy + (j > 0)) * cx + (i > 0)) * s.dof;
  for (int c = 0; c < s.dof; c++) {
    double *row = a.out + (row_start(s, i, j, k, c) - a.base_off);
    for (int t = 0; t < cnt; t++) row[t] = (t == self + c) ? diag : 0.0;
  }
}

}  // namespace femasm
