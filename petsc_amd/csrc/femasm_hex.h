// femasm_hex.h -- S3/R3 stiffness of interior nodes on ANY hexahedra with ONE (E, nu) per element: kernel level 1
// of the elastic kind (north_star (2) + (3): Jacobian, inverse and det per Gauss point in FP64 registers, S3:186-227;
// B^T D B, S3:450-518).  Round 3; DESIGN.md section 4.2c has the derivation and the measurements.
//
// Why it is not k_assemble_sweep<ELASTIC> made faster: that kernel holds a third of a node's rows per lane (81 FP64
// accumulators = 162 registers: two waves per SIMD at most), applies the material at every Gauss point (21 FMAs per
// node pair and Gauss point) and meets at nine s_barriers per element layer.  Here
//  * the material leaves the Gauss-point loop.  With one (E, nu) per element (tested bitwise on the records, flag
//    bad[1] -> the general kernel, like every fast path of this library)
//        K_ab[i][j] = c12 W_ab[i][j] + c44 W_ab[j][i]   (i != j)          W_ab[i][j] = sum_p wJ_p ga_i(p) gb_j(p)
//        K_ab[i][i] = c11 W_ab[i][i] + c44 (W_ab[j][j] + W_ab[k][k])
//    so a node pair costs 9 FMAs per Gauss point and 21 once per element;
//  * lane = (node, role, element): role = which z-third of the node's rows and which element layer feeds it (the four
//    roles of k_assemble_sweep: A, start, finish, D), element = one of the four elements (ox, oy) of that layer around
//    the node.  36 accumulators (four node pairs) per lane: three waves per SIMD.  Roles differ only in LDS addresses:
//    one instruction stream for all lanes;
//  * the four lanes of a (node, role) are a quad: their 3x3 blocks meet in the same neighbours and are added by two
//    quad-permute steps (a fixed tree: deterministic), then every lane keeps a disjoint 22 of the quad's 81 numbers;
//  * a step = one element layer L.  All 8 x 35 Gauss-point records of the halo'd 7 x 5 element tile are built in ONE
//    phase by 280 of the 384 lanes (lane = (Gauss point, element)), so a step has four barriers, not nine, and no wave
//    waits for a record wave;
//  * what a step leaves for the next one (the dz = -1 third and the lower half of the dz = 0 third of node layer L+1)
//    stays in those lanes' registers (22 numbers), so the staging area of the finished node layer can alias the records;
//  * the finished node layer leaves as four contiguous runs (6 nodes x 243 values), 16 B per lane from the 128-byte
//    boundary below each run: whole lines only (partial lines cost four writes each).
// LDS 74.8 KB: two workgroups (12 waves) per CU.
#pragma once
#include "femasm_sweep.h"

namespace femasm {

constexpr int HX_TX = 6, HX_TY = 4, HX_NN = HX_TX * HX_TY;  // node tile
constexpr int HX_SY = 7, HX_NS = 35;                        // element tile 7 x 5 (halo of one), slot = ex + 7 ey
constexpr int HX_NX = 8, HX_NL = 48;                        // halo'd node tile 8 x 6 of one layer
constexpr int HX_NT = 384;
constexpr int HX_NREC = 8 * HX_NS;                          // record lanes: (Gauss point, element)
constexpr int HX_CELLS = 16;                                // record = 16 cells of 16 B: node n -> (gx, gy), (gz, wJ)
constexpr int HX_NB = 243, HX_RUN = HX_TX * HX_NB;          // values of a node block, of a tile-row run
constexpr int HX_ZC = 42;                                   // default node layers per workgroup (KArgs::zc overrides)

__device__ inline int hx_quad(int bx, int by) { return bx == 0 ? by : 3 - by; }  // = quad_idx, branch-free

__global__ __launch_bounds__(HX_NT, 3) void k_assemble_s3_hex(const Slab s, const KArgs a) {
  if (a.cond && a.bad[1] == 0) return;
  __shared__ __attribute__((aligned(16))) double2 sh_ring[8 * HX_CELLS * HX_NS];  // 71 680 B; the stage aliases it
  __shared__ double sh_x[2][HX_NL * 3];                                           // node layers L, L+1 (ring of two)
  __shared__ double sh_c[HX_NS * 3];                                              // (c11, c12, c44) of the layer's elements
  double *sh_stage = reinterpret_cast<double *>(sh_ring);
  static_assert(HX_NN * HX_NB * 8 <= 8 * HX_CELLS * HX_NS * 16, "stage fits the ring");

  const int t = threadIdx.x;
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t i0 = a.x0 + (int64_t)blockIdx.x * HX_TX, j0 = a.y0 + (int64_t)blockIdx.y * HX_TY;
  const int ZC = a.zc > 0 ? a.zc : HX_ZC;
  const int64_t zc0 = a.s0 + (int64_t)blockIdx.z * ZC;
  const int64_t zc1 = (zc0 + ZC < a.s1) ? zc0 + ZC : a.s1;
  const int nsteps = (int)(zc1 - zc0) + 1;
  const int nvx = (a.x1 - i0 < HX_TX) ? (int)(a.x1 - i0) : HX_TX;
  const int nvy = (a.y1 - j0 < HX_TY) ? (int)(a.y1 - j0) : HX_TY;

  // ---- row lane: (role group, node, element of the quad) ----
  const int grp = t / 96, u = t - 96 * grp;  // 0: A, 1 / 2: start and finish in turn, 3: D
  const int node = u >> 2, qx = u & 1, qy = (u >> 1) & 1;
  const int nxl = node % HX_TX, nyl = node / HX_TX;
  const int slot = (nxl + qx) + HX_SY * (nyl + qy);  // element (ox, oy) = (qx - 1, qy - 1) of the node
  // local nodes of that element in its z-face: the node's own corner, and the four (near | far in x) x (near | far in y)
  const int qa = hx_quad(1 - qx, 1 - qy);
  const int qb00 = qa, qb10 = hx_quad(qx, 1 - qy), qb01 = hx_quad(1 - qx, qy), qb11 = hx_quad(qx, qy);  // [fx][fy]

  // ---- record lane: (Gauss point, element) ----
  const bool rlane = t < HX_NREC;
  const int rp_ = rlane ? t / HX_NS : 0, re = rlane ? t - HX_NS * rp_ : 0;
  const int rex = re % HX_SY, rey = re / HX_SY;
  const int64_t gex = i0 - 1 + rex, gey = j0 - 1 + rey;
  const bool rxy_ok = rlane && gex >= 0 && gex < M - 1 && gey >= 0 && gey < N - 1;
  // coefficient lane: one per element of the layer (a wave of its own: no divergence with the record lanes)
  const bool clane = t >= 320 && t < 320 + HX_NS;
  const int ce = clane ? t - 320 : 0;
  const int64_t cgex = i0 - 1 + (ce % HX_SY), cgey = j0 - 1 + (ce / HX_SY);
  const bool cxy_ok = clane && cgex >= 0 && cgex < M - 1 && cgey >= 0 && cgey < N - 1;
  const bool has_rec = a.mat_layout == 0, has_soa = a.mat_layout == 3;

  // node coordinates: one double per lane and layer (48 nodes x 3), clamped reads
  auto load_x = [&](int64_t k) -> double {
    if (t >= HX_NL * 3) return 0.0;
    const int nd = t / 3, d = t - 3 * nd;
    int64_t gi = i0 - 1 + (nd % HX_NX), gj = j0 - 1 + (nd / HX_NX), gk = k;
    gi = gi < 0 ? 0 : (gi > M - 1 ? M - 1 : gi);
    gj = gj < 0 ? 0 : (gj > N - 1 ? N - 1 : gj);
    gk = gk < s.cks ? s.cks : (gk > a.cke - 1 ? a.cke - 1 : gk);
    return a.coords[3 * (gi + M * (gj + N * (gk - s.cks))) + d];
  };
  {
    const double v0 = load_x(zc0 - 1), v1 = load_x(zc0);
    if (t < HX_NL * 3) sh_x[0][t] = v0, sh_x[1][t] = v1;
  }
  __syncthreads();

  double carry[22];
#pragma unroll
  for (int q = 0; q < 22; q++) carry[q] = 0.0;

  for (int st = 0; st < nsteps; st++) {
    const int64_t L = zc0 - 1 + st;  // element layer of this step: node layers L (sh_x[st & 1]) and L + 1
    const bool lay_ok = L >= s.els && L < s.ele;
    const double xnext = load_x(L + 2);  // next step's upper node layer, committed after the record phase

    // ================= record phase: 8 Gauss points x 35 elements, one record per lane =================
    if (rlane) {
      double2 *dst = sh_ring + (rp_ * HX_CELLS) * HX_NS + re;
      const bool eok = rxy_ok && lay_ok;
      if (!eok) {
#pragma unroll
        for (int c = 0; c < HX_CELLS; c++) dst[c * HX_NS] = make_double2(0.0, 0.0);
      } else {
        // premise: the element's (E, nu) at this lane's Gauss point equal those at Gauss point 0, bit for bit
        double mE = 0.0, mNu = 0.0, mE0 = 0.0, mNu0 = 0.0;
        if (has_rec) {
          const double *mq = a.gp + (gex + (M - 1) * (gey + (N - 1) * (L - s.mat_e_ks))) * a.mat_stride + 96;
          mE = mq[rp_], mNu = mq[8 + rp_], mE0 = mq[0], mNu0 = mq[8];
        }
        // reference gradients at this lane's Gauss point (A2, A5): G[r][n] = (s_r(n)/8) (1 + s_u(n) xi_u) (1 + s_v(n) xi_v)
        const double gx = HSX[0] * 0.0 + ((rp_ == 2 || rp_ == 3 || rp_ == 6 || rp_ == 7) ? GAUSS_G : -GAUSS_G);
        const double gy = (rp_ == 1 || rp_ == 2 || rp_ == 5 || rp_ == 6) ? GAUSS_G : -GAUSS_G;
        const double gz = (rp_ >= 4) ? GAUSS_G : -GAUSS_G;
        const double ax[2] = {1.0 - gx, 1.0 + gx}, ay[2] = {1.0 - gy, 1.0 + gy}, az[2] = {1.0 - gz, 1.0 + gz};
        const double *x0 = sh_x[st & 1], *x1 = sh_x[(st + 1) & 1];
        double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        double Gn[8][3];
#pragma unroll
        for (int n = 0; n < 8; n++) {
          const int bx = HDI[n], by = HDJ[n], bz = HDK[n];
          Gn[n][0] = (HSX[n] * 0.125) * ay[by] * az[bz];
          Gn[n][1] = (HSY[n] * 0.125) * ax[bx] * az[bz];
          Gn[n][2] = (HSZ[n] * 0.125) * ax[bx] * ay[by];
          const double *xn = (bz ? x1 : x0) + 3 * ((rex + bx) + HX_NX * (rey + by));
          const double c0 = xn[0], c1 = xn[1], c2 = xn[2];
#pragma unroll
          for (int r = 0; r < 3; r++) {
            J[r][0] = J[r][0] + Gn[n][r] * c0;
            J[r][1] = J[r][1] + Gn[n][r] * c1;
            J[r][2] = J[r][2] + Gn[n][r] * c2;
          }
        }
        double iJ[3][3], det;
        inv3_ref(J, iJ, det);
        const double wJ = 1.0 * det;
        if (!(wJ > 0.0)) atomicAdd(a.bad, 1ULL);
#pragma unroll
        for (int n = 0; n < 8; n++) {
          double g[3];
#pragma unroll
          for (int k = 0; k < 3; k++) g[k] = Gn[n][0] * iJ[k][0] + Gn[n][1] * iJ[k][1] + Gn[n][2] * iJ[k][2];
          dst[(2 * n) * HX_NS] = make_double2(g[0], g[1]);
          dst[(2 * n + 1) * HX_NS] = make_double2(g[2], wJ);
        }
        if (has_rec && (mE != mE0 || mNu != mNu0)) atomicAdd(a.bad + 1, 1ULL);
      }
    } else if (clane) {
      double c11 = 0.0, c12 = 0.0, c44 = 0.0;
      if (cxy_ok && lay_ok) {
        double E = a.E, nu = a.nu;
        const int64_t eidx = cgex + (M - 1) * (cgey + (N - 1) * (L - s.mat_e_ks));
        if (has_rec) {
          const double *mq = a.gp + eidx * a.mat_stride + 96;
          E = mq[0], nu = mq[8];
        } else if (has_soa) {
          E = a.gp[eidx], nu = a.gp[a.mat_stride + eidx];
        }
        scaled_coefs<KIND_ELASTIC>(E, nu, 1.0, c11, c12, c44);  // S3:487-495 without the w detJ of S3:496-500
      }
      sh_c[3 * ce] = c11, sh_c[3 * ce + 1] = c12, sh_c[3 * ce + 2] = c44;
    }
    __syncthreads();  // B1: records and coefficients of element layer L are complete
    if (t < HX_NL * 3) sh_x[st & 1][t] = xnext;  // node layer L + 2 over node layer L (read again after B4)

    // ================= contraction: W[fx][fy][i][j] += (wJ ga_i) gb_j over the 8 Gauss points =================
    const bool fin = (grp == 1) ? (st & 1) : !(st & 1);          // groups 1 / 2: finish (else start) in this step
    const bool up = grp == 0 || ((grp == 1 || grp == 2) && !fin);  // roles A and start serve node layer L + 1
    const int a_z = up ? 1 : 0;                                   // the node's face of the element
    const int b_z = (grp == 3 || ((grp == 1 || grp == 2) && !fin)) ? 1 : 0;  // column nodes: start and D the top face
    const int dz = grp == 0 ? 0 : (grp == 3 ? 2 : 1);             // third of the row: A 0, start / finish 1, D 2
    double W[4][9];
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
      for (int q = 0; q < 9; q++) W[b][q] = 0.0;
    {
      const double2 *base = sh_ring + slot;
      const int ca = 2 * (4 * a_z + qa) * HX_NS;
      const int cb[4] = {2 * (4 * b_z + qb00) * HX_NS, 2 * (4 * b_z + qb10) * HX_NS, 2 * (4 * b_z + qb01) * HX_NS,
                         2 * (4 * b_z + qb11) * HX_NS};  // [fx + 2 fy]
#pragma unroll 2
      for (int p = 0; p < 8; p++) {
        const double2 *rp = base + p * (HX_CELLS * HX_NS);
        const double2 a0 = rp[ca], a1 = rp[ca + HX_NS];
        const double l0 = a1.y * a0.x, l1 = a1.y * a0.y, l2 = a1.y * a1.x;
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const double2 b0 = rp[cb[b]], b1 = rp[cb[b] + HX_NS];
          W[b][0] = fma(l0, b0.x, W[b][0]); W[b][1] = fma(l0, b0.y, W[b][1]); W[b][2] = fma(l0, b1.x, W[b][2]);
          W[b][3] = fma(l1, b0.x, W[b][3]); W[b][4] = fma(l1, b0.y, W[b][4]); W[b][5] = fma(l1, b1.x, W[b][5]);
          W[b][6] = fma(l2, b0.x, W[b][6]); W[b][7] = fma(l2, b0.y, W[b][7]); W[b][8] = fma(l2, b1.x, W[b][8]);
        }
      }
    }
    // the element's material: K = c12 W + c44 W^T off the diagonal, c11 W_ii + c44 (W_jj + W_kk) on it (in place)
    {
      const double c11 = sh_c[3 * slot], c12 = sh_c[3 * slot + 1], c44 = sh_c[3 * slot + 2];
#pragma unroll
      for (int b = 0; b < 4; b++) {
        double *w = W[b];
        const double d0 = w[0], d1 = w[4], d2 = w[8];
        w[0] = fma(c11, d0, c44 * (d1 + d2));
        w[4] = fma(c11, d1, c44 * (d0 + d2));
        w[8] = fma(c11, d2, c44 * (d0 + d1));
#pragma unroll
        for (int pr = 0; pr < 3; pr++) {
          const int ij = pr == 0 ? 1 : (pr == 1 ? 2 : 5), ji = pr == 0 ? 3 : (pr == 1 ? 6 : 7);
          const double wij = w[ij], wji = w[ji];
          w[ij] = fma(c12, wij, c44 * wji);
          w[ji] = fma(c12, wji, c44 * wij);
        }
      }
    }
    // the quad's sums: near-x blocks meet the x partner's, near-y blocks the y partner's (the adds commute: the same
    // bits in every lane that holds a sum; fixed tree: deterministic)
#pragma unroll
    for (int q = 0; q < 9; q++) {
      W[0][q] += quad_perm<0xB1>(W[0][q]);  // [fx 0][fy 0]
      W[2][q] += quad_perm<0xB1>(W[2][q]);  // [fx 0][fy 1]
    }
#pragma unroll
    for (int q = 0; q < 9; q++) {
      W[0][q] += quad_perm<0x4E>(W[0][q]);
      W[1][q] += quad_perm<0x4E>(W[1][q]);  // [fx 1][fy 0]
    }
    // every lane keeps a disjoint part: the far corner block, 5 of the 9 of each edge block, 3 of the centre block
    double nk[22];
#pragma unroll
    for (int q = 0; q < 9; q++) nk[q] = W[3][q];
#pragma unroll
    for (int q = 0; q < 5; q++) {
      nk[9 + q] = qx ? W[2][4 + q] : W[2][q];
      nk[14 + q] = qy ? W[1][4 + q] : W[1][q];
    }
    const int qc = (qx + 2 * qy) < 2 ? (qx + 2 * qy) : 2;  // centre block: row qc (lanes 2 and 3 both hold row 2)
#pragma unroll
    for (int q = 0; q < 3; q++) nk[19 + q] = qc == 0 ? W[0][q] : (qc == 1 ? W[0][3 + q] : W[0][6 + q]);
    __syncthreads();  // B2: every lane has read its records; the stage may overwrite them

    // ================= deposit: node layer L, in HBM order =================
    const bool out_layer = st > 0;  // node layer L = zc0 - 1 + st belongs to this workgroup
    if (out_layer && grp != (fin ? -1 : ((grp == 1 || grp == 2) ? grp : -1))) {
      // (all roles but start: A writes what it carried, finish adds its carry to this step's, D writes this step's)
      double *d = sh_stage + node * HX_NB + dz * 27;
      const int fxo = 6 * qx, fyo = 18 * qy;  // far neighbour: x 0 | 2 -> 3 * nx, y 0 | 2 -> 9 * ny
      const bool isA = grp == 0, isF = (grp == 1 || grp == 2);
      auto val = [&](int q) { return isA ? carry[q] : (isF ? carry[q] + nk[q] : nk[q]); };
#pragma unroll
      for (int q = 0; q < 9; q++) d[(q / 3) * 81 + (q % 3) + fyo + fxo] = val(q);
#pragma unroll
      for (int q = 0; q < 5; q++) {
        const int ib = q + 4 * qx, ic = q + 4 * qy;
        d[(ib / 3) * 81 + (ib % 3) + fyo + 3] = val(9 + q);
        d[(ic / 3) * 81 + (ic % 3) + 9 + fxo] = val(14 + q);
      }
#pragma unroll
      for (int q = 0; q < 3; q++) d[qc * 81 + q + 9 + 3] = val(19 + q);
    }
    // what the next step needs from this one: A and start keep this step's numbers
    if (grp == 0 || ((grp == 1 || grp == 2) && !fin)) {
#pragma unroll
      for (int q = 0; q < 22; q++) carry[q] = nk[q];
    }
    __syncthreads();  // B3: the node layer is complete in the stage

    // ================= store: four runs of nvx nodes, whole 128-byte lines =================
    if (out_layer) {
      const int rr = t / 96, ct = t - 96 * rr;
      if (rr < nvy) {
        const int n = nvx * HX_NB;
        double *runp = a.out + (row_start(s, i0, j0 + rr, L, 0) - a.base_off);
        const double *src = sh_stage + rr * HX_RUN;
        const int head = (int)((reinterpret_cast<uintptr_t>(runp) & 127) >> 3);  // doubles above a line boundary
        const int nchunk = (head + n + 1) >> 1;
        constexpr int NIT = (HX_RUN / 2 + 8 + 95) / 96;
        double2 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {  // chunk q = the q-th 16 bytes from that boundary; LDS reads first
          const int s0 = 2 * (ct + 96 * it) - head;
          const int sa = s0 < 0 ? 0 : (s0 < n ? s0 : n - 1), sb = s0 + 1 < 0 ? 0 : (s0 + 1 < n ? s0 + 1 : n - 1);
          v[it] = make_double2(src[sa], src[sb]);
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
          const int q = ct + 96 * it, s0 = 2 * q - head;
          if (q < nchunk) {
            if (s0 >= 0 && s0 + 1 < n) store_pair(runp + s0, v[it]);
            else if (s0 == -1) runp[0] = v[it].y;
            else if (s0 == n - 1) runp[s0] = v[it].x;
          }
        }
      }
    }
    __syncthreads();  // B4: the stage has been read; the next step's records may overwrite it
  }
}

}  // namespace femasm
