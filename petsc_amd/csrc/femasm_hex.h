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
//    quad-permute steps (a fixed tree: deterministic), then every lane keeps part of the quad's 81 numbers (whole rows of blocks: 24);
//  * a step = one element layer L.  All 8 x 55 Gauss-point records of the halo'd 11 x 5 element tile are built in ONE
//    phase by 440 of the 640 row lanes (lane = (Gauss point, element)), so a step has four barriers, not nine;
//  * two loader waves are the only waves that read global memory (coordinates, (E, nu)), a whole step ahead, and they
//    never write any (gfx9 retires a wave's loads and stores through one in-order counter);
//  * what a step leaves for the next one (the dz = -1 third and the lower half of the dz = 0 third of node layer L+1)
//    waits in two small LDS areas, so the staging area of the finished node layer can alias the records and no lane
//    carries anything in registers across a step (the first cut kept them in registers and spilled 158 of them);
//  * the finished node layer leaves as four contiguous runs (10 nodes x 243 values), 16 B per lane from the 128-byte
//    boundary below each run: whole lines only (partial lines cost four writes each).
// 768 threads = 12 waves = three per SIMD, ONE workgroup per CU (LDS 148 KB, <= 168 VGPRs).  Measured on the way: two
// 6-wave workgroups per CU (LDS 78 KB each, what the occupancy API promises) never run together -- a workgroup's waves
// land on the SIMDs in a fixed pattern (2, 2, 1, 1), a second one would need a fourth wave's registers on SIMD 0 -- so
// a workgroup is a multiple of four waves here.
#pragma once
#include "femasm_sweep.h"
#ifndef HX_UNROLL
#define HX_UNROLL 2
#endif
#ifndef HX_EXP
#define HX_EXP 0  // profiling builds: 1 no global stores, 2 no store phase, 3 no record arithmetic, 4 no contraction,
#endif            // 9 phase clocks of waves 0 and 4 of one workgroup (printed by femasm_plan_destroy)
#if HX_EXP == 9
#define HXT(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); tk[i] += now_ - tk0; tk0 = now_; } while (0)
#else
#define HXT(i)
#endif

namespace femasm {

constexpr int HX_TX = 10, HX_TY = 4, HX_NN = HX_TX * HX_TY;  // node tile
constexpr int HX_SY = HX_TX + 1, HX_NS = HX_SY * (HX_TY + 1);  // element tile 11 x 5 (halo of one), slot = ex + 11 ey
constexpr int HX_NX = HX_TX + 2, HX_NL = HX_NX * (HX_TY + 2);  // halo'd node tile 12 x 6 of one layer
constexpr int HX_NROW = 16 * HX_NN;                           // 640 row lanes: (role group, node, element of the quad)
constexpr int HX_NA = 512;                                    // ... of which waves 0..7 are one role each (nodes 0..31)
constexpr int HX_NT = HX_NROW + 128;                          // + two loader waves: 12 waves, ONE workgroup per CU
constexpr int HX_NREC = 8 * HX_NS;                            // record lanes: (Gauss point, element)
// record = 13 cells of 16 B: cell n < 8: (gx, gy) of node n; cell 8 + m: gz of nodes 2m, 2m + 1; cell 12: (w detJ, -)
constexpr int HX_CELLS = 13;
constexpr int HX_NB = 243, HX_RUN = HX_TX * HX_NB;            // values of a node block, of a tile-row run
constexpr int HX_ZC = 42;                                     // default node layers per workgroup (KArgs::zc overrides)
static_assert(HX_NREC <= HX_NROW && 2 * HX_NS <= 128, "record lanes among the row waves, two loader lanes per element");
static_assert(HX_NL * 3 <= 256 && HX_TY == 4, "two coordinates per loader lane; two storing waves per run");
static_assert(HX_NN == 40, "lane map of the row lanes: eight waves of 16 nodes (two per role), two of 4 x 4 nodes");

#if HX_EXP == 9
__device__ unsigned hx_dbg_cnt[8192];  // resident workgroups per compute unit (profiling build)
#endif
__device__ inline int hx_quad(int bx, int by) { return bx == 0 ? by : 3 - by; }  // = quad_idx, branch-free

__global__ __launch_bounds__(HX_NT, 3) void k_assemble_s3_hex(const Slab s, const KArgs a) {
  if (a.cond && a.bad[1] == 0) return;
  __shared__ __attribute__((aligned(16))) double2 sh_ring[8 * HX_CELLS * HX_NS];  // 91 520 B; the stage aliases it
  __shared__ double sh_cA[HX_NN * 81], sh_cM[HX_NN * 81];  // node layer L + 1: its dz = -1 third, its dz = 0 third from below
  __shared__ double sh_x[2][HX_NL * 3];                    // node layers L, L + 1 (ring of two)
  __shared__ double sh_c[HX_NS * 3];                       // (c11, c12, c44) of the layer's elements
  double *sh_stage = reinterpret_cast<double *>(sh_ring);
  static_assert((HX_NN * HX_NB + 32) * 8 <= 8 * HX_CELLS * HX_NS * 16, "stage (+ slack) fits the ring");

  const int t = threadIdx.x;
#if HX_EXP == 9
  unsigned hwkey = 0;
  if (t == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
    hwkey = ((xcc & 15) << 9) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
    const unsigned old = atomicAdd(&hx_dbg_cnt[hwkey], 1u);
    atomicMax(a.bad + 25, (unsigned long long)(old + 1));
  }
#endif
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t i0 = a.x0 + (int64_t)blockIdx.x * HX_TX, j0 = a.y0 + (int64_t)blockIdx.y * HX_TY;
  const int ZC = a.zc > 0 ? a.zc : HX_ZC;
  const int64_t zc0 = a.s0 + (int64_t)blockIdx.z * ZC;
  const int64_t zc1 = (zc0 + ZC < a.s1) ? zc0 + ZC : a.s1;
  const int nsteps = (int)(zc1 - zc0) + 1;
  const int nvx = (a.x1 - i0 < HX_TX) ? (int)(a.x1 - i0) : HX_TX;
  const int nvy = (a.y1 - j0 < HX_TY) ? (int)(a.y1 - j0) : HX_TY;

  const bool has_rec = a.mat_layout == 0, has_soa = a.mat_layout == 3;

  // ---- the loader wave ----
  // Waves 10 and 11 are the only waves that read global memory, and they never write any: gfx9 retires a wave's loads and stores
  // through one in-order counter, so a load issued by a wave that has stores in flight waits for their acknowledgement
  // What they load is requested a whole step ahead and held in registers: two node
  // coordinates per lane, and per pair of lanes the (E, nu) of one element at its 8 Gauss points (lane 2e: E[8], lane
  // 2e + 1: nu[8]; their equality is the kernel's premise).
  const bool w4 = t >= HX_NROW;  // the loader wave (wave-uniform)
  auto load_x = [&](int idx, int64_t k) -> double {
    if (idx >= HX_NL * 3) return 0.0;
    const int nd = idx / 3, d = idx - 3 * nd;
    int64_t gi = i0 - 1 + (nd % HX_NX), gj = j0 - 1 + (nd / HX_NX), gk = k;
    gi = gi < 0 ? 0 : (gi > M - 1 ? M - 1 : gi);
    gj = gj < 0 ? 0 : (gj > N - 1 ? N - 1 : gj);
    gk = gk < s.cks ? s.cks : (gk > a.cke - 1 ? a.cke - 1 : gk);
    return a.coords[3 * (gi + M * (gj + N * (gk - s.cks))) + d];
  };
  auto load_mat = [&](int l, int64_t lay, double (&pm)[8]) {  // (clamped element indices: what lies outside is not used)
    const int e = (l >> 1) < HX_NS ? (l >> 1) : HX_NS - 1, half = l & 1;
    int64_t ex = i0 - 1 + (e % HX_SY), ey = j0 - 1 + (e / HX_SY), ez = lay;
    ex = ex < 0 ? 0 : (ex > M - 2 ? M - 2 : ex);
    ey = ey < 0 ? 0 : (ey > N - 2 ? N - 2 : ey);
    ez = ez > s.ele - 1 ? s.ele - 1 : ez;
    ez = ez < s.els ? s.els : ez;
    const int64_t eidx = ex + (M - 1) * (ey + (N - 1) * (ez - s.mat_e_ks));
    if (has_rec) {
      const double *mq = a.gp + eidx * a.mat_stride + 96 + 8 * half;
#pragma unroll
      for (int q = 0; q < 8; q++) pm[q] = mq[q];
    } else if (has_soa) {
      const double v = a.gp[(half ? (int64_t)a.mat_stride : 0) + eidx];
#pragma unroll
      for (int q = 0; q < 8; q++) pm[q] = v;
    }
  };
  if (w4) {
    // the loader wave's own loop: the same four barriers per step as the row waves
    const int l = t - HX_NROW, e = l >> 1, half = l & 1;
    double px[2], pm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    {
      const double v0 = load_x(l, zc0 - 1), v1 = load_x(l + 128, zc0 - 1), v2 = load_x(l, zc0), v3 = load_x(l + 128, zc0);
      sh_x[0][l] = v0, sh_x[1][l] = v2;
      if (l + 128 < HX_NL * 3) sh_x[0][l + 128] = v1, sh_x[1][l + 128] = v3;
    }
    px[0] = load_x(l, zc0 + 1), px[1] = load_x(l + 128, zc0 + 1);
    if (s.ele > s.els) load_mat(l, zc0 - 1, pm);
    __syncthreads();
    const int64_t cex = i0 - 1 + (e % HX_SY), cey = j0 - 1 + (e / HX_SY);
    const bool exy_ok = e < HX_NS && cex >= 0 && cex < M - 1 && cey >= 0 && cey < N - 1;
    for (int st = 0; st < nsteps; st++) {
      const int64_t L = zc0 - 1 + st;
      // coefficients of the layer's elements from what was requested a step ago; premise: one (E, nu) per element
      const bool ok = exy_ok && L >= s.els && L < s.ele;
      bool same = true;
#pragma unroll
      for (int q = 1; q < 8; q++) same = same && pm[q] == pm[0];
      if (ok && has_rec && !same) atomicAdd(a.bad + 1, 1ULL);
      const double mine = pm[0], other = quad_perm<0xB1>(mine);  // lane 2e: E, lane 2e + 1: nu
      if (e < HX_NS && half == 0) {
        double c11 = 0.0, c12 = 0.0, c44 = 0.0;
        if (ok) {
          const bool hm = has_rec || has_soa;
          scaled_coefs<KIND_ELASTIC>(hm ? mine : a.E, hm ? other : a.nu, 1.0, c11, c12, c44);  // S3:487-495, w detJ apart
        }
        sh_c[3 * e] = c11, sh_c[3 * e + 1] = c12, sh_c[3 * e + 2] = c44;
      }
      if (s.ele > s.els) load_mat(l, L + 1, pm);  // the next step's
      __syncthreads();                            // B1
      sh_x[st & 1][l] = px[0];                    // node layer L + 2 over node layer L (read again after B4)
      if (l + 128 < HX_NL * 3) sh_x[st & 1][l + 128] = px[1];
      px[0] = load_x(l, L + 3), px[1] = load_x(l + 128, L + 3);
      __syncthreads();  // B2
      __syncthreads();  // B3
      __syncthreads();  // B4
    }
    return;
  }
  __syncthreads();

#if HX_EXP == 9
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tk0 = __builtin_readcyclecounter();
#endif
  for (int st = 0; st < nsteps; st++) {
    const int64_t L = zc0 - 1 + st;  // element layer of this step: node layers L (sh_x[st & 1]) and L + 1
    const bool lay_ok = L >= s.els && L < s.ele;
    // (lane-derived indices are recomputed from an opaque copy of the lane id in every step: derived once before the
    // loop, the compiler keeps several dozen loop-invariant addresses alive and spills them -- DESIGN section 3)
    int tl = t;
    asm volatile("" : "+v"(tl));
    // ---- record lane: (Gauss point, element) ----
    const bool rlane = tl < HX_NREC;
    const int rp_ = rlane ? tl / HX_NS : 0, re = rlane ? tl - HX_NS * rp_ : 0;
    const int mex = re % HX_SY, mey = re / HX_SY;
    const int64_t gex = i0 - 1 + mex, gey = j0 - 1 + mey;
    const bool mxy_ok = rlane && gex >= 0 && gex < M - 1 && gey >= 0 && gey < N - 1;

    // ================= record phase: 8 Gauss points x 30 elements, one record per lane =================
    if (rlane) {
      double2 *dst = sh_ring + (rp_ * HX_CELLS) * HX_NS + re;
      if (!(mxy_ok && lay_ok) || HX_EXP == 3) {
#pragma unroll
        for (int c = 0; c < HX_CELLS; c++) dst[c * HX_NS] = make_double2(0.0, 0.0);
      } else {
        // reference gradients at this lane's Gauss point (A2, A5): G[r][n] = (s_r(n)/8) (1 + s_u(n) xi_u) (1 + s_v(n) xi_v)
        // (Gauss point p sits in the corner of node p: signs HSX / HSY / HSZ of p as bit masks)
        const double gx = ((0xCC >> rp_) & 1) ? GAUSS_G : -GAUSS_G;
        const double gy = ((0x66 >> rp_) & 1) ? GAUSS_G : -GAUSS_G;
        const double gz = (rp_ >= 4) ? GAUSS_G : -GAUSS_G;
        const double ax[2] = {1.0 - gx, 1.0 + gx}, ay[2] = {1.0 - gy, 1.0 + gy}, az[2] = {1.0 - gz, 1.0 + gz};
        const double *x0 = sh_x[st & 1], *x1 = sh_x[(st + 1) & 1];
        double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
        for (int n = 0; n < 8; n++) {
          const int bx = HDI[n], by = HDJ[n], bz = HDK[n];
          const double g0 = (HSX[n] * 0.125) * ay[by] * az[bz];
          const double g1 = (HSY[n] * 0.125) * ax[bx] * az[bz];
          const double g2 = (HSZ[n] * 0.125) * ax[bx] * ay[by];
          const double *xn = (bz ? x1 : x0) + 3 * ((mex + bx) + HX_NX * (mey + by));
          const double c0 = xn[0], c1 = xn[1], c2 = xn[2];
          J[0][0] = J[0][0] + g0 * c0; J[0][1] = J[0][1] + g0 * c1; J[0][2] = J[0][2] + g0 * c2;
          J[1][0] = J[1][0] + g1 * c0; J[1][1] = J[1][1] + g1 * c1; J[1][2] = J[1][2] + g1 * c2;
          J[2][0] = J[2][0] + g2 * c0; J[2][1] = J[2][1] + g2 * c1; J[2][2] = J[2][2] + g2 * c2;
        }
        double iJ[3][3], det;
        inv3_ref(J, iJ, det);
        const double wJ = 1.0 * det;
        if (!(wJ > 0.0)) atomicAdd(a.bad, 1ULL);
        double gzp = 0.0;
#pragma unroll
        for (int n = 0; n < 8; n++) {
          const int bx = HDI[n], by = HDJ[n], bz = HDK[n];
          const double g0 = (HSX[n] * 0.125) * ay[by] * az[bz];
          const double g1 = (HSY[n] * 0.125) * ax[bx] * az[bz];
          const double g2 = (HSZ[n] * 0.125) * ax[bx] * ay[by];
          double g[3];
#pragma unroll
          for (int k = 0; k < 3; k++) g[k] = g0 * iJ[k][0] + g1 * iJ[k][1] + g2 * iJ[k][2];
          dst[n * HX_NS] = make_double2(g[0], g[1]);
          if (n & 1) dst[(8 + n / 2) * HX_NS] = make_double2(gzp, g[2]);
          gzp = g[2];
        }
        dst[12 * HX_NS] = make_double2(wJ, 0.0);
      }
    }
    HXT(0);
    __syncthreads();  // B1: records and coefficients of element layer L are complete
    HXT(1);

    // ================= contraction: W[fx][fy][i][j] += (wJ ga_i) gb_j over the 8 Gauss points =================
    // roles: A (group 0) and start serve node layer L + 1 from below, finish and D (group 3) node layer L from above;
    // groups 1 and 2 take start and finish in turn (a node's dz = 0 third is started from below, finished from above).
    // Waves 0..7 are the four groups for nodes 0..31 (two waves each), waves 8 and 9 hold two groups each for nodes
    // 32..39.
    asm volatile("" : "+v"(tl));
    const int tw = tl - HX_NA;
    const int grp = tl < HX_NA ? (tl >> 7) : ((tw >> 5) & 3);  // 0: A, 1 / 2: start and finish in turn, 3: D
    const int node = tl < HX_NA ? ((tl & 127) >> 2) : 32 + ((tw & 31) >> 2);
    const int qx = tl & 1, qy = (tl >> 1) & 1;
    const int slot = (node % HX_TX + qx) + HX_SY * (node / HX_TX + qy);  // element (ox, oy) = (qx - 1, qy - 1) of the node
    const bool isA = grp == 0, isD = grp == 3, isM = !isA && !isD;
    const bool fin = isM && ((grp == 1) == ((st & 1) != 0));
    const bool isS = isM && !fin;
    double W[4][9];
    {
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
      for (int q = 0; q < 9; q++) W[b][q] = 0.0;
    {
      const int a_z = (isA || isS) ? 1 : 0;  // the node's face of the element
      const int b_z = (isS || isD) ? 1 : 0;  // the column nodes' face: A and finish the bottom, start and D the top
      // local nodes: the node's own corner, and the four (near | far in x) x (near | far in y) of the column face
      const int na = 4 * a_z + hx_quad(1 - qx, 1 - qy);
      const int nb[4] = {4 * b_z + hx_quad(1 - qx, 1 - qy), 4 * b_z + hx_quad(qx, 1 - qy),
                         4 * b_z + hx_quad(1 - qx, qy), 4 * b_z + hx_quad(qx, qy)};  // [fx + 2 fy]
      const char *base = reinterpret_cast<const char *>(sh_ring + slot);
      constexpr int CS = HX_NS * 16;  // bytes from cell to cell
      const int oa = na * CS, oaz = (8 + (na >> 1)) * CS + 8 * (na & 1);
      int ob[4], obz[4];
#pragma unroll
      for (int b = 0; b < 4; b++) ob[b] = nb[b] * CS, obz[b] = (8 + (nb[b] >> 1)) * CS + 8 * (nb[b] & 1);
#if HX_EXP != 4
#pragma unroll HX_UNROLL
      for (int p = 0; p < 8; p++) {
        const char *rp = base + p * (HX_CELLS * CS);
        // all eleven reads of this Gauss point first, one wait, then the 39 products
        const double2 a0 = *reinterpret_cast<const double2 *>(rp + oa);
        const double a2 = *reinterpret_cast<const double *>(rp + oaz);
        const double wJ = *reinterpret_cast<const double *>(rp + 12 * CS);
        double2 b0[4];
        double b2[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
          b0[b] = *reinterpret_cast<const double2 *>(rp + ob[b]);
          b2[b] = *reinterpret_cast<const double *>(rp + obz[b]);
        }
        const double l0 = wJ * a0.x, l1 = wJ * a0.y, l2 = wJ * a2;
#pragma unroll
        for (int b = 0; b < 4; b++) {
          W[b][0] = fma(l0, b0[b].x, W[b][0]); W[b][1] = fma(l0, b0[b].y, W[b][1]); W[b][2] = fma(l0, b2[b], W[b][2]);
          W[b][3] = fma(l1, b0[b].x, W[b][3]); W[b][4] = fma(l1, b0[b].y, W[b][4]); W[b][5] = fma(l1, b2[b], W[b][5]);
          W[b][6] = fma(l2, b0[b].x, W[b][6]); W[b][7] = fma(l2, b0[b].y, W[b][7]); W[b][8] = fma(l2, b2[b], W[b][8]);
        }
      }
#else
      (void)base, (void)oa, (void)oaz;
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int q = 0; q < 9; q++) W[b][q] = (double)(ob[b] + obz[b] + q);
#endif
    }
    HXT(2);
    __syncthreads();  // B2: every lane has read its records; the stage may overwrite them
    HXT(3);
    // (material, quad sums and the choice of what a lane keeps come AFTER the barrier: they touch registers only, and the
    // phase they now share with the deposit is bound by LDS, not by vector issue like the contraction)
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
    }
    // Every lane keeps part of the quad's 81 numbers: the far corner block (9), two of the three rows of each edge block
    // (6 + 6: the x partners split the x-edge block as rows {0,1} | {1,2}, the y partners the y-edge block), one row of
    // the centre block (3; lanes 2 and 3 both row 2).  Rows, so that a kept entry's place in a third [c][ny][nx][c']
    // is a lane base + a compile-time offset: nothing but the value selects is computed per entry.
    const int qc = (qx + 2 * qy) < 2 ? (qx + 2 * qy) : 2;
    double nk[24];
#pragma unroll
    for (int q = 0; q < 9; q++) nk[q] = W[3][q];
#pragma unroll
    for (int q = 0; q < 6; q++) {
      nk[9 + q] = qx ? W[2][3 + q] : W[2][q];
      nk[15 + q] = qy ? W[1][3 + q] : W[1][q];
    }
#pragma unroll
    for (int q = 0; q < 3; q++) nk[21 + q] = qc == 0 ? W[0][q] : (qc == 1 ? W[0][3 + q] : W[0][6 + q]);
    // lane bases for row stride RS (27 in a carry, 81 in the stage); far neighbour: nx 0 | 2 -> 3 nx, ny 0 | 2 -> 9 ny
    struct Pos { int corner, xedge, yedge, centre; };
    auto bases = [&](int RS) {
      return Pos{18 * qy + 6 * qx, 18 * qy + 3 + qx * RS, 9 + 6 * qx + qy * RS, 12 + qc * RS};
    };
    // entry q of nk: its base and its compile-time offset (i = 3 c + c' of the kept rows -> c RS + c')
#define HX_POS(P_, q, RS) ((q) < 9 ? (P_).corner + ((q) / 3) * (RS) + (q) % 3 : (q) < 15 ? (P_).xedge + (((q)-9) / 3) * (RS) + ((q)-9) % 3 \
                           : (q) < 21 ? (P_).yedge + (((q)-15) / 3) * (RS) + ((q)-15) % 3 : (P_).centre + ((q)-21))
    const Pos pc = bases(27), ps = bases(81);

    // ================= deposit: node layer L, in HBM order =================
    const bool out_layer = st > 0;  // node layer L = zc0 - 1 + st belongs to this workgroup
    double *stg = sh_stage + 16;    // (16 doubles of slack either side: the store loop reads whole chunks from a line boundary)
    // One instruction stream for all roles: A writes what it carried (the dz = -1 third of layer L), finish adds what
    // start left to this step's numbers (dz = 0), D writes this step's (dz = +1); start writes nothing here.  After the
    // barrier A and start leave this step's numbers for the next one (they belong to layer L + 1).
    double *cy = (isA ? sh_cA : sh_cM) + node * 81;
    {
      double *d = stg + node * HX_NB + (isA ? 0 : (isD ? 54 : 27));
      double v[24];
#pragma unroll
      for (int q = 0; q < 24; q++) v[q] = cy[HX_POS(pc, q, 27)];
#pragma unroll
      for (int q = 0; q < 24; q++) v[q] = isA ? v[q] : (fin ? v[q] + nk[q] : nk[q]);
      if (out_layer && !isS) {
#pragma unroll
        for (int q = 0; q < 24; q++) d[HX_POS(ps, q, 81)] = v[q];
      }
    }
    HXT(4);
    __syncthreads();  // B3: the node layer is complete in the stage; the finish lanes have read what start left
    HXT(5);
    if (isA || isS) {
#pragma unroll
      for (int q = 0; q < 24; q++) cy[HX_POS(pc, q, 27)] = nk[q];
    }
#undef HX_POS

    // ================= store: four runs of nvx nodes, whole 128-byte lines =================
#if HX_EXP != 2
    if (out_layer && tl < HX_NA) {  // waves 0..7: two per run
      asm volatile("" : "+v"(tl));
      const int rr = __builtin_amdgcn_readfirstlane(tl >> 7), ct = tl & 127;  // two waves = one run
      if (rr < nvy) {
        const int n = nvx * HX_NB;
        // rows j0 .. j0 + 3 are interior rows of an interior layer: their runs lie 81 (3M - 2) values apart
        double *runp = a.out + (row_start(s, i0, j0, L, 0) - a.base_off) + (int64_t)rr * (81 * (3 * M - 2));
        const int head = (int)((reinterpret_cast<uintptr_t>(runp) & 127) >> 3);  // doubles above a line boundary
        constexpr int LPR = 128, NIT = (HX_RUN / 2 + 8 + LPR - 1) / LPR;
        // chunk q = the q-th 16 bytes from that boundary = run[2q - head, 2q - head + 1]; whole chunks here, the two
        // possible half chunks (an odd head, an odd end) after the loop
        const int s0 = 2 * ct - head;
        const double *src = stg + rr * HX_RUN + s0;
        double *dstp = runp + s0;
        double2 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) v[it] = make_double2(src[2 * LPR * it], src[2 * LPR * it + 1]);
#pragma unroll
        for (int it = 0; it < NIT; it++) {
          const int sq = s0 + 2 * LPR * it;
          // (a full run: only the first round can start below it and only the last can end beyond it)
          const bool always = it > 0 && 2 * (LPR - 1) + 2 * LPR * it + 2 <= HX_RUN && nvx == HX_TX;
#if HX_EXP == 1
          if (always || (sq >= 0 && sq + 2 <= n)) asm volatile("" ::"v"(v[it].x), "v"(v[it].y));
#else
          if (always || (sq >= 0 && sq + 2 <= n)) store_pair(dstp + 2 * LPR * it, v[it]);
#endif
        }
        if ((head & 1) && ct == 0) runp[0] = stg[rr * HX_RUN];
        if (((head + n) & 1) && ct == 1) runp[n - 1] = stg[rr * HX_RUN + n - 1];
      }
    }
#else
    if (out_layer && tl < 8) a.out[tl] = stg[tl];
#endif
    HXT(6);
    __syncthreads();  // B4: the stage has been read; the next step's records may overwrite it
    HXT(7);
  }
#if HX_EXP == 9
  if (t == 0) atomicSub(&hx_dbg_cnt[hwkey], 1u);
  if (blockIdx.x == 3 && blockIdx.y == 7 && blockIdx.z == 1 && (t == 0 || t == HX_NA)) {
    for (int q = 0; q < 8; q++) atomicAdd(a.bad + 8 + (t ? 8 : 0) + q, tk[q]);
    if (t == 0) atomicAdd(a.bad + 24, (unsigned long long)nsteps);
  }
#endif
}

}  // namespace femasm
