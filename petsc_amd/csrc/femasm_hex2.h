// femasm_hex2.h -- S3/R3 stiffness of interior nodes on ANY hexahedra with ONE (E, nu) per element: kernel level 1 of
// the elastic kind (north_star (2) + (3): Jacobian, inverse and det per Gauss point in FP64 registers, S3:186-227;
// B^T D B, S3:450-518).  Round 3; DESIGN.md section 4.2c has the derivation and the measurements.
//
// Why it is not k_assemble_sweep<ELASTIC> made faster: that kernel holds a third of a node's rows per lane (81 FP64
// accumulators: two waves per SIMD at most), applies the material at every Gauss point (21 FMAs per node pair and Gauss
// point) and meets at nine s_barriers per element layer.  Here
//  * the material leaves the Gauss-point loop.  With one (E, nu) per element (tested bitwise on the records, flag
//    bad[1] -> the general kernel, like every fast path of this library)
//        K_ab[i][j] = c12 W_ab[i][j] + c44 W_ab[j][i]   (i != j)          W_ab[i][j] = sum_p wJ_p ga_i(p) gb_j(p)
//        K_ab[i][i] = c11 W_ab[i][i] + c44 (W_ab[j][j] + W_ab[k][k])
//    so a node pair costs 9 FMAs per Gauss point and 21 once per element;
//  * lane = (node, role, element): role = which z-third of the node's rows and which element layer feeds it (A, start,
//    finish, D), element = one of the four elements (ox, oy) of that layer around the node: 36 accumulators per lane;
//  * the four lanes of a (node, role) are a quad: their blocks are added by two quad-permute steps (a fixed tree:
//    deterministic), then every lane keeps 24 of the quad's 81 numbers (whole rows of blocks).
// The skeleton (the round's first cut, one 12-wave workgroup per CU on 10 x 4 nodes with two loader waves, spent its
// time at barriers -- 20.4 k cycles per step of 40 nodes: every phase of a step is bound by ONE resource, records by
// vector issue, the contraction by LDS reads, the deposit by LDS writes, and one workgroup runs them one after the other):
//  * a workgroup is FOUR waves on a tile of 8 x 2 nodes, one role per wave, and TWO workgroups share a CU: while one
//    contracts (LDS) the other builds records (VALU);
//  * what a step leaves for the next one (the dz = -1 third and the lower half of the dz = 0 third of node layer
//    L + 1) stays in 24 registers of the lane that made it -- with two waves per SIMD there are 256 registers per lane --
//    so the deposit neither reads nor writes a carry area in LDS (half of that phase's LDS traffic);
//  * the LDS image of a Gauss point is laid out so that every read of the contraction is conflict-free: the (gx, gy)
//    pairs of an element face are 16-byte units at  CB[c] + 2 ex + 48 ey,  CB = (0, 24, 1, 25) for the face's four nodes
//    (two nodes interleaved, the other two 24 units on; found by search over the lane groups of ds_read_b128), the gz at
//    CB' + 2 ex + 32 ey, CB' = (0, 112, 1, 113) (groups of ds_read_b64); the writes are two-way at most;
//  * no loader wave: wave 3 requests the next node layer and the next element layer's (E, nu) a step ahead and never
//    stores to global memory, waves 0, 1 and 2 store the two finished runs and never load (gfx9 retires a wave's loads
//    and stores through one in-order counter).
#pragma once
#include "femasm_sweep.h"
#ifndef H2_UNROLL
#define H2_UNROLL 8
#endif
#ifndef H2_EXP
#define H2_EXP 0  // profiling builds: 1 no global stores, 3 no record arithmetic, 4 no contraction, 9 phase clocks
#endif
#if H2_EXP == 9
#define H2T(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); tk[i] += now_ - tk0; tk0 = now_; } while (0)
#else
#define H2T(i)
#endif

namespace femasm {

constexpr int H2_TX = 8, H2_TY = 2, H2_NN = H2_TX * H2_TY;     // node tile
constexpr int H2_SY = H2_TX + 1, H2_NS = H2_SY * (H2_TY + 1);  // element tile 9 x 3 (halo of one), slot = ex + 9 ey
constexpr int H2_NX = H2_TX + 2, H2_NL = H2_NX * (H2_TY + 2);  // halo'd node tile 10 x 4 of one layer
constexpr int H2_NT = 16 * H2_NN;                              // 256 lanes: (role, node, element of the quad)
constexpr int H2_NREC = 8 * H2_NS;                             // record lanes: (Gauss point, element)
constexpr int H2_NB = 243, H2_RUN = H2_TX * H2_NB;             // values of a node block, of a tile-row run
constexpr int H2_RS = H2_RUN + 2;                               // distance of the two runs in the stage (even)
constexpr int H2_ZC = 127;                                     // default node layers per workgroup (KArgs::zc overrides)
// LDS image of one Gauss point (bytes): (gx, gy) units of the two element faces, gz of the two faces, w detJ
constexpr int H2_XY_B = 138 * 16, H2_Z_B = 194 * 8;
constexpr int H2_OFF_Z = 2 * H2_XY_B, H2_OFF_W = H2_OFF_Z + 2 * H2_Z_B, H2_PS = H2_OFF_W + 224;
static_assert(H2_NREC <= H2_NT && 2 * H2_NS <= 64 && H2_NL * 3 <= 128 && (3 * H2_NX) % 2 == 0, "record lanes and loader lanes among the row lanes");
static_assert(H2_PS % 16 == 0 && 7 * H2_PS + H2_PS < 65536, "Gauss-point images are reached through immediate offsets");

__device__ inline int h2_quad(int bx, int by) { return bx == 0 ? by : 3 - by; }  // = quad_idx, branch-free
// place of face-local node c of element (ex, ey) in a face image: 16-byte units of (gx, gy), doubles of gz
__device__ inline int h2_uxy(int c, int ex, int ey) { return ((c & 1) ? 24 : 0) + (c >> 1) + 2 * ex + 48 * ey; }
__device__ inline int h2_uz(int c, int ex, int ey) { return ((c & 1) ? 112 : 0) + (c >> 1) + 2 * ex + 32 * ey; }

typedef double2 __attribute__((aligned(8))) double2_a8;  // two consecutive doubles, 8-byte aligned

// Pre-pass of level 1: tests the premise (one (E, nu) per element, bitwise over its 8 Gauss points; flag bad[1]) and leaves
// the element's (c11, c12, c44, -) = D11, D12, D44 of S3:487-495 (w detJ apart) in a plan-owned array, element layers
// [els, ele), 32 bytes per element.  8 lanes per element read the 128-byte line of E[8], nu[8] of an AoS record.  So the
// sweep kernel issues ONE load per step for its 27 elements instead of eight, and no FP64 division: on a compute unit
// whose memory pipe is full of stores every load instruction costs its wave 500-1000 cycles.
__global__ __launch_bounds__(256) void k_s3_elem_coefs(const Slab s, const KArgs a, double *coef) {
  const int64_t nex = s.dim[0] - 1, ney = s.dim[1] - 1, nel = nex * ney * (s.ele - s.els);
  const int64_t eoff = nex * ney * (s.els - s.mat_e_ks);  // first element's place in the caller's material array
  const int j = threadIdx.x & 7, lane = threadIdx.x & 63;
  constexpr int U = 4;  // elements per lane group and round: four 128-byte lines in flight per 8 lanes
  for (int64_t e0 = ((int64_t)blockIdx.x * 32 + (threadIdx.x >> 3)) * U; e0 < nel; e0 += (int64_t)gridDim.x * 32 * U) {
    double v0[U], v1[U];
    if (a.mat_layout == 0) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int64_t e = e0 + u < nel ? e0 + u : nel - 1;
        const double *mq = a.gp + (e + eoff) * a.mat_stride + 96 + 2 * j;  // lanes 0..3: E[2j], E[2j+1]; lanes 4..7: nu[..]
        v0[u] = mq[0], v1[u] = mq[1];
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t e = e0 + u;
      double E = a.E, nu = a.nu;
      if (a.mat_layout == 0) {
        const double first = __shfl(v0[u], lane & ~3, 64);  // E[0] in lanes 0..3, nu[0] in lanes 4..7
        const bool same = v0[u] == first && v1[u] == first;
        const unsigned long long m = __ballot(same);
        if (j == 0 && e < nel && ((m >> lane) & 0xFFull) != 0xFFull) atomicAdd(a.bad + 1, 1ULL);
        E = v0[u];
        nu = __shfl(v0[u], lane + 4, 64);
      } else if (a.mat_layout == 3 && e < nel) {
        E = a.gp[e + eoff];
        nu = a.gp[(int64_t)a.mat_stride + e + eoff];
      }
      if (j == 0 && e < nel) {
        double c11, c12, c44;
        scaled_coefs<KIND_ELASTIC>(E, nu, 1.0, c11, c12, c44);
        double2 *o = reinterpret_cast<double2 *>(coef + 4 * e);
        o[0] = make_double2(c11, c12);
        o[1] = make_double2(c44, 0.0);
      }
    }
  }
}

__global__ __launch_bounds__(H2_NT, 2) void k_assemble_s3_hex2(const Slab s, const KArgs a) {
  if (a.cond && a.bad[1] == 0) return;
  __shared__ __attribute__((aligned(16))) char sh_rec[8 * H2_PS];  // 61 952 B; the stage aliases it
  __shared__ __attribute__((aligned(16))) double sh_x[2][H2_NL * 3];  // node layers L, L + 1 (ring of two)
  __shared__ __attribute__((aligned(16))) double sh_c[H2_NS * 4];     // (c11, c12, c44, -) of the layer's elements
  double *sh_stage = reinterpret_cast<double *>(sh_rec);
  static_assert((2 * H2_RS + 32) * 8 <= 8 * H2_PS, "stage (+ slack) fits the record area");

  const int t = threadIdx.x;
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t i0 = a.x0 + (int64_t)blockIdx.x * H2_TX, j0 = a.y0 + (int64_t)blockIdx.y * H2_TY;
  const int ZC = a.zc > 0 ? a.zc : H2_ZC;
  const int64_t zc0 = a.s0 + (int64_t)blockIdx.z * ZC;
  const int64_t zc1 = (zc0 + ZC < a.s1) ? zc0 + ZC : a.s1;
  const int nsteps = (int)(zc1 - zc0) + 1;
  const int nvx = (a.x1 - i0 < H2_TX) ? (int)(a.x1 - i0) : H2_TX;
  const int nvy = (a.y1 - j0 < H2_TY) ? (int)(a.y1 - j0) : H2_TY;

  // ---- what wave 3 requests from global memory: one 16-byte load of coordinates and one of coefficients per lane and
  // step (unconditional, clamped addresses; what lies outside the mesh is never used) ----
  const bool loader = t >= 192;  // wave 3 (wave-uniform): the only wave that reads global memory
  const int ll = t - 192;
  // coordinates: the halo'd tile of a node layer is 4 rows of 30 consecutive doubles = 60 pairs
  const int64_t LX = 3 * M * N, xtotal = LX * (a.cke - s.cks);
  int64_t xoff;
  {
    const int cw = 2 * (ll < H2_NL * 3 / 2 ? ll : H2_NL * 3 / 2 - 1), r = cw / (3 * H2_NX), w = cw - 3 * H2_NX * r;
    int64_t gj = j0 - 1 + r;
    gj = gj < 0 ? 0 : (gj > N - 1 ? N - 1 : gj);
    xoff = 3 * ((i0 - 1) + M * gj) + w;
  }
  auto load_x = [&](int64_t k) -> double2 {
    const int64_t gk = k < s.cks ? s.cks : (k > a.cke - 1 ? a.cke - 1 : k);
    const int64_t off = xoff + LX * (gk - s.cks);
    // (a pair that starts on the array's last double is read one double lower and shifted)
    const double2_a8 v = *reinterpret_cast<const double2_a8 *>(a.coords + (off > xtotal - 2 ? xtotal - 2 : off));
    return make_double2(off == xtotal - 1 ? v.y : v.x, v.y);
  };
  // coefficients: lane 2e: (c11, c12), lane 2e + 1: (c44, -) of element e (k_s3_elem_coefs)
  const int64_t LC = 4 * (M - 1) * (N - 1);
  int64_t coff;
  {
    const int e = (ll >> 1) < H2_NS ? (ll >> 1) : H2_NS - 1;
    int64_t ex = i0 - 1 + (e % H2_SY), ey = j0 - 1 + (e / H2_SY);
    ex = ex < 0 ? 0 : (ex > M - 2 ? M - 2 : ex);
    ey = ey < 0 ? 0 : (ey > N - 2 ? N - 2 : ey);
    coff = 4 * (ex + (M - 1) * ey) + 2 * (ll & 1);
  }
  auto load_c = [&](int64_t lay) -> double2 {
    int64_t ez = lay > s.ele - 1 ? s.ele - 1 : lay;
    ez = ez < s.els ? s.els : ez;
    return *reinterpret_cast<const double2 *>(a.ecoef + coff + LC * (ez - s.els));
  };
  double2 px = make_double2(0.0, 0.0), pc = make_double2(0.0, 0.0);
  if (loader) {
    const double2 v0 = load_x(zc0 - 1), v1 = load_x(zc0), c0 = load_c(zc0 - 1);
    if (ll < H2_NL * 3 / 2) reinterpret_cast<double2 *>(sh_x[0])[ll] = v0, reinterpret_cast<double2 *>(sh_x[1])[ll] = v1;
    if (ll < 2 * H2_NS) reinterpret_cast<double2 *>(sh_c)[ll] = c0;
    px = load_x(zc0 + 1);
    pc = load_c(zc0);
  }
  __syncthreads();

  // the run of tile row 0 of the first node layer; rows j0, j0 + 1 are interior rows of an interior layer: their runs lie
  // 81 (3M - 2) values apart, and the same run of the next layer 27 (3M - 2)(3N - 2) values on
  double *runp = a.out + (row_start(s, i0, j0, zc0, 0) - a.base_off);
  const int64_t rowstride = 81 * (3 * M - 2), laystride = 27 * (3 * M - 2) * (3 * N - 2);

  // the roles: wave 0 A, waves 1 and 2 start / finish in turn, wave 3 D (wave-uniform, so scalar branches)
  const int grp = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool isA = grp == 0, isD = grp == 3, isM = !isA && !isD;
  double old[24];  // what this lane left for the next step (A: the dz = -1 third; start: the lower half of dz = 0)
#pragma unroll
  for (int q = 0; q < 24; q++) old[q] = 0.0;

#if H2_EXP == 9
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tk0 = __builtin_readcyclecounter();
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), ck0 = tk0;
#endif
  for (int st = 0; st < nsteps; st++) {
    const int64_t L = zc0 - 1 + st;  // element layer of this step: node layers L (sh_x[st & 1]) and L + 1
    const bool lay_ok = L >= s.els && L < s.ele;
    // (lane-derived indices are recomputed from an opaque copy of the lane id in every step: derived once before the
    // loop, the compiler keeps several dozen loop-invariant addresses alive and spills them -- DESIGN section 3)
    int tl = t;
    asm volatile("" : "+v"(tl));

    // ================= record phase: 8 Gauss points x 27 elements, one record per lane =================
    if (tl < H2_NREC) {
      const int rp_ = tl / H2_NS, re = tl - H2_NS * rp_;
      const int mex = re % H2_SY, mey = re / H2_SY;
      const int64_t gex = i0 - 1 + mex, gey = j0 - 1 + mey;
      const bool mxy_ok = gex >= 0 && gex < M - 1 && gey >= 0 && gey < N - 1;
      char *img = sh_rec + rp_ * H2_PS;
      double2 *dxy = reinterpret_cast<double2 *>(img) + h2_uxy(0, mex, mey);
      double *dz = reinterpret_cast<double *>(img + H2_OFF_Z) + h2_uz(0, mex, mey);
      double *dw = reinterpret_cast<double *>(img + H2_OFF_W) + re;
      // node n = 4 f + c of the element: face f, cell c -> unit f * 138 + CB[c], double f * 194 + CB'[c]
#define H2_XYI(n) (((n) >> 2) * 138 + ((((n) & 3) & 1) ? 24 : 0) + (((n) & 3) >> 1))
#define H2_ZI(n) (((n) >> 2) * 194 + ((((n) & 3) & 1) ? 112 : 0) + (((n) & 3) >> 1))
      if (!(mxy_ok && lay_ok) || H2_EXP == 3) {
#pragma unroll
        for (int n = 0; n < 8; n++) dxy[H2_XYI(n)] = make_double2(0.0, 0.0), dz[H2_ZI(n)] = 0.0;
        *dw = 0.0;
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
          const double *xn = (bz ? x1 : x0) + 3 * ((mex + bx) + H2_NX * (mey + by));
          const double c0 = xn[0], c1 = xn[1], c2 = xn[2];
          J[0][0] = J[0][0] + g0 * c0; J[0][1] = J[0][1] + g0 * c1; J[0][2] = J[0][2] + g0 * c2;
          J[1][0] = J[1][0] + g1 * c0; J[1][1] = J[1][1] + g1 * c1; J[1][2] = J[1][2] + g1 * c2;
          J[2][0] = J[2][0] + g2 * c0; J[2][1] = J[2][1] + g2 * c1; J[2][2] = J[2][2] + g2 * c2;
        }
        double iJ[3][3], det;
        inv3_ref(J, iJ, det);
        const double wJ = 1.0 * det;
        if (!(wJ > 0.0)) atomicAdd(a.bad, 1ULL);
#pragma unroll
        for (int n = 0; n < 8; n++) {
          const int bx = HDI[n], by = HDJ[n], bz = HDK[n];
          const double g0 = (HSX[n] * 0.125) * ay[by] * az[bz];
          const double g1 = (HSY[n] * 0.125) * ax[bx] * az[bz];
          const double g2 = (HSZ[n] * 0.125) * ax[bx] * ay[by];
          double g[3];
#pragma unroll
          for (int k = 0; k < 3; k++) g[k] = g0 * iJ[k][0] + g1 * iJ[k][1] + g2 * iJ[k][2];
          dxy[H2_XYI(n)] = make_double2(g[0], g[1]);
          dz[H2_ZI(n)] = g[2];
        }
        *dw = wJ;
      }
#undef H2_XYI
#undef H2_ZI
    }
    H2T(0);
    __syncthreads();  // B1: records and coefficients of element layer L are complete
    H2T(1);

    // ================= contraction: W[fx][fy][i][j] += (wJ ga_i) gb_j over the 8 Gauss points =================
    // roles: A and start serve node layer L + 1 from below, finish and D node layer L from above; waves 1 and 2 take
    // start and finish in turn (a node's dz = 0 third is started from below, finished from above).
    asm volatile("" : "+v"(tl));
    const int node = (tl & 63) >> 2;
    const int qx = tl & 1, qy = (tl >> 1) & 1;
    const int nxn = node % H2_TX, nyn = node / H2_TX;
    const int sex = nxn + qx, sey = nyn + qy;  // element (ox, oy) = (qx - 1, qy - 1) of the node
    const int slot = sex + H2_SY * sey;
    const bool fin = isM && ((grp == 1) == ((st & 1) != 0));
    const bool isS = isM && !fin;
    double W[4][9];
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
      for (int q = 0; q < 9; q++) W[b][q] = 0.0;
    {
      const int a_z = (isA || isS) ? 1 : 0;  // the node's face of the element
      const int b_z = (isS || isD) ? 1 : 0;  // the column nodes' face: A and finish the bottom, start and D the top
      // face-local cells: the node's own corner, and the four (near | far in x) x (near | far in y) of the column face
      const int ca = h2_quad(1 - qx, 1 - qy);
      const int cb[4] = {h2_quad(1 - qx, 1 - qy), h2_quad(qx, 1 - qy), h2_quad(1 - qx, qy), h2_quad(qx, qy)};  // [fx + 2 fy]
      const char *base = sh_rec;
      const int oa = a_z * H2_XY_B + 16 * h2_uxy(ca, sex, sey), oaz = H2_OFF_Z + a_z * H2_Z_B + 8 * h2_uz(ca, sex, sey);
      const int ow = H2_OFF_W + 8 * slot;
      int ob[4], obz[4];
#pragma unroll
      for (int b = 0; b < 4; b++)
        ob[b] = b_z * H2_XY_B + 16 * h2_uxy(cb[b], sex, sey), obz[b] = H2_OFF_Z + b_z * H2_Z_B + 8 * h2_uz(cb[b], sex, sey);
#if H2_EXP != 4
#pragma unroll H2_UNROLL
      for (int p = 0; p < 8; p++) {
        const char *rp = base + p * H2_PS;
        // all eleven reads of this Gauss point first, one wait, then the 39 products
        const double2 a0 = *reinterpret_cast<const double2 *>(rp + oa);
        const double a2 = *reinterpret_cast<const double *>(rp + oaz);
        const double wJ = *reinterpret_cast<const double *>(rp + ow);
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
      (void)base, (void)oa, (void)oaz, (void)ow;
#pragma unroll
      for (int b = 0; b < 4; b++)
#pragma unroll
        for (int q = 0; q < 9; q++) W[b][q] = (double)(ob[b] + obz[b] + q);
#endif
    }
    H2T(2);
    __syncthreads();  // B2: every lane has read its records; the stage may overwrite them
    H2T(3);
    // (material, quad sums and the choice of what a lane keeps come AFTER the barrier: they touch registers only)
    // the element's material: K = c12 W + c44 W^T off the diagonal, c11 W_ii + c44 (W_jj + W_kk) on it (in place)
    {
      const double c11 = sh_c[4 * slot], c12 = sh_c[4 * slot + 1], c44 = sh_c[4 * slot + 2];
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
    // lane bases in a third of the stage (row stride 81); far neighbour: nx 0 | 2 -> 3 nx, ny 0 | 2 -> 9 ny
    const int p_corner = 18 * qy + 6 * qx, p_xedge = 18 * qy + 3 + qx * 81, p_yedge = 9 + 6 * qx + qy * 81, p_centre = 12 + qc * 81;
    // entry q of nk: its base and its compile-time offset (i = 3 c + c' of the kept rows -> 81 c + c')
#define H2_POS(q) ((q) < 9 ? p_corner + ((q) / 3) * 81 + (q) % 3 : (q) < 15 ? p_xedge + (((q)-9) / 3) * 81 + ((q)-9) % 3 \
                   : (q) < 21 ? p_yedge + (((q)-15) / 3) * 81 + ((q)-15) % 3 : p_centre + ((q)-21))

    // ================= deposit: node layer L, in HBM order =================
    const bool out_layer = st > 0;  // node layer L = zc0 - 1 + st belongs to this workgroup
    double *stg = sh_stage + 16;    // (16 doubles of slack either side: the store loop reads whole chunks from a line boundary)
    // run rr of the tile starts at stg + rr * H2_RS + (head_rr & 1): a 16-byte chunk of HBM is a 16-byte chunk of LDS
    const int head0 = (int)((reinterpret_cast<uintptr_t>(runp) & 127) >> 3), head1 = (int)((reinterpret_cast<uintptr_t>(runp + rowstride) & 127) >> 3);
    // A writes what it kept from the last step (the dz = -1 third of layer L), finish adds what it kept as start to this
    // step's numbers (dz = 0), D writes this step's (dz = +1); start writes nothing.  A and start keep this step's
    // numbers for the next one (they belong to layer L + 1).
    if (out_layer && !isS) {
      double *d = stg + (nyn ? H2_RS + (head1 & 1) : (head0 & 1)) + nxn * H2_NB + (isA ? 0 : (isD ? 54 : 27));
      if (isA) {
#pragma unroll
        for (int q = 0; q < 24; q++) d[H2_POS(q)] = old[q];
      } else if (fin) {
#pragma unroll
        for (int q = 0; q < 24; q++) d[H2_POS(q)] = old[q] + nk[q];
      } else {
#pragma unroll
        for (int q = 0; q < 24; q++) d[H2_POS(q)] = nk[q];
      }
    }
#undef H2_POS
#pragma unroll
    for (int q = 0; q < 24; q++) old[q] = nk[q];
    H2T(4);
    __syncthreads();  // B3: the node layer is complete in the stage
    H2T(5);

    // ================= store: two runs of nvx nodes, whole 128-byte lines =================
    // The two runs are 2 x 16 rounds of 64 chunks (16 bytes per lane); waves 0, 1, 2 take every third round.  Wave 3 stores
    // nothing: it is the wave that loads, and gfx9 retires a wave's loads and stores through one in-order counter (with
    // a share of the stores its wait for the next layer's inputs cost 3 400 cycles per step).
    if (out_layer && grp < 3) {
      asm volatile("" : "+v"(tl));
      const int ct = tl & 63;
      const int n = nvx * H2_NB;
      constexpr int NR = (H2_RUN / 2 + 8 + 63) / 64;  // rounds per run
      constexpr int NIT = (2 * NR + 2) / 3;
      double2 v[NIT];
#pragma unroll
      for (int k = 0; k < NIT; k++) {
        const int r = grp + 3 * k, rr = r >= NR ? 1 : 0, it = r - NR * rr;  // (scalar)
        const int head = rr ? head1 : head0;                                // doubles above a line boundary
        // chunk q = the q-th 16 bytes from that boundary = run[2q - head, 2q - head + 1]
        const int s0 = 2 * (ct + 64 * it) - head;
        const double *run_l = stg + rr * H2_RS + (head & 1);
        v[k] = *reinterpret_cast<const double2 *>(run_l + (r < 2 * NR ? s0 : 0));  // 16-byte aligned: s0 + (head & 1) is even
      }
#pragma unroll
      for (int k = 0; k < NIT; k++) {
        const int r = grp + 3 * k, rr = r >= NR ? 1 : 0, it = r - NR * rr;
        const int head = rr ? head1 : head0;
        const int s0 = 2 * (ct + 64 * it) - head;
        double *rp = rr ? runp + rowstride : runp;
        // whole chunks here, the two possible half chunks (an odd head, an odd end) below
        const bool in = r < 2 * NR && rr < nvy && s0 >= 0 && s0 + 2 <= n;
#if H2_EXP == 1
        if (in) asm volatile("" ::"v"(v[k].x), "v"(v[k].y));
#else
        if (in) store_pair(rp + s0, v[k]);
#endif
      }
      if (grp < nvy) {  // wave rr: the half chunks of run rr
        const int head = grp ? head1 : head0;
        double *rp = grp ? runp + rowstride : runp;
        const double *run_l = stg + grp * H2_RS + (head & 1);
        if ((head & 1) && ct == 0) rp[0] = run_l[0];
        if (((head + n) & 1) && ct == 1) rp[n - 1] = run_l[n - 1];
      }
    }
    if (loader) {
      // wave 3, while the others store: the next step's coefficients (read after its B2) and node layer L + 2 over node
      // layer L (last read before B1), both requested a step ago; the next requests
      const int l = tl - 192;
      if (l < 2 * H2_NS) reinterpret_cast<double2 *>(sh_c)[l] = pc;
      if (l < H2_NL * 3 / 2) reinterpret_cast<double2 *>(sh_x[st & 1])[l] = px;
      pc = load_c(L + 2);
      px = load_x(L + 3);
    }
    if (out_layer) runp += laystride;
    H2T(6);
    __syncthreads();  // B4: the stage has been read; the next step's records may overwrite it
    H2T(7);
  }
#if H2_EXP == 9
  if (blockIdx.x == 3 && blockIdx.y == 7 && blockIdx.z == 0 && (t == 0 || t == 128)) {
    for (int q = 0; q < 8; q++) atomicAdd(a.bad + 8 + (t ? 8 : 0) + q, tk[q]);
    if (t == 0) atomicAdd(a.bad + 24, (unsigned long long)nsteps);
  }
  if (t == 0) {  // wall time of the workgroups (100 MHz ticks): sum of durations, count, and the sampled launch's span
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    atomicAdd(a.bad + 26, rt1 - rt0);
    atomicAdd(a.bad + 27, 1ULL);
    atomicAdd(a.bad + 28, __builtin_readcyclecounter() - ck0);
  }
#endif
}

}  // namespace femasm
