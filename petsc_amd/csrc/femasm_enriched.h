// femasm_enriched.h -- config 5: SGFEM-enriched Q1 hexahedra (node-blocked D = 3 (1 + n_enr) dofs per node).
//
// The reference holds no arithmetic for this configuration: R3 has the bookkeeping only
// (R3:39-49 macros and flags, R3:52-68 record, R3:1265-1269 zero fill, R3:1467-1595 dead BlockRedundantDofs).  The
// specification is this build's own, DESIGN.md section 9 (the tests hold an independent CPU restatement of it).
//
// Scalar shape functions of an element: Phi_{a,0} = N_a, Phi_{a,1+m} = N_a phi_m with phi_m = psi_m - I_h psi_m
// (SGFEM, `type` 1 of R3:42) or psi_m (GFEM); fn = 8 s + a.  Every 3x3 block of the element matrix is the S3 formula
// (S3:450-518) on the gradients of two scalar functions, so the local matrix is (8 S)^2 such blocks.
//
// Two kernels (north_star (3) + (4): local B^T D B contraction, then a deterministic segmented reduction):
//   k_enr_elem<S, MFMA>   one wave per element.  Lanes (Gauss point p, node a) build the gradient table
//                         T[p][fn][3] and the w detJ scaled (D11, D12, D44) in LDS; then the contraction
//                           VALU: lane = node pair (a, b), all S^2 (s,t) blocks of that pair, 21 FMAs per block and point
//                           MFMA: S = 2 only: the 16 scalar functions are exactly one v_mfma_f64_16x16x4 tile; block
//                                 (r,c) of all 256 function pairs = sum over the K index (p, term) of
//                                 A = grad_r Phi_i, B = D12 grad_c Phi_j  and  A = grad_c Phi_i, B = D44 grad_r Phi_j
//                                 (diagonal: D11 and two D44 terms): 42 MFMAs per element, operands straight from T
//                         the element matrix goes to a scratch ring in the layout [element][a][s][r][c][j]: row node a, row
//                         family s, row component r, column component c, column function j = 8 t + b -- a row node's
//                         chunk is 8 D^2 contiguous doubles and, for S = 2, every (s, r, c) is one 128-byte line of 16
//                         column functions, which is exactly what 16 lanes of an MFMA result register (or of the VALU
//                         variant, which uses the same lane map) hold: each store instruction writes four whole lines
//                         straight from registers, no staging.  Other S: through LDS, 16 B per lane.
//   k_enr_gather<D>       one wave per node: adds the <= 8 chunks [a] of the elements around the node (each a contiguous
//                         8 D^2 doubles) into the node's CSR block in LDS, in the reference's element order
//                         (k -> j -> i: MatSetValuesStencil ADD order, S3:687-705), applies the row rules (Dirichlet faces,
//                         blocked redundant dofs) and writes the block, which is contiguous in the CSR array.
// Every CSR value has one writer, no atomics on values, bitwise reproducible.
#pragma once
#include "femasm_kernels.h"
#include "femasm_sweep.h"

namespace femasm {

enum { KIND_ENRICHED = 3 };
constexpr int ENR_MAX_S = 4;  // 1 + n_enr families supported by the instantiations below

struct EnrArgs {
  const double *coords;   // node layers [cks, cke)
  const double *gp;       // R3 records + enrichment blocks, stride doubles per element (never null)
  double *ke;             // element-matrix ring: slot = element layer % nring
  double *out;            // gather: CSR values of the processed rows / b
  double *elemF;          // RHS element vectors
  unsigned long long *bad;
  int64_t base_off;
  int64_t cke;
  int64_t L0, L1;         // k_enr_elem: element layers of this launch
  int64_t s0, s1;         // gather: node layers of this launch
  int32_t nring;
  int32_t stride, base;   // record stride and width of the R3 part (408: 32-bit PetscInt, 416: 64-bit)
  int32_t sgfem;          // 1: phi = psi - I_h psi (R3:42 `type` 1), 0: phi = psi
  int32_t bc, iface;
  int32_t npw;            // k_enr_box: consecutive nodes per wave
  double E, nu;           // FEMASM_MAT_UNIFORM is not offered for this kind (the enrichment data lives in the records)
};

__device__ inline int64_t enr_flag(const double *rec, int base, int idx) {
  // PetscInt enrich_type[8], n_enrich[8] inside a struct of PetscScalar, at double 400 (R3:65-67)
  if (base == 408) return (int64_t)reinterpret_cast<const int32_t *>(rec + 400)[idx];
  return reinterpret_cast<const int64_t *>(rec + 400)[idx];
}
// number of active families at local node a of the element whose record is rec
__device__ inline int enr_active(const double *rec, int base, int a, int n_enr) {
  if (n_enr <= 0) return 0;  // no families: the flags are not read (a 400-stride S3 record has none: double 400 is the next record)
  const int64_t t = enr_flag(rec, base, a), n = enr_flag(rec, base, 8 + a);
  if (t == 0 || n <= 0) return 0;
  return n < n_enr ? (int)n : n_enr;
}

template <int S> struct EnrCfg {
  static constexpr int NF = 8 * S, D = 3 * S, KE = 64 * D * D;
  static constexpr bool DIRECT = (S == 2);  // results go from registers to the ring (see the layout note above)
  static constexpr int NW = (S <= 2) ? 4 : (S == 3 ? 2 : 1);  // waves (= elements) per workgroup, bounded by the LDS staging
};
// position of (row family s, row component r, column component c, column function j) inside a row node's chunk
template <int NF> __device__ inline int enr_pos(int sa, int r, int c, int j) { return ((sa * 3 + r) * 3 + c) * NF + j; }

typedef double d4v __attribute__((ext_vector_type(4)));

template <int S, bool MFMA>
__global__ __launch_bounds__(64 * EnrCfg<S>::NW) void k_enr_elem(const Slab s, const EnrArgs a) {
  using Cf = EnrCfg<S>;
  constexpr int NF = Cf::NF, D = Cf::D, KE = Cf::KE, NW = Cf::NW, NE = S - 1;
  static_assert(!MFMA || S == 2, "the FP64 MFMA contraction is the 16-function tile of one enrichment family");
  __shared__ __attribute__((aligned(16))) double sh_ke[NW][Cf::DIRECT ? 2 : KE];
  __shared__ double sh_T[NW][8][NF][3];
  __shared__ double sh_cf[NW][8][4];
  __shared__ double sh_x[NW][24];
  __shared__ double sh_en[NW][NE > 0 ? NE : 1][40];  // psi at nodes [8], at Gauss points [8], grad psi [8][3]
  __shared__ int sh_act[NW][8];
  __shared__ double sh_G[8][3][8], sh_N[8][8];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int t = tid; t < 192; t += 64 * NW) (&sh_G[0][0][0])[t] = (&c_tab3.G[0][0][0])[t];
  for (int t = tid; t < 64; t += 64 * NW) (&sh_N[0][0])[t] = (&c_tab3.N[0][0])[t];
  __syncthreads();  // (the only workgroup barrier: every wave is still here)

  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t mx = M - 1, my = N - 1;
  const int64_t e = (int64_t)blockIdx.x * NW + w;  // element of this launch, k -> j -> i order
  if (e >= mx * my * (a.L1 - a.L0)) return;
  const int64_t ex = e % mx, ey = (e / mx) % my, ez = a.L0 + e / (mx * my);
  const double *rec = a.gp + (ex + mx * (ey + my * (ez - s.mat_e_ks))) * (int64_t)a.stride;
  auto wave_sync = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };

  // ---- inputs of the element -> LDS ----
  if (lane < 24) {
    const int n = lane / 3, d = lane - 3 * n;
    const int64_t node = (ex + HDI[n]) + M * ((ey + HDJ[n]) + N * (ez + HDK[n] - s.cks));
    sh_x[w][lane] = a.coords[3 * node + d];
  }
  if (lane < 8) sh_act[w][lane] = enr_active(rec, a.base, lane, NE);
  if constexpr (NE > 0) {
    if (lane < 40)
#pragma unroll
      for (int m = 0; m < NE; m++) sh_en[w][m][lane] = rec[a.base + 64 * m + lane];
  }
  wave_sync();

  // ---- lanes (p, n): Jacobian of Gauss point p, physical gradient of N_n, scaled coefficients ----
  const int p = lane >> 3, n = lane & 7;
  double gN[3];
  {
    double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
    for (int q = 0; q < 8; q++)
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) J[r][c] = J[r][c] + sh_G[p][r][q] * sh_x[w][3 * q + c];
    double iJ[3][3], det;
    inv3_ref(J, iJ, det);
    const double wJ = 1.0 * det;
#pragma unroll
    for (int k = 0; k < 3; k++) gN[k] = sh_G[p][0][n] * iJ[k][0] + sh_G[p][1][n] * iJ[k][1] + sh_G[p][2][n] * iJ[k][2];
#pragma unroll
    for (int k = 0; k < 3; k++) sh_T[w][p][n][k] = gN[k];
    if (n == 0) {
      if (!(wJ > 0.0)) atomicAdd(a.bad, 1ULL);
      double d11, d12, d44;
      scaled_coefs<KIND_ELASTIC>(rec[96 + p], rec[104 + p], wJ, d11, d12, d44);
      sh_cf[w][p][0] = d11;
      sh_cf[w][p][1] = d12;
      sh_cf[w][p][2] = d44;
    }
  }
  if constexpr (NE > 0) {
    wave_sync();  // the gradients of all eight N_b at p are in T
    const int nact = sh_act[w][n];
    const double Nn = sh_N[p][n];
#pragma unroll
    for (int m = 0; m < NE; m++) {
      const double *en = sh_en[w][m];
      double phi = en[8 + p], dphi[3] = {en[16 + 3 * p], en[16 + 3 * p + 1], en[16 + 3 * p + 2]};
      if (a.sgfem) {
#pragma unroll
        for (int b = 0; b < 8; b++) {
          phi = phi - sh_N[p][b] * en[b];
#pragma unroll
          for (int k = 0; k < 3; k++) dphi[k] = dphi[k] - sh_T[w][p][b][k] * en[b];
        }
      }
      const bool on = m < nact;
#pragma unroll
      for (int k = 0; k < 3; k++) sh_T[w][p][8 * (1 + m) + n][k] = on ? phi * gN[k] + Nn * dphi[k] : 0.0;
    }
  }
  wave_sync();

  // ---- contraction: 3x3 block of every pair of scalar functions ----
  const int slot = (int)(ez % a.nring);
  double *kg = a.ke + ((slot * my + ey) * mx + ex) * (int64_t)KE;  // this element's matrix in the ring
  constexpr int CH = 8 * D * D;                                     // doubles of a row node's chunk
  if constexpr (Cf::DIRECT) {
    // lane l: column function j = l & 15; row functions i = (l >> 4) + 4 g, g < 4 (the C/D map of v_mfma_f64_16x16x4)
    const int j = lane & 15, ib = lane >> 4;
    d4v acc[3][3];
    if constexpr (MFMA) {
      // A and B operands of lane l: function l & 15 at Gauss point (l >> 4) + 4 h of K-step h
      double ag[2][3], b11[2][3], b12[2][3], b44[2][3];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int q = ib + 4 * h;
        const double d11 = sh_cf[w][q][0], d12 = sh_cf[w][q][1], d44 = sh_cf[w][q][2];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const double g = sh_T[w][q][j][k];
          ag[h][k] = g;
          b11[h][k] = d11 * g;
          b12[h][k] = d12 * g;
          b44[h][k] = d44 * g;
        }
      }
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
          d4v v = {0.0, 0.0, 0.0, 0.0};
          if (r != c) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
              v = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[h][r], b12[h][c], v, 0, 0, 0);  // D12 g_i[r] g_j[c]
              v = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[h][c], b44[h][r], v, 0, 0, 0);  // D44 g_i[c] g_j[r]
            }
          } else {
#pragma unroll
            for (int h = 0; h < 2; h++) {
              v = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[h][r], b11[h][r], v, 0, 0, 0);
#pragma unroll
              for (int k = 0; k < 3; k++)
                if (k != r) v = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[h][k], b44[h][k], v, 0, 0, 0);
            }
          }
          acc[r][c] = v;
        }
    } else {
      double av[4][9];
#pragma unroll
      for (int g = 0; g < 4; g++)
#pragma unroll
        for (int q = 0; q < 9; q++) av[g][q] = 0.0;
#pragma unroll 1
      for (int q = 0; q < 8; q++) {
        const double d11 = sh_cf[w][q][0], d12 = sh_cf[w][q][1], d44 = sh_cf[w][q][2];
        const double *gjp = sh_T[w][q][j];
        const double gj[3] = {gjp[0], gjp[1], gjp[2]};
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const double *ga = sh_T[w][q][ib + 4 * g];
          double l[3], m[3], qq[3];
#pragma unroll
          for (int k = 0; k < 3; k++) {
            l[k] = d12 * ga[k];
            m[k] = d44 * ga[k];
            qq[k] = d11 * ga[k];
          }
          add_block<KIND_ELASTIC, 3>(av[g], l, m, qq, gj);
        }
      }
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
          d4v v = {av[0][3 * r + c], av[1][3 * r + c], av[2][3 * r + c], av[3][3 * r + c]};
          acc[r][c] = v;
        }
    }
    // register g of lane l: row function i = ib + 4 g = 8 s + a.  One store instruction = 4 whole 128-byte lines.
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int i = ib + 4 * g, sa = i >> 3, na = i & 7;
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
#if FEMASM_EXP == 1  // profiling build: the contraction without its stores
          asm volatile("" ::"v"(acc[r][c][g]));
#else
          kg[na * CH + enr_pos<NF>(sa, r, c, j)] = acc[r][c][g];
#endif
        }
    }
  } else {
    double *ke = sh_ke[w];
    const int na = lane >> 3, nb = lane & 7;  // node pair (a, b)
#pragma unroll 1
    for (int sa = 0; sa < S; sa++) {
      double acc[S][9];
#pragma unroll
      for (int t = 0; t < S; t++)
#pragma unroll
        for (int q = 0; q < 9; q++) acc[t][q] = 0.0;
#pragma unroll 1
      for (int q = 0; q < 8; q++) {
        const double d11 = sh_cf[w][q][0], d12 = sh_cf[w][q][1], d44 = sh_cf[w][q][2];
        const double *ga = sh_T[w][q][8 * sa + na];
        double l[3], m[3], qq[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          l[k] = d12 * ga[k];
          m[k] = d44 * ga[k];
          qq[k] = d11 * ga[k];
        }
#pragma unroll
        for (int t = 0; t < S; t++) {
          const double *gbp = sh_T[w][q][8 * t + nb];
          const double gb[3] = {gbp[0], gbp[1], gbp[2]};
          add_block<KIND_ELASTIC, 3>(acc[t], l, m, qq, gb);
        }
      }
#pragma unroll
      for (int t = 0; t < S; t++)
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
          for (int cc = 0; cc < 3; cc++) ke[na * CH + enr_pos<NF>(sa, c, cc, 8 * t + nb)] = acc[t][3 * c + cc];
    }
    wave_sync();
    // staging -> scratch ring, 16 B per lane, whole lines (KE * 8 bytes is a multiple of 128)
    double2 *dst = reinterpret_cast<double2 *>(kg);
    const double2 *src = reinterpret_cast<const double2 *>(ke);
#pragma unroll 2
    for (int t = lane; t < KE / 2; t += 64) dst[t] = src[t];
  }
}

// ---------------------------------------------------------------------------
// Segmented reduction: one wave per node of node layers [s0, s1).  D = DMAX is a compile-time number (the index
// arithmetic of the deposits is divisions by D and 8 S: with a runtime D they cost more than the memory traffic).
// ---------------------------------------------------------------------------
constexpr int ENR_GW = 4;  // waves per workgroup
template <int DMAX>
__global__ __launch_bounds__(64 * ENR_GW) void k_enr_gather(const Slab s, const EnrArgs a) {
  __shared__ double sh_acc[ENR_GW][27 * DMAX * DMAX];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t M = s.dim[0], N = s.dim[1], P = s.dim[2];
  constexpr int D = DMAX;
  const int64_t q = (int64_t)blockIdx.x * ENR_GW + w;
  if (q >= M * N * (a.s1 - a.s0)) return;
  const int64_t i = q % M, j = (q / M) % N, k = a.s0 + q / (M * N);
  const int64_t mx = M - 1, my = N - 1;
  const int cntx = span1(i, M), cnty = span1(j, N), cntz = span1(k, P);
  const int rowlen = cntx * cnty * cntz * D, nvals = D * rowlen, KN = 8 * D * D;
  const int xi0 = (i == 0), yi0 = (j == 0), zi0 = (k == 0);
  double *acc = sh_acc[w];
  auto wave_sync = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  for (int t = lane; t < nvals; t += 64) acc[t] = 0.0;
  wave_sync();
  const bool dirichlet = (a.bc == BC_S3) && on_boundary(s, i, j, k);
  constexpr int NF = 8 * (D / 3);
  // chunk position t -> (row dof, neighbour, column dof) -> position in the node's CSR block
  auto deposit = [&](int t, int ox, int oy, int oz, double v) {
    const int sa = t / (9 * NF), rem = t - sa * 9 * NF, r = rem / (3 * NF), rem2 = rem - r * 3 * NF, c = rem2 / NF;
    const int jf = rem2 - c * NF, tt = jf >> 3, b = jf & 7;
    const int dx = ox + HDI[b], dy = oy + HDJ[b], dz = oz + HDK[b];
    const int nbr = ((dz + 1 - zi0) * cnty + (dy + 1 - yi0)) * cntx + (dx + 1 - xi0);
    const int pos = (3 * sa + r) * rowlen + nbr * D + 3 * tt + c;
    acc[pos] = acc[pos] + v;
  };
  if (!dirichlet) {
    if constexpr (DMAX <= 6) {
      // all eight chunks are requested before the first is added (memory-level parallelism: 18 KB per wave in flight),
      // then added in the reference's element order (S3:687-689)
      constexpr int NIT = (8 * DMAX * DMAX + 63) / 64;
      double v[8][NIT];
#pragma unroll
      for (int q8 = 0; q8 < 8; q8++) {
        const int ox = (q8 & 1) - 1, oy = ((q8 >> 1) & 1) - 1, oz = (q8 >> 2) - 1;
        const int64_t ex = i + ox, ey = j + oy, ez = k + oz;
        const bool ok = !(ex < 0 || ex >= mx || ey < 0 || ey >= my || ez < s.els || ez >= s.ele);
        const int64_t exc = ok ? ex : 0, eyc = ok ? ey : 0, ezc = ok ? ez : s.els;
        const int slot = (int)(ezc % a.nring);
        const double *chunk = a.ke + (((slot * my + eyc) * mx + exc) * (int64_t)8 + hex_idx(-ox, -oy, -oz)) * KN;
#pragma unroll
        for (int it = 0; it < NIT; it++) {
          const int t = lane + 64 * it;
          v[q8][it] = (ok && t < KN) ? chunk[t] : 0.0;
        }
      }
#pragma unroll
      for (int q8 = 0; q8 < 8; q8++) {
        const int ox = (q8 & 1) - 1, oy = ((q8 >> 1) & 1) - 1, oz = (q8 >> 2) - 1;
        const int64_t ex = i + ox, ey = j + oy, ez = k + oz;
        if (ex < 0 || ex >= mx || ey < 0 || ey >= my || ez < s.els || ez >= s.ele) continue;
#pragma unroll
        for (int it = 0; it < NIT; it++) {
          const int t = lane + 64 * it;
          if (t < KN) deposit(t, ox, oy, oz, v[q8][it]);
        }
        wave_sync();  // the next element's read-modify-writes come after these
      }
    } else {
#pragma unroll 1
      for (int q8 = 0; q8 < 8; q8++) {
        const int ox = (q8 & 1) - 1, oy = ((q8 >> 1) & 1) - 1, oz = (q8 >> 2) - 1;
        const int64_t ex = i + ox, ey = j + oy, ez = k + oz;
        if (ex < 0 || ex >= mx || ey < 0 || ey >= my || ez < s.els || ez >= s.ele) continue;
        const int slot = (int)(ez % a.nring);
        const double *chunk = a.ke + (((slot * my + ey) * mx + ex) * (int64_t)8 + hex_idx(-ox, -oy, -oz)) * KN;
        for (int t = lane; t < KN; t += 64) deposit(t, ox, oy, oz, chunk[t]);
        wave_sync();
      }
    }
  }
  // row rules: Dirichlet faces (all dofs of the node), blocked redundant dofs (enriched dofs of inactive families)
  const int self = (((k > 0) * cnty + (j > 0)) * cntx + (i > 0)) * D;
  int nact = 0;
  if (!dirichlet && D > 3 && !a.iface) {
    const int64_t bx = i > 0 ? i - 1 : 0, by = j > 0 ? j - 1 : 0, bz = k > 0 ? k - 1 : 0;  // the element behind the node
    const double *rec = a.gp + (bx + mx * (by + my * (bz - s.mat_e_ks))) * (int64_t)a.stride;
    nact = enr_active(rec, a.base, hex_idx(i > 0, j > 0, k > 0), D / 3 - 1);
  }
  const double diag = a.iface ? 0.0 : 1.0;
  if (dirichlet || (D > 3 && !a.iface)) {
    for (int t = lane; t < nvals; t += 64) {
      const int qr = t / rowlen, c = t - qr * rowlen;
      const bool unit = dirichlet || (qr >= 3 && (qr / 3 - 1) >= nact);
      if (unit) acc[t] = (c == self + qr) ? diag : 0.0;
    }
  }
  wave_sync();
  double *o = a.out + (row_start(s, i, j, k, 0) - a.base_off);
  for (int t = lane; t < nvals; t += 64) o[t] = acc[t];
}

// ---------------------------------------------------------------------------
// One pass, node-centric: k_enr_node<S>.  The two kernels above move every element matrix through HBM (18 KB out and
// 18 KB back per element for S = 2: 5.4x the bytes of the CSR values they end up in).  Here one wave owns one node and
// nothing but the node's CSR block leaves the chip:
//   phase 1  lane (e, p): element e of the 8 around the node (k -> j -> i order), Gauss point p.  Jacobian, inverse
//            (reference order), physical gradients of the eight N_b -> T[e][p][b][3]; w detJ scaled (D11, D12, D44),
//            phi and grad phi of every family, and the gradients of the ROW node's S functions -> U[e][p][..]
//            (an element outside the mesh or outside the layers this rank assembles gets zero coefficients)
//   phase 2  lane (e, b): column node b of element e.  All S x S blocks (row family s, column family t) of the pair
//            (this node, b) over the eight Gauss points: 21 FMAs per block and point, accumulators in registers
//   phase 3  per row family (round 3): lane = two bits per axis (element side, node side); the two lanes that meet in a
//            neighbour are added in registers along x (quad permute) and y (row shift) and in LDS along z (the lanes below
//            the node write, the lanes above add in place): a fixed tree, one writer per slot of the node's CSR block in
//            LDS (rounds 1-2: a stage and eight ordered rounds); row rules (Dirichlet faces, blocked redundant dofs); the
//            block is contiguous in the CSR array and is written 16 B per lane from the 128-byte boundary below it.
//   ONEMAT   (level 1 of this kind) one (E, nu) per element, tested bitwise in phase 1 (bad[1]): the row gradients carry
//            w detJ, phase 2 accumulates W^{st}[i][j] with 9 FMAs per block and point, the material is applied once.
// Each element's geometry is computed by the eight waves of its nodes (8 x 64 Jacobians per node, ~20 % of the
// arithmetic); in exchange the kernel's HBM traffic is the CSR values plus the (cached) records.
// Every CSR value has one writer and a fixed order of additions: bitwise reproducible.
// ---------------------------------------------------------------------------
template <int S> struct EnrNodeCfg {
  static constexpr int NE = S - 1, D = 3 * S;
  static constexpr int SB = (S <= 2) ? S : 1;          // row families per contraction pass (9 S SB accumulators)
  static constexpr bool ALIAS = (SB == S);             // the node's block reuses the T/U area (single pass only)
  static constexpr int UW = 3 + 4 * NE + 3 * S;        // coefficients, (phi, grad phi) per family, row gradients
  static constexpr int UWP = (UW + 1) & ~1;
  static constexpr int TPS = 25, TES = 8 * TPS;        // T: 25 doubles per Gauss point (24 gradients + the activity flag
                                                       // of node p), 200 per element: with these strides the 8-byte
                                                       // accesses of 16 consecutive lanes fall on 16 distinct bank pairs
                                                       // both when lanes (e, p) write and when lanes (e, b) read
  static constexpr int UES = 8 * UWP + 2;
  static constexpr int IN = 8 * TES + 8 * UES;
  static constexpr int STG = 0, ACC = 27 * D * D;      // (no stage since round 3: the lanes of a neighbour meet in registers)
  static constexpr int OUT = (STG + ACC + 1) & ~1;
  static constexpr int PER_WAVE = ALIAS ? (IN > OUT ? IN : OUT) : IN + OUT;
  static constexpr int NW = (S <= 2) ? 8 : 2;          // waves per workgroup (LDS: 20 KB per wave for S = 2)
  static constexpr int NPW = 8;                        // consecutive nodes per wave
};

template <int S, bool ONEMAT>
__global__ __launch_bounds__(64 * EnrNodeCfg<S>::NW) void k_enr_node(const Slab s, const EnrArgs a) {
  using Cf = EnrNodeCfg<S>;
  constexpr int NE = Cf::NE, D = Cf::D, NW = Cf::NW, SB = Cf::SB, TES = Cf::TES, TPS = Cf::TPS, UES = Cf::UES, UWP = Cf::UWP;
  constexpr int RO = 3 + 4 * NE;  // RO: row gradients inside a U record
  __shared__ __attribute__((aligned(16))) double sh_w[NW][Cf::PER_WAVE];
  __shared__ double sh_G[8][3][8], sh_N[8][8];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int t = tid; t < 192; t += 64 * NW) (&sh_G[0][0][0])[t] = (&c_tab3.G[0][0][0])[t];
  for (int t = tid; t < 64; t += 64 * NW) (&sh_N[0][0])[t] = (&c_tab3.N[0][0])[t];
  __syncthreads();  // (the only workgroup barrier)
  double *T = sh_w[w], *U = T + 8 * TES;
  // activity flag of node n of element el: the 25th double of its T record of Gauss point n
  auto act_of = [&](int el, int n) { return reinterpret_cast<int *>(T + el * TES + n * TPS + 24); };
  double *stg = Cf::ALIAS ? T : U + 8 * UES, *acc = stg + Cf::STG;
  auto wave_sync = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  const int64_t M = s.dim[0], N = s.dim[1], P = s.dim[2];
  const int64_t mx = M - 1, my = N - 1;
  // lane = (element e of the 8 around the node, column node b); phase 1: Gauss point p = b.  Two bits per axis, (element
  // side, node side) = f >> 1, f & 1: the two lanes that meet in the neighbour at offset 0 of an axis are f = 1 and f = 2 of
  // that field -- one quad permute along x, one row shift along y (phase 3, as in k_enr_box)
  const int fx = lane & 3, fy = (lane >> 2) & 3, fz = lane >> 4;
  const int e = (fx >> 1) + 2 * (fy >> 1) + 4 * (fz >> 1), b = hex_idx(fx & 1, fy & 1, fz & 1);
  const int ox = (e & 1) - 1, oy = ((e >> 1) & 1) - 1, oz = (e >> 2) - 1;
  const int arow = 4 * (-oz) + ((-ox) == 0 ? ((-oy) == 0 ? 0 : 1) : ((-oy) == 0 ? 3 : 2));  // hex_idx(-ox, -oy, -oz)
  const int64_t nn = M * N * (a.s1 - a.s0);
  const int64_t q0 = ((int64_t)blockIdx.x * NW + w) * Cf::NPW;
#pragma unroll 1
  for (int64_t q = q0; q < q0 + Cf::NPW && q < nn; q++) {
    const int64_t i = q % M, j = (q / M) % N, k = a.s0 + q / (M * N);
    const int cntx = span1(i, M), cnty = span1(j, N), cntz = span1(k, P);
    const int rowlen = cntx * cnty * cntz * D, nvals = D * rowlen;
    const int xi0 = (i == 0), yi0 = (j == 0), zi0 = (k == 0);
    const bool dirichlet = (a.bc == BC_S3) && on_boundary(s, i, j, k);

    // ---- phase 1: lane (e, p) ----
    const int64_t ex = i + ox, ey = j + oy, ez = k + oz;
    const bool ok = !(ex < 0 || ex >= mx || ey < 0 || ey >= my || ez < s.els || ez >= s.ele);
    const int64_t exc = ex < 0 ? 0 : (ex >= mx ? mx - 1 : ex), eyc = ey < 0 ? 0 : (ey >= my ? my - 1 : ey);
    const int64_t ezc = ez < s.els ? s.els : (ez >= s.ele ? s.ele - 1 : ez);
    const double *rec = a.gp + (exc + mx * (eyc + my * (ezc - s.mat_e_ks))) * (int64_t)a.stride;
    int actp = 0;
    {
      const int p = b;
      double xs[24];
#pragma unroll
      for (int n = 0; n < 8; n++) {
        const double *xp = a.coords + 3 * ((exc + HDI[n]) + M * ((eyc + HDJ[n]) + N * (ezc + HDK[n] - s.cks)));
        xs[3 * n] = xp[0];
        xs[3 * n + 1] = xp[1];
        xs[3 * n + 2] = xp[2];
      }
      const double Ep = rec[96 + p], nup = rec[104 + p];
      double psin[NE > 0 ? NE : 1][8], psig[NE > 0 ? NE : 1], dpsi[NE > 0 ? NE : 1][3];
      if constexpr (NE > 0) {
        actp = enr_active(rec, a.base, p, NE);
#pragma unroll
        for (int m = 0; m < NE; m++) {
          const double *en = rec + a.base + 64 * m;
#pragma unroll
          for (int n = 0; n < 8; n++) psin[m][n] = en[n];
          psig[m] = en[8 + p];
#pragma unroll
          for (int kk = 0; kk < 3; kk++) dpsi[m][kk] = en[16 + 3 * p + kk];
        }
        *act_of(e, p) = actp;
      }
      double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
      for (int n = 0; n < 8; n++)
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) J[r][c] = J[r][c] + sh_G[p][r][n] * xs[3 * n + c];
      double iJ[3][3], det;
      inv3_ref(J, iJ, det);
      const double wJ = 1.0 * det;
      // the element is counted once, by the wave of its local node 6 (always an owned node of this rank)
      if (ok && e == 0 && !a.iface && !(wJ > 0.0)) atomicAdd(a.bad, 1ULL);
      double gN[8][3];
      double *Tp = T + e * TES + p * TPS;
#pragma unroll
      for (int n = 0; n < 8; n++)
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
          gN[n][kk] = sh_G[p][0][n] * iJ[kk][0] + sh_G[p][1][n] * iJ[kk][1] + sh_G[p][2][n] * iJ[kk][2];
          Tp[3 * n + kk] = gN[n][kk];
        }
      double *Up = U + e * UES + p * UWP;
      double d11 = 0.0, d12 = 0.0, d44 = 0.0;
      if constexpr (ONEMAT) {
        // level 1: one (E, nu) per element (every lane compares its Gauss point's pair with that of point 0; bad[1] ->
        // the general form); the coefficients stay unscaled and w detJ goes into the row node's gradients below
        if (ok && !(Ep == rec[96] && nup == rec[104]) && !a.iface) atomicAdd(a.bad + 1, 1ULL);
        scaled_coefs<KIND_ELASTIC>(Ep, nup, 1.0, d11, d12, d44);
      } else {
        if (ok) scaled_coefs<KIND_ELASTIC>(Ep, nup, wJ, d11, d12, d44);
      }
      Up[0] = d11;
      Up[1] = d12;
      Up[2] = d44;
      if constexpr (NE > 0) {
#pragma unroll
        for (int m = 0; m < NE; m++) {
          double phi = psig[m], dphi[3] = {dpsi[m][0], dpsi[m][1], dpsi[m][2]};
          if (a.sgfem) {
#pragma unroll
            for (int n = 0; n < 8; n++) {
              phi = phi - sh_N[p][n] * psin[m][n];
#pragma unroll
              for (int kk = 0; kk < 3; kk++) dphi[kk] = dphi[kk] - gN[n][kk] * psin[m][n];
            }
          }
          Up[3 + 4 * m] = phi;
#pragma unroll
          for (int kk = 0; kk < 3; kk++) Up[4 + 4 * m + kk] = dphi[kk];
        }
      }
      wave_sync();  // T and the activity flags of all eight nodes of the element are in LDS
      const double g0 = Tp[3 * arow], g1 = Tp[3 * arow + 1], g2 = Tp[3 * arow + 2];
      const double wz = ONEMAT ? (ok ? wJ : 0.0) : 1.0;  // (level 0: w detJ is in the coefficients)
      Up[RO] = wz * g0;
      Up[RO + 1] = wz * g1;
      Up[RO + 2] = wz * g2;
      if constexpr (NE > 0) {
        const int actr = *act_of(e, arow);
        const double Nr = sh_N[p][arow];
#pragma unroll
        for (int m = 0; m < NE; m++) {
          const double phi = Up[3 + 4 * m];
          const bool on = m < actr;
          Up[RO + 3 * (1 + m)] = on ? wz * (phi * g0 + Nr * Up[4 + 4 * m]) : 0.0;
          Up[RO + 3 * (1 + m) + 1] = on ? wz * (phi * g1 + Nr * Up[5 + 4 * m]) : 0.0;
          Up[RO + 3 * (1 + m) + 2] = on ? wz * (phi * g2 + Nr * Up[6 + 4 * m]) : 0.0;
        }
      }
    }
    wave_sync();

    const int self = (((k > 0) * cnty + (j > 0)) * cntx + (i > 0)) * D;
    if (!Cf::ALIAS) {
      for (int t = lane; t < nvals; t += 64) acc[t] = 0.0;
    }
    if (!dirichlet) {
#pragma unroll 1
      for (int s0 = 0; s0 < S; s0 += SB) {
        // ---- phase 2: lane (e, b), row families [s0, s0 + SB) ----
        double av[SB][S][9];
#pragma unroll
        for (int sa = 0; sa < SB; sa++)
#pragma unroll
          for (int t = 0; t < S; t++)
#pragma unroll
            for (int v = 0; v < 9; v++) av[sa][t][v] = 0.0;
#if FEMASM_EXP == 13 || FEMASM_EXP == 14  // profiling build: no contraction
#pragma unroll 1
        for (int p = 0; p < 0; p++) {
#else
#pragma unroll 2
        for (int p = 0; p < 8; p++) {
#endif
          const double *Tp = T + e * TES + p * TPS + 3 * b;
          const double *Up = U + e * UES + p * UWP;
          const double d11 = Up[0], d12 = Up[1], d44 = Up[2];
          double G[S][3];
          G[0][0] = Tp[0];
          G[0][1] = Tp[1];
          G[0][2] = Tp[2];
          if constexpr (NE > 0) {
            const double Nb = sh_N[p][b];
#pragma unroll
            for (int m = 0; m < NE; m++) {
              const bool on = m < actp;
              const double phi = Up[3 + 4 * m];
#pragma unroll
              for (int kk = 0; kk < 3; kk++) G[1 + m][kk] = on ? phi * G[0][kk] + Nb * Up[4 + 4 * m + kk] : 0.0;
            }
          }
#pragma unroll
          for (int sa = 0; sa < SB; sa++) {
            const double *R = Up + RO + 3 * (s0 + sa);
            if constexpr (ONEMAT) {
              // W^{st}[i][j] += (w detJ d_i Phi_{a,s}) d_j Phi_{b,t}: 9 FMAs per block and point
              const double l0 = R[0], l1 = R[1], l2 = R[2];
#pragma unroll
              for (int t = 0; t < S; t++) {
                double *wv = av[sa][t];
                wv[0] = fma(l0, G[t][0], wv[0]); wv[1] = fma(l0, G[t][1], wv[1]); wv[2] = fma(l0, G[t][2], wv[2]);
                wv[3] = fma(l1, G[t][0], wv[3]); wv[4] = fma(l1, G[t][1], wv[4]); wv[5] = fma(l1, G[t][2], wv[5]);
                wv[6] = fma(l2, G[t][0], wv[6]); wv[7] = fma(l2, G[t][1], wv[7]); wv[8] = fma(l2, G[t][2], wv[8]);
              }
            } else {
              double l[3], mm[3], qq[3];
#pragma unroll
              for (int kk = 0; kk < 3; kk++) {
                l[kk] = d12 * R[kk];
                mm[kk] = d44 * R[kk];
                qq[kk] = d11 * R[kk];
              }
#pragma unroll
              for (int t = 0; t < S; t++) add_block<KIND_ELASTIC, 3>(av[sa][t], l, mm, qq, G[t]);
            }
          }
        }
        if constexpr (ONEMAT) {
          // the material, once per block: K[i][j] = c12 W[i][j] + c44 W[j][i], K[i][i] = c11 W[i][i] + c44 (W[j][j] + W[k][k])
          const double c11 = U[e * UES], c12 = U[e * UES + 1], c44 = U[e * UES + 2];
#pragma unroll
          for (int sa = 0; sa < SB; sa++)
#pragma unroll
            for (int t = 0; t < S; t++) {
              double *wv = av[sa][t];
              const double w0 = wv[0], w1 = wv[4], w2 = wv[8];
              wv[0] = fma(c11, w0, c44 * (w1 + w2));
              wv[4] = fma(c11, w1, c44 * (w0 + w2));
              wv[8] = fma(c11, w2, c44 * (w0 + w1));
#pragma unroll
              for (int pr = 0; pr < 3; pr++) {
                const int ij = pr == 0 ? 1 : (pr == 1 ? 2 : 5), ji = pr == 0 ? 3 : (pr == 1 ? 6 : 7);
                const double wij = wv[ij], wji = wv[ji];
                wv[ij] = fma(c12, wij, c44 * wji);
                wv[ji] = fma(c12, wji, c44 * wij);
              }
            }
        }
        // ---- phase 3 (round 3): the lanes whose blocks belong to one neighbour meet in registers along x and y (the
        // partner's block through a quad permute / a row shift, added with a 0 / 1 factor so that every lane runs the same
        // code; a fixed tree: deterministic) and in LDS along z (the lanes below the node write, then the lanes above it add
        // in place: LDS executes a wave's accesses in order).  Every stored slot of the node's block has exactly one writing
        // lane: the block is not zeroed and nothing is staged.  (Rounds 1-2: stage + eight ordered rounds, 4 of 14.6 ms.)
        wave_sync();  // every lane is done with T and U (which the block may alias)
#if FEMASM_EXP == 12 || FEMASM_EXP == 14  // profiling build: no sums, no deposit
#pragma unroll
        for (int sa = 0; sa < SB; sa++)
#pragma unroll
          for (int t = 0; t < S; t++)
#pragma unroll
            for (int v = 0; v < 9; v++) asm volatile("" ::"v"(av[sa][t][v]));
        continue;
#endif
        const double mxs = (fx == 1) ? 1.0 : 0.0, mys = (fy == 1) ? 1.0 : 0.0;
#pragma unroll
        for (int sa = 0; sa < SB; sa++)
#pragma unroll
          for (int t = 0; t < S; t++)
#pragma unroll
            for (int v = 0; v < 9; v++) {
              double &wv = av[sa][t][v];
              wv = fma(mxs, quad_perm<0xD8>(wv), wv);   // fx = 1 takes fx = 2's
              wv = fma(mys, quad_perm<0x104>(wv), wv);  // row_shl:4 -- fy = 1 takes fy = 2's
            }
        const int nbx = ox + (fx & 1), nby = oy + (fy & 1), nbz = oz + (fz & 1);  // the neighbour, offsets -1 .. 1
        const bool exists = i + nbx >= 0 && i + nbx < M && j + nby >= 0 && j + nby < N && k + nbz >= 0 && k + nbz < P;
        const bool hold = exists && fx != 2 && fy != 2;
        const int nbpos = (((nbz + 1 - zi0) * cnty + (nby + 1 - yi0)) * cntx + (nbx + 1 - xi0)) * D;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
          if (hold && ((fz == 2) == (pass == 1))) {
#pragma unroll
            for (int sa = 0; sa < SB; sa++)
#pragma unroll
              for (int r = 0; r < 3; r++) {
                double *d = acc + (3 * (s0 + sa) + r) * rowlen + nbpos;
#pragma unroll
                for (int t = 0; t < S; t++)
#pragma unroll
                  for (int c = 0; c < 3; c++)
                    d[3 * t + c] = pass ? d[3 * t + c] + av[sa][t][3 * r + c] : av[sa][t][3 * r + c];
              }
          }
          wave_sync();
        }
      }
    } else if (Cf::ALIAS) {
      for (int t = lane; t < nvals; t += 64) acc[t] = 0.0;
    }
    wave_sync();
    // ---- row rules: Dirichlet faces (all dofs of the node), blocked redundant dofs ----
    {
      int nact = NE;
      if (!dirichlet && NE > 0 && !a.iface) {
        const int64_t bx = i > 0 ? i - 1 : 0, by = j > 0 ? j - 1 : 0, bz = k > 0 ? k - 1 : 0;  // the element behind the node
        const double *rb = a.gp + (bx + mx * (by + my * (bz - s.mat_e_ks))) * (int64_t)a.stride;
        nact = enr_active(rb, a.base, hex_idx(i > 0, j > 0, k > 0), NE);
      }
      const double diag = a.iface ? 0.0 : 1.0;
      const int qfirst = dirichlet ? 0 : 3 * (1 + nact);  // rows [qfirst, D) become unit rows
      if (dirichlet || !a.iface) {
        for (int qr = qfirst; qr < D; qr++)
          for (int t = lane; t < rowlen; t += 64) acc[qr * rowlen + t] = (t == self + qr) ? diag : 0.0;
      }
    }
    wave_sync();
    // ---- the block -> CSR array, 16 B per lane from the 128-byte boundary below it ----
    {
      double *o = a.out + (row_start(s, i, j, k, 0) - a.base_off);
      const int head = (int)((reinterpret_cast<uintptr_t>(o) >> 3) & 15);
      for (int t0 = -head; t0 < nvals; t0 += 128) {
        const int t = t0 + 2 * lane;
        if (t >= 0 && t + 1 < nvals) {
          const double v0 = acc[t], v1 = acc[t + 1];
#if FEMASM_EXP == 11 || FEMASM_EXP == 14  // profiling build: no global stores
          asm volatile("" ::"v"(v0), "v"(v1));
          continue;
#endif
          *reinterpret_cast<double2 *>(o + t) = make_double2(v0, v1);
        } else if (t + 1 >= 0 && t + 1 < nvals && t < 0) {
          o[t + 1] = acc[t + 1];
        } else if (t >= 0 && t < nvals) {
          o[t] = acc[t];
        }
      }
    }
    wave_sync();  // the next node's phase 1 overwrites the block
  }
}

// ---------------------------------------------------------------------------
// Level 2 of this kind (round 3): k_enr_box<S>.  Premises, both tested on the device (flag bad[1] -> the plan falls back
// to k_enr_node for good): the mesh is rectilinear (k_check_rectilinear's verdict before this kernel) and every element
// has ONE (E, nu) over its Gauss points (compared bitwise in phase 1).  Then
//   * J is diagonal and grad N_b(p) = G[p][.][b] / J_rr: no gradient table (12.8 KB of k_enr_node's 20 KB per wave), no
//     3x3 inverse, 6 coordinates per element instead of 24;
//   * the material leaves the Gauss-point loop as in k_assemble_s3_hex2: W^{st}[i][j] = sum_p (w detJ d_i Phi_{a,s})
//     d_j Phi_{b,t} -- 9 FMAs per block and point -- then K^{st}[i][j] = c12 W[i][j] + c44 W[j][i],
//     K^{st}[i][i] = c11 W[i][i] + c44 (W[j][j] + W[k][k]) once per block;
//   * lane = two bits per axis, (element side, node side): the two lanes that meet in the neighbour at offset 0 of an
//     axis are field values 1 and 2, so the blocks of a neighbour are added in REGISTERS along x (one quad permute) and
//     along y (one row shift), with a 0 / 1 factor so that every lane runs the same code, and in LDS along z (the lanes
//     below the node write, then the lanes above add in place: LDS executes a wave's accesses in order).  A fixed tree:
//     deterministic.  Every stored slot of the node's block has exactly one writing lane: the block is not zeroed, nothing
//     is staged, and the eight ordered rounds of k_enr_node (3.8 of its 11.7 ms at equal occupancy) are gone;
//   * a wave keeps phase 1 of its NEXT node in front of the stores of this one (the element table and the node's block do
//     not share LDS): gfx9 retires a wave's loads and stores through one in-order counter, so a load issued behind the
//     block's eight stores waits for their acknowledgement (1.2 of 9.1 ms before this);
//   * 13.4 KB of LDS per wave and <= 168 registers: 12 waves per CU (k_enr_node: 8).
// ---------------------------------------------------------------------------
constexpr int ENR_YS = 16;  // y-lines of a strip in k_enr_box's node order (4, 8, 16: 8.06-8.22, 8.19, 8.02-8.13 ms)
template <int S> struct EnrBoxCfg {
  static constexpr int NE = S - 1, D = 3 * S;
  static constexpr int SB = (S <= 2) ? S : 1;  // row families per contraction pass (9 S SB accumulators)
  // per element [1/J_00, 1/J_11, 1/J_22, c11, c12, c44, -, -], then per Gauss point (phi, grad phi) per family and the
  // w detJ scaled gradients of the row node's S functions
  static constexpr int BUP = (4 * NE + 3 * S + 1) & ~1, BUE = 8 + 8 * BUP;
  static constexpr int IN = 8 * BUE, ACC = 27 * D * D;
  static constexpr int PER_WAVE = (IN + ACC + 2 + 1) & ~1;
  static constexpr int NWMAX = (160 * 1024 - 2048 - 512) / (8 * PER_WAVE);  // (2 KB: the reference tables)
  static constexpr int NW = NWMAX > 12 ? 12 : NWMAX;  // waves per workgroup = per CU (12: three per SIMD at <= 168 registers)
  static constexpr int NPW_MAX = 32;                  // consecutive nodes per wave (EnrArgs::npw: set by the launch)
};

template <int S>
__global__ __launch_bounds__((64 * EnrBoxCfg<S>::NW)) void k_enr_box(const Slab s, const EnrArgs a) {
  using Cf = EnrBoxCfg<S>;
  constexpr int NE = Cf::NE, D = Cf::D, NW = Cf::NW, SB = Cf::SB, BUP = Cf::BUP, BUE = Cf::BUE;
  __shared__ __attribute__((aligned(16))) double sh_w[NW][Cf::PER_WAVE];
  __shared__ double sh_G[8][3][8], sh_N[8][8];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar node indices)
  for (int t = tid; t < 192; t += 64 * NW) (&sh_G[0][0][0])[t] = (&c_tab3.G[0][0][0])[t];
  for (int t = tid; t < 64; t += 64 * NW) (&sh_N[0][0])[t] = (&c_tab3.N[0][0])[t];
  __syncthreads();  // (the only workgroup barrier)
  double *U = sh_w[w], *acc0 = U + Cf::IN;
  auto wave_sync = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  const int64_t M = s.dim[0], N = s.dim[1], P = s.dim[2];
  const int64_t mx = M - 1, my = N - 1;
  const int fx = lane & 3, fy = (lane >> 2) & 3, fz = lane >> 4;
  const int e = (fx >> 1) + 2 * (fy >> 1) + 4 * (fz >> 1);  // element of the 8 around the node, k -> j -> i order
  const int b = hex_idx(fx & 1, fy & 1, fz & 1);            // column node of that element; phase 1: Gauss point p = b
  const int ox = (e & 1) - 1, oy = ((e >> 1) & 1) - 1, oz = (e >> 2) - 1;
  const int arow = hex_idx(-ox, -oy, -oz);  // the node itself in element e
  const int64_t nn = M * N * (a.s1 - a.s0);
  // Order of the nodes (round 3; each node's block is written by one wave whatever the order): an element's record is read
  // by the waves of its eight nodes, which lie on four x-lines (j, k), (j+1, k), (j, k+1), (j+1, k+1).  In the natural
  // order those lines are a plane apart (11 MB of records) and on different XCDs (workgroup b runs on XCD b % 8), so a
  // record came from HBM 3.3 times (4.6 GB read for 1.4 GB of records).  Now the lines go strip by strip -- ENR_YS
  // y-lines, all node layers of the pass, then the next strip -- so the four lines are 2 * ENR_YS lines apart, and the
  // workgroups of an XCD take consecutive stretches of that order (lb below): 3.4 GB read, the kernel's traffic 21.1 ->
  // 19.9 GB = 1.14 x its algorithmic bytes.  Its time does not change (8.09 -> 8.13 ms on one box): it is not bound by
  // its reads.
  const unsigned nb = gridDim.x, bq = nb >> 3, br = nb & 7u, xcd = blockIdx.x & 7u;
  const int64_t lb = (int64_t)xcd * bq + (xcd < br ? xcd : br) + (blockIdx.x >> 3);
  const int64_t q0 = (lb * NW + w) * a.npw;
  const int nq = (int)((nn - q0 < a.npw) ? (nn - q0 > 0 ? nn - q0 : 0) : a.npw);
  // (the wave's nodes are consecutive in that order: decoded once, then advanced)
  const int nk = (int)(a.s1 - a.s0);
  int ii, jj, kk_, j_lo, j_hi;  // j_lo, j_hi: the strip
  {
    const int64_t line = q0 / M;
    ii = (int)(q0 - line * M);
    const int64_t per_strip = (int64_t)ENR_YS * nk;
    int64_t st = line / per_strip;
    const int64_t nstrips = (N + ENR_YS - 1) / ENR_YS;
    if (st >= nstrips) st = nstrips - 1;  // (q0 beyond the last node: nq = 0)
    j_lo = (int)(st * ENR_YS);
    j_hi = (j_lo + ENR_YS < (int)N) ? j_lo + ENR_YS : (int)N;
    const int64_t r = line - st * per_strip;  // line within the strip: layer-major, then y (a short last strip: h < ENR_YS)
    const int h = j_hi - j_lo;
    kk_ = (int)(a.s0 + r / h);
    jj = j_lo + (int)(r % h);
  }
  int actp = 0;

  // ---- phase 1, lane (e, p): the element's box, premise, coefficients, (phi, grad phi), the row node's gradients ----
  auto phase1 = [&](int64_t i, int64_t j, int64_t k) {
    const int p = b;
    const int64_t ex = i + ox, ey = j + oy, ez = k + oz;
    const bool ok = !(ex < 0 || ex >= mx || ey < 0 || ey >= my || ez < s.els || ez >= s.ele);
    const int64_t exc = ex < 0 ? 0 : (ex >= mx ? mx - 1 : ex), eyc = ey < 0 ? 0 : (ey >= my ? my - 1 : ey);
    const int64_t ezc = ez < s.els ? s.els : (ez >= s.ele ? s.ele - 1 : ez);
    const double *rec = a.gp + (exc + mx * (eyc + my * (ezc - s.mat_e_ks))) * (int64_t)a.stride;
    double *Ue = U + e * BUE, *Up = Ue + 8 + p * BUP;
    // the element's box from its (-,-,-) and (+,+,+) nodes (the mesh is rectilinear: k_check_rectilinear)
    const double *xa = a.coords + 3 * (exc + M * (eyc + N * (ezc - s.cks)));
    const double *xb = a.coords + 3 * ((exc + 1) + M * ((eyc + 1) + N * (ezc + 1 - s.cks)));
    const double J0 = 0.5 * (xb[0] - xa[0]), J1 = 0.5 * (xb[1] - xa[1]), J2 = 0.5 * (xb[2] - xa[2]);
    const double Ep = rec[96 + p], nup = rec[104 + p];
    // premise: one (E, nu) per element -- every lane compares its Gauss point's pair with that of point 0
    if (ok && !(Ep == rec[96] && nup == rec[104]) && !a.iface) atomicAdd(a.bad + 1, 1ULL);
    const double wJ = 1.0 * (J0 * J1 * J2);
    // the element is counted once, by the wave of its local node 6 (always an owned node of this rank)
    if (ok && e == 0 && !a.iface && !(wJ > 0.0)) atomicAdd(a.bad, 1ULL);
    // 1 / J_rr: hardware reciprocal + two Newton steps (relative error ~1e-16; the quotient sequence costs three times
    // the instructions, and the results of this level are compared with the reference's at 1e-13)
    auto recip = [](double x) {
      double r = __builtin_amdgcn_rcp(x);
      r = fma(fma(-x, r, 1.0), r, r);
      return fma(fma(-x, r, 1.0), r, r);
    };
    const double ai[3] = {recip(J0), recip(J1), recip(J2)};
    if (p == 0) {
      double c11 = 0.0, c12 = 0.0, c44 = 0.0;
      scaled_coefs<KIND_ELASTIC>(Ep, nup, 1.0, c11, c12, c44);
      Ue[0] = ai[0], Ue[1] = ai[1], Ue[2] = ai[2];
      Ue[3] = c11, Ue[4] = c12, Ue[5] = c44;
    }
    // gradient of the row node's N at this point, scaled by w detJ (zero for an element that is not assembled here)
    const double wz = ok ? wJ : 0.0;
    const double r0[3] = {sh_G[p][0][arow] * ai[0], sh_G[p][1][arow] * ai[1], sh_G[p][2][arow] * ai[2]};
    Up[4 * NE] = wz * r0[0], Up[4 * NE + 1] = wz * r0[1], Up[4 * NE + 2] = wz * r0[2];
    if constexpr (NE > 0) {
      actp = enr_active(rec, a.base, p, NE);
      const int actr = enr_active(rec, a.base, arow, NE);
      const double Nr = sh_N[p][arow];
#pragma unroll
      for (int m = 0; m < NE; m++) {
        const double *en = rec + a.base + 64 * m;
        double phi = en[8 + p], dphi[3] = {en[16 + 3 * p], en[16 + 3 * p + 1], en[16 + 3 * p + 2]};
        if (a.sgfem) {
          double sN = 0.0, sG[3] = {0.0, 0.0, 0.0};
#pragma unroll
          for (int n = 0; n < 8; n++) {
            const double pn = en[n];
            sN = fma(sh_N[p][n], pn, sN);
#pragma unroll
            for (int kk = 0; kk < 3; kk++) sG[kk] = fma(sh_G[p][kk][n], pn, sG[kk]);
          }
          phi = phi - sN;
#pragma unroll
          for (int kk = 0; kk < 3; kk++) dphi[kk] = dphi[kk] - sG[kk] * ai[kk];
        }
        Up[4 * m] = phi;
#pragma unroll
        for (int kk = 0; kk < 3; kk++) Up[4 * m + 1 + kk] = dphi[kk];
        const bool on = m < actr;
#pragma unroll
        for (int kk = 0; kk < 3; kk++) Up[4 * NE + 3 * (1 + m) + kk] = on ? wz * (phi * r0[kk] + Nr * dphi[kk]) : 0.0;
      }
    }
  };

  if (nq > 0) phase1(ii, jj, kk_);
#pragma unroll 1
  for (int qi = 0; qi < nq; qi++) {
    const int64_t i = ii, j = jj, k = kk_;
    const int cntx = span1(i, M), cnty = span1(j, N), cntz = span1(k, P);
    const int rowlen = cntx * cnty * cntz * D, nvals = D * rowlen;
    const int xi0 = (i == 0), yi0 = (j == 0), zi0 = (k == 0);
    const bool dirichlet = (a.bc == BC_S3) && on_boundary(s, i, j, k);
    const int self = (((k > 0) * cnty + (j > 0)) * cntx + (i > 0)) * D;
    // the node's block in the CSR array; its LDS copy starts at the parity of its HBM address, so that a 16-byte chunk of
    // HBM is a 16-byte chunk of LDS (ds_read_b128 in the copy-out)
    double *o = a.out + (row_start(s, i, j, k, 0) - a.base_off);
    const int head = (int)((reinterpret_cast<uintptr_t>(o) >> 3) & 15);
    double *acc = acc0 + (head & 1);
    wave_sync();  // the element table of this node is complete

    if (!dirichlet) {
#pragma unroll 1
      for (int s0 = 0; s0 < S; s0 += SB) {
        // ---- phase 2, lane (e, b): W^{st}[i][j] over the eight Gauss points, then the material once ----
        double av[SB][S][9];
#pragma unroll
        for (int sa = 0; sa < SB; sa++)
#pragma unroll
          for (int t = 0; t < S; t++)
#pragma unroll
            for (int v = 0; v < 9; v++) av[sa][t][v] = 0.0;
        const double *Ue = U + e * BUE;
        const double ai0 = Ue[0], ai1 = Ue[1], ai2 = Ue[2], c11 = Ue[3], c12 = Ue[4], c44 = Ue[5];
#pragma unroll 2
        for (int p = 0; p < 8; p++) {
          const double *Up = Ue + 8 + p * BUP;
          double G[S][3];
          G[0][0] = sh_G[p][0][b] * ai0, G[0][1] = sh_G[p][1][b] * ai1, G[0][2] = sh_G[p][2][b] * ai2;
          if constexpr (NE > 0) {
            const double Nb = sh_N[p][b];
#pragma unroll
            for (int m = 0; m < NE; m++) {
              const double phi = Up[4 * m];
#pragma unroll
              for (int kk = 0; kk < 3; kk++) G[1 + m][kk] = phi * G[0][kk] + Nb * Up[4 * m + 1 + kk];
            }
          }
#pragma unroll
          for (int sa = 0; sa < SB; sa++) {
            const double *R = Up + 4 * NE + 3 * (s0 + sa);
            const double l0 = R[0], l1 = R[1], l2 = R[2];
#pragma unroll
            for (int t = 0; t < S; t++) {
              double *wv = av[sa][t];
              wv[0] = fma(l0, G[t][0], wv[0]); wv[1] = fma(l0, G[t][1], wv[1]); wv[2] = fma(l0, G[t][2], wv[2]);
              wv[3] = fma(l1, G[t][0], wv[3]); wv[4] = fma(l1, G[t][1], wv[4]); wv[5] = fma(l1, G[t][2], wv[5]);
              wv[6] = fma(l2, G[t][0], wv[6]); wv[7] = fma(l2, G[t][1], wv[7]); wv[8] = fma(l2, G[t][2], wv[8]);
            }
          }
        }
#pragma unroll
        for (int sa = 0; sa < SB; sa++)
#pragma unroll
          for (int t = 0; t < S; t++) {
            double *wv = av[sa][t];
            // an inactive family of column node b is zero on this element: zero coefficients (the sums are finite)
            const bool on = t == 0 || (t - 1) < actp;
            const double k11 = on ? c11 : 0.0, k12 = on ? c12 : 0.0, k44 = on ? c44 : 0.0;
            const double d0 = wv[0], d1 = wv[4], d2 = wv[8];
            wv[0] = fma(k11, d0, k44 * (d1 + d2));
            wv[4] = fma(k11, d1, k44 * (d0 + d2));
            wv[8] = fma(k11, d2, k44 * (d0 + d1));
#pragma unroll
            for (int pr = 0; pr < 3; pr++) {
              const int ij = pr == 0 ? 1 : (pr == 1 ? 2 : 5), ji = pr == 0 ? 3 : (pr == 1 ? 6 : 7);
              const double wij = wv[ij], wji = wv[ji];
              wv[ij] = fma(k12, wij, k44 * wji);
              wv[ji] = fma(k12, wji, k44 * wij);
            }
          }
        // ---- phase 3: the lanes of a neighbour meet in registers along x and y, in LDS along z ----
        const double mxs = (fx == 1) ? 1.0 : 0.0, mys = (fy == 1) ? 1.0 : 0.0;
#pragma unroll
        for (int sa = 0; sa < SB; sa++)
#pragma unroll
          for (int t = 0; t < S; t++)
#pragma unroll
            for (int v = 0; v < 9; v++) {
              double &wv = av[sa][t][v];
              wv = fma(mxs, quad_perm<0xD8>(wv), wv);   // fx = 1 takes fx = 2's
              wv = fma(mys, quad_perm<0x104>(wv), wv);  // row_shl:4 -- fy = 1 takes fy = 2's
            }
        const int nbx = ox + (fx & 1), nby = oy + (fy & 1), nbz = oz + (fz & 1);  // the neighbour, offsets -1 .. 1
        const bool exists = i + nbx >= 0 && i + nbx < M && j + nby >= 0 && j + nby < N && k + nbz >= 0 && k + nbz < P;
        const bool hold = exists && fx != 2 && fy != 2;
        const int nbpos = (((nbz + 1 - zi0) * cnty + (nby + 1 - yi0)) * cntx + (nbx + 1 - xi0)) * D;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
          if (hold && ((fz == 2) == (pass == 1))) {
#pragma unroll
            for (int sa = 0; sa < SB; sa++)
#pragma unroll
              for (int r = 0; r < 3; r++) {
                double *d = acc + (3 * (s0 + sa) + r) * rowlen + nbpos;
#pragma unroll
                for (int t = 0; t < S; t++)
#pragma unroll
                  for (int c = 0; c < 3; c++)
                    d[3 * t + c] = pass ? d[3 * t + c] + av[sa][t][3 * r + c] : av[sa][t][3 * r + c];
              }
          }
          wave_sync();
        }
      }
    } else {
      for (int t = lane; t < nvals; t += 64) acc[t] = 0.0;
      wave_sync();
    }
    // ---- row rules: Dirichlet faces (all dofs of the node), blocked redundant dofs ----
    {
      int nact = NE;
      if (!dirichlet && NE > 0 && !a.iface) {
        const int64_t bx = i > 0 ? i - 1 : 0, by = j > 0 ? j - 1 : 0, bz = k > 0 ? k - 1 : 0;  // the element behind the node
        const double *rb = a.gp + (bx + mx * (by + my * (bz - s.mat_e_ks))) * (int64_t)a.stride;
        nact = enr_active(rb, a.base, hex_idx(i > 0, j > 0, k > 0), NE);
      }
      const double diag = a.iface ? 0.0 : 1.0;
      const int qfirst = dirichlet ? 0 : 3 * (1 + nact);  // rows [qfirst, D) become unit rows
      if (dirichlet || !a.iface) {
        for (int qr = qfirst; qr < D; qr++)
          for (int t = lane; t < rowlen; t += 64) acc[qr * rowlen + t] = (t == self + qr) ? diag : 0.0;
      }
    }
    wave_sync();
    // ---- the next node's phase 1 BEFORE this node's stores (see the header) ----
    if (++ii == (int)M) {
      ii = 0;
      if (++jj == j_hi) {
        jj = j_lo;
        if (++kk_ == (int)a.s1) {  // the next strip
          kk_ = (int)a.s0;
          j_lo = j_hi;
          j_hi = (j_lo + ENR_YS < (int)N) ? j_lo + ENR_YS : (int)N;
          jj = j_lo;
        }
      }
    }
    if (qi + 1 < nq) phase1(ii, jj, kk_);
    // ---- the block -> CSR array, 16 B per lane from the 128-byte boundary below it ----
    for (int t0 = -head; t0 < nvals; t0 += 128) {
      const int t = t0 + 2 * lane;
      if (t >= 0 && t + 1 < nvals) {
        const double2 v = *reinterpret_cast<const double2 *>(acc + t);  // 16-byte aligned: t + (head & 1) is even
#if FEMASM_EXP == 11  // profiling build: no global stores
        asm volatile("" ::"v"(v.x), "v"(v.y));
        continue;
#endif
        store_pair(o + t, v);
      } else if (t + 1 >= 0 && t + 1 < nvals && t < 0) {
        o[t + 1] = acc[t + 1];
      } else if (t >= 0 && t < nvals) {
        o[t] = acc[t];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// RHS of the enriched element (S3:520-656 with Phi_{a,s} in place of N_a): one thread per element -> element
// vectors (blocks of 16 elements x 8 D lines, as ef_idx), then one thread per node.
// ---------------------------------------------------------------------------
__device__ inline int64_t ef_idx_rt(int64_t e, int t, int nout) { return (e >> 4) * ((int64_t)nout * 16) + t * 16 + (e & 15); }

template <int S>
__global__ __launch_bounds__(128) void k_enr_rhs_elem(const Slab s, const EnrArgs a, int64_t nel) {
  constexpr int D = 3 * S, NE = S - 1;
  const int64_t e = (int64_t)blockIdx.x * 128 + threadIdx.x;
  if (e >= nel) return;
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t mx = M - 1, my = N - 1;
  const int64_t ex = e % mx, ey = (e / mx) % my, ez = s.els + e / (mx * my);
  const double *rec = a.gp + (ex + mx * (ey + my * (ez - s.mat_e_ks))) * (int64_t)a.stride;
  double xc[24];
#pragma unroll
  for (int n = 0; n < 8; n++) {
    const int64_t node = (ex + HDI[n]) + M * ((ey + HDJ[n]) + N * (ez + HDK[n] - s.cks));
#pragma unroll
    for (int d = 0; d < 3; d++) xc[3 * n + d] = a.coords[3 * node + d];
  }
  int nact[8];
#pragma unroll
  for (int n = 0; n < 8; n++) nact[n] = enr_active(rec, a.base, n, NE);
  double F[8 * D];
#pragma unroll
  for (int t = 0; t < 8 * D; t++) F[t] = 0.0;
  // body force at the volume Gauss points
  for (int p = 0; p < 8; p++) {
    const double fac = 1.0 * det3_only(c_tab3.G[p], xc);
    const double f[3] = {rec[160 + p], rec[168 + p], rec[176 + p]};
    double phi[S];
    phi[0] = 1.0;
#pragma unroll
    for (int m = 0; m < NE; m++) {
      const double *en = rec + a.base + 64 * m;
      double v = en[8 + p];
      if (a.sgfem)
#pragma unroll
        for (int b = 0; b < 8; b++) v = v - c_tab3.N[p][b] * en[b];
      phi[1 + m] = v;
    }
#pragma unroll
    for (int n = 0; n < 8; n++)
#pragma unroll
      for (int sa = 0; sa < S; sa++) {
        const double val = (sa == 0 || sa - 1 < nact[n]) ? c_tab3.N[p][n] * phi[sa] : 0.0;
#pragma unroll
        for (int c = 0; c < 3; c++) F[n * D + 3 * sa + c] += fac * val * f[c];
      }
  }
  // natural-BC term over the 24 face points of the element (S3:548-656)
  const double dx = fabs(xc[0] - xc[9]), dy = fabs(xc[1] - xc[4]), dz = fabs(xc[2] - xc[14]);
  const double *gx = rec + 184, *gy = rec + 256, *gz = rec + 328, *EB = rec + 112, *nuB = rec + 136;
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
    double phi[S];
    phi[0] = 1.0;
#pragma unroll
    for (int m = 0; m < NE; m++) {
      const double *en = rec + a.base + 64 * m;
      double v = en[40 + p];
      if (a.sgfem)
#pragma unroll
        for (int b = 0; b < 8; b++) v = v - c_tab3.NF[p][b] * en[b];
      phi[1 + m] = v;
    }
#pragma unroll
    for (int n = 0; n < 8; n++)
#pragma unroll
      for (int sa = 0; sa < S; sa++) {
        const double val = (sa == 0 || sa - 1 < nact[n]) ? c_tab3.NF[p][n] * phi[sa] : 0.0;
#pragma unroll
        for (int c = 0; c < 3; c++) F[n * D + 3 * sa + c] += fac * val * sn[c];
      }
  }
#pragma unroll
  for (int t = 0; t < 8 * D; t++) a.elemF[ef_idx_rt(e, t, 8 * D)] = F[t];
}

// The same element vectors, 8 lanes per element and 8 elements per wave (the form the step uses; the thread-per-element
// kernel above stays as the plain statement of the formulas and is what FEMASM_ENR_RHS_SIMPLE=1 selects).
// Lane (x, j) of element x: face Gauss points j, j + 8, j + 16 (one per face pair) and volume Gauss point j -> traction
// sigma.n x face Jacobian, w detJ f and the enrichment factors phi_s of those points into LDS; then lane j = node j
// sums its D entries over the 24 + 8 points.  Same terms as the kernel above in another order (rounding only).
template <int S>
__global__ __launch_bounds__(256) void k_enr_rhs_elem8(const Slab s, const EnrArgs a, int64_t nel) {
  constexpr int D = 3 * S, NE = S - 1, NW = 4;
  __shared__ double sh_xc[NW][8][24];
  __shared__ double sh_tr[NW][8][32][3];       // [point: 24 face, then 8 volume][component]
  __shared__ double sh_phi[NW][8][NE > 0 ? NE : 1][32];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, x = lane >> 3, j = lane & 7;
  const int64_t e0 = ((int64_t)blockIdx.x * NW + w) * 8;
  if (e0 >= nel) return;
  const bool live = e0 + x < nel;
  const int64_t e = live ? e0 + x : nel - 1;  // idle lanes repeat the last element (clamped, never masked loads)
  const int64_t M = s.dim[0], N = s.dim[1];
  const int64_t mx = M - 1, my = N - 1;
  const int64_t ex = e % mx, ey = (e / mx) % my, ez = s.els + e / (mx * my);
  const double *rec = a.gp + (ex + mx * (ey + my * (ez - s.mat_e_ks))) * (int64_t)a.stride;
  auto wave_sync = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  {
    const int64_t node = (ex + HDI[j]) + M * ((ey + HDJ[j]) + N * (ez + HDK[j] - s.cks));
#pragma unroll
    for (int d = 0; d < 3; d++) sh_xc[w][x][3 * j + d] = a.coords[3 * node + d];
  }
  wave_sync();
  const double *xc = sh_xc[w][x];
  const double dx = fabs(xc[0] - xc[9]), dy = fabs(xc[1] - xc[4]), dz = fabs(xc[2] - xc[14]);  // S3:560-562
  const double *gx = rec + 184, *gy = rec + 256, *gz = rec + 328, *EB = rec + 112, *nuB = rec + 136;
#pragma unroll
  for (int ax = 0; ax < 3; ax++) {
    const int p = j + 8 * ax;
    const double pe = EB[p], pn = nuB[p];
    const double factor = pe * (1 - pn) / ((1.0 + pn) * (1.0 - 2.0 * pn));
    const double d11 = factor * 1.0, d12 = factor * (pn / (1 - pn)), d44 = factor * (0.5 * (1 - 2 * pn) / (1 - pn));
    const double e0_ = gx[3 * p], e1 = gy[3 * p + 1], e2 = gz[3 * p + 2];
    const double e3 = gx[3 * p + 1] + gy[3 * p], e4 = gy[3 * p + 2] + gz[3 * p + 1], e5 = gx[3 * p + 2] + gz[3 * p];
    const double z0 = d11 * e0_ + d12 * e1 + d12 * e2;
    const double z1 = d12 * e0_ + d11 * e1 + d12 * e2;
    const double z2 = d12 * e0_ + d12 * e1 + d11 * e2;
    const double z3 = d44 * e3, z4 = d44 * e4, z5 = d44 * e5;
    const double side = (j & 4) ? -1.0 : 1.0;  // points 0-3 of a face pair: +1 face, 4-7: -1 face
    double sn[3], Jp;
    if (ax == 0)      { sn[0] = z0 * side; sn[1] = z3 * side; sn[2] = z5 * side; Jp = (0.5 * dy) * (0.5 * dz); }
    else if (ax == 1) { sn[0] = z3 * side; sn[1] = z1 * side; sn[2] = z4 * side; Jp = (0.5 * dx) * (0.5 * dz); }
    else              { sn[0] = z5 * side; sn[1] = z4 * side; sn[2] = z2 * side; Jp = (0.5 * dx) * (0.5 * dy); }
    const double fac = 1.0 * Jp;
#pragma unroll
    for (int c = 0; c < 3; c++) sh_tr[w][x][p][c] = fac * sn[c];
#pragma unroll
    for (int m = 0; m < NE; m++) {
      const double *en = rec + a.base + 64 * m;
      double v = en[40 + p];
      if (a.sgfem)
#pragma unroll
        for (int b = 0; b < 8; b++) v = v - c_tab3.NF[p][b] * en[b];
      sh_phi[w][x][m][p] = v;
    }
  }
  {
    const double fac = 1.0 * det3_only(c_tab3.G[j], xc);
    sh_tr[w][x][24 + j][0] = fac * rec[160 + j];
    sh_tr[w][x][24 + j][1] = fac * rec[168 + j];
    sh_tr[w][x][24 + j][2] = fac * rec[176 + j];
#pragma unroll
    for (int m = 0; m < NE; m++) {
      const double *en = rec + a.base + 64 * m;
      double v = en[8 + j];
      if (a.sgfem)
#pragma unroll
        for (int b = 0; b < 8; b++) v = v - c_tab3.N[j][b] * en[b];
      sh_phi[w][x][m][24 + j] = v;
    }
  }
  wave_sync();
  const int nact = enr_active(rec, a.base, j, NE);
  double F[D];
#pragma unroll
  for (int t = 0; t < D; t++) F[t] = 0.0;
#pragma unroll 4
  for (int p = 0; p < 32; p++) {
    const double Nn = (p < 24) ? c_tab3.NF[p][j] : c_tab3.N[p - 24][j];
    const double t0 = sh_tr[w][x][p][0], t1 = sh_tr[w][x][p][1], t2 = sh_tr[w][x][p][2];
    F[0] += Nn * t0; F[1] += Nn * t1; F[2] += Nn * t2;
#pragma unroll
    for (int m = 0; m < NE; m++) {
      const double val = (m < nact) ? Nn * sh_phi[w][x][m][p] : 0.0;
      F[3 * (1 + m)] += val * t0; F[3 * (1 + m) + 1] += val * t1; F[3 * (1 + m) + 2] += val * t2;
    }
  }
  if (live)
#pragma unroll
    for (int t = 0; t < D; t++) a.elemF[ef_idx_rt(e, j * D + t, 8 * D)] = F[t];
}

// one thread per (node, dof) of node layers [s0, s1): element contributions in k -> j -> i order, then the row rules
__global__ __launch_bounds__(256) void k_enr_rhs_gather(const Slab s, const EnrArgs a) {
  const int64_t M = s.dim[0], N = s.dim[1];
  const int D = s.dof;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= M * N * (a.s1 - a.s0) * D) return;
  const int qd = (int)(t % D);
  const int64_t nq = t / D;
  const int64_t i = nq % M, j = (nq / M) % N, k = a.s0 + nq / (M * N);
  const int64_t mx = M - 1, my = N - 1;
  double b = 0.0;
  for (int oz = -1; oz <= 0; oz++)
    for (int oy = -1; oy <= 0; oy++)
      for (int ox = -1; ox <= 0; ox++) {
        const int64_t ex = i + ox, ey = j + oy, ez = k + oz;
        if (ex < 0 || ex >= mx || ey < 0 || ey >= my || ez < s.els || ez >= s.ele) continue;
        const int64_t e = ex + mx * (ey + my * (ez - s.els));
        b = b + a.elemF[ef_idx_rt(e, hex_idx(-ox, -oy, -oz) * D + qd, 8 * D)];
      }
  const bool dirichlet = (a.bc == BC_S3) && on_boundary(s, i, j, k);
  if (dirichlet) {
    if (a.iface) b = 0.0;
    else if (qd == 2) {
      constexpr double PI = 3.14159265358979323846264338327950288;
      const int64_t node = i + M * (j + N * (k - s.cks));
      b = sin(PI * a.coords[3 * node]) * sin(PI * a.coords[3 * node + 1]) * sin(PI * a.coords[3 * node + 2]);
    } else b = 0.0;  // S3:964-968 for the displacement dofs; enriched dofs of a Dirichlet node: 0
  } else if (qd >= 3 && !a.iface) {
    const int64_t bx = i > 0 ? i - 1 : 0, by = j > 0 ? j - 1 : 0, bz = k > 0 ? k - 1 : 0;
    const double *rec = a.gp + (bx + mx * (by + my * (bz - s.mat_e_ks))) * (int64_t)a.stride;
    if (qd / 3 - 1 >= enr_active(rec, a.base, hex_idx(i > 0, j > 0, k > 0), D / 3 - 1)) b = 0.0;  // blocked dof
  }
  a.out[t] = b;
}

}  // namespace femasm
