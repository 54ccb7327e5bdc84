// femasm_device.h -- shared host/device definitions of the gfx950 assembly engine.
//
// Index semantics restated from PETSc's DMDA: natural node order
// n = i + M*(j + N*k), row = dof*n + c, BOX stencil width 1, non-periodic, sorted columns.
// Everything here is closed form: the element->nonzero scatter map never touches memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace femasm {

// Slab description handed to every kernel by value.
struct Slab {
  int64_t dim[3];      // M, N, P (P = 1 in 2-D)
  int64_t ks, ke;      // owned node layers of the slowest direction
  int64_t cks;         // first node layer held in coords[]
  int64_t els, ele;    // element layers assembled by this rank: [els, ele)
  int64_t mat_e_ks;    // first element layer held in the material array
  int64_t before0;     // stencil pairs in front of the first owned node (for local rowptr)
  int32_t nsd, dof;
};

// number of stencil neighbours of index i along a direction of M nodes
__host__ __device__ inline int span1(int64_t i, int64_t M) {
  return (M == 1) ? 1 : 3 - (i == 0) - (i == M - 1);
}
// sum of span1 over indices < i
__host__ __device__ inline int64_t pre1(int64_t i, int64_t M) {
  if (M == 1) return i;
  if (i <= 0) return 0;
  if (i >= M) return 3 * M - 2;
  return 3 * i - 1;
}
__host__ __device__ inline int64_t tot1(int64_t M) { return (M == 1) ? 1 : 3 * M - 2; }

// stencil pairs (node, neighbour) of all nodes in front of node (i,j,k), natural order
__host__ __device__ inline int64_t pairs_before(const Slab &s, int64_t i, int64_t j, int64_t k) {
  const int64_t SX = tot1(s.dim[0]), SY = tot1(s.dim[1]);
  return pre1(k, s.dim[2]) * SY * SX +
         (int64_t)span1(k, s.dim[2]) * (pre1(j, s.dim[1]) * SX + (int64_t)span1(j, s.dim[1]) * pre1(i, s.dim[0]));
}

// local (slab) position of the first stored entry of row dof*n + c
__host__ __device__ inline int64_t row_start(const Slab &s, int64_t i, int64_t j, int64_t k, int c) {
  const int64_t cnt = (int64_t)span1(i, s.dim[0]) * span1(j, s.dim[1]) * span1(k, s.dim[2]);
  return (int64_t)s.dof * s.dof * (pairs_before(s, i, j, k) - s.before0) + (int64_t)c * s.dof * cnt;
}

// local node numbering of the reference elements (A13): (bx,by[,bz]) -> index
__host__ __device__ constexpr int quad_idx(int bx, int by) { return bx == 0 ? (by == 0 ? 0 : 1) : (by == 0 ? 3 : 2); }
__host__ __device__ constexpr int hex_idx(int bx, int by, int bz) { return 4 * bz + quad_idx(bx, by); }

}  // namespace femasm
