/*
 * femasm.h -- C ABI of the MI355X (gfx950) finite-element global-assembly engine.
 *
 * Drop-in boundary for the assembly hot path of the CuiCu-618/petsc drivers.  The
 * reference has no FFI/plugin interface of its own: the hot path is a set of static
 * functions called from main()/solve_elasticity_*().  Each entry point below names
 * the reference call site(s) it replaces (P3 = FEM_poisson3D/FEM.cpp,
 * E2 = FEM_elsatic2D/FEM.cpp, S3 = SGFEM/SGFEM_elastic3D.cpp,
 * R3 = SGFEM_real/SGFEM_elastic3D.cpp).  INTEGRATION.md shows the driver patch.
 *
 * Plain C: no C++ types, no exceptions, no PETSc or torch headers.  All buffers are
 * caller-owned; every data pointer may be a HOST pointer or a DEVICE (HIP) pointer,
 * detected with hipPointerGetAttributes.  Host buffers are staged over PCIe.
 * Return value: 0 = success, otherwise a FEMASM_ERR_* class; text from
 * femasm_last_error() (thread-local), PETSc style (wrap with CHKERRQ).
 */
#ifndef FEMASM_H
#define FEMASM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  FEMASM_POISSON3D_Q1      = 0, /* P3: 8-node hex, dof 1, 8 Gauss points                */
  FEMASM_PLANE_STRAIN2D_Q1 = 1, /* E2: 4-node quad, dof 2, 4 Gauss points               */
  FEMASM_ELASTIC3D_Q1      = 2, /* S3/R3 (all enrich_type == 0): 8-node hex, dof 3      */
  FEMASM_ELASTIC3D_ENRICHED = 3 /* BASELINE config 5: SGFEM-enriched hex, dof 3(1+n_enr): no reference arithmetic
                                   exists (R3 holds the scaffolding only, R3:39-68, 1265-1269, 1467-1595); the
                                   specification is this engine's, DESIGN.md section 9                  */
};

enum {
  FEMASM_OK = 0,
  FEMASM_ERR_ARG = 1,      /* bad argument / unsupported combination            */
  FEMASM_ERR_HIP = 2,      /* a HIP runtime call or kernel launch failed        */
  FEMASM_ERR_GEOMETRY = 3, /* non-positive or non-finite det(J) at a Gauss point */
  FEMASM_ERR_NOMEM = 4
};

/* "mesh": the DMDA the drivers create (P3:658-663, E2:567-578, S3:1185-1198):
 * BOX stencil, width 1, non-periodic, natural node order n = i + M*(j + N*k). */
typedef struct {
  int32_t problem;      /* FEMASM_* problem kind                                              */
  int32_t n_enr;        /* FEMASM_ELASTIC3D_ENRICHED: enrichment families per node, 0..3 (the DMDA's dof
                           is 3(1+n_enr); R3:43 test_dof); 0 for the other kinds                      */
  int64_t M, N, P;      /* global node counts per direction (P = 1 in 2-D)                    */
  int64_t ks, ke;       /* owned node layers [ks,ke) of the slowest direction (z in 3-D, y in
                           2-D): DMDA slab split.  0,P (0,N in 2-D) on one GPU.               */
  int64_t cks, cke;     /* node layers [cks,cke) held in coords[] (the DMDA ghosted local
                           coordinate vector: ks-1 .. ke clipped to the mesh); 0,P on one GPU.
                           Must cover [max(ks-1,0), ke).                                      */
  const double *coords; /* node coordinates, interleaved (x,y[,z]), layers [cks,cke)          */
} femasm_mesh;

/* "material": per-element GaussPointCoefficients array (A14), zero-copy from
 * VecGetArray(properties) (E2:29-35, S3:51-64, R3:52-68), or uniform constants. */
enum {
  FEMASM_MAT_AOS = 0,
  FEMASM_MAT_UNIFORM = 1,
  /* S3/R3 only, no counterpart in the reference: E, nu constant (the drivers' -iso_E / -iso_nu, S3:1311-1314) and the
   * loads of the drivers' manufactured solution (evaluate_Elastic, S3:958-985) evaluated inside the right-hand-side
   * kernel at the Gauss-point positions it derives from the coordinates -- the fill loop S3:1243-1363 and the
   * 2 304 bytes per element that AssembleF_Elasticity (S3:720-803) reads back from it are not needed.  Same b as from
   * the records the drivers fill, to rounding.  FEMASM_MAT_AOS stays the drop-in default. */
  FEMASM_MAT_MANUFACTURED = 2,
  /* Compact per-element form (A14: "should also accept a compact SoA/uniform form"; no counterpart in the reference,
   * whose records repeat one (E, nu) at every Gauss point wherever the material is element-wise constant): two planes of
   * doubles in device memory, E of element e at gp[e], nu at gp[stride + e], e = ei + mx*(ej + my*(ek - e_ks));
   * stride = distance between the planes in doubles (>= the elements held).  16 B per element instead of the 128-B
   * line of a 3 200-B record.  It carries no loads: K (and femasm_apply_operator) use it; a right-hand side asked for
   * in the same call holds the Dirichlet data only, as with FEMASM_MAT_UNIFORM.  Not for the enriched kind.
   * The same (E, nu) through FEMASM_MAT_AOS records gives the same K bit for bit (tested). */
  FEMASM_MAT_ELEMENT_SOA = 3
};
typedef struct {
  int32_t layout;       /* FEMASM_MAT_AOS | _UNIFORM | _MANUFACTURED | _ELEMENT_SOA (K only needs E,nu) */
  int32_t stride;       /* doubles per element: 24 (E2) | 400 (S3) | 408/416 (R3); _ELEMENT_SOA: plane distance */
  int64_t e_ks;         /* first element layer held in gp[] (owned elements start here)       */
  const double *gp;     /* element index ei + mx*(ej + my*(ek - e_ks))                        */
  double E, nu;         /* FEMASM_MAT_UNIFORM                                                 */
  int32_t enr_mode;     /* FEMASM_ELASTIC3D_ENRICHED: 0 = SGFEM, phi = psi - I_h psi (`type` 1, R3:42);
                           1 = GFEM, phi = psi (`type` 0).  Record stride = 408 | 416 (R3:52-68 with 32- /
                           64-bit PetscInt) + 64 n_enr: per family psi at the 8 nodes, at the 8 Gauss points,
                           grad psi at the 8 Gauss points [p][3], psi at the 24 face Gauss points.
                           REQUIRED: the records of the (up to 8) elements around a node carry the same
                           enrich_type / n_enrich for that node (the engine reads them from ONE of these elements,
                           as R3's GetNodalRedundantDofs would); records that disagree give rows that differ between
                           a one-plan and a slab assembly, and no error is raised                  */
  int32_t reserved;
} femasm_material;

/* "quad": rule 0 = the reference's 2-point Gauss rule with the literal abscissa
 * 0.57735026919 and unit weights (P3:152-174, E2:124-135, S3:229-289). */
typedef struct { int32_t ngp; int32_t rule; } femasm_quad;

/* Dirichlet boundary conditions (A10, A11).  Rows only, no column elimination, no RHS
 * lifting.  Elastic kinds: diagonal 1.0, exactly MatZeroRows(A,n,ids,1.0,0,0).  Poisson:
 * the _ZERO_ROW_i macro acts on every ELEMENT matrix before ADD_VALUES (P3:286-290,
 * 328-340), so the assembled diagonal of a boundary row is the number of assembled
 * elements sharing the node (1, 2 or 4), not 1.  The zeroed rows keep their slots in the
 * DMDA pattern (explicit zeros); femasm_compress_bc_rows() gives PETSc's truncated
 * form for the elastic kinds (the Poisson driver has no MatZeroRows step). */
enum {
  FEMASM_BC_NONE = 0,
  FEMASM_BC_POISSON_INLOOP = 1,  /* P3:286-290, 328-338, 624-630                       */
  FEMASM_BC_E2_COMPRESSION = 2,  /* E2:877-887: i=M-1: ux=-1,uy=0 ; i=0: ux=+1,uy=0    */
  FEMASM_BC_S3_ALLFACES_EXACT = 3 /* S3:1912-1941: 6 faces x 3 dofs, exact displacement */
};
typedef struct {
  int32_t kind;
  int32_t symmetric;  /* 0 (default): what the reference does, rows only.  1: OPT-IN symmetric elimination, NOT reference
                         parity: after the rows-only system is complete every stored column c that is a Dirichlet dof
                         is eliminated from the other rows with right-hand-side lifting, b[r] -= K[r][c] g_c, K[r][c] = 0
                         (north_star (5) "row/column elimination"); K stays in the DMDA pattern and becomes symmetric, the
                         solution is unchanged.  One slab: done by femasm_assemble_system; several slabs: call
                         femasm_eliminate_columns after femasm_interface_apply. */
} femasm_bc;

typedef struct {
  double  ms_total;        /* whole call, host wall clock incl. staging               */
  double  ms_kernel_K;     /* stiffness kernel(s), HIP events on the plan's stream    */
  double  ms_kernel_rhs;   /* RHS kernels                                             */
  double  ms_staging;      /* host<->device copies (0 for device buffers)             */
  double  bytes_K;         /* algorithmic bytes of K assembly           */
  double  bytes_system;    /* algorithmic bytes of the full system                    */
  int64_t nel, nrows, nnz; /* local (slab) sizes                                      */
  int64_t bad_detJ;        /* Gauss points with det(J) <= 0 or NaN                    */
  double  ms_kernel_main;  /* the dominant stiffness kernel alone (interior rows in 3-D), HIP events */
  double  bytes_main;      /* its algorithmic bytes: values stored + coordinates and E,nu read once   */
} femasm_stats;

typedef struct femasm_plan femasm_plan;

/* Build the device-resident plan: slab geometry, DMDA sparsity pattern (bit-exact to
 * DMCreateMatrix, replaces nothing in the driver -- it is compared against P3:667,
 * E2:681, S3:1373), reference-element tables.  `device` is the HIP device ordinal. */
int femasm_plan_create(const femasm_mesh *mesh, const femasm_quad *quad, int device, femasm_plan **out);
int femasm_plan_destroy(femasm_plan *plan);

/* Launch everything on `stream` (a hipStream_t) instead of the plan's own stream. */
int femasm_plan_set_stream(femasm_plan *plan, void *stream);

/* No counterpart in the reference (an optimisation switch of this engine).  Every kind
 * has a specialised stiffness kernel besides the general one:
 *   level 2  rectilinear mesh (x depends on i only, y on j, z on k: what DMDASetUniformCoordinates
 *            P3:663 / E2:577 / S3:1198 produces; every element an axis-aligned box) with one
 *            (E,nu) per element
 *   level 1  Poisson: affine elements (parallelepipeds); elastic (S3/R3) and the enriched kind: any hexahedra with
 *            one (E,nu) per element (the material leaves the Gauss-point loop); plane strain: as level 0
 *   level 0  general isoparametric elements, (E,nu) per Gauss point
 * The premise is tested exactly on every node / element touched (bitwise-equal coordinates, edge
 * vectors and material); on any failure femasm_assemble_system reassembles K one level down
 * (elastic and enriched 2 -> 1 -> 0, and 2 -> 0 at once when the failed premise is the material's, which level 1
 * shares; Poisson 2 -> 0: its level 1 is kept for callers that set it, the general kernel is the faster one) and the
 * plan stays there.  Default 2.
 * Results of the levels differ by rounding only. */
int femasm_plan_set_fast_path(femasm_plan *plan, int level);
int femasm_plan_get_fast_path(femasm_plan *plan, int *level);
/* No counterpart in the reference.  FEMASM_ELASTIC3D_ENRICHED: how the enriched blocks are assembled.
 *   2 (= -1, the default)  one node-centric pass: a wave owns a node, contracts the blocks of the eight elements
 *                          around it on the FP64 VALU and writes the node's CSR block; no scratch memory
 *   0 / 1                  element matrices first (local B^T D B contraction on the FP64 VALU / on
 *                          v_mfma_f64_16x16x4, n_enr = 1: the 16 scalar shape functions of an element are one MFMA
 *                          tile) into a scratch ring, then a deterministic segmented reduction per node
 * Measured on MI355X in DESIGN.md section 9.  Results differ by rounding only. */
int femasm_plan_set_enriched_contraction(femasm_plan *plan, int mode);
/* No counterpart in the reference.  Kernel variants and tile parameters that the engine otherwise chooses by itself,
 * for A/B measurements and for the tests that keep the alternative kernels covered (DESIGN.md section 4 lists them).
 * The library reads NOTHING from the environment: this call is the only way to leave the automatic choice.
 * value 0 = automatic / off.  Names:
 *   "zc"              node layers per workgroup of the row-sweep, P3 box and S3 hexahedron kernels
 *   "s3_units"        x-consecutive units per wave of k_assemble_s3_unit
 *   "e2_yc"           node rows per wave of k_assemble_e2_rect
 *   "rhs_grid"        cap on the resident grid of the S3 right-hand-side pass
 *   "enr_scratch_mb"  enriched kind, two-kernel form: size of the element-matrix ring
 *   "s3_rowsweep"     1: elastic boxes through k_assemble_rowsweep<ELASTIC> instead of k_assemble_s3_unit
 *   "p3_sweep"        Poisson on general hexahedra: 1 the node-row kernel, 2 the element-matrix kernel of round 2
 *                     (k_assemble_ke_p3) instead of k_assemble_p3_hex4 (element matrices in tensor-product form)
 *   "e2_brick"        1: plane strain on general quadrilaterals through the brick kernel for every row
 *   "e2_lane_loads"   1: plane strain level 2, one-pass form: every lane loads its two elements' records itself
 *                     instead of the record stream through LDS
 *   "e2_two_pass"     1: plane strain on boxes, K and b in two passes
 *   "p3_hex2w"        1: Poisson on general hexahedra: k_assemble_p3_hex (every wave does every phase) instead of
 *                     k_assemble_p3_hex4 (element waves and row waves)
 *   "rhs_simple"      1: Poisson load on general hexahedra per Gauss point as the reference states it (library sinpi)
 *                     instead of the tensor-product form
 *   "enr_rhs_simple"  1: enriched kind, the per-element right-hand side
 *   "debug"           1: launch facts on stderr
 * Results differ by rounding at most (every variant is tested against the oracle). */
int femasm_plan_set_option(femasm_plan *plan, const char *name, int64_t value);

/* Sparsity pattern of the owned rows (A12): rowptr is local (starts at 0, int64 so
 * 256^3 elastic fits), colidx holds GLOBAL column ids (= PETSc natural numbering for
 * z-slab layouts).  Device pointers, borrowed until destroy. */
int femasm_pattern(femasm_plan *plan, int64_t *nrows, int64_t *nnz, const int64_t **rowptr, const int32_t **colidx);
/* Same, copied into caller-owned HOST arrays (rowptr[nrows+1], colidx[nnz]). */
int femasm_pattern_copy(femasm_plan *plan, int64_t *rowptr, int32_t *colidx);

/* Explicit element->nonzero scatter map (A13): for each owned element e (k->j->i
 * order) and local (row r, col c) of its (nen*dof)^2 matrix, the position in the
 * local csr_values array, or -1 if the row belongs to another slab.  The assembly
 * kernels use the closed form of this map; this entry materialises it for checks
 * and for unstructured front-ends.  map_out: HOST or DEVICE, nel*(nen*dof)^2 int64. */
int femasm_scatter_map(femasm_plan *plan, int64_t *map_out);

/* THE hot path.  Replaces
 *   P3:675 AssembleA_Poisson + P3:678 AssembleF_Poisson              (bc = POISSON_INLOOP)
 *   E2:692 AssembleA_Elasticity + E2:694 AssembleF_Elasticity + E2:700 DMDABCApplyCompression
 *   S3:1384 + S3:1386 + S3:1392  (R3:1402, 1404, 1410)
 * csr_values[nnz] and b[nrows] receive the owned rows (overwritten, not added to; the
 * drivers zero A and F first, P3:672-673).  Either may be NULL to skip that half.
 * On a slab with ks > 0 the partial sums this rank owes to the rank below are left in
 * the plan's send buffer (femasm_interface_buffers). */
int femasm_assemble_system(femasm_plan *plan, const femasm_material *material, const femasm_bc *bc,
                           double *csr_values, double *b, femasm_stats *stats);

/* Multi-GPU slab interface: what MatAssemblyEnd's stash exchange and
 * DMLocalToGlobal(ADD_VALUES) move in the reference (P3:345-346, 638-639; S3:710-711,
 * 794-795).  send_*: partial rows of node layer ks-1, owed to rank-1, in PACKED form -- they
 * come from one element layer, so only the columns in node layers ks-1 and ks can be non-zero:
 * per node, dof rows of those two z-thirds (18 dof of an interior row's 27 dof) -- then the
 * b slice.  recv_*: same shape for node layer ke-1, filled by the caller from rank+1
 * (femasm_interface_exchange, or the caller's own RCCL/MPI).  Counts are in doubles; 0 where
 * there is no neighbour. */
int femasm_interface_buffers(femasm_plan *plan, double **send_buf, int64_t *send_count,
                             double **recv_buf, int64_t *recv_count);
/* Use caller-owned DEVICE buffers (e.g. torch / PETSc device vectors registered with RCCL)
 * of at least send_count / recv_count doubles instead of the plan's own; NULL keeps it. */
int femasm_interface_set_buffers(femasm_plan *plan, double *send_buf, double *recv_buf);
/* Optional first half of a split call, for overlapping the exchange with the bulk of the work: computes ONLY
 * the partial rows owed to the rank below (the send buffer) and returns without waiting, so that the caller can
 * start its send on the plan's stream and then call femasm_assemble_system, which skips that part once.  Device
 * material pointers only.  Nothing is read back here: the buffer is final in stream order whatever the mesh (the
 * general kernels are queued behind the fast ones and run only if the fast kernels' premise failed), and geometry
 * errors (det J <= 0) are reported by the femasm_assemble_system call that must follow on the same inputs.
 * No counterpart in the reference (MatAssemblyBegin/End P3:345-346 is the closest). */
int femasm_assemble_interface(femasm_plan *plan, const femasm_material *material, const femasm_bc *bc);
/* The exchange itself, for C/C++ MPI drivers (one rank per GPU): replaces what MatAssemblyBegin/End's stash and
 * DMLocalToGlobalBegin/End(ADD_VALUES) move (P3:345-346, 638-639; S3:710-711, 794-795).  comm is the caller's
 * ncclComm_t (RCCL; created with ncclCommInitRank from an id the driver broadcasts over MPI); one ncclSend of the send
 * buffer to rank_below and one ncclRecv of the receive buffer from rank_above, grouped, on a communication stream the
 * plan owns: it starts when the kernels queued on the plan's stream so far (femasm_assemble_interface, or all of
 * femasm_assemble_system) have filled the send buffer, and what is queued on the plan's stream afterwards -- the bulk of
 * the assembly -- does not wait for it.  Pass -1 where there is no neighbour.  Returns without waiting;
 * femasm_interface_apply makes the plan's stream wait for the pair before it adds the received rows.
 * RCCL is looked up at run time (already loaded by the driver, or librccl.so): the library does not link against it. */
int femasm_interface_exchange(femasm_plan *plan, void *comm, int rank_below, int rank_above);
/* own + received, fixed order (deterministic); Dirichlet rows are left untouched. */
int femasm_interface_apply(femasm_plan *plan, const femasm_bc *bc, double *csr_values, double *b);

/* The symmetric (row AND column) elimination of femasm_bc.symmetric as a call of its own: csr_values / b must hold the
 * complete rows-only system of this slab (after femasm_interface_apply), DEVICE pointers; b may be NULL (columns are
 * zeroed, nothing lifted).  No counterpart in the reference (MatZeroRowsColumns would be PETSc's). */
int femasm_eliminate_columns(femasm_plan *plan, const femasm_bc *bc, double *csr_values, double *b);

/* PETSc's default MatZeroRows storage (keepnonzeropattern off): zeroed rows keep only
 * their diagonal and the CSR arrays are recompressed.  Outputs HOST or DEVICE,
 * sized like the inputs; *nnz_out receives the new count.  Elastic BC kinds only
 * (FEMASM_BC_POISSON_INLOOP is rejected: P3 never calls MatZeroRows). */
int femasm_compress_bc_rows(femasm_plan *plan, const femasm_bc *bc, const double *csr_values,
                            int64_t *rowptr_out, int32_t *colidx_out, double *values_out, int64_t *nnz_out);

/* A11 in its general form, for Dirichlet sets other than the three built-in kinds: what
 * MatZeroRows(A, n, rows, 1.0, NULL, NULL) + VecSetValues(b, n, rows, vals, INSERT_VALUES) do (S3:1497-1512,
 * E2:741-875): every listed GLOBAL row becomes the unit row (its slots stay in the pattern as explicit zeros) and
 * b[row] = vals[i] (0 if vals is NULL).  Rows of other slabs are ignored; a row listed twice takes one of its values
 * (the drivers list edge nodes several times with the same value).  rows/vals: HOST or DEVICE; csr_values and b:
 * DEVICE, either may be NULL.  Assemble with FEMASM_BC_NONE first. */
int femasm_zero_rows(femasm_plan *plan, int64_t n, const int64_t *rows, const double *vals, double *csr_values,
                     double *b);

/* y = K x without forming K, from the same element arithmetic (any hexahedra / quads,
 * (E,nu) per Gauss point), Dirichlet rows as in the assembled matrix (row = diag * e_row, no column
 * elimination).  x: DEVICE vector on the node layers [cks,cke) of the plan's coordinate array; y: DEVICE, the
 * owned rows.  On a slab with ks > 0 the partial y of node layer ks-1 is left in the b part of the send buffer:
 * exchange it and call femasm_interface_apply(plan, bc, NULL, y), as after an assembly.  Deterministic. */
int femasm_apply_operator(femasm_plan *plan, const femasm_material *material, const femasm_bc *bc, const double *x,
                          double *y);

/* SURVEY 8(f) row N1, the step before the path: the drivers' fill of the GaussPointCoefficients array
 * (S3:1243-1363 / R3:1254-1380: Gauss-point and face-Gauss-point positions, E, nu, body force and displacement
 * gradient of the manufactured solution; E2:618-670: positions, E, nu, zero body force) for this slab's
 * element layers [max(ks-1,0), ke-1), element index as in femasm_material with e_ks = that first layer.
 * gp_out: HOST or DEVICE, nel*stride doubles; doubles beyond 400 (R3) are zeroed (enrich_type = 0). */
int femasm_fill_properties(femasm_plan *plan, double E, double nu, int32_t stride, double *gp_out);

/* SURVEY 8(f) row N3, the step after the solve: the element integrals of DMDAIntegrateErrors3D (P3:416-532,
 * S3:1041-1151) over this slab's elements against the drivers' manufactured solutions.  u: the solution on the
 * node layers [cks,cke) of the plan's coordinate array (the ghosted local vector of DMGlobalToLocal, P3:437),
 * HOST or DEVICE.  sums[0] = sum of the L2 integrals, sums[1] = of the H1 integrals; the caller adds the ranks'
 * sums (MPI_Allreduce, P3:510-513) and takes the square roots; h = 1/(M-1).  3-D kinds only. */
int femasm_error_integrals(femasm_plan *plan, const double *u, double sums[2]);

/* No counterpart in the reference.  femasm_error_integrals for callers that cannot allocate: code under HIP graph
 * capture, tight solver loops, harnesses that forbid hipMalloc.  Same kernels and the same bits; the per-workgroup
 * partial sums live in a static device array inside the library (1 MiB per GPU, held until the process ends) instead
 * of plan-owned memory, u and sums are DEVICE pointers (sums[2] is written in stream order, nothing is copied back
 * and the call does not wait), and no hipMalloc / hipFree / host copy happens.  Limit: at most 16 777 216 elements
 * (256^3) in the plan's slab, larger slabs are rejected with FEMASM_ERR_ARG.  Every plan holds a slice of its GPU's
 * array of its own from femasm_plan_create to femasm_plan_destroy (32 slots of 524 288 elements, first fit), so plans
 * driven on different streams do not meet; a plan for which no slice is left is rejected with FEMASM_ERR_ARG (use
 * femasm_error_integrals).  Calls on ONE plan that overlap (two streams, two host threads) still race -- order them. */
int femasm_error_integrals_static(femasm_plan *plan, const double *u, double *sums);

/* SURVEY 8(f) row N2, the step after the path: landing K in a (device-resident) PETSc Mat without the host.
 * Either hand femasm_assemble_system the Mat's own value array (MatSeqAIJGetCSRAndMemType /
 * MatSeqAIJGetArray: the pattern is DMCreateMatrix's), or use PETSc's COO interface: this entry writes the
 * GLOBAL row and column index of every slot of csr_values, in that array's order, for
 * MatSetPreallocationCOO(A, nnz, coo_i, coo_j) once and MatSetValuesCOO(A, csr_values, INSERT_VALUES) per
 * assembly (call sites P3:681-685, S3:1394-1397).  index_bytes = sizeof(PetscInt) (4 or 8); HOST or DEVICE. */
int femasm_coo_indices(femasm_plan *plan, int32_t index_bytes, void *coo_i, void *coo_j);

/* SURVEY 8(f) row N4, second half: the rigid-body near-nullspace vectors the elasticity drivers hand to the
 * preconditioner (MatNullSpaceCreateRigidBody on the DMDA coordinates, E2:682-685, S3:1374-1377): translations
 * 1/sqrt(Nnodes) e_k, then the rotations (y,-x,0), (0,-z,y), (z,0,-x) (2-D: (-y,x)), each orthogonalised
 * against the vectors before it and normalised (PETSc's published algorithm; PETSc is not in the reference
 * tree: parity unpinned).  Two calls so that the only global reduction stays with the caller:
 *   femasm_coord_moments: mom = sums over the OWNED nodes of [1, x, y, z, xx, yy, zz, xy, yz, zx]; the caller
 *     adds the ranks' arrays (MPI_Allreduce / RCCL all-reduce of 10 doubles; nothing to do on one GPU);
 *   femasm_rigid_body_modes: modes[m*nrows + r], m < 6 (3 in 2-D), owned rows r, from the GLOBAL moments;
 *     HOST or DEVICE. */
int femasm_coord_moments(femasm_plan *plan, double mom[10]);
int femasm_rigid_body_modes(femasm_plan *plan, const double mom[10], double *modes);

/* Block until everything queued on the plan's stream has finished; in asynchronous mode also the place where the
 * errors of the calls before it are reported (FEMASM_ERR_GEOMETRY ...). */
int femasm_plan_sync(femasm_plan *plan);

/* No counterpart in the reference.  on != 0: femasm_assemble_system (device pointers, stats == NULL; P3 / E2 / S3 kinds)
 * and femasm_interface_apply queue their kernels and return without waiting for the device, so that a driver can post
 * the exchange, start the next slab or set up its solver meanwhile.  The results are final in stream order whatever the
 * mesh: where a verified fast kernel is used, the general kernels are queued behind it and run on the device only if
 * its premise failed.  Geometry errors of such a call are returned by femasm_plan_sync or by the next call on the plan
 * (when the premise failed, both the fast and the general kernels saw the mesh: the number of bad Gauss points in that
 * message counts a point once per kernel that met it -- an upper bound, not the synchronous mode's exact count). */
int femasm_plan_set_async(femasm_plan *plan, int on);

const char *femasm_last_error(void);
const char *femasm_version(void);

#ifdef __cplusplus
}
#endif
#endif
