/*
 * femasm_oracle.h -- CPU restatement of the reference's FEM assembly hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under petsc_amd/ (the product) may include,
 * link or call this.  Allowed users: tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py -- as the checker / baseline, never as the thing
 * shipped or measured as the product.
 *
 * Every function follows the arithmetic of the reference driver files and cites
 * the lines it restates (abbreviations as in SURVEY.md):
 *   P3 = FEM_poisson3D/FEM.cpp        E2 = FEM_elsatic2D/FEM.cpp
 *   S3 = SGFEM/SGFEM_elastic3D.cpp    R3 = SGFEM_real/SGFEM_elastic3D.cpp
 *
 * PETSc library semantics (DMCreateMatrix pattern, MatSetValuesStencil index
 * map, MatZeroRows, DMDASetUniformCoordinates, DMDA ownership split) are NOT in
 * /root/reference (un-vendored, un-pinned dependency, roughly PETSc 3.8-3.18);
 * they are restated from PETSc's published behaviour.
 *
 * PARITY PINNING: the reference cannot be built here (needs PETSc headers and
 * libraries that are absent; no stand-ins are written).  The element kernels are
 * pinned by the golden values recorded in SURVEY.md section 8(c)
 * (tests/golden/survey_8c.json) and end-to-end by the reference's only known-answer
 * check (manufactured-solution error falling O(h^2), P3:524 / S3:1142).  Anything
 * those do not cover -- PETSc pattern/BC semantics in particular -- is
 * "parity unpinned".
 */
#ifndef FEMASM_ORACLE_H
#define FEMASM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- reference-element tables and geometry (A1-A5) ---------------------- */
void orc_hex_N(const double xi[3], double N[8]);                       /* P3:31-46  S3:107-122 */
void orc_hex_dNdxi(const double xi[3], double G[3][8]);                /* P3:48-83  S3:125-160 */
void orc_inv3x3(const double A[3][3], double B[3][3]);                 /* P3:86-107 S3:162-183 */
void orc_hex_dNdx(const double G[3][8], const double coords[24],
                  double Gx[3][8], double *detJ);                       /* P3:109-150 S3:186-227 */
void orc_gauss_hex8(double xi[8][3], double w[8]);                     /* P3:152-174 S3:229-251 */
void orc_gauss_hexface24(double xi[24][3], double w[24]);              /* S3:253-289 */
void orc_quad_N(const double xi[2], double N[4]);                      /* E2:64-73 */
void orc_quad_dNdxi(const double xi[2], double G[2][4]);               /* E2:75-89 */
void orc_quad_dNdx(const double G[2][4], const double coords[8],
                   double Gx[2][4], double *detJ);                      /* E2:91-121 */
void orc_gauss_quad4(double xi[4][2], double w[4]);                    /* E2:124-135 */

/* ---- element kernels (A6-A9b); all ACCUMULATE into their output -------- */
void orc_ke_poisson_hex(const double coords[24], double Ke[64]);        /* P3:257-284 */
void orc_fe_poisson_hex(const double coords[24], double Fe[8]);         /* P3:534-567, 352-359 */
void orc_ke_planestrain_quad(const double coords[8], const double E[4],
                             const double nu[4], double Ke[64]);        /* E2:266-323 */
void orc_fe_momentum_quad(const double coords[8], const double fx[4],
                          const double fy[4], double Fe[8]);            /* E2:325-350 */
void orc_ke_elastic_hex(const double coords[24], const double E[8],
                        const double nu[8], double Ke[576]);            /* S3:450-518 R3:454-522 */
void orc_fe_momentum_hex(const double coords[24], const double fx[8],
                         const double fy[8], const double fz[8],
                         double Fe[24]);                                /* S3:520-546 */
void orc_ge_naturalbc_hex(const double coords[24], const double gx[72],
                          const double gy[72], const double gz[72],
                          const double E_B[24], const double nu_B[24],
                          double Ge[24]);                               /* S3:548-656 */

/* ---- manufactured solutions (driver side; used to build inputs) -------- */
double orc_poisson_exact(const double pos[3]);                          /* P3:361-368 */
void   orc_elastic_exact(const double pos[3], double vel[3], double Fm[3],
                         double Gm[3][3], double E, double nu);         /* S3:958-985 */

/* ---- structured mesh, PETSc DMDA semantics (A12, A13) ------------------- */
/* DMDASetUniformCoordinates(da,0,1,0,1,0,1): x_i = 0 + i*((1-0)/(M-1)). interleaved. */
void    orc_dmda_uniform_coords(int nsd, int64_t M, int64_t N, int64_t P, double *coords);
int64_t orc_dmda_nnz(int nsd, int dof, int64_t M, int64_t N, int64_t P);
/* DMCreateMatrix on DMDA BOX stencil width 1, non-periodic, AIJ: sorted columns. */
void    orc_dmda_pattern(int nsd, int dof, int64_t M, int64_t N, int64_t P,
                         int64_t *rowptr, int32_t *colidx);

/* ---- GaussPointCoefficients producer (driver side, A14) ----------------- */
/* E2:618-670; stride 24 doubles per element: gp_coords[8],E[4],nu[4],fx[4],fy[4] */
void orc_props_e2_fill(int64_t mx, int64_t my, const double *coords,
                       double E, double nu, double *props);
/* S3:1243-1363; stride>=400 doubles per element:
 * gp_coords[24],gpb_coords[72],E[8],nu[8],E_B[24],nu_B[24],fx[8],fy[8],fz[8],gx[72],gy[72],gz[72] */
void orc_props_s3_fill(int64_t mx, int64_t my, int64_t mz, const double *coords,
                       double E, double nu, double *props, int stride);

/* ---- global assembly in the reference's order (A10, A13, A13b) ---------- */
/* Element z-layers (y-rows in 2-D) [e0,e1) are assembled, in k->j->i order, with
 * MatSetValues(ADD_VALUES) semantics into the CSR arrays; vals/b must be zeroed by
 * the caller (MatZeroEntries/VecZeroEntries, P3:672-673).  Returns 0, or -1 if a
 * column is missing from the pattern. */
int orc_assemble_poisson3d(int64_t M, int64_t N, int64_t P, const double *coords,
                           const int64_t *rowptr, const int32_t *colidx,
                           double *vals, double *b, int64_t e0, int64_t e1); /* P3:293-350, 581-644 */
int orc_assemble_elastic2d(int64_t M, int64_t N, const double *coords,
                           const double *props, int stride,
                           const int64_t *rowptr, const int32_t *colidx,
                           double *vals, double *b, int64_t e0, int64_t e1); /* E2:390-447, 461-529 */
int orc_assemble_elastic3d(int64_t M, int64_t N, int64_t P, const double *coords,
                           const double *props, int stride,
                           const int64_t *rowptr, const int32_t *colidx,
                           double *vals, double *b, int64_t e0, int64_t e1); /* S3:658-803 */

/* ---- post-assembly Dirichlet BCs (A11), MatZeroRows(diag=1) + VecSetValues INSERT.
 * keep-pattern form: rows are zeroed in place.  */
void orc_bc_e2_compression(int64_t M, int64_t N, const int64_t *rowptr,
                           const int32_t *colidx, double *vals, double *b);   /* E2:741-887 */
void orc_bc_s3_allfaces(int64_t M, int64_t N, int64_t P, const double *coords,
                        const int64_t *rowptr, const int32_t *colidx,
                        double *vals, double *b);                             /* S3:1451-1941 */
/* PETSc AIJ default (keepnonzeropattern off): a zeroed row keeps only its diagonal
 * and the arrays are recompressed.  flags[r]!=0 marks a zeroed row.  Returns new nnz. */
int64_t orc_compress_zeroed_rows(int64_t nrows, const uint8_t *flags,
                                 const int64_t *rowptr, const int32_t *colidx, const double *vals,
                                 int64_t *rowptr_out, int32_t *colidx_out, double *vals_out);
void orc_bc_flags_e2(int64_t M, int64_t N, uint8_t *flags);
void orc_bc_flags_s3(int64_t M, int64_t N, int64_t P, uint8_t *flags);

/* ---- error norms (N3; the reference's only known-answer print) ---------- */
void orc_errors_poisson3d(int64_t M, int64_t N, int64_t P, const double *coords,
                          const double *x, double *h, double *L2, double *H1); /* P3:416-532 */
void orc_errors_elastic3d(int64_t M, int64_t N, int64_t P, const double *coords,
                          const double *x, double *h, double *L2, double *H1); /* S3:1041-1151 */

/* ---- config 5: SGFEM-enriched hexahedra.  NO REFERENCE ARITHMETIC EXISTS; this restates
 * the build's own specification (femasm_oracle.c has the short form), shaped by R3:39-49,
 * 52-68, 1265-1269, 1467-1595.  base = 408 | 416 (width of PetscInt), record stride = base + 64 n_enr. */
void orc_enr_ke(const double coords[24], const double *rec, int base, int n_enr, int sgfem, double *Ke);
void orc_enr_fe(const double coords[24], const double *rec, int base, int n_enr, int sgfem, double *F);
int  orc_assemble_enriched3d(int64_t M, int64_t N, int64_t P, const double *coords, const double *props, int stride,
                             int base, int n_enr, int sgfem, const int64_t *rowptr, const int32_t *colidx,
                             double *vals, double *b, int64_t e0, int64_t e1);
void orc_bc_enriched3d(int64_t M, int64_t N, int64_t P, const double *coords, const double *props, int stride,
                       int base, int n_enr, int dirichlet, const int64_t *rowptr, const int32_t *colidx,
                       double *vals, double *b);

#ifdef __cplusplus
}
#endif
#endif
