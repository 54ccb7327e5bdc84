/*
 * san_check.c -- the CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (host code only; SURVEY.md
 * section 5: "CI run of the CPU restatement under -fsanitize=address,undefined").  TEST INFRASTRUCTURE: built by
 * `make -C oracle san_check` into oracle/san_check and run by tests/test_oracle_golden.py.  It drives every assembly
 * routine of femasm_oracle.c on small meshes (distorted, per-Gauss-point data, enriched records of both PetscInt
 * widths) and checks two SURVEY 8(c) golden values on the way, so that the instrumented code is the code the parity
 * tests trust.  Exit code 0 = no sanitizer report (they abort) and the values agree.
 */
#include "femasm_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double *dalloc(size_t n) { double *p = (double *)calloc(n ? n : 1, sizeof(double)); if (!p) exit(3); return p; }

static void perturb(double *c, int nsd, int64_t M, int64_t N, int64_t P)
{
    for (int64_t k = 0; k < P; k++)
        for (int64_t j = 0; j < N; j++)
            for (int64_t i = 0; i < M; i++) {
                int in = i > 0 && i < M - 1 && j > 0 && j < N - 1 && (nsd == 2 || (k > 0 && k < P - 1));
                double *x = c + nsd * (i + M * (j + N * k));
                if (in) for (int d = 0; d < nsd; d++) x[d] += 0.1 / (double)(M - 1) * sin(7.0 * x[(d + 1) % nsd] + d);
            }
}

int main(void)
{
    int fail = 0;
    /* golden element values of SURVEY 8(c) */
    {
        double h = 1.0 / 32, c[24], Ke[64];
        static const int di[8] = {0, 0, 1, 1, 0, 0, 1, 1}, dj[8] = {0, 1, 1, 0, 0, 1, 1, 0}, dk[8] = {0, 0, 0, 0, 1, 1, 1, 1};
        for (int n = 0; n < 8; n++) { c[3 * n] = h * di[n]; c[3 * n + 1] = h * dj[n]; c[3 * n + 2] = h * dk[n]; }
        memset(Ke, 0, sizeof Ke);
        orc_ke_poisson_hex(c, Ke);
        if (Ke[0] != 0.010416666666673416) { printf("P3 golden Ke[0] = %.17g\n", Ke[0]); fail = 1; }
        double E[8], nu[8], K3[576];
        h = 1.0 / 128;
        for (int n = 0; n < 8; n++) { c[3 * n] = h * di[n]; c[3 * n + 1] = h * dj[n]; c[3 * n + 2] = h * dk[n]; E[n] = 90.0; nu[n] = 0.28; }
        memset(K3, 0, sizeof K3);
        orc_ke_elastic_hex(c, E, nu, K3);
        if (K3[0] != 0.16091086647737698) { printf("S3 golden Ke[0] = %.17g\n", K3[0]); fail = 1; }
    }
    /* P3 */
    {
        int64_t M = 5, N = 4, P = 6, nn = M * N * P, nnz = orc_dmda_nnz(3, 1, M, N, P);
        double *co = dalloc(3 * nn), *K = dalloc(nnz), *b = dalloc(nn);
        int64_t *rp = (int64_t *)calloc(nn + 1, sizeof(int64_t));
        int32_t *ci = (int32_t *)calloc(nnz, sizeof(int32_t));
        orc_dmda_uniform_coords(3, M, N, P, co);
        perturb(co, 3, M, N, P);
        orc_dmda_pattern(3, 1, M, N, P, rp, ci);
        if (rp[nn] != nnz || orc_assemble_poisson3d(M, N, P, co, rp, ci, K, b, 0, P - 1)) fail = 1;
        double h, l2, h1;
        orc_errors_poisson3d(M, N, P, co, b, &h, &l2, &h1);
        free(co); free(K); free(b); free(rp); free(ci);
    }
    /* E2 */
    {
        int64_t M = 7, N = 5, nn = M * N, nnz = orc_dmda_nnz(2, 2, M, N, 1);
        double *co = dalloc(2 * nn), *K = dalloc(nnz), *b = dalloc(2 * nn), *pr = dalloc(24 * (M - 1) * (N - 1));
        int64_t *rp = (int64_t *)calloc(2 * nn + 1, sizeof(int64_t)), *rp2 = (int64_t *)calloc(2 * nn + 1, sizeof(int64_t));
        int32_t *ci = (int32_t *)calloc(nnz, sizeof(int32_t)), *ci2 = (int32_t *)calloc(nnz, sizeof(int32_t));
        uint8_t *fl = (uint8_t *)calloc(2 * nn, 1);
        double *v2 = dalloc(nnz);
        orc_dmda_uniform_coords(2, M, N, 1, co);
        perturb(co, 2, M, N, 1);
        orc_dmda_pattern(2, 2, M, N, 1, rp, ci);
        orc_props_e2_fill(M - 1, N - 1, co, 1.0, 0.33, pr);
        if (orc_assemble_elastic2d(M, N, co, pr, 24, rp, ci, K, b, 0, N - 1)) fail = 1;
        orc_bc_e2_compression(M, N, rp, ci, K, b);
        orc_bc_flags_e2(M, N, fl);
        if (orc_compress_zeroed_rows(2 * nn, fl, rp, ci, K, rp2, ci2, v2) <= 0) fail = 1;
        free(co); free(K); free(b); free(pr); free(rp); free(rp2); free(ci); free(ci2); free(fl); free(v2);
    }
    /* S3 and the enriched kind (n_enr = 0..2, both PetscInt widths) */
    for (int n_enr = 0; n_enr <= 2; n_enr++)
        for (int base = 408; base <= 416; base += 8) {
            int64_t M = 4, N = 3, P = 4, nn = M * N * P, nel = (M - 1) * (N - 1) * (P - 1);
            int D = 3 * (1 + n_enr), stride = base + 64 * n_enr;
            int64_t nnz = orc_dmda_nnz(3, D, M, N, P);
            double *co = dalloc(3 * nn), *K = dalloc(nnz), *b = dalloc(D * nn), *s3 = dalloc(400 * nel), *pr = dalloc((size_t)stride * nel);
            int64_t *rp = (int64_t *)calloc(D * nn + 1, sizeof(int64_t));
            int32_t *ci = (int32_t *)calloc(nnz, sizeof(int32_t));
            orc_dmda_uniform_coords(3, M, N, P, co);
            perturb(co, 3, M, N, P);
            orc_dmda_pattern(3, D, M, N, P, rp, ci);
            orc_props_s3_fill(M - 1, N - 1, P - 1, co, 90.0, 0.28, s3, 400);
            for (int64_t e = 0; e < nel; e++) {
                double *q = pr + (size_t)stride * e;
                memcpy(q, s3 + 400 * e, 400 * sizeof(double));
                for (int a = 0; a < 16; a++) {
                    int64_t v = (a < 8) ? ((e + a) % 3 != 0) : (int64_t)((e + a) % (n_enr + 1));
                    if (base == 408) ((int32_t *)(q + 400))[a] = (int32_t)v; else ((int64_t *)(q + 400))[a] = v;
                }
                for (int t = base; t < stride; t++) q[t] = 0.3 + 0.01 * (double)((t * 7 + e) % 11);
            }
            if (n_enr == 0 && base == 408) {
                double *K3 = dalloc(nnz), *b3 = dalloc(3 * nn);
                if (orc_assemble_elastic3d(M, N, P, co, pr, stride, rp, ci, K3, b3, 0, P - 1)) fail = 1;
                orc_bc_s3_allfaces(M, N, P, co, rp, ci, K3, b3);
                double h, l2, h1;
                orc_errors_elastic3d(M, N, P, co, b3, &h, &l2, &h1);
                free(K3); free(b3);
            }
            if (orc_assemble_enriched3d(M, N, P, co, pr, stride, base, n_enr, base == 408, rp, ci, K, b, 0, P - 1)) fail = 1;
            orc_bc_enriched3d(M, N, P, co, pr, stride, base, n_enr, 1, rp, ci, K, b);
            for (int64_t t = 0; t < nnz; t++) if (!(K[t] == K[t])) fail = 1;
            free(co); free(K); free(b); free(s3); free(pr); free(rp); free(ci);
        }
    printf(fail ? "san_check: FAILED\n" : "san_check: ok\n");
    return fail;
}
