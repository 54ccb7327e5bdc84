/*
 * femasm_oracle.c -- CPU restatement of the reference's FEM assembly hot path.
 * TEST INFRASTRUCTURE ONLY (see femasm_oracle.h for the rules and the parity
 * pinning status).  Build with -ffp-contract=off so the operation order below
 * is the operation order executed (the reference is plain x86-64 C++).
 *
 * The restatement is table-driven (sign tables for the Q1 bases instead of the
 * reference's spelled-out lines) but keeps, product by product, the evaluation
 * order of the reference expressions: multiplying by +-1.0 and a + (-b) are exact
 * in IEEE-754, so the results are bit-identical to the spelled-out forms.
 */
#include "femasm_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

#define ORC_PI 3.14159265358979323846264338327950288 /* PETSC_PI */
#define ORC_G  0.57735026919                         /* literal of P3:155, E2:127, S3:232 */

/* local node -> (di,dj,dk) offsets; A13: P3:181-201, S3:296-338 */
static const int HEX_DI[8] = {0, 0, 1, 1, 0, 0, 1, 1};
static const int HEX_DJ[8] = {0, 1, 1, 0, 0, 1, 1, 0};
static const int HEX_DK[8] = {0, 0, 0, 0, 1, 1, 1, 1};
/* same thing as signs of the reference coordinates of each node */
static const double HEX_SX[8] = {-1, -1, 1, 1, -1, -1, 1, 1};
static const double HEX_SY[8] = {-1, 1, 1, -1, -1, 1, 1, -1};
static const double HEX_SZ[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
/* E2:357-377 */
static const int QUAD_DI[4] = {0, 0, 1, 1};
static const int QUAD_DJ[4] = {0, 1, 1, 0};
static const double QUAD_SX[4] = {-1, -1, 1, 1};
static const double QUAD_SY[4] = {-1, 1, 1, -1};

/* ------------------------------------------------------------------------ */
void orc_hex_N(const double xi[3], double N[8])
{
    for (int n = 0; n < 8; n++)
        N[n] = 0.125 * (1.0 + HEX_SX[n] * xi[0]) * (1.0 + HEX_SY[n] * xi[1]) * (1.0 + HEX_SZ[n] * xi[2]);
}

void orc_hex_dNdxi(const double xi[3], double G[3][8])
{
    for (int n = 0; n < 8; n++) {
        double ax = 1.0 + HEX_SX[n] * xi[0];
        double ay = 1.0 + HEX_SY[n] * xi[1];
        double az = 1.0 + HEX_SZ[n] * xi[2];
        G[0][n] = (HEX_SX[n] * 0.125) * ay * az;
        G[1][n] = (HEX_SY[n] * 0.125) * ax * az;
        G[2][n] = (HEX_SZ[n] * 0.125) * ax * ay;
    }
}

void orc_inv3x3(const double A[3][3], double B[3][3])
{
    double t4 = A[2][0] * A[0][1];
    double t6 = A[2][0] * A[0][2];
    double t8 = A[1][0] * A[0][1];
    double t10 = A[1][0] * A[0][2];
    double t12 = A[0][0] * A[1][1];
    double t14 = A[0][0] * A[1][2];
    double t17 = 0.1e1 / (t4 * A[1][2] - t6 * A[1][1] - t8 * A[2][2] + t10 * A[2][1] + t12 * A[2][2] - t14 * A[2][1]);

    B[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) * t17;
    B[0][1] = -(A[0][1] * A[2][2] - A[0][2] * A[2][1]) * t17;
    B[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * t17;
    B[1][0] = -(-A[2][0] * A[1][2] + A[1][0] * A[2][2]) * t17;
    B[1][1] = (-t6 + A[0][0] * A[2][2]) * t17;
    B[1][2] = -(-t10 + t14) * t17;
    B[2][0] = (-A[2][0] * A[1][1] + A[1][0] * A[2][1]) * t17;
    B[2][1] = -(-t4 + A[0][0] * A[2][1]) * t17;
    B[2][2] = (-t8 + t12) * t17;
}

void orc_hex_dNdx(const double G[3][8], const double coords[24], double Gx[3][8], double *detJ)
{
    double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, iJ[3][3];
    for (int n = 0; n < 8; n++)
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) J[r][c] = J[r][c] + G[r][n] * coords[3 * n + c];

    orc_inv3x3(J, iJ);

    *detJ = J[0][0] * J[1][1] * J[2][2] - J[0][0] * J[1][2] * J[2][1] - J[1][0] * J[0][1] * J[2][2]
          + J[1][0] * J[0][2] * J[2][1] + J[2][0] * J[0][1] * J[1][2] - J[2][0] * J[0][2] * J[1][1];

    for (int n = 0; n < 8; n++)
        for (int r = 0; r < 3; r++)
            Gx[r][n] = G[0][n] * iJ[r][0] + G[1][n] * iJ[r][1] + G[2][n] * iJ[r][2];
}

void orc_gauss_hex8(double xi[8][3], double w[8])
{
    /* Gauss point p sits at the reference position of node p, scaled by ORC_G */
    for (int p = 0; p < 8; p++) {
        xi[p][0] = HEX_SX[p] * ORC_G;
        xi[p][1] = HEX_SY[p] * ORC_G;
        xi[p][2] = HEX_SZ[p] * ORC_G;
        w[p] = 1.0;
    }
}

void orc_gauss_hexface24(double xi[24][3], double w[24])
{
    /* faces in the order xi=+1, xi=-1, eta=+1, eta=-1, zeta=+1, zeta=-1; inside a face
     * the first free axis alternates fastest for the xi faces (eta), and the LOWER
     * axis alternates fastest for the others -- S3:256-284 */
    static const double s0[4] = {-1, 1, -1, 1};
    static const double s1[4] = {-1, -1, 1, 1};
    for (int f = 0; f < 6; f++) {
        int axis = f / 2;
        double side = (f % 2 == 0) ? 1.0 : -1.0;
        int a0 = (axis == 0) ? 1 : 0;           /* fastest free axis */
        int a1 = (axis == 2) ? 1 : 2;           /* slowest free axis */
        for (int q = 0; q < 4; q++) {
            int p = 4 * f + q;
            xi[p][axis] = side;
            xi[p][a0] = s0[q] * ORC_G;
            xi[p][a1] = s1[q] * ORC_G;
            w[p] = 1.0;
        }
    }
}

void orc_quad_N(const double xi[2], double N[4])
{
    for (int n = 0; n < 4; n++) N[n] = 0.25 * (1.0 + QUAD_SX[n] * xi[0]) * (1.0 + QUAD_SY[n] * xi[1]);
}

void orc_quad_dNdxi(const double xi[2], double G[2][4])
{
    for (int n = 0; n < 4; n++) {
        G[0][n] = (QUAD_SX[n] * 0.25) * (1.0 + QUAD_SY[n] * xi[1]);
        G[1][n] = (QUAD_SY[n] * 0.25) * (1.0 + QUAD_SX[n] * xi[0]);
    }
}

void orc_quad_dNdx(const double G[2][4], const double coords[8], double Gx[2][4], double *detJ)
{
    double J00 = 0, J01 = 0, J10 = 0, J11 = 0;
    for (int i = 0; i < 4; i++) {
        double cx = coords[2 * i], cy = coords[2 * i + 1];
        J00 = J00 + G[0][i] * cx;
        J01 = J01 + G[0][i] * cy;
        J10 = J10 + G[1][i] * cx;
        J11 = J11 + G[1][i] * cy;
    }
    double J = (J00 * J11) - (J01 * J10);
    double iJ00 = J11 / J, iJ01 = -J01 / J, iJ10 = -J10 / J, iJ11 = J00 / J;
    for (int i = 0; i < 4; i++) {
        Gx[0][i] = G[0][i] * iJ00 + G[1][i] * iJ01;
        Gx[1][i] = G[0][i] * iJ10 + G[1][i] * iJ11;
    }
    if (detJ) *detJ = J;
}

void orc_gauss_quad4(double xi[4][2], double w[4])
{
    for (int p = 0; p < 4; p++) {
        xi[p][0] = QUAD_SX[p] * ORC_G;
        xi[p][1] = QUAD_SY[p] * ORC_G;
        w[p] = 1.0;
    }
}

/* ---- element kernels ---------------------------------------------------- */
void orc_ke_poisson_hex(const double coords[24], double Ke[64])
{
    double xi[8][3], w[8], G[3][8], Gx[3][8], J;
    orc_gauss_hex8(xi, w);
    for (int p = 0; p < 8; p++) {
        orc_hex_dNdxi(xi[p], G);
        orc_hex_dNdx(G, coords, Gx, &J);
        for (int i = 0; i < 8; i++)
            for (int j = 0; j < 8; j++)
                for (int k = 0; k < 3; k++) Ke[i * 8 + j] += Gx[k][i] * Gx[k][j] * w[p] * J;
    }
}

static double poisson_forcing(const double pos[3])
{
    return 3.0 * ORC_PI * ORC_PI * sin(ORC_PI * pos[0]) * sin(ORC_PI * pos[1]) * sin(ORC_PI * pos[2]);
}

void orc_fe_poisson_hex(const double coords[24], double Fe[8])
{
    double xi[8][3], w[8], N[8], G[3][8], Gx[3][8], J;
    orc_gauss_hex8(xi, w);
    for (int p = 0; p < 8; p++) {
        orc_hex_N(xi[p], N);
        orc_hex_dNdxi(xi[p], G);
        orc_hex_dNdx(G, coords, Gx, &J);
        double fac = w[p] * J;
        double pos[3] = {0.0, 0.0, 0.0};
        for (int n = 0; n < 8; n++) {
            pos[0] += N[n] * coords[3 * n];
            pos[1] += N[n] * coords[3 * n + 1];
            pos[2] += N[n] * coords[3 * n + 2];
        }
        double fv = poisson_forcing(pos);
        for (int i = 0; i < 8; i++) Fe[i] += fac * fv * N[i];
    }
}

void orc_ke_planestrain_quad(const double coords[8], const double E[4], const double nu[4], double Ke[64])
{
    double xi[4][2], w[4], G[2][4], Gx[2][4], J, B[3][8], D[3][3];
    orc_gauss_quad4(xi, w);
    for (int p = 0; p < 4; p++) {
        orc_quad_dNdxi(xi[p], G);
        orc_quad_dNdx(G, coords, Gx, &J);
        for (int i = 0; i < 4; i++) {
            double dx = Gx[0][i], dy = Gx[1][i];
            B[0][2 * i] = dx;  B[0][2 * i + 1] = 0.0;
            B[1][2 * i] = 0.0; B[1][2 * i + 1] = dy;
            B[2][2 * i] = dy;  B[2][2 * i + 1] = dx;
        }
        double pe = E[p], pn = nu[p];
        double factor = pe / ((1.0 + pn) * (1.0 - 2.0 * pn));
        D[0][0] = 1.0 - pn; D[0][1] = pn;       D[0][2] = 0.0;
        D[1][0] = pn;       D[1][1] = 1.0 - pn; D[1][2] = 0.0;
        D[2][0] = 0.0;      D[2][1] = 0.0;      D[2][2] = 0.5 * (1.0 - 2.0 * pn);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) D[i][j] = factor * D[i][j] * w[p] * J;
        for (int i = 0; i < 8; i++)
            for (int j = 0; j < 8; j++)
                for (int k = 0; k < 3; k++)
                    for (int l = 0; l < 3; l++) Ke[8 * i + j] = Ke[8 * i + j] + B[k][i] * D[k][l] * B[l][j];
    }
}

void orc_fe_momentum_quad(const double coords[8], const double fx[4], const double fy[4], double Fe[8])
{
    double xi[4][2], w[4], N[4], G[2][4], Gx[2][4], J;
    orc_gauss_quad4(xi, w);
    for (int p = 0; p < 4; p++) {
        orc_quad_N(xi[p], N);
        orc_quad_dNdxi(xi[p], G);
        orc_quad_dNdx(G, coords, Gx, &J);
        double fac = w[p] * J;
        for (int i = 0; i < 4; i++) {
            Fe[2 * i] += fac * N[i] * fx[p];
            Fe[2 * i + 1] += fac * N[i] * fy[p];
        }
    }
}

/* isotropic 6x6 D of S3:487-495 (unscaled by w*J); shared by Ke and the natural BC */
static void elastic_D(double pe, double pn, double D[6][6], double *factor)
{
    *factor = pe * (1 - pn) / ((1.0 + pn) * (1.0 - 2.0 * pn));
    memset(D, 0, 36 * sizeof(double));
    D[0][0] = 1.0;            D[0][1] = pn / (1 - pn);  D[0][2] = pn / (1 - pn);
    D[1][0] = pn / (1 - pn);  D[1][1] = 1.0;            D[1][2] = pn / (1 - pn);
    D[2][0] = pn / (1 - pn);  D[2][1] = pn / (1 - pn);  D[2][2] = 1.0;
    D[3][3] = 0.5 * (1 - 2 * pn) / (1 - pn);
    D[4][4] = 0.5 * (1 - 2 * pn) / (1 - pn);
    D[5][5] = 0.5 * (1 - 2 * pn) / (1 - pn);
}

void orc_ke_elastic_hex(const double coords[24], const double E[8], const double nu[8], double Ke[576])
{
    double xi[8][3], w[8], G[3][8], Gx[3][8], J, B[6][24], D[6][6], factor;
    orc_gauss_hex8(xi, w);
    for (int p = 0; p < 8; p++) {
        orc_hex_dNdxi(xi[p], G);
        orc_hex_dNdx(G, coords, Gx, &J);
        for (int i = 0; i < 8; i++) {
            double dx = Gx[0][i], dy = Gx[1][i], dz = Gx[2][i];
            B[0][3 * i] = dx;  B[0][3 * i + 1] = 0.0; B[0][3 * i + 2] = 0.0;
            B[1][3 * i] = 0.0; B[1][3 * i + 1] = dy;  B[1][3 * i + 2] = 0.0;
            B[2][3 * i] = 0.0; B[2][3 * i + 1] = 0.0; B[2][3 * i + 2] = dz;
            B[3][3 * i] = dy;  B[3][3 * i + 1] = dx;  B[3][3 * i + 2] = 0.0;
            B[4][3 * i] = 0.0; B[4][3 * i + 1] = dz;  B[4][3 * i + 2] = dy;
            B[5][3 * i] = dz;  B[5][3 * i + 1] = 0.0; B[5][3 * i + 2] = dx;
        }
        elastic_D(E[p], nu[p], D, &factor);
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) D[i][j] = factor * D[i][j] * w[p] * J;
        for (int i = 0; i < 24; i++)
            for (int j = 0; j < 24; j++)
                for (int k = 0; k < 6; k++)
                    for (int l = 0; l < 6; l++) Ke[24 * i + j] = Ke[24 * i + j] + B[k][i] * D[k][l] * B[l][j];
    }
}

void orc_fe_momentum_hex(const double coords[24], const double fx[8], const double fy[8], const double fz[8],
                         double Fe[24])
{
    double xi[8][3], w[8], N[8], G[3][8], Gx[3][8], J;
    orc_gauss_hex8(xi, w);
    for (int p = 0; p < 8; p++) {
        orc_hex_N(xi[p], N);
        orc_hex_dNdxi(xi[p], G);
        orc_hex_dNdx(G, coords, Gx, &J);
        double fac = w[p] * J;
        for (int i = 0; i < 8; i++) {
            Fe[3 * i] += fac * N[i] * fx[p];
            Fe[3 * i + 1] += fac * N[i] * fy[p];
            Fe[3 * i + 2] += fac * N[i] * fz[p];
        }
    }
}

void orc_ge_naturalbc_hex(const double coords[24], const double gx[72], const double gy[72], const double gz[72],
                          const double E_B[24], const double nu_B[24], double Ge[24])
{
    double xi[24][3], w[24], N[8], D[6][6], factor, strain[6], z[6], sig[3][3], nhat[3], sn[3], Jp;
    double dx = fabs(coords[0] - coords[9]);
    double dy = fabs(coords[1] - coords[4]);
    double dz = fabs(coords[2] - coords[14]);
    orc_gauss_hexface24(xi, w);
    for (int p = 0; p < 24; p++) {
        orc_hex_N(xi[p], N);
        elastic_D(E_B[p], nu_B[p], D, &factor);
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) D[i][j] = factor * D[i][j];
        memset(z, 0, sizeof z);
        strain[0] = gx[3 * p]; strain[1] = gy[3 * p + 1]; strain[2] = gz[3 * p + 2];
        strain[3] = gx[3 * p + 1] + gy[3 * p];
        strain[4] = gy[3 * p + 2] + gz[3 * p + 1];
        strain[5] = gx[3 * p + 2] + gz[3 * p];
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) z[i] = z[i] + D[i][j] * strain[j];
        sig[0][0] = z[0]; sig[0][1] = z[3]; sig[0][2] = z[5];
        sig[1][0] = z[3]; sig[1][1] = z[1]; sig[1][2] = z[4];
        sig[2][0] = z[5]; sig[2][1] = z[4]; sig[2][2] = z[2];

        int axis = p / 8;                         /* 0: xi faces, 1: eta faces, 2: zeta faces */
        double side = ((p / 4) % 2 == 0) ? 1.0 : -1.0;
        nhat[0] = nhat[1] = nhat[2] = 0.0;
        nhat[axis] = side;
        if (axis == 0)      Jp = (0.5 * dy) * (0.5 * dz);
        else if (axis == 1) Jp = (0.5 * dx) * (0.5 * dz);
        else                Jp = (0.5 * dx) * (0.5 * dy);
        double fac = w[p] * Jp;
        for (int i = 0; i < 3; i++) sn[i] = sig[i][0] * nhat[0] + sig[i][1] * nhat[1] + sig[i][2] * nhat[2];
        for (int i = 0; i < 8; i++) {
            Ge[3 * i] += fac * N[i] * sn[0];
            Ge[3 * i + 1] += fac * N[i] * sn[1];
            Ge[3 * i + 2] += fac * N[i] * sn[2];
        }
    }
}

/* ---- manufactured solutions -------------------------------------------- */
double orc_poisson_exact(const double pos[3])
{
    return sin(ORC_PI * pos[0]) * sin(ORC_PI * pos[1]) * sin(ORC_PI * pos[2]);
}

void orc_elastic_exact(const double pos[3], double vel[3], double Fm[3], double Gm[3][3], double E, double nu)
{
    double x = pos[0], y = pos[1], z = pos[2];
    if (vel) {
        vel[2] = sin(ORC_PI * x) * sin(ORC_PI * y) * sin(ORC_PI * z);
        vel[0] = 0.0;
        vel[1] = 0.0;
    }
    if (Fm) {
        Fm[2] = (E * ORC_PI * ORC_PI * sin(ORC_PI * x) * sin(ORC_PI * y) * sin(ORC_PI * z) * (3 * nu - 2))
              / (2 * nu * nu + nu - 1);
        Fm[0] = (E * ORC_PI * ORC_PI * cos(ORC_PI * x) * cos(ORC_PI * z) * sin(ORC_PI * y))
              / (2 * (2 * nu * nu + nu - 1));
        Fm[1] = (E * ORC_PI * ORC_PI * cos(ORC_PI * y) * sin(ORC_PI * x) * cos(ORC_PI * z))
              / (2 * (2 * nu * nu + nu - 1));
    }
    if (Gm) {
        Gm[2][0] = ORC_PI * cos(ORC_PI * x) * sin(ORC_PI * y) * sin(ORC_PI * z);
        Gm[2][1] = ORC_PI * sin(ORC_PI * x) * cos(ORC_PI * y) * sin(ORC_PI * z);
        Gm[2][2] = ORC_PI * sin(ORC_PI * x) * sin(ORC_PI * y) * cos(ORC_PI * z);
        Gm[0][0] = Gm[0][1] = Gm[0][2] = 0;
        Gm[1][0] = Gm[1][1] = Gm[1][2] = 0;
    }
}

/* ---- structured mesh ---------------------------------------------------- */
void orc_dmda_uniform_coords(int nsd, int64_t M, int64_t N, int64_t P, double *coords)
{
    /* PETSc DMDASetUniformCoordinates: h = (max-min)/(M-1); x = min + h*i */
    double hx = (1.0 - 0.0) / (double)(M - 1);
    double hy = (1.0 - 0.0) / (double)(N - 1);
    double hz = (nsd == 3) ? (1.0 - 0.0) / (double)(P - 1) : 0.0;
    int64_t c = 0;
    if (nsd == 2) P = 1;
    for (int64_t k = 0; k < P; k++)
        for (int64_t j = 0; j < N; j++)
            for (int64_t i = 0; i < M; i++) {
                coords[c++] = 0.0 + hx * (double)i;
                coords[c++] = 0.0 + hy * (double)j;
                if (nsd == 3) coords[c++] = 0.0 + hz * (double)k;
            }
}

static int64_t span(int64_t i, int64_t M) { return (M == 1) ? 1 : 3 - (i == 0) - (i == M - 1); }

int64_t orc_dmda_nnz(int nsd, int dof, int64_t M, int64_t N, int64_t P)
{
    int64_t sx = 0, sy = 0, sz = 0;
    if (nsd == 2) P = 1;
    for (int64_t i = 0; i < M; i++) sx += span(i, M);
    for (int64_t j = 0; j < N; j++) sy += span(j, N);
    for (int64_t k = 0; k < P; k++) sz += span(k, P);
    return (int64_t)dof * dof * sx * sy * sz;
}

void orc_dmda_pattern(int nsd, int dof, int64_t M, int64_t N, int64_t P, int64_t *rowptr, int32_t *colidx)
{
    int64_t pos = 0, row = 0;
    if (nsd == 2) P = 1;
    rowptr[0] = 0;
    for (int64_t k = 0; k < P; k++)
        for (int64_t j = 0; j < N; j++)
            for (int64_t i = 0; i < M; i++)
                for (int c = 0; c < dof; c++) {
                    for (int64_t kk = (k > 0 ? k - 1 : 0); kk <= (k < P - 1 ? k + 1 : P - 1); kk++)
                        for (int64_t jj = (j > 0 ? j - 1 : 0); jj <= (j < N - 1 ? j + 1 : N - 1); jj++)
                            for (int64_t ii = (i > 0 ? i - 1 : 0); ii <= (i < M - 1 ? i + 1 : M - 1); ii++)
                                for (int cc = 0; cc < dof; cc++)
                                    colidx[pos++] = (int32_t)(dof * (ii + M * (jj + N * kk)) + cc);
                    rowptr[++row] = pos;
                }
}

/* ---- GaussPointCoefficients producers ----------------------------------- */
static void hex_coords(const double *coords, int64_t M, int64_t N, int64_t i, int64_t j, int64_t k, double el[24])
{
    for (int n = 0; n < 8; n++) {
        int64_t node = (i + HEX_DI[n]) + M * ((j + HEX_DJ[n]) + N * (k + HEX_DK[n]));
        el[3 * n] = coords[3 * node];
        el[3 * n + 1] = coords[3 * node + 1];
        el[3 * n + 2] = coords[3 * node + 2];
    }
}

static void quad_coords(const double *coords, int64_t M, int64_t i, int64_t j, double el[8])
{
    for (int n = 0; n < 4; n++) {
        int64_t node = (i + QUAD_DI[n]) + M * (j + QUAD_DJ[n]);
        el[2 * n] = coords[2 * node];
        el[2 * n + 1] = coords[2 * node + 1];
    }
}

void orc_props_e2_fill(int64_t mx, int64_t my, const double *coords, double E, double nu, double *props)
{
    double xi[4][2], w[4], N[4], el[8];
    orc_gauss_quad4(xi, w);
    for (int64_t j = 0; j < my; j++)
        for (int64_t i = 0; i < mx; i++) {
            double *q = props + 24 * (i + mx * j);
            quad_coords(coords, mx + 1, i, j, el);
            for (int p = 0; p < 4; p++) {
                orc_quad_N(xi[p], N);
                double gx = 0.0, gy = 0.0;
                for (int n = 0; n < 4; n++) {
                    gx = gx + N[n] * el[2 * n];
                    gy = gy + N[n] * el[2 * n + 1];
                }
                q[2 * p] = gx;
                q[2 * p + 1] = gy;
                q[8 + p] = E;
                q[12 + p] = nu;
                q[16 + p] = 0.0;
                q[20 + p] = 0.0;
            }
        }
}

void orc_props_s3_fill(int64_t mx, int64_t my, int64_t mz, const double *coords, double E, double nu,
                       double *props, int stride)
{
    double xi[8][3], w[8], xib[24][3], wb[24], N[8], el[24];
    orc_gauss_hex8(xi, w);
    orc_gauss_hexface24(xib, wb);
    for (int64_t k = 0; k < mz; k++)
        for (int64_t j = 0; j < my; j++)
            for (int64_t i = 0; i < mx; i++) {
                double *q = props + (int64_t)stride * (i + mx * (j + my * k));
                double *gp = q, *gpb = q + 24, *pE = q + 96, *pnu = q + 104, *pEB = q + 112, *pnuB = q + 136;
                double *fx = q + 160, *fy = q + 168, *fz = q + 176, *gx = q + 184, *gy = q + 256, *gz = q + 328;
                for (int t = 400; t < stride; t++) q[t] = 0.0; /* R3: enrich_type/n_enrich = 0 */
                hex_coords(coords, mx + 1, my + 1, i, j, k, el);
                for (int p = 0; p < 8; p++) {
                    orc_hex_N(xi[p], N);
                    double x = 0.0, y = 0.0, z = 0.0;
                    for (int n = 0; n < 8; n++) {
                        x = x + N[n] * el[3 * n];
                        y = y + N[n] * el[3 * n + 1];
                        z = z + N[n] * el[3 * n + 2];
                    }
                    gp[3 * p] = x; gp[3 * p + 1] = y; gp[3 * p + 2] = z;
                }
                for (int p = 0; p < 24; p++) {
                    orc_hex_N(xib[p], N);
                    double x = 0.0, y = 0.0, z = 0.0;
                    for (int n = 0; n < 8; n++) {
                        x = x + N[n] * el[3 * n];
                        y = y + N[n] * el[3 * n + 1];
                        z = z + N[n] * el[3 * n + 2];
                    }
                    gpb[3 * p] = x; gpb[3 * p + 1] = y; gpb[3 * p + 2] = z;
                }
                for (int p = 0; p < 8; p++) {
                    double Fm[3];
                    orc_elastic_exact(gp + 3 * p, NULL, Fm, NULL, E, nu);
                    pE[p] = E; pnu[p] = nu;
                    fx[p] = Fm[0]; fy[p] = Fm[1]; fz[p] = Fm[2];
                }
                for (int p = 0; p < 24; p++) {
                    double Gm[3][3];
                    orc_elastic_exact(gpb + 3 * p, NULL, NULL, Gm, E, nu);
                    pEB[p] = E; pnuB[p] = nu;
                    for (int c = 0; c < 3; c++) {
                        gx[3 * p + c] = Gm[0][c];
                        gy[3 * p + c] = Gm[1][c];
                        gz[3 * p + c] = Gm[2][c];
                    }
                }
            }
}

/* ---- MatSetValues(ADD_VALUES) into a sorted CSR row --------------------- */
static int csr_add(const int64_t *rowptr, const int32_t *colidx, double *vals, int64_t row, int64_t col, double v)
{
    int64_t lo = rowptr[row], hi = rowptr[row + 1] - 1;
    while (lo <= hi) {
        int64_t mid = (lo + hi) / 2;
        if (colidx[mid] == col) { vals[mid] += v; return 0; }
        if (colidx[mid] < col) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

int orc_assemble_poisson3d(int64_t M, int64_t N, int64_t P, const double *coords, const int64_t *rowptr,
                           const int32_t *colidx, double *vals, double *b, int64_t e0, int64_t e1)
{
    double el[24], Ae[64], Fe[8];
    int64_t eq[8];
    int bnd[8];
    for (int64_t ek = e0; ek < e1; ek++)
        for (int64_t ej = 0; ej < N - 1; ej++)
            for (int64_t ei = 0; ei < M - 1; ei++) {
                hex_coords(coords, M, N, ei, ej, ek, el);
                for (int n = 0; n < 8; n++) {
                    int64_t i = ei + HEX_DI[n], j = ej + HEX_DJ[n], k = ek + HEX_DK[n];
                    eq[n] = i + M * (j + N * k);
                    bnd[n] = (i == 0 || i == M - 1 || j == 0 || j == N - 1 || k == 0 || k == P - 1);
                }
                if (vals) {
                    memset(Ae, 0, sizeof Ae);
                    orc_ke_poisson_hex(el, Ae);
                    for (int n = 0; n < 8; n++)
                        if (bnd[n]) {                       /* _ZERO_ROW_i, P3:286-290 */
                            for (int q = 0; q < 8; q++) Ae[8 * n + q] = 0.0;
                            Ae[8 * n + n] = 1.0;
                        }
                    for (int r = 0; r < 8; r++)
                        for (int c = 0; c < 8; c++)
                            if (csr_add(rowptr, colidx, vals, eq[r], eq[c], Ae[8 * r + c])) return -1;
                }
                if (b) {
                    memset(Fe, 0, sizeof Fe);
                    orc_fe_poisson_hex(el, Fe);
                    for (int n = 0; n < 8; n++)
                        if (bnd[n]) Fe[n] = 0.0;
                    for (int n = 0; n < 8; n++) b[eq[n]] += Fe[n];
                }
            }
    return 0;
}

int orc_assemble_elastic2d(int64_t M, int64_t N, const double *coords, const double *props, int stride,
                           const int64_t *rowptr, const int32_t *colidx, double *vals, double *b,
                           int64_t e0, int64_t e1)
{
    double el[8], Ae[64], Fe[8];
    int64_t eq[8];
    (void)N;
    for (int64_t ej = e0; ej < e1; ej++)
        for (int64_t ei = 0; ei < M - 1; ei++) {
            const double *q = props + (int64_t)stride * (ei + (M - 1) * ej);
            quad_coords(coords, M, ei, ej, el);
            for (int n = 0; n < 4; n++)
                for (int c = 0; c < 2; c++) eq[2 * n + c] = 2 * ((ei + QUAD_DI[n]) + M * (ej + QUAD_DJ[n])) + c;
            if (vals) {
                memset(Ae, 0, sizeof Ae);
                orc_ke_planestrain_quad(el, q + 8, q + 12, Ae);
                for (int r = 0; r < 8; r++)
                    for (int c = 0; c < 8; c++)
                        if (csr_add(rowptr, colidx, vals, eq[r], eq[c], Ae[8 * r + c])) return -1;
            }
            if (b) {
                memset(Fe, 0, sizeof Fe);
                orc_fe_momentum_quad(el, q + 16, q + 20, Fe);
                for (int n = 0; n < 4; n++) {
                    b[eq[2 * n]] = b[eq[2 * n]] + Fe[2 * n];
                    b[eq[2 * n + 1]] = b[eq[2 * n + 1]] + Fe[2 * n + 1];
                }
            }
        }
    return 0;
}

int orc_assemble_elastic3d(int64_t M, int64_t N, int64_t P, const double *coords, const double *props, int stride,
                           const int64_t *rowptr, const int32_t *colidx, double *vals, double *b,
                           int64_t e0, int64_t e1)
{
    double el[24], Fe[24], Ge[24];
    double *Ae = (double *)malloc(576 * sizeof(double));
    int64_t eq[24];
    (void)P;
    for (int64_t ek = e0; ek < e1; ek++)
        for (int64_t ej = 0; ej < N - 1; ej++)
            for (int64_t ei = 0; ei < M - 1; ei++) {
                const double *q = props + (int64_t)stride * (ei + (M - 1) * (ej + (N - 1) * ek));
                hex_coords(coords, M, N, ei, ej, ek, el);
                for (int n = 0; n < 8; n++)
                    for (int c = 0; c < 3; c++)
                        eq[3 * n + c] = 3 * ((ei + HEX_DI[n]) + M * ((ej + HEX_DJ[n]) + N * (ek + HEX_DK[n]))) + c;
                if (vals) {
                    memset(Ae, 0, 576 * sizeof(double));
                    orc_ke_elastic_hex(el, q + 96, q + 104, Ae);
                    for (int r = 0; r < 24; r++)
                        for (int c = 0; c < 24; c++)
                            if (csr_add(rowptr, colidx, vals, eq[r], eq[c], Ae[24 * r + c])) { free(Ae); return -1; }
                }
                if (b) {
                    memset(Fe, 0, sizeof Fe);
                    memset(Ge, 0, sizeof Ge);
                    orc_fe_momentum_hex(el, q + 160, q + 168, q + 176, Fe);
                    orc_ge_naturalbc_hex(el, q + 184, q + 256, q + 328, q + 112, q + 136, Ge);
                    for (int t = 0; t < 24; t++) b[eq[t]] = b[eq[t]] + Fe[t];
                    for (int t = 0; t < 24; t++) b[eq[t]] = b[eq[t]] + Ge[t];
                }
            }
    free(Ae);
    return 0;
}

/* ---- Dirichlet BCs ------------------------------------------------------- */
static void zero_row_unit_diag(const int64_t *rowptr, const int32_t *colidx, double *vals, int64_t row)
{
    for (int64_t q = rowptr[row]; q < rowptr[row + 1]; q++) vals[q] = (colidx[q] == row) ? 1.0 : 0.0;
}

void orc_bc_e2_compression(int64_t M, int64_t N, const int64_t *rowptr, const int32_t *colidx, double *vals,
                           double *b)
{
    /* E2:882-885: EAST(i=M-1) ux=-1, uy=0 ; WEST(i=0) ux=+1, uy=0 */
    const double val[2][2] = {{-1.0, 0.0}, {1.0, 0.0}};
    for (int side = 0; side < 2; side++)
        for (int d = 0; d < 2; d++)
            for (int64_t j = 0; j < N; j++) {
                int64_t i = (side == 0) ? M - 1 : 0;
                int64_t row = 2 * (i + M * j) + d;
                if (b) b[row] = val[side][d];
                if (vals) zero_row_unit_diag(rowptr, colidx, vals, row);
            }
}

void orc_bc_flags_e2(int64_t M, int64_t N, uint8_t *flags)
{
    for (int64_t j = 0; j < N; j++)
        for (int64_t i = 0; i < M; i++)
            for (int d = 0; d < 2; d++) flags[2 * (i + M * j) + d] = (i == 0 || i == M - 1);
}

void orc_bc_s3_allfaces(int64_t M, int64_t N, int64_t P, const double *coords, const int64_t *rowptr,
                        const int32_t *colidx, double *vals, double *b)
{
    /* S3:1917-1939: FRONT(j=0) BACK(j=N-1) LEFT(i=0) RIGHT(i=M-1) UP(k=P-1) DOWN(k=0), dofs 0..2,
     * value = exact displacement at the node (bc_val == NULL, S3:1501-1505). */
    for (int64_t k = 0; k < P; k++)
        for (int64_t j = 0; j < N; j++)
            for (int64_t i = 0; i < M; i++) {
                if (!(i == 0 || i == M - 1 || j == 0 || j == N - 1 || k == 0 || k == P - 1)) continue;
                int64_t node = i + M * (j + N * k);
                double vel[3];
                orc_elastic_exact(coords + 3 * node, vel, NULL, NULL, 0, 0);
                for (int d = 0; d < 3; d++) {
                    if (b) b[3 * node + d] = vel[d];
                    if (vals) zero_row_unit_diag(rowptr, colidx, vals, 3 * node + d);
                }
            }
}

void orc_bc_flags_s3(int64_t M, int64_t N, int64_t P, uint8_t *flags)
{
    for (int64_t k = 0; k < P; k++)
        for (int64_t j = 0; j < N; j++)
            for (int64_t i = 0; i < M; i++) {
                uint8_t f = (i == 0 || i == M - 1 || j == 0 || j == N - 1 || k == 0 || k == P - 1);
                for (int d = 0; d < 3; d++) flags[3 * (i + M * (j + N * k)) + d] = f;
            }
}

int64_t orc_compress_zeroed_rows(int64_t nrows, const uint8_t *flags, const int64_t *rowptr, const int32_t *colidx,
                                 const double *vals, int64_t *rowptr_out, int32_t *colidx_out, double *vals_out)
{
    int64_t pos = 0;
    rowptr_out[0] = 0;
    for (int64_t r = 0; r < nrows; r++) {
        if (flags[r]) {
            colidx_out[pos] = (int32_t)r;
            vals_out[pos] = 1.0;
            pos++;
        } else {
            for (int64_t q = rowptr[r]; q < rowptr[r + 1]; q++) {
                colidx_out[pos] = colidx[q];
                vals_out[pos] = vals[q];
                pos++;
            }
        }
        rowptr_out[r + 1] = pos;
    }
    return pos;
}

/* ---- error norms ---------------------------------------------------------- */
void orc_errors_poisson3d(int64_t M, int64_t N, int64_t P, const double *coords, const double *x, double *h,
                          double *L2, double *H1)
{
    double xi[8][3], w[8], Nn[8], G[3][8], Gx[3][8], J, el[24], fe[8], fa[8];
    double tL2 = 0.0, tH1 = 0.0;
    orc_gauss_hex8(xi, w);
    for (int64_t ek = 0; ek < P - 1; ek++)
        for (int64_t ej = 0; ej < N - 1; ej++)
            for (int64_t ei = 0; ei < M - 1; ei++) {
                hex_coords(coords, M, N, ei, ej, ek, el);
                for (int n = 0; n < 8; n++) {
                    int64_t node = (ei + HEX_DI[n]) + M * ((ej + HEX_DJ[n]) + N * (ek + HEX_DK[n]));
                    fe[n] = x[node];
                    fa[n] = orc_poisson_exact(coords + 3 * node);
                }
                double eL2 = 0.0, eH1 = 0.0;
                for (int p = 0; p < 8; p++) {
                    orc_hex_N(xi[p], Nn);
                    orc_hex_dNdxi(xi[p], G);
                    orc_hex_dNdx(G, el, Gx, &J);
                    double fac = w[p] * J;
                    for (int i = 0; i < 8; i++) {
                        double e = fe[i] - fa[i];
                        eL2 += fac * Nn[i] * e * e;
                        eH1 += fac * (Gx[0][i] * e * Gx[0][i] * e + Gx[1][i] * e * Gx[1][i] * e
                                      + Gx[2][i] * e * Gx[2][i] * e);
                    }
                }
                tL2 += eL2;
                tH1 += eH1;
            }
    *h = (1.0 - 0.0) / (double)(M - 1);
    *L2 = sqrt(tL2);
    *H1 = sqrt(tH1);
}

void orc_errors_elastic3d(int64_t M, int64_t N, int64_t P, const double *coords, const double *x, double *h,
                          double *L2, double *H1)
{
    double xi[8][3], w[8], Nn[8], G[3][8], Gx[3][8], J, el[24], ue[8][3], ua[8][3];
    double tL2 = 0.0, tH1 = 0.0;
    orc_gauss_hex8(xi, w);
    for (int64_t ek = 0; ek < P - 1; ek++)
        for (int64_t ej = 0; ej < N - 1; ej++)
            for (int64_t ei = 0; ei < M - 1; ei++) {
                hex_coords(coords, M, N, ei, ej, ek, el);
                for (int n = 0; n < 8; n++) {
                    int64_t node = (ei + HEX_DI[n]) + M * ((ej + HEX_DJ[n]) + N * (ek + HEX_DK[n]));
                    for (int c = 0; c < 3; c++) ue[n][c] = x[3 * node + c];
                    orc_elastic_exact(coords + 3 * node, ua[n], NULL, NULL, 0, 0);
                }
                double eL2 = 0.0, eH1 = 0.0;
                for (int p = 0; p < 8; p++) {
                    orc_hex_N(xi[p], Nn);
                    orc_hex_dNdxi(xi[p], G);
                    orc_hex_dNdx(G, el, Gx, &J);
                    double fac = w[p] * J;
                    for (int i = 0; i < 8; i++) {
                        double u = ue[i][0] - ua[i][0], v = ue[i][1] - ua[i][1], ww = ue[i][2] - ua[i][2];
                        eL2 += fac * Nn[i] * (u * u + v * v + ww * ww);
                        eH1 = eH1 + fac * (Gx[0][i] * u * Gx[0][i] * u + Gx[1][i] * u * Gx[1][i] * u
                                           + Gx[2][i] * u * Gx[2][i] * u + Gx[0][i] * v * Gx[0][i] * v
                                           + Gx[1][i] * v * Gx[1][i] * v + Gx[2][i] * v * Gx[2][i] * v
                                           + Gx[0][i] * ww * Gx[0][i] * ww + Gx[1][i] * ww * Gx[1][i] * ww
                                           + Gx[2][i] * ww * Gx[2][i] * ww);
                    }
                }
                tL2 += eL2;
                tH1 += eH1;
            }
    *h = (1.0 - 0.0) / (double)(M - 1);
    *L2 = sqrt(tL2);
    *H1 = sqrt(tH1);
}

/* ==========================================================================================================
 * Config 5: SGFEM-enriched Q1 hexahedra.  NO REFERENCE ARITHMETIC EXISTS for this: R3
 * ("SGFEM_real") holds only bookkeeping -- the macros Q, L1, L2, type, test_dof (R3:39-43), the per-node fields
 * enrich_type[8] / n_enrich[8] of the element record (R3:46-49, 65-67), their zero fill (R3:1265-1269) and the
 * dead, non-compiling BlockRedundantDofs / GetNodalRedundantDofs (R3:1467-1595).  The specification below is
 * owned by this build and shaped by that scaffolding; it is a restatement of the SPEC,
 * written independently of the HIP kernels (naive B^T D B over the full B matrix, as S3:450-518 does).
 *
 *   scalar shape functions of an element:  Phi_{a,0} = N_a,   Phi_{a,1+m} = N_a * phi_m   (m < n_enr),
 *     phi_m = psi_m - sum_b N_b psi_m(x_b)   (`type` 1, R3:42: SGFEM, enrichment minus its Q1 interpolant)
 *     phi_m = psi_m                           (`type` 0: GFEM)
 *     family m is active at local node a iff enrich_type[a] != 0 and m < n_enrich[a]; else Phi_{a,1+m} = 0 here
 *   node-blocked dofs, D = 3 (1 + n_enr) per node, q = 3 s + c  (ElasticityDOF R3:70-74 repeated per family)
 *   record = R3's GaussPointCoefficients (R3:52-68; `base` = 408 doubles with 32-bit PetscInt, 416 with 64-bit)
 *     followed by 64 doubles per family:  psi at the 8 nodes, psi at the 8 Gauss points, grad psi (physical) at the
 *     8 Gauss points [p][3], psi at the 24 face Gauss points
 *   Ke, Fe, Ge: the S3 formulas (S3:450-656) with the 8 S scalar functions in place of the 8 N_a
 *   rows:  all-face Dirichlet as S3 on the displacement dofs (exact value) and 0 on the enriched dofs of boundary
 *     nodes; every enriched dof whose family is inactive at its node is blocked: unit row, b = 0
 *     (BlockRedundantDofs R3:1509-1595: MatZeroRows(A, ..., 1.0) on the redundant dofs).  A node takes its flags from
 *     the element behind it, (max(i-1,0), max(j-1,0), max(k-1,0)), the one element of a node every slab holds.
 * ========================================================================================================== */
static int64_t enr_flag(const double *rec, int base, int idx)
{
    /* PetscInt enrich_type[8], n_enrich[8] stored inside a struct of PetscScalar, at double 400 (R3:65-67) */
    if (base == 408) return (int64_t)((const int32_t *)(rec + 400))[idx];
    return ((const int64_t *)(rec + 400))[idx];
}

/* gradients [fn][3] and values of the 8 S scalar functions at volume Gauss point p; fn = 8 s + a */
static void enr_functions_gp(const double coords[24], const double *rec, int base, int n_enr, int sgfem, int p,
                             double (*g)[3], double *val, double *detJ)
{
    double xi[8][3], w[8], N[8], G[3][8], Gx[3][8];
    orc_gauss_hex8(xi, w);
    orc_hex_N(xi[p], N);
    orc_hex_dNdxi(xi[p], G);
    orc_hex_dNdx(G, coords, Gx, detJ);
    for (int a = 0; a < 8; a++) {
        val[a] = N[a];
        for (int k = 0; k < 3; k++) g[a][k] = Gx[k][a];
    }
    for (int m = 0; m < n_enr; m++) {
        const double *e = rec + base + 64 * m, *psi_n = e, *psi = e + 8, *dpsi = e + 16;
        double phi = psi[p], dphi[3] = {dpsi[3 * p], dpsi[3 * p + 1], dpsi[3 * p + 2]};
        if (sgfem) {
            for (int b = 0; b < 8; b++) {
                phi = phi - N[b] * psi_n[b];
                for (int k = 0; k < 3; k++) dphi[k] = dphi[k] - Gx[k][b] * psi_n[b];
            }
        }
        for (int a = 0; a < 8; a++) {
            int active = enr_flag(rec, base, a) != 0 && m < enr_flag(rec, base, 8 + a);
            int fn = 8 * (1 + m) + a;
            val[fn] = active ? N[a] * phi : 0.0;
            for (int k = 0; k < 3; k++) g[fn][k] = active ? phi * Gx[k][a] + N[a] * dphi[k] : 0.0;
        }
    }
}

void orc_enr_ke(const double coords[24], const double *rec, int base, int n_enr, int sgfem, double *Ke)
{
    const int S = 1 + n_enr, D = 3 * S, NR = 8 * D;
    double g[8 * 8][3], val[8 * 8], J, D6[6][6], factor;
    double *B = (double *)malloc((size_t)6 * NR * sizeof(double));
    for (int p = 0; p < 8; p++) {
        enr_functions_gp(coords, rec, base, n_enr, sgfem, p, g, val, &J);
        memset(B, 0, (size_t)6 * NR * sizeof(double));
        for (int a = 0; a < 8; a++)
            for (int s = 0; s < S; s++) {
                const double dx = g[8 * s + a][0], dy = g[8 * s + a][1], dz = g[8 * s + a][2];
                const int c0 = a * D + 3 * s;                /* columns of (a, s, c = 0..2): S3:470-482 */
                B[0 * NR + c0] = dx; B[1 * NR + c0 + 1] = dy; B[2 * NR + c0 + 2] = dz;
                B[3 * NR + c0] = dy; B[3 * NR + c0 + 1] = dx;
                B[4 * NR + c0 + 1] = dz; B[4 * NR + c0 + 2] = dy;
                B[5 * NR + c0] = dz; B[5 * NR + c0 + 2] = dx;
            }
        elastic_D(rec[96 + p], rec[104 + p], D6, &factor);
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) D6[i][j] = factor * D6[i][j] * 1.0 * J;
        for (int i = 0; i < NR; i++)
            for (int j = 0; j < NR; j++)
                for (int k = 0; k < 6; k++)
                    for (int l = 0; l < 6; l++) Ke[NR * i + j] = Ke[NR * i + j] + B[k * NR + i] * D6[k][l] * B[l * NR + j];
    }
    free(B);
}

/* Fe + Ge of the enriched element (S3:520-656 with Phi_{a,s} in place of N_a), added into F[8 D] */
void orc_enr_fe(const double coords[24], const double *rec, int base, int n_enr, int sgfem, double *F)
{
    const int S = 1 + n_enr, D = 3 * S;
    double g[8 * 8][3], val[8 * 8], J;
    const double *fx = rec + 160, *fy = rec + 168, *fz = rec + 176;
    for (int p = 0; p < 8; p++) {
        enr_functions_gp(coords, rec, base, n_enr, sgfem, p, g, val, &J);
        double fac = 1.0 * J;
        for (int a = 0; a < 8; a++)
            for (int s = 0; s < S; s++) {
                F[a * D + 3 * s] += fac * val[8 * s + a] * fx[p];
                F[a * D + 3 * s + 1] += fac * val[8 * s + a] * fy[p];
                F[a * D + 3 * s + 2] += fac * val[8 * s + a] * fz[p];
            }
    }
    /* natural-BC term over the 24 face points (S3:548-656) */
    const double *gx = rec + 184, *gy = rec + 256, *gz = rec + 328, *E_B = rec + 112, *nu_B = rec + 136;
    double xi[24][3], w[24], N[8], D6[6][6], factor, strain[6], z[6], sn[3], Jp;
    double dx = fabs(coords[0] - coords[9]), dy = fabs(coords[1] - coords[4]), dz = fabs(coords[2] - coords[14]);
    orc_gauss_hexface24(xi, w);
    for (int p = 0; p < 24; p++) {
        orc_hex_N(xi[p], N);
        elastic_D(E_B[p], nu_B[p], D6, &factor);
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) D6[i][j] = factor * D6[i][j];
        memset(z, 0, sizeof z);
        strain[0] = gx[3 * p]; strain[1] = gy[3 * p + 1]; strain[2] = gz[3 * p + 2];
        strain[3] = gx[3 * p + 1] + gy[3 * p];
        strain[4] = gy[3 * p + 2] + gz[3 * p + 1];
        strain[5] = gx[3 * p + 2] + gz[3 * p];
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) z[i] = z[i] + D6[i][j] * strain[j];
        int axis = p / 8;
        double side = ((p / 4) % 2 == 0) ? 1.0 : -1.0;
        if (axis == 0)      { sn[0] = z[0] * side; sn[1] = z[3] * side; sn[2] = z[5] * side; Jp = (0.5 * dy) * (0.5 * dz); }
        else if (axis == 1) { sn[0] = z[3] * side; sn[1] = z[1] * side; sn[2] = z[4] * side; Jp = (0.5 * dx) * (0.5 * dz); }
        else                { sn[0] = z[5] * side; sn[1] = z[4] * side; sn[2] = z[2] * side; Jp = (0.5 * dx) * (0.5 * dy); }
        double fac = w[p] * Jp;
        for (int s = 0; s < S; s++) {
            double phi = 1.0;
            if (s > 0) {
                const double *e = rec + base + 64 * (s - 1), *psi_n = e, *psi_b = e + 40;
                phi = psi_b[p];
                if (sgfem)
                    for (int b = 0; b < 8; b++) phi = phi - N[b] * psi_n[b];
            }
            for (int a = 0; a < 8; a++) {
                int active = (s == 0) || (enr_flag(rec, base, a) != 0 && (s - 1) < enr_flag(rec, base, 8 + a));
                double v = active ? N[a] * phi : 0.0;
                for (int c = 0; c < 3; c++) F[a * D + 3 * s + c] += fac * v * sn[c];
            }
        }
    }
}

int orc_assemble_enriched3d(int64_t M, int64_t N, int64_t P, const double *coords, const double *props, int stride,
                            int base, int n_enr, int sgfem, const int64_t *rowptr, const int32_t *colidx,
                            double *vals, double *b, int64_t e0, int64_t e1)
{
    const int D = 3 * (1 + n_enr), NR = 8 * D;
    double el[24];
    double *Ae = (double *)malloc((size_t)NR * NR * sizeof(double)), *Fe = (double *)malloc((size_t)NR * sizeof(double));
    int64_t *eq = (int64_t *)malloc((size_t)NR * sizeof(int64_t));
    int rc = 0;
    (void)P;
    for (int64_t ek = e0; ek < e1 && !rc; ek++)
        for (int64_t ej = 0; ej < N - 1 && !rc; ej++)
            for (int64_t ei = 0; ei < M - 1 && !rc; ei++) {
                const double *q = props + (int64_t)stride * (ei + (M - 1) * (ej + (N - 1) * ek));
                hex_coords(coords, M, N, ei, ej, ek, el);
                for (int n = 0; n < 8; n++)
                    for (int c = 0; c < D; c++)
                        eq[D * n + c] = D * ((ei + HEX_DI[n]) + M * ((ej + HEX_DJ[n]) + N * (ek + HEX_DK[n]))) + c;
                if (vals) {
                    memset(Ae, 0, (size_t)NR * NR * sizeof(double));
                    orc_enr_ke(el, q, base, n_enr, sgfem, Ae);
                    for (int r = 0; r < NR && !rc; r++)
                        for (int c = 0; c < NR; c++)
                            if (csr_add(rowptr, colidx, vals, eq[r], eq[c], Ae[NR * r + c])) { rc = -1; break; }
                }
                if (b) {
                    memset(Fe, 0, (size_t)NR * sizeof(double));
                    orc_enr_fe(el, q, base, n_enr, sgfem, Fe);
                    for (int t = 0; t < NR; t++) b[eq[t]] = b[eq[t]] + Fe[t];
                }
            }
    free(Ae); free(Fe); free(eq);
    return rc;
}

/* rows: S3's all-face Dirichlet (if dirichlet != 0) and the blocked redundant dofs (always) */
void orc_bc_enriched3d(int64_t M, int64_t N, int64_t P, const double *coords, const double *props, int stride,
                       int base, int n_enr, int dirichlet, const int64_t *rowptr, const int32_t *colidx,
                       double *vals, double *b)
{
    const int D = 3 * (1 + n_enr);
    for (int64_t k = 0; k < P; k++)
        for (int64_t j = 0; j < N; j++)
            for (int64_t i = 0; i < M; i++) {
                const int64_t node = i + M * (j + N * k);
                const int bnd = (i == 0 || i == M - 1 || j == 0 || j == N - 1 || k == 0 || k == P - 1);
                /* the element behind the node and the node's place in it */
                const int64_t ei = i > 0 ? i - 1 : 0, ej = j > 0 ? j - 1 : 0, ek = k > 0 ? k - 1 : 0;
                int a = -1;
                for (int n = 0; n < 8; n++)
                    if (HEX_DI[n] == (i > 0) && HEX_DJ[n] == (j > 0) && HEX_DK[n] == (k > 0)) a = n;
                const double *q = props + (int64_t)stride * (ei + (M - 1) * (ej + (N - 1) * ek));
                double vel[3];
                orc_elastic_exact(coords + 3 * node, vel, NULL, NULL, 0, 0);
                for (int d = 0; d < D; d++) {
                    const int m = d / 3 - 1;
                    const int blocked = (d >= 3) && !(enr_flag(q, base, a) != 0 && m < enr_flag(q, base, 8 + a));
                    if (!(blocked || (dirichlet && bnd))) continue;
                    if (b) b[D * node + d] = (d < 3 && dirichlet && bnd) ? vel[d] : 0.0;
                    if (vals) zero_row_unit_diag(rowptr, colidx, vals, D * node + d);
                }
            }
}
