/*
 * oracle/ndt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (C99) of the reference's NDT scan-to-map registration,
 * PCR::NdtRegister::scan2Map -> pclomp::NormalDistributionsTransform (DIRECT7).
 * Follows (reference tree):
 *   PCR/src/NdtRegister.cpp:12-13,21-31                resolution 1.0, DIRECT7, f32 pose in/out
 *   third_parties/pclomp/src/voxel_grid_covariance_omp_impl.hpp:49-370   voxel Gaussians
 *       (leaf index in float, single-pass covariance, eigenvalue inflation 0.01, inverse)
 *   ...:374-433                                         getNeighborhoodAtPoint7 (<= 7 leaves, >= 6 points)
 *   third_parties/pclomp/src/ndt_omp_impl.hpp:81-171    computeTransformation (Newton + line search)
 *   ...:180-285  computeDerivatives   ...:289-395 computeAngleDerivatives
 *   ...:399-440  computePointDerivatives (float)   ...:485-537 updateDerivatives (float inner math)
 *   ...:541-645  computeHessian / updateHessian (double; note h_ang_d1_ differs in sign of its 3rd
 *                component from the float table's row 6 -- both are restated as written)
 *   ...:649-932  More-Thuente line search
 *   pclomp/ndt_omp.h:430-447  auxiliary psi functions
 *
 * PARITY STATUS: "parity unpinned" (no reference vectors; PCL/Eigen absent here).  Restated from
 * published semantics: pcl::transformPointCloud (float R p + t), Eigen AngleAxis/Translation
 * products in float, Matrix3f::eulerAngles(0,1,2), JacobiSVD::solve (one-sided Jacobi here),
 * SelfAdjointEigenSolver (cyclic Jacobi here), Matrix3d::inverse (cofactors).
 * Deviations: Transform<float,3,Affine>::rotation() (an SVD-based polar factor in Eigen) is taken as
 * the linear part; expf for exp(float).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void oracle_sym3_eig(const double A[9], double w[3], double V[9]);   /* vgicp_oracle.c: descending */

typedef struct {
    double resolution;      /* 1.0  NdtRegister.hpp:11 */
    double step_size;       /* 0.1  ndt_omp_impl.hpp:50 */
    double outlier_ratio;   /* 0.55 ndt_omp_impl.hpp:51 */
    double trans_eps;       /* 0.1  ndt_omp_impl.hpp:71 */
    int max_iters;          /* 35   ndt_omp_impl.hpp:72 */
    int min_points;         /* 6    voxel_grid_covariance_omp.h:210 */
    double eig_mult;        /* 0.01 voxel_grid_covariance_omp.h:211 */
    int threads;            /* num_threads_ = `cores` (PCR/src/NdtRegister.cpp:18; ndt_omp_impl.hpp:206): computeDerivatives only */
    int pad_;
} oracle_ndt_params;

void oracle_ndt_default_params(oracle_ndt_params *p)
{
    p->resolution = 1.0; p->step_size = 0.1; p->outlier_ratio = 0.55; p->trans_eps = 0.1; p->max_iters = 35;
    p->min_points = 6; p->eig_mult = 0.01; p->threads = 1; p->pad_ = 0;
}

/* ---- voxel grid ---- */
typedef struct { int n; double mean[3]; double cov[9]; double icov[9]; } leaf_t;
typedef struct {
    int min_b[3], max_b[3], div_b[3];
    float leaf, inv_leaf;
    int min_points;
    leaf_t *leaves;     /* dense div_b[0]*div_b[1]*div_b[2] (the reference: std::map keyed by the same index) */
} ndt_grid;

static void inv3(const double m[9], double out[9])
{
    double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    double det = m[0] * c00 + m[1] * c01 + m[2] * c02, id = 1.0 / det;
    out[0] = c00 * id; out[1] = (m[2] * m[7] - m[1] * m[8]) * id; out[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    out[3] = c01 * id; out[4] = (m[0] * m[8] - m[2] * m[6]) * id; out[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    out[6] = c02 * id; out[7] = (m[1] * m[6] - m[0] * m[7]) * id; out[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

static ndt_grid *ndt_grid_build(const float *pts, size_t n, size_t stride, const oracle_ndt_params *prm)
{
    ndt_grid *g = (ndt_grid *)calloc(1, sizeof *g);
    g->leaf = (float)prm->resolution; g->inv_leaf = 1.0f / g->leaf; g->min_points = prm->min_points;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    size_t nf = 0;
    for (size_t i = 0; i < n; ++i) {
        const float *p = pts + i * stride;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        for (int d = 0; d < 3; ++d) { if (p[d] < mn[d]) mn[d] = p[d]; if (p[d] > mx[d]) mx[d] = p[d]; }
        ++nf;
    }
    if (!nf) { g->div_b[0] = g->div_b[1] = g->div_b[2] = 0; return g; }
    for (int d = 0; d < 3; ++d) {
        g->min_b[d] = (int)floorf(mn[d] * g->inv_leaf); g->max_b[d] = (int)floorf(mx[d] * g->inv_leaf);
        g->div_b[d] = g->max_b[d] - g->min_b[d] + 1;
    }
    size_t nl = (size_t)g->div_b[0] * g->div_b[1] * g->div_b[2];
    g->leaves = (leaf_t *)calloc(nl, sizeof(leaf_t));
    for (size_t i = 0; i < n; ++i) {
        const float *p = pts + i * stride;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        int ijk[3];
        for (int d = 0; d < 3; ++d) ijk[d] = (int)(floorf(p[d] * g->inv_leaf) - (float)g->min_b[d]);   /* :218-220 */
        leaf_t *l = &g->leaves[(size_t)ijk[0] + (size_t)ijk[1] * g->div_b[0] + (size_t)ijk[2] * g->div_b[0] * g->div_b[1]];
        double x[3] = {p[0], p[1], p[2]};
        for (int r = 0; r < 3; ++r) { l->mean[r] += x[r]; for (int c = 0; c < 3; ++c) l->cov[r * 3 + c] += x[r] * x[c]; }
        l->n++;
    }
    for (size_t k = 0; k < nl; ++k) {
        leaf_t *l = &g->leaves[k];
        if (l->n == 0) continue;
        double sum[3] = {l->mean[0], l->mean[1], l->mean[2]};
        for (int r = 0; r < 3; ++r) l->mean[r] /= l->n;
        if (l->n < g->min_points) continue;
        /* :329-330 single-pass covariance */
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
            l->cov[r * 3 + c] = (l->cov[r * 3 + c] - 2 * (sum[r] * l->mean[c])) / l->n + l->mean[r] * l->mean[c];
        for (int e = 0; e < 9; ++e) l->cov[e] *= (l->n - 1.0) / l->n;
        /* SelfAdjointEigenSolver reads the lower triangle; eigenvalues ascending */
        double S[9] = {l->cov[0], l->cov[3], l->cov[6], l->cov[3], l->cov[4], l->cov[7], l->cov[6], l->cov[7], l->cov[8]};
        double w[3], V[9];
        oracle_sym3_eig(S, w, V);                          /* descending: w[0] largest */
        double ev[3] = {w[2], w[1], w[0]};                 /* ascending, as Eigen */
        if (ev[0] < 0 || ev[1] < 0 || ev[2] <= 0) { l->n = -1; continue; }
        double minev = prm->eig_mult * ev[2];
        if (ev[0] < minev) {
            ev[0] = minev;
            if (ev[1] < minev) ev[1] = minev;
            /* cov = evecs * diag * evecs^-1, columns ascending */
            double E[9], Ei[9], D[9] = {ev[0], 0, 0, 0, ev[1], 0, 0, 0, ev[2]}, T[9];
            for (int r = 0; r < 3; ++r) { E[r * 3 + 0] = V[r * 3 + 2]; E[r * 3 + 1] = V[r * 3 + 1]; E[r * 3 + 2] = V[r * 3 + 0]; }
            inv3(E, Ei);
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k2 = 0; k2 < 3; ++k2) s += E[r * 3 + k2] * D[k2 * 3 + c]; T[r * 3 + c] = s; }
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k2 = 0; k2 < 3; ++k2) s += T[r * 3 + k2] * Ei[k2 * 3 + c]; l->cov[r * 3 + c] = s; }
        }
        inv3(l->cov, l->icov);
        double mxc = -DBL_MAX, mnc = DBL_MAX;
        for (int e = 0; e < 9; ++e) { if (l->icov[e] > mxc) mxc = l->icov[e]; if (l->icov[e] < mnc) mnc = l->icov[e]; }
        if (mxc == (double)INFINITY || mnc == -(double)INFINITY) l->n = -1;
    }
    return g;
}
static void ndt_grid_free(ndt_grid *g) { if (g) { free(g->leaves); free(g); } }

/* :374-404,419-433 neighbourhood of <= 7 leaves at a (float) point */
static int neighborhood7(const ndt_grid *g, const float p[3], const leaf_t *out[7])
{
    static const int off[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    if (!g->leaves) return 0;
    int ijk[3], n = 0;
    for (int d = 0; d < 3; ++d) ijk[d] = (int)floorf(p[d] / g->leaf);
    for (int k = 0; k < 7; ++k) {
        int c[3], ok = 1;
        for (int d = 0; d < 3; ++d) { c[d] = ijk[d] + off[k][d]; if (c[d] < g->min_b[d] || c[d] > g->max_b[d]) ok = 0; }
        if (!ok) continue;
        const leaf_t *l = &g->leaves[(size_t)(c[0] - g->min_b[0]) + (size_t)(c[1] - g->min_b[1]) * g->div_b[0] +
                                     (size_t)(c[2] - g->min_b[2]) * g->div_b[0] * g->div_b[1]];
        if (l->n >= g->min_points) out[n++] = l;
    }
    return n;
}

/* query helper for tests */
int oracle_ndt_leaf_at(const float *pts, size_t n, size_t stride, const oracle_ndt_params *prm, const float p[3], double mean[3], double cov[9], double icov[9])
{
    ndt_grid *g = ndt_grid_build(pts, n, stride, prm);
    const leaf_t *nb[7];
    int cnt = 0;
    /* only the centre cell */
    if (g->leaves) {
        int ijk[3], ok = 1;
        for (int d = 0; d < 3; ++d) { ijk[d] = (int)floorf(p[d] / g->leaf); if (ijk[d] < g->min_b[d] || ijk[d] > g->max_b[d]) ok = 0; }
        if (ok) {
            const leaf_t *l = &g->leaves[(size_t)(ijk[0] - g->min_b[0]) + (size_t)(ijk[1] - g->min_b[1]) * g->div_b[0] + (size_t)(ijk[2] - g->min_b[2]) * g->div_b[0] * g->div_b[1]];
            cnt = l->n;
            memcpy(mean, l->mean, sizeof l->mean); memcpy(cov, l->cov, sizeof l->cov); memcpy(icov, l->icov, sizeof l->icov);
        }
    }
    (void)nb;
    ndt_grid_free(g);
    return cnt;
}

/* ---- transform helpers (float, as Eigen/PCL evaluate them) ---- */
static void angle_axis_f(float angle, int axis, float R[9])
{
    /* Eigen::AngleAxisf(angle, UnitX/Y/Z).toRotationMatrix() */
    float ax[3] = {0, 0, 0}; ax[axis] = 1.0f;
    float s = sinf(angle), c = cosf(angle);
    float sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
    float c1[3] = {(1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2]};
    float tmp;
    tmp = c1[0] * ax[1]; R[0 * 3 + 1] = tmp - sa[2]; R[1 * 3 + 0] = tmp + sa[2];
    tmp = c1[0] * ax[2]; R[0 * 3 + 2] = tmp + sa[1]; R[2 * 3 + 0] = tmp - sa[1];
    tmp = c1[1] * ax[2]; R[1 * 3 + 2] = tmp - sa[0]; R[2 * 3 + 1] = tmp + sa[0];
    for (int d = 0; d < 3; ++d) R[d * 3 + d] = c1[d] * ax[d] + c;
}
static void mul33f(const float A[9], const float B[9], float C[9])
{
    float o[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { float s = A[r * 3] * B[c]; s += A[r * 3 + 1] * B[3 + c]; s += A[r * 3 + 2] * B[6 + c]; o[r * 3 + c] = s; }
    memcpy(C, o, sizeof o);
}
/* Translation(x[0:3]) * Rx * Ry * Rz in float -> row-major 3x3 R and t */
static void pose_from_p(const double x[6], float R[9], float t[3])
{
    float Rx[9], Ry[9], Rz[9], T[9];
    angle_axis_f((float)x[3], 0, Rx); angle_axis_f((float)x[4], 1, Ry); angle_axis_f((float)x[5], 2, Rz);
    mul33f(Rx, Ry, T); mul33f(T, Rz, R);
    t[0] = (float)x[0]; t[1] = (float)x[1]; t[2] = (float)x[2];
}
static void transform_cloud_f(const float *src, size_t n, size_t stride, const float R[9], const float t[3], float *out /* n*3 */)
{
    for (size_t i = 0; i < n; ++i) {
        const float *p = src + i * stride;
        for (int r = 0; r < 3; ++r) { float v = R[r * 3] * p[0]; v += R[r * 3 + 1] * p[1]; v += R[r * 3 + 2] * p[2]; v += t[r]; out[i * 3 + r] = v; }
    }
}

/* ---- angle derivative tables ---- */
typedef struct {
    float j_ang[8][3];      /* float table rows (ndt_omp_impl.hpp:338-346) */
    float h_ang[15][3];     /* float table rows a2..f3 (:374-397); row 6 has +sy */
    double jd[8][3];        /* double vectors j_ang_a_.. (:328-335) */
    double hd[15][3];       /* double vectors h_ang_a2_.. (:351-372); d1 has -sy */
} ang_t;

static void angle_derivatives(const double p[6], ang_t *a)
{
    double cx, cy, cz, sx, sy, sz;
    if (fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = cos(p[3]); sx = sin(p[3]); }
    if (fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = cos(p[4]); sy = sin(p[4]); }
    if (fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = cos(p[5]); sz = sin(p[5]); }
    const double J[8][3] = {
        {(-sx * sz + cx * sy * cz), (-sx * cz - cx * sy * sz), (-cx * cy)},
        {(cx * sz + sx * sy * cz), (cx * cz - sx * sy * sz), (-sx * cy)},
        {(-sy * cz), sy * sz, cy},
        {sx * cy * cz, (-sx * cy * sz), sx * sy},
        {(-cx * cy * cz), cx * cy * sz, (-cx * sy)},
        {(-cy * sz), (-cy * cz), 0},
        {(cx * cz - sx * sy * sz), (-cx * sz - sx * sy * cz), 0},
        {(sx * cz + cx * sy * sz), (cx * sy * cz - sx * sz), 0}};
    const double Hh[15][3] = {
        {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), sx * cy},
        {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), (-cx * cy)},
        {(cx * cy * cz), (-cx * cy * sz), (cx * sy)},
        {(sx * cy * cz), (-sx * cy * sz), (sx * sy)},
        {(-sx * cz - cx * sy * sz), (sx * sz - cx * sy * cz), 0},
        {(cx * cz - sx * sy * sz), (-sx * sy * cz - cx * sz), 0},
        {(-cy * cz), (cy * sz), (-sy)},
        {(-sx * sy * cz), (sx * sy * sz), (sx * cy)},
        {(cx * sy * cz), (-cx * sy * sz), (-cx * cy)},
        {(sy * sz), (sy * cz), 0},
        {(-sx * cy * sz), (-sx * cy * cz), 0},
        {(cx * cy * sz), (cx * cy * cz), 0},
        {(-cy * cz), (cy * sz), 0},
        {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), 0},
        {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), 0}};
    for (int r = 0; r < 8; ++r) for (int c = 0; c < 3; ++c) { a->jd[r][c] = J[r][c]; a->j_ang[r][c] = (float)J[r][c]; }
    for (int r = 0; r < 15; ++r) for (int c = 0; c < 3; ++c) { a->hd[r][c] = Hh[r][c]; a->h_ang[r][c] = (float)Hh[r][c]; }
    a->h_ang[6][2] = (float)(sy);      /* the float table writes (sy) where the double vector has (-sy) */
}

typedef struct { double d1, d2, d3; } gauss_t;
static gauss_t gauss_consts(const oracle_ndt_params *prm)
{
    gauss_t g;
    float res = (float)prm->resolution;
    double c1 = 10 * (1 - prm->outlier_ratio), c2 = prm->outlier_ratio / pow((double)res, 3);
    g.d3 = -log(c2);
    g.d1 = -log(c1 + c2) - g.d3;
    g.d2 = -2 * log((-log(c1 * exp(-0.5) + c2) - g.d3) / g.d1);
    return g;
}

/* computeDerivatives (:180-285) with the float inner math of updateDerivatives (:485-537).
 * The reference runs the loop over the points on num_threads_ OpenMP threads (schedule(guided, 8), :206), every point writing
 * its own score / gradient / Hessian into scores[idx], score_gradients[idx], hessians[idx] (:270-272), and then adds those up
 * SERIALLY in point order (:277-282, "Ensure that the result is invariant against the summing up order").  So do we: the
 * result is bit-identical for every thread count, which tests/test_oracle_vgicp_ndt.py pins. */
static void derivatives_point(const ndt_grid *g, const float *xp, const float *tp, const gauss_t *gc, int compute_hessian, const ang_t *ang,
                              double *score_out, double g_pt[6], double h_pt[36])
{
    const float gauss_d2 = (float)gc->d2;
    const leaf_t *nb[7];
    int nn = neighborhood7(g, tp, nb);
    /* computePointDerivatives, float (:399-440): x4 = (float)x */
    float x4[3] = {(float)(double)xp[0], (float)(double)xp[1], (float)(double)xp[2]};
    float pg[3][6];
    memset(pg, 0, sizeof pg); pg[0][0] = pg[1][1] = pg[2][2] = 1.0f;
    float xj[8], xh[15];
    for (int r = 0; r < 8; ++r) { float s = ang->j_ang[r][0] * x4[0]; s += ang->j_ang[r][1] * x4[1]; s += ang->j_ang[r][2] * x4[2]; xj[r] = s; }
    pg[1][3] = xj[0]; pg[2][3] = xj[1]; pg[0][4] = xj[2]; pg[1][4] = xj[3]; pg[2][4] = xj[4]; pg[0][5] = xj[5]; pg[1][5] = xj[6]; pg[2][5] = xj[7];
    for (int r = 0; r < 15; ++r) { float s = ang->h_ang[r][0] * x4[0]; s += ang->h_ang[r][1] * x4[1]; s += ang->h_ang[r][2] * x4[2]; xh[r] = s; }
    /* point_hessian_ blocks: ph[i][j] = 3-vector for parameters (i,j), i,j in 3..5 */
    float ph[6][6][3];
    memset(ph, 0, sizeof ph);
    const float a_[3] = {0, xh[0], xh[1]}, b_[3] = {0, xh[2], xh[3]}, c_[3] = {0, xh[4], xh[5]}, d_[3] = {xh[6], xh[7], xh[8]},
                e_[3] = {xh[9], xh[10], xh[11]}, f_[3] = {xh[12], xh[13], xh[14]};
    memcpy(ph[3][3], a_, 12); memcpy(ph[4][3], b_, 12); memcpy(ph[5][3], c_, 12);
    memcpy(ph[3][4], b_, 12); memcpy(ph[4][4], d_, 12); memcpy(ph[5][4], e_, 12);
    memcpy(ph[3][5], c_, 12); memcpy(ph[4][5], e_, 12); memcpy(ph[5][5], f_, 12);
    double score_pt = 0;
    memset(g_pt, 0, 6 * sizeof(double)); memset(h_pt, 0, 36 * sizeof(double));
    for (int k = 0; k < nn; ++k) {
        const leaf_t *cell = nb[k];
        double xt[3] = {(double)tp[0] - cell->mean[0], (double)tp[1] - cell->mean[1], (double)tp[2] - cell->mean[2]};
        float x4t[3] = {(float)xt[0], (float)xt[1], (float)xt[2]};
        float ci[3][3];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) ci[r][c] = (float)cell->icov[r * 3 + c];
        float xc[3];   /* x_trans4 * c_inv4 */
        for (int c = 0; c < 3; ++c) { float s = x4t[0] * ci[0][c]; s += x4t[1] * ci[1][c]; s += x4t[2] * ci[2][c]; xc[c] = s; }
        float dot = x4t[0] * xc[0]; dot += x4t[1] * xc[1]; dot += x4t[2] * xc[2];
        float e = expf(-gauss_d2 * dot * 0.5f);
        float score_inc = (float)(-gc->d1 * (double)e);
        e = gauss_d2 * e;
        if (e > 1 || e < 0 || e != e) continue;
        e = (float)((double)e * gc->d1);
        float cpg[3][6];  /* c_inv4 * point_gradient4 */
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 6; ++c) { float s = ci[r][0] * pg[0][c]; s += ci[r][1] * pg[1][c]; s += ci[r][2] * pg[2][c]; cpg[r][c] = s; }
        float xcpg[6];
        for (int c = 0; c < 6; ++c) { float s = x4t[0] * cpg[0][c]; s += x4t[1] * cpg[1][c]; s += x4t[2] * cpg[2][c]; xcpg[c] = s; }
        for (int c = 0; c < 6; ++c) g_pt[c] += (double)(e * xcpg[c]);
        if (compute_hessian) {
            float pgcpg[6][6];  /* point_gradient4^T * c_inv4_x_point_gradient4 */
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { float s = pg[0][r] * cpg[0][c]; s += pg[1][r] * cpg[1][c]; s += pg[2][r] * cpg[2][c]; pgcpg[r][c] = s; }
            for (int i = 0; i < 6; ++i) {
                float xph[6];
                for (int j = 0; j < 6; ++j) { float s = xc[0] * ph[i][j][0]; s += xc[1] * ph[i][j][1]; s += xc[2] * ph[i][j][2]; xph[j] = s; }
                for (int j = 0; j < 6; ++j)
                    h_pt[i * 6 + j] += (double)(e * (-gauss_d2 * xcpg[i] * xcpg[j] + xph[j] + pgcpg[j][i]));
            }
        }
        score_pt += (double)score_inc;
    }
    *score_out = score_pt;
}

static __thread int g_ndt_threads = 1;      /* set per call from oracle_ndt_params.threads (the entry points below) */

static double compute_derivatives(const ndt_grid *g, const float *src, size_t n, size_t stride, const float *trans, const double p[6],
                                  const gauss_t *gc, int compute_hessian, ang_t *ang, double grad[6], double hess[36])
{
    memset(grad, 0, 6 * sizeof(double)); memset(hess, 0, 36 * sizeof(double));
    angle_derivatives(p, ang);
    double score = 0;
    const int threads = g_ndt_threads > 1 ? g_ndt_threads : 1;
    if (threads == 1 || n < 64) {
        for (size_t idx = 0; idx < n; ++idx) {
            double s_pt, g_pt[6], h_pt[36];
            derivatives_point(g, src + idx * stride, trans + idx * 3, gc, compute_hessian, ang, &s_pt, g_pt, h_pt);
            score += s_pt;
            for (int c = 0; c < 6; ++c) grad[c] += g_pt[c];
            for (int c = 0; c < 36; ++c) hess[c] += h_pt[c];
        }
        return score;
    }
    /* scores[idx], score_gradients[idx], hessians[idx] of :189-197, then the ordered sum of :277-282 */
    const int w = compute_hessian ? 43 : 7;
    double *per = (double *)malloc(n * (size_t)w * sizeof(double));
    if (!per) { g_ndt_threads = 1; return compute_derivatives(g, src, n, stride, trans, p, gc, compute_hessian, ang, grad, hess); }
#pragma omp parallel for num_threads(threads) schedule(guided, 8)
    for (long long idx = 0; idx < (long long)n; ++idx) {
        double s_pt, g_pt[6], h_pt[36];
        derivatives_point(g, src + (size_t)idx * stride, trans + (size_t)idx * 3, gc, compute_hessian, ang, &s_pt, g_pt, h_pt);
        double *o = per + (size_t)idx * w;
        o[0] = s_pt;
        for (int c = 0; c < 6; ++c) o[1 + c] = g_pt[c];
        if (compute_hessian) for (int c = 0; c < 36; ++c) o[7 + c] = h_pt[c];
    }
    for (size_t idx = 0; idx < n; ++idx) {
        const double *o = per + idx * w;
        score += o[0];
        for (int c = 0; c < 6; ++c) grad[c] += o[1 + c];
        if (compute_hessian) for (int c = 0; c < 36; ++c) hess[c] += o[7 + c];
    }
    free(per);
    return score;
}

/* computeHessian / updateHessian (:541-645), double */
static void compute_hessian_d(const ndt_grid *g, const float *src, size_t n, size_t stride, const float *trans, const gauss_t *gc,
                              const ang_t *ang, double hess[36])
{
    memset(hess, 0, 36 * sizeof(double));
    for (size_t idx = 0; idx < n; ++idx) {
        const float *xp = src + idx * stride;
        const float *tp = trans + idx * 3;
        const leaf_t *nb[7];
        int nn = neighborhood7(g, tp, nb);
        double x[3] = {xp[0], xp[1], xp[2]};
        double pg[3][6];
        memset(pg, 0, sizeof pg); pg[0][0] = pg[1][1] = pg[2][2] = 1.0;
        double dj[8], dh[15];
        for (int r = 0; r < 8; ++r) dj[r] = x[0] * ang->jd[r][0] + x[1] * ang->jd[r][1] + x[2] * ang->jd[r][2];
        for (int r = 0; r < 15; ++r) dh[r] = x[0] * ang->hd[r][0] + x[1] * ang->hd[r][1] + x[2] * ang->hd[r][2];
        pg[1][3] = dj[0]; pg[2][3] = dj[1]; pg[0][4] = dj[2]; pg[1][4] = dj[3]; pg[2][4] = dj[4]; pg[0][5] = dj[5]; pg[1][5] = dj[6]; pg[2][5] = dj[7];
        double ph[6][6][3];
        memset(ph, 0, sizeof ph);
        const double a_[3] = {0, dh[0], dh[1]}, b_[3] = {0, dh[2], dh[3]}, c_[3] = {0, dh[4], dh[5]}, d_[3] = {dh[6], dh[7], dh[8]},
                     e_[3] = {dh[9], dh[10], dh[11]}, f_[3] = {dh[12], dh[13], dh[14]};
        memcpy(ph[3][3], a_, 24); memcpy(ph[4][3], b_, 24); memcpy(ph[5][3], c_, 24);
        memcpy(ph[3][4], b_, 24); memcpy(ph[4][4], d_, 24); memcpy(ph[5][4], e_, 24);
        memcpy(ph[3][5], c_, 24); memcpy(ph[4][5], e_, 24); memcpy(ph[5][5], f_, 24);
        for (int k = 0; k < nn; ++k) {
            const leaf_t *cell = nb[k];
            double xt[3] = {(double)tp[0] - cell->mean[0], (double)tp[1] - cell->mean[1], (double)tp[2] - cell->mean[2]};
            const double *ci = cell->icov;
            double cx[3];
            for (int r = 0; r < 3; ++r) cx[r] = ci[r * 3] * xt[0] + ci[r * 3 + 1] * xt[1] + ci[r * 3 + 2] * xt[2];
            double e = gc->d2 * exp(-gc->d2 * (xt[0] * cx[0] + xt[1] * cx[1] + xt[2] * cx[2]) / 2);
            if (e > 1 || e < 0 || e != e) continue;
            e *= gc->d1;
            double cpg[6][3];   /* c_inv * point_gradient.col(i) */
            for (int i = 0; i < 6; ++i) for (int r = 0; r < 3; ++r) cpg[i][r] = ci[r * 3] * pg[0][i] + ci[r * 3 + 1] * pg[1][i] + ci[r * 3 + 2] * pg[2][i];
            for (int i = 0; i < 6; ++i) {
                double xd_i = xt[0] * cpg[i][0] + xt[1] * cpg[i][1] + xt[2] * cpg[i][2];
                for (int j = 0; j < 6; ++j) {
                    double xd_j = xt[0] * cpg[j][0] + xt[1] * cpg[j][1] + xt[2] * cpg[j][2];
                    double cph[3];
                    for (int r = 0; r < 3; ++r) cph[r] = ci[r * 3] * ph[i][j][0] + ci[r * 3 + 1] * ph[i][j][1] + ci[r * 3 + 2] * ph[i][j][2];
                    double t2 = xt[0] * cph[0] + xt[1] * cph[1] + xt[2] * cph[2];
                    double t3 = pg[0][j] * cpg[i][0] + pg[1][j] * cpg[i][1] + pg[2][j] * cpg[i][2];
                    hess[i * 6 + j] += e * (-gc->d2 * xd_i * xd_j + t2 + t3);
                }
            }
        }
    }
}

/* ---- 6x6 SVD solve (JacobiSVD::solve): one-sided Jacobi, pseudo-inverse with Eigen's default threshold ---- */
void oracle_svd6_solve(const double A_in[36], const double b[6], double x[6])
{
    double U[6][6], V[6][6];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { U[i][j] = A_in[i * 6 + j]; V[i][j] = i == j; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < 5; ++p)
            for (int q = p + 1; q < 6; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < 6; ++k) { alpha += U[k][p] * U[k][p]; beta += U[k][q] * U[k][q]; gamma += U[k][p] * U[k][q]; }
                if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 1e-17 * sqrt(alpha * beta)) continue;
                rotated = 1;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < 6; ++k) {
                    double up = U[k][p], uq = U[k][q]; U[k][p] = c * up - s * uq; U[k][q] = s * up + c * uq;
                    double vp = V[k][p], vq = V[k][q]; V[k][p] = c * vp - s * vq; V[k][q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double sig[6], smax = 0;
    for (int j = 0; j < 6; ++j) { double s = 0; for (int k = 0; k < 6; ++k) s += U[k][j] * U[k][j]; sig[j] = sqrt(s); if (sig[j] > smax) smax = sig[j]; }
    const double thr = 6.0 * DBL_EPSILON * smax;
    for (int i = 0; i < 6; ++i) x[i] = 0;
    for (int j = 0; j < 6; ++j) {
        if (!(sig[j] > thr)) continue;
        double ub = 0; for (int k = 0; k < 6; ++k) ub += (U[k][j] / sig[j]) * b[k];
        for (int i = 0; i < 6; ++i) x[i] += V[i][j] * (ub / sig[j]);
    }
}

/* ---- More-Thuente (:649-769) ---- */
static int update_interval(double *a_l, double *f_l, double *g_l, double *a_u, double *f_u, double *g_u, double a_t, double f_t, double g_t)
{
    if (f_t > *f_l) { *a_u = a_t; *f_u = f_t; *g_u = g_t; return 0; }
    else if (g_t * (*a_l - a_t) > 0) { *a_l = a_t; *f_l = f_t; *g_l = g_t; return 0; }
    else if (g_t * (*a_l - a_t) < 0) { *a_u = *a_l; *f_u = *f_l; *g_u = *g_l; *a_l = a_t; *f_l = f_t; *g_l = g_t; return 0; }
    return 1;
}
static double trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t)
{
    if (f_t > f_l) {
        double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
        double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
        return fabs(a_c - a_l) < fabs(a_q - a_l) ? a_c : 0.5 * (a_q + a_c);
    } else if (g_t * g_l < 0) {
        double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
        double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        return fabs(a_c - a_t) >= fabs(a_s - a_t) ? a_c : a_s;
    } else if (fabs(g_t) <= fabs(g_l)) {
        double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
        double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        double nx = fabs(a_c - a_t) < fabs(a_s - a_t) ? a_c : a_s;
        double lim = a_t + 0.66 * (a_u - a_t);
        /* std::min(lim, nx) / std::max(lim, nx) exactly as the STL evaluates them (:758-761): a NaN second argument is
         * dropped, which is how the reference survives a collapsed interval (a_t == a_l gives nx = NaN) */
        return a_t > a_l ? (nx < lim ? nx : lim) : (lim < nx ? nx : lim);
    }
    double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u, w = sqrt(z * z - g_t * g_u);
    return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}

/* test hook: trialValueSelectionMT (:690-769) */
double oracle_ndt_trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t)
{
    return trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, f_t, g_t);
}

typedef struct { const ndt_grid *g; const float *src; size_t n, stride; float *trans; gauss_t gc; ang_t ang; float R[9], t[3]; long n_deriv, n_hess; } ndt_ctx;

/* computeStepLengthMT (:773-932) */
static double step_length_mt(ndt_ctx *c, const double x[6], double dir[6], double step_init, double step_max, double step_min,
                             double *score, double grad[6], double hess[36])
{
    double phi_0 = -*score, d_phi_0 = 0;
    for (int i = 0; i < 6; ++i) d_phi_0 += grad[i] * dir[i];
    d_phi_0 = -d_phi_0;
    if (d_phi_0 >= 0) {
        if (d_phi_0 == 0) return 0;
        d_phi_0 *= -1; for (int i = 0; i < 6; ++i) dir[i] *= -1;
    }
    const int max_it = 10; int it = 0;
    const double mu = 1.e-4, nu = 0.9;
    double a_l = 0, a_u = 0;
    double f_l = phi_0 - phi_0 - mu * d_phi_0 * a_l, g_l = d_phi_0 - mu * d_phi_0;
    double f_u = phi_0 - phi_0 - mu * d_phi_0 * a_u, g_u = d_phi_0 - mu * d_phi_0;
    int interval_converged = (step_max - step_min) < 0, open_interval = 1;
    double a_t = step_init; if (a_t > step_max) a_t = step_max; if (a_t < step_min) a_t = step_min;
    double x_t[6];
    for (int i = 0; i < 6; ++i) x_t[i] = x[i] + dir[i] * a_t;
    pose_from_p(x_t, c->R, c->t);
    transform_cloud_f(c->src, c->n, c->stride, c->R, c->t, c->trans);
    *score = compute_derivatives(c->g, c->src, c->n, c->stride, c->trans, x_t, &c->gc, 1, &c->ang, grad, hess); c->n_deriv++;
    double phi_t = -*score, d_phi_t = 0;
    for (int i = 0; i < 6; ++i) d_phi_t += grad[i] * dir[i];
    d_phi_t = -d_phi_t;
    double psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t, d_psi_t = d_phi_t - mu * d_phi_0;
    while (!interval_converged && it < max_it && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
        if (open_interval) a_t = trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
        else a_t = trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
        if (a_t > step_max) a_t = step_max;
        if (a_t < step_min) a_t = step_min;
        for (int i = 0; i < 6; ++i) x_t[i] = x[i] + dir[i] * a_t;
        pose_from_p(x_t, c->R, c->t);
        transform_cloud_f(c->src, c->n, c->stride, c->R, c->t, c->trans);
        *score = compute_derivatives(c->g, c->src, c->n, c->stride, c->trans, x_t, &c->gc, 0, &c->ang, grad, hess); c->n_deriv++;
        phi_t = -*score; d_phi_t = 0;
        for (int i = 0; i < 6; ++i) d_phi_t += grad[i] * dir[i];
        d_phi_t = -d_phi_t;
        psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t; d_psi_t = d_phi_t - mu * d_phi_0;
        if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
            open_interval = 0;
            f_l = f_l + phi_0 - mu * d_phi_0 * a_l; g_l = g_l + mu * d_phi_0;
            f_u = f_u + phi_0 - mu * d_phi_0 * a_u; g_u = g_u + mu * d_phi_0;
        }
        if (open_interval) interval_converged = update_interval(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, psi_t, d_psi_t);
        else interval_converged = update_interval(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, phi_t, d_phi_t);
        it++;
    }
    if (it) { compute_hessian_d(c->g, c->src, c->n, c->stride, c->trans, &c->gc, &c->ang, hess); c->n_hess++; }
    return a_t;
}

int g_ndt_rotation_polar = 0;
/* orthogonal polar factor U V^T of a 3x3 (Newton iteration X <- (X + X^-T) / 2 in double, rounded to float): what
 * Eigen::Transform<float,3,Affine>::rotation() extracts by SVD.  Only used to MEASURE how much the documented deviation
 * (rotation() taken as the linear part) can matter: the guess handed over is a rotation rounded to float. */
static void polar_factor_f(const float R[9], float out[9])
{
    double X[9];
    for (int i = 0; i < 9; ++i) X[i] = (double)R[i];
    for (int it = 0; it < 12; ++it) {
        const double c00 = X[4] * X[8] - X[5] * X[7], c01 = X[5] * X[6] - X[3] * X[8], c02 = X[3] * X[7] - X[4] * X[6];
        const double det = X[0] * c00 + X[1] * c01 + X[2] * c02;
        /* inverse transpose = cofactor matrix / det */
        const double C[9] = {c00, c01, c02,
                             X[2] * X[7] - X[1] * X[8], X[0] * X[8] - X[2] * X[6], X[1] * X[6] - X[0] * X[7],
                             X[1] * X[5] - X[2] * X[4], X[2] * X[3] - X[0] * X[5], X[0] * X[4] - X[1] * X[3]};
        for (int i = 0; i < 9; ++i) X[i] = 0.5 * (X[i] + C[i] / det);
    }
    for (int i = 0; i < 9; ++i) out[i] = (float)X[i];
}

/* Matrix3f::eulerAngles(0,1,2) (Eigen 3.3/3.4), R row-major float */
static void euler_xyz_f(const float R[9], float out[3])
{
#define RM(i, j) R[(i) * 3 + (j)]
    const float pi = 3.14159265358979323846f;
    float r0 = atan2f(RM(1, 2), RM(2, 2));
    float c2 = sqrtf(RM(0, 0) * RM(0, 0) + RM(0, 1) * RM(0, 1));
    float r1;
    if (r0 > 0.f) { if (r0 > 0.f) r0 -= pi; else r0 += pi; r1 = atan2f(-RM(0, 2), -c2); }
    else r1 = atan2f(-RM(0, 2), c2);
    float s1 = sinf(r0), c1 = cosf(r0);
    float r2 = atan2f(s1 * RM(2, 0) - c1 * RM(1, 0), c1 * RM(1, 1) - s1 * RM(2, 1));
    out[0] = -r0; out[1] = -r1; out[2] = -r2;
#undef RM
}

/*
 * Full scan2Map.  pose in/out column-major f64 (cast to f32 and back, NdtRegister.cpp:27-28).
 * info: [0] iterations (nr_iterations_), [1] derivative passes, [2] double-Hessian passes.  Returns hasConverged().
 */
int oracle_ndt_scan2map(const float *src, size_t n_src, const float *dst, size_t n_dst, size_t stride, double pose[16],
                        const oracle_ndt_params *prm, long info[3], double *final_score)
{
    g_ndt_threads = prm && prm->threads > 1 ? prm->threads : 1;
    ndt_ctx c; memset(&c, 0, sizeof c);
    ndt_grid *g = ndt_grid_build(dst, n_dst, stride, prm);
    c.g = g; c.src = src; c.n = n_src; c.stride = stride; c.gc = gauss_consts(prm);
    c.trans = (float *)malloc(sizeof(float) * 3 * (n_src ? n_src : 1));
    float G[16]; for (int i = 0; i < 16; ++i) G[i] = (float)pose[i];
    int is_identity = 1; for (int i = 0; i < 16; ++i) if (G[i] != ((i % 5 == 0) ? 1.0f : 0.0f)) is_identity = 0;
    float Rg[9], tg[3];
    for (int r = 0; r < 3; ++r) { for (int cc = 0; cc < 3; ++cc) Rg[r * 3 + cc] = G[cc * 4 + r]; tg[r] = G[12 + r]; }
    float final_R[9], final_t[3];
    memcpy(final_R, Rg, sizeof Rg); memcpy(final_t, tg, sizeof tg);
    if (!is_identity) transform_cloud_f(src, n_src, stride, Rg, tg, c.trans);
    else for (size_t i = 0; i < n_src; ++i) for (int d = 0; d < 3; ++d) c.trans[i * 3 + d] = src[i * stride + d];
    float Rp[9];
    memcpy(Rp, Rg, sizeof Rp);
    if (g_ndt_rotation_polar) polar_factor_f(Rg, Rp);      /* Eigen's Transform<float,3,Affine>::rotation(): see oracle_set_variant */
    float eul[3]; euler_xyz_f(Rp, eul);
    double p[6] = {tg[0], tg[1], tg[2], eul[0], eul[1], eul[2]}, delta_p[6], grad[6], hess[36];
    double score = compute_derivatives(g, src, n_src, stride, c.trans, p, &c.gc, 1, &c.ang, grad, hess); c.n_deriv++;
    int converged = 0, nr_it = 0;
    while (!converged) {
        double rhs[6]; for (int i = 0; i < 6; ++i) rhs[i] = -grad[i];
        oracle_svd6_solve(hess, rhs, delta_p);
        double nrm = 0; for (int i = 0; i < 6; ++i) nrm += delta_p[i] * delta_p[i];
        nrm = sqrt(nrm);
        if (nrm == 0 || nrm != nrm) { converged = nrm == nrm; break; }
        for (int i = 0; i < 6; ++i) delta_p[i] /= nrm;
        nrm = step_length_mt(&c, p, delta_p, nrm, prm->step_size, prm->trans_eps / 2, &score, grad, hess);
        for (int i = 0; i < 6; ++i) delta_p[i] *= nrm;
        memcpy(final_R, c.R, sizeof final_R); memcpy(final_t, c.t, sizeof final_t);
        for (int i = 0; i < 6; ++i) p[i] += delta_p[i];
        if (nr_it > prm->max_iters || (nr_it && fabs(nrm) < prm->trans_eps)) converged = 1;
        nr_it++;
    }
    memset(pose, 0, 16 * sizeof(double));
    for (int r = 0; r < 3; ++r) { for (int cc = 0; cc < 3; ++cc) pose[cc * 4 + r] = (double)final_R[r * 3 + cc]; pose[12 + r] = (double)final_t[r]; }
    pose[15] = 1.0;
    if (info) { info[0] = nr_it; info[1] = c.n_deriv; info[2] = c.n_hess; }
    if (final_score) *final_score = score;
    free(c.trans); ndt_grid_free(g);
    return converged;
}

/* One computeDerivatives pass at the pose p = [t; euler xyz] (score, gradient, Hessian) and, when
 * hess_d != NULL, the double-precision Hessian of computeHessian at the same pose. */
double oracle_ndt_derivatives(const float *src, size_t n_src, const float *dst, size_t n_dst, size_t stride, const double p[6],
                              const oracle_ndt_params *prm, double grad[6], double hess[36], double *hess_d)
{
    g_ndt_threads = prm && prm->threads > 1 ? prm->threads : 1;
    ndt_grid *g = ndt_grid_build(dst, n_dst, stride, prm);
    gauss_t gc = gauss_consts(prm);
    ang_t ang; float R[9], t[3];
    float *trans = (float *)malloc(sizeof(float) * 3 * (n_src ? n_src : 1));
    pose_from_p(p, R, t);
    transform_cloud_f(src, n_src, stride, R, t, trans);
    double score = compute_derivatives(g, src, n_src, stride, trans, p, &gc, 1, &ang, grad, hess);
    if (hess_d) compute_hessian_d(g, src, n_src, stride, trans, &gc, &ang, hess_d);
    free(trans); ndt_grid_free(g);
    return score;
}
