/*
 * oracle/vgicp_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (C99 + OpenMP) of the reference's voxelised GICP scan-to-map
 * registration, PCR::VgicpRegister::scan2Map.  Follows (reference tree):
 *   PCR/src/VgicpRegister.cpp:13,30-45                wrapper, resolution 1.0, f32 pose in/out
 *   third_parties/pclomp/src/fast_gicp_impl.hpp:103-112,241-297   per-point covariances
 *       (20-NN, 4x20 f64 neighbours, cov/20, JacobiSVD, PLANE: U diag(1,1,1e-3) V^T)
 *   third_parties/pclomp/src/pclomp/fast_vgicp_voxel.hpp:105-122,129-174   ADDITIVE voxels,
 *       voxel_coord = floor(x/res - 0.5), lookup
 *   third_parties/pclomp/src/fast_vgicp_impl.hpp:73-204   correspondences (DIRECT1),
 *       Mahalanobis, linearize, compute_error
 *   third_parties/pclomp/src/lsq_registration_impl.hpp:53-171   LM driver, is_converged
 *   third_parties/pclomp/src/so3/so3.hpp:21-77        skewd, so3_exp
 *
 * PARITY STATUS: "parity unpinned".  The reference ships no vectors for this path and it
 * cannot be built here (needs PCL, FLANN, Eigen, Boost).  PCL/FLANN/Eigen behaviour is
 * restated from published semantics: pcl::search::KdTree::nearestKSearch = exact k-NN with
 * float squared distances (FLANN L2_Simple<float>); JacobiSVD of a symmetric PSD 3x3 =
 * its eigen-decomposition (cyclic Jacobi here); Matrix4d::inverse of blkdiag(S,1) =
 * blkdiag(S^-1,1) (adjugate here).  tests/ cross-check each piece against numpy.
 * Documented deviations: equal-distance k-NN ties broken by lower index (FLANN: traversal
 * order); H, b, error summed in source order (the reference: per-OpenMP-thread partials).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* k-d tree from loam_oracle.c */
typedef struct kd_tree kd_tree;
kd_tree *oracle_kd_build(const float *pts, size_t n, size_t stride_floats);
void oracle_kd_free(kd_tree *t);
int oracle_kd_knn_f32(const kd_tree *t, const float q[3], int k, int32_t *idx_out, float *d_out);

typedef struct {
    double resolution;   /* 1.0   VgicpRegister.cpp:13 */
    int k_corr;          /* 20    fast_gicp_impl.hpp:16 */
    int max_iters;       /* 64    lsq_registration_impl.hpp:11 */
    int lm_inner;        /* 10    lsq_registration_impl.hpp:17 */
    double rot_eps;      /* 2e-3  lsq_registration_impl.hpp:12 */
    double trans_eps;    /* 5e-4  lsq_registration_impl.hpp:13 */
    double lm_init;      /* 1e-9  lsq_registration_impl.hpp:18 */
    int threads;
} oracle_vgicp_params;

void oracle_vgicp_default_params(oracle_vgicp_params *p)
{
    p->resolution = 1.0; p->k_corr = 20; p->max_iters = 64; p->lm_inner = 10;
    p->rot_eps = 2e-3; p->trans_eps = 5e-4; p->lm_init = 1e-9; p->threads = 1;
}

/* ---- symmetric 3x3 eigen-decomposition (cyclic Jacobi), eigenvalues descending ---- */
void oracle_sym3_eig(const double A[9], double w[3], double V[9])
{
    double a[3][3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a[i][j] = A[i * 3 + j];
    for (int sweep = 0; sweep < 32; ++sweep) {
        double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-22 * diag) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {       /* A <- A J */
                    double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {       /* A <- J^T A */
                    double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {       /* V <- V J */
                    double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq; v[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i) for (int j = i + 1; j < 3; ++j) if (a[ord[j]][ord[j]] > a[ord[i]][ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    for (int i = 0; i < 3; ++i) { w[i] = a[ord[i]][ord[i]]; for (int k = 0; k < 3; ++k) V[k * 3 + i] = v[k][ord[i]]; }
}

/* fast_gicp_impl.hpp:241-297, PLANE regularisation.  covs: n x 9 (3x3 row-major; the
 * reference's 4x4 has a zero 4th row/column). */
void oracle_vgicp_covariances(const float *pts, size_t n, size_t stride, int k, double *covs, int threads)
{
    kd_tree *t = oracle_kd_build(pts, n, stride);
    int nth = threads > 0 ? threads : 1;
#pragma omp parallel for num_threads(nth) schedule(static)
    for (long i = 0; i < (long)n; ++i) {
        int32_t idx[32]; float d2[32];
        const float *q = pts + (size_t)i * stride;
        int found = oracle_kd_knn_f32(t, q, k, idx, d2);
        double mean[3] = {0, 0, 0}, nb[32][3];
        for (int j = 0; j < found; ++j) for (int d = 0; d < 3; ++d) nb[j][d] = (double)pts[(size_t)idx[j] * stride + d];
        /* neighbors has k_correspondences_ columns; with fewer points than k the missing
         * columns stay uninitialised in the reference -- clouds here always have >= k points */
        for (int j = 0; j < found; ++j) for (int d = 0; d < 3; ++d) mean[d] += nb[j][d];
        for (int d = 0; d < 3; ++d) mean[d] /= (double)k;
        double C[9] = {0};
        for (int j = 0; j < found; ++j) {
            double c0 = nb[j][0] - mean[0], c1 = nb[j][1] - mean[1], c2 = nb[j][2] - mean[2];
            C[0] += c0 * c0; C[1] += c0 * c1; C[2] += c0 * c2; C[4] += c1 * c1; C[5] += c1 * c2; C[8] += c2 * c2;
        }
        C[3] = C[1]; C[6] = C[2]; C[7] = C[5];
        for (int e = 0; e < 9; ++e) C[e] /= (double)k;
        double w[3], V[9];
        oracle_sym3_eig(C, w, V);
        const double val[3] = {1.0, 1.0, 1e-3};
        double *out = covs + (size_t)i * 9;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
            double s = 0; for (int e = 0; e < 3; ++e) s += V[r * 3 + e] * val[e] * V[c * 3 + e];
            out[r * 3 + c] = s;
        }
    }
    oracle_kd_free(t);
}

/* ---- voxel map (ADDITIVE), open-addressing hash ---- */
typedef struct { int32_t c[3]; int32_t n; double mean[3]; double cov[9]; } vox;
typedef struct { vox *v; size_t cap; double res; } voxmap;

static inline size_t vox_hash(int32_t x, int32_t y, int32_t z) {
    uint64_t h = (uint64_t)(uint32_t)x * 73856093u ^ (uint64_t)(uint32_t)y * 19349663u ^ (uint64_t)(uint32_t)z * 83492791u;
    h ^= h >> 29; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 32;
    return (size_t)h;
}
static inline void vox_coord(double res, const double p[3], int32_t c[3]) {
    for (int d = 0; d < 3; ++d) c[d] = (int32_t)floor(p[d] / res - 0.5);   /* fast_vgicp_voxel.hpp:158-160 */
}
static vox *vox_find(const voxmap *m, const int32_t c[3], int create) {
    size_t i = vox_hash(c[0], c[1], c[2]) & (m->cap - 1);
    for (;;) {
        vox *v = &m->v[i];
        if (v->n == 0) { if (!create) return NULL; v->c[0] = c[0]; v->c[1] = c[1]; v->c[2] = c[2]; return v; }
        if (v->c[0] == c[0] && v->c[1] == c[1] && v->c[2] == c[2]) return v;
        i = (i + 1) & (m->cap - 1);
    }
}
static voxmap *voxmap_build(const float *pts, size_t n, size_t stride, const double *covs, double res) {
    voxmap *m = (voxmap *)calloc(1, sizeof *m);
    m->cap = 1024; while (m->cap < 2 * n + 16) m->cap <<= 1;
    m->v = (vox *)calloc(m->cap, sizeof(vox)); m->res = res;
    for (size_t i = 0; i < n; ++i) {
        double p[3] = {pts[i * stride], pts[i * stride + 1], pts[i * stride + 2]};
        int32_t c[3]; vox_coord(res, p, c);
        vox *v = vox_find(m, c, 1);
        v->n++;
        for (int d = 0; d < 3; ++d) v->mean[d] += p[d];
        for (int e = 0; e < 9; ++e) v->cov[e] += covs[i * 9 + e];
    }
    for (size_t i = 0; i < m->cap; ++i) if (m->v[i].n) {
        for (int d = 0; d < 3; ++d) m->v[i].mean[d] /= m->v[i].n;
        for (int e = 0; e < 9; ++e) m->v[i].cov[e] /= m->v[i].n;
    }
    return m;
}
static void voxmap_free(voxmap *m) { if (m) { free(m->v); free(m); } }

/* query helper for tests: voxel (n, mean, cov) containing point p; returns n (0 = none) */
int oracle_vgicp_voxel_at(const float *pts, size_t n, size_t stride, const double *covs, double res, const double p[3],
                          double mean[3], double cov[9])
{
    voxmap *m = voxmap_build(pts, n, stride, covs, res);
    int32_t c[3]; vox_coord(res, p, c);
    vox *v = vox_find(m, c, 0);
    int cnt = 0;
    if (v) { cnt = v->n; memcpy(mean, v->mean, sizeof v->mean); memcpy(cov, v->cov, sizeof v->cov); }
    voxmap_free(m);
    return cnt;
}

static void inv3_sym(const double S[9], double out[9]) {
    double a = S[0], b = S[1], c = S[2], d = S[4], e = S[5], f = S[8];
    double A = d * f - e * e, B = c * e - b * f, Cc = b * e - c * d;
    double det = a * A + b * B + c * Cc;
    double id = 1.0 / det;
    out[0] = A * id; out[1] = B * id; out[2] = Cc * id;
    out[3] = out[1]; out[4] = (a * f - c * c) * id; out[5] = (b * c - a * e) * id;
    out[6] = out[2]; out[7] = out[5]; out[8] = (a * d - b * b) * id;
}

typedef struct { int32_t src; const vox *v; double M[9]; } corr_t;

static void apply_T(const double T[16], const double p[3], double out[3]) {
    for (int i = 0; i < 3; ++i) out[i] = T[0 * 4 + i] * p[0] + T[1 * 4 + i] * p[1] + T[2 * 4 + i] * p[2] + T[3 * 4 + i] * 1.0;
}

/* fast_vgicp_impl.hpp:73-116 */
static size_t update_corr(const voxmap *m, const float *src, size_t n, size_t stride, const double *src_covs,
                          const double T[16], corr_t *corr)
{
    size_t nc = 0;
    for (size_t i = 0; i < n; ++i) {
        double p[3] = {src[i * stride], src[i * stride + 1], src[i * stride + 2]}, tp[3];
        apply_T(T, p, tp);
        int32_t c[3]; vox_coord(m->res, tp, c);
        const vox *v = vox_find(m, c, 0);
        if (!v) continue;
        /* RCR = cov_B + T cov_A T^T (the 4th row/col of cov_A is zero) */
        const double *CA = src_covs + i * 9;
        double RC[9], RCR[9];
        for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) {
            double s = 0; for (int k = 0; k < 3; ++k) s += T[k * 4 + r] * CA[k * 3 + cc];
            RC[r * 3 + cc] = s;
        }
        for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) {
            double s = 0; for (int k = 0; k < 3; ++k) s += RC[r * 3 + k] * T[k * 4 + cc];
            RCR[r * 3 + cc] = v->cov[r * 3 + cc] + s;
        }
        corr[nc].src = (int32_t)i; corr[nc].v = v;
        inv3_sym(RCR, corr[nc].M);
        ++nc;
    }
    return nc;
}

/* fast_vgicp_impl.hpp:119-204.  H (36 row-major) / b (6) optional.  d = [rot; trans]. */
static double eval_cost(const corr_t *corr, size_t nc, const float *src, size_t stride, const double T[16], double *H, double *b)
{
    double sum = 0;
    if (H) { memset(H, 0, 36 * sizeof(double)); memset(b, 0, 6 * sizeof(double)); }
    for (size_t k = 0; k < nc; ++k) {
        size_t i = (size_t)corr[k].src;
        double p[3] = {src[i * stride], src[i * stride + 1], src[i * stride + 2]}, tp[3], e[3], Me[3];
        apply_T(T, p, tp);
        const vox *v = corr[k].v; const double *M = corr[k].M;
        for (int d = 0; d < 3; ++d) e[d] = v->mean[d] - tp[d];
        double w = sqrt((double)v->n);
        for (int r = 0; r < 3; ++r) Me[r] = M[r * 3] * e[0] + M[r * 3 + 1] * e[1] + M[r * 3 + 2] * e[2];
        sum += w * (e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2]);
        if (!H) continue;
        /* J = [skew(tp) | -I]  (3x6) */
        double J[3][6] = {{0, -tp[2], tp[1], -1, 0, 0}, {tp[2], 0, -tp[0], 0, -1, 0}, {-tp[1], tp[0], 0, 0, 0, -1}};
        double MJ[3][6];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 6; ++c) MJ[r][c] = M[r * 3] * J[0][c] + M[r * 3 + 1] * J[1][c] + M[r * 3 + 2] * J[2][c];
        for (int r = 0; r < 6; ++r) {
            for (int c = 0; c < 6; ++c) H[r * 6 + c] += w * (J[0][r] * MJ[0][c] + J[1][r] * MJ[1][c] + J[2][r] * MJ[2][c]);
            b[r] += w * (J[0][r] * Me[0] + J[1][r] * Me[1] + J[2][r] * Me[2]);
        }
    }
    return sum;
}

void oracle_ldlt6_solve(const double M_in[36], const double rhs[6], double x[6]);

/* so3.hpp:58-77 + Quaterniond::toRotationMatrix; delta = [exp(d[0:3]) ; d[3:6]] column-major */
static void make_delta(const double d[6], double D[16])
{
    double th2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], imag, real;
    if (th2 < 1e-10) {
        double q4 = th2 * th2;
        imag = 0.5 - 1.0 / 48.0 * th2 + 1.0 / 3840.0 * q4;
        real = 1.0 - 1.0 / 8.0 * th2 + 1.0 / 384.0 * q4;
    } else {
        double th = sqrt(th2), h = 0.5 * th;
        imag = sin(h) / th; real = cos(h);
    }
    double w = real, x = imag * d[0], y = imag * d[1], z = imag * d[2];
    double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x,
           tyy = ty * y, tyz = tz * y, tzz = tz * z;
    memset(D, 0, 16 * sizeof(double));
#define Dm(i, j) D[(j) * 4 + (i)]
    Dm(0, 0) = 1 - (tyy + tzz); Dm(0, 1) = txy - twz; Dm(0, 2) = txz + twy;
    Dm(1, 0) = txy + twz; Dm(1, 1) = 1 - (txx + tzz); Dm(1, 2) = tyz - twx;
    Dm(2, 0) = txz - twy; Dm(2, 1) = tyz + twx; Dm(2, 2) = 1 - (txx + tyy);
    Dm(0, 3) = d[3]; Dm(1, 3) = d[4]; Dm(2, 3) = d[5]; Dm(3, 3) = 1;
#undef Dm
}

static void mul44(const double A[16], const double B[16], double C[16]) {
    double o[16];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) { double s = 0; for (int k = 0; k < 4; ++k) s += A[k * 4 + r] * B[c * 4 + k]; o[c * 4 + r] = s; }
    /* keep the result an Isometry: last row (0,0,0,1) */
    o[3] = o[7] = o[11] = 0; o[15] = 1;
    memcpy(C, o, sizeof o);
}

/* lsq_registration_impl.hpp:82-91 */
static int is_converged(const double D[16], double rot_eps, double trans_eps) {
    double rmax = 0, tmax = 0;
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) { double v = fabs(D[c * 4 + r] - (r == c ? 1.0 : 0.0)) * (1.0 / rot_eps); if (v > rmax) rmax = v; }
    for (int r = 0; r < 3; ++r) { double v = fabs(D[12 + r]) * (1.0 / trans_eps); if (v > tmax) tmax = v; }
    return (rmax > tmax ? rmax : tmax) < 1;
}

/*
 * Full scan2Map.  pose in/out column-major f64 (cast to f32 and back like VgicpRegister.cpp:36-37).
 * Optional precomputed covariances (NULL -> computed).  info[0] = outer iterations run,
 * info[1] = linearisations, info[2] = error evaluations, info[3] = correspondences of the last linearisation.
 * Returns converged.
 */
int oracle_vgicp_scan2map(const float *src, size_t n_src, const float *dst, size_t n_dst, size_t stride, double pose[16],
                          const oracle_vgicp_params *prm, const double *src_covs_in, const double *dst_covs_in, long info[4])
{
    double *sc = NULL, *dc = NULL;
    if (!src_covs_in) { sc = (double *)malloc(sizeof(double) * 9 * (n_src ? n_src : 1)); oracle_vgicp_covariances(src, n_src, stride, prm->k_corr, sc, prm->threads); src_covs_in = sc; }
    if (!dst_covs_in) { dc = (double *)malloc(sizeof(double) * 9 * (n_dst ? n_dst : 1)); oracle_vgicp_covariances(dst, n_dst, stride, prm->k_corr, dc, prm->threads); dst_covs_in = dc; }
    voxmap *vm = voxmap_build(dst, n_dst, stride, dst_covs_in, prm->resolution);
    corr_t *corr = (corr_t *)malloc(sizeof(corr_t) * (n_src ? n_src : 1));
    double x0[16];
    for (int i = 0; i < 16; ++i) x0[i] = (double)(float)pose[i];     /* guess as Matrix4f */
    double lambda = -1.0;
    int converged = 0;
    long n_lin = 0, n_err = 0, outer = 0, last_nc = 0;
    for (int it = 0; it < prm->max_iters && !converged; ++it) {
        outer = it + 1;
        double H[36], b[6], D[16];
        size_t nc = update_corr(vm, src, n_src, stride, src_covs_in, x0, corr);
        last_nc = (long)nc;
        double y0 = eval_cost(corr, nc, src, stride, x0, H, b); ++n_lin;
        if (lambda < 0.0) { double mx = 0; for (int i = 0; i < 6; ++i) if (fabs(H[i * 7]) > mx) mx = fabs(H[i * 7]); lambda = prm->lm_init * mx; }
        double nu = 2.0;
        int ok = 0;
        for (int i = 0; i < prm->lm_inner; ++i) {
            double A[36], rhs[6], d[6], xi[16];
            memcpy(A, H, sizeof A);
            for (int k = 0; k < 6; ++k) { A[k * 7] += lambda; rhs[k] = -b[k]; }
            oracle_ldlt6_solve(A, rhs, d);
            make_delta(d, D);
            mul44(D, x0, xi);
            double yi = eval_cost(corr, nc, src, stride, xi, NULL, NULL); ++n_err;
            double den = 0; for (int k = 0; k < 6; ++k) den += d[k] * (lambda * d[k] - b[k]);
            double rho = (y0 - yi) / den;
            if (rho < 0) {
                if (is_converged(D, prm->rot_eps, prm->trans_eps)) { ok = 1; break; }
                lambda = nu * lambda; nu = 2 * nu;
                continue;
            }
            memcpy(x0, xi, sizeof xi);
            double f = 1 - pow(2 * rho - 1, 3);
            lambda = lambda * (f > 1.0 / 3.0 ? f : 1.0 / 3.0);
            ok = 1; break;
        }
        if (!ok) break;                                       /* "lm not converged!!" */
        converged = is_converged(D, prm->rot_eps, prm->trans_eps);
    }
    for (int i = 0; i < 16; ++i) pose[i] = (double)(float)x0[i];     /* final_transformation_ is Matrix4f */
    if (info) { info[0] = outer; info[1] = n_lin; info[2] = n_err; info[3] = last_nc; }
    free(corr); voxmap_free(vm); free(sc); free(dc);
    return converged;
}

/* One linearisation at `pose` (used as given, f64): H, b, error, number of correspondences. */
long oracle_vgicp_linearize(const float *src, size_t n_src, const float *dst, size_t n_dst, size_t stride, const double pose[16],
                            const oracle_vgicp_params *prm, const double *src_covs, const double *dst_covs, double H[36], double b[6],
                            double *err)
{
    voxmap *vm = voxmap_build(dst, n_dst, stride, dst_covs, prm->resolution);
    corr_t *corr = (corr_t *)malloc(sizeof(corr_t) * (n_src ? n_src : 1));
    size_t nc = update_corr(vm, src, n_src, stride, src_covs, pose, corr);
    *err = eval_cost(corr, nc, src, stride, pose, H, b);
    free(corr); voxmap_free(vm);
    return (long)nc;
}

/* compute_error (fast_vgicp_impl.hpp:183-204) at `pose_eval` on the correspondences of a linearisation at `pose_lin`: what an LM
 * trial evaluates (lsq_registration_impl.hpp:141).  Test infrastructure for the optimiser state machine (csrc/vgicp_opt.h). */
double oracle_vgicp_error(const float *src, size_t n_src, const float *dst, size_t n_dst, size_t stride, const double pose_lin[16],
                          const double pose_eval[16], const oracle_vgicp_params *prm, const double *src_covs, const double *dst_covs)
{
    voxmap *vm = voxmap_build(dst, n_dst, stride, dst_covs, prm->resolution);
    corr_t *corr = (corr_t *)malloc(sizeof(corr_t) * (n_src ? n_src : 1));
    size_t nc = update_corr(vm, src, n_src, stride, src_covs, pose_lin, corr);
    const double e = eval_cost(corr, nc, src, stride, pose_eval, NULL, NULL);
    free(corr); voxmap_free(vm);
    return e;
}

/* pcl::Registration::getFitnessScore(): mean squared 1-NN distance (float kd-tree) of the
 * source transformed by the final (f32) transformation; -1... PCL returns DBL_MAX when empty */
double oracle_fitness_score(const float *src, size_t n_src, const float *dst, size_t n_dst, size_t stride, const double pose[16], double max_range)
{
    kd_tree *t = oracle_kd_build(dst, n_dst, stride);
    float Tf[16]; for (int i = 0; i < 16; ++i) Tf[i] = (float)pose[i];
    double sum = 0; long nr = 0;
    for (size_t i = 0; i < n_src; ++i) {
        const float *p = src + i * stride; float q[3];
        /* pcl::transformPointCloud: float arithmetic */
        for (int r = 0; r < 3; ++r) q[r] = Tf[0 * 4 + r] * p[0] + Tf[1 * 4 + r] * p[1] + Tf[2 * 4 + r] * p[2] + Tf[3 * 4 + r];
        int32_t idx; float d2;
        if (oracle_kd_knn_f32(t, q, 1, &idx, &d2) == 1 && (double)d2 <= max_range) { sum += (double)d2; ++nr; }
    }
    oracle_kd_free(t);
    return nr > 0 ? sum / (double)nr : DBL_MAX;
}
