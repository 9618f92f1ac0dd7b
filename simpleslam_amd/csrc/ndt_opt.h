// ndt_opt.h -- the NDT optimiser as a state machine that is fed one evaluation pass at a time.
//
// pclomp::NormalDistributionsTransform::computeTransformation (ndt_omp_impl.hpp:81-171) is a Newton iteration whose step
// length comes from a More-Thuente line search (computeStepLengthMT, :735-932; trialValueSelectionMT :649-711,
// updateIntervalMT :713-733).  Every decision it takes depends on 43 numbers -- score, gradient, Hessian -- summed over the
// scan at one pose.  Here the loop is turned inside out: ndt_ctl_step() consumes the sums of the pass that has just run and
// leaves in NdtCtl what the NEXT pass must evaluate (pose, angle tables, which of the three kernels), or `done`.
// The same function runs
//   * on the device, in the single-block kernel that folds a pass's partial sums (ndt.hip: ndt_fold_ctl_kernel), so that an
//     alignment is a chain of launches with no host round trip;
//   * on the host, for sharded targets, where the sums of every pass cross the ranks through a host collective (capi.hip).
// Host and device differ in two places, both marked below: the libm behind sin/cos, and the 6x6 Newton solve (the host keeps the
// restated JacobiSVD::solve; the device eliminates with partial pivoting and hands the call back to the host -- `bail` -- when the
// pivots say that the pseudo-inverse of the SVD could differ from the inverse).
#pragma once
#include <math.h>
#include <stdint.h>

#include "pcr_internal.h"

namespace pcr {

#if defined(__HIPCC__)
#define NDT_HD __host__ __device__
#define NDT_HD_FLAT __host__ __device__ __forceinline__      // the controller's own code: a call on the device saves and restores ~60 registers through scratch memory
#else
#define NDT_HD
#define NDT_HD_FLAT inline
#endif

enum : int { kNdtPassDerivH = 0, kNdtPassDeriv = 1, kNdtPassHessian = 2, kNdtPassNone = 3 };
enum : int { kNdtPhaseInit = 0, kNdtPhaseLsFirst = 1, kNdtPhaseLsLoop = 2, kNdtPhaseLsHess = 3, kNdtPhaseLsFirstH = 4 };

struct NdtCtl {
    // ---- what the next pass evaluates (read by the pass kernel) ----
    NdtPose T;
    NdtAngles ang;
    int32_t kind, phase;
    // ---- parameters ----
    double step_size, trans_eps;
    int32_t max_iters, late_h;      // late_h: float Hessians evaluated at a first trial point AFTER the search had ended there (see ctl_advance)
    // ---- optimiser state ----
    double p[6], dir[6], x_t[6], grad[6], hess[36], score;
    double phi_0, d_phi_0, a_l, f_l, g_l, a_u, f_u, g_u, a_t, step_min, step_max;
    int32_t open_interval, interval_converged, it, nr_it, conv, n_deriv, n_hess, bail;
    NdtPose final_T;
    int32_t done, passes;
    uint32_t ticks[4];      // profiling: 100 MHz ticks spent in fold / controller step / write-back, summed over the passes
    // ---- the point (score, grad) were last evaluated at: a request for the same sums at the same point is answered from here ----
    double x_eval[6];
    int32_t eval_valid, replayed, replay_off, need_h;      // need_h: see ctl_advance; replay_off: PCR_NDT_NO_REPLAY (A/B runs): every request becomes a pass
};

// what the device-resident loop reports to the host (host-mapped memory; `seq` is written last)
struct NdtOut {
    NdtPose final_T;
    double score;
    int32_t conv, nr_it, n_deriv, n_hess, bail, passes, grid_overflow, grid_empty, grid_stale;
    int32_t roi_escapes;    // lookups that hit a qualifying voxel outside the region the target was prepared for (RoiView): > 0 = repeat on the whole target
    uint64_t grid_cells;
    uint32_t ticks[4];
    double batch;           // sharded device loop: seq * 65536 + 2 * (batches finished) + done, written by the last controller step of a batch
    double progress;        // number of passes consumed so far (written after every pass: the host keeps the queue ahead of it)
    double seq;             // == the call's sequence number once the loop has finished
};

namespace ndt_opt {

NDT_HD inline double min_std(double a, double b) { return (b < a) ? b : a; }      // std::min / std::max, including what they do with NaN
NDT_HD inline double max_std(double a, double b) { return (a < b) ? b : a; }

// sinf / cosf the way this image's glibc (2.35) evaluates them on a host with FMA (sysdeps/ieee754/flt-32/s_sincosf.h, the "ARM optimized
// routines" algorithm: reduction by pi/2 in double, a degree-7 sine or degree-6 cosine polynomial, rounding to float). glibc compiles that
// file with contraction for its FMA ifunc variant, so the multiply-adds below are explicit fma()s (the library is built with
// -ffp-contract=off); in this form the routine reproduced glibc's sinf and cosf bit for bit on 100 million arguments over [-96, 96], and
// differed in 1e-7 of them from the variant glibc selects on a host WITHOUT FMA. The result is not the correctly rounded value: rounding
// sin((double)x) to float differs from it for 0.65 % of the arguments, which is why that shortcut is not used.
// which = 0: sine, 1: cosine. |x| >= 120 (never an angle of a pose that converges) falls back to the double routine.
NDT_HD inline float glibc_sincosf(float y, int which) {
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    uint32_t bits;
    memcpy(&bits, &y, 4);
    const uint32_t top = (bits >> 20) & 0x7ff;
    double x = (double)y;
    int n = which;
    double flip = 1.0;       // -1 selects the negated polynomial table glibc switches to when n & 2
    if (top < 0x3f4u) {      // |y| < pi/4
        if (top < 0x398u) return which ? 1.0f : y;      // |y| < 2^-12
    } else if (top < 0x42fu) {                          // |y| < 120
        const double r = x * hpi_inv;
        const int q = ((int32_t)r + 0x800000) >> 24;
        x = fma(-(double)q, hpi, x);
        if (q & 2) flip = -1.0;
        n = q ^ which;
        if (!(n & 1)) x *= ((q & 3) == 1 || (q & 3) == 2) ? -1.0 : 1.0;   // sign[q & 3] = {1, -1, -1, 1}, applied to the sine's argument
    } else {
        return (float)(which ? cos((double)y) : sin((double)y));
    }
    const double x2 = x * x;
    if (!(n & 1)) {
        const double x3 = x * x2;
        const double t1 = fma(x2, s3, s2);
        const double x7 = x3 * x2;
        const double t = fma(x3, s1, x);
        return (float)fma(x7, t1, t);
    }
    const double x4 = x2 * x2;
    const double k2 = fma(x2, flip * c4, flip * c3);
    const double k1 = fma(x2, flip * c1, flip * c0);
    const double x6 = x4 * x2;
    const double k = fma(x4, flip * c2, k1);
    return (float)fma(x6, k2, k);
}

// The six sine/cosine pairs a request needs, so that the device can evaluate them in six lanes at once:
//   k = 0..2  the float angles of the pose, Eigen::AngleAxisf(float(x[3+k])) -> sinf / cosf
//   k = 3..5  the double angles of computeAngleDerivatives (ndt_omp_impl.hpp:289-312), with its small-angle shortcut
// sc[2k] = sine, sc[2k+1] = cosine.
NDT_HD inline void trig_pair(const double x[6], int k, double sc[2]) {
    if (k < 3) {
        const float a = (float)x[3 + k];
        sc[0] = (double)glibc_sincosf(a, 0); sc[1] = (double)glibc_sincosf(a, 1);
    } else {
        const double a = x[k];
        if (fabs(a) < 10e-5) { sc[0] = 0.0; sc[1] = 1.0; } else { sc[0] = sin(a); sc[1] = cos(a); }
    }
}

// Eigen::AngleAxisf(angle, Unit{X,Y,Z}).toRotationMatrix(), row-major, from the angle's sine and cosine
NDT_HD inline void angle_axis(float s, float c, int axis, float R[9]) {
    float ax[3] = {0, 0, 0};
    ax[axis] = 1.0f;
    const float sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
    const float c1[3] = {(1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2]};
    float tmp;
    tmp = c1[0] * ax[1]; R[1] = tmp - sa[2]; R[3] = tmp + sa[2];
    tmp = c1[0] * ax[2]; R[2] = tmp + sa[1]; R[6] = tmp - sa[1];
    tmp = c1[1] * ax[2]; R[5] = tmp - sa[0]; R[7] = tmp + sa[0];
#pragma unroll
    for (int d = 0; d < 3; ++d) R[d * 3 + d] = c1[d] * ax[d] + c;
}
NDT_HD inline void mul33(const float A[9], const float B[9], float C[9]) {
    float o[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) { float s = A[r * 3] * B[c]; s += A[r * 3 + 1] * B[3 + c]; s += A[r * 3 + 2] * B[6 + c]; o[r * 3 + c] = s; }
#pragma unroll
    for (int i = 0; i < 9; ++i) C[i] = o[i];
}
// Translation(x[0:3]) * Rx * Ry * Rz evaluated in float (ndt_omp_impl.hpp:146-149,827-830)
NDT_HD_FLAT void pose_from_trig(const double x[6], const double sc[12], NdtPose* T) {
    float Rx[9], Ry[9], Rz[9], M[9];
    angle_axis((float)sc[0], (float)sc[1], 0, Rx); angle_axis((float)sc[2], (float)sc[3], 1, Ry); angle_axis((float)sc[4], (float)sc[5], 2, Rz);
    mul33(Rx, Ry, M); mul33(M, Rz, T->R);
    T->t[0] = (float)x[0]; T->t[1] = (float)x[1]; T->t[2] = (float)x[2];
}
// computeAngleDerivatives (ndt_omp_impl.hpp:289-395)
NDT_HD_FLAT void angle_tables_from_trig(const double sc[12], NdtAngles* a, int part = -1) {
    const double sx = sc[6], cx = sc[7], sy = sc[8], cy = sc[9], sz = sc[10], cz = sc[11];
    const double J[8][3] = {
        {(-sx * sz + cx * sy * cz), (-sx * cz - cx * sy * sz), (-cx * cy)}, {(cx * sz + sx * sy * cz), (cx * cz - sx * sy * sz), (-sx * cy)},
        {(-sy * cz), sy * sz, cy}, {sx * cy * cz, (-sx * cy * sz), sx * sy}, {(-cx * cy * cz), cx * cy * sz, (-cx * sy)},
        {(-cy * sz), (-cy * cz), 0}, {(cx * cz - sx * sy * sz), (-cx * sz - sx * sy * cz), 0}, {(sx * cz + cx * sy * sz), (cx * sy * cz - sx * sz), 0}};
    const double Hh[15][3] = {
        {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), sx * cy}, {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), (-cx * cy)},
        {(cx * cy * cz), (-cx * cy * sz), (cx * sy)}, {(sx * cy * cz), (-sx * cy * sz), (sx * sy)},
        {(-sx * cz - cx * sy * sz), (sx * sz - cx * sy * cz), 0}, {(cx * cz - sx * sy * sz), (-sx * sy * cz - cx * sz), 0},
        {(-cy * cz), (cy * sz), (-sy)}, {(-sx * sy * cz), (sx * sy * sz), (sx * cy)}, {(cx * sy * cz), (-cx * sy * sz), (-cx * cy)},
        {(sy * sz), (sy * cz), 0}, {(-sx * cy * sz), (-sx * cy * cz), 0}, {(cx * cy * sz), (cx * cy * cz), 0},
        {(-cy * cz), (cy * sz), 0}, {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), 0}, {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), 0}};
    // (fully unrolled: indexed with a loop variable the two tables would live in scratch memory on the device)
    // `part` lets the device spread the rows over four waves (0: j, 1..3: five rows of h each); -1 = everything
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (part >= 1) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) { a->jd[r][c] = J[r][c]; a->j[r][c] = (float)J[r][c]; }
    }
#pragma unroll
    for (int r = 0; r < 15; ++r) {
        if (part >= 0 && part != 1 + r / 5) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) { a->hd[r][c] = Hh[r][c]; a->h[r][c] = (float)Hh[r][c]; }
    }
    if (part < 0 || part == 2) a->h[6][2] = (float)(sy);   // the float table (:384) writes (sy) where h_ang_d1_ (:362) has (-sy)
}
NDT_HD inline void pose_from_p(const double x[6], NdtPose* T) {
    double sc[12];
    for (int k = 0; k < 6; ++k) trig_pair(x, k, sc + 2 * k);
    pose_from_trig(x, sc, T);
}
NDT_HD inline void angle_derivatives(const double p[6], NdtAngles* a) {
    double sc[12];
    for (int k = 0; k < 6; ++k) trig_pair(p, k, sc + 2 * k);
    angle_tables_from_trig(sc, a);
}

// JacobiSVD<Matrix6d>::solve restated: one-sided Jacobi, pseudo-inverse with Eigen's default threshold
NDT_HD inline void svd6_solve(const double A_in[36], const double b[6], double x[6]) {
    double U[6][6], V[6][6];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { U[i][j] = A_in[i * 6 + j]; V[i][j] = i == j; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 5; ++p)
            for (int q = p + 1; q < 6; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int k = 0; k < 6; ++k) { alpha += U[k][p] * U[k][p]; beta += U[k][q] * U[k][q]; gamma += U[k][p] * U[k][q]; }
                if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 1e-17 * sqrt(alpha * beta)) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < 6; ++k) {
                    const double up = U[k][p], uq = U[k][q]; U[k][p] = c * up - s * uq; U[k][q] = s * up + c * uq;
                    const double vp = V[k][p], vq = V[k][q]; V[k][p] = c * vp - s * vq; V[k][q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double sig[6], smax = 0;
    for (int j = 0; j < 6; ++j) { double s = 0; for (int k = 0; k < 6; ++k) s += U[k][j] * U[k][j]; sig[j] = sqrt(s); smax = sig[j] > smax ? sig[j] : smax; }
    const double thr = 6.0 * 2.220446049250313e-16 * smax;
    for (int i = 0; i < 6; ++i) x[i] = 0;
    for (int j = 0; j < 6; ++j) {
        if (!(sig[j] > thr)) continue;
        double ub = 0;
        for (int k = 0; k < 6; ++k) ub += (U[k][j] / sig[j]) * b[k];
        for (int i = 0; i < 6; ++i) x[i] += V[i][j] * (ub / sig[j]);
    }
}

// Device flavour of the Newton solve: Gaussian elimination with partial pivoting.  For a matrix of full numerical rank it and the
// Jacobi SVD both give A^-1 b to cond(A) * eps.  When a pivot falls below 1e-9 of the largest entry -- or anything is not finite --
// the SVD's pseudo-inverse may drop a direction that elimination keeps: returns false and the caller hands the whole alignment to
// the host path.  Every index below is a compile-time constant once the loops are unrolled (the row exchange is a chain of
// conditional swaps, of which at most one fires): the 6x7 system then lives in registers.  Indexed by the pivot row it lived in
// scratch memory and one solve took ~20 us on the device's single lane, a quarter of a whole NDT iteration's controller time.
NDT_HD_FLAT bool lu6_solve_guarded(const double A_in[36], const double b[6], double x[6]) {
    double M[6][7];
    double amax = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j < 6; ++j) { M[i][j] = A_in[i * 6 + j]; const double a = fabs(M[i][j]); amax = a > amax ? a : amax; }
        M[i][6] = b[i];
    }
    if (!(amax > 0) || !(amax < 1e300)) return false;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int piv = k;
        double best = fabs(M[k][k]);
#pragma unroll
        for (int i = k + 1; i < 6; ++i) { const double a = fabs(M[i][k]); if (a > best) { best = a; piv = i; } }
        if (!(best > 1e-9 * amax)) return false;
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            const bool sw = piv == i;
#pragma unroll
            for (int j = k; j < 7; ++j) { const double u = M[k][j], v = M[i][j]; M[k][j] = sw ? v : u; M[i][j] = sw ? u : v; }
        }
        const double inv = 1.0 / M[k][k];
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            const double f = M[i][k] * inv;
#pragma unroll
            for (int j = k + 1; j < 7; ++j) M[i][j] -= f * M[k][j];
        }
    }
    double y[6];
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = M[i][6];
#pragma unroll
        for (int j = i + 1; j < 6; ++j) s -= M[i][j] * y[j];
        y[i] = s / M[i][i];
    }
    bool finite = true;
#pragma unroll
    for (int i = 0; i < 6; ++i) { x[i] = y[i]; if (!(fabs(y[i]) < 1e300)) finite = false; }
    return finite;
}

// updateIntervalMT (:713-733)
NDT_HD_FLAT bool update_interval(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t, double f_t, double g_t) {
    if (f_t > f_l) { a_u = a_t; f_u = f_t; g_u = g_t; return false; }
    else if (g_t * (a_l - a_t) > 0) { a_l = a_t; f_l = f_t; g_l = g_t; return false; }
    else if (g_t * (a_l - a_t) < 0) { a_u = a_l; f_u = f_l; g_u = g_l; a_l = a_t; f_l = f_t; g_l = g_t; return false; }
    return true;
}
// trialValueSelectionMT (:649-711)
NDT_HD_FLAT double trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {
    if (f_t > f_l) {
        const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
        const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
        return fabs(a_c - a_l) < fabs(a_q - a_l) ? a_c : 0.5 * (a_q + a_c);
    } else if (g_t * g_l < 0) {
        const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
        const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        return fabs(a_c - a_t) >= fabs(a_s - a_t) ? a_c : a_s;
    } else if (fabs(g_t) <= fabs(g_l)) {
        const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
        const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        const double nx = fabs(a_c - a_t) < fabs(a_s - a_t) ? a_c : a_s;
        const double lim = a_t + 0.66 * (a_u - a_t);
        return a_t > a_l ? min_std(lim, nx) : max_std(lim, nx);
    }
    const double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u, w = sqrt(z * z - g_t * g_u);
    return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}

// ---- the state machine -------------------------------------------------------------------------------------------------
// Start: the first pass evaluates score, gradient and Hessian at the initial guess (computeTransformation :113-118).
NDT_HD inline void ctl_init(NdtCtl* c, const NdtPose& T0, const double p0[6], double step_size, double trans_eps, int max_iters) {
    c->T = T0; c->final_T = T0;
    for (int i = 0; i < 6; ++i) { c->p[i] = p0[i]; c->dir[i] = 0; c->x_t[i] = p0[i]; c->grad[i] = 0; }
    for (int i = 0; i < 36; ++i) c->hess[i] = 0;
    angle_derivatives(p0, &c->ang);
    c->kind = kNdtPassDerivH; c->phase = kNdtPhaseInit;
    c->step_size = step_size; c->trans_eps = trans_eps; c->max_iters = max_iters; c->late_h = 0;
    c->score = 0; c->phi_0 = c->d_phi_0 = c->a_l = c->f_l = c->g_l = c->a_u = c->f_u = c->g_u = c->a_t = 0;
    c->step_min = trans_eps / 2; c->step_max = step_size;
    c->open_interval = 1; c->interval_converged = 0; c->it = 0; c->nr_it = 0; c->conv = 0; c->n_deriv = c->n_hess = 0; c->bail = 0;
    c->done = 0; c->passes = 0;
    for (int i = 0; i < 4; ++i) c->ticks[i] = 0;
    for (int i = 0; i < 6; ++i) c->x_eval[i] = 0;
    c->eval_valid = 0; c->replayed = 0; c->replay_off = 0; c->need_h = 0;
}

// the pose and the angle tables of the next pass from the six sine/cosine pairs of x_t (trig_pair)
NDT_HD_FLAT void ctl_tables(NdtCtl* c, const double sc[12]) {
    pose_from_trig(c->x_t, sc, &c->T);
    angle_tables_from_trig(sc, &c->ang);
}

// One step of the loops with the sums of the last request in c->score / grad / hess: leaves the next request in *c (or done).
// Returns true when that request is at a NEW point x_t.
NDT_HD_FLAT bool ctl_advance(NdtCtl* c) {
    const double mu = 1.e-4, nu = 0.9;
    const int max_it = 10;
    bool line_search_over = false;
    if (c->phase == kNdtPhaseLsFirst || c->phase == kNdtPhaseLsLoop) {
        // computeStepLengthMT after updateDerivatives at x_t (:832-852 the first time, :880-926 inside the loop)
        const double phi_t = -c->score;
        double d_phi_t = 0;
        for (int i = 0; i < 6; ++i) d_phi_t += c->grad[i] * c->dir[i];
        d_phi_t = -d_phi_t;
        const double psi_t = phi_t - c->phi_0 - mu * c->d_phi_0 * c->a_t, d_psi_t = d_phi_t - mu * c->d_phi_0;
        if (c->phase == kNdtPhaseLsLoop) {
            if (c->open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
                c->open_interval = 0;
                c->f_l = c->f_l + c->phi_0 - mu * c->d_phi_0 * c->a_l; c->g_l = c->g_l + mu * c->d_phi_0;
                c->f_u = c->f_u + c->phi_0 - mu * c->d_phi_0 * c->a_u; c->g_u = c->g_u + mu * c->d_phi_0;
            }
            if (c->open_interval) c->interval_converged = update_interval(c->a_l, c->f_l, c->g_l, c->a_u, c->f_u, c->g_u, c->a_t, psi_t, d_psi_t) ? 1 : 0;
            else c->interval_converged = update_interval(c->a_l, c->f_l, c->g_l, c->a_u, c->f_u, c->g_u, c->a_t, phi_t, d_phi_t) ? 1 : 0;
            c->it += 1;
        }
        if (!c->interval_converged && c->it < max_it && !(psi_t <= 0 && d_phi_t <= -nu * c->d_phi_0)) {
            if (c->open_interval) c->a_t = trial_value(c->a_l, c->f_l, c->g_l, c->a_u, c->f_u, c->g_u, c->a_t, psi_t, d_psi_t);
            else c->a_t = trial_value(c->a_l, c->f_l, c->g_l, c->a_u, c->f_u, c->g_u, c->a_t, phi_t, d_phi_t);
            c->a_t = min_std(c->a_t, c->step_max);
            c->a_t = max_std(c->a_t, c->step_min);
            for (int i = 0; i < 6; ++i) c->x_t[i] = c->p[i] + c->dir[i] * c->a_t;
            c->kind = kNdtPassDeriv; c->phase = kNdtPhaseLsLoop;
            return true;
        }
        if (c->it) {      // computeHessian (:928-929): double precision, with the pose and the tables of the last pass
            c->kind = kNdtPassHessian; c->phase = kNdtPhaseLsHess;
            return false;
        }
        // The search ended at its first trial point: the float Hessian computeDerivatives(.., true) left there (:832) is the one the
        // next Newton step uses.  That first evaluation was asked for WITHOUT the Hessian (below): ask for it now, at the same point.
        // -- but only if another Newton step follows: first the convergence test of the loop below (need_h).
        if (c->kind == kNdtPassDeriv) c->need_h = 1;
        line_search_over = true;
    } else if (c->phase == kNdtPhaseLsHess) {
        line_search_over = true;
    }      // (kNdtPhaseLsFirstH: the late Hessian has arrived; the step it follows was applied before it was asked for)
    // ---- computeTransformation's loop (:120-160); passes that need no evaluation are walked through right here ----
    for (;;) {
        if (line_search_over) {
            const double nrm = c->a_t;
            c->final_T = c->T;
            for (int i = 0; i < 6; ++i) c->p[i] += c->dir[i] * nrm;
            if (c->nr_it > c->max_iters || (c->nr_it && fabs(nrm) < c->trans_eps)) c->conv = 1;
            c->nr_it += 1;
            if (c->conv) { c->done = 1; c->kind = kNdtPassNone; return false; }
            if (c->need_h) {      // the float Hessian at the point the search ended at (= p now; pose and tables are still the ones of that pass)
                c->need_h = 0;
                c->kind = kNdtPassDerivH; c->phase = kNdtPhaseLsFirstH;
                return false;
            }
        }
        // Newton direction (:124-135)
        double rhs[6], dp[6];
        for (int i = 0; i < 6; ++i) rhs[i] = -c->grad[i];
#if defined(__HIP_DEVICE_COMPILE__)
        if (!lu6_solve_guarded(c->hess, rhs, dp)) { c->bail = 1; c->done = 1; c->kind = kNdtPassNone; return false; }
#else
        svd6_solve(c->hess, rhs, dp);
#endif
        double nrm = 0;
        for (int i = 0; i < 6; ++i) nrm += dp[i] * dp[i];
        nrm = sqrt(nrm);
        if (nrm == 0 || nrm != nrm) { c->conv = nrm == nrm ? 1 : 0; c->done = 1; c->kind = kNdtPassNone; return false; }
        for (int i = 0; i < 6; ++i) c->dir[i] = dp[i] / nrm;
        // computeStepLengthMT up to its first evaluation (:735-830)
        c->phi_0 = -c->score;
        double d0 = 0;
        for (int i = 0; i < 6; ++i) d0 += c->grad[i] * c->dir[i];
        d0 = -d0;
        if (d0 >= 0) {
            if (d0 == 0) { c->a_t = 0; line_search_over = true; continue; }      // "not a descent direction": step 0, T untouched
            d0 *= -1;
            for (int i = 0; i < 6; ++i) c->dir[i] *= -1;
        }
        c->d_phi_0 = d0;
        c->it = 0;
        c->a_l = 0; c->a_u = 0;
        c->f_l = c->phi_0 - c->phi_0 - mu * d0 * c->a_l; c->g_l = d0 - mu * d0;
        c->f_u = c->phi_0 - c->phi_0 - mu * d0 * c->a_u; c->g_u = d0 - mu * d0;
        c->step_max = c->step_size; c->step_min = c->trans_eps / 2;
        c->interval_converged = (c->step_max - c->step_min) < 0 ? 1 : 0;
        c->open_interval = 1;
        c->a_t = min_std(nrm, c->step_max);
        c->a_t = max_std(c->a_t, c->step_min);
        for (int i = 0; i < 6; ++i) c->x_t[i] = c->p[i] + c->dir[i] * c->a_t;
        // The reference evaluates the first trial point WITH the float Hessian (computeDerivatives(.., true), :832) -- and throws that
        // Hessian away whenever the search goes on to a second point: every further computeDerivatives zeroes it (:183) and
        // computeHessian replaces it (:928).  A pass with the Hessian costs twice a pass without (ndt.hip: 35 vs 18 us at 131 072
        // points), and on the reference's constants nearly every search goes on.  So the first trial is asked for without it, and the
        // rare search that ends right there gets its Hessian by a pass of its own (kNdtPhaseLsFirstH): same numbers either way -- score
        // and gradient do not depend on whether the Hessian is accumulated beside them.  (replay_off = the reference's schedule.)
        c->kind = c->replay_off ? kNdtPassDerivH : kNdtPassDeriv; c->phase = kNdtPhaseLsFirst;
        return true;
    }
}

// `sums` = score, gradient[6], Hessian[36] of the pass that ctl->kind asked for.  Leaves the next request in *c (or done).
// Returns true when the next pass evaluates at a point whose pose and tables are not in *c yet: the caller then owes ctl_tables()
// (the trigonometry is kept out of this function so that the device can spread it over lanes).
//
// A request for score + gradient at the point they were last evaluated at is not passed on: the pass would return, bit for bit, the
// sums it returned before (same pose, same tables, same fixed-order sums), so they are fed again on the spot.  That is not a corner
// case -- pclomp clamps every trial step into [epsilon / 2, step_size] (ndt_omp_impl.hpp:821-823, 905-907), and once the search
// wants a longer step than step_size every further trial is that same clamped step: with the reference's constants a More-Thuente
// search that runs to its cap evaluates ONE point nine times over.  The reference recomputes it each time; n_deriv counts what the
// reference evaluates, `replayed` how many of those were answered from here.
// Repeated evaluations, fast: after ctl_decide has answered ONE request from the sums it holds (same point, same sums), the next More-Thuente
// updates see the same trial value again and again -- phi_t, d_phi_t, psi_t and d_psi_t are constants, only the interval (a_l .. g_u) and the
// iteration count move -- until the search ends or asks for another point.  This runs those updates on LOCAL copies of the dozen scalars involved
// (on the device the state lives in LDS, and a step through ctl_advance is a chain of LDS round trips: up to nine such steps in one pass were what
// one lane spent most of its 5.8 us on).  Each update is first SIMULATED -- the statements of ctl_advance's line-search branch, word for word -- and
// committed only if it ends in yet another repeat; the update that ends the run is discarded here and taken by ctl_advance itself, so whatever
// happens next has one source: ctl_advance.  Host and device run the same code.
NDT_HD_FLAT void ctl_replay_run(NdtCtl* c) {
    if (c->phase != kNdtPhaseLsLoop || c->kind != kNdtPassDeriv) return;
    const double mu = 1.e-4, nu = 0.9;
    const int max_it = 10;
    const double phi_t = -c->score;
    double d_phi_t = 0;
    for (int i = 0; i < 6; ++i) d_phi_t += c->grad[i] * c->dir[i];
    d_phi_t = -d_phi_t;
    const double phi_0 = c->phi_0, d_phi_0 = c->d_phi_0, step_min = c->step_min, step_max = c->step_max;
    double a_l = c->a_l, f_l = c->f_l, g_l = c->g_l, a_u = c->a_u, f_u = c->f_u, g_u = c->g_u, a_t = c->a_t;
    int it = c->it, open_interval = c->open_interval, n = 0;
    double xt_new[6];
    for (;;) {
        const double psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t, d_psi_t = d_phi_t - mu * d_phi_0;
        double al = a_l, fl = f_l, gl = g_l, au = a_u, fu = f_u, gu = g_u;
        int oi = open_interval;
        if (oi && (psi_t <= 0 && d_psi_t >= 0)) {
            oi = 0;
            fl = fl + phi_0 - mu * d_phi_0 * al; gl = gl + mu * d_phi_0;
            fu = fu + phi_0 - mu * d_phi_0 * au; gu = gu + mu * d_phi_0;
        }
        const bool ic = oi ? update_interval(al, fl, gl, au, fu, gu, a_t, psi_t, d_psi_t) : update_interval(al, fl, gl, au, fu, gu, a_t, phi_t, d_phi_t);
        const int it2 = it + 1;
        if (ic || !(it2 < max_it) || (psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) break;      // the search ends: ctl_advance's turn
        double at2 = oi ? trial_value(al, fl, gl, au, fu, gu, a_t, psi_t, d_psi_t) : trial_value(al, fl, gl, au, fu, gu, a_t, phi_t, d_phi_t);
        at2 = min_std(at2, step_max);
        at2 = max_std(at2, step_min);
        bool same = true;
        for (int i = 0; i < 6; ++i) { xt_new[i] = c->p[i] + c->dir[i] * at2; same = same && xt_new[i] == c->x_eval[i]; }
        if (!same) break;                                                                   // another point: ctl_advance's turn
        a_l = al; f_l = fl; g_l = gl; a_u = au; f_u = fu; g_u = gu; a_t = at2; it = it2; open_interval = oi;
        ++n;
    }
    if (!n) return;
    c->a_l = a_l; c->f_l = f_l; c->g_l = g_l; c->a_u = a_u; c->f_u = f_u; c->g_u = g_u; c->a_t = a_t;
    c->it = it; c->open_interval = open_interval; c->interval_converged = 0;
    for (int i = 0; i < 6; ++i) c->x_t[i] = c->p[i] + c->dir[i] * a_t;      // (what the last committed update stored: the values compare equal to x_eval; the bits are these)
    c->n_deriv += n; c->replayed += n;
}

// What a pass's sums do to entry i of the Hessian the state machine keeps (kind / phase: of the request the pass answered).  Apart from ctl_decide so
// that the device can let 36 lanes take the entries in while one lane decides (hess_taken below).
NDT_HD_FLAT double ctl_hess_entry(int kind, int phase, const double sums[43], int i) {
    if (kind == kNdtPassHessian || phase == kNdtPhaseLsFirstH || kind == kNdtPassDerivH) return sums[7 + i];
    return 0.0;      // (computeDerivatives zeroes it either way, :183)
}
NDT_HD_FLAT bool ctl_decide(NdtCtl* c, const double sums[43], bool hess_taken = false) {
    if (c->done) return false;
    c->passes += 1;
    // ---- take the sums in ----
    if (!hess_taken) { const int k0 = c->kind, p0 = c->phase; for (int i = 0; i < 36; ++i) c->hess[i] = ctl_hess_entry(k0, p0, sums, i); }
    bool hess_zero = false;      // the kept Hessian holds zeros: a repeated evaluation, which zeroes it again, has nothing to store (36 stores per repeat, up to nine repeats)
    if (c->kind == kNdtPassHessian) {
        c->n_hess += 1;
    } else if (c->phase == kNdtPhaseLsFirstH) {
        c->late_h += 1;      // (not an evaluation of the reference's: it computed this Hessian with the first trial)
    } else {
        c->n_deriv += 1;
        c->score = sums[0];
        hess_zero = c->kind != kNdtPassDerivH;
        for (int i = 0; i < 6; ++i) c->grad[i] = sums[1 + i];
        for (int i = 0; i < 6; ++i) c->x_eval[i] = c->x_t[i];
        c->eval_valid = 1;
    }
    bool moved = false;
    for (;;) {
        if (ctl_advance(c)) moved = true;
        if (c->done) return false;
        bool same = c->kind == kNdtPassDeriv && c->eval_valid != 0 && c->replay_off == 0;
        for (int i = 0; i < 6; ++i) same = same && c->x_t[i] == c->x_eval[i];
        if (!same) return moved;
        // (x_t == x_eval compares values: +0 / -0 would pass as equal and give the same pose; NaN never passes)
        c->n_deriv += 1; c->replayed += 1;
        if (!hess_zero) { for (int i = 0; i < 36; ++i) c->hess[i] = 0.0; hess_zero = true; }
        ctl_replay_run(c);      // (the repeats that follow, on local copies; whatever ends them is left to ctl_advance)
    }
}

// the whole step on one thread (host loop)
NDT_HD inline void ctl_step(NdtCtl* c, const double sums[43]) {
    if (ctl_decide(c, sums)) {
        double sc[12];
        for (int k = 0; k < 6; ++k) trig_pair(c->x_t, k, sc + 2 * k);
        ctl_tables(c, sc);
    }
}

}  // namespace ndt_opt
}  // namespace pcr
