// VGICP's optimiser as a state machine the host and the device both run: fast_gicp::LsqRegistration::computeTransformation
// and step_lm (lsq_registration_impl.hpp:53-79, 125-171), is_converged (:82-91), so3_exp (so3/so3.hpp:58-77).
//
// Every decision of the Levenberg-Marquardt loop depends on 29 numbers summed over the scan: H (21, upper triangle), b (6)
// and the error at the linearisation point (linearize, fast_vgicp_impl.hpp:119-180), and the error at the trial pose
// (compute_error, :183-204).  A PASS evaluates them at one pose: the first at the initial guess (linearisation only), every
// later one at an LM trial pose -- the error on the correspondences of the last accepted linearisation and, speculatively, the
// linearisation AT the trial pose, which is the next linearisation point whenever the trial is accepted (vgicp.hip).  Between two
// passes stands one call of vg_ctl_step: it takes the sums in and leaves the pose of the next pass in VgCtl, or `done`.
//
// On the device the step is the prologue of the next pass's launch (vgicp_pass_pro_kernel): no host round trip per pass.  The
// host-driven loop of capi.hip (sharded targets, pcr_params.host_optimiser) is the same arithmetic written as the reference's loops.
#pragma once
#include <math.h>
#include <stdint.h>

#include "pcr_internal.h"
#include "small_math.h"

namespace pcr {

#if defined(__HIPCC__)
#define VG_HD __host__ __device__ __forceinline__
#else
#define VG_HD inline
#endif

enum : int { kVgPassLinearize = 0, kVgPassTrial = 1 };

struct VgCtl {
    Pose16 x0;               // the linearisation point (x0 of computeTransformation)
    Pose16 xi;               // the pose the next pass evaluates: x0 itself for the first pass, delta * x0 of an LM trial afterwards
    double lin[28];          // H upper triangle (21), b (6), error y0 at x0
    double d[6], D[16];      // the trial's step and its delta matrix
    double lambda, nu;
    double lm_init_scale, rot_eps, trans_eps;
    int32_t max_iters, lm_inner;
    int32_t kind, it, inner, parity;      // parity: which of the two correspondence buffers belongs to x0
    int32_t conv, done, passes, n_lin, n_err, outer;
    uint32_t ticks[2];
};

// what the device-resident loop reports (host-mapped memory; `seq` is written last)
struct VgOut {
    Pose16 x0;
    int32_t conv, outer, n_lin, n_err, passes;
    int32_t roi_escapes;     // lookups that hit an occupied voxel outside the region the target was prepared for (RoiView): > 0 = repeat on the whole target
    double progress;         // seq * kProgressWindow + passes consumed so far
    double seq;
};

namespace vg_opt {

// so3_exp (so3.hpp:58-77) -> Quaterniond::toRotationMatrix; translation d[3:6]
VG_HD void make_delta(const double d[6], double D[16]) {
    const double th2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    double imag, real;
    if (th2 < 1e-10) {
        const double q4 = th2 * th2;
        imag = 0.5 - 1.0 / 48.0 * th2 + 1.0 / 3840.0 * q4;
        real = 1.0 - 1.0 / 8.0 * th2 + 1.0 / 384.0 * q4;
    } else {
        const double th = sqrt(th2), hf = 0.5 * th;
        imag = sin(hf) / th; real = cos(hf);
    }
    const double w = real, x = imag * d[0], y = imag * d[1], z = imag * d[2];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x,
                 tyy = ty * y, tyz = tz * y, tzz = tz * z;
#pragma unroll
    for (int i = 0; i < 16; ++i) D[i] = 0;
    D[0] = 1 - (tyy + tzz); D[4] = txy - twz; D[8] = txz + twy;
    D[1] = txy + twz; D[5] = 1 - (txx + tzz); D[9] = tyz - twx;
    D[2] = txz - twy; D[6] = tyz + twx; D[10] = 1 - (txx + tyy);
    D[12] = d[3]; D[13] = d[4]; D[14] = d[5]; D[15] = 1;
}

// Isometry3d product, column-major
VG_HD void mul44(const double A[16], const double B[16], double C[16]) {
    double o[16];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += A[k * 4 + r] * B[c * 4 + k];
            o[c * 4 + r] = s;
        }
    o[3] = o[7] = o[11] = 0; o[15] = 1;
#pragma unroll
    for (int i = 0; i < 16; ++i) C[i] = o[i];
}

// is_converged (lsq_registration_impl.hpp:82-91)
VG_HD bool is_converged(const double D[16], double rot_eps, double trans_eps) {
    double rmax = 0, tmax = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) { const double v = fabs(D[c * 4 + r] - (r == c ? 1.0 : 0.0)) * (1.0 / rot_eps); rmax = rmax < v ? v : rmax; }
#pragma unroll
    for (int r = 0; r < 3; ++r) { const double v = fabs(D[12 + r]) * (1.0 / trans_eps); tmax = tmax < v ? v : tmax; }
    return (rmax < tmax ? tmax : rmax) < 1;
}

VG_HD void ctl_init(VgCtl* c, const Pose16& guess, int max_iters, int lm_inner, double lm_init_scale, double rot_eps, double trans_eps) {
    c->x0 = guess; c->xi = guess;
    for (int i = 0; i < 28; ++i) c->lin[i] = 0;
    for (int i = 0; i < 6; ++i) c->d[i] = 0;
    for (int i = 0; i < 16; ++i) c->D[i] = 0;
    c->lambda = -1.0; c->nu = 2.0;
    c->lm_init_scale = lm_init_scale; c->rot_eps = rot_eps; c->trans_eps = trans_eps;
    c->max_iters = max_iters; c->lm_inner = lm_inner;
    c->kind = kVgPassLinearize; c->it = 0; c->inner = 0; c->parity = 0;
    c->conv = 0; c->done = max_iters <= 0 ? 1 : 0; c->passes = 0; c->n_lin = c->n_err = 0; c->outer = 0;
    c->ticks[0] = c->ticks[1] = 0;
}

// one LM trial from the linearisation in c->lin: (H + lambda I) d = -b, delta = [so3_exp(d[0:3]); d[3:6]], xi = delta * x0   (:130-140)
VG_HD void next_trial(VgCtl* c) {
    double A[36], rhs[6];
    int q = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int cc = r; cc < 6; ++cc) { A[r * 6 + cc] = A[cc * 6 + r] = c->lin[q++]; }
#pragma unroll
    for (int k = 0; k < 6; ++k) { A[k * 7] += c->lambda; rhs[k] = -c->lin[21 + k]; }
    double d[6];
    ldlt6_solve(A, rhs, d);
#pragma unroll
    for (int k = 0; k < 6; ++k) c->d[k] = d[k];
    double D[16];
    make_delta(d, D);
#pragma unroll
    for (int k = 0; k < 16; ++k) c->D[k] = D[k];
    mul44(D, c->x0.m, c->xi.m);
    c->kind = kVgPassTrial;
}

// the head of one iteration of computeTransformation's loop (:60-72), with the linearisation already in c->lin
VG_HD void begin_outer(VgCtl* c) {
    c->outer = c->it + 1;
    if (c->lambda < 0.0) {      // (:126-128)
        double mx = 0;
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const double v = fabs(c->lin[q]);
            mx = mx < v ? v : mx;
            q += 6 - r;
        }
        c->lambda = c->lm_init_scale * mx;
    }
    c->nu = 2.0;
    c->inner = 0;
    next_trial(c);
}

// `sums` = H (21), b (6), error at the pose of the pass, [28] = error of the trial pose on the old correspondences.
VG_HD void ctl_step(VgCtl* c, const double sums[29]) {
    if (c->done) return;
    c->passes += 1;
    if (c->kind == kVgPassLinearize) {
        c->n_lin += 1;
#pragma unroll
        for (int k = 0; k < 28; ++k) c->lin[k] = sums[k];
        begin_outer(c);
        return;
    }
    c->n_err += 1;
    const double y0 = c->lin[27], yi = sums[28];
    double den = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) den += c->d[k] * (c->lambda * c->d[k] - c->lin[21 + k]);
    const double rho = (y0 - yi) / den;
    if (rho < 0) {                                         // (:146-154)
        if (is_converged(c->D, c->rot_eps, c->trans_eps)) { c->conv = 1; c->done = 1; return; }
        c->lambda = c->nu * c->lambda; c->nu = 2 * c->nu;
        c->inner += 1;
        if (c->inner >= c->lm_inner) { c->done = 1; return; }      // "lm not converged!!": the loop ends where it stands
        next_trial(c);
        return;
    }
    // accepted (:156-162): the trial pose is the next linearisation point, and the pass linearised there already
    c->x0 = c->xi;
#pragma unroll
    for (int k = 0; k < 28; ++k) c->lin[k] = sums[k];
    c->parity ^= 1;
    const double f = 1 - pow(2 * rho - 1, 3);
    c->lambda = c->lambda * (1.0 / 3.0 < f ? f : 1.0 / 3.0);      // std::max(1/3, f)
    c->conv = is_converged(c->D, c->rot_eps, c->trans_eps) ? 1 : 0;
    c->it += 1;
    if (c->conv || c->it >= c->max_iters) { c->done = 1; return; }
    begin_outer(c);
}

}  // namespace vg_opt
}  // namespace pcr
