// ndt.hip -- NDT scan-to-map (pclomp::NormalDistributionsTransform, DIRECT7) on gfx950.
//
// Kernels for the reference's PCR::NdtRegister::scan2Map (PCR/src/NdtRegister.cpp:21-31):
//   N1 voxel Gaussians   voxel_grid_covariance_omp_impl.hpp:49-370 (serial std::map there)
//      -> ndt_voxel_kernel: one thread per cell of the uniform index (cell = resolution):
//         centred fixed-point sums (order independent) -> the reference's single-pass covariance expression, Jacobi eigen
//         decomposition, eigenvalue inflation 0.01 * lambda_max, inverse
//   N3 computeDerivatives / updateDerivatives   ndt_omp_impl.hpp:180-285,399-440,485-537
//      -> ndt_derivatives_kernel: per source point, <= 7 neighbour cells (N6, :374-433),
//         FLOAT inner math exactly as the reference orders it, f64 accumulation, fixed-order sums
//   N5 computeHessian / updateHessian           ndt_omp_impl.hpp:541-645 (serial, double)
//      -> ndt_hessian_kernel
//   N2/N4 Newton step + More-Thuente line search: host code in capi.hip.
#include <string.h>

#include "pcr_internal.h"
#include "ndt_opt.h"
#include "peer_exchange.h"
#include "small_math.h"

namespace pcr {

__device__ inline void sym3_eig_asc(const double A[6], double w[3], double V[3][3]);   // below

// cyclic Jacobi, eigenvalues ASCENDING (Eigen::SelfAdjointEigenSolver order), V columns
__device__ inline void sym3_eig_asc(const double A[6] /* xx xy xz yy yz zz */, double w[3], double V[3][3]) {
    double a[3][3] = {{A[0], A[1], A[2]}, {A[1], A[3], A[4]}, {A[2], A[4], A[5]}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-22 * diag) break;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] != 0.0) {
                    const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                    for (int k = 0; k < 3; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq; }
#pragma unroll
                    for (int k = 0; k < 3; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk; }
#pragma unroll
                    for (int k = 0; k < 3; ++k) { const double vkp = v[k][p], vkq = v[k][q]; v[k][p] = c * vkp - s * vkq; v[k][q] = s * vkp + c * vkq; }
                }
            }
        }
    }
    double e0 = a[0][0], e1 = a[1][1], e2 = a[2][2];
    double c0[3] = {v[0][0], v[1][0], v[2][0]}, c1[3] = {v[0][1], v[1][1], v[2][1]}, c2[3] = {v[0][2], v[1][2], v[2][2]};
#define SWAPCOL(ea, ca, eb, cb) if (eb < ea) { double t_ = ea; ea = eb; eb = t_; for (int k_ = 0; k_ < 3; ++k_) { double u_ = ca[k_]; ca[k_] = cb[k_]; cb[k_] = u_; } }
    SWAPCOL(e0, c0, e1, c1) SWAPCOL(e0, c0, e2, c2) SWAPCOL(e1, c1, e2, c2)
#undef SWAPCOL
    w[0] = e0; w[1] = e1; w[2] = e2;
#pragma unroll
    for (int k = 0; k < 3; ++k) { V[k][0] = c0[k]; V[k][1] = c1[k]; V[k][2] = c2[k]; }
}

__device__ inline void inv3_general(const double m[9], double out[9]) {
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02, id = 1.0 / det;
    out[0] = c00 * id; out[1] = (m[2] * m[7] - m[1] * m[8]) * id; out[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    out[3] = c01 * id; out[4] = (m[0] * m[8] - m[2] * m[6]) * id; out[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    out[6] = c02 * id; out[7] = (m[1] * m[6] - m[0] * m[7]) * id; out[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// ------------------------------------------------------------------------------
// N1: voxel Gaussians, one thread per index cell
// ------------------------------------------------------------------------------
static constexpr double kFix1 = 17592186044416.0;   // 2^44 for sums of (x - c)
static constexpr double kFix2 = 1099511627776.0;    // 2^40 for sums of (x - c)(x - c)^T

// Pass 1: the cells with enough points, compacted into a list; every slot is cleared.  Only a tenth of the cells of a lidar map
// qualify: with one thread per cell nearly every wave carried a few of them and ran the whole eigen-decomposition path at ~10 %
// lane utilisation.  A block owns a contiguous range of cells, counts its candidates, claims room for all of them with ONE
// atomic on the shared counter (an atomic per 1024 cells was 1 600 same-address atomics for a 5 M-point map: they serialise at
// the memory side, 31 us) and then writes them; the order of the list is immaterial.
// roi (a target prepared for one scan, pcr_internal.h: RoiView): a qualifying cell OUTSIDE the region is not listed; its slot is set to
// kNdtUnprepared, which the lookups of the optimiser count as an escape.
__device__ __forceinline__ bool ndt_cell_in_roi(const GridHeader& h, const RoiView& roi, uint64_t t) {
    return roi_mask_holds_cell(h, roi.mask, roi.mshift, (uint32_t)t);
}
__global__ __launch_bounds__(256) void ndt_candidates_kernel(GridView g, uint32_t* __restrict__ vox_slot, uint32_t* __restrict__ list,
                                                             uint32_t* __restrict__ count, uint32_t* __restrict__ count_next, int min_points, uint32_t capacity,
                                                             const RoiView roi) {
    __shared__ uint32_t sh_w[4];
    __shared__ uint32_t sh_base;
    // two counters, used alternately: this call counts in `count` (left at zero by the previous call) and clears the other for the next
    // one -- a memset launch less per call.  Before anything can return.
    if (blockIdx.x == 0 && threadIdx.x == 0) *count_next = 0u;
    const GridHeader h = *g.hdr;
    if (h.overflow) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // this block's cells: [c_lo, c_hi), whole steps of 1024 (four consecutive cells per thread)
    const uint64_t steps = (h.n_cells + 1023) / 1024, per = (steps + gridDim.x - 1) / gridDim.x;
    const uint64_t s_lo = (uint64_t)blockIdx.x * per, s_hi = s_lo + per < steps ? s_lo + per : steps;
    if (s_lo >= s_hi) return;
    // round 1: count (and clear the slots)
    uint32_t mine_total = 0;
    for (uint64_t st = s_lo; st < s_hi; ++st) {
        const uint64_t t0 = st * 1024 + (uint64_t)threadIdx.x * 4;
        // (an index of the region's points only holds nothing outside the mask: EVERY cell there counts as unprepared -- the four cells
        //  of a thread are neighbours along x: one decode serves them unless the row ends in between)
        const bool every = roi.mask && roi.filtered;
        uint32_t cx0 = 0, mrow = 0;
        if (every && t0 < h.n_cells) {
            const uint32_t d0 = (uint32_t)h.dims[0], d1 = (uint32_t)h.dims[1], row = (uint32_t)t0 / d0, cz = row / d1;
            cx0 = (uint32_t)t0 - row * d0;
            mrow = roi_macro(h, roi.mshift, 0, (int)(row - cz * d1), (int)cz);
        }
        uint32_t sl[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t t = t0 + u;
            if (t < h.n_cells) {
                bool in = true, full;
                if (every) {      // (the mask first: two cells in three lie outside it and their counts need not be read)
                    in = cx0 + u < (uint32_t)h.dims[0] ? roi.mask[mrow + ((cx0 + u) >> roi.mshift)] != 0 : ndt_cell_in_roi(h, roi, t);
                    full = in && !h.empty && (int)(g.cell_start[t + 1] - g.cell_start[t]) >= min_points;
                } else {
                    full = !h.empty && (int)(g.cell_start[t + 1] - g.cell_start[t]) >= min_points;
                    if (full && roi.mask) in = ndt_cell_in_roi(h, roi, t);
                }
                mine_total += (full && in) ? 1u : 0u;
                sl[u] = ((full || every) && !in && !h.empty) ? kNdtUnprepared : 0u;
            }
        }
        // (the table is written in full by every target: one 16-byte store per thread -- the slot table's allocation is 256-byte aligned
        //  and t0 a multiple of four)
        if (t0 + 3 < h.n_cells) *reinterpret_cast<uint4*>(vox_slot + t0) = make_uint4(sl[0], sl[1], sl[2], sl[3]);
        else {
#pragma unroll
            for (int u = 0; u < 4; ++u) if (t0 + u < h.n_cells) vox_slot[t0 + u] = sl[u];
        }
    }
    uint32_t inc = mine_total;                     // inclusive prefix over the block
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(inc, d, 64); if (lane >= d) inc += v; }
    if (lane == 63) sh_w[wave] = inc;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { if (w < wave) off += sh_w[w]; tot += sh_w[w]; }
    if (threadIdx.x == 0) sh_base = tot ? atomicAdd(count, tot) : 0u;
    __syncthreads();
    if (!tot) return;
    // round 2: the same cells again (cell_start is L2-resident), now with a place to put them
    uint32_t pos = sh_base + off + inc - mine_total;
    for (uint64_t st = s_lo; st < s_hi; ++st) {
        const uint64_t t0 = st * 1024 + (uint64_t)threadIdx.x * 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t t = t0 + u;
            // (same cells as round 1; with an index of the region only the mask is asked first, as there)
            const bool every = roi.mask && roi.filtered;
            if (t < h.n_cells && !h.empty && (!every || ndt_cell_in_roi(h, roi, t)) && (int)(g.cell_start[t + 1] - g.cell_start[t]) >= min_points &&
                (every || !roi.mask || ndt_cell_in_roi(h, roi, t))) {
                if (pos < capacity) list[pos] = (uint32_t)t;      // (capacity = points / min_points: never short)
                ++pos;
            }
        }
    }
}

// Pass 2: one thread per listed cell
__global__ __launch_bounds__(256) void ndt_voxel_kernel(GridView g, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                        uint32_t* __restrict__ vox_slot, NdtVoxel* __restrict__ vox, double eig_mult, uint32_t capacity) {
    const GridHeader h = *g.hdr;
    if (h.overflow) return;
    const uint32_t n_list = min(*count, capacity);
    for (uint32_t li = blockIdx.x * 256 + threadIdx.x; li < n_list; li += gridDim.x * 256) {
        const uint64_t t = list[li];
        const uint32_t s = g.cell_start[t], e = g.cell_start[t + 1];
        const int n = (int)(e - s);
        uint32_t slot = 0;
        {
            // (32-bit: the list holds 32-bit cell numbers, and a 64-bit division is a hundred instructions of this instruction-bound kernel)
            const uint32_t t32 = (uint32_t)t, d0 = (uint32_t)h.dims[0], d1 = (uint32_t)h.dims[1], row = t32 / d0;
            const int cx = (int)(t32 - row * d0), cz = (int)(row / d1), cy = (int)(row - (uint32_t)cz * d1);
            const double ox = (h.org[0] + cx + 0.5) * h.cell, oy = (h.org[1] + cy + 0.5) * h.cell, oz = (h.org[2] + cz + 0.5) * h.cell;
            // Order-independent exact sums: every term is rounded to an integer multiple of 2^-44 (2^-40 for the products) first.
            // The integers are added up as DOUBLES while their sum provably stays below 2^53 (|dx| <= cell / 2, so n * cell <= 512 and
            // n * cell^2 <= 16384 do it with a factor of two to spare): v_rndne_f64 + v_add_f64 per term.  The same integers in
            // int64 cost a dozen instructions per term for the conversion alone (there is no f64 -> i64 instruction), and this loop
            // was a third of the kernel's instructions; voxels too crowded or too large for the bound take that path.
            double c1[3], c2s[6];
            if ((double)n * h.cell <= 512.0 && (double)n * h.cell * h.cell <= 16384.0) {
                double d1[3] = {0, 0, 0}, d2[6] = {0, 0, 0, 0, 0, 0};
                // four points per step, their loads issued together (one dependent load per iteration left the lanes of crowded cells
                // waiting a memory round trip per point; the sums are integers, so the grouping does not change a bit)
                for (uint32_t j = s; j < e; j += 4) {
                    float4 q[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) q[u] = g.pts[j + u < e ? j + u : j];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (j + u < e) {
                            const double dx = (double)q[u].x - ox, dy = (double)q[u].y - oy, dz = (double)q[u].z - oz;
                            d1[0] += rint(dx * kFix1); d1[1] += rint(dy * kFix1); d1[2] += rint(dz * kFix1);
                            d2[0] += rint(dx * dx * kFix2); d2[1] += rint(dx * dy * kFix2); d2[2] += rint(dx * dz * kFix2);
                            d2[3] += rint(dy * dy * kFix2); d2[4] += rint(dy * dz * kFix2); d2[5] += rint(dz * dz * kFix2);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) c1[k] = d1[k] / kFix1;
#pragma unroll
                for (int k = 0; k < 6; ++k) c2s[k] = d2[k] / kFix2;
            } else {
                long long s1[3] = {0, 0, 0}, s2[6] = {0, 0, 0, 0, 0, 0};
                for (uint32_t j = s; j < e; j += 4) {
                    float4 q[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) q[u] = g.pts[j + u < e ? j + u : j];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (j + u < e) {
                            const double dx = (double)q[u].x - ox, dy = (double)q[u].y - oy, dz = (double)q[u].z - oz;
                            s1[0] += llrint(dx * kFix1); s1[1] += llrint(dy * kFix1); s1[2] += llrint(dz * kFix1);
                            s2[0] += llrint(dx * dx * kFix2); s2[1] += llrint(dx * dy * kFix2); s2[2] += llrint(dx * dz * kFix2);
                            s2[3] += llrint(dy * dy * kFix2); s2[4] += llrint(dy * dz * kFix2); s2[5] += llrint(dz * dz * kFix2);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) c1[k] = (double)s1[k] / kFix1;
#pragma unroll
                for (int k = 0; k < 6; ++k) c2s[k] = (double)s2[k] / kFix2;
            }
            // The reference accumulates sum(x) and sum(x x^T) about the ORIGIN and then evaluates
            //   cov = (sum_xx - 2 (sum_x mean^T)) / n + mean mean^T,  cov *= (n-1)/n      (:329-330, all nine entries)
            // whose cancellation noise decides whether a (nearly) flat voxel keeps a non-negative smallest eigenvalue (:337-341).
            // Same expression here, from the order-independent centred sums moved back to the origin: identical to the
            // reference whenever its own sums are exact (coordinates on a binary lattice), as close as its noise otherwise.
            const double dn = (double)n, o[3] = {ox, oy, oz};
            const double c2[3][3] = {{c2s[0], c2s[1], c2s[2]}, {c2s[1], c2s[3], c2s[4]}, {c2s[2], c2s[4], c2s[5]}};
            double sum[3], sxx[9], mean[3], cov[9];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                sum[r] = dn * o[r] + c1[r];
#pragma unroll
                for (int c = 0; c < 3; ++c) sxx[r * 3 + c] = ((dn * o[r]) * o[c] + (o[r] * c1[c] + o[c] * c1[r])) + c2[r][c];
            }
            double w[3], V[3][3];
            for (int attempt = 0;; ++attempt) {
#pragma unroll
                for (int r = 0; r < 3; ++r) mean[r] = sum[r] / dn;
                const double f = (dn - 1.0) / dn;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) cov[r * 3 + c] = ((sxx[r * 3 + c] - 2 * (sum[r] * mean[c])) / dn + mean[r] * mean[c]) * f;
                // SelfAdjointEigenSolver reads the lower triangle
                const double C[6] = {cov[0], cov[3], cov[6], cov[4], cov[7], cov[8]};
                sym3_eig_asc(C, w, V);
                // A covariance that is singular but for rounding (duplicated points, collinear points, an exactly flat patch): whether
                // the voxel is kept (:337-341) hangs on the SIGN of that rounding, i.e. on the order the reference added the points up in
                // -- input order (:233-237).  Such a voxel is summed again in exactly that order, in double, about the origin: the sums,
                // and with them the decision, are then the reference's bit for bit.  (Up to 64 points: beyond that the voxel keeps the
                // order-independent sums -- exact on a binary lattice, where the crowded flat voxels of a synthetic map come from.)
                const bool shaky = !(fabs(w[0]) > 1e-9 * fabs(w[2])) || w[0] < 0 || w[1] < 0;
                if (attempt != 0 || !shaky || n > 64) break;
                uint32_t last = 0;      // original index + 1 of the point added last
#pragma unroll
                for (int r = 0; r < 3; ++r) sum[r] = 0.0;
#pragma unroll
                for (int e2 = 0; e2 < 9; ++e2) sxx[e2] = 0.0;
                for (int k = 0; k < n; ++k) {
                    uint32_t best = 0xffffffffu;
                    float4 bp = g.pts[s];
                    for (uint32_t j = s; j < e; ++j) {
                        const float4 q = g.pts[j];
                        const uint32_t id = __float_as_uint(q.w) + 1u;
                        if (id > last && id < best) { best = id; bp = q; }
                    }
                    last = best;
                    const double x[3] = {(double)bp.x, (double)bp.y, (double)bp.z};
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        sum[r] += x[r];
#pragma unroll
                        for (int c = 0; c < 3; ++c) sxx[r * 3 + c] += x[r] * x[c];
                    }
                }
            }
            bool ok = !(w[0] < 0 || w[1] < 0 || w[2] <= 0);            // :337-341
            if (ok) {
                const double minev = eig_mult * w[2];                   // :345-356
                if (w[0] < minev) {
                    w[0] = minev;
                    if (w[1] < minev) w[1] = minev;
                    double E[9], Ei[9], T[9];
#pragma unroll
                    for (int r = 0; r < 3; ++r) { E[r * 3] = V[r][0]; E[r * 3 + 1] = V[r][1]; E[r * 3 + 2] = V[r][2]; }
                    inv3_general(E, Ei);
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) T[r * 3 + c] = E[r * 3 + c] * w[c];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) cov[r * 3 + c] = T[r * 3] * Ei[c] + T[r * 3 + 1] * Ei[3 + c] + T[r * 3 + 2] * Ei[6 + c];
                }
                NdtVoxel v;
                inv3_general(cov, v.icov);                               // :359
                double mx = v.icov[0], mn = v.icov[0];
#pragma unroll
                for (int k = 1; k < 9; ++k) { mx = fmax(mx, v.icov[k]); mn = fmin(mn, v.icov[k]); }
                if (isinf(mx) || isinf(mn)) ok = false;
                if (ok) {
                    v.mean[0] = mean[0]; v.mean[1] = mean[1]; v.mean[2] = mean[2];
                    v.n = n; v.pad = 0;
                    slot = li + 1;                                      // the voxel lives at its position in the list
                    vox[li] = v;
                }
            }
        }
        if (slot) vox_slot[t] = slot;
    }
}

// ------------------------------------------------------------------------------
// neighbourhood: centre cell then +x -x +y -y +z -z (:419-433)
// ------------------------------------------------------------------------------
__device__ __forceinline__ int ndt_neighbours(const GridHeader& h, const uint32_t* __restrict__ vox_slot, float tx, float ty, float tz,
                                              uint32_t slots[7], uint32_t* __restrict__ roi_escapes) {
    int n = 0;
    if (h.overflow || h.empty) return 0;
    // floor(p / leaf) evaluated in float like the reference (:380-382)
    const float leaf = (float)h.cell;
    const double fx = (double)floorf(tx / leaf) - h.org[0], fy = (double)floorf(ty / leaf) - h.org[1], fz = (double)floorf(tz / leaf) - h.org[2];
    if (!(fx >= -1.0 && fx <= (double)h.dims[0] && fy >= -1.0 && fy <= (double)h.dims[1] && fz >= -1.0 && fz <= (double)h.dims[2])) return 0;
    const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
    const int off[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int x = cx + off[k][0], y = cy + off[k][1], z = cz + off[k][2];
        if (x >= 0 && x < h.dims[0] && y >= 0 && y < h.dims[1] && z >= 0 && z < h.dims[2]) {
            const uint32_t s = vox_slot[((uint64_t)z * (uint64_t)h.dims[1] + (uint64_t)y) * (uint64_t)h.dims[0] + (uint64_t)x];
            if (s == kNdtUnprepared) { if (roi_escapes) atomicAdd(roi_escapes, 1u); }      // a voxel the target was not prepared for: the call is repeated on the whole target
            else if (s) slots[n++] = s;
        }
    }
    return n;
}

// sharded target (pcr_set_shard): a source point belongs to the rank whose tile holds its transformed position
__device__ __forceinline__ bool ndt_in_tile(const NdtArgs& a, const float tp[3]) {
    return (double)tp[0] >= a.tile_lo[0] && (double)tp[0] < a.tile_hi[0] && (double)tp[1] >= a.tile_lo[1] && (double)tp[1] < a.tile_hi[1] &&
           (double)tp[2] >= a.tile_lo[2] && (double)tp[2] < a.tile_hi[2];
}

// expf as glibc computes it (2.27 and later: the ARM optimized-routines algorithm, sysdeps/ieee754/flt-32/e_expf.c): the argument is
// scaled by 32/ln2 in double, split into an integer k and a remainder r in [-1/2, 1/2] with the 1.5 * 2^52 trick, 2^(k/32) comes from a
// 32-entry table whose exponent field takes k / 32, and a cubic in r finishes it -- all in double, rounded to float once.  The device's
// own expf differs from it in the last bit for some arguments, which made every term of the NDT sums differ "by an ulp of the
// exponential"; this routine reproduced this image's glibc bit for bit on 50 million arguments (with and without FMA contraction).
// The table is 2^(i/32) correctly rounded, minus i << 47 (generated with 60-digit decimal arithmetic).
__device__ __constant__ unsigned long long kExp2fTab[32] = {
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL, 0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL,
    0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL, 0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL, 0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL,
    0x3feea11473eb0187ULL, 0x3feea589994cce13ULL, 0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL, 0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL,
    0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL};
__device__ __forceinline__ float ndt_expf(float x, const unsigned long long* __restrict__ tab /* kExp2fTab, staged in LDS */) {
    const double N = 32.0, inv_ln2_n = 0x1.71547652b82fep+0 * N, shift = 0x1.8p+52;
    const double c0 = 0x1.c6af84b912394p-5 / N / N / N, c1 = 0x1.ebfce50fac4f3p-3 / N / N, c2 = 0x1.62e42ff0c52d6p-1 / N;
    const uint32_t abstop = (__float_as_uint(x) >> 20) & 0x7ffu;
    if (abstop >= (0x42b00000u >> 20)) {                          // |x| >= 88, infinities, NaN
        if (__float_as_uint(x) == 0xff800000u) return 0.0f;
        if (abstop >= (0x7f800000u >> 20)) return x + x;
        if (x > 0x1.62e42ep6f) return __uint_as_float(0x7f800000u);
        if (x < -0x1.9fe368p6f) return 0.0f;
    }
    double z = inv_ln2_n * (double)x;
    double kd = z + shift;
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= shift;
    const double r = z - kd;
    const double sc = __longlong_as_double((long long)(tab[ki & 31u] + (ki << 47)));
    z = c0 * r + c1;
    const double r2 = r * r;
    double y = c2 * r + 1.0;
    y = z * r2 + y;
    return (float)(y * sc);
}

static constexpr int kNdtBlock = 128;
static constexpr int kNdtStride = 130;
static constexpr int kNdtComp = 43;    // score, gradient 6, Hessian 36

// Fold v[kComp] of every thread of the block into partials[block][48] in a fixed order.  The components go through LDS sixteen
// at a time ([16][kNdtStride] doubles = 16.6 KB: with all 43 at once a block held 45 KB and only three fitted on a CU, so a quarter
// of the 1024 blocks of a pass ran as a second round): thread (e = tid & 15, part = tid >> 4) adds 16 of the 128 values of
// component c0 + e, the parts of a wave are combined with two shuffles, and lanes 0..15 of each wave carry the running sums.
static constexpr int kNdtChunk = 16;
static constexpr int kNdtChunks = (kNdtComp + kNdtChunk - 1) / kNdtChunk;      // 3
template <int kComp, int kB>     // 43 with the Hessian, 7 (score + gradient) in the passes of the line search that do not need it; kB threads
__device__ __forceinline__ void ndt_block_reduce(double* sh /* [16][kB + 2] */, double* sh2 /* [kB / 64][48] */, const double v[kComp],
                                                 double acc[kNdtChunks], bool last, double* __restrict__ partials) {
    constexpr int kStride = kB + 2, kWaves = kB / 64;
    const int tid = threadIdx.x, e = tid & 15, part = tid >> 4, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int ch = 0; ch * kNdtChunk < kComp; ++ch) {
        const int c0 = ch * kNdtChunk;
#pragma unroll
        for (int k = 0; k < kNdtChunk; ++k) if (c0 + k < kComp) sh[k * kStride + tid] = v[c0 + k];
        __syncthreads();
        double s = 0.0;
        if (c0 + e < kComp) {
            const double* row = sh + e * kStride + part * 16;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += row[k];
        }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (lane < 16) acc[ch] += s;
        __syncthreads();
    }
    if (last) {
#pragma unroll
        for (int ch = 0; ch * kNdtChunk < kComp; ++ch)
            if (lane < 16 && ch * kNdtChunk + lane < 48) sh2[wave * 48 + ch * kNdtChunk + lane] = acc[ch];
        __syncthreads();
        if (tid < 48) {
            double r = sh2[tid];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) r += sh2[w * 48 + tid];
            partials[(size_t)blockIdx.x * 48 + tid] = tid < kComp ? r : 0.0;
        }
    }
}

// ------------------------------------------------------------------------------
// N3: computeDerivatives
// ------------------------------------------------------------------------------
template <bool kHessian, int kB>
__device__ __forceinline__ void ndt_derivatives_body(const NdtArgs& a, const NdtPose& T, const NdtAngles& ang, double* sh, double* sh2) {
    constexpr int kComp = kHessian ? kNdtComp : 7;
    // the exponential's table in LDS (a per-lane index into constant memory serialises; the slot is the tail of sh2, which the
    // block reduction uses only up to entry 96)
    unsigned long long* const exp_tab = reinterpret_cast<unsigned long long*>(sh2 + (kB / 64) * 48);
    if (threadIdx.x < 32) exp_tab[threadIdx.x] = kExp2fTab[threadIdx.x];
    __syncthreads();
    const GridHeader h = *a.hdr;
    const float gauss_d2 = (float)a.d2;
    double acc[kNdtChunks] = {0.0, 0.0, 0.0};
    const uint32_t step = gridDim.x * kB;
    for (uint32_t base = blockIdx.x * kB; base < a.n_src; base += step) {
        const uint32_t idx = base + threadIdx.x;
        double v[kComp];
#pragma unroll
        for (int k = 0; k < kComp; ++k) v[k] = 0.0;
        if (idx < a.n_src) {
            const float* xp = a.src + (size_t)idx * a.src_stride;
            const float x4[3] = {xp[0], xp[1], xp[2]};
            // pcl::transformPointCloud in float
            float tp[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) { float s = T.R[r * 3] * x4[0]; s += T.R[r * 3 + 1] * x4[1]; s += T.R[r * 3 + 2] * x4[2]; s += T.t[r]; tp[r] = s; }
            uint32_t slots[7];
            int nn = ndt_neighbours(h, a.vox_slot, tp[0], tp[1], tp[2], slots, a.roi_escapes);
            if (a.use_tile && !ndt_in_tile(a, tp)) nn = 0;       // another rank's query
            if (a.pair_count && nn > 0) atomicAdd(a.pair_count + (kHessian ? 48 : 32), (uint32_t)nn);      // (profiling passes only)
            if (nn > 0) {
                // computePointDerivatives (float): :399-440
                float pg[3][6];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 6; ++c) pg[r][c] = (r == c) ? 1.0f : 0.0f;
                float xj[8], xh[15];
#pragma unroll
                for (int r = 0; r < 8; ++r) { float s = ang.j[r][0] * x4[0]; s += ang.j[r][1] * x4[1]; s += ang.j[r][2] * x4[2]; xj[r] = s; }
                pg[1][3] = xj[0]; pg[2][3] = xj[1]; pg[0][4] = xj[2]; pg[1][4] = xj[3]; pg[2][4] = xj[4]; pg[0][5] = xj[5]; pg[1][5] = xj[6]; pg[2][5] = xj[7];
#pragma unroll
                for (int r = 0; r < 15; ++r) { float s = ang.h[r][0] * x4[0]; s += ang.h[r][1] * x4[1]; s += ang.h[r][2] * x4[2]; xh[r] = s; }
                // second derivatives of the transform for parameter pairs (i,j), i,j in 3..5; index (i-3)*3+(j-3)
                const float ph[9][3] = {{0, xh[0], xh[1]}, {0, xh[2], xh[3]}, {0, xh[4], xh[5]},
                                        {0, xh[2], xh[3]}, {xh[6], xh[7], xh[8]}, {xh[9], xh[10], xh[11]},
                                        {0, xh[4], xh[5]}, {xh[9], xh[10], xh[11]}, {xh[12], xh[13], xh[14]}};
                for (int k = 0; k < nn; ++k) {
                    const NdtVoxel cell = a.vox[slots[k] - 1];
                    const double xt[3] = {(double)tp[0] - cell.mean[0], (double)tp[1] - cell.mean[1], (double)tp[2] - cell.mean[2]};
                    const float x4t[3] = {(float)xt[0], (float)xt[1], (float)xt[2]};
                    float ci[3][3];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) ci[r][c] = (float)cell.icov[r * 3 + c];
                    float xc[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) { float s = x4t[0] * ci[0][c]; s += x4t[1] * ci[1][c]; s += x4t[2] * ci[2][c]; xc[c] = s; }
                    float dot = x4t[0] * xc[0]; dot += x4t[1] * xc[1]; dot += x4t[2] * xc[2];
                    float e = ndt_expf(-gauss_d2 * dot * 0.5f, exp_tab);         // :497 (the C library's expf: see ndt_expf)
                    const float score_inc = (float)(-a.d1 * (double)e);          // :499
                    e = gauss_d2 * e;
                    if (e > 1 || e < 0 || e != e) continue;                      // :504-505
                    e = (float)((double)e * a.d1);
                    // c_inv * point_gradient.  The first three columns of the point gradient are the identity and pg[0][3] is zero by
                    // construction (:399-440).  Written out, ci * 1 + cj * 0 + ck * 0 is ci to the bit for finite entries (a product with
                    // 0 is +-0, and adding +-0 changes nothing but the sign of a zero), but a compiler that may not assume finite values
                    // has to keep every one of those multiplies and adds: a fifth of a pass's arithmetic.  They are left out here.
                    float cpg[3][6];
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        cpg[r][0] = ci[r][0]; cpg[r][1] = ci[r][1]; cpg[r][2] = ci[r][2];
                        { float s = ci[r][1] * pg[1][3]; s += ci[r][2] * pg[2][3]; cpg[r][3] = s; }
#pragma unroll
                        for (int c = 4; c < 6; ++c) { float s = ci[r][0] * pg[0][c]; s += ci[r][1] * pg[1][c]; s += ci[r][2] * pg[2][c]; cpg[r][c] = s; }
                    }
                    float xcpg[6];
#pragma unroll
                    for (int c = 0; c < 6; ++c) { float s = x4t[0] * cpg[0][c]; s += x4t[1] * cpg[1][c]; s += x4t[2] * cpg[2][c]; xcpg[c] = s; }
#pragma unroll
                    for (int c = 0; c < 6; ++c) v[1 + c] += (double)(e * xcpg[c]);
                    if constexpr (kHessian) {
#pragma unroll
                        for (int i = 0; i < 6; ++i) {
#pragma unroll
                            for (int j = 0; j < 6; ++j) {
                                // (PG^T C PG)(j,i) and x^T C (second derivative), with the structural zeros and ones left out as above:
                                // column j < 3 of the point gradient is a unit vector, pg[0][3] and the first component of five of the
                                // nine second derivatives are zero
                                float pgc;
                                if (j < 3) pgc = cpg[j][i];
                                else if (j == 3) { pgc = pg[1][3] * cpg[1][i]; pgc += pg[2][3] * cpg[2][i]; }
                                else { pgc = pg[0][j] * cpg[0][i]; pgc += pg[1][j] * cpg[1][i]; pgc += pg[2][j] * cpg[2][i]; }
                                float t = -gauss_d2 * xcpg[i] * xcpg[j];
                                if (i >= 3 && j >= 3) {
                                    const int qi = (i - 3) * 3 + (j - 3);
                                    const float* q = ph[qi];
                                    float xph;
                                    if (qi == 0 || qi == 1 || qi == 2 || qi == 3 || qi == 6) { xph = xc[1] * q[1]; xph += xc[2] * q[2]; }
                                    else { xph = xc[0] * q[0]; xph += xc[1] * q[1]; xph += xc[2] * q[2]; }
                                    t += xph;
                                }
                                v[7 + i * 6 + j] += (double)(e * (t + pgc));
                            }
                        }
                    }
                    v[0] += (double)score_inc;
                }
            }
        }
        ndt_block_reduce<kComp, kB>(sh, sh2, v, acc, base + step >= a.n_src, a.partials);
    }
}

// ------------------------------------------------------------------------------
// N5: computeHessian (double)
// ------------------------------------------------------------------------------
template <int kB>
__device__ __forceinline__ void ndt_hessian_body(const NdtArgs& a, const NdtPose& T, const NdtAngles& ang, double* sh, double* sh2) {
    const GridHeader h = *a.hdr;
    double acc[kNdtChunks] = {0.0, 0.0, 0.0};
    const uint32_t step = gridDim.x * kB;
    for (uint32_t base = blockIdx.x * kB; base < a.n_src; base += step) {
        const uint32_t idx = base + threadIdx.x;
        double v[kNdtComp];
#pragma unroll
        for (int k = 0; k < kNdtComp; ++k) v[k] = 0.0;
        if (idx < a.n_src) {
            const float* xp = a.src + (size_t)idx * a.src_stride;
            float tp[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) { float s = T.R[r * 3] * xp[0]; s += T.R[r * 3 + 1] * xp[1]; s += T.R[r * 3 + 2] * xp[2]; s += T.t[r]; tp[r] = s; }
            uint32_t slots[7];
            int nn = ndt_neighbours(h, a.vox_slot, tp[0], tp[1], tp[2], slots, a.roi_escapes);
            if (a.use_tile && !ndt_in_tile(a, tp)) nn = 0;
            if (nn > 0) {
                const double x[3] = {(double)xp[0], (double)xp[1], (double)xp[2]};
                double pg[3][6];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 6; ++c) pg[r][c] = (r == c) ? 1.0 : 0.0;
                double dj[8], dh[15];
#pragma unroll
                for (int r = 0; r < 8; ++r) dj[r] = x[0] * ang.jd[r][0] + x[1] * ang.jd[r][1] + x[2] * ang.jd[r][2];
#pragma unroll
                for (int r = 0; r < 15; ++r) dh[r] = x[0] * ang.hd[r][0] + x[1] * ang.hd[r][1] + x[2] * ang.hd[r][2];
                pg[1][3] = dj[0]; pg[2][3] = dj[1]; pg[0][4] = dj[2]; pg[1][4] = dj[3]; pg[2][4] = dj[4]; pg[0][5] = dj[5]; pg[1][5] = dj[6]; pg[2][5] = dj[7];
                const double ph[9][3] = {{0, dh[0], dh[1]}, {0, dh[2], dh[3]}, {0, dh[4], dh[5]},
                                         {0, dh[2], dh[3]}, {dh[6], dh[7], dh[8]}, {dh[9], dh[10], dh[11]},
                                         {0, dh[4], dh[5]}, {dh[9], dh[10], dh[11]}, {dh[12], dh[13], dh[14]}};
                for (int k = 0; k < nn; ++k) {
                    const NdtVoxel cell = a.vox[slots[k] - 1];
                    const double xt[3] = {(double)tp[0] - cell.mean[0], (double)tp[1] - cell.mean[1], (double)tp[2] - cell.mean[2]};
                    const double* ci = cell.icov;
                    double cx[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r) cx[r] = ci[r * 3] * xt[0] + ci[r * 3 + 1] * xt[1] + ci[r * 3 + 2] * xt[2];
                    double e = a.d2 * exp(-a.d2 * (xt[0] * cx[0] + xt[1] * cx[1] + xt[2] * cx[2]) / 2);     // :623
                    if (e > 1 || e < 0 || e != e) continue;
                    e *= a.d1;
                    // (structural zeros and ones of the point gradient and of the second derivatives left out, as in the float pass above)
                    double cpg[6][3], xd[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            if (i < 3) cpg[i][r] = ci[r * 3 + i];
                            else if (i == 3) cpg[i][r] = ci[r * 3 + 1] * pg[1][3] + ci[r * 3 + 2] * pg[2][3];
                            else cpg[i][r] = ci[r * 3] * pg[0][i] + ci[r * 3 + 1] * pg[1][i] + ci[r * 3 + 2] * pg[2][i];
                        }
                        xd[i] = xt[0] * cpg[i][0] + xt[1] * cpg[i][1] + xt[2] * cpg[i][2];
                    }
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            double t = -a.d2 * xd[i] * xd[j];
                            if (i >= 3 && j >= 3) {
                                const int qi = (i - 3) * 3 + (j - 3);
                                const double* q = ph[qi];
                                double cph[3];
                                const bool q0_zero = qi == 0 || qi == 1 || qi == 2 || qi == 3 || qi == 6;
#pragma unroll
                                for (int r = 0; r < 3; ++r) cph[r] = q0_zero ? ci[r * 3 + 1] * q[1] + ci[r * 3 + 2] * q[2] : ci[r * 3] * q[0] + ci[r * 3 + 1] * q[1] + ci[r * 3 + 2] * q[2];
                                t += xt[0] * cph[0] + xt[1] * cph[1] + xt[2] * cph[2];
                            }
                            double t3;
                            if (j < 3) t3 = cpg[i][j];
                            else if (j == 3) t3 = pg[1][3] * cpg[i][1] + pg[2][3] * cpg[i][2];
                            else t3 = pg[0][j] * cpg[i][0] + pg[1][j] * cpg[i][1] + pg[2][j] * cpg[i][2];
                            v[7 + i * 6 + j] += e * (t + t3);
                        }
                    }
                }
            }
        }
        ndt_block_reduce<kNdtComp, kB>(sh, sh2, v, acc, base + step >= a.n_src, a.partials);
    }
}

// the three evaluation kernels of the host-driven loop (pose and tables as kernel arguments)
template <bool kHessian>
__global__ __launch_bounds__(kNdtBlock, 2) void ndt_derivatives_kernel(const NdtArgs a, const NdtPose T, const NdtAngles ang) {
    __shared__ double sh[kNdtChunk * kNdtStride];
    __shared__ double sh2[2 * 48 + 32];
    ndt_derivatives_body<kHessian, kNdtBlock>(a, T, ang, sh, sh2);
}
__global__ __launch_bounds__(kNdtBlock, 2) void ndt_hessian_kernel(const NdtArgs a, const NdtPose T, const NdtAngles ang) {
    __shared__ double sh[kNdtChunk * kNdtStride];
    __shared__ double sh2[2 * 48 + 32];
    ndt_hessian_body<kNdtBlock>(a, T, ang, sh, sh2);
}

// ------------------------------------------------------------------------------
// Device-resident optimiser: one PASS = ndt_pass_kernel (whatever the controller asked for: derivatives with or without
// the Hessian, or computeHessian, at the pose it left in NdtCtl) + ndt_fold_ctl_kernel (fold of the partial sums and one step
// of the state machine of ndt_opt.h).  The host enqueues passes ahead of the device and only watches a progress word; passes
// enqueued beyond the end of the optimisation return at once.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(kNdtBlock, 2) void ndt_pass_kernel(const NdtArgs a, const NdtCtl* __restrict__ ctl) {
    __shared__ double sh[kNdtChunk * kNdtStride];
    __shared__ double sh2[2 * 48 + 32];
    const int kind = ctl->kind;
    if (kind == kNdtPassNone) return;
    const NdtPose T = ctl->T;
    if (kind == kNdtPassDerivH) ndt_derivatives_body<true, kNdtBlock>(a, T, ctl->ang, sh, sh2);
    else if (kind == kNdtPassDeriv) ndt_derivatives_body<false, kNdtBlock>(a, T, ctl->ang, sh, sh2);
    else ndt_hessian_body<kNdtBlock>(a, T, ctl->ang, sh, sh2);
}

// the controller's initial state, prepared on the host (ndt_opt::ctl_init) and handed over as a kernel argument
struct NdtCtlArg { uint32_t w[(sizeof(NdtCtl) + 3) / 4]; };
__global__ __launch_bounds__(512) void ndt_ctl_store_kernel(NdtCtl* __restrict__ ctl, const NdtCtlArg init, uint32_t* __restrict__ roi_escapes) {
    const int n = (int)(sizeof(NdtCtl) / 4);
    for (int t = threadIdx.x; t < n; t += 512) reinterpret_cast<uint32_t*>(ctl)[t] = init.w[t];
    if (roi_escapes && threadIdx.x == 0) *roi_escapes = 0u;
}

static constexpr int kCtlWords = (int)((sizeof(NdtCtl) + 3) / 4);
// One block of 768 threads.  Fold: a line-search pass carries 7 sums (score + gradient), the others 43; the threads are laid
// out as [slice][component] with 8 or 48 components per slice, so that the 7-sum fold is a single round of loads (96 slices x
// 11 blocks) instead of eight.  Controller: thread 0 takes the decisions (ndt_opt::ctl_decide), six lanes evaluate the six
// sine/cosine pairs of the new pose at once, thread 0 fills the tables.
__global__ __launch_bounds__(768) void ndt_fold_ctl_kernel(const double* __restrict__ partials, uint32_t nblocks, NdtCtl* __restrict__ ctl,
                                                           const GridHeader* __restrict__ hdr, NdtOut* __restrict__ out, double seq, const uint32_t* __restrict__ roi_escapes) {
    __shared__ double sh[96 * 8];                // = 16 * 48
    __shared__ double sh_sums[48];
    __shared__ double sh_sc[12];
    __shared__ int sh_need;
    __shared__ __attribute__((aligned(16))) uint32_t sh_ctl[kCtlWords];
    const unsigned long long t_in = wall_clock64();
    const int t = threadIdx.x;
    // Everything this kernel needs from memory is requested in ONE round trip: the controller's state, and -- on the guess that this was
    // a line-search pass, which ten of thirteen are -- the 7 sums of every block in the [96 slices][8 components] layout.
    const int compL = t & 7, sliceL = t >> 3;
    double vL[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) { const uint32_t b = (uint32_t)(sliceL + 96 * u); vL[u] = b < nblocks ? partials[(size_t)b * 48 + compL] : 0.0; }
    // the controller's state to LDS (thread 0 then works at LDS latency instead of one memory round trip per field)
    for (int w = t; w < kCtlWords; w += 768) sh_ctl[w] = reinterpret_cast<const uint32_t*>(ctl)[w];
    const int done = ctl->done, kind = ctl->kind;
#pragma unroll
    for (int u = 0; u < 12; ++u) asm volatile("" ::"v"(vL[u]));      // (keeps the speculative loads above the branches below)
    if (done) return;                            // a pass enqueued beyond the end
    const bool light = kind == kNdtPassDeriv;
    const int cw = light ? 8 : 48, ns = 768 / cw;             // components per slice, slices
    const int comp = t % cw, slice = t / cw;
    double acc = 0.0;
    if (light) {
#pragma unroll
        for (int u = 0; u < 12; ++u) acc += vL[u];
    } else {
        for (uint32_t b0 = slice; b0 < nblocks; b0 += (uint32_t)ns * 12u) {
            double v[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) { const uint32_t b = b0 + (uint32_t)(ns * u); v[u] = b < nblocks ? partials[(size_t)b * 48 + comp] : 0.0; }
#pragma unroll
            for (int u = 0; u < 12; ++u) acc += v[u];
        }
    }
    sh[slice * cw + comp] = acc;
    __syncthreads();
    if (t < 48) {
        double v = 0.0;
        if (t < cw) {
            v = sh[t];
            for (int s2 = 1; s2 < ns; ++s2) v += sh[s2 * cw + t];
        }
        sh_sums[t] = v;                          // (a light pass leaves the Hessian slots at zero: ctl_decide does not read them)
    }
    __syncthreads();
    unsigned long long t_a = 0;
    if (t == 0) {
        t_a = wall_clock64();
        sh_need = ndt_opt::ctl_decide(reinterpret_cast<NdtCtl*>(sh_ctl), sh_sums) ? 1 : 0;
        reinterpret_cast<NdtCtl*>(sh_ctl)->ticks[2] += (uint32_t)(wall_clock64() - t_a);
    }
    __syncthreads();
    if (sh_need) {
        if (t < 6) {
            double sc[2];
            ndt_opt::trig_pair(reinterpret_cast<const NdtCtl*>(sh_ctl)->x_t, t, sc);
            sh_sc[2 * t] = sc[0]; sh_sc[2 * t + 1] = sc[1];
        }
        __syncthreads();
        if (t == 0) {
            const unsigned long long t_c = wall_clock64();
            ndt_opt::ctl_tables(reinterpret_cast<NdtCtl*>(sh_ctl), sh_sc);
            reinterpret_cast<NdtCtl*>(sh_ctl)->ticks[3] += (uint32_t)(wall_clock64() - t_c);
        }
    }
    if (t == 0) {
        NdtCtl* c = reinterpret_cast<NdtCtl*>(sh_ctl);
        const unsigned long long t_b = wall_clock64();
        c->ticks[0] += (uint32_t)(t_a - t_in); c->ticks[1] += (uint32_t)(t_b - t_a);
    }
    __syncthreads();
    for (int w = t; w < kCtlWords; w += 768) reinterpret_cast<uint32_t*>(ctl)[w] = sh_ctl[w];
    if (t == 0) {
        const NdtCtl* c = reinterpret_cast<const NdtCtl*>(sh_ctl);
        if (c->done) {
            out->final_T = c->final_T; out->score = c->score;
            out->conv = c->conv; out->nr_it = c->nr_it; out->n_deriv = c->n_deriv; out->n_hess = c->n_hess; out->bail = c->bail; out->passes = c->passes;
            for (int i = 0; i < 4; ++i) out->ticks[i] = c->ticks[i];
            out->grid_overflow = hdr->overflow; out->grid_empty = hdr->empty; out->grid_stale = hdr->stale; out->roi_escapes = roi_escapes ? (int32_t)min(*roi_escapes, 0x7fffffffu) : 0; out->grid_cells = hdr->n_cells;
            __threadfence_system();
            __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            __hip_atomic_store(&out->progress, seq * kProgressWindow + (double)c->passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ------------------------------------------------------------------------------
// One launch per pass (the unsharded device loop).  A launch on this stream costs ~5 us of device time whatever it does, and the
// fold/controller kernel above is a second one per pass: here its work is the PROLOGUE of the next pass instead -- every block
// folds the rows of the previous launch and takes the controller step itself (same instructions on the same numbers: the blocks
// agree to the bit, as loam_iterate_kernel's do), block 0 writes the new state and the progress word.  So that a block needs
// 256 rows and not 1024, the blocks are 512 threads, one per CU.  State and rows are double-buffered by launch parity: launch k
// reads what launch k - 1 wrote (buffer (k - 1) & 1) and writes buffer k & 1.  The optimisation ends in the prologue of the launch
// AFTER its last pass; that launch and any queued behind it leave at once, each handing the finished state on to the other buffer.
// ------------------------------------------------------------------------------
static constexpr int kProBlock = 512;
static constexpr int kProRows = 256;
struct NdtProArgs {
    const double* rows_prev;     // [rows_prev_n][48], written by the previous launch
    const NdtCtl* ctl_prev;      // the state the previous launch evaluated
    NdtCtl* ctl_next;            // the state this launch evaluates (written by block 0)
    NdtOut* out;
    double seq;
    uint32_t rows_prev_n;
    int32_t first;               // launch 0 of a call: ctl_prev is the initial state, nothing to fold
};

__device__ __forceinline__ float lds_uniform(const float* p) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, *p)));
}
__device__ __forceinline__ double lds_uniform(const double* p) {
    const double v = *p;
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// pose and tables out of LDS into scalar registers (what a body does not use is never read)
__device__ __forceinline__ void lds_uniform_copy(const NdtCtl* c, NdtPose& T, NdtAngles& ang) {
#pragma unroll
    for (int i = 0; i < 9; ++i) T.R[i] = lds_uniform(&c->T.R[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) T.t[i] = lds_uniform(&c->T.t[i]);
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) { ang.j[r][k] = lds_uniform(&c->ang.j[r][k]); ang.jd[r][k] = lds_uniform(&c->ang.jd[r][k]); }
#pragma unroll
    for (int r = 0; r < 15; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) { ang.h[r][k] = lds_uniform(&c->ang.h[r][k]); ang.hd[r][k] = lds_uniform(&c->ang.hd[r][k]); }
}

__global__ __launch_bounds__(kProBlock, 1) void ndt_pass_pro_kernel(const NdtArgs a, const NdtProArgs pa) {
    __shared__ double sh[kNdtChunk * (kProBlock + 2)];
    __shared__ double sh2[(kProBlock / 64) * 48 + 32];
    __shared__ __attribute__((aligned(16))) uint32_t sh_ctl[kCtlWords];
    __shared__ double sh_sums[48];
    __shared__ double sh_sc[12];
    __shared__ int sh_need;
    const unsigned long long t_in = wall_clock64();
    const int t = threadIdx.x;
    NdtCtl* const c = reinterpret_cast<NdtCtl*>(sh_ctl);
    // ONE round trip: the state and the rows of the previous launch, in the [10 slices][48 components] layout whatever that launch was
    // (since repeated line-search evaluations are no longer passes, most launches follow a pass with a Hessian; a line-search pass
    // fills 7 of the 48 slots of a row and the rest is ignored)
    const int comp = t % 48, slice = t / 48;      // 10 slices: t < 480
    double v[26];
#pragma unroll
    for (int u = 0; u < 26; ++u) {
        const uint32_t row = (uint32_t)(slice + 10 * u);
        v[u] = (!pa.first && t < 480 && row < pa.rows_prev_n) ? pa.rows_prev[(size_t)row * 48 + comp] : 0.0;
    }
    for (int w = t; w < kCtlWords; w += kProBlock) sh_ctl[w] = reinterpret_cast<const uint32_t*>(pa.ctl_prev)[w];
#pragma unroll
    for (int u = 0; u < 26; ++u) asm volatile("" ::"v"(v[u]));
    __syncthreads();
    if (c->done) {      // finished in an earlier launch: hand the state on, so that whatever is queued behind reads it too
        if (blockIdx.x == 0) for (int w = t; w < kCtlWords; w += kProBlock) reinterpret_cast<uint32_t*>(pa.ctl_next)[w] = sh_ctl[w];
        return;
    }
    if (!pa.first) {
        const bool light = c->kind == kNdtPassDeriv;
        if (t < 480) {
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < 26; ++u) acc += v[u];
            sh[slice * 48 + comp] = acc;
        }
        __syncthreads();
        if (t < 48) {
            double r = 0.0;
            if (t < (light ? 7 : kNdtComp)) {
                r = sh[t];
#pragma unroll
                for (int s2 = 1; s2 < 10; ++s2) r += sh[s2 * 48 + t];
            }
            sh_sums[t] = r;      // (a light pass leaves the Hessian slots at zero: ctl_decide does not read them)
        }
        __syncthreads();
        unsigned long long t_a = 0;
        // the pass's 36 Hessian entries are taken in by lanes 1..36 of the wave whose lane 0 decides (same wave: their LDS stores are queued before
        // lane 0's loads of the Newton solve)
        if (t >= 1 && t <= 36) c->hess[t - 1] = ndt_opt::ctl_hess_entry(c->kind, c->phase, sh_sums, t - 1);
        if (t == 0) {
            t_a = wall_clock64();
            sh_need = ndt_opt::ctl_decide(c, sh_sums, true) ? 1 : 0;
            c->ticks[2] += (uint32_t)(wall_clock64() - t_a);
        }
        __syncthreads();
        if (sh_need) {
            if (t < 6) {
                double sc[2];
                ndt_opt::trig_pair(c->x_t, t, sc);
                sh_sc[2 * t] = sc[0]; sh_sc[2 * t + 1] = sc[1];
            }
            __syncthreads();
            // pose and the four parts of the angle tables: five waves, one lane each
            const unsigned long long t_c = wall_clock64();
            if ((t & 63) == 0) {
                const int wave = t >> 6;
                if (wave == 0) ndt_opt::pose_from_trig(c->x_t, sh_sc, &c->T);
                else if (wave == 1) ndt_opt::angle_tables_from_trig(sh_sc, &c->ang, 0);
                else if (wave == 2) ndt_opt::angle_tables_from_trig(sh_sc, &c->ang, 1);
                else if (wave == 3) ndt_opt::angle_tables_from_trig(sh_sc, &c->ang, 2);
                else if (wave == 4) ndt_opt::angle_tables_from_trig(sh_sc, &c->ang, 3);
            }
            __syncthreads();
            if (t == 0) c->ticks[3] += (uint32_t)(wall_clock64() - t_c);
        }
        if (t == 0) {
            const unsigned long long t_b = wall_clock64();
            c->ticks[0] += (uint32_t)(t_a - t_in); c->ticks[1] += (uint32_t)(t_b - t_a);
        }
        __syncthreads();
        if (blockIdx.x == 0) {
            for (int w = t; w < kCtlWords; w += kProBlock) reinterpret_cast<uint32_t*>(pa.ctl_next)[w] = sh_ctl[w];
            if (t == 0) {
                NdtOut* const out = pa.out;
                if (c->done) {
                    const GridHeader* hdr = a.hdr;
                    out->final_T = c->final_T; out->score = c->score;
                    out->conv = c->conv; out->nr_it = c->nr_it; out->n_deriv = c->n_deriv; out->n_hess = c->n_hess; out->bail = c->bail; out->passes = c->passes;
                    for (int i = 0; i < 4; ++i) out->ticks[i] = c->ticks[i];
                    out->grid_overflow = hdr->overflow; out->grid_empty = hdr->empty; out->grid_stale = hdr->stale; out->roi_escapes = a.roi_escapes ? (int32_t)min(*a.roi_escapes, 0x7fffffffu) : 0; out->grid_cells = hdr->n_cells;
                    __threadfence_system();
                    __hip_atomic_store(&out->seq, pa.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                } else {
                    __hip_atomic_store(&out->progress, pa.seq * kProgressWindow + (double)c->passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        if (c->done) return;
    }
    const int kind = __builtin_amdgcn_readfirstlane(c->kind);
    // (the copy inside each branch: a readfirstlane cannot be sunk into one, and 220 scalars live across the branch all spill)
    if (kind == kNdtPassDerivH) {
        NdtPose T; NdtAngles ang;
        lds_uniform_copy(c, T, ang);
        ndt_derivatives_body<true, kProBlock>(a, T, ang, sh, sh2);
    } else if (kind == kNdtPassDeriv) {
        NdtPose T; NdtAngles ang;
        lds_uniform_copy(c, T, ang);
        ndt_derivatives_body<false, kProBlock>(a, T, ang, sh, sh2);
    } else {
        NdtPose T; NdtAngles ang;
        lds_uniform_copy(c, T, ang);
        ndt_hessian_body<kProBlock>(a, T, ang, sh, sh2);
    }
}

// ------------------------------------------------------------------------------
// Sharded targets over RCCL: the sums of a pass have to cross the ranks between the fold and the controller step, so the
// fold/controller kernel is cut in two around an ncclAllReduce on the same stream -- ndt_fold_kernel (this rank's 48 sums into
// a device buffer), all-reduce, ndt_ctl_kernel (one step of the state machine on the summed values).  Every rank holds the same
// controller state and feeds it the same sums: the ranks decide alike and stay in step without ever talking about it.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(768) void ndt_fold_kernel(const double* __restrict__ partials, uint32_t nblocks, const NdtCtl* __restrict__ ctl,
                                                       double* __restrict__ sums48) {
    __shared__ double sh[16 * 48];
    const int t = threadIdx.x, comp = t % 48, slice = t / 48;
    const int kind = ctl->done ? kNdtPassNone : ctl->kind;
    // (a pass enqueued beyond the end still takes part in the collective that follows: with zeros)
    double acc = 0.0;
    if (kind != kNdtPassNone) {
        for (uint32_t b0 = slice; b0 < nblocks; b0 += 16 * 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const uint32_t b = b0 + 16 * u; v[u] = b < nblocks ? partials[(size_t)b * 48 + comp] : 0.0; }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
    }
    sh[slice * 48 + comp] = acc;
    __syncthreads();
    if (t < 48) {
        double v = sh[t];
#pragma unroll
        for (int s2 = 1; s2 < 16; ++s2) v += sh[s2 * 48 + t];
        // a line-search pass fills 7 of the 48 slots of every block; the others hold what an earlier pass left there
        sums48[t] = (kind == kNdtPassDeriv && t >= 7) || t >= kNdtComp ? 0.0 : v;
    }
}

__global__ __launch_bounds__(64) void ndt_ctl_kernel(NdtCtl* __restrict__ ctl, const double* __restrict__ sums48, const GridHeader* __restrict__ hdr,
                                                     NdtOut* __restrict__ out, double seq, int batch_mark, const uint32_t* __restrict__ roi_escapes) {
    __shared__ double sh_sums[48];
    __shared__ double sh_sc[12];
    __shared__ int sh_need;
    __shared__ __attribute__((aligned(16))) uint32_t sh_ctl[kCtlWords];
    const int t = threadIdx.x;
    const int done_in = ctl->done;
    if (!done_in) {
        for (int w = t; w < kCtlWords; w += 64) sh_ctl[w] = reinterpret_cast<const uint32_t*>(ctl)[w];
        if (t < 48) sh_sums[t] = sums48[t];
        __syncthreads();
        if (t == 0) sh_need = ndt_opt::ctl_decide(reinterpret_cast<NdtCtl*>(sh_ctl), sh_sums) ? 1 : 0;
        __syncthreads();
        if (sh_need) {
            if (t < 6) {
                double sc[2];
                ndt_opt::trig_pair(reinterpret_cast<const NdtCtl*>(sh_ctl)->x_t, t, sc);
                sh_sc[2 * t] = sc[0]; sh_sc[2 * t + 1] = sc[1];
            }
            __syncthreads();
            if (t == 0) ndt_opt::ctl_tables(reinterpret_cast<NdtCtl*>(sh_ctl), sh_sc);
        }
        __syncthreads();
        for (int w = t; w < kCtlWords; w += 64) reinterpret_cast<uint32_t*>(ctl)[w] = sh_ctl[w];
    }
    if (t == 0) {
        const NdtCtl* c = done_in ? ctl : reinterpret_cast<const NdtCtl*>(sh_ctl);
        if (c->done && !done_in) {
            out->final_T = c->final_T; out->score = c->score;
            out->conv = c->conv; out->nr_it = c->nr_it; out->n_deriv = c->n_deriv; out->n_hess = c->n_hess; out->bail = c->bail; out->passes = c->passes;
            out->grid_overflow = hdr->overflow; out->grid_empty = hdr->empty; out->grid_stale = hdr->stale; out->roi_escapes = roi_escapes ? (int32_t)min(*roi_escapes, 0x7fffffffu) : 0; out->grid_cells = hdr->n_cells;
            __threadfence_system();
            __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // the end of a batch is marked whether or not the loop has finished: the host decides only then whether another batch is
        // due -- a decision every rank takes from the same state, so that all of them enqueue the same collectives
        if (batch_mark) {
            __threadfence_system();
            __hip_atomic_store(&out->batch, seq * 65536.0 + 2.0 * (double)batch_mark + (c->done ? 1.0 : 0.0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Sharded targets over the PEER exchange (pcr_comm_init_peer): the three pieces above in ONE launch -- this rank's rows folded, the 48 sums pushed
// into every peer's receive buffer and the ranks' contributions folded in rank order (peer_exchange.h), one step of the state machine on the result.
// A pass is then two launches (ndt_pass_kernel + this one) and no collective call, where the RCCL loop needs three launches and an all-reduce and a
// host-supplied collective a round trip through the host per pass.  Every rank holds the same controller state and gets the same bits out of the
// exchange: the ranks decide alike.  A launch queued beyond the end of the optimisation exchanges nothing (every rank's `done` flips in the same
// launch; sequence numbers are per enqueued launch, so the ones that follow still agree).
__global__ __launch_bounds__(768) void ndt_fold_exchange_ctl_kernel(const double* __restrict__ partials, uint32_t nblocks, NdtCtl* __restrict__ ctl, const PeerComm pc,
                                                                    const double xseq, const GridHeader* __restrict__ hdr, NdtOut* __restrict__ out, double seq,
                                                                    int batch_mark, const uint32_t* __restrict__ roi_escapes) {
    __shared__ double sh[16 * 48];
    __shared__ double sh_sums[64];
    __shared__ double sh_sc[12];
    __shared__ int sh_need, sh_fail;
    __shared__ __attribute__((aligned(16))) uint32_t sh_ctl[kCtlWords];
    const int t = threadIdx.x, comp = t % 48, slice = t / 48;
    const int done_in = ctl->done;
    if (!done_in) {
        const int kind = ctl->kind;
        for (int w = t; w < kCtlWords; w += 768) sh_ctl[w] = reinterpret_cast<const uint32_t*>(ctl)[w];
        double acc = 0.0;
        if (kind != kNdtPassNone) {
            for (uint32_t b0 = slice; b0 < nblocks; b0 += 16 * 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const uint32_t b = b0 + 16 * u; v[u] = b < nblocks ? partials[(size_t)b * 48 + comp] : 0.0; }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
        }
        sh[slice * 48 + comp] = acc;
        __syncthreads();
        double mine = 0.0;
        if (t < 48) {
            double v = sh[t];
#pragma unroll
            for (int s2 = 1; s2 < 16; ++s2) v += sh[s2 * 48 + t];
            // a line-search pass fills 7 of the 48 slots of every block; the others hold what an earlier pass left there
            mine = (kind == kNdtPassDeriv && t >= 7) || t >= kNdtComp ? 0.0 : v;
        }
        const bool ok = peer_exchange_block(pc, xseq, nullptr, mine, 48, 0, sh_sums);
        if (t == 0) sh_fail = ok ? 0 : 1;
        __syncthreads();
        if (sh_fail) {      // a rank never arrived (the status word says so to the host): the loop ends here, on every rank that waited
            if (t == 0) { NdtCtl* c = reinterpret_cast<NdtCtl*>(sh_ctl); c->done = 1; c->bail = 0; c->conv = 0; c->kind = kNdtPassNone; }
        } else {
            NdtCtl* const c = reinterpret_cast<NdtCtl*>(sh_ctl);
            if (t >= 1 && t <= 36) c->hess[t - 1] = ndt_opt::ctl_hess_entry(c->kind, c->phase, sh_sums, t - 1);      // (lanes of the deciding lane's wave: see ndt_pass_pro_kernel)
            if (t == 0) sh_need = ndt_opt::ctl_decide(c, sh_sums, true) ? 1 : 0;
            __syncthreads();
            if (sh_need) {
                if (t < 6) {
                    double sc[2];
                    ndt_opt::trig_pair(c->x_t, t, sc);
                    sh_sc[2 * t] = sc[0]; sh_sc[2 * t + 1] = sc[1];
                }
                __syncthreads();
                // pose and the four parts of the angle tables: five waves, one lane each
                if ((t & 63) == 0) {
                    const int wave = t >> 6;
                    if (wave == 0) ndt_opt::pose_from_trig(c->x_t, sh_sc, &c->T);
                    else if (wave == 1) ndt_opt::angle_tables_from_trig(sh_sc, &c->ang, 0);
                    else if (wave == 2) ndt_opt::angle_tables_from_trig(sh_sc, &c->ang, 1);
                    else if (wave == 3) ndt_opt::angle_tables_from_trig(sh_sc, &c->ang, 2);
                    else if (wave == 4) ndt_opt::angle_tables_from_trig(sh_sc, &c->ang, 3);
                }
            }
        }
        __syncthreads();
        for (int w = t; w < kCtlWords; w += 768) reinterpret_cast<uint32_t*>(ctl)[w] = sh_ctl[w];
    }
    if (t == 0) {
        const NdtCtl* c = done_in ? ctl : reinterpret_cast<const NdtCtl*>(sh_ctl);
        if (c->done && !done_in) {
            out->final_T = c->final_T; out->score = c->score;
            out->conv = c->conv; out->nr_it = c->nr_it; out->n_deriv = c->n_deriv; out->n_hess = c->n_hess; out->bail = c->bail; out->passes = c->passes;
            out->grid_overflow = hdr->overflow; out->grid_empty = hdr->empty; out->grid_stale = hdr->stale; out->roi_escapes = roi_escapes ? (int32_t)min(*roi_escapes, 0x7fffffffu) : 0; out->grid_cells = hdr->n_cells;
            __threadfence_system();
            __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        else if (!c->done) __hip_atomic_store(&out->progress, seq * kProgressWindow + (double)c->passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (the host keeps its queue a fixed number of passes ahead of this)
        if (batch_mark) {
            __threadfence_system();
            __hip_atomic_store(&out->batch, seq * 65536.0 + 2.0 * (double)batch_mark + (c->done ? 1.0 : 0.0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// fold per-block partials (48 doubles each) into 48 doubles, fixed order; 16 slices of 48 components, each
// slice keeps 8 independent loads in flight
// out lives in host-mapped memory: out[47] receives `seq` LAST (system-scope release), the word the host spins on
__global__ __launch_bounds__(768) void ndt_sum_partials_kernel(const double* __restrict__ partials, uint32_t nblocks, double* __restrict__ out, double seq) {
    __shared__ double sh[16 * 48];
    const int t = threadIdx.x, comp = t % 48, slice = t / 48;
    double acc = 0.0;
    for (uint32_t b0 = slice; b0 < nblocks; b0 += 16 * 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const uint32_t b = b0 + 16 * u; v[u] = b < nblocks ? partials[(size_t)b * 48 + comp] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    sh[slice * 48 + comp] = acc;
    __syncthreads();
    if (t < 47) {
        double v = sh[t];
#pragma unroll
        for (int s2 = 1; s2 < 16; ++s2) v += sh[s2 * 48 + t];
        out[t] = v;
        __threadfence_system();
    }
    __syncthreads();
    if (t == 0) __hip_atomic_store(&out[47], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- host launchers ---------------------------------------------------------------
uint32_t ndt_blocks(uint32_t n_src) {
    uint32_t b = (n_src + kNdtBlock - 1) / kNdtBlock;
    return b < 1 ? 1 : (b > 1024 ? 1024 : b);
}

hipError_t ndt_launch_voxels(const GridIndex& grid, uint32_t* d_slot, NdtVoxel* d_vox, uint32_t* d_count, uint32_t* d_count_next, uint32_t* d_list, size_t list_capacity,
                             int min_points, double eig_mult, hipStream_t s, const RoiView* roi, bool listed_by_tile_pass) {
    const int blocks = (int)std::min<size_t>(512, grid.cell_capacity / 1024 + 1);
    RoiView rv;
    memset(&rv, 0, sizeof rv);
    if (roi) rv = *roi;
    // (a region-only index has listed the cells and written the slots in its own tile pass: TileTail, grid_index.hip)
    if (!listed_by_tile_pass) hipLaunchKernelGGL(ndt_candidates_kernel, dim3(blocks), dim3(256), 0, s, grid.view(), d_slot, d_list, d_count, d_count_next, min_points, (uint32_t)std::min<size_t>(list_capacity, 0xffffffffu), rv);
    const int vblocks = (int)std::min<size_t>(65535, list_capacity / 256 + 1);
    hipLaunchKernelGGL(ndt_voxel_kernel, dim3(vblocks), dim3(256), 0, s, grid.view(), d_list, d_count, d_slot, d_vox, eig_mult, (uint32_t)std::min<size_t>(list_capacity, 0xffffffffu));
    return hipGetLastError();
}

hipError_t ndt_launch_derivatives(const NdtArgs& a, const NdtPose& T, const NdtAngles& ang, int compute_hessian, double* d_out48, hipStream_t s, double seq) {
    const uint32_t nb = ndt_blocks(a.n_src);
    if (compute_hessian) hipLaunchKernelGGL((ndt_derivatives_kernel<true>), dim3(nb), dim3(kNdtBlock), 0, s, a, T, ang);
    else hipLaunchKernelGGL((ndt_derivatives_kernel<false>), dim3(nb), dim3(kNdtBlock), 0, s, a, T, ang);
    hipLaunchKernelGGL(ndt_sum_partials_kernel, dim3(1), dim3(768), 0, s, a.partials, nb, d_out48, seq);
    return hipGetLastError();
}

static void ndt_initial_ctl(NdtCtl* c, const NdtPose& T0, const double p[6], double step_size, double trans_eps, int max_iters, int no_replay_arg) {
    static_assert(sizeof(NdtCtl) % 4 == 0, "NdtCtl is copied word by word");
    memset(c, 0, sizeof *c);
    ndt_opt::ctl_init(c, T0, p, step_size, trans_eps, max_iters);
    static const bool no_replay = dev_env("PCR_NDT_NO_REPLAY") != nullptr;
    c->replay_off = (no_replay || no_replay_arg) ? 1 : 0;
}
hipError_t ndt_launch_ctl_init(NdtCtl* d_ctl, const NdtPose& T0, const double p[6], double step_size, double trans_eps, int max_iters, hipStream_t s, int no_replay_arg,
                               uint32_t* d_roi_escapes) {
    NdtCtl c;
    ndt_initial_ctl(&c, T0, p, step_size, trans_eps, max_iters, no_replay_arg);
    NdtCtlArg a;
    memcpy(a.w, &c, sizeof c);
    hipLaunchKernelGGL(ndt_ctl_store_kernel, dim3(1), dim3(512), 0, s, d_ctl, a, d_roi_escapes);
    return hipGetLastError();
}
// the same state as a rider of another launch (pcr_internal.h: BlobStore; `zero` = the escape counter is filled in by whoever launches it)
void ndt_ctl_init_blob(BlobStore* b, NdtCtl* d_ctl, const NdtPose& T0, const double p[6], double step_size, double trans_eps, int max_iters, int no_replay_arg) {
    static_assert(sizeof(NdtCtl) <= sizeof(b->w), "NdtCtl must fit a BlobStore");
    NdtCtl c;
    ndt_initial_ctl(&c, T0, p, step_size, trans_eps, max_iters, no_replay_arg);
    memcpy(b->w, &c, sizeof c);
    b->dst = reinterpret_cast<uint32_t*>(d_ctl); b->zero = nullptr; b->n = (uint32_t)(sizeof c / 4); b->pad = 0;
}
hipError_t ndt_launch_pass(const NdtArgs& a, NdtCtl* d_ctl, NdtOut* d_out, hipStream_t s, double seq) {
    const uint32_t nb = ndt_blocks(a.n_src);
    hipLaunchKernelGGL(ndt_pass_kernel, dim3(nb), dim3(kNdtBlock), 0, s, a, d_ctl);
    hipLaunchKernelGGL(ndt_fold_ctl_kernel, dim3(1), dim3(768), 0, s, a.partials, nb, d_ctl, a.hdr, d_out, seq, (const uint32_t*)a.roi_escapes);
    return hipGetLastError();
}
// launch `index` of the one-launch-per-pass loop: d_ctl2 = two NdtCtl, d_rows2 = two buffers of kProRows * 48 doubles
hipError_t ndt_launch_pass_pro(const NdtArgs& a_in, NdtCtl* d_ctl2, double* d_rows2, NdtOut* d_out, hipStream_t s, double seq, int index,
                               hipEvent_t start, hipEvent_t stop) {
    uint32_t nb = (a_in.n_src + kProBlock - 1) / kProBlock;
    nb = nb < 1 ? 1 : (nb > (uint32_t)kProRows ? (uint32_t)kProRows : nb);
    NdtArgs a = a_in;
    a.partials = d_rows2 + (size_t)(index & 1) * kProRows * 48;
    NdtProArgs pa;
    pa.rows_prev = d_rows2 + (size_t)((index + 1) & 1) * kProRows * 48;
    pa.ctl_prev = index == 0 ? d_ctl2 : d_ctl2 + ((index + 1) & 1);
    pa.ctl_next = d_ctl2 + (index & 1);
    pa.out = d_out; pa.seq = seq; pa.rows_prev_n = nb; pa.first = index == 0 ? 1 : 0;
    // (start / stop: events the packet processor stamps at the kernel's own begin and end -- profiling passes)
    if (start || stop) hipExtLaunchKernelGGL(ndt_pass_pro_kernel, dim3(nb), dim3(kProBlock), 0, s, start, stop, 0, a, pa);
    else hipLaunchKernelGGL(ndt_pass_pro_kernel, dim3(nb), dim3(kProBlock), 0, s, a, pa);
    return hipGetLastError();
}

// one pass of the sharded device-resident loop, in three pieces around the caller's all-reduce of d_sums48
hipError_t ndt_launch_pass_fold(const NdtArgs& a, NdtCtl* d_ctl, double* d_sums48, hipStream_t s) {
    const uint32_t nb = ndt_blocks(a.n_src);
    hipLaunchKernelGGL(ndt_pass_kernel, dim3(nb), dim3(kNdtBlock), 0, s, a, d_ctl);
    hipLaunchKernelGGL(ndt_fold_kernel, dim3(1), dim3(768), 0, s, a.partials, nb, d_ctl, d_sums48);
    return hipGetLastError();
}
// one pass of the sharded loop over the peer exchange: the evaluation, then fold + exchange + controller step in one launch
hipError_t ndt_launch_pass_peer(const NdtArgs& a, NdtCtl* d_ctl, const PeerComm& pc, double xseq, NdtOut* d_out, hipStream_t s, double seq, int batch_mark) {
    const uint32_t nb = ndt_blocks(a.n_src);
    hipLaunchKernelGGL(ndt_pass_kernel, dim3(nb), dim3(kNdtBlock), 0, s, a, d_ctl);
    hipLaunchKernelGGL(ndt_fold_exchange_ctl_kernel, dim3(1), dim3(768), 0, s, a.partials, nb, d_ctl, pc, xseq, a.hdr, d_out, seq, batch_mark, (const uint32_t*)a.roi_escapes);
    return hipGetLastError();
}
hipError_t ndt_launch_ctl(const NdtArgs& a, NdtCtl* d_ctl, const double* d_sums48, NdtOut* d_out, hipStream_t s, double seq, int batch_mark) {
    hipLaunchKernelGGL(ndt_ctl_kernel, dim3(1), dim3(64), 0, s, d_ctl, d_sums48, a.hdr, d_out, seq, batch_mark, (const uint32_t*)a.roi_escapes);
    return hipGetLastError();
}

hipError_t ndt_launch_hessian(const NdtArgs& a, const NdtPose& T, const NdtAngles& ang, double* d_out48, hipStream_t s, double seq) {
    const uint32_t nb = ndt_blocks(a.n_src);
    hipLaunchKernelGGL(ndt_hessian_kernel, dim3(nb), dim3(kNdtBlock), 0, s, a, T, ang);
    hipLaunchKernelGGL(ndt_sum_partials_kernel, dim3(1), dim3(768), 0, s, a.partials, nb, d_out48, seq);
    return hipGetLastError();
}

}  // namespace pcr
