// loam.hip -- LOAM scan-to-map Gauss-Newton on gfx950 (hand-written HIP).
//
// One launch = one Gauss-Newton iteration of the reference's
//   PCR::LoamRegister::scan2Map            PCR/src/LoamRegister.cpp:99-223
// with its per-point body fused into a single kernel:
//   L2 transform + f32 cast                LoamRegister.cpp:126-130
//   L3 exact 5-NN + squared-distance gate  LoamRegister.cpp:47-72 (nanoflann there, uniform grid here)
//   L4 plane LS (col-piv Householder QR)   LoamRegister.cpp:29-45
//   L5 residual, weight, weight gate       LoamRegister.cpp:144-151, LoamRegister.hpp:75-77
//   L6 Jacobian row s*n^T [I | -p^]        LoamRegister.cpp:153-159, manifolds.hpp:63-68
//   L7 J^T J / J^T r reduction             LoamRegister.cpp:170-188 (omp critical + dense GEMM there)
// and, in the PROLOGUE of the next launch, executed redundantly by every block
// (bitwise identical, so no broadcast and no extra kernel boundary):
//   L8 LDLT solve, convergence test, SE(3) exp update   LoamRegister.cpp:198-216, manifolds.hpp:33-60
// The last launch (loam_finalize_kernel) runs only that prologue plus
//   L9 T2SE3 re-orthonormalisation        LoamRegister.cpp:220, trans.hpp:54-65
//
// Arithmetic is f64 with contraction off (-ffp-contract=off), in the reference's
// operation order, so per-point results are bit-identical to the CPU oracle
// wherever only IEEE +,-,*,/,sqrt are involved.
#include "pcr_internal.h"
#include "small_math.h"

namespace pcr {

// ------------------------------------------------------------------------------
// per-lane exact 5-NN on the grid
// ------------------------------------------------------------------------------
struct Knn5 {
    double d[5];
    uint32_t idx[5];   // original target index: the tie-break key
    uint32_t pos[5];   // position in the cell-sorted array
};

// (d, idx) lexicographic order, branch-free: distance ties are broken on the original index
__device__ __forceinline__ bool knn_less(double d, uint32_t i, double d2, uint32_t i2) {
    return (d < d2) | ((d == d2) & (i < i2));
}

__device__ __forceinline__ void knn_insert(Knn5& s, double d, uint32_t idx, uint32_t pos) {
    // caller guarantees (d,idx) < slot 4; straight-line sorted insert, selects only
    const bool c3 = knn_less(d, idx, s.d[3], s.idx[3]);
    const bool c2 = knn_less(d, idx, s.d[2], s.idx[2]);
    const bool c1 = knn_less(d, idx, s.d[1], s.idx[1]);
    const bool c0 = knn_less(d, idx, s.d[0], s.idx[0]);
    s.d[4] = c3 ? s.d[3] : d;   s.idx[4] = c3 ? s.idx[3] : idx;   s.pos[4] = c3 ? s.pos[3] : pos;
    s.d[3] = c3 ? (c2 ? s.d[2] : d) : s.d[3];   s.idx[3] = c3 ? (c2 ? s.idx[2] : idx) : s.idx[3];   s.pos[3] = c3 ? (c2 ? s.pos[2] : pos) : s.pos[3];
    s.d[2] = c2 ? (c1 ? s.d[1] : d) : s.d[2];   s.idx[2] = c2 ? (c1 ? s.idx[1] : idx) : s.idx[2];   s.pos[2] = c2 ? (c1 ? s.pos[1] : pos) : s.pos[2];
    s.d[1] = c1 ? (c0 ? s.d[0] : d) : s.d[1];   s.idx[1] = c1 ? (c0 ? s.idx[0] : idx) : s.idx[1];   s.pos[1] = c1 ? (c0 ? s.pos[0] : pos) : s.pos[1];
    s.d[0] = c0 ? d : s.d[0];   s.idx[0] = c0 ? idx : s.idx[0];   s.pos[0] = c0 ? pos : s.pos[0];
}

// Per-block LDS scratch of the search: the 9 row ranges of every lane.
struct KnnShared {
    uint32_t rs[9][256];
    uint32_t re[9][256];
};

// Rows of the 3x3 (y,z) neighbourhood in the order centre, faces, corners, so that the lower
// bound on a row's distance prunes late rows.  4-bit code per row: (dy+1) | (dz+1) << 2.
static constexpr unsigned long long kRowOrder = 0xA82091645ull;
__device__ __forceinline__ constexpr int row_dy(int r) { return (int)((kRowOrder >> (4 * r)) & 3) - 1; }
__device__ __forceinline__ constexpr int row_dz(int r) { return (int)((kRowOrder >> (4 * r + 2)) & 3) - 1; }

struct KnnQuery {
    double qx, qy, qz;          // query (a float widened to double)
    double ylo, yhi, zlo, zhi;  // distance to the faces of the query's cell
    double l6;                  // lower bound of the squared distance of every point NOT in the top 5
};

__device__ __forceinline__ double row_bound(const KnnQuery& q, int r) {
    const unsigned code = (unsigned)((kRowOrder >> (4 * r)) & 15);
    const int dy = (int)(code & 3) - 1, dz = (int)(code >> 2) - 1;
    const double gy = dy < 0 ? q.ylo : (dy > 0 ? q.yhi : 0.0);
    const double gz = dz < 0 ? q.zlo : (dz > 0 ? q.zhi : 0.0);
    // every point of the row is at least sqrt(gy^2+gz^2) away (same rounding order as d below)
    return gy * gy + gz * gz;
}

// One candidate.  No single-precision pre-screen: with 64 independent searches in lockstep some
// lane nearly always needs the exact distance, so the screen would only add instructions.
__device__ __forceinline__ void knn_consider(Knn5& s, KnnQuery& q, const float4 p, uint32_t pos) {
    const double dx = q.qx - (double)p.x, dy = q.qy - (double)p.y, dz = q.qz - (double)p.z;
    double d = dx * dx;      // nanoflann L2_Simple_Adaptor::evalMetric order (nanoflann.hpp:523-535)
    d += dy * dy;
    d += dz * dz;
    const uint32_t idx = __float_as_uint(p.w);
    const bool take = knn_less(d, idx, s.d[4], s.idx[4]);
    // whatever leaves or never enters the top 5 bounds the "6th neighbour" from below
    const double out = take ? s.d[4] : d;
    q.l6 = out < q.l6 ? out : q.l6;
    if (take) knn_insert(s, d, idx, pos);
}

struct KnnCursor { int r; uint32_t j, e; };

// move the cursor to the next row that still has candidates (rows whose lower bound already
// exceeds the current 5th distance are skipped; the bound only shrinks, so this is safe ahead of time)
__device__ __forceinline__ bool knn_advance(KnnCursor& c, const KnnShared& sh, KnnQuery& q, const Knn5& s, int tid) {
    while (c.j >= c.e) {
        if (++c.r > 8) return false;
        c.j = sh.rs[c.r][tid];
        c.e = sh.re[c.r][tid];
        const double b = row_bound(q, c.r);
        if (b > s.d[4]) { c.j = c.e; q.l6 = b < q.l6 ? b : q.l6; }   // skipped points are at least sqrt(b) away
    }
    return true;
}

static constexpr int kChunk = 8;   // candidates per lane per step (the sorted array is padded by kChunk)

// Exact 5 nearest target points with squared distance <= max_sq (ties on the original
// index), searching the 3x3x3 cell block as 9 contiguous x-runs.  Returns false when the
// query lies outside the searchable grid.  On return s.d[4] < max_sq  <=>  the reference's
// gate pointSearchSqDis[4] < mKdtreeMaxSearchDist (LoamRegister.cpp:59) passes.
//
// Memory-level parallelism is explicit: the 18 range loads of a query are issued together, and
// candidates stream in chunks of kChunk float4 loads with the next chunk already in flight.
__device__ __forceinline__ bool knn5_grid(const GridHeader& h, const float4* __restrict__ pts,
                                          const uint32_t* __restrict__ cell_start, double qx, double qy, double qz,
                                          double max_sq, Knn5& s, KnnShared& sh, bool active, double* l6_out, bool keep) {
    const int tid = threadIdx.x;
    if (!keep) {
#pragma unroll
        for (int j = 0; j < 5; ++j) { s.d[j] = max_sq; s.idx[j] = 0xffffffffu; s.pos[j] = 0; }
    }
    // cell coordinates (exact: q is a float widened to double, origin a multiple of cell)
    const double rx = qx - h.origin[0], ry = qy - h.origin[1], rz = qz - h.origin[2];
    const double fx = floor(rx * h.inv_cell), fy = floor(ry * h.inv_cell), fz = floor(rz * h.inv_cell);
    // queries in the outermost cell (or beyond, or NaN) are >= one cell away from every point
    const bool inside = active && (fx >= 1.0 && fx <= (double)(h.dims[0] - 2) && fy >= 1.0 && fy <= (double)(h.dims[1] - 2) &&
                                   fz >= 1.0 && fz <= (double)(h.dims[2] - 2));
    if (inside) {
        const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
        uint32_t ra[9], rb[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {   // 18 independent loads in flight
            const uint32_t key = ((uint32_t)(cz + row_dz(r)) * (uint32_t)h.dims[1] + (uint32_t)(cy + row_dy(r))) * (uint32_t)h.dims[0] + (uint32_t)cx;
            ra[r] = cell_start[key - 1]; rb[r] = cell_start[key + 2];
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) { sh.rs[r][tid] = ra[r]; sh.re[r][tid] = rb[r]; }
    } else {
#pragma unroll
        for (int r = 0; r < 9; ++r) { sh.rs[r][tid] = 0u; sh.re[r][tid] = 0u; }
    }
    KnnQuery q;
    q.qx = qx; q.qy = qy; q.qz = qz;
    q.ylo = ry - fy * h.cell; q.yhi = (fy + 1.0) * h.cell - ry;
    q.zlo = rz - fz * h.cell; q.zhi = (fz + 1.0) * h.cell - rz;
    q.l6 = max_sq;   // points outside the 3x3x3 block are >= one cell (>= sqrt(max_sq)) away
    // (each lane reads back only what it wrote itself: no barrier needed)
    KnnCursor cur{-1, 0u, 0u};
    bool has = knn_advance(cur, sh, q, s, tid);
    float4 c[kChunk];
#pragma unroll
    for (int i = 0; i < kChunk; ++i) c[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has) {
#pragma unroll
        for (int i = 0; i < kChunk; ++i) c[i] = pts[cur.j + i];
    }
    while (has) {
        KnnCursor nxt = cur;
        nxt.j += kChunk;
        const bool has_n = knn_advance(nxt, sh, q, s, tid);
        float4 n[kChunk];
#pragma unroll
        for (int i = 0; i < kChunk; ++i) n[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_n) {
#pragma unroll
            for (int i = 0; i < kChunk; ++i) n[i] = pts[nxt.j + i];
        }
#pragma unroll
        for (int i = 0; i < kChunk; ++i) {
            if (cur.j + i < cur.e) knn_consider(s, q, c[i], cur.j + i);
        }
#pragma unroll
        for (int i = 0; i < kChunk; ++i) c[i] = n[i];
        cur = nxt; has = has_n;
    }
    *l6_out = q.l6;
    return inside;
}

// ------------------------------------------------------------------------------
// Temporal coherence between Gauss-Newton iterations (exact, not approximate).
// After a full search we keep, per scan point: the query position q0, the sorted-array
// positions of its 5 neighbours and L6, a lower bound of the squared distance from q0 to
// every OTHER target point.  At the next iteration the query has moved by delta = |q - q0|.
// For any other point p: |p - q| >= |p - q0| - delta >= sqrt(L6) - delta, so if the farthest of
// the 5 cached neighbours is strictly closer to q than that, the cached set IS the exact
// 5-NN set of q and only its order has to be recomputed.  Otherwise the full search runs.
// ------------------------------------------------------------------------------
struct NnCacheEntry {          // 32 bytes
    uint32_t pos[5];
    float l6;                  // rounded down
    uint32_t valid;
    uint32_t pad;
};

__device__ __forceinline__ void knn_cswap(Knn5& s, int i, int j) {
    const bool sw = knn_less(s.d[j], s.idx[j], s.d[i], s.idx[i]);
    const double di = s.d[i], dj = s.d[j];
    const uint32_t ii = s.idx[i], ij = s.idx[j], pi = s.pos[i], pj = s.pos[j];
    s.d[i] = sw ? dj : di; s.d[j] = sw ? di : dj;
    s.idx[i] = sw ? ij : ii; s.idx[j] = sw ? ii : ij;
    s.pos[i] = sw ? pj : pi; s.pos[j] = sw ? pi : pj;
}

// returns true when the cached neighbours were proven to be the exact 5-NN of (qx,qy,qz)
__device__ __forceinline__ bool knn5_from_cache(const float4* __restrict__ pts, const NnCacheEntry& ce, const float4 q0,
                                                double qx, double qy, double qz, Knn5& s, float* l6_new) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const float4 p = pts[ce.pos[j]];
        const double dx = qx - (double)p.x, dy = qy - (double)p.y, dz = qz - (double)p.z;
        double d = dx * dx;
        d += dy * dy;
        d += dz * dz;
        s.d[j] = d; s.idx[j] = __float_as_uint(p.w); s.pos[j] = ce.pos[j];
    }
    // 9-comparator sorting network for 5 keys
    knn_cswap(s, 0, 1); knn_cswap(s, 3, 4); knn_cswap(s, 2, 4); knn_cswap(s, 2, 3); knn_cswap(s, 0, 3);
    knn_cswap(s, 0, 2); knn_cswap(s, 1, 4); knn_cswap(s, 1, 3); knn_cswap(s, 1, 2);
    const float ex = (float)qx - q0.x, ey = (float)qy - q0.y, ez = (float)qz - q0.z;
    // single precision with 1e-6 relative margins on every term (rounding is < 2e-7)
    const float delta = sqrtf(fmaf(ex, ex, fmaf(ey, ey, ez * ez))) * 1.000001f + 1e-7f;
    const float r5 = sqrtf((float)s.d[4]) * 1.000001f;
    const float l6 = sqrtf(ce.l6) * 0.999999f;
    const float slack = l6 - delta;
    *l6_new = slack > 0.f ? slack * slack * 0.999999f : 0.f;
    return r5 < slack * 0.999999f;
}

// ------------------------------------------------------------------------------
// plane fit: Eigen::ColPivHouseholderQR<Matrix<double,5,3>>::solve(-1) restated
// (LoamRegister.cpp:29-35), all indices static so everything stays in registers.
// ------------------------------------------------------------------------------
__device__ __forceinline__ void swap_d(double& a, double& b) { const double t = a; a = b; b = t; }

__device__ __forceinline__ void plane_qr_solve(double a[5][3], double x[3]) {
    double c[5] = {-1.0, -1.0, -1.0, -1.0, -1.0};
    int perm[3] = {0, 1, 2};
    double tau[3] = {0, 0, 0};
    double nu[3], nd[3];
    double maxn = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double s = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) s += a[i][j] * a[i][j];
        nd[j] = nu[j] = sqrt(s);
        if (nu[j] > maxn) maxn = nu[j];
    }
    const double eps = 2.220446049250313e-16;
    const double thr_helper = (maxn * eps) * (maxn * eps) / 5.0;
    const double downdate_thr = 1.4901161193847656e-08;  // sqrt(eps)
    int nonzero = 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int big = k; double bign = nu[k];
#pragma unroll
        for (int j = k + 1; j < 3; ++j) if (nu[j] > bign) { bign = nu[j]; big = j; }
        const double big_sq = bign * bign;
        if (nonzero == 3 && big_sq < thr_helper * (double)(5 - k)) nonzero = k;
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            if (big == j) {
#pragma unroll
                for (int i = 0; i < 5; ++i) swap_d(a[i][k], a[i][j]);
                swap_d(nu[k], nu[j]); swap_d(nd[k], nd[j]);
                const int t = perm[k]; perm[k] = perm[j]; perm[j] = t;
            }
        }
        double tail_sq = 0;
#pragma unroll
        for (int i = k + 1; i < 5; ++i) tail_sq += a[i][k] * a[i][k];
        const double c0 = a[k][k];
        double beta;
        if (tail_sq <= 2.2250738585072014e-308) {
            tau[k] = 0; beta = c0;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) a[i][k] = 0;
        } else {
            beta = sqrt(c0 * c0 + tail_sq);
            if (c0 >= 0) beta = -beta;
            const double den = c0 - beta;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) a[i][k] /= den;
            tau[k] = (beta - c0) / beta;
        }
        a[k][k] = beta;
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            double tmp = 0;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) tmp += a[i][k] * a[i][j];
            tmp += a[k][j];
            a[k][j] -= tau[k] * tmp;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) a[i][j] -= tau[k] * a[i][k] * tmp;
        }
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            if (nu[j] != 0) {
                double temp = fabs(a[k][j]) / nu[j];
                temp = (1.0 + temp) * (1.0 - temp);
                if (temp < 0) temp = 0;
                const double r = nu[j] / nd[j];
                const double temp2 = temp * r * r;
                if (temp2 <= downdate_thr) {
                    double s = 0;
#pragma unroll
                    for (int i = k + 1; i < 5; ++i) s += a[i][j] * a[i][j];
                    nd[j] = sqrt(s); nu[j] = nd[j];
                } else {
                    nu[j] *= sqrt(temp);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k < nonzero) {
            double tmp = 0;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) tmp += a[i][k] * c[i];
            tmp += c[k];
            c[k] -= tau[k] * tmp;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) c[i] -= tau[k] * a[i][k] * tmp;
        }
    }
    double y[3] = {0, 0, 0};
#pragma unroll
    for (int i = 2; i >= 0; --i) {
        if (i < nonzero) {
            double s = c[i];
#pragma unroll
            for (int j = i + 1; j < 3; ++j) if (j < nonzero) s -= a[i][j] * y[j];
            y[i] = s / a[i][i];
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double v = (i < nonzero) ? y[i] : 0.0;
#pragma unroll
        for (int j = 0; j < 3; ++j) if (perm[i] == j) x[j] = v;
    }
}

// ------------------------------------------------------------------------------
// one scan point: returns status (0 accepted, 1 k-NN gate, 2 plane gate, 3 weight gate,
// 4 outside this rank's query tile / no point);  row[0..5] = s*[n ; p x n], row[6] = s*d
// ------------------------------------------------------------------------------
__device__ __forceinline__ int loam_point(const LoamArgs& a, const GridHeader& h, const double* __restrict__ pose,
                                          const float* __restrict__ sp, bool valid, KnnShared& sh, double row[7],
                                          uint32_t nn_idx[5], uint32_t qi, bool use_cache) {
    float sx = 0.f, sy = 0.f, sz = 0.f;
    if (valid) { sx = sp[0]; sy = sp[1]; sz = sp[2]; }
    const double ox = (double)sx, oy = (double)sy, oz = (double)sz;
    // LoamRegister.cpp:126-130: Isometry3d * Vector4d in f64, then cast to f32
    const float px = (float)(pose[0] * ox + pose[4] * oy + pose[8] * oz + pose[12] * 1.0);
    const float py = (float)(pose[1] * ox + pose[5] * oy + pose[9] * oz + pose[13] * 1.0);
    const float pz = (float)(pose[2] * ox + pose[6] * oy + pose[10] * oz + pose[14] * 1.0);
    const double qx = (double)px, qy = (double)py, qz = (double)pz;
    bool active = valid && !h.empty && !h.overflow;
    if (a.use_tile) {
        const bool in_tile = qx >= a.tile_lo[0] && qx < a.tile_hi[0] && qy >= a.tile_lo[1] && qy < a.tile_hi[1] &&
                             qz >= a.tile_lo[2] && qz < a.tile_hi[2];
        if (!in_tile) { active = false; valid = false; }
    }
    Knn5 s;
    bool searched = false;
    NnCacheEntry* cache = a.nn_cache ? &a.nn_cache[qi] : nullptr;
    float l6f = 0.f;
    if (cache && use_cache && active) {
        const NnCacheEntry ce = *cache;
        if (ce.valid) searched = knn5_from_cache(a.grid.pts, ce, a.q_cache[qi], qx, qy, qz, s, &l6f);
    }
    const bool hit = searched;
    double l6 = 0.0;
    // (the full search is entered by the whole wave; lanes served by the cache sit it out)
    const bool full = knn5_grid(h, a.grid.pts, a.grid.cell_start, qx, qy, qz, a.c.knn_max_sq, s, sh,
                                active && !hit && !(a.ablate & 1), &l6, hit);
    searched = hit || full;
    if (cache && active) {
        NnCacheEntry ce;
#pragma unroll
        for (int j = 0; j < 5; ++j) ce.pos[j] = s.pos[j];
        // all five slots must hold real points for the entry to be reusable
        ce.valid = searched && s.idx[4] != 0xffffffffu;
        ce.l6 = hit ? l6f : (float)l6 * 0.999999f;
        ce.pad = hit;
        *cache = ce;
        a.q_cache[qi] = make_float4((float)qx, (float)qy, (float)qz, 0.f);
    }
    if (!valid) return 4;
    if (!searched) return 1;
#pragma unroll
    for (int j = 0; j < 5; ++j) nn_idx[j] = s.idx[j];
    if (!(s.d[4] < a.c.knn_max_sq)) return 1;
    double A[5][3], Aq[5][3];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const float4 p = a.grid.pts[s.pos[j]];
        A[j][0] = (double)p.x; A[j][1] = (double)p.y; A[j][2] = (double)p.z;
        Aq[j][0] = A[j][0]; Aq[j][1] = A[j][1]; Aq[j][2] = A[j][2];
    }
    if (a.ablate & 2) return 2;
    double x[3];
    plane_qr_solve(Aq, x);
    const double xn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    // LoamRegister.cpp:38-43
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const double dot = x[0] * A[i][0] + x[1] * A[i][1] + x[2] * A[i][2];
        if (fabs(dot + 1.0) > a.c.plane_thresh * xn) ok = false;
    }
    if (!ok) return 2;
    const double dist = (qx * x[0] + qy * x[1] + qz * x[2] + 1.0) / xn;   // LoamRegister.hpp:75-77
    const float r2 = sx * sx + sy * sy + sz * sz;                           // LoamRegister.cpp:147-148 (float)
    const float rr = sqrtf(sqrtf(r2));
    const double w = 1 - 0.9 * fabs(dist) / (double)rr;
    if (!(w > a.c.point_thresh)) return 3;
    const double n0 = x[0] / xn, n1 = x[1] / xn, n2 = x[2] / xn;            // LoamRegister.hpp:70-73
    const double s0 = w * n0, s1 = w * n1, s2 = w * n2;
    row[0] = s0; row[1] = s1; row[2] = s2;
    row[3] = s1 * (-qz) + s2 * qy;       // s*n^T * (-skew(p)), manifolds.hpp:63-68, matrix.hpp:13-18
    row[4] = s0 * qz + s2 * (-qx);
    row[5] = s0 * (-qy) + s1 * qx;
    row[6] = w * dist;
    return 0;
}

// ------------------------------------------------------------------------------
// prologue: fold the previous launch's partial sums, solve, update the pose.
// Every block runs it and obtains bit-identical results.  Returns true when this
// launch must not linearise (loop finished).
// ------------------------------------------------------------------------------
struct Prologue {
    double pose[16];
    int done;
};

__device__ bool loam_prologue(const LoamArgs& a, int k, double* sh_sum /* 8*32 */, Prologue* sh) {
    const LoamState* prev = &a.state[(k + 1) & 1];
    LoamState* cur = &a.state[k & 1];
    const int t = threadIdx.x;
    if (k == 0) {
        if (t < 16) sh->pose[t] = a.init_pose[t];
        if (t == 0) sh->done = 0;
        if (blockIdx.x == 0 && t == 0) {
            for (int i = 0; i < 16; ++i) cur->pose[i] = a.init_pose[i];
            cur->done = 0; cur->converged = 0; cur->iters_run = 0; cur->fail = 0;
        }
        __syncthreads();
        return false;
    }
    if (prev->done) {
        if (t < 16) sh->pose[t] = prev->pose[t];
        if (t == 0) sh->done = 1;
        if (blockIdx.x == 0 && t == 0) { *cur = *prev; }
        __syncthreads();
        return true;
    }
    if (a.ablate & 4) {
        if (t < 16) sh->pose[t] = prev->pose[t];
        if (t == 0) sh->done = k >= a.c.iters;
        if (blockIdx.x == 0 && t == 0) { *cur = *prev; cur->done = k >= a.c.iters; cur->iters_run = k; }
        __syncthreads();
        return k >= a.c.iters;
    }
    // fixed-order reduction of the partial sums of launch k-1
    const int comp = t & 31, slice = t >> 5;
    double acc = 0.0;
    if (a.reduced) {
        if (slice == 0) acc = a.reduced[comp];
    } else {
        // 8 independent loads in flight per batch; the additions keep their fixed order
        const double* part = a.partials + (size_t)((k + 1) & 1) * kMaxPartials * kAccum + comp;
        for (uint32_t b0 = slice; b0 < a.n_partials; b0 += 64) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t b = b0 + 8 * u;
                v[u] = b < a.n_partials ? part[(size_t)b * kAccum] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
    }
    sh_sum[slice * 32 + comp] = acc;
    __syncthreads();
    if (t < 32) {
        double v = sh_sum[t];
#pragma unroll
        for (int s = 1; s < 8; ++s) v += sh_sum[s * 32 + t];
        sh_sum[t] = v;
    }
    __syncthreads();
    if (t == 0) {
        double JtJ[36], rhs[6], x[6] = {0, 0, 0, 0, 0, 0};
        int q = 0;
        for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { JtJ[r * 6 + c] = JtJ[c * 6 + r] = sh_sum[q++]; }
        for (int r = 0; r < 6; ++r) rhs[r] = -sh_sum[21 + r];
        const double n = sh_sum[27];
        int done = 0, conv = 0, fail = 0;
        double pose[16];
        for (int i = 0; i < 16; ++i) pose[i] = prev->pose[i];
        if (n < 6.0) { done = 1; fail = 1; }                      // LoamRegister.cpp:173-176
        else {
            ldlt6_solve(JtJ, rhs, x);                              // LoamRegister.cpp:198
            const double np = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
            const double nr = sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5]);
            if (a.c.early_exit && np <= a.c.pos_conv && nr <= a.c.rot_conv) { done = 1; conv = 1; }   // :202-206
            else {
                double E[16], out[16];
                se3_exp(x, E);                                     // :213-216
                for (int c = 0; c < 4; ++c)
                    for (int r = 0; r < 4; ++r) {
                        double s = 0;
                        for (int kk = 0; kk < 4; ++kk) s += E[kk * 4 + r] * pose[c * 4 + kk];
                        out[c * 4 + r] = s;
                    }
                for (int i = 0; i < 16; ++i) pose[i] = out[i];
            }
        }
        if (k >= a.c.iters) done = 1;
        for (int i = 0; i < 16; ++i) sh->pose[i] = pose[i];
        sh->done = done;
        if (blockIdx.x == 0) {
            for (int i = 0; i < 16; ++i) cur->pose[i] = pose[i];
            cur->done = done; cur->converged = conv; cur->fail = fail; cur->iters_run = k;
            if (a.trace) {
                LoamTrace* tr = &a.trace[k - 1];
                for (int i = 0; i < 36; ++i) tr->JtJ[i] = JtJ[i];
                for (int i = 0; i < 6; ++i) { tr->JtE[i] = -rhs[i]; tr->x[i] = x[i]; }
                tr->n = (int64_t)n;
            }
        }
    }
    __syncthreads();
    return sh->done != 0;
}

// ------------------------------------------------------------------------------
// the iteration kernel
// ------------------------------------------------------------------------------
static constexpr int kRowStride = 258;   // doubles per component row in LDS (256 + pad: conflict-free reads)

__global__ __launch_bounds__(256) void loam_iterate_kernel(const LoamArgs a, const int k) {
    __shared__ double sh_sum[8 * 32];
    __shared__ Prologue sh_pro;
    __shared__ KnnShared sh_knn;
    __shared__ double sh_rows[8 * kRowStride];   // [component][point]: s*J row (6), s*d, accepted flag
    if (loam_prologue(a, k, sh_sum, &sh_pro)) return;
    const GridHeader h = *a.grid.hdr;
    double pose[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) pose[i] = sh_pro.pose[i];
    const int tid = threadIdx.x;

    // XCD-aware mapping: consecutive logical blocks (adjacent lidar rings) share an XCD's L2
    uint32_t blk = blockIdx.x;
    if ((gridDim.x & 7u) == 0) blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);

    // This thread accumulates ONE entry of the normal equations over a 32-point chunk:
    // e < 21: JtJ(er,ec) upper triangle; 21..26: JtE(er) = sum row[er]*row[6]; 27: accepted count.
    const int e = tid & 31, ch = tid >> 5;
    int er = 7, ec = 7;
    {
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
            for (int c = r; c < 6; ++c) { if (q == e) { er = r; ec = c; } ++q; }
        }
        if (e >= 21 && e < 27) { er = e - 21; ec = 6; }
    }
    double acc = 0.0;
    for (uint32_t base = blk * 256; base < a.n_src; base += gridDim.x * 256) {
        const uint32_t q = base + tid;
        const bool valid = q < a.n_src;
        double row[7] = {0, 0, 0, 0, 0, 0, 0};
        uint32_t nn[5] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        const int st = loam_point(a, h, pose, a.src + (size_t)(valid ? q : 0) * a.src_stride, valid, sh_knn, row, nn, q, k > 0);
        if (valid && (a.dbg_status || a.dbg_nn || a.dbg_rows)) {
            const float* spq = a.src + (size_t)q * a.src_stride;
            const size_t oi = a.src_indexed ? (size_t)__float_as_uint(spq[3]) : (size_t)q;   // original scan index
            if (a.dbg_status) a.dbg_status[oi] = (int8_t)st;
            if (a.dbg_nn) { for (int j = 0; j < 5; ++j) a.dbg_nn[oi * 5 + j] = (int32_t)nn[j]; }
            if (a.dbg_rows) { for (int j = 0; j < 7; ++j) a.dbg_rows[oi * 7 + j] = st == 0 ? row[j] : 0.0; }
        }
#pragma unroll
        for (int c = 0; c < 7; ++c) sh_rows[c * kRowStride + tid] = st == 0 ? row[c] : 0.0;
        sh_rows[7 * kRowStride + tid] = st == 0 ? 1.0 : 0.0;
        __syncthreads();
        if (e < 28) {
            const double* ra = sh_rows + er * kRowStride + ch * 32;
            const double* rb = sh_rows + ec * kRowStride + ch * 32;
#pragma unroll 8
            for (int i = 0; i < 32; ++i) acc += ra[i] * rb[i];
        }
        __syncthreads();
    }
    sh_sum[ch * 32 + e] = e < 28 ? acc : 0.0;
    __syncthreads();
    if (tid < 32) {
        double v = sh_sum[tid];
#pragma unroll
        for (int c = 1; c < 8; ++c) v += sh_sum[c * 32 + tid];
        a.partials[((size_t)(k & 1) * kMaxPartials + blockIdx.x) * kAccum + tid] = v;
    }
}

// last launch: prologue only, then T2SE3 and the output pose (LoamRegister.cpp:220)
__global__ __launch_bounds__(256) void loam_finalize_kernel(const LoamArgs a, const int k) {
    __shared__ double sh_sum[8 * 32];
    __shared__ Prologue sh_pro;
    loam_prologue(a, k, sh_sum, &sh_pro);
    if (threadIdx.x == 0) {
        double T[16];
        for (int i = 0; i < 16; ++i) T[i] = sh_pro.pose[i];
        t2se3(T);
        const LoamState* cur = &a.state[k & 1];   // written by this thread in the prologue
        LoamResult* r = a.result;
        for (int i = 0; i < 16; ++i) r->pose[i] = T[i];
        r->converged = cur->converged; r->iters_run = cur->iters_run; r->fail = cur->fail;
        r->grid_overflow = a.grid.hdr->overflow; r->grid_empty = a.grid.hdr->empty; r->grid_cells = a.grid.hdr->n_cells;
        r->pad = 1;   // completion marker
    }
}

// sharded (multi-GPU) mode: fold this rank's partial sums into kAccum doubles so that
// RCCL can all-reduce them before the next launch's prologue reads a.reduced
__global__ __launch_bounds__(256) void loam_reduce_kernel(const LoamArgs a, const int k, double* __restrict__ out) {
    __shared__ double sh_sum[8 * 32];
    const int t = threadIdx.x, comp = t & 31, slice = t >> 5;
    const LoamState* st = &a.state[k & 1];
    double acc = 0.0;
    if (!st->done) {
        const double* part = a.partials + (size_t)(k & 1) * kMaxPartials * kAccum;
        for (uint32_t b = slice; b < a.n_partials; b += 8) acc += part[(size_t)b * kAccum + comp];
    }
    sh_sum[slice * 32 + comp] = acc;
    __syncthreads();
    if (t < 32) {
        double v = sh_sum[t];
#pragma unroll
        for (int s = 1; s < 8; ++s) v += sh_sum[s * 32 + t];
        out[t] = v;
    }
}

uint32_t loam_grid_blocks(uint32_t n_src) {
    uint32_t b = (n_src + 255) / 256;
    if (b < 1) b = 1;
    if (b > (uint32_t)kMaxPartials) b = kMaxPartials;
    return b;
}

hipError_t loam_launch_iteration(const LoamArgs& a, int k, hipStream_t s) {
    hipLaunchKernelGGL(loam_iterate_kernel, dim3(a.n_partials), dim3(256), 0, s, a, k);
    return hipGetLastError();
}
hipError_t loam_launch_finalize(const LoamArgs& a, int k, hipStream_t s) {
    hipLaunchKernelGGL(loam_finalize_kernel, dim3(1), dim3(256), 0, s, a, k);
    return hipGetLastError();
}
hipError_t loam_launch_reduce(const LoamArgs& a, int k, double* d_out, hipStream_t s) {
    hipLaunchKernelGGL(loam_reduce_kernel, dim3(1), dim3(256), 0, s, a, k, d_out);
    return hipGetLastError();
}

}  // namespace pcr
