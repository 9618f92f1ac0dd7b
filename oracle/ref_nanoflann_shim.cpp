// oracle/ref_nanoflann_shim.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Thin C entry points around the REFERENCE's own vendored nanoflann.hpp,
// compiled from where it lies under /root/reference (never copied into this
// repo) into oracle/_ref/libref_nanoflann.so by oracle/Makefile.  It pins the
// k-NN stage of the oracle (and of the HIP path) to the reference's actual
// index: KDTreeSingleIndexAdaptor over float xyz with f64 distances, exactly
// as PointCloudKdtree<pt_t, double> instantiates it
// (third_parties/nanoflann/include/nanoflann/pcl_adaptor.hpp:33-34,47-58;
// the adaptor there needs PCL, so this file supplies a raw-array data source
// with the same three members).  Also used as the "reference" flavour of the
// CPU baseline's kd-tree (build per call + 5-NN), bench.py cpu_baseline.
#include <nanoflann.hpp>
#include <cstddef>
#include <cstdint>

namespace {
struct RawCloud {
    const float* p; size_t n; size_t stride;
    size_t kdtree_get_point_count() const { return n; }
    double kdtree_get_pt(size_t i, int d) const { return p[i * stride + d]; }
    template <class B> bool kdtree_get_bbox(B&) const { return false; }
};
using Metric = nanoflann::metric_L2_Simple::traits<double, RawCloud>::distance_t;
using Tree = nanoflann::KDTreeSingleIndexAdaptor<Metric, RawCloud, 3, size_t>;
struct Index { RawCloud cloud; Tree tree; Index(const float* p, size_t n, size_t s) : cloud{p, n, s}, tree(3, cloud) { tree.buildIndex(); } };
}

extern "C" {
void* ref_kd_build(const float* pts, size_t n, size_t stride_floats) { return new Index(pts, n, stride_floats); }
void ref_kd_free(void* h) { delete static_cast<Index*>(h); }
// k nearest of one query (f64 coordinates), ascending; returns count found
int ref_kd_knn(void* h, const double* q, int k, size_t* idx, double* d2)
{
    nanoflann::KNNResultSet<double> rs(k);
    rs.init(idx, d2);
    static_cast<Index*>(h)->tree.findNeighbors(rs, q);
    return (int)rs.size();
}
// batch: queries are float xyz (stride floats), converted to f64 like LoamRegister.cpp:54-55
void ref_kd_knn_batch(void* h, const float* q, size_t nq, size_t stride_floats, int k, int64_t* idx, double* d2)
{
    Index* ix = static_cast<Index*>(h);
    std::vector<size_t> ii(k);
    for (size_t i = 0; i < nq; ++i) {
        double qq[3] = {q[i * stride_floats], q[i * stride_floats + 1], q[i * stride_floats + 2]};
        nanoflann::KNNResultSet<double> rs(k);
        rs.init(ii.data(), d2 + i * k);
        ix->tree.findNeighbors(rs, qq);
        for (int j = 0; j < k; ++j) idx[i * k + j] = j < (int)rs.size() ? (int64_t)ii[j] : -1;
    }
}
}

// ---- the reference's two other nanoflann clients, compiled AS THEY LIE ------------------------------------------------
//   kfs_adaptor.hpp  KeyFramesKdtree: radius / nearest search over key-frame positions (frontend/src/MapManager.cpp:139-140,176-177)
//   vov_adaptor.h    VectorOfVectorsKdTree: k-NN over ScanContext ring keys, metric_L2 (backend/include/backend/ScanContext.hpp:39,
//                    backend/src/ScanContext.cpp:250)
// Both headers include only nanoflann.hpp; what they need from their containers is element access -- `c[i].pose.translation()(d)`
// and `c[i](d)` -- which the two element types below provide in place of the reference's Eigen-based KeyFrame and VectorXd.
#include <kfs_adaptor.hpp>
#include <vov_adaptor.h>
#include <vector>

namespace {
struct Coord3 { const double* p; double operator()(int d) const { return p[d]; } const double* data() const { return p; } };
struct PoseStandIn { double t[3]; Coord3 translation() const { return Coord3{t}; } };
struct KeyFrameStandIn { PoseStandIn pose; };
using Kfs = std::vector<KeyFrameStandIn>;
struct KeyVec { const double* p; double operator()(int d) const { return p[d]; } };
using Keys = std::vector<KeyVec>;
constexpr int kRingDim = 20;      // PC_NUM_RING (backend/include/backend/ScanContext.hpp)
}

extern "C" {
// pcl_adaptor.hpp:60-78 radiusSearch on the raw-array tree above (squared radius, strict '<', unsorted unless asked)
size_t ref_kd_radius(void* h, const double* q, double radius, int sorted, size_t* idx, double* d2, size_t cap)
{
    std::vector<nanoflann::ResultItem<size_t, double>> out;
    nanoflann::SearchParameters sp(0, sorted != 0);
    static_cast<Index*>(h)->tree.radiusSearch(q, radius * radius, out, sp);
    for (size_t i = 0; i < out.size() && i < cap; ++i) { idx[i] = out[i].first; d2[i] = out[i].second; }
    return out.size();
}
// MapManager::updateMap: KeyFramesKdtree<kfs_t, scalar_t, 3>(keyframes).radiusSearch(position, radius, ...), order as returned
size_t ref_kfs_radius(const double* pos, size_t n, const double* q, double radius, size_t* idx, double* d2, size_t cap)
{
    Kfs kfs(n);
    for (size_t i = 0; i < n; ++i) for (int d = 0; d < 3; ++d) kfs[i].pose.t[d] = pos[3 * i + d];
    nanoflann::KeyFramesKdtree<Kfs, double, 3> tree(kfs);
    std::vector<size_t> ki; std::vector<double> kd;
    const size_t m = tree.radiusSearch(Coord3{q}, radius, ki, kd);
    for (size_t i = 0; i < m && i < cap; ++i) { idx[i] = ki[i]; d2[i] = kd[i]; }
    return m;
}
size_t ref_kfs_nearest(const double* pos, size_t n, const double* q, size_t k, size_t* idx, double* d2)
{
    Kfs kfs(n);
    for (size_t i = 0; i < n; ++i) for (int d = 0; d < 3; ++d) kfs[i].pose.t[d] = pos[3 * i + d];
    nanoflann::KeyFramesKdtree<Kfs, double, 3> tree(kfs);
    std::vector<size_t> ki; std::vector<double> kd;
    const size_t m = tree.nearestKSearch(Coord3{q}, k, ki, kd);
    for (size_t i = 0; i < m; ++i) { idx[i] = ki[i]; d2[i] = kd[i]; }
    return m;
}
// ScanContext::query: ring_kdtree_.nearestKSearch(key.data(), NUM_CANDIDATES_FROM_TREE, ...) over the first n ring keys
size_t ref_vov_knn(const double* keys, size_t n, const double* q, int k, size_t* idx, double* d2)
{
    Keys v(n);
    for (size_t i = 0; i < n; ++i) v[i].p = keys + i * kRingDim;
    nanoflann::VectorOfVectorsKdTree<Keys, double, kRingDim> tree;
    tree.setInput(&v);
    std::vector<size_t> ki; std::vector<double> kd;
    const size_t m = tree.nearestKSearch(q, k, ki, kd);
    for (size_t i = 0; i < m; ++i) { idx[i] = ki[i]; d2[i] = kd[i]; }
    return m;
}
}
