// PCR/HipRegister.hpp -- header-only C++ mirror of the reference's registration plugin over the
// C ABI of libpcr_hip.so (include/pcr_hip.h).  No PCL, no Eigen: the cloud and pose types below have
// the memory layout of the reference's (pcl::PointXYZI = 32 bytes, Eigen::Isometry3d = 16 doubles
// column-major), so an adapter for a tree that has PCL+Eigen only forwards pointers.
//
// Mirrors (reference): PCR/include/PCR/PointCloudRegister.hpp:12-38, LoamRegister.hpp, NdtRegister.hpp,
// VgicpRegister.hpp; the factory of frontend/src/LidarOdometry.cpp:44-54.
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include <cstdio>

#include "../../../include/pcr_hip.h"
#include "../config/params.hpp"

namespace PCR {

using scalar_t = double;                       // common/types/basic.hpp:16

struct alignas(16) PointXYZI {                 // pcl::PointXYZI: x y z 1 | intensity pad pad pad
    float x = 0, y = 0, z = 0, w = 1.f;
    float intensity = 0, pad[3] = {0, 0, 0};
};
static_assert(sizeof(PointXYZI) == 32, "must match pcl::PointXYZI");

struct PointCloud {
    std::vector<PointXYZI> points;
    size_t size() const { return points.size(); }
};
using PC_Ptr = std::shared_ptr<PointCloud>;
using PC_cPtr = std::shared_ptr<const PointCloud>;

struct pose_t {                                // Eigen::Isometry3d: 4x4 column-major
    double m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double* data() { return m; }
    const double* data() const { return m; }
    double& operator()(int r, int c) { return m[c * 4 + r]; }
    double operator()(int r, int c) const { return m[c * 4 + r]; }
};

class PointCloudRegister {
protected:
    bool isConverge = false;
    int cores = 1;                             // cfg["cores"]: OpenMP team of the CPU reference; the GPU path has no use for it
    std::string lastError_;
    // the reference logs through its spdlog singleton (`this->lg->error(...)`, LoamRegister.cpp:174); here: stderr
    void logError(const std::string& msg) { lastError_ = msg; std::fprintf(stderr, "[error] %s\n", msg.c_str()); }
public:
    using Ptr = std::shared_ptr<PointCloudRegister>;
    // PointCloudRegister.hpp:28-32: `cores = cfg["cores"].get<int>()` -- when a configuration has been loaded (config::Params::load)
    PointCloudRegister() { if (config::Params::loaded()) cores = config::Params::getInstance()["cores"].get<int>(); }
    int getCores() const { return cores; }
    const std::string& lastError() const { return lastError_; }
    virtual scalar_t getFitnessScore() { return 0; }
    virtual bool scan2Map(const PC_cPtr& src, const PC_cPtr& dst, pose_t& res) = 0;
    virtual ~PointCloudRegister() {}
};

// Common implementation: one pcr_handle per registrar (own HIP stream and buffers, so the odometry and the
// loop-closure registrars of the reference can run on their own threads concurrently).
class HipRegister : public PointCloudRegister {
protected:
    pcr_handle* h_ = nullptr;
    explicit HipRegister(const char* method, const pcr_params* p = nullptr) {
        h_ = pcr_create(method, p);
        if (!h_) throw std::runtime_error(pcr_last_error(nullptr));
    }
public:
    HipRegister(const HipRegister&) = delete;
    HipRegister& operator=(const HipRegister&) = delete;
    ~HipRegister() override { pcr_destroy(h_); }

    // Error behaviour of the reference: scan2Map never throws -- it logs and returns false (LoamRegister.cpp:173-176,222; `res` keeps
    // what the iterations had reached, here the caller's guess).  lastError() holds the library's message.
    bool scan2Map(const PC_cPtr& src, const PC_cPtr& dst, pose_t& res) override {
        int conv = 0;
        if (pcr_scan2map(h_, src->points.data(), src->size(), dst->points.data(), dst->size(), sizeof(PointXYZI), res.data(), &conv)) {
            logError(pcr_last_error(h_));
            return isConverge = false;
        }
        lastError_.clear();
        isConverge = conv != 0;
        return isConverge;
    }
    // static-map localisation (test/loc.cpp): index the map once.  (No counterpart in the reference's interface: failures throw.)
    void setTarget(const PC_cPtr& dst) {
        if (pcr_set_target(h_, dst->points.data(), dst->size(), sizeof(PointXYZI), 0)) throw std::runtime_error(pcr_last_error(h_));
    }
    bool align(const PC_cPtr& src, pose_t& res) {
        int conv = 0;
        if (pcr_align(h_, src->points.data(), src->size(), sizeof(PointXYZI), 0, res.data(), &conv)) {
            logError(pcr_last_error(h_));
            return isConverge = false;
        }
        lastError_.clear();
        isConverge = conv != 0;
        return isConverge;
    }
    pcr_handle* handle() { return h_; }
    // getFitnessScore of the reference's test/align.cpp:29-61: mean squared 1-NN distance (<= max_sq) of the source under `pose`
    // against the target of the last registration; -1 when no point is that close
    scalar_t gatedFitness(const PC_cPtr& src, const pose_t& pose, double max_sq = 1.0, int64_t* n_in = nullptr) {
        double score = -1.0;
        if (pcr_fitness_gated(h_, src->points.data(), src->size(), sizeof(PointXYZI), 0, pose.data(), max_sq, &score, n_in))
            throw std::runtime_error(pcr_last_error(h_));
        return score;
    }

    // scan2Map with `dst` = the sub-map a SubMap keeps in HBM (the scan is uploaded, the map never leaves the device)
    bool scan2Map(const PC_cPtr& src, const class SubMap& dst, pose_t& res);
};

// MapManager's key-frame store and sub-map (frontend/src/MapManager.cpp:151-201), kept in HBM
class SubMap {
    pcr_map* m_ = nullptr;
public:
    SubMap() : m_(pcr_map_create(-1)) { if (!m_) throw std::runtime_error(pcr_map_last_error(nullptr)); }
    SubMap(const SubMap&) = delete;
    SubMap& operator=(const SubMap&) = delete;
    ~SubMap() { pcr_map_destroy(m_); }
    void addKeyFrame(const PC_cPtr& pc, const pose_t& pose) {                 // KeyFrame{pc, pose}, common/types/basic.hpp:33-40
        if (pcr_map_add_keyframe(m_, pc->points.data(), pc->size(), sizeof(PointXYZI), 0, pose.data())) throw std::runtime_error(pcr_map_last_error(m_));
    }
    // MapManager::updateMap around `position`; returns the number of sub-map points
    size_t updateMap(const double position[3], double radius = 8.0 /* mSurroundingKeyframeSearchRadius */, double grid_size = 0.4) {
        size_t n = 0;
        if (pcr_map_update(m_, position, radius, grid_size, &n)) throw std::runtime_error(pcr_map_last_error(m_));
        return n;
    }
    // ... in two halves (MapManager's own thread as a stream, MapManager.cpp:109-119): queue the assembly; wait() -- or the next registration against this map -- collects it
    void updateMapBegin(const double position[3], double radius = 8.0, double grid_size = 0.4) {
        if (pcr_map_update_begin(m_, position, radius, grid_size)) throw std::runtime_error(pcr_map_last_error(m_));
    }
    size_t wait() {
        size_t n = 0;
        if (pcr_map_wait(m_, &n)) throw std::runtime_error(pcr_map_last_error(m_));
        return n;
    }
    std::vector<int64_t> submapIdx() const {                                    // mSubmapIdx
        size_t n = 0;
        pcr_map_submap_indices(m_, nullptr, 0, &n);
        std::vector<int64_t> idx(n);
        if (n) pcr_map_submap_indices(m_, idx.data(), n, &n);
        return idx;
    }
    const void* devicePointer(size_t* n, size_t* stride_bytes) const { return pcr_map_submap(m_, n, stride_bytes); }
    const pcr_map* handle() const { return m_; }
    uint64_t generation() const { uint64_t id = 0, g = 0; pcr_map_generation(m_, &id, &g); return g; }
};

inline bool HipRegister::scan2Map(const PC_cPtr& src, const SubMap& dst, pose_t& res) {
    // The handle keeps the target structures it builds from the sub-map for as long as the map stays at the same generation
    // (pcr_scan2map_submap): LidarOdometry registers several scans between two MapManager::updateMap calls.
    int conv = 0;
    if (pcr_scan2map_submap(h_, src->points.data(), src->size(), 0, dst.handle(), res.data(), &conv)) {
        logError(pcr_last_error(h_));
        return isConverge = false;
    }
    lastError_.clear();
    isConverge = conv != 0;
    return isConverge;
}

class LoamRegister : public HipRegister {
public:
    LoamRegister() : HipRegister("loam") {}
    explicit LoamRegister(const pcr_params& p) : HipRegister("loam", &p) {}
};

class NdtRegister : public HipRegister {
public:
    NdtRegister() : HipRegister("ndt") {}
    explicit NdtRegister(const pcr_params& p) : HipRegister("ndt", &p) {}
};

class VgicpRegister : public HipRegister {
    static pcr_params lc_params() {            // VgicpRegister::initForLC (VgicpRegister.cpp:21-28)
        pcr_params p;
        pcr_default_params(&p);
        p.vgicp_max_iters = 100;
        p.vgicp_trans_eps = 1e-6;
        return p;
    }
public:
    VgicpRegister() : HipRegister("vgicp") {}
    explicit VgicpRegister(const pcr_params& p) : HipRegister("vgicp", &p) {}
    static std::shared_ptr<VgicpRegister> makeForLC() { return std::make_shared<VgicpRegister>(lc_params()); }
    // VgicpRegister::initForLC (VgicpRegister.cpp:21-28) on the live object, as LoopClosureManager's constructor calls it
    // (backend/src/LoopClosureManager.cpp:21-22): 100 iterations, transformation epsilon 1e-6
    void initForLC() {
        pcr_params p;
        if (pcr_get_params(h_, &p)) throw std::runtime_error(pcr_last_error(h_));
        p.vgicp_max_iters = 100;
        p.vgicp_trans_eps = 1e-6;
        if (pcr_set_params(h_, &p)) throw std::runtime_error(pcr_last_error(h_));
    }
    scalar_t getFitnessScore() override { return pcr_fitness(h_); }
};

// The static-map adapter: test/loc.cpp loads the global map ONCE (MapManager::MapManager(pcd_file), frontend/src/MapManager.cpp:52-78) and
// registers every scan against it -- through the unchanged scan2Map(src, dst, res) interface.  Wrapped around any registrar above it
// indexes `dst` at its first call (pcr_set_target: the map crosses PCIe once) and aligns every later scan against what the device holds
// (pcr_align: only the scan is uploaded).  The CALLER says when the map's content has changed -- invalidate() -- the adapter never keys
// anything on the pointer it was handed (a cloud edited in place keeps its address).  Same poses as scan2Map, bit for bit.
class StaticMapRegister : public PointCloudRegister {
    std::shared_ptr<HipRegister> reg_;
    bool have_target_ = false;
public:
    explicit StaticMapRegister(std::shared_ptr<HipRegister> reg) : reg_(std::move(reg)) {}
    void invalidate() { have_target_ = false; }
    bool scan2Map(const PC_cPtr& src, const PC_cPtr& dst, pose_t& res) override {
        if (!have_target_) {
            try { reg_->setTarget(dst); } catch (const std::exception& e) { logError(e.what()); return isConverge = false; }
            have_target_ = true;
        }
        isConverge = reg_->align(src, res);
        lastError_ = reg_->lastError();
        return isConverge;
    }
    scalar_t getFitnessScore() override { return reg_->getFitnessScore(); }
    HipRegister& inner() { return *reg_; }
};

// frontend/src/LidarOdometry.cpp:44-54
inline PointCloudRegister::Ptr makeRegister(const std::string& pcr_type) {
    if (pcr_type == "loam") return std::make_shared<LoamRegister>();
    if (pcr_type == "ndt") return std::make_shared<NdtRegister>();
    if (pcr_type == "vgicp") return std::make_shared<VgicpRegister>();
    throw std::runtime_error("such pcr type(" + pcr_type + ") is not exist, please implemented your self!");
}
// ... the same registrar behind the static-map adapter (localisation against a map that is loaded once)
inline std::shared_ptr<StaticMapRegister> makeStaticMapRegister(const std::string& pcr_type) {
    return std::make_shared<StaticMapRegister>(std::dynamic_pointer_cast<HipRegister>(makeRegister(pcr_type)));
}

}  // namespace PCR

// backend/include/backend/ScanContext.hpp: the loop-closure descriptor database (addContext, query)
namespace context {
class ScanContext {
    pcr_sc* s_ = nullptr;
public:
    using QueryResult = std::pair<int, float>;                                  // {matched key frame or -1, yaw in rad}
    explicit ScanContext(const pcr_sc_params* p = nullptr) : s_(pcr_sc_create(-1, p)) { if (!s_) throw std::runtime_error(pcr_sc_last_error(nullptr)); }
    ScanContext(const ScanContext&) = delete;
    ScanContext& operator=(const ScanContext&) = delete;
    ~ScanContext() { pcr_sc_destroy(s_); }
    void addContext(const PCR::PointCloud& scan_down) {
        if (pcr_sc_add(s_, scan_down.points.data(), scan_down.size(), sizeof(PCR::PointXYZI), 0)) throw std::runtime_error(pcr_sc_last_error(s_));
    }
    QueryResult query(int id) {
        long long match = -1;
        float yaw = 0;
        if (pcr_sc_query(s_, id, &match, &yaw, nullptr)) throw std::out_of_range(pcr_sc_last_error(s_));     // ringcontexts_.at(id)
        return {(int)match, yaw};
    }
    size_t size() const { size_t n = 0; pcr_sc_size(s_, &n); return n; }
};
}  // namespace context

// common/pcp/pcp.hpp:14-28 (pcl::VoxelGrid with leaf = grid_size on all axes), on the device
namespace pcp {
inline void voxelDownSample(const PCR::PC_cPtr& cloudIn, PCR::PointCloud& cloudOut, float grid_size) {
    pcr_handle* h = pcr_create("loam", nullptr);       // any method: the filter only needs a device context
    if (!h) throw std::runtime_error(pcr_last_error(nullptr));
    const size_t n = cloudIn->size();
    std::vector<PCR::PointXYZI> tmp(n ? n : 1);
    size_t n_out = 0;
    const int rc = pcr_voxel_filter(h, cloudIn->points.data(), n, sizeof(PCR::PointXYZI), 0, grid_size, tmp.data(), n, 0, &n_out);
    const std::string msg = rc ? pcr_last_error(h) : "";
    pcr_destroy(h);
    if (rc) throw std::runtime_error(msg);
    tmp.resize(n_out);
    cloudOut.points.swap(tmp);
}
inline void voxelDownSample(PCR::PC_Ptr& cloud, float grid_size) {      // in place, like pcp.hpp:14-20
    PCR::PointCloud out;
    voxelDownSample(PCR::PC_cPtr(cloud), out, grid_size);
    cloud->points.swap(out.points);
}
}  // namespace pcp
