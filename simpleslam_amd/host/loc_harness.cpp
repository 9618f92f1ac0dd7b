// loc_harness.cpp -- ROS-free counterpart of one pass of the reference's test/loc.cpp +
// LidarOdometry::generateOdom (frontend/src/LidarOdometry.cpp:170-184), reading what test/loc.cpp reads:
//   loc_harness <params.json> <scan.pcd> <init_pose.txt> [--no-downsample] [--static <more_scans.txt>]
// params.json is the reference's configuration file (JSON with comments, config/params.hpp:30); the keys used are
//   "cores"                    -> PointCloudRegister's OpenMP team (PointCloudRegister.hpp:28-32; printed, the GPU has no use for it)
//   "downSampleVoxelGridSize"  -> leaf of the voxel filter of the map (MapManager.cpp:57,78) and of every scan (LidarOdometry.cpp:33,36,170-171)
//   "pcd_file"                 -> the global map, loaded with loadPCDFile (test/loc.cpp:34, MapManager.cpp:68)
//   "frontend"."pcr"           -> loam | ndt | vgicp, the factory of LidarOdometry.cpp:32,44-54 (unknown: throws, as there)
// The scan (what LidarDataProxy would deliver) is a PCD too; init_pose.txt is a 4x4 row-major pose (test/align.cpp:85-93).
// --no-downsample skips both voxel filters for fixtures that already have the sizes BASELINE config 1 fixes (65 536 x 100 k).
// --static <list>: the localisation loop of test/loc.cpp -- the map is loaded once, scan after scan is registered against it.  The list holds
// one line per FURTHER scan, "<scan.pcd> <init_pose.txt>"; every scan (the first included) goes through PCR::StaticMapRegister (the map is
// indexed at the first call and stays in HBM) AND through a fresh plain registrar's scan2Map: both poses are printed and must be equal.
// Prints the refined pose (17 significant digits), the converged flag and the elapsed time of scan2Map.
//
// Older forms, kept for the raw-float fixtures of the test suite (x y z intensity records):
//   loc_harness <method> <map.f32> <scan.f32> <init_pose.txt> [downSampleVoxelGridSize]
//   loc_harness <method> submap:<keyframes.txt> <scan.f32> <init_pose.txt> <grid>
// (the second builds the map like MapManager::updateMap: keyframes.txt holds one line per key frame, "<cloud.f32> r00 r01 .. r33";
// the sub-map is assembled on the device around the initial position (8 m, grid) and never copied back).
#include <chrono>
#include <cstdio>
#include <fstream>
#include <string>
#include <utility>
#include <vector>

#include "PCR/HipRegister.hpp"
#include "config/params.hpp"
#include "pcp/pcd_io.hpp"

static PCR::PC_Ptr load(const char* path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    auto pc = std::make_shared<PCR::PointCloud>();
    float r[4];
    while (f.read(reinterpret_cast<char*>(r), sizeof r)) {
        PCR::PointXYZI p;
        p.x = r[0]; p.y = r[1]; p.z = r[2]; p.intensity = r[3];
        pc->points.push_back(p);
    }
    return pc;
}

static PCR::pose_t read_pose(const char* path) {
    PCR::pose_t pose;
    std::ifstream pf(path);
    if (!pf) throw std::runtime_error(std::string("cannot open ") + path);
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) if (!(pf >> pose(r, c))) throw std::runtime_error("bad pose file");
    return pose;
}

static bool ends_with(const std::string& s, const char* suffix) {
    const size_t n = std::char_traits<char>::length(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

// test/loc.cpp as it is wired: configuration -> MapManager(pcd_file) -> LidarOdometry (factory on frontend.pcr) -> one scan
static int run_from_config(int argc, char** argv) {
    config::Params::load(argv[1]);
    auto cfg = config::Params::getInstance();
    const auto pcr_type = cfg["frontend"]["pcr"].get<std::string>();                 // LidarOdometry.cpp:32
    const auto grid_size = cfg["downSampleVoxelGridSize"].get<float>();              // LidarOdometry.cpp:33, MapManager.cpp:57
    const std::string pcd_file = cfg["pcd_file"];                                    // test/loc.cpp:34
    bool downsample = true;
    for (int i = 4; i < argc; ++i) if (std::string(argv[i]) == "--no-downsample") downsample = false;
    auto reg = PCR::makeRegister(pcr_type);                                          // throws on an unknown type, LidarOdometry.cpp:50-54
    // MapManager::MapManager(pcd_file), MapManager.cpp:66-78
    auto map = std::make_shared<PCR::PointCloud>();
    if (pcp::loadPCDFile(pcd_file, *map) == -1) throw std::runtime_error("can't load globalmap from: " + pcd_file);
    const size_t map_before = map->size();
    if (downsample) pcp::voxelDownSample(map, grid_size);
    auto scan = std::make_shared<PCR::PointCloud>();
    if (pcp::loadPCDFile(argv[2], *scan) == -1) throw std::runtime_error(std::string("can't load scan from: ") + argv[2]);
    const size_t scan_before = scan->size();
    if (downsample) pcp::voxelDownSample(scan, grid_size);                           // LidarOdometry.cpp:170-171
    PCR::pose_t pose = read_pose(argv[3]);
    for (int i = 4; i + 1 < argc; ++i) {
        if (std::string(argv[i]) != "--static") continue;
        // the static-map loop: scans[0] = the one of the command line, the others from the list
        std::vector<std::pair<PCR::PC_Ptr, PCR::pose_t>> work{{scan, pose}};
        std::ifstream lf(argv[i + 1]);
        if (!lf) throw std::runtime_error(std::string("cannot open ") + argv[i + 1]);
        std::string sf, pf;
        while (lf >> sf >> pf) {
            auto sc = std::make_shared<PCR::PointCloud>();
            if (pcp::loadPCDFile(sf, *sc) == -1) throw std::runtime_error("can't load scan from: " + sf);
            if (downsample) pcp::voxelDownSample(sc, grid_size);
            work.push_back({sc, read_pose(pf.c_str())});
        }
        auto loc = PCR::makeStaticMapRegister(pcr_type);
        int differing = 0;
        double sec_static = 0;
        std::printf("pcr %s  static map %zu -> %zu  scans %zu\n", pcr_type.c_str(), map_before, map->size(), work.size());
        for (auto& w : work) {
            PCR::pose_t a = w.second, b = w.second;
            const auto t0 = std::chrono::steady_clock::now();
            const bool ca = loc->scan2Map(w.first, map, a);
            sec_static += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const bool cb = PCR::makeRegister(pcr_type)->scan2Map(w.first, map, b);
            bool same = ca == cb;
            for (int k = 0; k < 16; ++k) same = same && a.m[k] == b.m[k];
            differing += same ? 0 : 1;
            std::printf("scan %zu converged %d %d  %s\n", w.first->size(), (int)ca, (int)cb, same ? "same pose" : "POSES DIFFER");
            for (int r = 0; r < 4; ++r) std::printf("%.17g %.17g %.17g %.17g\n", a(r, 0), a(r, 1), a(r, 2), a(r, 3));
        }
        std::printf("static-map registrations %.6f s in all, differing %d\n", sec_static, differing);
        return differing ? 4 : 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const bool conv = reg->scan2Map(scan, map, pose);                                // LidarOdometry.cpp:184
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("pcr %s  cores %d  grid %.9g  map %zu -> %zu  scan %zu -> %zu  converged %d  scan2map %.6f s\n", pcr_type.c_str(), reg->getCores(),
                (double)grid_size, map_before, map->size(), scan_before, scan->size(), (int)conv, sec);
    for (int r = 0; r < 4; ++r) std::printf("%.17g %.17g %.17g %.17g\n", pose(r, 0), pose(r, 1), pose(r, 2), pose(r, 3));
    if (!reg->lastError().empty()) return 3;      // scan2Map logged an error and returned false (LoamRegister.cpp:173-176)
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <params.json> <scan.pcd> <init_pose.txt> [--no-downsample]\n"
                             "       %s <loam|ndt|vgicp> <map.f32> <scan.f32> <init_pose.txt> [grid]\n", argv[0], argv[0]);
        return 2;
    }
    try {
        if (ends_with(argv[1], ".json")) return run_from_config(argc, argv);
        if (argc < 5) throw std::runtime_error("the raw-float form needs <method> <map.f32> <scan.f32> <init_pose.txt>");
        auto reg = PCR::makeRegister(argv[1]);
        const std::string map_arg = argv[2];
        if (map_arg.rfind("submap:", 0) == 0) {
            if (argc < 6) throw std::runtime_error("submap mode needs the grid size");
            const float grid = std::stof(argv[5]);
            PCR::SubMap sm;
            std::ifstream kf(map_arg.substr(7));
            std::string file;
            size_t n_kf = 0;
            while (kf >> file) {
                PCR::pose_t T;
                for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) if (!(kf >> T(r, c))) throw std::runtime_error("bad key-frame list");
                sm.addKeyFrame(load(file.c_str()), T);
                ++n_kf;
            }
            auto scan = load(argv[3]);
            PCR::pose_t pose;
            std::ifstream pf(argv[4]);
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) if (!(pf >> pose(r, c))) throw std::runtime_error("bad pose file");
            const double pos[3] = {pose(0, 3), pose(1, 3), pose(2, 3)};
            const size_t n_sub = sm.updateMap(pos, 8.0, grid);
            const size_t before = scan->size();
            pcp::voxelDownSample(scan, grid);
            auto* hr = dynamic_cast<PCR::HipRegister*>(reg.get());
            const bool conv = hr->scan2Map(PCR::PC_cPtr(scan), sm, pose);
            std::printf("method %s  key frames %zu (used %zu)  submap %zu  scan %zu -> %zu  converged %d\n", argv[1], n_kf, sm.submapIdx().size(), n_sub,
                        before, scan->size(), (int)conv);
            for (int r = 0; r < 4; ++r) std::printf("%.17g %.17g %.17g %.17g\n", pose(r, 0), pose(r, 1), pose(r, 2), pose(r, 3));
            return 0;
        }
        auto map = load(argv[2]);
        auto scan = load(argv[3]);
        PCR::pose_t pose;
        std::ifstream pf(argv[4]);
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) if (!(pf >> pose(r, c))) throw std::runtime_error("bad pose file");
        if (argc > 5) {
            const size_t before = scan->size();
            pcp::voxelDownSample(scan, std::stof(argv[5]));
            std::printf("voxel ds %zu -> %zu\n", before, scan->size());
        }
        const auto t0 = std::chrono::steady_clock::now();
        const bool conv = reg->scan2Map(scan, map, pose);
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("method %s  map %zu  scan %zu  converged %d  scan2map %.6f s  fitness %.6f\n", argv[1], map->size(), scan->size(), (int)conv, sec,
                    reg->getFitnessScore());
        for (int r = 0; r < 4; ++r) std::printf("%.17g %.17g %.17g %.17g\n", pose(r, 0), pose(r, 1), pose(r, 2), pose(r, 3));
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
