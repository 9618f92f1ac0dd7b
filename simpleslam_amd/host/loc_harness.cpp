// loc_harness.cpp -- ROS-free counterpart of one pass of the reference's test/loc.cpp +
// LidarOdometry::generateOdom (frontend/src/LidarOdometry.cpp:170-184): load a map and a scan
// (raw float32 x y z intensity records), read the initial pose, call scan2Map through the C++ mirror
// of the plugin interface, print the refined pose.
//   loc_harness <method> <map.f32> <scan.f32> <init_pose.txt (4x4 row-major, test/align.cpp:85-93)> [downSampleVoxelGridSize]
// With the optional grid size the scan is voxel-filtered first, as LidarOdometry does (LidarOdometry.cpp:36,170-171).
//   loc_harness <method> submap:<keyframes.txt> <scan.f32> <init_pose.txt> <grid>
// builds the map like MapManager::updateMap instead: keyframes.txt holds one line per key frame, "<cloud.f32> r00 r01 .. r33"
// (4x4 row-major pose); the sub-map is assembled on the device around the initial position (8 m, grid) and never copied back.
#include <chrono>
#include <cstdio>
#include <fstream>
#include <string>

#include "PCR/HipRegister.hpp"

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

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s <loam|ndt|vgicp> <map.f32> <scan.f32> <init_pose.txt>\n", argv[0]); return 2; }
    try {
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
