"""The caller's workload: scan after scan along a trajectory, against a sub-map that grows by key frames.

One pass of the reference's front end in "lo" mode, reduced to what touches the registration path
(frontend/src/LidarOdometry.cpp:160-200, frontend/src/MapManager.cpp:109-201):

    scan -> voxel filter (downSampleVoxelGridSize)                         LidarOdometry.cpp:170-171
         -> scan2Map(filtered scan, sub-map, init = previous pose o commanded motion)   :179-181
         -> setCurPose: the sub-map is assembled again once the pose has moved minKFGap since the last assembly   MapManager.cpp:109-119
         -> putKeyFrame: the scan becomes a key frame when no key frame lies within minKFGap   :121-149
         -> updateMap: key frames within 8 m, transformed, concatenated, voxel-filtered        :151-201

`drive()` runs that loop over any front that offers the four steps -- `GpuFront` here (the C ABI: pcr_voxel_filter,
pcr_scan2map_submap, pcr_map_*), the CPU oracle's in oracle/ (tests and bench.py only).  The reference notifies a map thread and goes on;
here the assembly happens at the end of the step that asked for it, key frame of that step included: deterministic, and the same on
both fronts.  `GpuFront` queues the assembly (pcr_map_update_begin) and collects it when the sub-map is next needed -- after the next scan's voxel
filter, whose kernels the assembly's then run beside, as the reference's map thread works beside its front end; the registration that follows always
sees the finished sub-map, so the poses do not depend on it."""
import time

import numpy as np

from . import synth

MIN_KF_GAP = 1.0          # MapManager.hpp:67
SEARCH_RADIUS = 8.0       # MapManager.hpp:68


def make_drive(n_scans, seed, map_points=200_000, beams=64, azimuths=1024, step=0.5, odo_trans=0.02, odo_rot_deg=0.2):
    """-> (scans [n][N,4] float32 in the sensor frame, true poses, commanded motions).  commanded[k] = the motion from scan k-1 to scan k as an
    odometer reports it: the true relative motion o a small error (|t| <= odo_trans, |r| <= odo_rot_deg); commanded[0] = identity."""
    world, _ = synth.make_map(map_points, seed=seed)
    truth = [synth.trajectory_pose(world, k, step) for k in range(n_scans)]
    scans = [synth.make_scan(world, k, seed=seed, beams=beams, azimuths=azimuths, pose=truth[k])[0] for k in range(n_scans)]
    cmds = [np.eye(4)]
    for k in range(1, n_scans):
        rel = np.linalg.inv(truth[k - 1]) @ truth[k]
        cmds.append(synth.perturb(rel, seed * 1000 + k, trans=odo_trans, rot_deg=odo_rot_deg))
    return scans, truth, cmds


def drive(front, scans, cmds, start_pose, grid=0.5, radius=SEARCH_RADIUS, kf_gap=MIN_KF_GAP, prefetch=False):
    """Run the loop; -> dict(poses, converged, iterations, submap_points, keyframes, updates, seconds, scan2map_seconds, step_seconds).
    step_seconds: host time per step of the loop, summed over the drive (every step but a queued assembly returns when its result is there, so no
    synchronisation is added to measure them): voxel, wait (collecting a queued sub-map), scan2map, add_keyframe, update_map.
    prefetch: the NEXT scan's voxel filter is queued (front.prefetch) before this scan is registered -- a caller that has the next scan already (a
    recording replayed, a front end that lags its sensor); a front without prefetch() filters in place, so the results are the same either way."""
    pose = np.array(start_pose, float)
    kf_pos, n_kf, last_update = np.zeros((len(scans), 3)), 0, None      # (one array: the nearest key frame by one numpy expression, not a Python loop over them)
    poses, conv, iters, sub_n = [], [], [], []
    updates = 0
    t_s2m = 0.0
    ts = dict(voxel=0.0, wait=0.0, scan2map=0.0, add_keyframe=0.0, update_map=0.0)
    clock = time.perf_counter
    t0 = time.perf_counter()
    for k, scan in enumerate(scans):
        t1 = clock(); ds = front.voxel(scan, grid); ts["voxel"] += clock() - t1          # (collects the prefetched filter of this scan, if one was queued)
        if prefetch and k + 1 < len(scans) and hasattr(front, "prefetch"):
            t1 = clock(); front.prefetch(scans[k + 1], grid); ts["voxel"] += clock() - t1
        pose = pose @ cmds[k]
        t1 = clock(); n_sub = front.submap_points(); ts["wait"] += clock() - t1          # (collects an assembly the previous step queued)
        if k: sub_n.append(n_sub)              # sub-map after step k - 1 = the one scan k is registered against
        if n_sub > 0:
            t1 = time.perf_counter()
            c, it = front.scan2map(ds, pose)          # refines `pose` in place
            t_s2m += time.perf_counter() - t1
            conv.append(bool(c)); iters.append(int(it))
        else:
            conv.append(True); iters.append(0)
        t = pose[:3, 3]
        need_update = last_update is None or float(np.linalg.norm(last_update - t)) > kf_gap       # setCurPose
        # putKeyFrame: nearestKSearch's squared distance against minKFGap (MapManager.cpp:141-143)
        if n_kf == 0 or float(np.min(np.sum((kf_pos[:n_kf] - t) ** 2, axis=1))) > kf_gap:
            t1 = clock(); front.add_keyframe(scan, pose); ts["add_keyframe"] += clock() - t1
            kf_pos[n_kf] = t; n_kf += 1
        if need_update:
            t1 = clock(); front.update_map(t, radius, grid); ts["update_map"] += clock() - t1
            last_update = t.copy()
            updates += 1
        poses.append(pose.copy())
    t1 = clock(); sub_n.append(front.submap_points()); ts["wait"] += clock() - t1
    front.finish()
    ts["scan2map"] = t_s2m
    return dict(poses=poses, converged=conv, iterations=iters, submap_points=sub_n, keyframes=n_kf, updates=updates,
                seconds=time.perf_counter() - t0, scan2map_seconds=t_s2m, step_seconds=ts)


class GpuFront:
    """The four steps through the C ABI; scans are CUDA tensors (uploaded once, outside the timed loop), the sub-map never leaves HBM."""

    def __init__(self, register, submap=None, record=False):
        """record: keep, per scan2map call, what went in and what came out -- (filtered scan, sub-map, initial pose, result), all on the host -- so that a
        checker can repeat every call on exactly those inputs (tests, bench.py's parity figure; never in a timed pass)."""
        from .pcr import SubMap
        self.reg = register
        self.map = submap or SubMap()
        self._n = 0
        self.calls = [] if record else None
        self._sub_host = None
        self._filt, self._pre = None, None      # prefetch(): the filter's own handle, the filter in flight (scan, grid, token)

    def voxel(self, scan, grid):
        if self._pre is not None and self._pre[0] is scan and self._pre[1] == grid:      # queued by prefetch(): collect it
            tok, self._pre = self._pre[2], None
            return self._filt.voxelDownSampleEnd(tok)
        return self.reg.voxelDownSample(scan, grid)

    def prefetch(self, scan, grid):
        """Queue the voxel filter of a scan that is registered LATER, on a handle (a stream) of its own: pcr_voxel_filter_begin."""
        if self._filt is None:
            from .pcr import make_register
            self._filt = make_register("loam")      # (any method: the filter is the handle's, not the registrar's)
        if self._pre is not None:
            self._filt.voxelDownSampleEnd(self._pre[2])
        self._pre = (scan, grid, self._filt.voxelDownSampleBegin(scan, grid))

    def scan2map(self, ds, pose):
        init = pose.copy() if self.calls is not None else None
        c = self.reg.scan2MapSubmap(ds, self.map, pose)
        if self.calls is not None:
            self.calls.append((ds.cpu().numpy(), self._sub_host, init, pose.copy(), bool(c)))
        return c, self.reg.stats()["iterations"]

    def add_keyframe(self, scan, pose):
        self.map.addKeyFrame(scan, pose)

    def update_map(self, position, radius, grid):
        if self.calls is not None:
            self._n = self.map.updateMap(position, radius, grid)
            self._sub_host = self.map.download()
        else:
            self.map.updateMapBegin(position, radius, grid)      # queued: collected by submap_points() / scan2map of the next step
            self._n = None

    def submap_points(self):
        if self._n is None:
            self._n = self.map.wait()
        return self._n

    def reset(self):
        """A new session on the same handles: key frames and sub-map forgotten (pcr_map_clear), the device memory they grew to kept."""
        self.map.clear()
        self._n = 0
        if self._pre is not None:
            self._filt.voxelDownSampleEnd(self._pre[2]); self._pre = None

    def finish(self):
        import torch
        torch.cuda.synchronize()
