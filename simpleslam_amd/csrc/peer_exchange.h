// peer_exchange.h -- the ranks' sums without a collective library (pcr_comm_init_peer): one block's exchange, shared by the LOAM, NDT and VGICP loops.
#pragma once
#include "pcr_internal.h"

namespace pcr {

// ------------------------------------------------------------------------------
// Peer exchange (prototype, pcr_comm_init_peer): the ranks' sums without a collective library.  Every rank owns a RECEIVE buffer in its HBM
// (fine-grained, exported by hipIpcGetMemHandle and mapped by every peer): [2 parities][kMaxPeers writers][kPeerSlot doubles], word
// kPeerFlag of a slot = the sequence number of what the slot holds.  An exchange: write my values into MY slot of EVERY peer's buffer
// (stores that go out over xGMI), release to system scope, write the sequence number behind them; wait until every writer's slot of my OWN
// buffer carries this exchange's number (local polls), acquire, fold the slots IN RANK ORDER (so every rank gets the same bits).  Two parities:
// a rank can be at most one exchange ahead of the slowest (it needs that rank's contribution to go on), so the slot it overwrites has been read.
// The wait is bounded (kPeerTimeoutTicks of the 100 MHz clock): a peer that never arrives makes the exchange FAIL -- a status word in host-mapped memory
// (PeerComm::status), after which the host refuses every further exchange of the session: the ranks' sequence counters no longer agree -- it does not
// hang the device.
// ------------------------------------------------------------------------------
__device__ __forceinline__ bool peer_exchange_block(const PeerComm& pc, double seq, const double* vals /* LDS or regs by thread t < n */, double v_mine, int n, int op,
                                                    double* __restrict__ out) {
    const int t = threadIdx.x;
    const int par = (int)((unsigned long long)seq & 1ull);
    __shared__ int sh_ok;
    if (t == 0) sh_ok = 1;
    // my values into my slot of every peer's buffer
    if (t < n) {
        for (int p = 0; p < pc.nranks; ++p) {
            double* slot = pc.buf[p] + ((size_t)par * kMaxPeers + (size_t)pc.rank) * kPeerSlot;
            __hip_atomic_store(slot + t, v_mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // Every store above is a system-scope store (write-through past every cache) and so is every load below: once this wave's stores have
    // been acknowledged they are where the peers will read them, and the sequence word may follow.  (A release FENCE here would write back
    // the whole L2 -- the iterate kernel has just left 12 MB of cache entries in it: measured, the exchange then cost more than
    // reduce + ncclAllReduce.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t < pc.nranks) {
        double* slot = pc.buf[t] + ((size_t)par * kMaxPeers + (size_t)pc.rank) * kPeerSlot;
        __hip_atomic_store(slot + kPeerFlag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // every writer's slot of MY buffer
    if (t < pc.nranks) {
        const double* flag = pc.buf[pc.rank] + ((size_t)par * kMaxPeers + (size_t)t) * kPeerSlot + kPeerFlag;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (wall_clock64() - t0 > kPeerTimeoutTicks) { sh_ok = 0; break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool ok = sh_ok != 0;
    if (t < n) {
        double acc = op == 1 ? -1e308 : 0.0;
        for (int r = 0; r < pc.nranks; ++r) {
            const double* slot = pc.buf[pc.rank] + ((size_t)par * kMaxPeers + (size_t)r) * kPeerSlot;
            const double x = __hip_atomic_load(slot + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            acc = op == 1 ? fmax(acc, x) : acc + x;
        }
        out[t] = acc;      // (a peer never arrived: the sum is of whatever the slots held -- the STATUS word says so, not a value: a sum may be NaN in its own right)
    }
    if (!ok && t == 0 && pc.status) __hip_atomic_store(pc.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    (void)vals;
    return ok;
}

}  // namespace pcr
