// spk_k_comm.hip -- stand-alone launches of the peer-store collectives (granule all-reduce, halo exchange in granule
// and bulk form); the in-kernel forms live in spk_device.hpp.
#include "spk_device.hpp"

namespace spk {
namespace k {

// ---------------------------------------------------------------------------
// peer-store collectives: stand-alone launches (primitives: top of this file)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(2 * 64) void peer_allreduce_kernel(PeerAR a, double *buf, int count)
{
    peer_allreduce_block(a, buf, count, buf);
}
void peer_allreduce(const PeerAR &a, double *buf, int count, hipStream_t s)
{
    if (count < 1 || 2 * count > kArGranules) fail(SPK_ERR_COMM, "peer all-reduce: %d values (1..%d)", count, kArGranules / 2);
    hipLaunchKernelGGL(peer_allreduce_kernel, dim3(1), dim3(128), 0, s, a, buf, count);
}

// Loop-back form for tests: workgroup r plays rank r of a P-rank all-reduce through P windows that all live
// in this process (spk_debug_peer_allreduce_loopback): every lane of the window layout is exercised on one
// device, including lanes 4..7 that a box with at most six GPU processes cannot reach otherwise.
__global__ __launch_bounds__(2 * 64) void peer_allreduce_loopback_kernel(PeerAR a, double *buf, int count)
{
    a.me = (int)blockIdx.x;
    peer_allreduce_block(a, buf + (size_t)a.me * 64, count, buf + (size_t)a.me * 64);
}
void peer_allreduce_loopback(const PeerAR &a, double *buf, int count, hipStream_t s)
{
    if (count < 1 || 2 * count > kArGranules || a.P < 1 || a.P > kPeerMax) fail(SPK_ERR_ARG, "loop-back all-reduce: bad shape");
    hipLaunchKernelGGL(peer_allreduce_loopback_kernel, dim3(a.P), dim3(128), 0, s, a, buf, count);
}

// one thread per granule: send first, then wait for the granule with the same index of my own staging
__global__ __launch_bounds__(kThreads) void peer_exchange_kernel(PeerHalo h, const double *__restrict__ sendbuf,
                                                                 double *__restrict__ recvbuf)
{
    const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t ns2 = 2 * h.send_off[h.npeers], nr2 = 2 * h.recv_off[h.npeers];
    if (g < ns2) {
        const int64_t e = g >> 1;
        int p = 0;
        while (p + 1 < h.npeers && e >= h.send_off[p + 1]) ++p;
        const uint32_t half = reinterpret_cast<const uint32_t *>(sendbuf)[g];
        st_sys(h.remote[p] + (g - 2 * h.send_off[p]), ((unsigned long long)h.seq << 32) | half);
    }
    if (g < nr2) {
        uint32_t lo;
        const unsigned long long tw0 = (h.stats && threadIdx.x == 0) ? wall_clock64() : 0ull;
        const bool ok = granule_wait(h.mine + g, h.seq, h.timeout_ms, lo, h.err);
        if (h.stats && threadIdx.x == 0) {
            atomicAdd(h.stats + 2 * kStatHalo, wall_clock64() - tw0);
            atomicAdd(h.stats + 2 * kStatHalo + 1, 1ull);
        }
        const uint32_t other = __shfl_xor(lo, 1, kWave);
        if (!(g & 1)) recvbuf[g >> 1] = join_halves(lo, other);
        if (!ok) raise_comm_error(h.err, 19, h.seq);
    }
}
void peer_exchange(const PeerHalo &h, const double *sendbuf, double *recvbuf, hipStream_t s)
{
    const int64_t g = 2 * std::max(h.send_off[h.npeers], h.recv_off[h.npeers]);
    if (g == 0) return;
    hipLaunchKernelGGL(peer_exchange_kernel, dim3((unsigned)((g + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, h,
                       sendbuf, recvbuf);
}

// Bulk form (PeerBulk): one workgroup per chunk of kBulkChunk doubles.  Send workgroups come first in
// the grid: copy the chunk into the peer's staging, release at system scope, then ONE flag store.
// Receive workgroups wait for their chunk's flag, acquire, copy the chunk out.
__global__ __launch_bounds__(kThreads) void peer_exchange_bulk_kernel(PeerBulk h, const double *__restrict__ sendbuf,
                                                                      double *__restrict__ recvbuf)
{
    __shared__ int okf;
    const int nsend = h.send_chunk0[h.npeers];
    int b = blockIdx.x;
    if (b < nsend) {
        int i = 0;
        while (i + 1 < h.npeers && b >= h.send_chunk0[i + 1]) ++i;
        const int64_t e0 = (int64_t)(b - h.send_chunk0[i]) * kBulkChunk;
        const int64_t len = h.send_off[i + 1] - h.send_off[i];
        const int64_t n = len - e0 < kBulkChunk ? len - e0 : kBulkChunk;
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(sendbuf + h.send_off[i] + e0);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(h.rdata[i] + e0);
        for (int64_t j = threadIdx.x; j < n; j += kThreads) st_sys(dst + j, src[j]);
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) st_sys(h.rflag[i] + e0, (unsigned long long)h.seq);
        return;
    }
    b -= nsend;
    int i = 0;
    while (i + 1 < h.npeers && b >= h.recv_chunk0[i + 1]) ++i;
    const int64_t e0 = h.recv_off[i] + (int64_t)(b - h.recv_chunk0[i]) * kBulkChunk;
    const int64_t rem = h.recv_off[i + 1] - e0;
    const int64_t n = rem < kBulkChunk ? rem : kBulkChunk;
    if (threadIdx.x == 0) {
        bool ok = true;
        const unsigned long long tw0 = h.stats ? wall_clock64() : 0ull;
        if (ld_sys(h.mflag + e0) != (unsigned long long)h.seq) {
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                __builtin_amdgcn_s_sleep(4);
                if (ld_sys(h.mflag + e0) == (unsigned long long)h.seq) break;
                if (__hip_atomic_load(h.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                    wall_clock64() - t0 > (unsigned long long)h.timeout_ms * 100000ull) {
                    ok = false;
                    break;
                }
            }
        }
        if (h.stats) {
            atomicAdd(h.stats + 2 * kStatHalo, wall_clock64() - tw0);
            atomicAdd(h.stats + 2 * kStatHalo + 1, 1ull);
        }
        if (!ok) raise_comm_error(h.err, 20, h.seq);
        okf = ok;
    }
    __syncthreads();
    if (!okf) return;
    __threadfence_system();
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(h.mdata + e0);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(recvbuf + e0);
    for (int64_t j = threadIdx.x; j < n; j += kThreads) dst[j] = ld_sys(src + j);
}
void peer_exchange_bulk(const PeerBulk &h, const double *sendbuf, double *recvbuf, hipStream_t s)
{
    const int grid = h.send_chunk0[h.npeers] + h.recv_chunk0[h.npeers];
    if (grid == 0) return;
    hipLaunchKernelGGL(peer_exchange_bulk_kernel, dim3((unsigned)grid), dim3(kThreads), 0, s, h, sendbuf, recvbuf);
}


}  // namespace k
}  // namespace spk
