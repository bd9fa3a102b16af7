// spk_comm.cpp -- the two collectives KSPSolve needs across ranks, replacing
// what PETSc does with MPI inside the call at
// /root/reference/src/SaddlePointProblem.c:70:
//   * MPI_Allreduce of <= restart+2 doubles per Krylov reduction (VecMDot,
//     VecNorm)                       -> ncclAllReduce on the solver's stream
//   * VecScatter of the ghost entries MatMult_MPIAIJ needs
//                                    -> grouped ncclSend/ncclRecv with the two
//                                       slab neighbours over xGMI
// Backends:
//   RcclComm   one process per GPU (production; RCCL is dlopen'ed so that the
//              library loads, and the CPU tests run, on a box without it and so
//              that the process shares whichever librccl is already mapped)
//   LocalComm  several logical ranks inside one process on ONE device, each
//              driven by its own host thread -- lets a 1-GPU box run the
//              partitioned algorithm (parity tests); host barriers, slow.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>

#include "spk_internal.hpp"

// ---------------------------------------------------------------------------
// local group (in-process logical ranks)
// ---------------------------------------------------------------------------
struct spk_local_group {
    int nranks = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    // all-reduce slots: nranks x 256 doubles, device memory (allocated lazily by rank 0)
    double *slots = nullptr;
    // exchange bookkeeping published by every rank
    std::vector<const double *> sendbuf;
    std::vector<std::vector<int>> peers;
    std::vector<std::vector<int64_t>> send_off;
    // host staging for allgather
    std::vector<std::vector<char>> stage;

    // bounded: a rank that failed (exception) must not leave the others waiting forever
    void barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) spk::fail(SPK_ERR_COMM, "local group: a rank has failed");
        const long gen = generation;
        if (++arrived == nranks) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return generation != gen || broken; }) || broken) {
            broken = true;
            cv.notify_all();
            spk::fail(SPK_ERR_COMM, "local group: barrier timed out or a rank failed");
        }
    }
    bool broken = false;
};

namespace spk {
namespace {

// ---------------------------------------------------------------------------
// RCCL, resolved at run time
// ---------------------------------------------------------------------------
struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi &rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // share an already-mapped librccl (e.g. torch's) before opening ROCm's
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            api.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (api.h) break;
        }
        for (const char *n : names) {
            if (api.h) break;
            api.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!api.h) return;
        auto sym = [&](const char *s) { return dlsym(api.h, s); };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    });
    if (!api.h || !api.CommInitRank || !api.AllReduce || !api.Send || !api.Recv)
        fail(SPK_ERR_COMM, "RCCL (librccl.so.1) could not be loaded: %s", dlerror() ? dlerror() : "missing symbols");
    return api;
}

#define SPK_NCCL(call)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (call);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            fail(SPK_ERR_COMM, "%s failed: %s", #call,                                          \
                 rccl().GetErrorString ? rccl().GetErrorString(r_) : "rccl error");             \
    } while (0)

class SelfComm : public Comm {};

class RcclComm : public Comm {
public:
    RcclComm(int rank, int nranks, const void *id128) : rank_(rank), n_(nranks)
    {
        ncclUniqueId id;
        static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
        std::memcpy(&id, id128, sizeof id);
        SPK_NCCL(rccl().CommInitRank(&comm_, nranks, id, rank));
    }
    ~RcclComm() override
    {
        if (comm_) (void)rccl().CommDestroy(comm_);
    }
    int rank() const override { return rank_; }
    int size() const override { return n_; }
    void allreduce_sum(double *dev, int count, hipStream_t s) override
    {
        if (count <= 0) return;
        SPK_NCCL(rccl().AllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, comm_, s));
    }
    void exchange(const double *sendbuf, const std::vector<int> &peers, const std::vector<int64_t> &send_off,
                  double *recvbuf, const std::vector<int64_t> &recv_off, hipStream_t s) override
    {
        if (peers.empty()) return;
        SPK_NCCL(rccl().GroupStart());
        for (size_t i = 0; i < peers.size(); ++i) {
            const int64_t ns = send_off[i + 1] - send_off[i], nr = recv_off[i + 1] - recv_off[i];
            if (ns > 0) SPK_NCCL(rccl().Send(sendbuf + send_off[i], (size_t)ns, ncclDouble, peers[i], comm_, s));
            if (nr > 0) SPK_NCCL(rccl().Recv(recvbuf + recv_off[i], (size_t)nr, ncclDouble, peers[i], comm_, s));
        }
        SPK_NCCL(rccl().GroupEnd());
    }
    void host_allgather(const void *in, void *out, size_t bytes_each) override
    {
        DevBuf<char> din, dout;
        din.upload((const char *)in, bytes_each);
        dout.alloc(bytes_each * (size_t)n_);
        SPK_NCCL(rccl().AllGather(din.p, dout.p, bytes_each, ncclChar, comm_, nullptr));
        SPK_HIP(hipStreamSynchronize(nullptr));
        SPK_HIP(hipMemcpy(out, dout.p, bytes_each * (size_t)n_, hipMemcpyDeviceToHost));
    }
    void host_allgatherv(const void *in, size_t bytes_in, std::vector<std::vector<char>> &out) override
    {
        std::vector<int64_t> sizes((size_t)n_);
        const int64_t mine = (int64_t)bytes_in;
        host_allgather(&mine, sizes.data(), sizeof(int64_t));
        int64_t mx = 16;
        for (auto v : sizes) mx = std::max(mx, v);
        std::vector<char> padded((size_t)mx, 0), all((size_t)mx * (size_t)n_);
        if (bytes_in) std::memcpy(padded.data(), in, bytes_in);
        host_allgather(padded.data(), all.data(), (size_t)mx);
        out.resize((size_t)n_);
        for (int r = 0; r < n_; ++r)
            out[(size_t)r].assign(all.begin() + (size_t)mx * r, all.begin() + (size_t)mx * r + (size_t)sizes[(size_t)r]);
    }

private:
    int rank_, n_;
    ncclComm_t comm_ = nullptr;
};

class LocalComm : public Comm {
public:
    LocalComm(spk_local_group *g, int rank) : g_(g), rank_(rank)
    {
        if (rank == 0 && !g->slots) {
            SPK_HIP(hipMalloc((void **)&g->slots, sizeof(double) * 256 * (size_t)g->nranks));
            SPK_HIP(hipMemset(g->slots, 0, sizeof(double) * 256 * (size_t)g->nranks));
        }
        g->barrier();
    }
    int rank() const override { return rank_; }
    int size() const override { return g_->nranks; }
    void allreduce_sum(double *dev, int count, hipStream_t s) override
    {
        if (count <= 0) return;
        if (count > 256) fail(SPK_ERR_COMM, "local all-reduce limited to 256 values");
        SPK_HIP(hipMemcpyAsync(g_->slots + 256 * (size_t)rank_, dev, sizeof(double) * (size_t)count,
                               hipMemcpyDeviceToDevice, s));
        SPK_HIP(hipStreamSynchronize(s));
        g_->barrier();
        k::sum_slots(g_->slots, g_->nranks, 256, count, dev, s);
        SPK_HIP(hipStreamSynchronize(s));
        g_->barrier();
    }
    void exchange(const double *sendbuf, const std::vector<int> &peers, const std::vector<int64_t> &send_off,
                  double *recvbuf, const std::vector<int64_t> &recv_off, hipStream_t s) override
    {
        g_->sendbuf[(size_t)rank_] = sendbuf;
        g_->peers[(size_t)rank_] = peers;
        g_->send_off[(size_t)rank_] = send_off;
        SPK_HIP(hipStreamSynchronize(s));
        g_->barrier();
        for (size_t i = 0; i < peers.size(); ++i) {
            const int p = peers[i];
            const int64_t nr = recv_off[i + 1] - recv_off[i];
            if (nr <= 0) continue;
            const auto &pp = g_->peers[(size_t)p];
            size_t j = 0;
            while (j < pp.size() && pp[j] != rank_) ++j;
            if (j == pp.size()) fail(SPK_ERR_COMM, "local exchange: rank %d not a peer of %d", rank_, p);
            const auto &po = g_->send_off[(size_t)p];
            if (po[j + 1] - po[j] != nr) fail(SPK_ERR_COMM, "local exchange: size mismatch %d<->%d", rank_, p);
            SPK_HIP(hipMemcpyAsync(recvbuf + recv_off[i], g_->sendbuf[(size_t)p] + po[j], sizeof(double) * (size_t)nr,
                                   hipMemcpyDeviceToDevice, s));
        }
        SPK_HIP(hipStreamSynchronize(s));
        g_->barrier();
    }
    void host_allgather(const void *in, void *out, size_t bytes_each) override
    {
        g_->stage[(size_t)rank_].assign((const char *)in, (const char *)in + bytes_each);
        g_->barrier();
        for (int r = 0; r < g_->nranks; ++r) std::memcpy((char *)out + bytes_each * (size_t)r, g_->stage[(size_t)r].data(), bytes_each);
        g_->barrier();
    }
    void host_allgatherv(const void *in, size_t bytes_in, std::vector<std::vector<char>> &out) override
    {
        g_->stage[(size_t)rank_].assign((const char *)in, (const char *)in + bytes_in);
        g_->barrier();
        out = g_->stage;
        g_->barrier();
    }

private:
    spk_local_group *g_;
    int rank_;
};

// Host-callback transport: every operation goes device -> host -> callback -> device.
class HostCbComm : public Comm {
public:
    HostCbComm(int rank, int nranks, const spk_host_comm &cb) : rank_(rank), n_(nranks), cb_(cb) {}
    int rank() const override { return rank_; }
    int size() const override { return n_; }
    void allreduce_sum(double *dev, int count, hipStream_t s) override
    {
        if (count <= 0) return;
        std::vector<double> h((size_t)count);
        SPK_HIP(hipMemcpyAsync(h.data(), dev, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s));
        SPK_HIP(hipStreamSynchronize(s));
        if (cb_.allreduce(cb_.user, h.data(), count)) fail(SPK_ERR_COMM, "host all-reduce callback failed");
        SPK_HIP(hipMemcpyAsync(dev, h.data(), sizeof(double) * (size_t)count, hipMemcpyHostToDevice, s));
        SPK_HIP(hipStreamSynchronize(s));
    }
    void exchange(const double *sendbuf, const std::vector<int> &peers, const std::vector<int64_t> &send_off,
                  double *recvbuf, const std::vector<int64_t> &recv_off, hipStream_t s) override
    {
        if (peers.empty()) return;
        const int64_t ns = send_off.back(), nr = recv_off.back();
        std::vector<double> hs((size_t)std::max<int64_t>(ns, 1)), hr((size_t)std::max<int64_t>(nr, 1));
        if (ns) SPK_HIP(hipMemcpyAsync(hs.data(), sendbuf, sizeof(double) * (size_t)ns, hipMemcpyDeviceToHost, s));
        SPK_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i < peers.size(); ++i)
            if (cb_.exchange(cb_.user, peers[i], hs.data() + send_off[i], send_off[i + 1] - send_off[i],
                             hr.data() + recv_off[i], recv_off[i + 1] - recv_off[i]))
                fail(SPK_ERR_COMM, "host exchange callback failed");
        if (nr) SPK_HIP(hipMemcpyAsync(recvbuf, hr.data(), sizeof(double) * (size_t)nr, hipMemcpyHostToDevice, s));
        SPK_HIP(hipStreamSynchronize(s));
    }
    void host_allgather(const void *in, void *out, size_t bytes_each) override
    {
        if (cb_.allgather(cb_.user, in, out, (int64_t)bytes_each)) fail(SPK_ERR_COMM, "host all-gather callback failed");
    }
    void host_allgatherv(const void *in, size_t bytes_in, std::vector<std::vector<char>> &out) override
    {
        std::vector<int64_t> sizes((size_t)n_);
        const int64_t mine = (int64_t)bytes_in;
        host_allgather(&mine, sizes.data(), sizeof(int64_t));
        int64_t mx = 16;
        for (auto v : sizes) mx = std::max(mx, v);
        std::vector<char> padded((size_t)mx, 0), all((size_t)mx * (size_t)n_);
        if (bytes_in) std::memcpy(padded.data(), in, bytes_in);
        host_allgather(padded.data(), all.data(), (size_t)mx);
        out.resize((size_t)n_);
        for (int r = 0; r < n_; ++r)
            out[(size_t)r].assign(all.begin() + (size_t)mx * r, all.begin() + (size_t)mx * r + (size_t)sizes[(size_t)r]);
    }

private:
    int rank_, n_;
    spk_host_comm cb_;
};

}  // namespace

Comm *make_host_comm(int rank, int nranks, const spk_host_comm &cb) { return new HostCbComm(rank, nranks, cb); }
Comm *make_self_comm() { return new SelfComm(); }
Comm *make_rccl_comm(int rank, int nranks, const void *id128, int device)
{
    (void)device;
    return new RcclComm(rank, nranks, id128);
}
Comm *make_local_comm(spk_local_group *grp, int rank) { return new LocalComm(grp, rank); }

void rccl_unique_id(void *id128)
{
    ncclUniqueId id;
    SPK_NCCL(rccl().GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof id);
}

}  // namespace spk

extern "C" int spk_local_group_create(spk_local_group **grp, int nranks)
{
    if (!grp || nranks < 1 || nranks > 64) return SPK_ERR_ARG;
    auto *g = new spk_local_group();
    g->nranks = nranks;
    g->sendbuf.assign((size_t)nranks, nullptr);
    g->peers.resize((size_t)nranks);
    g->send_off.resize((size_t)nranks);
    g->stage.resize((size_t)nranks);
    *grp = g;
    return SPK_OK;
}

extern "C" int spk_local_group_destroy(spk_local_group *grp)
{
    if (!grp) return SPK_OK;
    if (grp->slots) (void)hipFree(grp->slots);
    delete grp;
    return SPK_OK;
}
