// Development probe: latency of the granule all-reduce / halo kernels between P "ranks" that are P
// non-blocking streams of ONE process on one GPU (run with GPU_MAX_HW_QUEUES=8 so that every
// stream owns a hardware queue).  What it measures is launch + store + poll on one chip; xGMI adds
// its flight time on top.  Build: hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/peer_latency.cpp -o tools/peer_latency
// -Lsaddle_point_petsc_amd -lspk -Wl,-rpath,'$ORIGIN/../saddle_point_petsc_amd'
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../saddle_point_petsc_amd/csrc/spk_internal.hpp"

using namespace spk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv)
{
    const int P = argc > 1 ? atoi(argv[1]) : 2, reps = argc > 2 ? atoi(argv[2]) : 2000, count = argc > 3 ? atoi(argv[3]) : 32;
    std::vector<hipStream_t> st(P);
    std::vector<unsigned long long *> win(P);
    std::vector<double *> buf(P);
    std::vector<int32_t *> err(P);
    const size_t bytes = sizeof(unsigned long long) * k::kArSlots * P * k::kArGranules;
    for (int r = 0; r < P; ++r) {
        CK(hipStreamCreateWithFlags(&st[r], hipStreamNonBlocking));
        CK(hipExtMallocWithFlags((void **)&win[r], bytes, hipDeviceMallocUncached));
        CK(hipMemset(win[r], 0, bytes));
        CK(hipMalloc((void **)&buf[r], 64 * 8));
        CK(hipMemset(buf[r], 0, 64 * 8));
        CK(hipMalloc((void **)&err[r], 4));
        CK(hipMemset(err[r], 0, 4));
    }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    uint32_t seq = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const int n = pass == 0 ? 50 : reps;
        CK(hipEventRecord(e0, st[0]));
        for (int i = 0; i < n; ++i) {
            ++seq;
            for (int r = 0; r < P; ++r) {
                k::PeerAR a{};
                a.P = P; a.me = r; a.seq = seq; a.timeout_ms = 2000; a.err = err[r];
                for (int p = 0; p < P; ++p) a.win[p] = win[p];
                k::peer_allreduce(a, buf[r], count, st[r]);
            }
        }
        CK(hipEventRecord(e1, st[0]));
        for (int r = 0; r < P; ++r) CK(hipStreamSynchronize(st[r]));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (pass) printf("P=%d count=%d: %.2f us per all-reduce (stream 0, %d back-to-back launches)\n", P, count, ms * 1e3 / n, n);
    }
    int32_t e = 0;
    CK(hipMemcpy(&e, err[0], 4, hipMemcpyDeviceToHost));
    printf("err word: %d\n", e);
    return 0;
}
