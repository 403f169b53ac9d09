// What the packets between two dependent kernels cost on this runtime: event records, cross-stream waits (already satisfied), empty kernels.
// hipcc --offload-arch=gfx950 -O2 tools/micro/stream_packets.hip -o build/stream_packets && ./build/stream_packets
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) {} }
__global__ void nop() {}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    hipStream_t s, side;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    const int N = 200;
    std::vector<hipEvent_t> ev(4 * N), done(N);
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : done) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    auto run = [&](const char *name, int records, int waits, int nops) -> int {
        // side stream: events that are long complete when the main stream reaches its waits
        for (int k = 0; k < N; ++k) { hipLaunchKernelGGL(nop, 1, 64, 0, side); CK(hipEventRecord(done[k], side)); }
        CK(hipStreamSynchronize(side));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(t0, s));
            for (int k = 0; k < N; ++k) {
                hipLaunchKernelGGL(spin, 256, 64, 0, s, 2000);       // 20 us at 100 MHz
                for (int r = 0; r < records; ++r) CK(hipEventRecord(ev[4 * k + r], s));
                for (int w = 0; w < waits; ++w) CK(hipStreamWaitEvent(s, done[k], 0));
                for (int n = 0; n < nops; ++n) hipLaunchKernelGGL(nop, 64, 1024, 0, s);
            }
            CK(hipEventRecord(t1, s)); CK(hipEventSynchronize(t1));
            float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
            if (rep) printf("%-58s %7.2f us per iteration (20 us of it the kernel)\n", name, ms * 1e3 / N);
        }
        return 0;
    };
    if (run("kernel only", 0, 0, 0)) return 1;
    if (run("+ 1 event record", 1, 0, 0)) return 1;
    if (run("+ 3 event records", 3, 0, 0)) return 1;
    if (run("+ 2 waits on events of another stream (long complete)", 0, 2, 0)) return 1;
    if (run("+ 1 record + 2 waits", 1, 2, 0)) return 1;
    if (run("+ 1 empty 64 x 1024 kernel", 0, 0, 1)) return 1;
    if (run("+ 2 empty kernels", 0, 0, 2)) return 1;
    if (run("+ 1 record + 2 waits + 2 empty kernels (a frame's boundary)", 1, 2, 2)) return 1;
    return 0;
}
