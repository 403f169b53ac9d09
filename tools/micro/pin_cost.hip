// What a page-locked frame-sized host buffer costs to make, two ways: hipHostMalloc against 2 MiB-aligned memory with MADV_HUGEPAGE,
// touched, then hipHostRegister -- and what a device-to-host copy of a C4 frame into each (and into pageable memory of both kinds) takes.
// hipcc --offload-arch=gfx950 -O2 tools/micro/pin_cost.hip -o build/pin_cost && ./build/pin_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double ms(std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); }
static void *huge_alloc(size_t n, bool touch)
{
    void *p = nullptr;
    if (posix_memalign(&p, (size_t)2 << 20, (n + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1)) != 0) return nullptr;
    madvise(p, n, MADV_HUGEPAGE);
    if (touch) for (size_t k = 0; k < n; k += 4096) ((volatile char *)p)[k] = 0;
    return p;
}
int main()
{
    const size_t n = (size_t)64 << 20;
    void *d = nullptr;
    CK(hipMalloc(&d, n));
    CK(hipMemset(d, 7, n));
    CK(hipDeviceSynchronize());
    hipStream_t s;
    CK(hipStreamCreate(&s));
    auto copy_ms = [&](void *h) { auto t0 = std::chrono::steady_clock::now(); (void)hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); return ms(t0); };
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        void *a = nullptr;
        CK(hipHostMalloc(&a, n, hipHostMallocDefault));
        const double t_a = ms(t0);
        const double c_a1 = copy_ms(a), c_a2 = copy_ms(a);
        t0 = std::chrono::steady_clock::now();
        void *b = huge_alloc(n, true);
        const double t_b0 = ms(t0);
        CK(hipHostRegister(b, n, hipHostRegisterDefault));
        const double t_b = ms(t0);
        const double c_b1 = copy_ms(b), c_b2 = copy_ms(b);
        t0 = std::chrono::steady_clock::now();
        void *c = huge_alloc(n, false);
        CK(hipHostRegister(c, n, hipHostRegisterDefault));
        const double t_c = ms(t0);
        const double c_c1 = copy_ms(c);
        // pageable destinations, fresh: ordinary and huge-page
        void *p1 = std::malloc(n);
        const double c_p1 = copy_ms(p1), c_p1b = copy_ms(p1);
        void *p2 = huge_alloc(n, false);
        const double c_p2 = copy_ms(p2), c_p2b = copy_ms(p2);
        std::printf("hipHostMalloc %.2f ms (copies %.2f, %.2f) | huge pages touched %.2f + register = %.2f ms (copies %.2f, %.2f) | huge pages untouched + register %.2f ms (copy %.2f) | "
                    "pageable: fresh malloc %.2f, again %.2f; fresh huge pages %.2f, again %.2f ms\n", t_a, c_a1, c_a2, t_b0, t_b, c_b1, c_b2, t_c, c_c1, c_p1, c_p1b, c_p2, c_p2b);
        t0 = std::chrono::steady_clock::now();
        CK(hipHostFree(a));
        const double f_a = ms(t0);
        t0 = std::chrono::steady_clock::now();
        CK(hipHostUnregister(b)); std::free(b);
        const double f_b = ms(t0);
        CK(hipHostUnregister(c)); std::free(c); std::free(p1); std::free(p2);
        std::printf("   free: hipHostFree %.2f ms, unregister + free %.2f ms\n", f_a, f_b);
    }
    return 0;
}
