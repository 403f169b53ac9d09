// First touch of a frame-sized host buffer: ordinary malloc against a 2 MiB-aligned allocation with MADV_HUGEPAGE (round 5: a cold
// process's first read-back lands in fresh memory and is page-fault bound).  g++ -O2 thp_touch.cpp -o thp_touch && ./thp_touch
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
static double ms(std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); }
int main()
{
    const size_t n = (size_t)64 << 20;
    FILE *f = std::fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    char line[128] = "?";
    if (f) { if (!std::fgets(line, sizeof line, f)) line[0] = 0; std::fclose(f); }
    std::printf("transparent_hugepage/enabled: %s", line);
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        char *a = (char *)std::malloc(n);
        std::memset(a, 1, n);
        const double t_malloc = ms(t0);
        t0 = std::chrono::steady_clock::now();
        void *p = nullptr;
        if (posix_memalign(&p, (size_t)2 << 20, n) != 0) return 1;
        const int rc = madvise(p, n, MADV_HUGEPAGE);
        std::memset(p, 1, n);
        const double t_thp = ms(t0);
        std::printf("64 MiB first touch: malloc %.2f ms, 2 MiB-aligned + MADV_HUGEPAGE (rc %d) %.2f ms\n", t_malloc, rc, t_thp);
        std::free(a); std::free(p);
    }
    return 0;
}
