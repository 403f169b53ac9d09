// Hardware probe: is the LAST register of a wave's VGPR allocation safe while other waves start and end on the same SIMD?
// (Found while bisecting a wrong triangle frame: k_triangle with .vgpr_count 16 lost the value of v15 in ~6 % of its waves;
//  the same instructions with .vgpr_count 17 -- a 24-register allocation -- were right.)
// Each kernel parks a pattern in one named register, idles for a while, reads it back and counts the lanes that changed.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/vgpr_top.hip -o build/vgpr_top ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define PROBE(NAME, REG, TOP)                                                                                        \
    __global__ __launch_bounds__(256) void NAME(uint32_t iters, uint32_t early, unsigned long long *bad)             \
    {                                                                                                                \
        uint32_t pattern = 0xA5000000u ^ (blockIdx.x * 256u + threadIdx.x), got;                                   \
        if (early && (blockIdx.x & 3) == 3 && threadIdx.x >= 64) return; /* some waves end at once, like k_triangle */ \
        asm volatile("v_mov_b32 " REG ", %1\n"                                                                       \
                     "s_mov_b32 s20, 0\n"                                                                            \
                     "L_%=:\n s_sleep 1\n s_add_u32 s20, s20, 1\n s_cmp_lt_u32 s20, %2\n s_cbranch_scc1 L_%=\n"      \
                     "v_mov_b32 %0, " REG "\n"                                                                       \
                     : "=v"(got) : "v"(pattern), "s"(iters) : REG, TOP, "s20", "scc");                               \
        if (got != pattern) atomicAdd(bad, 1ull);                                                                    \
    }

PROBE(k_v7_of_8,    "v7",  "v7")
PROBE(k_v6_of_8,    "v6",  "v7")
PROBE(k_v15_of_16,  "v15", "v15")
PROBE(k_v14_of_16,  "v14", "v15")
PROBE(k_v8_of_16,   "v8",  "v15")
PROBE(k_v15_of_24,  "v15", "v16")
PROBE(k_v23_of_24,  "v23", "v23")
PROBE(k_v31_of_32,  "v31", "v31")
PROBE(k_v63_of_64,  "v63", "v63")
PROBE(k_v95_of_96,  "v95", "v95")
PROBE(k_v127_of_128, "v127", "v127")

struct P { const char *name; void (*fn)(uint32_t, uint32_t, unsigned long long *); };

int main()
{
    const P probes[] = { { "v7  in an  8-register allocation", k_v7_of_8 },   { "v6  in an  8-register allocation", k_v6_of_8 },
                         { "v15 in a  16-register allocation", k_v15_of_16 }, { "v14 in a  16-register allocation", k_v14_of_16 },
                         { "v8  in a  16-register allocation", k_v8_of_16 },  { "v15 in a  24-register allocation", k_v15_of_24 },
                         { "v23 in a  24-register allocation", k_v23_of_24 }, { "v31 in a  32-register allocation", k_v31_of_32 },
                         { "v63 in a  64-register allocation", k_v63_of_64 }, { "v95 in a  96-register allocation", k_v95_of_96 },
                         { "v127 in a 128-register allocation", k_v127_of_128 } };
    unsigned long long *d_bad;
    if (hipMalloc(&d_bad, 8) != hipSuccess) return 2;
    for (uint32_t early = 0; early < 2; ++early)
        for (const P &p : probes) {
            (void)hipMemset(d_bad, 0, 8);
            const unsigned blocks = 32768;               // 128 blocks per CU: waves start and end all the time
            hipLaunchKernelGGL(p.fn, dim3(blocks), dim3(256), 0, 0, 40u, early, d_bad);
            unsigned long long bad = 0;
            if (hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 2; }
            printf("%-36s %s: %llu of %u lanes changed\n", p.name, early ? "(a quarter of the blocks end 3 waves early)" : "(all waves idle equally)       ", bad,
                   blocks * 256);
        }
    return 0;
}
