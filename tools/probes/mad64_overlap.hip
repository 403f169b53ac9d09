// Hardware probe: does v_mad_u64_u32 / v_mad_i64_i32 on gfx950 give the right answer when its destination registers overlap its
// sources?  (Found while bisecting a wrong triangle frame: the compiler emitted  v_mad_u64_u32 v[6:7], s[6:7], v6, s6, 0 .)
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/mad64_overlap.hip -o build/mad64_overlap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CASE(NAME, SETUP, INSN, LO, HI)                                                                             \
    __global__ void NAME(const uint32_t *a, uint32_t b, uint32_t clo, uint32_t chi, uint64_t *out)                   \
    {                                                                                                                \
        uint32_t x = a[threadIdx.x], lo, hi;                                                                         \
        asm volatile("s_mov_b32 s6, %3\n s_mov_b32 s7, %3\n v_mov_b32 v10, %4\n v_mov_b32 v11, %5\n" SETUP "\n s_nop 7\n" INSN \
                     "\n s_nop 7\n v_mov_b32 %0, " LO "\n v_mov_b32 %1, " HI "\n"                                     \
                     : "=v"(lo), "=v"(hi)                                                                            \
                     : "v"(x), "s"(b), "v"(clo), "v"(chi)                                                            \
                     : "v6", "v7", "v8", "v9", "v10", "v11", "s6", "s7", "s8", "s9", "vcc");                         \
        out[threadIdx.x] = ((uint64_t)hi << 32) | lo;                                                                \
    }

CASE(k_ref,      "v_mov_b32 v6, %2", "v_mad_u64_u32 v[8:9], s[8:9], v6, s6, 0", "v8", "v9")
CASE(k_vlo_src0, "v_mov_b32 v6, %2", "v_mad_u64_u32 v[6:7], s[8:9], v6, s6, 0", "v6", "v7")
CASE(k_vhi_src0, "v_mov_b32 v7, %2", "v_mad_u64_u32 v[6:7], s[8:9], v7, s6, 0", "v6", "v7")
CASE(k_slo_src1, "v_mov_b32 v6, %2", "v_mad_u64_u32 v[8:9], s[6:7], v6, s6, 0", "v8", "v9")
CASE(k_shi_src1, "v_mov_b32 v6, %2", "v_mad_u64_u32 v[8:9], s[6:7], v6, s7, 0", "v8", "v9")
CASE(k_both,     "v_mov_b32 v6, %2", "v_mad_u64_u32 v[6:7], s[6:7], v6, s6, 0", "v6", "v7")
CASE(k_vlo_src1, "v_mov_b32 v8, %2\n v_mov_b32 v6, %3", "v_mad_u64_u32 v[6:7], s[8:9], v8, v6, 0", "v6", "v7")
CASE(k_vlo_add,  "v_mov_b32 v6, %2", "v_mad_u64_u32 v[6:7], s[8:9], v6, s6, v[10:11]", "v6", "v7")
CASE(k_inplace,  "v_mov_b32 v8, %2\n v_mov_b32 v6, %4\n v_mov_b32 v7, %5", "v_mad_u64_u32 v[6:7], s[8:9], v8, s6, v[6:7]", "v6", "v7")
CASE(k_i_ref,    "v_mov_b32 v6, %2", "v_mad_i64_i32 v[8:9], s[8:9], v6, s6, 0", "v8", "v9")
CASE(k_i_vlo,    "v_mov_b32 v6, %2", "v_mad_i64_i32 v[6:7], s[8:9], v6, s6, 0", "v6", "v7")
CASE(k_i_vhi,    "v_mov_b32 v7, %2", "v_mad_i64_i32 v[6:7], s[8:9], v7, s6, 0", "v6", "v7")

struct Case { const char *name; void (*fn)(const uint32_t *, uint32_t, uint32_t, uint32_t, uint64_t *); bool is_signed; bool adds; };

int main()
{
    const Case cases[] = {
        { "reference (no overlap)          v[8:9], s[8:9], v6, s6, 0", k_ref, false, false },
        { "vdst.lo == src0                 v[6:7], s[8:9], v6, s6, 0", k_vlo_src0, false, false },
        { "vdst.hi == src0                 v[6:7], s[8:9], v7, s6, 0", k_vhi_src0, false, false },
        { "sdst.lo == src1                 v[8:9], s[6:7], v6, s6, 0", k_slo_src1, false, false },
        { "sdst.hi == src1                 v[8:9], s[6:7], v6, s7, 0", k_shi_src1, false, false },
        { "vdst.lo == src0, sdst.lo==src1  v[6:7], s[6:7], v6, s6, 0", k_both, false, false },
        { "vdst.lo == src1 (vgpr)          v[6:7], s[8:9], v8, v6, 0", k_vlo_src1, false, false },
        { "vdst.lo == src0, + addend       v[6:7], s[8:9], v6, s6, v[10:11]", k_vlo_add, false, true },
        { "vdst == src2 (accumulate)       v[6:7], s[8:9], v8, s6, v[6:7]", k_inplace, false, true },
        { "i64 reference                   v[8:9], s[8:9], v6, s6, 0", k_i_ref, true, false },
        { "i64 vdst.lo == src0             v[6:7], s[8:9], v6, s6, 0", k_i_vlo, true, false },
        { "i64 vdst.hi == src0             v[6:7], s[8:9], v7, s6, 0", k_i_vhi, true, false },
    };
    std::vector<uint32_t> a(64);
    for (int i = 0; i < 64; ++i) a[i] = 0x9E3779B9u * (uint32_t)(i + 1) ^ (i << 27);
    a[0] = 0xFFFFFFFFu; a[1] = 1; a[2] = 0x80000000u; a[3] = 0x00012345u;
    uint32_t *d_a; uint64_t *d_o;
    hipMalloc(&d_a, 256); hipMalloc(&d_o, 512);
    hipMemcpy(d_a, a.data(), 256, hipMemcpyHostToDevice);
    const uint32_t bs[] = { 0xFFFFFFF1u, 0x0003A980u, 0x7FFFFFFFu, 0xFFFC5680u };
    const uint64_t addend = 0xFEDCBA9876543210ull;
    int bad_cases = 0;
    for (const Case &c : cases) {
        long bad = 0; int first_lane = -1; uint64_t got0 = 0, want0 = 0;
        for (uint32_t b : bs) {
            hipLaunchKernelGGL(c.fn, dim3(1), dim3(64), 0, 0, d_a, b, (uint32_t)addend, (uint32_t)(addend >> 32), d_o);
            uint64_t o[64];
            if (hipMemcpy(o, d_o, 512, hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 2; }
            for (int i = 0; i < 64; ++i) {
                uint64_t want = c.is_signed ? (uint64_t)((int64_t)(int32_t)a[i] * (int64_t)(int32_t)b) : (uint64_t)a[i] * b;
                if (c.adds) want += addend;
                if (o[i] != want) { if (!bad) { first_lane = i; got0 = o[i]; want0 = want; } ++bad; }
            }
        }
        printf("%-66s %s", c.name, bad ? "WRONG" : "ok");
        if (bad) printf("  (%ld of 256 lanes; first lane %d got %016llx want %016llx)", bad, first_lane, (unsigned long long)got0, (unsigned long long)want0);
        printf("\n");
        bad_cases += bad != 0;
    }
    printf("%d overlapping forms give wrong results\n", bad_cases);
    return 0;
}
