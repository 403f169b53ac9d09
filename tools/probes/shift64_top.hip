// Hardware probe: 64-bit VALU operations whose 32-bit operand sits in the LAST register of the wave's VGPR allocation.
// (k_triangle with .vgpr_count 16 computed  v_lshlrev_b64 v[2:3], v15, v[6:7]  wrongly in ~7 % of its waves; the same code
//  with a 24-register allocation was right, and a value parked in v15 survived -- so the register is intact and the READ is
//  what fails.)  Each kernel runs one instruction form in every lane of many waves and counts the lanes with a wrong result.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/shift64_top.hip -o build/shift64_top ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

// OP: 0 shl, 1 lshr, 2 ashr, 3 sub (32-bit control), 4 mad_u64_u32
#define PROBE(NAME, INSN, TOPREG, OP)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(uint32_t spin, unsigned long long *bad, unsigned long long *first)      \
    {                                                                                                                  \
        uint32_t t = blockIdx.x * 256u + threadIdx.x;                                                                  \
        uint32_t amt = (t * 7u + 3u) & 63u, xlo = t * 0x9E3779B9u + 12345u, xhi = (t ^ 0x5bd1e995u) * 0x85EBCA6Bu;     \
        uint32_t lo, hi;                                                                                               \
        asm volatile("s_mov_b32 s20, 0\n"                                                                              \
                     "L_%=:\n s_sleep 1\n s_add_u32 s20, s20, 1\n s_cmp_lt_u32 s20, %5\n s_cbranch_scc1 L_%=\n"        \
                     "v_mov_b32 v6, %3\n v_mov_b32 v7, %4\n v_mov_b32 v14, %2\n v_mov_b32 " TOPREG ", %2\n s_nop 4\n"  \
                     INSN "\n s_nop 4\n v_mov_b32 %0, v2\n v_mov_b32 %1, v3\n"                                         \
                     : "=v"(lo), "=v"(hi) : "v"(amt), "v"(xlo), "v"(xhi), "s"(spin & 15u)                              \
                     : "v2", "v3", "v6", "v7", "v8", "v14", TOPREG, "s8", "s9", "s20", "scc");                               \
        uint64_t x = ((uint64_t)xhi << 32) | xlo, want;                                                                \
        if (OP == 0) want = x << amt; else if (OP == 1) want = x >> amt; else if (OP == 2) want = (uint64_t)((int64_t)x >> amt); \
        else if (OP == 3) want = ((uint64_t)xhi << 32) | (uint32_t)(32u - amt); else if (OP == 4) want = (uint64_t)amt * xlo; \
        else if (OP == 5) want = (uint64_t)__double_as_longlong((double)amt);                                          \
        else if (OP == 6) want = (uint64_t)__double_as_longlong(ldexp(__longlong_as_double((long long)((x & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull)), (int)amt)); \
        else if (OP == 7) want = (x << (amt & 3u)) + x;                                                                \
        else want = x << amt;                                                                                          \
        if (OP == 3) hi = xhi;                                                                                         \
        uint64_t got = ((uint64_t)hi << 32) | lo;                                                                      \
        if (got != want && atomicAdd(bad, 1ull) == 0) { first[0] = got; first[1] = want; first[2] = amt; }             \
    }

PROBE(k_shl_v15_of16,  "v_lshlrev_b64 v[2:3], v15, v[6:7]", "v15", 0)
PROBE(k_shl_v14_of16,  "v_lshlrev_b64 v[2:3], v14, v[6:7]", "v15", 0)
PROBE(k_lshr_v15_of16, "v_lshrrev_b64 v[2:3], v15, v[6:7]", "v15", 1)
PROBE(k_ashr_v15_of16, "v_ashrrev_i64 v[2:3], v15, v[6:7]", "v15", 2)
PROBE(k_sub_v15_of16,  "v_sub_u32 v2, 32, v15",             "v15", 3)
PROBE(k_mad_v15_of16,  "v_mad_u64_u32 v[2:3], s[8:9], v15, v6, 0", "v15", 4)
PROBE(k_mad1_v15_of16, "v_mad_u64_u32 v[2:3], s[8:9], v6, v15, 0", "v15", 4)
PROBE(k_cvtu_v15_of16, "v_cvt_f64_u32 v[2:3], v15", "v15", 5)
PROBE(k_cvti_v15_of16, "v_cvt_f64_i32 v[2:3], v15", "v15", 5)
PROBE(k_ldexp_v15_of16, "v_and_b32 v7, 0xfffff, v7\n v_or_b32 v7, 0x3ff00000, v7\n s_nop 2\n v_ldexp_f64 v[2:3], v[6:7], v15", "v15", 6)
PROBE(k_lshladd_v15_of16, "v_and_b32 v15, 3, v15\n s_nop 2\n v_lshl_add_u64 v[2:3], v[6:7], v15, v[6:7]", "v15", 7)
PROBE(k_shl_src_top_of16, "v_mov_b32 v14, v6\n v_mov_b32 v15, v7\n v_mov_b32 v8, %2\n s_nop 2\n v_lshlrev_b64 v[2:3], v8, v[14:15]", "v15", 8)
PROBE(k_shl_dst_top_of16, "v_mov_b32 v8, %2\n s_nop 2\n v_lshlrev_b64 v[14:15], v8, v[6:7]\n s_nop 2\n v_mov_b32 v2, v14\n v_mov_b32 v3, v15", "v15", 8)
PROBE(k_shl_v23_of24,  "v_lshlrev_b64 v[2:3], v23, v[6:7]", "v23", 0)
PROBE(k_shl_v31_of32,  "v_lshlrev_b64 v[2:3], v31, v[6:7]", "v31", 0)
PROBE(k_shl_v30_of32,  "v_lshlrev_b64 v[2:3], v14, v[6:7]", "v31", 0)
PROBE(k_shl_v63_of64,  "v_lshlrev_b64 v[2:3], v63, v[6:7]", "v63", 0)
PROBE(k_shl_v127_of128, "v_lshlrev_b64 v[2:3], v127, v[6:7]", "v127", 0)

struct P { const char *name; void (*fn)(uint32_t, unsigned long long *, unsigned long long *); };
int main()
{
    const P probes[] = { { "v_lshlrev_b64 v[2:3], v15, v[6:7]   (16-register allocation, v15 is its last)", k_shl_v15_of16 },
                         { "v_lshlrev_b64 v[2:3], v14, v[6:7]   (16-register allocation)", k_shl_v14_of16 },
                         { "v_lshrrev_b64 v[2:3], v15, v[6:7]   (16-register allocation)", k_lshr_v15_of16 },
                         { "v_ashrrev_i64 v[2:3], v15, v[6:7]   (16-register allocation)", k_ashr_v15_of16 },
                         { "v_sub_u32     v2, 32, v15           (16-register allocation, 32-bit control)", k_sub_v15_of16 },
                         { "v_mad_u64_u32 v[2:3], s[8:9], v15, v6, 0 (16-register allocation)", k_mad_v15_of16 },
                         { "v_mad_u64_u32 v[2:3], s[8:9], v6, v15, 0 (16-register allocation)", k_mad1_v15_of16 },
                         { "v_cvt_f64_u32 v[2:3], v15           (16-register allocation)", k_cvtu_v15_of16 },
                         { "v_cvt_f64_i32 v[2:3], v15           (16-register allocation)", k_cvti_v15_of16 },
                         { "v_ldexp_f64   v[2:3], v[6:7], v15   (16-register allocation)", k_ldexp_v15_of16 },
                         { "v_lshl_add_u64 v[2:3], v[6:7], v15, v[6:7] (16-register allocation)", k_lshladd_v15_of16 },
                         { "v_lshlrev_b64 v[2:3], v8, v[14:15]  (16-register allocation, 64-bit SOURCE in the last pair)", k_shl_src_top_of16 },
                         { "v_lshlrev_b64 v[14:15], v8, v[6:7]  (16-register allocation, RESULT in the last pair)", k_shl_dst_top_of16 },
                         { "v_lshlrev_b64 v[2:3], v23, v[6:7]   (24-register allocation, v23 is its last)", k_shl_v23_of24 },
                         { "v_lshlrev_b64 v[2:3], v31, v[6:7]   (32-register allocation, v31 is its last)", k_shl_v31_of32 },
                         { "v_lshlrev_b64 v[2:3], v14, v[6:7]   (32-register allocation)", k_shl_v30_of32 },
                         { "v_lshlrev_b64 v[2:3], v63, v[6:7]   (64-register allocation, v63 is its last)", k_shl_v63_of64 },
                         { "v_lshlrev_b64 v[2:3], v127, v[6:7]  (128-register allocation, v127 is its last)", k_shl_v127_of128 } };
    unsigned long long *d;
    if (hipMalloc(&d, 64) != hipSuccess) return 2;
    for (const P &p : probes) {
        (void)hipMemset(d, 0, 64);
        const unsigned blocks = 32768;
        hipLaunchKernelGGL(p.fn, dim3(blocks), dim3(256), 0, 0, 7u, d, d + 1);
        unsigned long long h[4] = {};
        if (hipMemcpy(h, d, 32, hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 2; }
        printf("%-86s %llu of %u lanes wrong", p.name, h[0], blocks * 256);
        if (h[0]) printf("  (first: shift %llu got %016llx want %016llx)", h[3], h[1], h[2]);
        printf("\n");
    }
    return 0;
}
