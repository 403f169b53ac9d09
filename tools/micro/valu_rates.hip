// Issue cost of the vector instructions the raster path is made of, with the tile kernel's occupancy (one 1024-thread workgroup per CU =
// 4 waves per SIMD): cycles per wave-instruction per SIMD = waves_per_simd * clock * time / instructions_per_wave.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o build/valu_rates && build/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 4096
#define UNROLL 8
template <int OP>
__global__ __launch_bounds__(1024) void k(float *out, float seed)
{
    uint32_t sc[UNROLL]; for (int u = 0; u < UNROLL; ++u) sc[u] = __builtin_amdgcn_readfirstlane((int)seed + u);
    float a[UNROLL]; double d[UNROLL]; int32_t q[UNROLL]; uint64_t m[UNROLL]; int64_t w[UNROLL];
    for (int u = 0; u < UNROLL; ++u) { a[u] = seed + u + threadIdx.x; d[u] = a[u]; q[u] = (int)a[u]; m[u] = (uint64_t)a[u] * 0x9E3779B97F4A7C15ull; w[u] = (int64_t)m[u]; }
    const float b = seed * 1.0001f, c = seed * 0.5f;
    const double bd = b, cd = c;
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (OP == 0) a[u] = __builtin_fmaf(a[u], b, c);
            if (OP == 1) d[u] = __builtin_fma(d[u], bd, cd);
            if (OP == 2) w[u] = (int64_t)q[u] * (int32_t)(w[u] >> 7) + w[u];      // v_mad_i64_i32
            if (OP == 3) m[u] = (m[u] << (q[u] & 31)) ^ (uint64_t)it;                // v_lshlrev_b64 (+ xor)
            if (OP == 4) { d[u] = (double)q[u] + d[u]; q[u] += it; }                  // v_cvt_f64_i32 + v_add_f64 (+ add)
            if (OP == 5) a[u] = __builtin_amdgcn_rcpf(a[u]) + c;                      // v_rcp_f32 + add
            if (OP == 6) a[u] = (float)d[u] * a[u] + 1.0f, d[u] += 1.0;               // v_cvt_f32_f64 + mul/add + add_f64
            if (OP == 7) q[u] = ((q[u] << 8) >> 8) * 3 + it;                               // small int mul + add
            if (OP == 8) q[u] = q[u] * (q[u] | 5) + it;                               // v_mul_lo_u32
            if (OP == 9) a[u] = __builtin_fminf(__builtin_fmaxf(a[u] * b, -4.0f), c);   // mul + med3
            if (OP == 10) m[u] = (m[u] & ~(uint64_t)q[u]) + 1;                        // 64-bit logic + add
            if (OP == 11) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc[u]) : : "scc");          // scalar ALU only
            if (OP == 12) { asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc[u]) : : "scc"); a[u] = __builtin_fmaf(a[u], b, c); }   // 1 SALU + 1 VALU
            if (OP == 13) { asm volatile("s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 5" : "+s"(sc[u]) : : "scc"); a[u] = __builtin_fmaf(a[u], b, c); }   // 2 SALU + 1 VALU
            if (OP == 14) { const unsigned long long bm = __ballot(a[u] > c); a[u] = __builtin_fmaf(a[u], b, (float)__popcll(bm)); }     // cmp + ballot popcount (SALU) + cvt + fma
            if (OP == 15) { a[u] = __builtin_amdgcn_readfirstlane(__float_as_int(a[u])) > 0 ? __builtin_fmaf(a[u], b, c) : a[u]; }   // readfirstlane + scalar cmp + branch/select
        }
    }
    float r = 0; for (int u = 0; u < UNROLL; ++u) r += (float)sc[u]; for (int u = 0; u < UNROLL; ++u) r += a[u] + (float)d[u] + (float)q[u] + (float)m[u] + (float)w[u];
    out[blockIdx.x * 1024 + threadIdx.x] = r;
}
template <int OP> void run(const char *name, float *d_out, int cus, double insts_per_iter)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP>), dim3(cus), dim3(1024), 0, 0, d_out, 1.5f);
    hipEventRecord(e0); hipLaunchKernelGGL((k<OP>), dim3(cus), dim3(1024), 0, 0, d_out, 1.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves, each N_ITER * UNROLL * insts wave-instructions
    const double per_simd = 4.0 * N_ITER * UNROLL * insts_per_iter;
    printf("%-44s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz, %.2f at 2.1)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, ms * 1e6 / per_simd * 2.1);
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    float *d; hipMalloc(&d, (size_t)p.multiProcessorCount * 1024 * 4);
    const int cus = p.multiProcessorCount;
    run<0>("v_fma_f32", d, cus, 1); run<1>("v_fma_f64", d, cus, 1); run<2>("v_mad_i64_i32 (+shift)", d, cus, 1); run<3>("v_lshlrev_b64 + 2 xor", d, cus, 1);
    run<4>("v_cvt_f64_i32 + v_add_f64 + v_add_u32", d, cus, 1); run<5>("v_rcp_f32 + v_add_f32", d, cus, 1); run<6>("v_cvt_f32_f64 + fma + add_f64", d, cus, 1);
    run<7>("mad_i24", d, cus, 1); run<8>("v_mul_lo_u32 + or + add", d, cus, 1); run<9>("v_mul_f32 + v_med3_f32", d, cus, 1); run<10>("64-bit andn + add", d, cus, 1);
    run<11>("s_add_u32", d, cus, 1); run<12>("s_add_u32 + v_fma_f32", d, cus, 1); run<13>("2 SALU + v_fma_f32", d, cus, 1); run<14>("v_cmp + s_bcnt1 + cvt + fma", d, cus, 1); run<15>("readfirstlane + s_cmp + cndmask/branch + fma", d, cus, 1);
    return 0;
}
