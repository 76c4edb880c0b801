// ubench_issue.hip -- lone-wave issue / latency calibration for the latency-bound tail kernels (rk_coop.h).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_issue.hip -o tools/ubench_issue ; run on the MI355X.
// Every kernel runs ITER x UNROLL copies of one instruction pattern in a single wave (grid 1) or one wave per SIMD
// (grid 1024) and reports ns per instruction (hipEvent time / instruction count) plus the s_memtime delta.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 2000;

#define REP8(X) X X X X X X X X
#define REP16(X) REP8(X) REP8(X)
#define REP32(X) REP16(X) REP16(X)

template <int K>
__global__ __launch_bounds__(64) void k_bench(double *out, unsigned long long *clk, double seed)
{
    double a = seed + threadIdx.x, b = 1.0000001, c = 0.5, d = a + 1, e = a + 2, f = a + 3, g = a + 4, h = a + 5, i2 = a + 6, j = a + 7;
    int u = threadIdx.x, v = u + 1;
    int addr = ((threadIdx.x & ~7) | 1) << 2;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
        if constexpr (K == 0) {   // dependent v_fma_f64 chain
            REP32(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
        } else if constexpr (K == 1) {   // 8 independent v_fma_f64 chains
            REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9"
                              : "+v"(a), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i2), "+v"(j) : "v"(b), "v"(c));)
        } else if constexpr (K == 2) {   // independent fma interleaved with s_mov (SALU slot cost)
            REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n s_mov_b32 s20, 0x3ff00000\n v_fma_f64 %1, %1, %2, %3\n s_mov_b32 s21, 0x40000000"
                               : "+v"(a), "+v"(d) : "v"(b), "v"(c) : "s20", "s21");)
        } else if constexpr (K == 3) {   // dependent v_mul_f64 chain
            REP32(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));)
        } else if constexpr (K == 4) {   // dependent v_add_f64 chain
            REP32(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c));)
        } else if constexpr (K == 5) {   // independent v_rcp_f64
            REP8(asm volatile("v_rcp_f64 %0, %4\n v_rcp_f64 %1, %4\n v_rcp_f64 %2, %4\n v_rcp_f64 %3, %4" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a));)
        } else if constexpr (K == 6) {   // dependent v_rcp_f64
            REP32(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));)
        } else if constexpr (K == 7) {   // dependent v_rsq_f64
            REP32(asm volatile("v_rsq_f64 %0, %0" : "+v"(a));)
        } else if constexpr (K == 8) {   // dependent ds_bpermute_b32 + wait
            REP32(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(u) : "v"(addr));)
        } else if constexpr (K == 9) {   // dependent v_mov_b32 dpp quad_perm
            REP32(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u));)
        } else if constexpr (K == 10) {  // dependent v_mov_b32 dpp row_shr:4
            REP32(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(u));)
        } else if constexpr (K == 11) {  // dependent v_mov_b64 dpp row_newbcast
            REP32(asm volatile("s_nop 1\n v_mov_b64_dpp %0, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(a));)
        } else if constexpr (K == 12) {  // fma feeding a dpp mov feeding an fma (hazard cost in a real chain)
            REP16(asm volatile("v_fma_f64 %0, %0, %1, %2\n s_nop 1\n v_mov_b64_dpp %0, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(c));)
        } else if constexpr (K == 13) {  // fma -> 2x ds_bpermute -> wait (what __shfl(double) does)
            REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); a = __shfl(a, (int)((threadIdx.x & ~7u) | 1u));)
        } else if constexpr (K == 14) {  // independent v_cndmask
            REP16(asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc" : "+v"(u), "+v"(v) : "v"(addr) : "vcc");)
        } else if constexpr (K == 15) {  // s_nop 0
            REP32(asm volatile("s_nop 0");)
        } else if constexpr (K == 16) {  // v_fmac_f64 with dpp row_newbcast operand (DPALU DPP)
            REP32(asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(c));)
        } else if constexpr (K == 17) {  // dependent v_div_fmas / div_fixup style: v_div_scale chain
            REP32(asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(a) : "v"(b) : "vcc");)
        } else if constexpr (K == 18) {  // dependent v_ldexp_f64
            REP32(asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a) : "v"(0));)
        } else if constexpr (K == 19) {  // two independent fma chains (ILP 2)
            REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));)
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a + d + e + f + g + h + i2 + j + (double)u + (double)v;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

struct Case { const char *name; int per_iter; void (*fn)(double *, unsigned long long *, double); };

int main(int argc, char **argv)
{
    double *out;
    unsigned long long *clk, hclk;
    CHECK(hipMalloc(&out, sizeof(double) * 64 * 4096));
    CHECK(hipMalloc(&clk, 8));
    Case cases[] = {
        {"dep v_fma_f64", 32, k_bench<0>}, {"indep x8 v_fma_f64", 32, k_bench<1>}, {"fma + s_mov interleaved (per instr)", 64, k_bench<2>},
        {"dep v_mul_f64", 32, k_bench<3>}, {"dep v_add_f64", 32, k_bench<4>}, {"indep v_rcp_f64", 32, k_bench<5>}, {"dep v_rcp_f64", 32, k_bench<6>},
        {"dep v_rsq_f64", 32, k_bench<7>}, {"dep ds_bpermute+wait", 32, k_bench<8>}, {"dep s_nop1+mov_dpp quad_perm (pair)", 32, k_bench<9>},
        {"dep s_nop1+mov_dpp row_shr4 (pair)", 32, k_bench<10>}, {"dep s_nop1+mov_b64_dpp newbcast (pair)", 32, k_bench<11>},
        {"fma->nop->b64 dpp (triple)", 16, k_bench<12>}, {"fma->2 bpermute->wait (quad)", 16, k_bench<13>}, {"indep v_cndmask", 32, k_bench<14>},
        {"s_nop 0", 32, k_bench<15>}, {"dep s_nop1+v_fmac_f64_dpp (pair)", 32, k_bench<16>}, {"dep v_div_scale_f64", 32, k_bench<17>},
        {"dep v_ldexp_f64", 32, k_bench<18>}, {"2 indep fma chains", 32, k_bench<19>},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int grid : {1, 1024, 2048}) {
        printf("---- grid %d (one 64-lane wave per block) ----\n", grid);
        for (const Case &c : cases) {
            hipLaunchKernelGGL(c.fn, dim3(grid), dim3(64), 0, 0, out, clk, 1.0);   // warm
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(c.fn, dim3(grid), dim3(64), 0, 0, out, clk, 1.0);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(&hclk, clk, 8, hipMemcpyDeviceToHost));
            const double n = (double)ITER * c.per_iter;
            printf("%-42s %8.3f ns/unit   %8.2f memtime-ticks/unit   (%.1f us total)\n", c.name, ms * 1e6 / n, (double)hclk / n, ms * 1e3);
        }
    }
    return 0;
}
