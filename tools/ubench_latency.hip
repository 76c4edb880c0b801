// ubench_latency.hip -- what does a DEPENDENT instruction cost a lone wave on gfx950?
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_latency.hip -o tools/ubench_latency ; run on the MI355X.
// The cooperative tail of a solve runs at most one wave per SIMD, so its time is (instructions per attempt) x (issue
// interval of a lone wave) + dependency stalls.  tools/ubench_select.hip measured the issue interval of independent
// instructions (about 2.6 ns whatever the instruction); this file measures chains where every instruction needs the
// previous result, and chains of 2 / 3 / 4 interleaved independent streams, to see how much of the tail's per-attempt
// time is dependency stall and how much is issue.
// Reported: ns per instruction of ONE wave, for a lone wave (grid 1) and one wave per SIMD (grid 1024).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 2000;
#define REP4(X) X X X X
#define REP8(X) REP4(X) REP4(X)
#define REP16(X) REP8(X) REP8(X)
#define REP32(X) REP16(X) REP16(X)

__device__ __forceinline__ double dpp_add(double a)
{
    int lo = __double2loint(a);
    int hi = __double2hiint(a);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, false);   // row_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, false);
    a = a + __hiloint2double(hi, lo);
    asm volatile("" : "+v"(a));
    return a;
}

template <int K>
__global__ __launch_bounds__(64) void k_lat(double *out, double seed)
{
    double a = seed + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-9, d = a + 1, e = a + 2, f = a + 3, t = 0;
    int u = threadIdx.x, w = u + 2;
    for (int it = 0; it < ITER; ++it) {
        if constexpr (K == 0) {          // v_fma_f64, one dependent chain
            REP32(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
        } else if constexpr (K == 1) {   // v_add_f64 dependent
            REP32(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c));)
        } else if constexpr (K == 2) {   // v_mul_f64 dependent
            REP32(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));)
        } else if constexpr (K == 3) {   // v_rcp_f64 dependent
            REP32(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));)
        } else if constexpr (K == 4) {   // v_rsq_f64 dependent
            REP32(asm volatile("v_rsq_f64 %0, %0" : "+v"(a));)
        } else if constexpr (K == 5) {   // v_fma_f64, two interleaved chains
            REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));)
        } else if constexpr (K == 6) {   // three interleaved chains (30 + 2)
            REP8(asm volatile("v_fma_f64 %0, %0, %3, %4\n v_fma_f64 %1, %1, %3, %4\n v_fma_f64 %2, %2, %3, %4\n v_fma_f64 %0, %0, %3, %4" : "+v"(a), "+v"(d), "+v"(e) : "v"(b), "v"(c));)
        } else if constexpr (K == 7) {   // four interleaved chains
            REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));)
        } else if constexpr (K == 8) {   // the cooperative kernels' cross-lane add: 2 x v_mov_b32_dpp + v_add_f64, dependent (3 instr, s_nop not counted)
            REP16(a = dpp_add(a);)
        } else if constexpr (K == 9) {   // dependent v_fma_f64 with an independent 32-bit VALU instruction between (does it fill the stall?)
            REP32(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_and_b32 %1, %1, %4" : "+v"(a), "+v"(u) : "v"(b), "v"(c), "v"(w));)
        } else if constexpr (K == 10) {  // v_cndmask_b32 (VOP3) dependent
            REP32(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(u) : "v"(w) : "s20", "s21");)
        } else if constexpr (K == 11) {  // v_and_b32 dependent
            REP32(asm volatile("v_and_b32 %0, %0, %1" : "+v"(u) : "v"(w));)
        } else if constexpr (K == 12) {  // dependent v_fma_f64 with a SALU instruction between
            REP32(asm volatile("v_fma_f64 %0, %0, %1, %2\n s_and_b64 s[20:21], s[20:21], exec" : "+v"(a) : "v"(b), "v"(c) : "s20", "s21", "scc");)
        } else if constexpr (K == 13) {  // dependent v_fma_f64 with s_nop 0 between
            REP32(asm volatile("v_fma_f64 %0, %0, %1, %2\n s_nop 0" : "+v"(a) : "v"(b), "v"(c));)
        } else if constexpr (K == 14) {  // v_div_fmas_f64 / v_div_fixup_f64 / v_ldexp_f64 / v_div_scale dependent mix
            REP8(asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0\n v_div_fmas_f64 %0, %0, %1, %2\n v_div_fixup_f64 %0, %0, %1, %2\n v_ldexp_f64 %0, %0, 0" : "+v"(a) : "v"(b), "v"(c) : "vcc");)
        } else if constexpr (K == 15) {  // full IEEE division chain a = a / b  (compiler expansion, counted as 1 op of ~ 11 instr)
            REP32(a = a / b; asm volatile("" : "+v"(a));)
            asm volatile("s_nop 0");
        } else if constexpr (K == 16) {  // full IEEE sqrt chain
            REP32(a = __builtin_sqrt(a + 2.0); asm volatile("" : "+v"(a));)
        }
    }
    if (a + d + e + f + u + t == 12345.678) out[threadIdx.x] = a + d + u;
}

struct Case { const char *name; void (*fn)(double *, double); int per_iter; };

int main()
{
    double *out;
    CHECK(hipMalloc(&out, 64 * sizeof(double)));
    const Case cases[] = {
        {"v_fma_f64 dependent", k_lat<0>, 32}, {"v_add_f64 dependent", k_lat<1>, 32}, {"v_mul_f64 dependent", k_lat<2>, 32},
        {"v_rcp_f64 dependent", k_lat<3>, 32}, {"v_rsq_f64 dependent", k_lat<4>, 32},
        {"v_fma_f64 2 chains", k_lat<5>, 32}, {"v_fma_f64 3 chains", k_lat<6>, 32}, {"v_fma_f64 4 chains", k_lat<7>, 32},
        {"dpp cross-lane add (per ADD, 6 instr)", k_lat<8>, 16}, {"dep fma + indep v_and (per 2)", k_lat<9>, 64},
        {"v_cndmask_b32_e64 dependent", k_lat<10>, 32}, {"v_and_b32 dependent", k_lat<11>, 32},
        {"dep fma + s_and_b64 (per 2)", k_lat<12>, 64}, {"dep fma + s_nop 0 (per 2)", k_lat<13>, 64},
        {"div_scale/fmas/fixup/ldexp dep", k_lat<14>, 32},
        {"IEEE a/b dependent (per DIVISION)", k_lat<15>, 32}, {"IEEE sqrt dependent (per SQRT)", k_lat<16>, 32},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grids[] = {1, 1024, 2048};
    printf("%-36s", "ns per instruction, grid =");
    for (int g : grids) printf(" %8d", g);
    printf("\n");
    for (const Case &c : cases) {
        printf("%-36s", c.name);
        for (int grid : grids) {
            hipLaunchKernelGGL(c.fn, dim3(grid), dim3(64), 0, 0, out, 1.0);   // warm
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(c.fn, dim3(grid), dim3(64), 0, 0, out, 1.0);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %8.3f", ms * 1e6 / ((double)ITER * c.per_iter));
        }
        printf("\n");
    }
    return 0;
}
