// ubench_select.hip -- which VALU / SALU instructions keep their rate when every SIMD of a CU runs a wave?
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_select.hip -o tools/ubench_select ; run on the MI355X.
// One 64-lane wave per block; grid 1 = a lone wave, 256 = one wave per CU, 1024 = one per SIMD, 2048 = two per SIMD.
// Reported: ns per instruction of ONE wave (kernel time / instructions per wave).  A rate that degrades from grid 256 to
// grid 1024 is limited per CU, not per SIMD (motivation: the select-chain heavy BDF kernel runs 2.1x slower per wave
// with four waves on a CU than with one).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 2000;
#define REP8(X) X X X X X X X X
#define REP16(X) REP8(X) REP8(X)

template <int K>
__global__ __launch_bounds__(64) void k_bench(double *out, double seed)
{
    double a = seed + threadIdx.x, b = 1.0000001, c = 0.5, d = a + 1;
    int u = threadIdx.x, v = u + 1, w = u + 2, m = (threadIdx.x & 1) ? -1 : 0;
    for (int it = 0; it < ITER; ++it) {
        if constexpr (K == 0) {          // v_cndmask_b32 (VOP2, mask in vcc), two independent
            REP16(asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc" : "+v"(u), "+v"(v) : "v"(w) : "vcc");)
        } else if constexpr (K == 1) {   // v_cndmask_b32 (VOP3, mask in an SGPR pair)
            REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %2, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]" : "+v"(u), "+v"(v) : "v"(w) : "s20", "s21");)
        } else if constexpr (K == 2) {   // v_bfi_b32 (mask in a VGPR)
            REP16(asm volatile("v_bfi_b32 %0, %3, %0, %2\n v_bfi_b32 %1, %3, %1, %2" : "+v"(u), "+v"(v) : "v"(w), "v"(m));)
        } else if constexpr (K == 3) {   // v_cmp_lt_f64 -> vcc
            REP16(asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %0" : : "v"(a), "v"(d) : "vcc");)
        } else if constexpr (K == 4) {   // v_cmp_lt_f64 -> SGPR pair
            REP16(asm volatile("v_cmp_lt_f64_e64 s[20:21], %0, %1\n v_cmp_lt_f64_e64 s[22:23], %1, %0" : : "v"(a), "v"(d) : "s20", "s21", "s22", "s23");)
        } else if constexpr (K == 5) {   // v_and_b32 (plain 32-bit VALU)
            REP16(asm volatile("v_and_b32 %0, %0, %2\n v_and_b32 %1, %1, %2" : "+v"(u), "+v"(v) : "v"(w));)
        } else if constexpr (K == 6) {   // v_mov_b32
            REP16(asm volatile("v_mov_b32 %0, %2\n v_mov_b32 %1, %2" : "=v"(u), "=v"(v) : "v"(w));)
        } else if constexpr (K == 7) {   // v_max_f64
            REP16(asm volatile("v_max_f64 %0, %0, %2\n v_max_f64 %1, %1, %2" : "+v"(a), "+v"(d) : "v"(b));)
        } else if constexpr (K == 8) {   // v_fma_f64, two independent chains
            REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));)
        } else if constexpr (K == 9) {   // v_readfirstlane_b32
            REP16(asm volatile("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1" : : "v"(u), "v"(v) : "s20", "s21");)
        } else if constexpr (K == 10) {  // s_and_b64 (SALU)
            REP16(asm volatile("s_and_b64 s[20:21], s[20:21], exec\n s_and_b64 s[22:23], s[22:23], exec" : : : "s20", "s21", "s22", "s23", "scc");)
        } else if constexpr (K == 11) {  // exec save / restore pair (what a divergent branch costs)
            REP16(asm volatile("s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]" : : : "s20", "s21", "scc", "vcc");)
        } else if constexpr (K == 12) {  // v_cmp + dependent v_cndmask (the select idiom)
            REP16(asm volatile("v_cmp_lt_f64 vcc, %2, %3\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u) : "v"(w), "v"(a), "v"(d) : "vcc");)
        } else if constexpr (K == 13) {  // v_fma_f64 with an SGPR operand
            REP16(asm volatile("v_fma_f64 %0, %0, s[20:21], %2\n v_fma_f64 %1, %1, s[20:21], %2" : "+v"(a), "+v"(d) : "v"(c) : "s20", "s21");)
        } else if constexpr (K == 14) {  // v_fma_f64 with a literal/inline constant
            REP16(asm volatile("v_fma_f64 %0, %0, 1.0, %2\n v_fma_f64 %1, %1, 1.0, %2" : "+v"(a), "+v"(d) : "v"(c));)
        } else if constexpr (K == 15) {  // s_mov_b32 literal
            REP16(asm volatile("s_mov_b32 s20, 0x3ff00000\n s_mov_b32 s21, 0x40000000" : : : "s20", "s21");)
        } else if constexpr (K == 16) {  // taken branch (s_branch over one instruction)
            REP16(asm volatile("s_branch 1\n s_nop 0\n s_branch 1\n s_nop 0");)
        } else if constexpr (K == 17) {  // the f64 select idiom: one compare, two v_cndmask (lo / hi dword) on the same vcc
            REP8(asm volatile("v_cmp_neq_f64 vcc, 0, %3\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_and_b32 %2, %2, %2" : "+v"(u), "+v"(v), "+v"(w) : "v"(a) : "vcc");)
        } else if constexpr (K == 18) {  // one compare, six v_cndmask on the same vcc, one filler
            REP8(asm volatile("v_cmp_neq_f64 vcc, 0, %3\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_and_b32 %2, %2, %2" : "+v"(u), "+v"(v), "+v"(w) : "v"(a) : "vcc");)
        } else if constexpr (K == 19) {  // v_readlane_b32 / v_writelane_b32 (SGPR spill slots in a VGPR)
            REP8(asm volatile("v_writelane_b32 %0, s20, 3\n v_readlane_b32 s21, %0, 5\n v_writelane_b32 %1, s22, 7\n v_readlane_b32 s23, %1, 9" : "+v"(u), "+v"(v) : : "s20", "s21", "s22", "s23");)
        } else if constexpr (K == 20) {  // v_div_scale / v_div_fmas / v_div_fixup / v_rcp (IEEE division skeleton, independent)
            REP8(asm volatile("v_div_scale_f64 %0, vcc, %2, %2, %3\n v_rcp_f64 %1, %2\n v_div_fmas_f64 %0, %0, %2, %3\n v_div_fixup_f64 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c) : "vcc");)
        } else if constexpr (K == 21) {  // v_lshl_add_u64 / v_mov_b64
            REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %2\n v_mov_b64 %1, %2\n v_lshl_add_u64 %0, %0, 0, %2\n v_mov_b64 %1, %2" : "+v"(a), "+v"(d) : "v"(b));)
        } else if constexpr (K == 22) {  // divergent region: saveexec, cbranch_execz (not taken), body, restore
            REP8(asm volatile("v_cmp_neq_f64 vcc, 0, %1\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1\n v_and_b32 %0, %0, %0\n s_or_b64 exec, exec, s[20:21]" : "+v"(u) : "v"(a) : "vcc", "s20", "s21", "scc");)
        } else if constexpr (K == 23) {  // one compare -> vcc, six v_cndmask in VOP3 ENCODING that read vcc, one filler
            REP8(asm volatile("v_cmp_neq_f64 vcc, 0, %3\n v_cndmask_b32_e64 %0, %0, %2, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc\n v_cndmask_b32_e64 %0, %0, %2, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc\n v_cndmask_b32_e64 %0, %0, %2, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc\n v_and_b32 %2, %2, %2" : "+v"(u), "+v"(v), "+v"(w) : "v"(a) : "vcc");)
        } else if constexpr (K == 24) {  // one compare -> SGPR pair, six v_cndmask_e64 on that pair, one filler
            REP8(asm volatile("v_cmp_neq_f64_e64 s[20:21], 0, %3\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_and_b32 %2, %2, %2" : "+v"(u), "+v"(v), "+v"(w) : "v"(a) : "s20", "s21");)
        } else if constexpr (K == 25) {  // one compare -> vcc, copy to an SGPR pair (s_mov_b64), six v_cndmask_e64 on the copy
            REP8(asm volatile("v_cmp_neq_f64 vcc, 0, %3\n s_mov_b64 s[20:21], vcc\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_and_b32 %2, %2, %2" : "+v"(u), "+v"(v), "+v"(w) : "v"(a) : "vcc", "s20", "s21");)
        }
    }
    if (a + d + u + v == 12345.678) out[threadIdx.x] = a + d + u + v;
}

struct Case { const char *name; void (*fn)(double *, double); };

int main()
{
    double *out;
    CHECK(hipMalloc(&out, 64 * sizeof(double)));
    const Case cases[] = {
        {"v_cndmask_b32 vcc", k_bench<0>}, {"v_cndmask_b32_e64 sgpr mask", k_bench<1>}, {"v_bfi_b32", k_bench<2>},
        {"v_cmp_lt_f64 -> vcc", k_bench<3>}, {"v_cmp_lt_f64 -> sgpr pair", k_bench<4>}, {"v_and_b32", k_bench<5>}, {"v_mov_b32", k_bench<6>},
        {"v_max_f64", k_bench<7>}, {"v_fma_f64 x2", k_bench<8>}, {"v_readfirstlane_b32", k_bench<9>}, {"s_and_b64", k_bench<10>},
        {"saveexec + restore (pair)", k_bench<11>}, {"v_cmp -> v_cndmask (pair)", k_bench<12>}, {"v_fma_f64 sgpr operand", k_bench<13>},
        {"v_fma_f64 inline constant", k_bench<14>}, {"s_mov_b32 literal", k_bench<15>}, {"taken s_branch + skipped nop", k_bench<16>},
        {"cmp + 2 cndmask + and (x4 per 32)", k_bench<17>}, {"cmp + 6 cndmask + and (x8 per 32)", k_bench<18>}, {"writelane/readlane", k_bench<19>},
        {"div_scale/rcp/div_fmas/div_fixup", k_bench<20>}, {"v_lshl_add_u64 / v_mov_b64", k_bench<21>}, {"divergent region (5 instr per 32/8)", k_bench<22>},
        {"cmp->vcc + 6 cndmask_e64 vcc + and", k_bench<23>}, {"cmp->sgpr + 6 cndmask_e64 sgpr + and", k_bench<24>},
        {"cmp->vcc, s_mov, 6 cndmask_e64 sgpr", k_bench<25>},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grids[] = {1, 256, 512, 1024, 2048};
    printf("%-32s", "ns per instruction, grid =");
    for (int g : grids) printf(" %8d", g);
    printf("\n");
    for (const Case &c : cases) {
        printf("%-32s", c.name);
        for (int grid : grids) {
            hipLaunchKernelGGL(c.fn, dim3(grid), dim3(64), 0, 0, out, 1.0);   // warm
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(c.fn, dim3(grid), dim3(64), 0, 0, out, 1.0);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %8.3f", ms * 1e6 / ((double)ITER * 32));
        }
        printf("\n");
    }
    return 0;
}
