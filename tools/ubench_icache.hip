// ubench_icache.hip -- instruction-fetch behaviour of a loop body of KB kilobytes (8-byte VALU instructions, no branches
// inside the body): ns per instruction for a lone wave, one wave per CU, one per SIMD; waves start staggered so that they
// sit at different places of the body.  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_icache.hip -o tools/ubench_icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define I1 asm volatile("v_and_b32_e64 %0, %0, %2\n v_and_b32_e64 %1, %1, %2" : "+v"(u), "+v"(v) : "v"(w));
#define I8 I1 I1 I1 I1 I1 I1 I1 I1
#define I64 I8 I8 I8 I8 I8 I8 I8 I8          /* 64 x 2 x 8 B = 1 KB */
#define KB1 I64
#define KB4 KB1 KB1 KB1 KB1
#define KB16 KB4 KB4 KB4 KB4

template <int KB>
__global__ __launch_bounds__(64) void k_body(int *out, int iters, int stagger)
{
    int u = threadIdx.x, v = u + 1, w = -1;
    for (int s = 0; s < (int)(blockIdx.x % 16) * stagger; ++s) asm volatile("s_sleep 8");
    for (int it = 0; it < iters; ++it) {
        if constexpr (KB == 4) { KB4 }
        else if constexpr (KB == 8) { KB4 KB4 }
        else if constexpr (KB == 16) { KB16 }
        else if constexpr (KB == 24) { KB16 KB4 KB4 }
        else if constexpr (KB == 32) { KB16 KB16 }
        else if constexpr (KB == 40) { KB16 KB16 KB4 KB4 }
        else if constexpr (KB == 48) { KB16 KB16 KB16 }
        else if constexpr (KB == 64) { KB16 KB16 KB16 KB16 }
        else if constexpr (KB == 96) { KB16 KB16 KB16 KB16 KB16 KB16 }
        else if constexpr (KB == 128) { KB16 KB16 KB16 KB16 KB16 KB16 KB16 KB16 }
    }
    if (u + v == 123456789) out[threadIdx.x] = u + v;
}

struct Case { int kb; void (*fn)(int *, int, int); };

int main()
{
    int *out;
    CHECK(hipMalloc(&out, 64 * sizeof(int)));
    const Case cases[] = {{4, k_body<4>}, {8, k_body<8>}, {16, k_body<16>}, {24, k_body<24>}, {32, k_body<32>}, {40, k_body<40>},
                          {48, k_body<48>}, {64, k_body<64>}, {96, k_body<96>}, {128, k_body<128>}};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grids[] = {1, 256, 512, 1024, 2048};
    for (int stagger : {0, 7}) {
        printf("stagger %d: ns per instruction of one wave; body KB vs grid", stagger);
        for (int g : grids) printf(" %8d", g);
        printf("\n");
        for (const Case &c : cases) {
            printf("%-58d", c.kb);
            const int iters = 4096 / c.kb * 8;   // 32 MB of instructions per wave
            for (int grid : grids) {
                hipLaunchKernelGGL(c.fn, dim3(grid), dim3(64), 0, 0, out, iters, stagger);
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(c.fn, dim3(grid), dim3(64), 0, 0, out, iters, stagger);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                printf(" %8.3f", ms * 1e6 / ((double)iters * c.kb * 128));
            }
            printf("\n");
        }
    }
    return 0;
}
