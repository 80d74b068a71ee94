// What the vector ALU of a gfx950 SIMD sustains for the instruction streams of the
// exact resampling kernel: cycles per wave-instruction and SIMD at 1, 2, 4 and 8
// waves per SIMD, for
//   chain    v_mul_f32 (scalar-register coefficient) feeding a v_add_f32 into one
//            accumulator: the half-band stages' stream as the compiler emits it
//   four     the same with four accumulators taking turns
//   vgpr     chain with the coefficient in a vector register
//   fma      independent v_fma_f32 (the peak the data sheet counts)
//   packed   v_pk_mul_f32 / v_pk_add_f32, four accumulators of two floats
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R32(x) R8(x) R8(x) R8(x) R8(x)

constexpr int kLoops = 2000;   // x 64 vector instructions

template <int MODE>
__global__ __launch_bounds__(1024) void probe (long long * cycles, float * sink)
{
    float seed = static_cast<float>(threadIdx.x) * 1e-3f;

    asm volatile
    (
        "v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, 0.5\n v_mov_b32 v13, %0\n"
        "v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n"
        "v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n"
        "v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n"
        "s_mov_b32 s20, 0.5\n s_mov_b32 s21, 0.5\n"
        :: "v"(seed) : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17",
           "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "s20", "s21"
    );

    __syncthreads();

    long long const start = __builtin_readcyclecounter();

    for (int loop = 0; loop < kLoops; loop++)
    {
        if (MODE == 0)
        {
            asm volatile (R32("v_mul_f32 v20, s20, v10\n v_add_f32 v30, v30, v20\n") ::: "v20", "v30");
        }
        else if (MODE == 1)
        {
            asm volatile
            (
                R8("v_mul_f32 v20, s20, v10\n v_mul_f32 v21, s20, v11\n v_mul_f32 v22, s20, v13\n v_mul_f32 v23, s20, v14\n"
                   "v_add_f32 v30, v30, v20\n v_add_f32 v31, v31, v21\n v_add_f32 v32, v32, v22\n v_add_f32 v33, v33, v23\n")
                ::: "v20", "v21", "v22", "v23", "v30", "v31", "v32", "v33"
            );
        }
        else if (MODE == 2)
        {
            asm volatile (R32("v_mul_f32 v20, v12, v10\n v_add_f32 v30, v30, v20\n") ::: "v20", "v30");
        }
        else if (MODE == 3)
        {
            asm volatile
            (
                R8("v_fma_f32 v30, v10, v12, v30\n v_fma_f32 v31, v10, v12, v31\n v_fma_f32 v32, v10, v12, v32\n v_fma_f32 v33, v10, v12, v33\n"
                   "v_fma_f32 v34, v10, v12, v34\n v_fma_f32 v35, v10, v12, v35\n v_fma_f32 v36, v10, v12, v36\n v_fma_f32 v37, v10, v12, v37\n")
                ::: "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37"
            );
        }
        else
        {
            asm volatile
            (
                R8("v_pk_mul_f32 v[20:21], v[10:11], s[20:21]\n v_pk_mul_f32 v[22:23], v[12:13], s[20:21]\n"
                   "v_pk_mul_f32 v[24:25], v[14:15], s[20:21]\n v_pk_mul_f32 v[26:27], v[16:17], s[20:21]\n"
                   "v_pk_add_f32 v[30:31], v[30:31], v[20:21]\n v_pk_add_f32 v[32:33], v[32:33], v[22:23]\n"
                   "v_pk_add_f32 v[34:35], v[34:35], v[24:25]\n v_pk_add_f32 v[36:37], v[36:37], v[26:27]\n")
                ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37"
            );
        }
    }

    long long const stop = __builtin_readcyclecounter();
    float out;

    asm volatile ("v_add_f32 %0, v30, v31\n v_add_f32 %0, %0, v32\n v_add_f32 %0, %0, v36" : "=v"(out));

    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = stop - start;
    if (out == 12345.678f) sink[0] = out;
}

template <int MODE>
static void run (char const * name, long long * cycles, float * sink)
{
    int const shapes [4][2] = { { 256, 256 }, { 256, 512 }, { 256, 1024 }, { 512, 1024 } };
    int const per_simd [4] = { 1, 2, 4, 8 };

    for (int s = 0; s < 4; s++)
    {
        int const blocks = shapes[s][0], threads = shapes[s][1];
        int const waves = blocks * threads / 64;
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);

        probe<MODE><<<blocks, threads>>>(cycles, sink);
        hipDeviceSynchronize();
        hipEventRecord(a);
        probe<MODE><<<blocks, threads>>>(cycles, sink);
        hipEventRecord(b);
        hipDeviceSynchronize();

        float ms = 0;
        hipEventElapsedTime(&ms, a, b);

        std::vector<long long> host(waves);
        hipMemcpy(host.data(), cycles, waves * sizeof(long long), hipMemcpyDeviceToHost);

        double mean = 0; long long most = 0;
        for (long long c : host) { mean += c; most = c > most ? c : most; }
        mean /= waves;

        double const instructions = 64.0 * kLoops;
        printf("%-7s %d waves/SIMD: %6.2f cycles per instruction and SIMD (mean wave %9.0f cycles, slowest %9lld; %.3f ms: %.2f G instr/s/SIMD)\n",
               name, per_simd[s], mean / (instructions * per_simd[s]), mean, most, ms,
               instructions * per_simd[s] / (ms * 1e6));
        hipEventDestroy(a); hipEventDestroy(b);
    }
}

int main ()
{
    long long * cycles; float * sink;
    hipMalloc(&cycles, 512 * 16 * sizeof(long long));
    hipMalloc(&sink, 4);

    run<0>("chain", cycles, sink);
    run<1>("four", cycles, sink);
    run<2>("vgpr", cycles, sink);
    run<3>("fma", cycles, sink);
    run<4>("packed", cycles, sink);
    return 0;
}
