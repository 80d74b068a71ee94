// LDS access-pattern microbenchmark for gfx950: cycles per wave-level LDS
// instruction for aligned / misaligned 128-, 64- and 32-bit accesses at the
// strides the resampler's tile layouts produce.  Build:
//   hipcc --offload-arch=gfx950 -O3 -o lds_patterns lds_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));

constexpr int kRepeat = 64, kLoops = 16;

template <int WIDTH, bool WRITE>
__global__ __launch_bounds__(256) void probe (int col_stride, int quad_stride, int shift, long long * cycles, float * sink, int * check)
{
    extern __shared__ float lds [];

    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = static_cast<float>(i);
    __syncthreads();

    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int const col = lane & 15, quad = lane >> 4;
    unsigned const address = 4u * (col * col_stride + quad * quad_stride + shift + wave * 4);

    float4_t sum = { 0, 0, 0, 0 };
    long long const start = __builtin_readcyclecounter();

    for (int loop = 0; loop < kLoops; loop++)
    {
        #pragma unroll
        for (int r = 0; r < kRepeat; r++)
        {
            if (WRITE)
            {
                if (WIDTH == 4) asm volatile ("ds_write_b128 %0, %1" :: "v"(address), "v"(sum) : "memory");
                if (WIDTH == 2) asm volatile ("ds_write_b64 %0, %1" :: "v"(address), "v"(sum.xy) : "memory");
                if (WIDTH == 1) asm volatile ("ds_write_b32 %0, %1" :: "v"(address), "v"(sum.x) : "memory");
            }
            else
            {
                float4_t v = { 0, 0, 0, 0 };
                float2_t w;
                float x;
                if (WIDTH == 4) { asm volatile ("ds_read_b128 %0, %1" : "=v"(v) : "v"(address) : "memory"); }
                if (WIDTH == 2) { asm volatile ("ds_read_b64 %0, %1" : "=v"(w) : "v"(address) : "memory"); v.x = w.x; v.y = w.y; }
                if (WIDTH == 1) { asm volatile ("ds_read_b32 %0, %1" : "=v"(x) : "v"(address) : "memory"); v.x = x; }
                if ((r & 15) == 15) { asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory"); sum += v; }
            }
        }
    }

    asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long const stop = __builtin_readcyclecounter();

    if (lane == 0) cycles[blockIdx.x * 4 + wave] = stop - start;
    sink[blockIdx.x * 256 + threadIdx.x] = sum.x + sum.y + sum.z + sum.w;

    if (!WRITE && WIDTH == 4 && blockIdx.x == 0)
    {
        // correctness of a possibly misaligned 128-bit read
        float4_t v;
        asm volatile ("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(address) : "memory");
        int const first = col * col_stride + quad * quad_stride + shift + wave * 4;
        bool const good = v.x == first && v.y == first + 1 && v.z == first + 2 && v.w == first + 3;
        if (!good) atomicAdd(check, 1);
    }
}

template <int WIDTH, bool WRITE>
static void run (char const * label, int col_stride, int quad_stride, int shift, int blocks)
{
    long long * cycles; float * sink; int * check;
    hipMalloc(&cycles, blocks * 4 * sizeof(long long));
    hipMalloc(&sink, blocks * 256 * sizeof(float));
    hipMalloc(&check, sizeof(int));
    hipMemset(check, 0, sizeof(int));
    probe<WIDTH, WRITE><<<blocks, 256, 65536>>>(col_stride, quad_stride, shift, cycles, sink, check);
    probe<WIDTH, WRITE><<<blocks, 256, 65536>>>(col_stride, quad_stride, shift, cycles, sink, check);
    if (hipDeviceSynchronize() != hipSuccess) { std::printf("%s: launch failed\n", label); std::exit(1); }
    std::vector<long long> host(blocks * 4);
    int bad = 0;
    hipMemcpy(host.data(), cycles, host.size() * sizeof(long long), hipMemcpyDeviceToHost);
    hipMemcpy(&bad, check, sizeof(int), hipMemcpyDeviceToHost);
    double mean = 0;
    for (long long c : host) mean += static_cast<double>(c);
    mean /= static_cast<double>(host.size());
    // four waves share one CU's LDS: cycles of LDS time per wave-instruction
    std::printf("%-44s %s b%-3d stride %4d quad %3d shift %d : %7.2f cycles per instruction per CU (4 waves), %d bad lanes\n",
        label, WRITE ? "write" : "read ", WIDTH * 32, col_stride, quad_stride, shift,
        mean / (kRepeat * kLoops) / 4.0, bad);
    hipFree(cycles); hipFree(sink); hipFree(check);
}

int main ()
{
    int const blocks = 256;   // one workgroup per CU (64 KB of LDS each keeps it to one or two)
    run<4, false>("linear, aligned", 4, 64, 0, blocks);
    run<4, false>("columns 148 apart (aligned, banks spread)", 148, 12, 0, blocks);
    run<4, false>("columns 147 apart (misaligned)", 147, 12, 0, blocks);
    run<4, false>("columns 147 apart, odd start", 147, 12, 1, blocks);
    run<4, false>("columns 294 apart", 294, 12, 0, blocks);
    run<4, false>("columns 160 apart (aligned, same banks)", 160, 12, 0, blocks);
    run<2, false>("columns 147 apart", 147, 12, 0, blocks);
    run<2, false>("columns 148 apart", 148, 12, 0, blocks);
    run<1, false>("rows of 16 columns (current layout)", 1, 16, 0, blocks);
    run<1, false>("columns 147 apart", 147, 12, 0, blocks);
    run<4, true>("linear, aligned", 4, 64, 0, blocks);
    run<4, true>("columns 148 apart", 148, 4, 0, blocks);
    run<4, true>("columns 147 apart (misaligned)", 147, 4, 0, blocks);
    run<4, true>("columns 160 apart", 160, 4, 0, blocks);
    run<2, true>("columns 147 apart", 147, 4, 0, blocks);
    run<1, true>("rows of 16 columns (current layout)", 1, 16, 0, blocks);
    run<1, true>("columns 147 apart", 147, 4, 0, blocks);
    return 0;
}
