// What v_mfma_f32_16x16x32_f16 does with split float32 operands on gfx950:
//   1. operand layout, checked with exact integer data;
//   2. subnormal half inputs (are they flushed?);
//   3. accuracy of a 128-term dot product of float32 data carried as two halves
//      per value (x = h + l / 2048) with three products (h h, h l, l h), against
//      double precision and against the float32 matrix instruction;
//   4. cycles per instruction on one dependent chain and on three chains.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f16_split mfma_f16_split.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));

constexpr int kK = 128;   // terms per dot product: four 32-deep instructions

// A [16][kK], B [kK][16] as float32 in memory; D [16][16].  mode 0: float32
// instruction; 1: halves, high parts only; 2: three products.
__global__ __launch_bounds__(64) void product (float const * A, float const * B, float * D, int mode)
{
    int const lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    float4_t main_sum = { 0, 0, 0, 0 }, cross = { 0, 0, 0, 0 };

    if (mode == 0)
    {
        for (int s = 0; s < kK / 4; s++)
        {
            main_sum = __builtin_amdgcn_mfma_f32_16x16x4f32(A[n * kK + 4 * s + g], B[(4 * s + g) * 16 + n], main_sum, 0, 0, 0);
        }
    }
    else
    {
        for (int s = 0; s < kK / 32; s++)
        {
            half8_t ah, al, bh, bl;

            for (int j = 0; j < 8; j++)
            {
                float const a = A[n * kK + 32 * s + 8 * g + j], b = B[(32 * s + 8 * g + j) * 16 + n];
                _Float16 const a_high = static_cast<_Float16>(a), b_high = static_cast<_Float16>(b);

                ah[j] = a_high;
                bh[j] = b_high;
                al[j] = static_cast<_Float16>((a - static_cast<float>(a_high)) * 2048.0f);
                bl[j] = static_cast<_Float16>((b - static_cast<float>(b_high)) * 2048.0f);
            }

            main_sum = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, main_sum, 0, 0, 0);

            if (mode == 2)
            {
                cross = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, cross, 0, 0, 0);
                cross = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, cross, 0, 0, 0);
            }
        }
    }

    for (int r = 0; r < 4; r++) D[(4 * g + r) * 16 + n] = main_sum[r] + cross[r] * (1.0f / 2048.0f);
}

__global__ __launch_bounds__(64) void pace (long long * cycles, float * sink, int chains)
{
    half8_t a, b;

    for (int j = 0; j < 8; j++) { a[j] = static_cast<_Float16>(0.001f * (threadIdx.x + j)); b[j] = static_cast<_Float16>(0.002f * (threadIdx.x - j)); }

    float4_t c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0;
    long long const start = __builtin_readcyclecounter();

    for (int i = 0; i < 256; i++)
    {
        if (chains == 1)
        {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        }
        else
        {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        }
    }

    long long const stop = __builtin_readcyclecounter();

    if (threadIdx.x == 0) cycles[blockIdx.x] = stop - start;
    sink[blockIdx.x * 64 + threadIdx.x] = c0[0] + c1[1] + c2[2];
}

#define CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { printf("%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

int main ()
{
    float * A, * B, * D;
    long long * cycles;
    float * sink;

    CHECK(hipMalloc(&A, 16 * kK * 4));
    CHECK(hipMalloc(&B, 16 * kK * 4));
    CHECK(hipMalloc(&D, 256 * 4));
    CHECK(hipMalloc(&cycles, 1024 * 8));
    CHECK(hipMalloc(&sink, 1024 * 64 * 4));

    std::vector<float> a(16 * kK), b(16 * kK), d(256);

    // 1. layout: small integers, asymmetric
    for (int m = 0; m < 16; m++) for (int k = 0; k < kK; k++) a[m * kK + k] = static_cast<float>((m * 7 + k * 3) % 11 - 5);
    for (int k = 0; k < kK; k++) for (int n = 0; n < 16; n++) b[k * 16 + n] = static_cast<float>((k * 5 + n * 2) % 13 - 6);

    CHECK(hipMemcpy(A, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(B, b.data(), b.size() * 4, hipMemcpyHostToDevice));

    for (int mode = 0; mode < 3; mode++)
    {
        product<<<1, 64>>>(A, B, D, mode);
        CHECK(hipMemcpy(d.data(), D, 1024, hipMemcpyDeviceToHost));

        int wrong = 0;

        for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++)
        {
            double want = 0;
            for (int k = 0; k < kK; k++) want += static_cast<double>(a[m * kK + k]) * b[k * 16 + n];
            if (d[m * 16 + n] != static_cast<float>(want)) wrong++;
        }

        printf("layout, mode %d: %d of 256 wrong\n", mode, wrong);
    }

    // 2. subnormal halves: a = 2^-20 (subnormal as a half), b = 1
    for (auto & v : a) v = 0.0f;
    for (auto & v : b) v = 0.0f;
    for (int m = 0; m < 16; m++) a[m * kK + m] = ldexpf(1.0f, -20);
    for (int n = 0; n < 16; n++) for (int k = 0; k < 16; k++) b[k * 16 + n] = 1.0f;

    CHECK(hipMemcpy(A, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(B, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    product<<<1, 64>>>(A, B, D, 1);
    CHECK(hipMemcpy(d.data(), D, 1024, hipMemcpyDeviceToHost));
    printf("subnormal half operand 2^-20 x 1: got %g (2^-20 = %g)\n", d[0], ldexp(1.0, -20));

    // 3. accuracy on filter-like data
    srand(7);

    double worst [3] = { 0, 0, 0 }, sum_sq [3] = { 0, 0, 0 };
    long count = 0;

    for (int trial = 0; trial < 200; trial++)
    {
        // coefficients: a windowed sinc-ish row, data: uniform [-1, 1)
        for (int m = 0; m < 16; m++) for (int k = 0; k < kK; k++)
        {
            double const t = (k - 64 + 0.37 * m + 0.01 * trial) * 0.92;
            double const w = 0.5 + 0.5 * cos(M_PI * (k - 64) / 64.0);
            a[m * kK + k] = static_cast<float>((fabs(t) < 1e-9 ? 1.0 : sin(M_PI * t) / (M_PI * t)) * w * 0.92);
        }

        for (auto & v : b) v = static_cast<float>(rand() / (RAND_MAX + 1.0) * 2.0 - 1.0);

        CHECK(hipMemcpy(A, a.data(), a.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(B, b.data(), b.size() * 4, hipMemcpyHostToDevice));

        for (int mode = 0; mode < 3; mode++)
        {
            product<<<1, 64>>>(A, B, D, mode);
            CHECK(hipMemcpy(d.data(), D, 1024, hipMemcpyDeviceToHost));

            for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++)
            {
                double want = 0;
                for (int k = 0; k < kK; k++) want += static_cast<double>(a[m * kK + k]) * b[k * 16 + n];

                double const e = fabs(d[m * 16 + n] - want);

                if (e > worst[mode]) worst[mode] = e;
                sum_sq[mode] += e * e;
            }
        }

        count += 256;
    }

    for (int mode = 0; mode < 3; mode++)
    {
        printf("accuracy, mode %d (%s): worst %.3g, rms %.3g over %ld dot products of %d terms\n", mode,
               mode == 0 ? "float32 instruction" : mode == 1 ? "high halves only" : "three half products",
               worst[mode], sqrt(sum_sq[mode] / count), count, kK);
    }

    // 4. pace
    for (int chains = 1; chains <= 3; chains += 2)
    {
        pace<<<1024, 64>>>(cycles, sink, chains);

        std::vector<long long> c(1024);

        CHECK(hipMemcpy(c.data(), cycles, 1024 * 8, hipMemcpyDeviceToHost));

        double mean = 0;
        for (auto v : c) mean += static_cast<double>(v);
        printf("pace, %d chain(s): %.1f cycles per instruction (one wave per workgroup, 1024 workgroups)\n", chains, mean / 1024 / 768);
    }

    return 0;
}
