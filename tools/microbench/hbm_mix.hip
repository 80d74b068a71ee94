// What the memory system of one MI355X sustains for the traffic of the resampler's
// headline step: 3.6 GB read and 3.9 GB written per launch, in 16-byte accesses.
//   1. read only, write only, and a flat copy, one 16-byte access per lane and
//      loop trip, grid-stride, for several grid sizes;
//   2. the same copy in the resampler's own pattern: 2048 rows (stream x channel)
//      of 1.76 MB in and 1.92 MB out, a workgroup walking over 10 KB tiles of one
//      row with a stride, 1024 workgroups of 1024 lanes.
// Build: hipcc --offload-arch=gfx950 -O3 -o hbm_mix hbm_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float float4_t __attribute__((ext_vector_type(4)));

#define CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { printf("%s: %s\n", #call, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void read_only (float4_t const * __restrict__ in, long long count, float * sink)
{
    float4_t sum = { 0, 0, 0, 0 };

    for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < count; i += static_cast<long long>(gridDim.x) * blockDim.x)
    {
        sum += __builtin_nontemporal_load(in + i);
    }

    if (sum[0] + sum[1] + sum[2] + sum[3] == 12345.678f) *sink = 1.0f;
}

__global__ __launch_bounds__(1024) void write_only (float4_t * __restrict__ out, long long count)
{
    float4_t const value = { 1.0f, 2.0f, 3.0f, static_cast<float>(threadIdx.x) };

    for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < count; i += static_cast<long long>(gridDim.x) * blockDim.x)
    {
        out[i] = value;
    }
}

__global__ __launch_bounds__(1024) void copy_flat (float4_t const * __restrict__ in, float4_t * __restrict__ out, long long count)
{
    for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < count; i += static_cast<long long>(gridDim.x) * blockDim.x)
    {
        out[i] = in[i];
    }
}

// rows x (tiles of 2352 floats in, 2560 floats out); workgroup w: row w % rows,
// tiles w / rows, + stride, ...; 640 lanes move a tile (588 load, 640 store)
__global__ __launch_bounds__(1024) void copy_tiles (float const * __restrict__ in, float * __restrict__ out, int rows, int tiles, int stride, long long in_row, long long out_row)
{
    int const row = blockIdx.x % rows;
    float const * const from = in + row * in_row;
    float * const to = out + row * out_row;
    int const lane = threadIdx.x;

    for (int tile = blockIdx.x / rows; tile < tiles; tile += stride)
    {
        float4_t v = { 0, 0, 0, 0 };

        if (lane < 588) v = *reinterpret_cast<float4_t const *>(from + 2352LL * tile + 4 * lane);
        if (lane < 640) *reinterpret_cast<float4_t *>(to + 2560LL * tile + 4 * lane) = v;
    }
}

// the tile pattern with the loads of BATCH tiles in front of their stores
template <int BATCH>
__global__ __launch_bounds__(1024) void copy_tiles_batched (float const * __restrict__ in, float * __restrict__ out, int rows, int tiles, int stride, long long in_row, long long out_row)
{
    int const row = blockIdx.x % rows;
    float const * const from = in + row * in_row;
    float * const to = out + row * out_row;
    int const lane = threadIdx.x;

    for (int tile = blockIdx.x / rows; tile < tiles; tile += stride * BATCH)
    {
        float4_t v [BATCH];

        #pragma unroll
        for (int b = 0; b < BATCH; b++)
        {
            v[b] = float4_t{ 0, 0, 0, 0 };
            if (lane < 588 && tile + b * stride < tiles) v[b] = *reinterpret_cast<float4_t const *>(from + 2352LL * (tile + b * stride) + 4 * lane);
        }

        #pragma unroll
        for (int b = 0; b < BATCH; b++)
        {
            if (lane < 640 && tile + b * stride < tiles) *reinterpret_cast<float4_t *>(to + 2560LL * (tile + b * stride) + 4 * lane) = v[b];
        }
    }
}

// the same with twice as much read as written: tiles of 5120 floats in, 2560 out
__global__ __launch_bounds__(1024) void copy_tiles_2to1 (float const * __restrict__ in, float * __restrict__ out, int rows, int tiles, int stride, long long in_row, long long out_row)
{
    int const row = blockIdx.x % rows;
    float const * const from = in + row * in_row;
    float * const to = out + row * out_row;
    int const lane = threadIdx.x;

    for (int tile = blockIdx.x / rows; tile < tiles; tile += stride)
    {
        float4_t a = *reinterpret_cast<float4_t const *>(from + 5120LL * tile + 4 * lane);
        float4_t b = { 0, 0, 0, 0 };

        if (lane < 256) b = *reinterpret_cast<float4_t const *>(from + 5120LL * tile + 4096 + 4 * lane);
        if (lane < 640) *reinterpret_cast<float4_t *>(to + 2560LL * tile + 4 * lane) = a + b;
    }
}

int main ()
{
    long long const in_bytes = 3640000000LL / 16 * 16, out_bytes = 3940000000LL / 16 * 16;
    float * in, * out, * sink;

    CHECK(hipMalloc(&in, in_bytes + (1 << 20)));
    CHECK(hipMalloc(&out, out_bytes + (1 << 20)));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(in, 0, in_bytes));
    CHECK(hipMemset(out, 0, out_bytes));

    hipEvent_t start, stop;

    CHECK(hipEventCreate(&start));
    CHECK(hipEventCreate(&stop));

    auto timed = [&] (char const * name, double bytes, auto launch)
    {
        for (int i = 0; i < 3; i++) launch();

        CHECK(hipEventRecord(start, 0));
        for (int i = 0; i < 20; i++) launch();
        CHECK(hipEventRecord(stop, 0));
        CHECK(hipEventSynchronize(stop));

        float ms = 0;

        CHECK(hipEventElapsedTime(&ms, start, stop));
        CHECK(hipGetLastError());
        printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms / 20, bytes / (ms / 20) * 1e-6);
    };

    int const grids [] = { 256, 512, 1024, 2048, 4096, 16384 };
    char name [96];

    for (int threads : { 256, 1024 })
    {
        for (int grid : grids)
        {
            snprintf(name, sizeof(name), "read only, %d x %d", grid, threads);
            timed(name, static_cast<double>(in_bytes), [&] { read_only<<<grid, threads>>>(reinterpret_cast<float4_t const *>(in), in_bytes / 16, sink); });
        }

        for (int grid : grids)
        {
            snprintf(name, sizeof(name), "write only, %d x %d", grid, threads);
            timed(name, static_cast<double>(out_bytes), [&] { write_only<<<grid, threads>>>(reinterpret_cast<float4_t *>(out), out_bytes / 16); });
        }

        for (int grid : grids)
        {
            snprintf(name, sizeof(name), "flat copy, %d x %d", grid, threads);
            timed(name, 2.0 * in_bytes, [&] { copy_flat<<<grid, threads>>>(reinterpret_cast<float4_t const *>(in), reinterpret_cast<float4_t *>(out), in_bytes / 16); });
        }
    }

    // the resampler's pattern: 2048 rows of 441088 frames in, 480000 out: 188 tiles
    {
        int const rows = 2048, tiles = 187;
        long long const in_row = 441088, out_row = 480000;

        for (int workgroups : { 1024, 2048, 4096, 8192 })
        {
            int const stride = workgroups / rows > 0 ? workgroups / rows : 1;
            int const grid = rows * stride;

            snprintf(name, sizeof(name), "tiles, %d workgroups x 1024", grid);
            timed(name, 4.0 * rows * tiles * (2352.0 + 2560.0),
                  [&] { copy_tiles<<<grid, 1024>>>(in, out, rows, tiles, stride, in_row, out_row); });
        }
    }

    {
        int const rows = 2048, tiles = 187;
        long long const in_row = 441088, out_row = 480000;
        double const bytes = 4.0 * rows * tiles * (2352.0 + 2560.0);

        timed("tiles, 2048 workgroups, loads of 2 tiles batched", bytes, [&] { copy_tiles_batched<2><<<rows, 1024>>>(in, out, rows, tiles, 1, in_row, out_row); });
        timed("tiles, 2048 workgroups, loads of 4 tiles batched", bytes, [&] { copy_tiles_batched<4><<<rows, 1024>>>(in, out, rows, tiles, 1, in_row, out_row); });
        timed("tiles, 2048 workgroups, loads of 8 tiles batched", bytes, [&] { copy_tiles_batched<8><<<rows, 1024>>>(in, out, rows, tiles, 1, in_row, out_row); });
        timed("tiles, 2048 workgroups, loads of 16 tiles batched", bytes, [&] { copy_tiles_batched<16><<<rows, 1024>>>(in, out, rows, tiles, 1, in_row, out_row); });
    }

    // a half-band decimator's pattern: 4096 x 8 rows of 960 000 frames in and 480 000 out
    // would be 31 GB; a tenth of it here: rows of 96 000 in (tiles of 5120) and 48 000 out (2560)
    {
        int const rows = 8192, tiles = 18;   // 8192 x 98 048 x 4 B = 3.2 GB in, 1.6 GB out: inside both buffers
        long long const in_row = 96000 + 2048, out_row = 48000 + 1024;

        // (3.2 GB + 1.6 GB: fits the two buffers)
        snprintf(name, sizeof(name), "2:1 tiles, %d workgroups x 1024", rows);
        timed(name, 4.0 * rows * tiles * (5120.0 + 2560.0), [&]
        {
            copy_tiles_2to1<<<rows, 1024>>>(in, out, rows, tiles, 1, in_row, out_row);
        });
    }

    return 0;
}
