// What a copy of the resampler's headline traffic (3.64 GB read, 3.94 GB written per
// launch, 16 bytes a lane) can reach on one MI355X, swept over what a kernel can
// choose: cache policy of loads and stores (plain / nontemporal), how many loads a
// lane has in flight before its stores, workgroup size, grid size, and how the
// addresses are dealt out (grid-stride, one contiguous chunk per workgroup, or the
// resampler's rows: 2048 (stream, channel) rows of 1.76 MB in and 1.92 MB out).
// Prints one line per variant; tools/microbench/hbm_sweep.sh keeps the listing.
// Build: hipcc --offload-arch=gfx950 -O3 -o hbm_sweep hbm_sweep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float float4_t __attribute__((ext_vector_type(4)));

#define CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { printf("%s: %s\n", #call, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ float4_t load16 (float4_t const * p)
{
    return NT ? __builtin_nontemporal_load(p) : *p;
}

template <bool NT> __device__ __forceinline__ void store16 (float4_t * p, float4_t v)
{
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// grid-stride: lane i of the grid takes elements i, i + lanes, ...; UNROLL loads in
// front of their UNROLL stores
template <int UNROLL, bool NTL, bool NTS>
__global__ __launch_bounds__(1024) void copy_stride (float4_t const * __restrict__ in, float4_t * __restrict__ out, long long count)
{
    long long const lanes = static_cast<long long>(gridDim.x) * blockDim.x;

    for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < count; i += lanes * UNROLL)
    {
        float4_t v [UNROLL];

        #pragma unroll
        for (int k = 0; k < UNROLL; k++) if (i + k * lanes < count) v[k] = load16<NTL>(in + i + k * lanes);

        #pragma unroll
        for (int k = 0; k < UNROLL; k++) if (i + k * lanes < count) store16<NTS>(out + i + k * lanes, v[k]);
    }
}

// one contiguous chunk per workgroup
template <int UNROLL, bool NTL, bool NTS>
__global__ __launch_bounds__(1024) void copy_chunk (float4_t const * __restrict__ in, float4_t * __restrict__ out, long long count)
{
    long long const chunk = (count + gridDim.x - 1) / gridDim.x;
    long long const begin = blockIdx.x * chunk, end = begin + chunk < count ? begin + chunk : count;

    for (long long i = begin + threadIdx.x; i < end; i += static_cast<long long>(blockDim.x) * UNROLL)
    {
        float4_t v [UNROLL];

        #pragma unroll
        for (int k = 0; k < UNROLL; k++) if (i + k * blockDim.x < end) v[k] = load16<NTL>(in + i + k * blockDim.x);

        #pragma unroll
        for (int k = 0; k < UNROLL; k++) if (i + k * blockDim.x < end) store16<NTS>(out + i + k * blockDim.x, v[k]);
    }
}

template <bool NTS>
__global__ __launch_bounds__(1024) void write_stride (float4_t * __restrict__ out, long long count)
{
    float4_t const value = { 1.0f, 2.0f, 3.0f, static_cast<float>(threadIdx.x) };
    long long const lanes = static_cast<long long>(gridDim.x) * blockDim.x;

    for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < count; i += lanes) store16<NTS>(out + i, value);
}

template <bool NTS>
__global__ __launch_bounds__(1024) void write_chunk (float4_t * __restrict__ out, long long count)
{
    float4_t const value = { 1.0f, 2.0f, 3.0f, static_cast<float>(threadIdx.x) };
    long long const chunk = (count + gridDim.x - 1) / gridDim.x;
    long long const begin = blockIdx.x * chunk, end = begin + chunk < count ? begin + chunk : count;

    for (long long i = begin + threadIdx.x; i < end; i += blockDim.x) store16<NTS>(out + i, value);
}

// The resampler's rows.  Workgroup w: row w % rows, tiles w / rows, + stride, ...; a
// tile is 2352 floats in (588 lanes of 16 bytes) and 2560 floats out (640 lanes).
// DEPTH tiles' loads are in flight in front of their stores.
template <int DEPTH, bool NTL, bool NTS>
__global__ __launch_bounds__(1024) void copy_rows (float const * __restrict__ in, float * __restrict__ out, int rows, int tiles, int stride, long long in_row, long long out_row)
{
    int const row = blockIdx.x % rows;
    float const * const from = in + row * in_row;
    float * const to = out + row * out_row;
    int const lane = threadIdx.x;

    for (int tile = blockIdx.x / rows; tile < tiles; tile += stride * DEPTH)
    {
        float4_t v [DEPTH];

        #pragma unroll
        for (int b = 0; b < DEPTH; b++)
        {
            v[b] = float4_t{ 0, 0, 0, 0 };
            if (lane < 588 && tile + b * stride < tiles) v[b] = load16<NTL>(reinterpret_cast<float4_t const *>(from + 2352LL * (tile + b * stride) + 4 * lane));
        }

        #pragma unroll
        for (int b = 0; b < DEPTH; b++)
        {
            if (lane < 640 && tile + b * stride < tiles) store16<NTS>(reinterpret_cast<float4_t *>(to + 2560LL * (tile + b * stride) + 4 * lane), v[b]);
        }
    }
}

// The rows again, as the resampler's workgroup is organised: LOADERS waves load,
// each a whole tile of its own (10 x 16 bytes a lane, the last load partly
// masked) and own every LOADERS-th tile; they hand the tile over through LDS; ten
// storing waves store 1 KB runs of it a barrier later.  RING planes in LDS.
template <int LOADERS, bool NTL, bool NTS>
__global__ __launch_bounds__(1024) void copy_rows_teams (float const * __restrict__ in, float * __restrict__ out, int rows, int tiles, int stride, long long in_row, long long out_row)
{
    constexpr int RING = 3;
    __shared__ float4_t planes [RING][640];

    int const row = blockIdx.x % rows;
    float const * const from = in + row * in_row;
    float * const to = out + row * out_row;
    int const wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int const first = blockIdx.x / rows;
    int const items = first < tiles ? (tiles - first + stride - 1) / stride : 0;

    if (wave >= 10)
    {
        int const me = wave - 10;

        if (me >= LOADERS)
        {
            for (int u = 0; u <= items; u++) __syncthreads();
            return;
        }

        float4_t v [10];
        int passed = 0;

        auto issue = [&] (int item)
        {
            long long const base = 2352LL * (first + item * stride);

            #pragma unroll
            for (int j = 0; j < 10; j++)
            {
                int const at = 64 * j + lane;

                v[j] = at < 588 && item < items ? load16<NTL>(reinterpret_cast<float4_t const *>(from + base) + at) : float4_t{ 0, 0, 0, 0 };
            }
        };

        issue(me);

        for (int u = me; u < items; u += LOADERS)
        {
            while (passed < u) { __syncthreads(); passed++; }   // barriers 0 .. u - 1: the plane's last readers are done

            #pragma unroll
            for (int j = 0; j < 10; j++) planes[u % RING][64 * j + lane] = v[j];

            issue(u + LOADERS);
        }

        while (passed <= items) { __syncthreads(); passed++; }
        return;
    }

    // barrier u: item u is in LDS; storing waves store item u behind barrier u
    for (int u = 0; u < items; u++)
    {
        __syncthreads();

        float4_t const v = planes[u % RING][64 * wave + lane];

        store16<NTS>(reinterpret_cast<float4_t *>(to + 2560LL * (first + u * stride)) + 64 * wave + lane, v);
    }

    __syncthreads();
}

// The same workgroup, but the chip sweeps memory front to back: workgroup w takes
// tiles w, w + W, w + 2 W, ... of the list of all tiles, rows of `band` tiles
// one after the other (a row is cut into bands of `band` tiles; band b of every
// row comes before band b + 1 of any), so that at any time the W workgroups
// read one contiguous stretch of a few rows and write another.
template <int LOADERS, bool NTL, bool NTS>
__global__ __launch_bounds__(1024) void copy_sweep_teams (float const * __restrict__ in, float * __restrict__ out, int rows, int tiles, int band, long long in_row, long long out_row)
{
    constexpr int RING = 3;
    __shared__ float4_t planes [RING][640];

    int const wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int const bands = (tiles + band - 1) / band;
    long long const all = static_cast<long long>(rows) * tiles;
    int const items = static_cast<int>((all - blockIdx.x + gridDim.x - 1) / gridDim.x);

    // item -> (row, tile)
    auto place = [&] (int item, int & row, int & tile)
    {
        long long id = blockIdx.x + static_cast<long long>(item) * gridDim.x;
        int b = 0;

        // bands but the last are `band` tiles wide
        long long const full = static_cast<long long>(rows) * band;

        if (id >= full * (bands - 1)) { id -= full * (bands - 1); b = bands - 1; }
        else { b = static_cast<int>(id / full); id -= b * full; }

        int const width = b == bands - 1 ? tiles - band * (bands - 1) : band;

        row = static_cast<int>(id / width);
        tile = b * band + static_cast<int>(id - static_cast<long long>(row) * width);
    };

    if (wave >= 10)
    {
        int const me = wave - 10;

        if (me >= LOADERS)
        {
            for (int u = 0; u <= items; u++) __syncthreads();
            return;
        }

        float4_t v [10];
        int passed = 0;

        auto issue = [&] (int item)
        {
            int row = 0, tile = 0;

            if (item < items) place(item, row, tile);

            float const * const from = in + row * in_row + 2352LL * tile;

            #pragma unroll
            for (int j = 0; j < 10; j++)
            {
                int const at = 64 * j + lane;

                v[j] = at < 588 && item < items ? load16<NTL>(reinterpret_cast<float4_t const *>(from) + at) : float4_t{ 0, 0, 0, 0 };
            }
        };

        issue(me);

        for (int u = me; u < items; u += LOADERS)
        {
            while (passed < u) { __syncthreads(); passed++; }

            #pragma unroll
            for (int j = 0; j < 10; j++) planes[u % RING][64 * j + lane] = v[j];

            issue(u + LOADERS);
        }

        while (passed <= items) { __syncthreads(); passed++; }
        return;
    }

    for (int u = 0; u < items; u++)
    {
        __syncthreads();

        int row, tile;

        place(u, row, tile);

        float4_t const v = planes[u % RING][64 * wave + lane];

        store16<NTS>(reinterpret_cast<float4_t *>(out + row * out_row + 2560LL * tile) + 64 * wave + lane, v);
    }

    __syncthreads();
}

int main (int argc, char ** argv)
{
    bool const quick = argc > 1 && !strcmp(argv[1], "quick");
    long long const in_bytes = 3640000000LL / 16 * 16, out_bytes = 3940000000LL / 16 * 16;
    float * in, * out;

    CHECK(hipMalloc(&in, in_bytes + (1 << 20)));
    CHECK(hipMalloc(&out, out_bytes + (1 << 20)));
    CHECK(hipMemset(in, 0x11, in_bytes));
    CHECK(hipMemset(out, 0, out_bytes));

    hipEvent_t start, stop;

    CHECK(hipEventCreate(&start));
    CHECK(hipEventCreate(&stop));

    auto timed = [&] (char const * name, double bytes, auto launch)
    {
        for (int i = 0; i < 3; i++) launch();

        CHECK(hipEventRecord(start, 0));
        for (int i = 0; i < 10; i++) launch();
        CHECK(hipEventRecord(stop, 0));
        CHECK(hipEventSynchronize(stop));

        float ms = 0;

        CHECK(hipEventElapsedTime(&ms, start, stop));
        CHECK(hipGetLastError());
        printf("%-64s %8.3f ms  %7.1f GB/s\n", name, ms / 10, bytes / (ms / 10) * 1e-6);
        fflush(stdout);
    };

    char name [128];
    float4_t const * const in4 = reinterpret_cast<float4_t const *>(in);
    float4_t * const out4 = reinterpret_cast<float4_t *>(out);
    long long const count = in_bytes / 16;

    // ---- writes alone ----
    if (!quick) for (int threads : { 256, 1024 })
    {
        for (int grid : { 256, 512, 1024, 2048, 8192 })
        {
            snprintf(name, sizeof(name), "write stride plain, %d x %d", grid, threads);
            timed(name, static_cast<double>(out_bytes), [&] { write_stride<false><<<grid, threads>>>(out4, out_bytes / 16); });
            snprintf(name, sizeof(name), "write stride nt,    %d x %d", grid, threads);
            timed(name, static_cast<double>(out_bytes), [&] { write_stride<true><<<grid, threads>>>(out4, out_bytes / 16); });
            snprintf(name, sizeof(name), "write chunk plain,  %d x %d", grid, threads);
            timed(name, static_cast<double>(out_bytes), [&] { write_chunk<false><<<grid, threads>>>(out4, out_bytes / 16); });
            snprintf(name, sizeof(name), "write chunk nt,     %d x %d", grid, threads);
            timed(name, static_cast<double>(out_bytes), [&] { write_chunk<true><<<grid, threads>>>(out4, out_bytes / 16); });
        }
    }

    // ---- flat copies ----
    auto copies = [&] (auto unroll_tag)
    {
        constexpr int U = decltype(unroll_tag)::value;

        for (int threads : { 256, 512, 1024 })
        {
            for (int grid : { 256, 512, 1024, 2048, 4096 })
            {
                if (quick && threads == 512) continue;

                snprintf(name, sizeof(name), "copy stride u%d plain/plain, %d x %d", U, grid, threads);
                timed(name, 2.0 * in_bytes, [&] { copy_stride<U, false, false><<<grid, threads>>>(in4, out4, count); });
                snprintf(name, sizeof(name), "copy stride u%d nt/plain,    %d x %d", U, grid, threads);
                timed(name, 2.0 * in_bytes, [&] { copy_stride<U, true, false><<<grid, threads>>>(in4, out4, count); });
                snprintf(name, sizeof(name), "copy stride u%d plain/nt,    %d x %d", U, grid, threads);
                timed(name, 2.0 * in_bytes, [&] { copy_stride<U, false, true><<<grid, threads>>>(in4, out4, count); });
                snprintf(name, sizeof(name), "copy stride u%d nt/nt,       %d x %d", U, grid, threads);
                timed(name, 2.0 * in_bytes, [&] { copy_stride<U, true, true><<<grid, threads>>>(in4, out4, count); });
                snprintf(name, sizeof(name), "copy chunk  u%d plain/plain, %d x %d", U, grid, threads);
                timed(name, 2.0 * in_bytes, [&] { copy_chunk<U, false, false><<<grid, threads>>>(in4, out4, count); });
                snprintf(name, sizeof(name), "copy chunk  u%d nt/nt,       %d x %d", U, grid, threads);
                timed(name, 2.0 * in_bytes, [&] { copy_chunk<U, true, true><<<grid, threads>>>(in4, out4, count); });
            }
        }
    };

    if (!quick)
    {
        copies(std::integral_constant<int, 1>{});
        copies(std::integral_constant<int, 4>{});
        copies(std::integral_constant<int, 8>{});
    }

    // ---- the resampler's rows: 2048 rows of 441088 frames in, 480000 out: 187 tiles ----
    {
        int const rows = 2048, tiles = 187;
        long long const in_row = 441088, out_row = 480000;
        double const bytes = 4.0 * rows * tiles * (2352.0 + 2560.0);

        for (int workgroups : { 2048, 4096, 8192 })
        {
            int const stride = workgroups / rows;

            snprintf(name, sizeof(name), "rows d1 plain/plain, %d workgroups x 1024", workgroups);
            timed(name, bytes, [&] { copy_rows<1, false, false><<<workgroups, 1024>>>(in, out, rows, tiles, stride, in_row, out_row); });
            snprintf(name, sizeof(name), "rows d1 nt/nt,       %d workgroups x 1024", workgroups);
            timed(name, bytes, [&] { copy_rows<1, true, true><<<workgroups, 1024>>>(in, out, rows, tiles, stride, in_row, out_row); });
            snprintf(name, sizeof(name), "rows d1 plain/nt,    %d workgroups x 1024", workgroups);
            timed(name, bytes, [&] { copy_rows<1, false, true><<<workgroups, 1024>>>(in, out, rows, tiles, stride, in_row, out_row); });
            snprintf(name, sizeof(name), "rows d1 nt/plain,    %d workgroups x 1024", workgroups);
            timed(name, bytes, [&] { copy_rows<1, true, false><<<workgroups, 1024>>>(in, out, rows, tiles, stride, in_row, out_row); });
            snprintf(name, sizeof(name), "rows d4 plain/plain, %d workgroups x 1024", workgroups);
            timed(name, bytes, [&] { copy_rows<4, false, false><<<workgroups, 1024>>>(in, out, rows, tiles, stride, in_row, out_row); });
            snprintf(name, sizeof(name), "rows d4 nt/nt,       %d workgroups x 1024", workgroups);
            timed(name, bytes, [&] { copy_rows<4, true, true><<<workgroups, 1024>>>(in, out, rows, tiles, stride, in_row, out_row); });
            snprintf(name, sizeof(name), "rows d8 nt/nt,       %d workgroups x 1024", workgroups);
            timed(name, bytes, [&] { copy_rows<8, true, true><<<workgroups, 1024>>>(in, out, rows, tiles, stride, in_row, out_row); });
        }

        // one workgroup per CU and row at a time (the resampler's launch: 1024 threads, one per CU)
        for (int workgroups : { 2048, 4096 })
        {
            int const stride = workgroups / rows;

            snprintf(name, sizeof(name), "rows teams 2 loaders plain/plain, %d workgroups", workgroups);
            timed(name, bytes, [&] { copy_rows_teams<2, false, false><<<workgroups, 1024>>>(in, out, rows, tiles, stride, in_row, out_row); });
            snprintf(name, sizeof(name), "rows teams 3 loaders plain/plain, %d workgroups", workgroups);
            timed(name, bytes, [&] { copy_rows_teams<3, false, false><<<workgroups, 1024>>>(in, out, rows, tiles, stride, in_row, out_row); });
            snprintf(name, sizeof(name), "rows teams 6 loaders plain/plain, %d workgroups", workgroups);
            timed(name, bytes, [&] { copy_rows_teams<6, false, false><<<workgroups, 1024>>>(in, out, rows, tiles, stride, in_row, out_row); });
            snprintf(name, sizeof(name), "rows teams 3 loaders nt/nt,       %d workgroups", workgroups);
            timed(name, bytes, [&] { copy_rows_teams<3, true, true><<<workgroups, 1024>>>(in, out, rows, tiles, stride, in_row, out_row); });
            snprintf(name, sizeof(name), "rows teams 6 loaders nt/nt,       %d workgroups", workgroups);
            timed(name, bytes, [&] { copy_rows_teams<6, true, true><<<workgroups, 1024>>>(in, out, rows, tiles, stride, in_row, out_row); });
            snprintf(name, sizeof(name), "rows teams 6 loaders plain/nt,    %d workgroups", workgroups);
            timed(name, bytes, [&] { copy_rows_teams<6, false, true><<<workgroups, 1024>>>(in, out, rows, tiles, stride, in_row, out_row); });
            snprintf(name, sizeof(name), "rows teams 6 loaders nt/plain,    %d workgroups", workgroups);
            timed(name, bytes, [&] { copy_rows_teams<6, true, false><<<workgroups, 1024>>>(in, out, rows, tiles, stride, in_row, out_row); });
        }

        // the chip sweeping memory front to back, one workgroup per CU and more
        for (int workgroups : { 256, 512, 1024 })
        {
            for (int band : { 187, 128, 64 })
            {
                snprintf(name, sizeof(name), "sweep teams 3 loaders plain/plain, %d workgroups, band %d", workgroups, band);
                timed(name, bytes, [&] { copy_sweep_teams<3, false, false><<<workgroups, 1024>>>(in, out, rows, tiles, band, in_row, out_row); });
                snprintf(name, sizeof(name), "sweep teams 6 loaders plain/plain, %d workgroups, band %d", workgroups, band);
                timed(name, bytes, [&] { copy_sweep_teams<6, false, false><<<workgroups, 1024>>>(in, out, rows, tiles, band, in_row, out_row); });
                snprintf(name, sizeof(name), "sweep teams 3 loaders nt/nt,       %d workgroups, band %d", workgroups, band);
                timed(name, bytes, [&] { copy_sweep_teams<3, true, true><<<workgroups, 1024>>>(in, out, rows, tiles, band, in_row, out_row); });
                snprintf(name, sizeof(name), "sweep teams 6 loaders nt/nt,       %d workgroups, band %d", workgroups, band);
                timed(name, bytes, [&] { copy_sweep_teams<6, true, true><<<workgroups, 1024>>>(in, out, rows, tiles, band, in_row, out_row); });
            }
        }
    }

    return 0;
}
