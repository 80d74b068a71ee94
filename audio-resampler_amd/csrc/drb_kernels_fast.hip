// Matrix-core ("fast") cascade for gfx950: [half-band up2x] -> windowed-sinc
// fractional stage -> [half-band down2x], every filter run as a banded
// (Toeplitz) matrix product on v_mfma_f32_16x16x4_f32.
//
// Layout.  A workgroup owns one channel of one stream and walks over tiles of
// 16 segments.  Every buffer in LDS is stored "transposed": element (segment
// c, row t) sits at float index (t + lookback) * 16 + c, so the 16 segments
// are the 16 columns of the MFMA and a block of 16 consecutive rows times 16
// segments is one 16 x 16 product tile:
//
//     D[row m][segment c] = sum_k A[m][k] * B[k][c],   B[k][c] = in(c, t0 + k)
//
// with A a banded matrix of filter coefficients (constant for the half-bands;
// for the fractional stage built from the exact int64 phase, where row m of a
// block uses its own blended bank row, shifted by its own window position).
// Segments are whole periods of the rate ratio apart, so all 16 share A.  The
// first `lookback` rows of a segment repeat the tail of the previous segment;
// segment 0 has no previous segment, so it only warms the filters up and the
// tile's output comes from segments 1..15.
//
// The fixed-point phase is not exactly periodic (it drifts by ~1e-10 source
// frames per output), so the fractional stage's coefficients are rebuilt from
// the exact phase every kRebuild tiles and in between follow the drift to
// first order: a + (blend now - blend at build) * da/dblend, exact for the
// piecewise-linear bank interpolation except at its breakpoints.
//
// Arithmetic is fused multiply-add in an order of the hardware's choosing and
// the fractional stage applies ONE blended kernel per output instead of
// blending two (or four) dot products; results agree with the upstream scalar
// code to a few 1e-7 for inputs in [-1, 1] (bound 1e-5 relative / 1e-6
// absolute, checked by the tests).  Upstream formulas:
// drb-audio-converter.c:1984-2004 (up), 2033-2057 (down), 1617-1648 /
// 1677-1707 + 881-941 (fractional stage).

#include <hip/hip_runtime.h>

#include "drb_hip.h"

namespace
{

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kBankRowStride = 65;
constexpr int kMaxTypes = 3;         // row-block types a wave keeps sinc coefficients for
constexpr int kRebuild = 16;         // tiles between exact rebuilds of those coefficients   // floats between taps in the LDS copy of the sinc bank

enum { kSampleF32 = 0, kSampleI16 = 1, kSampleI32 = 2 };
enum { kPlanar = 0, kInterleaved = 1 };

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float4_unaligned __attribute__((ext_vector_type(4), aligned(4)));

// Sample buffers arrive as integers inside the step descriptors; telling the
// compiler that they are global memory keeps their accesses off the (slower)
// flat path and lets four consecutive floats travel as one dwordx4.
#define DRB_GLOBAL __attribute__((address_space(1)))

template <typename T>
__device__ __forceinline__ DRB_GLOBAL T * global_pointer (unsigned long long address)
{
    return reinterpret_cast<DRB_GLOBAL T *>(address);
}

__constant__ unsigned int drb_fast_table_words [] =
{
#include "../../data/drb_tables_words.inc"
};

__device__ __forceinline__ float const * table_at (int float_offset)
{
    return reinterpret_cast<float const *>(drb_fast_table_words) + float_offset;
}

// Float index of (row, column) in a transposed buffer: consecutive rows
// alternate between the two halves of the 32 banks.
__device__ __forceinline__ int cell (int row, int column)
{
    return (row << 4) + column;
}

// The even plane of d is read four rows apart per lane group (the
// down-sampler's centre tap): flipping the half on bit 2 of the row keeps
// those reads to two per bank.
__device__ __forceinline__ int cell_centre (int row, int column)
{
    return ((row << 4) + column) ^ (((row >> 2) & 1) << 4);
}

__device__ __forceinline__ float load_sample (unsigned long long base, long long index, int type)
{
    switch (type)
    {
      case kSampleI16:
        return static_cast<float>(global_pointer<short const>(base)[index]) * (1.0f / 32768.0f);
      case kSampleI32:
        return static_cast<float>(global_pointer<int const>(base)[index]) * (1.0f / 2147483648.0f);
      default:
        return global_pointer<float const>(base)[index];
    }
}

__device__ __forceinline__ void store_sample (unsigned long long base, long long index, int type, float value)
{
    switch (type)
    {
      case kSampleI16:
      {
        float const scaled = rintf(value * 32768.0f);
        int const q = scaled >= 32767.0f ? 32767
                    : scaled <= -32768.0f ? -32768
                    : scaled != scaled ? 0 : static_cast<int>(scaled);
        global_pointer<short>(base)[index] = static_cast<short>(q);
        break;
      }
      case kSampleI32:
      {
        float const scaled = rintf(value * 2147483648.0f);
        int const q = scaled >= 2147483648.0f ? 2147483647
                    : scaled <= -2147483648.0f ? (-2147483647 - 1)
                    : scaled != scaled ? 0 : static_cast<int>(scaled);
        global_pointer<int>(base)[index] = q;
        break;
      }
      default:
        global_pointer<float>(base)[index] = value;
    }
}

// Source frame of one channel; negative frames come from the stream's history,
// anything older is silence.  PLAIN = planar float32 buffers (known at compile
// time so that the common case carries no sample-type or layout code).
template <bool PLAIN>
__device__ __forceinline__ float source_frame
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call,
        float const * history, long long channel_stride, int channel, int frame
    )
{
    if (frame < 0)
    {
        return frame >= -chain.history ? history[channel * chain.history + chain.history + frame] : 0.0f;
    }

    if (frame >= call.source_frames) return 0.0f;

    if (PLAIN) return global_pointer<float const>(call.source)[channel * channel_stride + frame];

    long long const index = chain.layout == kInterleaved
        ? static_cast<long long>(frame) * chain.channels + channel
        : channel * channel_stride + frame;

    return load_sample(call.source, index, chain.sample_type);
}

template <bool PLAIN>
__device__ __forceinline__ void target_frame
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call,
        long long channel_stride, int channel, int frame, float value
    )
{
    if (PLAIN)
    {
        global_pointer<float>(call.target)[channel * channel_stride + frame] = value;
        return;
    }

    long long const index = chain.layout == kInterleaved
        ? static_cast<long long>(frame) * chain.channels + channel
        : channel * channel_stride + frame;

    store_sample(call.target, index, chain.sample_type, value);
}

// Source frames consumed before the output with accumulator value `phase`, and
// its sub-sample position (drb-audio-converter.c:1617-1648 push, 1677-1707 pull).
__device__ __forceinline__ void sinc_position
    (
        long long phase, long long sd, double scale, int pull, int & consumed, double & blend
    )
{
    long long q = static_cast<long long>(floor(static_cast<double>(phase) * scale));
    long long r = phase - q * sd;

    while (r < 0) { q -= 1; r += sd; }
    while (r >= sd) { q += 1; r -= sd; }

    if (pull)
    {
        blend = static_cast<double>(r) * scale;
    }
    else
    {
        if (r == 0) { q -= 1; r = sd; }
        blend = static_cast<double>(r - 1) * scale;
    }

    consumed = static_cast<int>(q);
}

// ---- banded products -----------------------------------------------------------

// The B operands of one row block: STEPS floats `STRIDE` floats apart.
template <int STEPS, int STRIDE>
__device__ __forceinline__ void load_rows (float const * first, float (& b) [STEPS])
{
    #pragma unroll
    for (int s = 0; s < STEPS; s++) b[s] = first[STRIDE * s];
}

template <int STEPS>
__device__ __forceinline__ float4_t multiply (float const (& a) [STEPS], float const (& b) [STEPS])
{
    float4_t acc = { 0.0f, 0.0f, 0.0f, 0.0f };

    #pragma unroll
    for (int s = 0; s < STEPS; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);

    return acc;
}

// Two independent row blocks with the same coefficients, interleaved so that
// neither accumulator waits on the other.
template <int STEPS>
__device__ __forceinline__ void multiply_pair
    (
        float const (& a) [STEPS], float const (& b0) [STEPS], float const (& b1) [STEPS],
        float4_t & acc0, float4_t & acc1
    )
{
    acc0 = float4_t{ 0.0f, 0.0f, 0.0f, 0.0f };
    acc1 = float4_t{ 0.0f, 0.0f, 0.0f, 0.0f };

    #pragma unroll
    for (int s = 0; s < STEPS; s++)
    {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b0[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1[s], acc1, 0, 0, 0);
    }
}

struct Tile
{
    int base_y;   // target index of (segment 0, row 0)
    int base_s;   // sinc-output index of (segment 0, row 0)
    int base_u;   // sinc-input index of (segment 0, row 0)
    int base_x;   // source index of (segment 0, row 0)
};

struct Buffers
{
    float * x, * u, * d_odd, * d_even, * bank;
};

// ---- up-sampler: odd sinc-input samples ------------------------------------------
// u[2 f + 1] = 2 sum_j h[j] x[f - N + 1 + j], stored by source row f; the even
// ones, u[2 f] = x[f - N / 2], are read straight from the source plane.
__device__ __forceinline__ void write_up (Drb_Fast_Chain const & chain, Buffers const & buf, int block, float4_t acc)
{
    int const lane = threadIdx.x & 63, column = lane & 15, quad = lane >> 4;

    #pragma unroll
    for (int r = 0; r < 4; r++)
    {
        int const t = 16 * block + 4 * quad + r;

        if (t < chain.seg_x)
        {
            buf.u[cell(t + chain.look_u, column)] = acc[r];

            // tail of a segment = lookback of the next one
            if (t >= chain.seg_x - chain.look_u && column < 15)
            {
                buf.u[cell(t - chain.seg_x + chain.look_u, column + 1)] = acc[r];
            }
        }
    }
}

template <int STEPS>
__device__ __forceinline__ void stage_up (Drb_Fast_Chain const & chain, Buffers const & buf, float const (& a) [STEPS])
{
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int const blocks = (chain.seg_x + 15) / 16;

    // row of x[16 block - N] for block 0, this lane's quarter of the 4-tap step
    float const * const origin = buf.x + cell(chain.look_x - chain.up_taps + (lane >> 4), lane & 15);

    int block = wave;

    for (; block + kWaves < blocks; block += 2 * kWaves)
    {
        float b0 [STEPS], b1 [STEPS];
        float4_t acc0, acc1;

        load_rows<STEPS, 64>(origin + 256 * block, b0);
        load_rows<STEPS, 64>(origin + 256 * (block + kWaves), b1);
        multiply_pair<STEPS>(a, b0, b1, acc0, acc1);
        write_up(chain, buf, block, acc0);
        write_up(chain, buf, block + kWaves, acc1);
    }

    if (block < blocks)
    {
        float b0 [STEPS];

        load_rows<STEPS, 64>(origin + 256 * block, b0);
        write_up(chain, buf, block, multiply<STEPS>(a, b0));
    }
}

// ---- fractional stage --------------------------------------------------------------

// What a wave keeps per row-block type between exact rebuilds.
template <int STEPS>
struct Sinc_Type
{
    float a [STEPS];   // coefficients at the build phase: row (lane & 15) of the block, taps 4 s + quad - skew
    float g [STEPS];   // their derivative with respect to the sub-sample position
    int first_row;     // sinc-input row of the block's first tap
};

// Reference point of the drift correction: the middle segment's first row.
struct Sinc_Anchor
{
    long long rest;    // phase of that row minus the source frames it has consumed
    int base_s, base_u, base_x;
};

template <int STEPS>
__device__ __forceinline__ void build_sinc_type
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call, Sinc_Anchor const & anchor,
        float const * bank, int block, Sinc_Type<STEPS> & type, int * fault_flag
    )
{
    int const lane = threadIdx.x & 63, column = lane & 15, quad = lane >> 4;
    int const taps = chain.sinc_taps;

    // exact position of this lane's row in the middle segment (8)
    int consumed;
    double blend;

    sinc_position(call.phase + static_cast<long long>(anchor.base_s + 8 * chain.seg_s + 16 * block + column) * chain.target_delta,
                  chain.source_delta, chain.scale, chain.sinc_pull, consumed, blend);

    int const end = consumed - (anchor.base_u + 8 * chain.seg_u);   // window = rows [end - taps, end)
    int const first_row = __builtin_amdgcn_readfirstlane(end) - taps;
    int const skew = end - taps - first_row;

    if (column == 15 && skew + taps > 4 * STEPS) atomicOr(fault_flag, 8);

    float const rows_per_unit = chain.sinc_cubic ? 29.0f : 63.0f;
    double const scaled = blend * static_cast<double>(rows_per_unit);
    int const bank_row = static_cast<int>(scaled);
    float const b = static_cast<float>(scaled - static_cast<double>(bank_row));
    float const * const rows = bank + bank_row;

    // weights of the bank rows and their derivatives with respect to b
    float w0, w1, w2 = 0.0f, w3 = 0.0f, v0, v1, v2 = 0.0f, v3 = 0.0f;

    if (chain.sinc_cubic)
    {
        // Catmull-Rom (drb-audio-converter.c:861-874)
        float const b2 = b * b, b3 = b2 * b;

        w0 = -0.5f * b + b2 - 0.5f * b3;
        w1 = 1.0f - 2.5f * b2 + 1.5f * b3;
        w2 = 0.5f * b + 2.0f * b2 - 1.5f * b3;
        w3 = -0.5f * b2 + 0.5f * b3;
        v0 = -0.5f + 2.0f * b - 1.5f * b2;
        v1 = -5.0f * b + 4.5f * b2;
        v2 = 0.5f + 4.0f * b - 4.5f * b2;
        v3 = -b + 1.5f * b2;
    }
    else
    {
        w0 = 1.0f - b;
        w1 = b;
        v0 = -1.0f;
        v1 = 1.0f;
    }

    #pragma unroll
    for (int s = 0; s < STEPS; s++)
    {
        int const t = 4 * s + quad - skew;
        bool const live = t >= 0 && t < taps;
        float const * const k = rows + (live ? t : 0) * kBankRowStride;
        float value = w0 * k[0] + w1 * k[1];
        float slope = v0 * k[0] + v1 * k[1];

        if (chain.sinc_cubic)
        {
            value += w2 * k[2] + w3 * k[3];
            slope += v2 * k[2] + v3 * k[3];
        }

        type.a[s] = live ? value : 0.0f;
        type.g[s] = live ? slope * rows_per_unit : 0.0f;
    }

    type.first_row = first_row;
}

template <bool DOWN, bool PLAIN>
__device__ __forceinline__ void write_sinc
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call, Tile const & tile, Buffers const & buf,
        int channel, long long target_channel_stride, int block, int period, float4_t acc
    )
{
    int const lane = threadIdx.x & 63, column = lane & 15, quad = lane >> 4;
    int const in_period = 16 * block + 4 * quad;
    int const t0 = period * chain.period_out + in_period;   // row of the sinc output = row of d

    if (DOWN)
    {
        // rows alternate between the planes of d; a period may start on either
        int const half = chain.seg_s >> 1;

        #pragma unroll
        for (int r = 0; r < 4; r++)
        {
            int const t = t0 + r, g = t >> 1;

            if (in_period + r >= chain.period_out) continue;

            if (t & 1)
            {
                buf.d_odd[cell(g + chain.look_d, column)] = acc[r];

                if (g >= half - chain.look_d && column < 15) buf.d_odd[cell(g - half + chain.look_d, column + 1)] = acc[r];
            }
            else
            {
                buf.d_even[cell_centre(g + chain.look_d, column)] = acc[r];

                if (g >= half - chain.look_d && column < 15) buf.d_even[cell_centre(g - half + chain.look_d, column + 1)] = acc[r];
            }
        }
    }
    else
    {
        #pragma unroll
        for (int r = 0; r < 4; r++)
        {
            int const frame = tile.base_y + column * chain.seg_y + t0 + r;

            if (in_period + r < chain.period_out && column >= 1 && frame < call.target_frames)
            {
                target_frame<PLAIN>(chain, call, target_channel_stride, channel, frame, acc[r]);
            }
        }
    }
}

// First B operand of a sinc row block for this lane: sinc-input row `row`
// (segment-relative, any parity) plus the lane's quarter; later steps are
// STRIDE floats further.  Behind an up-sampler even rows live in the source
// plane and odd rows in the up-sampler's output plane, two rows per step.
template <bool PLANES>
__device__ __forceinline__ float const * sinc_operand (Drb_Fast_Chain const & chain, Buffers const & buf, int row)
{
    int const lane = threadIdx.x & 63, column = lane & 15, quad = lane >> 4;
    int const u = row + quad;

    if (!PLANES) return buf.u + cell(u + chain.look_u, column);

    return (u & 1) ? buf.u + cell(((u - 1) >> 1) + chain.look_u, column)
                   : buf.x + cell(((u - chain.up_taps) >> 1) + chain.look_x, column);
}

template <int STEPS, bool PLANES, bool DOWN, bool PLAIN>
__device__ __forceinline__ void stage_sinc
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call, Tile const & tile, Buffers const & buf,
        int channel, long long target_channel_stride,
        Sinc_Type<STEPS> const (& types) [kMaxTypes], float drift
    )
{
    constexpr int kStride = PLANES ? 32 : 64;

    int const wave = threadIdx.x >> 6;
    int const chains = chain.blocks * chain.periods;
    int const share = (chains + kWaves - 1) / kWaves;
    int const first = wave * share, last = min(first + share, chains);
    int const first_block = first / chain.periods;

    // One copy of the code for all of the wave's row blocks: the block's
    // coefficient set is picked out of the (statically indexed) registers.
    for (int q = first; q < last; )
    {
        int const block = q / chain.periods, period = q - block * chain.periods;
        int const slot = block - first_block;
        bool const pair = period + 1 < chain.periods && q + 1 < last;

        float a [STEPS];
        int first_row = period * chain.period_in;

        // coefficients at this tile's phase; the slot is uniform over the wave
        #pragma unroll
        for (int i = 0; i < kMaxTypes; i++)
        {
            if (__builtin_amdgcn_readfirstlane(slot) == i)
            {
                #pragma unroll
                for (int s = 0; s < STEPS; s++) a[s] = fmaf(drift, types[i].g[s], types[i].a[s]);

                first_row += types[i].first_row;
            }
        }

        float b0 [STEPS], b1 [STEPS];
        float4_t acc0, acc1;

        load_rows<STEPS, kStride>(sinc_operand<PLANES>(chain, buf, first_row), b0);

        if (pair)
        {
            load_rows<STEPS, kStride>(sinc_operand<PLANES>(chain, buf, first_row + chain.period_in), b1);
            multiply_pair<STEPS>(a, b0, b1, acc0, acc1);
            write_sinc<DOWN, PLAIN>(chain, call, tile, buf, channel, target_channel_stride, block, period, acc0);
            write_sinc<DOWN, PLAIN>(chain, call, tile, buf, channel, target_channel_stride, block, period + 1, acc1);
            q += 2;
        }
        else
        {
            write_sinc<DOWN, PLAIN>(chain, call, tile, buf, channel, target_channel_stride, block, period, multiply<STEPS>(a, b0));
            q += 1;
        }
    }
}

// ---- down-sampler --------------------------------------------------------------------
// y[f] = 0.5 d[2 f - N] + sum_j h[j] d[2 (f - N + j) + 1]
template <bool PLAIN>
__device__ __forceinline__ void write_down
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call, Tile const & tile, Buffers const & buf,
        int channel, long long target_channel_stride, int block, float4_t acc
    )
{
    int const lane = threadIdx.x & 63, column = lane & 15, quad = lane >> 4;
    int const t0 = 16 * block + 4 * quad;
    int const frame0 = tile.base_y + column * chain.seg_y + t0;
    float value [4];

    #pragma unroll
    for (int r = 0; r < 4; r++)
    {
        float const centre = buf.d_even[cell_centre(t0 + r - chain.down_taps / 2 + chain.look_d, column)];
        value[r] = fmaf(0.5f, centre, acc[r]);
    }

    if (column < 1) return;

    if (PLAIN && t0 + 3 < chain.seg_y && frame0 + 3 < call.target_frames)
    {
        DRB_GLOBAL float * const out = global_pointer<float>(call.target) + channel * target_channel_stride + frame0;
        *reinterpret_cast<DRB_GLOBAL float4_unaligned *>(out) = float4_unaligned{ value[0], value[1], value[2], value[3] };
        return;
    }

    #pragma unroll
    for (int r = 0; r < 4; r++)
    {
        if (t0 + r < chain.seg_y && frame0 + r < call.target_frames)
        {
            target_frame<PLAIN>(chain, call, target_channel_stride, channel, frame0 + r, value[r]);
        }
    }
}

template <int STEPS, bool PLAIN>
__device__ __forceinline__ void stage_down
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call, Tile const & tile, Buffers const & buf,
        int channel, long long target_channel_stride, float const (& a) [STEPS]
    )
{
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int const blocks = (chain.seg_y + 15) / 16;
    float const * const origin = buf.d_odd + cell(chain.look_d - chain.down_taps + (lane >> 4), lane & 15);

    int block = wave;

    for (; block + kWaves < blocks; block += 2 * kWaves)
    {
        float b0 [STEPS], b1 [STEPS];
        float4_t acc0, acc1;

        load_rows<STEPS, 64>(origin + 256 * block, b0);
        load_rows<STEPS, 64>(origin + 256 * (block + kWaves), b1);
        multiply_pair<STEPS>(a, b0, b1, acc0, acc1);
        write_down<PLAIN>(chain, call, tile, buf, channel, target_channel_stride, block, acc0);
        write_down<PLAIN>(chain, call, tile, buf, channel, target_channel_stride, block + kWaves, acc1);
    }

    if (block < blocks)
    {
        float b0 [STEPS];

        load_rows<STEPS, 64>(origin + 256 * block, b0);
        write_down<PLAIN>(chain, call, tile, buf, channel, target_channel_stride, block, multiply<STEPS>(a, b0));
    }
}

// ---- source tile: global -> registers -> LDS -------------------------------------------

constexpr int kFetchMax = 4;   // float4 per thread: up to 4 * 64 = 256 rows of 16 segments

struct Fetch
{
    float4_t v [kFetchMax];
};

// Thread t holds rows 4 (t >> 4) + 64 i + {0..3} of segment t & 15.
template <bool PLAIN>
__device__ __forceinline__ void fetch_tile
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call, int base_x,
        float const * history, long long source_channel_stride, int channel, Fetch & fetch
    )
{
    int const segment = threadIdx.x & 15, group = threadIdx.x >> 4;
    int const rows = chain.seg_x + chain.look_x;
    int const first_frame = base_x + segment * chain.seg_x + 4 * group - chain.look_x;

    // Interior tiles of planar float streams (the common case, decided for the
    // whole workgroup): every row is a plain dwordx4 load and nothing waits.
    bool const interior = PLAIN
        && base_x - chain.look_x >= 0
        && base_x + 15 * chain.seg_x + ((rows + 3) & ~3) + 64 <= call.source_frames;

    if (interior)
    {
        DRB_GLOBAL float const * const planar =
            global_pointer<float const>(call.source) + channel * source_channel_stride + first_frame;

        #pragma unroll
        for (int i = 0; i < kFetchMax; i++)
        {
            if (64 * i < rows)
            {
                fetch.v[i] = *reinterpret_cast<DRB_GLOBAL float4_unaligned const *>(planar + 64 * i);
            }
        }

        return;
    }

    // stream edges, history, other sample types and layouts: one by one
    #pragma unroll
    for (int i = 0; i < kFetchMax; i++)
    {
        if (64 * i < rows)
        {
            #pragma unroll
            for (int e = 0; e < 4; e++)
            {
                fetch.v[i][e] = source_frame<PLAIN>(chain, call, history, source_channel_stride, channel,
                                             first_frame + 64 * i + e);
            }
        }
    }
}

__device__ __forceinline__ void store_tile (Drb_Fast_Chain const & chain, Buffers const & buf, Fetch const & fetch)
{
    int const segment = threadIdx.x & 15, group = threadIdx.x >> 4;
    int const rows = chain.seg_x + chain.look_x;

    #pragma unroll
    for (int i = 0; i < kFetchMax; i++)
    {
        #pragma unroll
        for (int e = 0; e < 4; e++)
        {
            int const row = 4 * group + 64 * i + e;

            if (row < rows) buf.x[cell(row, segment)] = fetch.v[i][e];
        }
    }
}

// Diagnostic build only (-DDRB_FAST_STAMPS): where a wave's cycles go, summed
// per phase into fault_flag[1 + 16 wave_class + phase] (the flag word itself is
// word 0).  Never compiled into the library that ships.
#ifdef DRB_FAST_STAMPS
#  define STAMP(phase) do { long long const now_ = __builtin_readcyclecounter(); spent[phase] += now_ - mark; mark = now_; } while (0)
#else
#  define STAMP(phase) do { } while (0)
#endif

template <int UP_STEPS, int SINC_STEPS, int DOWN_STEPS, bool PLAIN>
__global__ __launch_bounds__(kThreads, 2) void drb_fast_kernel
    (
        Drb_Fast_Chain const chain,
        Drb_Device_Call const * __restrict__ calls, int stream_count, int tile_stride,
        float const * __restrict__ history,
        long long source_channel_stride, long long target_channel_stride,
        int * fault_flag
    )
{
    extern __shared__ __attribute__((aligned(16))) float lds [];

    int const pair = blockIdx.x % (stream_count * chain.channels);
    int const first_tile = blockIdx.x / (stream_count * chain.channels);
    int const stream = pair / chain.channels, channel = pair - stream * chain.channels;
    Drb_Device_Call const call = calls[stream];

    if (first_tile * chain.tile_frames >= call.target_frames) return;

    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6, column = lane & 15, quad = lane >> 4;

    Buffers const buf = { lds + chain.off_x, lds + chain.off_u, lds + chain.off_d_odd,
                          lds + chain.off_d_even, lds + chain.off_bank };
    float const * const stream_history = history + static_cast<long long>(stream) * chain.channels * chain.history;

    // ---- once per workgroup -----------------------------------------------------
    // coefficient bank of the fractional stage -> LDS as [tap][row]
    {
        int const taps = chain.sinc_taps, rows = chain.sinc_cubic ? 32 : 64;
        float const * const source = table_at(chain.sinc_table);

        for (int i = threadIdx.x; i < rows * taps; i += kThreads)
        {
            int const row = i / taps, tap = i - row * taps;
            buf.bank[tap * kBankRowStride + row] = source[i];
        }
    }

    // half-band coefficients as banded A fragments, parked in LDS as [step][lane]
    // (the same for every wave) and fetched by the stage that uses them:
    //   up:   A[m][k] = 2 h[k - m - 1]   (block window starts at row 16 i - N)
    //   down: A[m][k] =   h[k - m]
    float * const park_up = buf.bank + chain.sinc_taps * kBankRowStride + 8;
    float * const park_down = park_up + 64 * UP_STEPS;

    if (wave == 0)
    {
        #pragma unroll
        for (int s = 0; s < UP_STEPS; s++)
        {
            int const j = 4 * s + quad - column - 1;
            park_up[64 * s + lane] = j >= 0 && j < chain.up_taps ? 2.0f * table_at(chain.up_table)[j] : 0.0f;
        }

        #pragma unroll
        for (int s = 0; s < DOWN_STEPS; s++)
        {
            int const j = 4 * s + quad - column;
            park_down[64 * s + lane] = j >= 0 && j < chain.down_taps ? table_at(chain.down_table)[j] : 0.0f;
        }
    }

    int const delay_behind = chain.delay_place == 1 ? call.delay : 0;
    int const delay_front = chain.delay_place == 2 ? call.delay : 0;
    int const tile_advance_s = (DOWN_STEPS ? 2 : 1) * chain.tile_frames;   // sinc outputs per tile

#ifdef DRB_FAST_STAMPS
    long long spent [10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    long long mark = __builtin_readcyclecounter();
#endif

    Sinc_Type<SINC_STEPS> types [kMaxTypes];
    Sinc_Anchor anchor = { 0, 0, 0, 0 };
    int since_build = kRebuild;
    Fetch fetch;
    bool fetched = false;

    // ---- tiles of this stream and channel ------------------------------------------
    for (int tile_index = first_tile; tile_index * chain.tile_frames < call.target_frames; tile_index += tile_stride)
    {
        Tile tile;

        // segment 1 starts at the tile's first target frame
        tile.base_y = tile_index * chain.tile_frames - chain.seg_y;
        tile.base_s = (DOWN_STEPS ? 2 * tile.base_y : tile.base_y) - delay_behind;

        bool const rebuild = since_build >= kRebuild;
        float drift = 0.0f;

        if (rebuild)
        {
            // window ends of the middle segment (8) define the phases of all segments
            int consumed;
            double unused;
            long long const phase = call.phase + static_cast<long long>(tile.base_s + 8 * chain.seg_s) * chain.target_delta;

            sinc_position(phase, chain.source_delta, chain.scale, chain.sinc_pull, consumed, unused);

            // the first row's window ends at row -2 or -1 (so that the last row's
            // window still ends inside the segment); the producer-side index must
            // be even when an up-sampler feeds the stage
            tile.base_u = consumed - 8 * chain.seg_u + 2;

            if (UP_STEPS && ((tile.base_u - delay_front) & 1)) tile.base_u -= 1;

            tile.base_x = UP_STEPS ? (tile.base_u - delay_front) / 2 : tile.base_u - delay_front;

            anchor.rest = phase - static_cast<long long>(consumed) * chain.source_delta;
            anchor.base_s = tile.base_s;
            anchor.base_u = tile.base_u;
            anchor.base_x = tile.base_x;
            since_build = 0;
        }
        else
        {
            // between rebuilds every edge advances by whole periods, and the
            // phase's departure from that is the drift the coefficients follow
            int const tiles = (tile.base_s - anchor.base_s) / tile_advance_s;

            tile.base_u = anchor.base_u + tiles * 15 * chain.seg_u;
            tile.base_x = anchor.base_x + tiles * 15 * chain.seg_x;

            long long const advance = static_cast<long long>(tile.base_s - anchor.base_s) * chain.target_delta
                                    - static_cast<long long>(tile.base_u - anchor.base_u) * chain.source_delta;

            drift = static_cast<float>(static_cast<double>(advance) * chain.scale);
        }

        since_build += 1;

        if (!fetched && !(chain.ablate & 1))
        {
            fetch_tile<PLAIN>(chain, call, tile.base_x, stream_history, source_channel_stride, channel, fetch);
        }

        STAMP(0);

        if (!(chain.ablate & 1)) store_tile(chain, buf, fetch);

        STAMP(1);

        __syncthreads();   // source tile (and bank) in place

        STAMP(2);

        // the next tile's source travels while this one is filtered
        {
            int const next_index = tile_index + tile_stride;

            fetched = next_index * chain.tile_frames < call.target_frames && since_build < kRebuild;

            if (fetched && !(chain.ablate & 1))
            {
                fetch_tile<PLAIN>(chain, call, tile.base_x + tile_stride * 15 * chain.seg_x, stream_history,
                           source_channel_stride, channel, fetch);
            }
        }

        STAMP(3);

        if (rebuild && !(chain.ablate & 4))
        {
            int const chains = chain.blocks * chain.periods;
            int const share = (chains + kWaves - 1) / kWaves;
            int const first_block = wave * share / chain.periods;

            #pragma unroll
            for (int i = 0; i < kMaxTypes; i++)
            {
                if (first_block + i < chain.blocks && i < chain.types_per_wave)
                {
                    build_sinc_type<SINC_STEPS>(chain, call, anchor, buf.bank, first_block + i, types[i], fault_flag);
                }
            }
        }

        STAMP(4);

        if (UP_STEPS > 0)
        {
            if (!(chain.ablate & 2))
            {
                float a_up [UP_STEPS ? UP_STEPS : 1];

                load_rows<UP_STEPS ? UP_STEPS : 1, 64>(park_up + lane, a_up);
                stage_up<UP_STEPS ? UP_STEPS : 1>(chain, buf, a_up);
            }

            STAMP(5);

            __syncthreads();   // odd plane of the sinc input complete

            STAMP(6);
        }

        if (!(chain.ablate & 4))
        {
            stage_sinc<SINC_STEPS, (UP_STEPS > 0), (DOWN_STEPS > 0), PLAIN>(chain, call, tile, buf, channel, target_channel_stride, types, drift);
        }

        STAMP(7);

        __syncthreads();   // both planes of d complete; source plane free again

        STAMP(8);

        if (DOWN_STEPS > 0 && !(chain.ablate & 8))
        {
            float a_down [DOWN_STEPS ? DOWN_STEPS : 1];

            load_rows<DOWN_STEPS ? DOWN_STEPS : 1, 64>(park_down + lane, a_down);
            stage_down<DOWN_STEPS ? DOWN_STEPS : 1, PLAIN>(chain, call, tile, buf, channel, target_channel_stride, a_down);
        }

        STAMP(9);
    }

#ifdef DRB_FAST_STAMPS
    if (lane == 0)
    {
        for (int phase = 0; phase < 10; phase++)
        {
            atomicAdd(reinterpret_cast<unsigned long long *>(fault_flag) + 1 + 16 * wave + phase,
                      static_cast<unsigned long long>(spent[phase]));
        }
    }
#endif
}

typedef void Fast_Kernel
    (
        Drb_Fast_Chain, Drb_Device_Call const *, int, int, float const *, long long, long long, int *
    );

// The kernel images that exist: (up steps, sinc steps, down steps), steps =
// (taps + 16) / 4 for a half-band, chain.ksteps for the fractional stage.
Fast_Kernel * pick_kernel (Drb_Fast_Chain const * chain)
{
    int const up = chain->has_up ? (chain->up_taps + 16) / 4 : 0;
    int const down = chain->has_down ? (chain->down_taps + 16) / 4 : 0;
    int const key = up * 10000 + chain->ksteps * 100 + down;

    bool const plain = chain->sample_type == kSampleF32 && chain->layout == kPlanar;

#define DRB_FAST_CASE(KEY, UP, SINC, DOWN) \
      case KEY: return plain ? drb_fast_kernel<UP, SINC, DOWN, true> : drb_fast_kernel<UP, SINC, DOWN, false>;

    switch (key)
    {
      DRB_FAST_CASE(141214, 14, 12, 14)   // good, rate going up
      DRB_FAST_CASE(141314, 14, 13, 14)   // good, rate going down
      DRB_FAST_CASE(120812, 12, 8, 12)    // fine
      DRB_FAST_CASE(120912, 12, 9, 12)
      DRB_FAST_CASE(161216, 16, 12, 16)   // best
      DRB_FAST_CASE(161316, 16, 13, 16)
      DRB_FAST_CASE(800, 0, 8, 0)         // poor
      DRB_FAST_CASE(900, 0, 9, 0)
    }

#undef DRB_FAST_CASE

    return nullptr;
}

} // namespace

extern "C" int drb_hip_fast_kernel_exists (Drb_Fast_Chain const * chain)
{
    return pick_kernel(chain) != nullptr;
}

extern "C" int drb_hip_launch_fast
    (
        Drb_Fast_Chain const * chain,
        Drb_Device_Call const * calls, int stream_count, int max_tiles,
        float const * history,
        long long source_channel_stride, long long target_channel_stride,
        int * fault_flag, void * stream
    )
{
    if (stream_count <= 0 || max_tiles <= 0) return 0;

    size_t const dynamic = static_cast<size_t>(chain->lds_floats) * sizeof(float);

    Fast_Kernel * const kernel = pick_kernel(chain);

    if (!kernel) return hipErrorInvalidValue;

    if (dynamic > 48 * 1024)
    {
        // per kernel image; cheap enough to repeat, and correct on every device
        hipError_t const status = hipFuncSetAttribute
        (
            reinterpret_cast<void const *>(kernel),
            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(dynamic)
        );

        if (status != hipSuccess) return status;
    }

    // Workgroups walk over the tiles of their stream and channel with a stride;
    // about eight workgroups per CU are enough to keep the chip busy, more only
    // repeat the per-workgroup set-up.
    long long const pairs = static_cast<long long>(stream_count) * chain->channels;
    long long tile_stride = (2048 + pairs - 1) / pairs;

    if (tile_stride > max_tiles) tile_stride = max_tiles;
    if (tile_stride < 1) tile_stride = 1;

    long long const blocks = pairs * tile_stride;

    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;

    kernel<<<dim3(static_cast<unsigned>(blocks)), dim3(kThreads), dynamic, static_cast<hipStream_t>(stream)>>>
    (
        *chain, calls, stream_count, static_cast<int>(tile_stride), history,
        source_channel_stride, target_channel_stride, fault_flag
    );

    return hipGetLastError();
}
