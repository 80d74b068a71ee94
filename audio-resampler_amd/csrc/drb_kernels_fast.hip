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

#include <mutex>

#include "drb_hip.h"

namespace
{

constexpr int kMaxThreads = 320;     // four or five waves, whichever deals the row blocks out more evenly
constexpr int kBankTapStride = 65;    // floats between taps in the LDS copy of the sinc bank
constexpr int kMaxTypes = 3;         // row-block types a wave keeps sinc coefficients for
// Tiles of a stream between exact rebuilds of those coefficients.  Measured on 4 s
// of noise, 44.1 -> 48 kHz good, worst |error| against upstream: 4.8e-7 rebuilding
// every tile or every 8, 7.2e-7 every 16, 9.5e-7 every 32, 1.7e-6 (over the
// bound) every 64 (tools/drift_probe.py).  20 leaves a stream of 19 tiles (1 s at
// 48 kHz) with one rebuild, in its middle.  (44.1 -> 48 kHz moves 1.7e-8 source
// frames per segment, 176.4 -> 192 kHz 4.5e-8, the most among the 13 rates.)
constexpr int kRebuild = 20;

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

    long long const index = chain.target_layout == kInterleaved
        ? static_cast<long long>(frame) * chain.channels + channel
        : channel * channel_stride + frame;

    store_sample(call.target, index, chain.target_sample_type, value);
}

// Four consecutive target frames of one channel (not planar float32): where
// they lie together in the buffer they leave as one store.
__device__ __forceinline__ void target_four
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call,
        long long channel_stride, int channel, int frame, float const (& value) [4]
    )
{
    typedef int int4_unaligned __attribute__((ext_vector_type(4), aligned(4)));
    typedef int int2_unaligned __attribute__((ext_vector_type(2), aligned(2)));

    bool const contiguous = chain.target_layout == kPlanar || chain.channels == 1;
    long long const index = chain.target_layout == kInterleaved
        ? static_cast<long long>(frame) * chain.channels + channel
        : channel * channel_stride + frame;

    if (contiguous && chain.target_sample_type == kSampleI16)
    {
        int q [4];

        #pragma unroll
        for (int r = 0; r < 4; r++)
        {
            float const scaled = rintf(value[r] * 32768.0f);

            q[r] = scaled >= 32767.0f ? 32767 : scaled <= -32768.0f ? -32768 : scaled != scaled ? 0 : static_cast<int>(scaled);
        }

        *reinterpret_cast<DRB_GLOBAL int2_unaligned *>(global_pointer<short>(call.target) + index) =
            int2_unaligned{ (q[0] & 0xffff) | (q[1] << 16), (q[2] & 0xffff) | (q[3] << 16) };
        return;
    }

    if (contiguous && chain.target_sample_type == kSampleI32)
    {
        int q [4];

        #pragma unroll
        for (int r = 0; r < 4; r++)
        {
            float const scaled = rintf(value[r] * 2147483648.0f);

            q[r] = scaled >= 2147483648.0f ? 2147483647 : scaled <= -2147483648.0f ? (-2147483647 - 1)
                 : scaled != scaled ? 0 : static_cast<int>(scaled);
        }

        *reinterpret_cast<DRB_GLOBAL int4_unaligned *>(global_pointer<int>(call.target) + index) = int4_unaligned{ q[0], q[1], q[2], q[3] };
        return;
    }

    #pragma unroll
    for (int r = 0; r < 4; r++) target_frame<false>(chain, call, channel_stride, channel, frame + r, value[r]);
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

struct Tile
{
    int index;       // which of the workgroup's tiles (0 = its first)
    int segment;     // index of the tile's segment 0: target frames [segment seg_y, (segment + 1) seg_y)
    int first_column;// 1: segment 0 only warms the filters up; 0: its lookback came from the tile before
    int base_y;      // target index of (segment 0, row 0)
    int base_x;      // source index of (segment 0, row 0)
    float drift;     // sub-sample position now minus at the last exact build
    bool rebuild;    // sinc coefficients are rebuilt from the exact phase for this tile
    bool inside;     // all 16 segments end before the stream's last target frame
};

struct Buffers
{
    float * x, * u, * d_odd, * d_even, * bank;
};

// Float offsets of a lane inside a [row][16 segments] buffer.  Everything a
// stage adds to them is the same for the whole wave (scalar registers) or a
// constant of the instruction, so an access costs at most one vector add.
struct Lane
{
    int column, quad;
    int operand;        // (row quad, column): B operands, rows 4 steps apart
    int result;         // (row 4 quad, column): the lane's four result rows
    int planes;         // (row 2 quad, column): result rows dealt to two planes
    int sinc_even;      // B operand of the fractional stage behind an up-sampler,
    int sinc_odd;       //   relative to `lds`, for an even / odd first row
};

// ---- up-sampler: odd sinc-input samples ------------------------------------------
// u[2 f + 1] = 2 sum_j h[j] x[f - N + 1 + j], stored by source row f; the even
// ones, u[2 f] = x[f - N / 2], are read straight from the source plane.
__device__ __forceinline__ void write_up
    (
        Drb_Fast_Chain const & chain, Buffers const & buf, Lane const & lane, int block, float4_t acc
    )
{
    float * const rows = buf.u + 16 * (16 * block + chain.look_u) + lane.result;

    // blocks that end before the segment's tail: four plain stores
    if (16 * block + 16 <= chain.seg_x - chain.look_u)
    {
        rows[0] = acc[0];
        rows[16] = acc[1];
        rows[32] = acc[2];
        rows[48] = acc[3];
        return;
    }

    // the tail of a segment is also the lookback of the next one
    float * const next = rows - 16 * chain.seg_x + 1;

    if (16 * block >= chain.seg_x - chain.look_u && 16 * block + 16 <= chain.seg_x)
    {
        rows[0] = acc[0];
        rows[16] = acc[1];
        rows[32] = acc[2];
        rows[48] = acc[3];

        if (lane.column < 15)
        {
            next[0] = acc[0];
            next[16] = acc[1];
            next[32] = acc[2];
            next[48] = acc[3];
        }

        return;
    }

    #pragma unroll
    for (int r = 0; r < 4; r++)
    {
        int const t = 16 * block + 4 * lane.quad + r;

        if (t < chain.seg_x)
        {
            rows[16 * r] = acc[r];

            if (t >= chain.seg_x - chain.look_u && lane.column < 15) next[16 * r] = acc[r];
        }
    }
}

template <int STEPS>
__device__ __forceinline__ void stage_up
    (
        Drb_Fast_Chain const & chain, Buffers const & buf, Lane const & lane, int wave, float const (& a) [STEPS]
    )
{
    int const blocks = (chain.seg_x + 15) / 16;

    // row of x[16 block - N] for block 0, this lane's quarter of the 4-tap step
    float const * const origin = buf.x + 16 * (chain.look_x - chain.up_taps) + lane.operand;

    for (int block = wave; block < blocks; block += chain.waves)
    {
        float b [STEPS];

        load_rows<STEPS, 64>(origin + 256 * block, b);
        write_up(chain, buf, lane, block, multiply<STEPS>(a, b));
    }
}

// ---- fractional stage --------------------------------------------------------------

// What a wave keeps per row-block type between exact rebuilds.
template <int STEPS>
struct Sinc_Type
{
    float a [STEPS];                 // coefficients now: row (lane & 15) of the block, taps 4 s + quad - skew
    unsigned g [(STEPS + 1) / 2];    // their derivative with respect to the sub-sample position, as bfloat16
                                     //   pairs (it only scales a correction of a few 1e-6): even step low
    int first_row;                   // sinc-input row of the block's first tap
};

// Reference point of the drift correction: the middle segment's first row.
struct Sinc_Anchor
{
    long long rest;    // phase of that row minus the source frames it has consumed
    int segment;       // the anchor tile's segment 0
    int base_s, base_u, base_x;
};

// Where a lane's row of a block type sits at the anchor tile, and how its
// bank rows blend.
struct Sinc_Spot
{
    int first_row, skew;       // window = rows [first_row + skew, first_row + skew + taps)
    float const * rows;        // bank row of the blend's integer part in the LDS copy, [tap][row]
    float w [4], v [4];        // weights of bank rows 0 .. 3 from there, and their derivatives
};

template <int STEPS, bool CUBIC>
__device__ __forceinline__ void spot_sinc_type
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call, Sinc_Anchor const & anchor,
        float const * bank, int block, Sinc_Spot & spot, int * fault_flag
    )
{
    int const column = threadIdx.x & 15;
    int const taps = chain.sinc_taps;

    // exact position of this lane's row in the middle segment (8)
    int consumed;
    double blend;

    sinc_position(call.phase + static_cast<long long>(anchor.base_s + 8 * chain.seg_s + 16 * block + column) * chain.target_delta,
                  chain.source_delta, chain.scale, chain.sinc_pull, consumed, blend);

    int const end = consumed - (anchor.base_u + 8 * chain.seg_u);   // window = rows [end - taps, end)

    spot.first_row = __builtin_amdgcn_readfirstlane(end) - taps;
    spot.skew = end - taps - spot.first_row;

    if (column == 15 && spot.skew + taps > 4 * STEPS) atomicOr(fault_flag, 8);

    double const scaled = blend * (CUBIC ? 29.0 : 63.0);
    int const bank_row = static_cast<int>(scaled);
    float const b = static_cast<float>(scaled - static_cast<double>(bank_row));

    spot.rows = bank + bank_row;

    if (CUBIC)
    {
        // Catmull-Rom (drb-audio-converter.c:861-874)
        float const b2 = b * b, b3 = b2 * b;

        spot.w[0] = -0.5f * b + b2 - 0.5f * b3;
        spot.w[1] = 1.0f - 2.5f * b2 + 1.5f * b3;
        spot.w[2] = 0.5f * b + 2.0f * b2 - 1.5f * b3;
        spot.w[3] = -0.5f * b2 + 0.5f * b3;
        spot.v[0] = -0.5f + 2.0f * b - 1.5f * b2;
        spot.v[1] = -5.0f * b + 4.5f * b2;
        spot.v[2] = 0.5f + 4.0f * b - 4.5f * b2;
        spot.v[3] = -b + 1.5f * b2;
    }
    else
    {
        spot.w[0] = 1.0f - b;
        spot.w[1] = b;
        spot.w[2] = spot.w[3] = 0.0f;
        spot.v[0] = -1.0f;
        spot.v[1] = 1.0f;
        spot.v[2] = spot.v[3] = 0.0f;
    }
}

// Coefficients of up to kMaxTypes block types at once, from the copy of the
// bank in LDS (a gather of 64 different entries per instruction, which LDS
// takes in its stride and the vector memory path does not): the bank rows of
// all types are asked for before the first is used, in four parts for the
// cubic bank, which needs twice the registers.
template <int STEPS, bool CUBIC>
__device__ __forceinline__ void build_sinc_types
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call, Sinc_Anchor const & anchor,
        float const * bank, int first_block, int count, Sinc_Type<STEPS> (& types) [kMaxTypes], int * fault_flag
    )
{
    constexpr int kReach = CUBIC ? 4 : 2;
    constexpr int kHalves = CUBIC ? 4 : 1;
    constexpr int kHalf = (STEPS + kHalves - 1) / kHalves;

    int const quad = (threadIdx.x & 63) >> 4;
    int const taps = chain.sinc_taps;
    float const rows_per_unit = CUBIC ? 29.0f : 63.0f;
    Sinc_Spot spots [kMaxTypes];

    #pragma unroll
    for (int i = 0; i < kMaxTypes; i++)
    {
        // (types past the wave's run are placed like its last one: harmless, and
        // every lane keeps to one path)
        int const block = min(first_block + min(i, count - 1), chain.blocks - 1);

        spot_sinc_type<STEPS, CUBIC>(chain, call, anchor, bank, block, spots[i], fault_flag);
    }

    #pragma unroll
    for (int half = 0; half < kHalves; half++)
    {
        float k [kMaxTypes][kHalf][kReach];

        #pragma unroll
        for (int i = 0; i < kMaxTypes; i++)
        {
            #pragma unroll
            for (int n = 0; n < kHalf; n++)
            {
                int const t = 4 * (half * kHalf + n) + quad - spots[i].skew;
                int const tap = t >= 0 && t < taps ? t : 0;

                #pragma unroll
                for (int j = 0; j < kReach; j++) k[i][n][j] = spots[i].rows[tap * kBankTapStride + j];
            }
        }

        #pragma unroll
        for (int i = 0; i < kMaxTypes; i++)
        {
            #pragma unroll
            for (int n = 0; n < kHalf; n++)
            {
                int const s = half * kHalf + n;

                if (s >= STEPS) continue;

                int const t = 4 * s + quad - spots[i].skew;
                bool const live = t >= 0 && t < taps;
                float value = 0.0f, slope = 0.0f;

                #pragma unroll
                for (int j = 0; j < kReach; j++)
                {
                    value = fmaf(spots[i].w[j], k[i][n][j], value);
                    slope = fmaf(spots[i].v[j], k[i][n][j], slope);
                }

                types[i].a[s] = live ? value : 0.0f;

                unsigned const rounded = (__float_as_uint(live ? slope * rows_per_unit : 0.0f) + 0x8000u) >> 16;

                if (s & 1) types[i].g[s / 2] |= rounded << 16; else types[i].g[s / 2] = rounded;
            }
        }
    }

    #pragma unroll
    for (int i = 0; i < kMaxTypes; i++) types[i].first_row = spots[i].first_row;
}

// Rows first .. first + 15 of the sinc output (counted from the segment's
// start; `in_period` from the period's start), four per lane.
template <bool DOWN, bool PLAIN>
__device__ __forceinline__ void write_sinc
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call, Tile const & tile, Buffers const & buf,
        Lane const & lane, DRB_GLOBAL float * target_lane, int channel, long long target_channel_stride,
        int in_period, int first, float4_t acc
    )
{
    bool const whole = in_period + 16 <= chain.period_out;

    if (DOWN)
    {
        // rows alternate between the planes of d; a period may start on either
        int const half = chain.seg_s >> 1;

        if (whole && ((first + 15) >> 1) < half - chain.look_d)
        {
            int const origin = 16 * ((first >> 1) + chain.look_d) + lane.planes;
            float * const even = buf.d_even + origin, * const odd = buf.d_odd + origin;

            if (first & 1)
            {
                odd[0] = acc[0];
                even[16] = acc[1];
                odd[16] = acc[2];
                even[32] = acc[3];
            }
            else
            {
                even[0] = acc[0];
                odd[0] = acc[1];
                even[16] = acc[2];
                odd[16] = acc[3];
            }

            return;
        }

        // blocks wholly inside the segment's tail: the same four stores, and
        // again into the next segment's lookback
        if (whole && (first >> 1) >= half - chain.look_d)
        {
            int const origin = 16 * ((first >> 1) + chain.look_d) + lane.planes;
            float * const even = buf.d_even + origin, * const odd = buf.d_odd + origin;
            int const back = 1 - 16 * half;
            bool const more = lane.column < 15;

            if (first & 1)
            {
                odd[0] = acc[0];
                even[16] = acc[1];
                odd[16] = acc[2];
                even[32] = acc[3];

                if (more)
                {
                    odd[back] = acc[0];
                    even[back + 16] = acc[1];
                    odd[back + 16] = acc[2];
                    even[back + 32] = acc[3];
                }
            }
            else
            {
                even[0] = acc[0];
                odd[0] = acc[1];
                even[16] = acc[2];
                odd[16] = acc[3];

                if (more)
                {
                    even[back] = acc[0];
                    odd[back] = acc[1];
                    even[back + 16] = acc[2];
                    odd[back + 16] = acc[3];
                }
            }

            return;
        }

        #pragma unroll
        for (int r = 0; r < 4; r++)
        {
            int const t = first + 4 * lane.quad + r, g = t >> 1;

            if (in_period + 4 * lane.quad + r >= chain.period_out) continue;

            float * const plane = (t & 1) ? buf.d_odd : buf.d_even;

            plane[cell(g + chain.look_d, lane.column)] = acc[r];

            // the tail of a segment is also the lookback of the next one
            if (g >= half - chain.look_d && lane.column < 15) plane[cell(g - half + chain.look_d, lane.column + 1)] = acc[r];
        }
    }
    else
    {
        if (lane.column < tile.first_column) return;

        if (PLAIN && whole && tile.inside)
        {
            *reinterpret_cast<DRB_GLOBAL float4_unaligned *>(target_lane + first) =
                float4_unaligned{ acc[0], acc[1], acc[2], acc[3] };
            return;
        }

        if (whole && tile.inside)
        {
            int const frame = tile.base_y + lane.column * chain.seg_y + first + 4 * lane.quad;

            float const four [4] = { acc[0], acc[1], acc[2], acc[3] };

            target_four(chain, call, target_channel_stride, channel, frame, four);

            return;
        }

        #pragma unroll
        for (int r = 0; r < 4; r++)
        {
            int const frame = tile.base_y + lane.column * chain.seg_y + first + 4 * lane.quad + r;

            if (in_period + 4 * lane.quad + r < chain.period_out && frame < call.target_frames)
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
__device__ __forceinline__ float const * sinc_operand
    (
        Drb_Fast_Chain const & chain, float const * lds, Buffers const & buf, Lane const & lane, int row
    )
{
    if (!PLANES) return buf.u + 16 * (row + chain.look_u) + lane.operand;

    return lds + 16 * (row >> 1) + ((row & 1) ? lane.sinc_odd : lane.sinc_even);
}

// Row-block chains [first, last) of the tile belong to this wave; chain q is
// period q % periods of block q / periods, and types[i] belongs to the i-th
// block of the run.
template <int STEPS, bool PLANES, bool DOWN, bool PLAIN>
__device__ __forceinline__ void stage_sinc
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call, Tile const & tile,
        float const * lds, Buffers const & buf, Lane const & lane, DRB_GLOBAL float * target_lane,
        int channel, long long target_channel_stride,
        Sinc_Type<STEPS> const (& types) [kMaxTypes], int first, int last
    )
{
    constexpr int kStride = PLANES ? 32 : 64;

    int const first_block = first / chain.periods;

    #pragma unroll
    for (int i = 0; i < kMaxTypes; i++)
    {
        int const block = first_block + i;
        int period = max(first - block * chain.periods, 0);
        int const period_end = min(last - block * chain.periods, chain.periods);

        for (; period < period_end; period++)
        {
            float b [STEPS];
            int const row = types[i].first_row + period * chain.period_in;

            load_rows<STEPS, kStride>(sinc_operand<PLANES>(chain, lds, buf, lane, row), b);

            float4_t acc = multiply<STEPS>(types[i].a, b);

            // Where the phase differs noticeably from segment to segment, every
            // segment (column) adds its own first-order term: the same product
            // with the coefficients' derivatives, times its distance from the
            // middle segment, whose phase the coefficients have.
            if (chain.column_drift != 0.0f)
            {
                float slope [STEPS];

                #pragma unroll
                for (int s = 0; s < STEPS; s++)
                {
                    unsigned const pair = types[i].g[s / 2];

                    slope[s] = __uint_as_float((s & 1) ? pair & 0xffff0000u : pair << 16);
                }

                float4_t const tilt = multiply<STEPS>(slope, b);
                float const away = static_cast<float>(lane.column - 8) * chain.column_drift;

                #pragma unroll
                for (int r = 0; r < 4; r++) acc[r] = fmaf(away, tilt[r], acc[r]);
            }

            write_sinc<DOWN, PLAIN>(chain, call, tile, buf, lane, target_lane, channel, target_channel_stride,
                                    16 * block, period * chain.period_out + 16 * block, acc);
        }
    }
}

// ---- down-sampler --------------------------------------------------------------------
// y[f] = 0.5 d[2 f - N] + sum_j h[j] d[2 (f - N + j) + 1]
template <bool PLAIN>
__device__ __forceinline__ void write_down
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call, Tile const & tile, Buffers const & buf,
        Lane const & lane, DRB_GLOBAL float * target_lane, int channel, long long target_channel_stride,
        int block, float4_t acc, int first_column = 1
    )
{
    float const * const centre = buf.d_even + 16 * (16 * block - chain.down_taps / 2 + chain.look_d) + lane.result;
    float value [4];

    #pragma unroll
    for (int r = 0; r < 4; r++) value[r] = fmaf(0.5f, centre[16 * r], acc[r]);

    if (lane.column < first_column) return;

    if (PLAIN && tile.inside && 16 * block + 16 <= chain.seg_y)
    {
        *reinterpret_cast<DRB_GLOBAL float4_unaligned *>(target_lane + 16 * block) =
            float4_unaligned{ value[0], value[1], value[2], value[3] };
        return;
    }

    if (tile.inside && 16 * block + 16 <= chain.seg_y)
    {
        int const frame = tile.base_y + lane.column * chain.seg_y + 16 * block + 4 * lane.quad;

        target_four(chain, call, target_channel_stride, channel, frame, value);

        return;
    }

    #pragma unroll
    for (int r = 0; r < 4; r++)
    {
        int const t = 16 * block + 4 * lane.quad + r;
        int const frame = tile.base_y + lane.column * chain.seg_y + t;

        if (t < chain.seg_y && frame < call.target_frames)
        {
            target_frame<PLAIN>(chain, call, target_channel_stride, channel, frame, value[r]);
        }
    }
}

template <int STEPS, bool PLAIN>
__device__ __forceinline__ void stage_down
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call, Tile const & tile, Buffers const & buf,
        Lane const & lane, DRB_GLOBAL float * target_lane, int channel, long long target_channel_stride,
        int wave, float const (& a) [STEPS], int first_column = 1
    )
{
    int const blocks = (chain.seg_y + 15) / 16;
    float const * const origin = buf.d_odd + 16 * (chain.look_d - chain.down_taps) + lane.operand;

    // where the blocks do not divide evenly, the waves that carried one more in
    // the up-sampler carry one less here
    int block = wave + (chain.waves >> 1);

    if (block >= chain.waves) block -= chain.waves;

    for (; block < blocks; block += chain.waves)
    {
        float b [STEPS];

        load_rows<STEPS, 64>(origin + 256 * block, b);
        write_down<PLAIN>(chain, call, tile, buf, lane, target_lane, channel, target_channel_stride, block,
                          multiply<STEPS>(a, b), first_column);
    }
}

// ---- source tile: global -> registers -> LDS -------------------------------------------

// What a thread holds of the next tile: FETCH groups of four consecutive
// frames, (threads / 4) rows apart -- up to (threads / 4) FETCH rows of 16
// segments per workgroup.  Planar float32 sources travel as the floats they
// are, other sample types and layouts as raw words, converted when stored.
template <int FETCH, bool PLAIN>
struct Fetch
{
    float4_t v [FETCH];
};

template <int FETCH>
struct Fetch<FETCH, false>
{
    int raw [FETCH][4];
};

__device__ __forceinline__ long long chunk_index
    (
        Drb_Fast_Chain const & chain, long long channel_stride, int channel, int frame
    )
{
    return chain.layout == kInterleaved ? static_cast<long long>(frame) * chain.channels + channel
                                        : channel * channel_stride + frame;
}

// How four consecutive frames of one channel lie in the chunk:
//   0  apart: one load per sample
//   1  16-bit, two interleaved channels: the four frames are one dwordx4, a dword per frame
//   2  16-bit, contiguous (planar, or one channel): one dwordx2
//   3  32-bit, contiguous: one dwordx4
__device__ __forceinline__ int chunk_packing (Drb_Fast_Chain const & chain)
{
    bool const contiguous = chain.layout == kPlanar || chain.channels == 1;

    if (chain.sample_type == kSampleI16) return contiguous ? 2 : chain.channels == 2 ? 1 : 0;

    return contiguous ? 3 : 0;
}

__device__ __forceinline__ float raw_sample (int raw, int type)
{
    return type == kSampleI16 ? static_cast<float>(raw) * (1.0f / 32768.0f)
         : type == kSampleI32 ? static_cast<float>(raw) * (1.0f / 2147483648.0f)
         : __int_as_float(raw);
}

// Whether every frame a tile's fetch touches (FETCH groups per thread, whether
// or not the tile has that many rows) lies inside the step's chunk.
template <int FETCH>
__device__ __forceinline__ bool tile_in_chunk (Drb_Fast_Chain const & chain, Drb_Device_Call const & call, int base_x)
{
    return base_x - chain.look_x >= 0
        && base_x + 15 * chain.seg_x + 16 * chain.waves * FETCH - chain.look_x <= call.source_frames;
}

// Four neighbouring lanes fetch 16 consecutive frames of one segment (thread t:
// segment (t >> 2) & 15, rows 4 (t & 3) + 16 (t >> 6) + (threads / 4) i + {0..3}), a
// dwordx4 per group from the step's chunk or from the stream's history,
// whichever holds all four frames.  The loads are unconditional (a group that
// straddles a seam or an end reads the last history frames instead and is
// mended when the tile is stored), so that nothing waits for them before then.
template <int FETCH>
__device__ __forceinline__ void fetch_tile
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call, int base_x,
        float const * history, long long source_channel_stride, int channel, Fetch<FETCH, true> & fetch
    )
{
    int const segment = (threadIdx.x >> 2) & 15, group = (threadIdx.x & 3) + 4 * (threadIdx.x >> 6);
    int const first = base_x + segment * chain.seg_x + 4 * group - chain.look_x;
    int const pass = 16 * chain.waves;   // rows the workgroup fetches per group of four each
    DRB_GLOBAL float const * const source = global_pointer<float const>(call.source) + channel * source_channel_stride;
    DRB_GLOBAL float const * const past =
        global_pointer<float const>(reinterpret_cast<unsigned long long>(history)) + channel * chain.history + chain.history;

    // Tiles wholly inside the step's chunk (all but a stream's first and last):
    // one address, the rest are constants of the instructions.
    if (tile_in_chunk<FETCH>(chain, call, base_x))
    {
        DRB_GLOBAL float const * const from = source + first;

        #pragma unroll
        for (int i = 0; i < FETCH; i++)
        {
            fetch.v[i] = *reinterpret_cast<DRB_GLOBAL float4_unaligned const *>(from + pass * i);
        }

        return;
    }

    #pragma unroll
    for (int i = 0; i < FETCH; i++)
    {
        // (also for groups past the tile's last row: a load that depends on
        // nothing is cheaper than one the next tile's loads have to wait for)
        int const frame = first + pass * i;
        bool const in_chunk = frame >= 0 && frame + 4 <= call.source_frames;
        bool const in_past = frame + 4 <= 0 && frame >= -chain.history;
        DRB_GLOBAL float const * const from = in_chunk ? source + frame : past + (in_past ? frame : -4);

        fetch.v[i] = *reinterpret_cast<DRB_GLOBAL float4_unaligned const *>(from);
    }
}

// Other sample types and layouts: tiles wholly inside the step's chunk travel
// as raw words (four frames as one load where they lie together); the first
// and last tiles of a stream are not fetched ahead but loaded when stored.
template <int FETCH>
__device__ __forceinline__ void fetch_tile
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call, int base_x,
        float const *, long long source_channel_stride, int channel, Fetch<FETCH, false> & fetch
    )
{
    typedef int int4_unaligned __attribute__((ext_vector_type(4), aligned(4)));
    typedef int int2_unaligned __attribute__((ext_vector_type(2), aligned(2)));

    if (!tile_in_chunk<FETCH>(chain, call, base_x)) return;

    int const segment = (threadIdx.x >> 2) & 15, group = (threadIdx.x & 3) + 4 * (threadIdx.x >> 6);
    int const first = base_x + segment * chain.seg_x + 4 * group - chain.look_x;
    int const pass = 16 * chain.waves;
    int const packing = chunk_packing(chain);

    #pragma unroll
    for (int i = 0; i < FETCH; i++)
    {
        int const frame = first + pass * i;
        long long const index = chunk_index(chain, source_channel_stride, channel, frame);

        // (every path leaves all four words, through one variable each: paths
        // that wrote different elements would turn the registers into an array
        // in memory)
        int w0, w1, w2 = 0, w3 = 0;

        if (packing == 1)
        {
            int4_unaligned const four = *reinterpret_cast<DRB_GLOBAL int4_unaligned const *>(global_pointer<short const>(call.source) + (index - channel));

            w0 = four[0]; w1 = four[1]; w2 = four[2]; w3 = four[3];
        }
        else if (packing == 2)
        {
            int2_unaligned const two = *reinterpret_cast<DRB_GLOBAL int2_unaligned const *>(global_pointer<short const>(call.source) + index);

            w0 = two[0]; w1 = two[1];
        }
        else if (packing == 3)
        {
            int4_unaligned const four = *reinterpret_cast<DRB_GLOBAL int4_unaligned const *>(global_pointer<int const>(call.source) + index);

            w0 = four[0]; w1 = four[1]; w2 = four[2]; w3 = four[3];
        }
        else if (chain.sample_type == kSampleI16)
        {
            DRB_GLOBAL short const * const samples = global_pointer<short const>(call.source);

            w0 = samples[chunk_index(chain, source_channel_stride, channel, frame)];
            w1 = samples[chunk_index(chain, source_channel_stride, channel, frame + 1)];
            w2 = samples[chunk_index(chain, source_channel_stride, channel, frame + 2)];
            w3 = samples[chunk_index(chain, source_channel_stride, channel, frame + 3)];
        }
        else
        {
            DRB_GLOBAL int const * const samples = global_pointer<int const>(call.source);

            w0 = samples[chunk_index(chain, source_channel_stride, channel, frame)];
            w1 = samples[chunk_index(chain, source_channel_stride, channel, frame + 1)];
            w2 = samples[chunk_index(chain, source_channel_stride, channel, frame + 2)];
            w3 = samples[chunk_index(chain, source_channel_stride, channel, frame + 3)];
        }

        fetch.raw[i][0] = w0;
        fetch.raw[i][1] = w1;
        fetch.raw[i][2] = w2;
        fetch.raw[i][3] = w3;
    }
}

template <int FETCH>
__device__ __forceinline__ void store_tile
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call, Buffers const & buf, int base_x,
        float const * history, long long source_channel_stride, int channel, Fetch<FETCH, true> & fetch
    )
{
    int const segment = (threadIdx.x >> 2) & 15, group = (threadIdx.x & 3) + 4 * (threadIdx.x >> 6);
    int const rows = chain.seg_x + chain.look_x;
    int const first = base_x + segment * chain.seg_x + 4 * group - chain.look_x;
    int const pass = 16 * chain.waves;
    bool const inside = tile_in_chunk<FETCH>(chain, call, base_x);   // no seams: as decided for the fetch
    float * const mine = buf.x + cell(4 * group, segment);

    #pragma unroll
    for (int i = 0; i < FETCH; i++)
    {
        // every fetched group counts as used here, whichever rows below take
        // it: no load may look pending once the tile is stored
        asm volatile ("" : "+v"(fetch.v[i]));

        if (pass * i >= rows) continue;

        int const frame = first + pass * i;
        bool const in_chunk = inside || (frame >= 0 && frame + 4 <= call.source_frames);
        bool const in_past = !inside && frame + 4 <= 0 && frame >= -chain.history;
        float4_t & value = fetch.v[i];

        // groups of four frames across a seam or an end: one by one
        if (!in_chunk && !in_past)
        {
            #pragma unroll
            for (int e = 0; e < 4; e++)
            {
                value[e] = source_frame<true>(chain, call, history, source_channel_stride, channel, frame + e);
            }

            asm volatile ("" : "+v"(value));   // as above
        }

        if (pass * i + pass <= rows)
        {
            #pragma unroll
            for (int e = 0; e < 4; e++) mine[16 * (pass * i + e)] = value[e];
        }
        else
        {
            #pragma unroll
            for (int e = 0; e < 4; e++)
            {
                if (4 * group + pass * i + e < rows) mine[16 * (pass * i + e)] = value[e];
            }
        }
    }
}

template <int FETCH>
__device__ __forceinline__ void store_tile
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call, Buffers const & buf, int base_x,
        float const * history, long long source_channel_stride, int channel, Fetch<FETCH, false> & fetch
    )
{
    int const rows = chain.seg_x + chain.look_x;

    #pragma unroll
    for (int i = 0; i < FETCH; i++)
    {
        #pragma unroll
        for (int e = 0; e < 4; e++) asm volatile ("" : "+v"(fetch.raw[i][e]));   // counts as used, whatever follows
    }

    if (!tile_in_chunk<FETCH>(chain, call, base_x))
    {
        // stream edges and history: one sample at a time, straight into LDS
        #pragma unroll 1
        for (int i = threadIdx.x; i < 16 * rows; i += blockDim.x)
        {
            int const segment = i & 15, row = i >> 4;

            buf.x[i] = source_frame<false>(chain, call, history, source_channel_stride, channel,
                                           base_x + segment * chain.seg_x + row - chain.look_x);
        }

        return;
    }

    int const segment = (threadIdx.x >> 2) & 15, group = (threadIdx.x & 3) + 4 * (threadIdx.x >> 6);
    int const pass = 16 * chain.waves;
    int const packing = chunk_packing(chain);
    float * const mine = buf.x + cell(4 * group, segment);

    #pragma unroll
    for (int i = 0; i < FETCH; i++)
    {
        if (pass * i >= rows) continue;

        // (branches, not selects between array elements: those would turn the
        // registers into an array in memory)
        int word [4];

        if (packing == 1)
        {
            #pragma unroll
            for (int e = 0; e < 4; e++) word[e] = channel ? fetch.raw[i][e] >> 16 : (fetch.raw[i][e] << 16) >> 16;
        }
        else if (packing == 2)
        {
            word[0] = (fetch.raw[i][0] << 16) >> 16;
            word[1] = fetch.raw[i][0] >> 16;
            word[2] = (fetch.raw[i][1] << 16) >> 16;
            word[3] = fetch.raw[i][1] >> 16;
        }
        else
        {
            #pragma unroll
            for (int e = 0; e < 4; e++) word[e] = fetch.raw[i][e];
        }

        #pragma unroll
        for (int e = 0; e < 4; e++)
        {
            if (4 * group + pass * i + e < rows) mine[16 * (pass * i + e)] = raw_sample(word[e], chain.sample_type);
        }
    }
}

// Diagnostic build only (-DDRB_FAST_STAMPS): where a wave's cycles go, summed
// per phase into fault_flag[1 + 16 wave + phase] (the flag word itself is
// word 0), and chain.ablate switches stages off.  Never compiled into the
// library that ships.
#ifdef DRB_FAST_STAMPS
// (the counter is a scalar-memory read: left pending it turns every later LDS
// wait into a wait for everything, so it is waited for on the spot)
#  define STAMP(phase) do { long long const now_ = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory"); \
                            spent[phase] += now_ - mark; mark = now_; } while (0)
#  define ABLATED(bit) (chain.ablate & (bit))
#else
#  define STAMP(phase) do { } while (0)
#  define ABLATED(bit) false
#endif

template <int UP_STEPS, int SINC_STEPS, int DOWN_STEPS, int FETCH, bool CUBIC, bool PLAIN>
__global__ __launch_bounds__(kMaxThreads, 3) void drb_fast_kernel
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

    // LDS arrives as the last workgroup left it, and the zero-padded products
    // also read rows that nobody writes: they must hold numbers (0 x NaN = NaN)
    for (int i = threadIdx.x; i < chain.lds_floats; i += blockDim.x) lds[i] = 0.0f;

    __syncthreads();

    // the same for all lanes of a wave: kept in a scalar register so that what
    // depends on it branches instead of masking lanes
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    Lane lane;

    lane.column = threadIdx.x & 15;
    lane.quad = (threadIdx.x & 63) >> 4;
    lane.operand = cell(lane.quad, lane.column);
    lane.result = cell(4 * lane.quad, lane.column);
    lane.planes = cell(2 * lane.quad, lane.column);

    // Behind an up-sampler, sinc-input row F + quad (F even) is source-plane
    // row (F + quad - N) / 2 for even quads and odd-plane row (F + quad - 1) / 2
    // for odd ones; one row further on, the planes swap.
    {
        int const source = chain.off_x + 16 * (chain.look_x - chain.up_taps / 2) + lane.column;
        int const odd = chain.off_u + 16 * chain.look_u + lane.column;

        lane.sinc_even = (lane.quad & 1) ? odd + 16 * (lane.quad >> 1) : source + 16 * (lane.quad >> 1);
        lane.sinc_odd = (lane.quad & 1) ? source + 16 * ((lane.quad + 1) >> 1) : odd + 16 * (lane.quad >> 1);
    }

    Buffers const buf = { lds + chain.off_x, lds + chain.off_u, lds + chain.off_d_odd,
                          lds + chain.off_d_even, lds + chain.off_bank };
    float const * const stream_history = history + static_cast<long long>(stream) * chain.channels * chain.history;

    // ---- once per workgroup -----------------------------------------------------
    // Half-band taps with 16 zeros in front and behind.  Row m of a 16-row block
    // multiplies input k of the block's window by
    //   up:   2 h[k - m - 1]   (window starts at row 16 block - N)
    //   down:   h[k - m]
    // so a lane (m = column, k = 4 step + quad) reads its A operands four floats
    // apart from one place in the table.
    float * const taps_up = lds + chain.off_taps;
    float * const taps_down = taps_up + chain.up_taps + 32;

    for (int i = threadIdx.x; i < chain.up_taps + 32; i += blockDim.x)
    {
        int const j = i - 16;
        taps_up[i] = UP_STEPS && j >= 0 && j < chain.up_taps ? 2.0f * table_at(chain.up_table)[j] : 0.0f;
    }

    for (int i = threadIdx.x; i < chain.down_taps + 32; i += blockDim.x)
    {
        int const j = i - 15;
        taps_down[i] = DOWN_STEPS && j >= 0 && j < chain.down_taps ? table_at(chain.down_table)[j] : 0.0f;
    }

    float const * const a_up_lane = taps_up + 15 + lane.quad - lane.column;
    float const * const a_down_lane = taps_down + 15 + lane.quad - lane.column;

    int const delay_behind = chain.delay_place == 1 ? call.delay : 0;
    int const delay_front = chain.delay_place == 2 ? call.delay : 0;
    // how far the sub-sample position moves from one segment to the next, on top
    // of the whole periods a segment is long
    float const drift_per_segment = static_cast<float>(static_cast<double>(
          static_cast<long long>(chain.seg_s) * chain.target_delta
        - static_cast<long long>(chain.seg_u) * chain.source_delta) * chain.scale);

    // this wave's run of the fractional stage's row-block chains
    int const chains = chain.blocks * chain.periods;
    int const share = (chains + chain.waves - 1) / chain.waves;
    int const first_chain = wave * share, last_chain = min(first_chain + share, chains);

#ifdef DRB_FAST_STAMPS
    long long spent [11] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    long long mark = __builtin_readcyclecounter();
#endif

    Sinc_Type<SINC_STEPS> types [kMaxTypes];
    Sinc_Anchor anchor = { 0, 0, 0, 0, 0 };
    // of this workgroup's tiles, which lie tile_stride apart
    // (a walk over all of a stream's tiles spreads its rebuilds evenly: 19 tiles
    // are one stretch of 19, not 16 + 3)
    // kRebuild was measured at 44.1 -> 48 kHz, which drifts 1.74e-8 source frames per
    // segment; rate pairs that drift faster (176.4 <-> 192 kHz: 4.5e-8 and 4.8e-8) rebuild
    // that much more often, so that the drift a build is followed over stays the same
    float const drift_size = fabsf(drift_per_segment);
    int const cadence = drift_size > 1.8e-8f ? max(2, static_cast<int>(static_cast<float>(kRebuild) * 1.8e-8f / drift_size)) : kRebuild;
    int rebuild_every = max(1, (chain.rebuild_tiles > 0 ? chain.rebuild_tiles : cadence) / tile_stride);

    if (tile_stride == 1 && !chain.no_carry && chain.rebuild_tiles <= 0)
    {
        int const segments = (call.target_frames + chain.seg_y - 1) / chain.seg_y;
        int const tiles = segments > 15 ? 1 + (segments - 15 + 15) / 16 : 1;
        int const stretches = (tiles + cadence - 1) / cadence;

        rebuild_every = (tiles + stretches - 1) / stretches;
    }


    float applied_drift = 0.0f;
    int since_build = rebuild_every;
    Fetch<FETCH, PLAIN> fetch;

    // Places tile `index`: from the exact phase every kRebuild tiles, by whole
    // periods in between.
    // A workgroup that walks over ALL tiles of its stream (tile_stride 1) hands
    // the tail of each tile's last segment on to the next tile as the lookback of
    // its segment 0, so that only the stream's first tile spends a segment on
    // warming up; otherwise every tile does.
    bool const carry = tile_stride == 1 && !chain.no_carry;
    // segments from the first to the last of the tiles that one build serves (this
    // workgroup's tiles lie tile_stride apart: a build per tile serves that tile alone,
    // however far away the workgroup's next one is)
    int const rebuild_span = (rebuild_every - 1) * (carry ? 16 : 15 * tile_stride);

    auto place = [&] (int index) -> Tile
    {
        Tile tile;

        tile.index = index;
        tile.first_column = carry && index > 0 ? 0 : 1;
        tile.segment = carry ? (index > 0 ? 16 * index - 1 : -1) : 15 * (first_tile + index * tile_stride) - 1;
        tile.base_y = tile.segment * chain.seg_y;
        tile.inside = tile.base_y + 16 * chain.seg_y <= call.target_frames;
        tile.rebuild = since_build >= rebuild_every;
        tile.drift = 0.0f;

        if (tile.rebuild)
        {
            // The coefficients are built for a tile half way to the next
            // rebuild, so that the drift they follow is at most half as long.
            // Window ends of that tile's middle segment (8) define the phases
            // of all segments.
            anchor.segment = tile.segment + rebuild_span / 2;

            int const anchor_y = anchor.segment * chain.seg_y;
            int const anchor_s = (DOWN_STEPS ? 2 * anchor_y : anchor_y) - delay_behind;
            int consumed;
            double unused;
            long long const phase = call.phase + static_cast<long long>(anchor_s + 8 * chain.seg_s) * chain.target_delta;

            sinc_position(phase, chain.source_delta, chain.scale, chain.sinc_pull, consumed, unused);

            // the first row's window ends at row -2 or -1 (so that the last row's
            // window still ends inside the segment); the producer-side index must
            // be even when an up-sampler feeds the stage
            anchor.base_u = consumed - 8 * chain.seg_u + 2;

            if (UP_STEPS && ((anchor.base_u - delay_front) & 1)) anchor.base_u -= 1;

            anchor.base_x = UP_STEPS ? (anchor.base_u - delay_front) / 2 : anchor.base_u - delay_front;
            anchor.rest = phase - static_cast<long long>(consumed) * chain.source_delta;
            anchor.base_s = anchor_s;
            since_build = 0;
        }

        // Segments lie whole periods from the anchor on every edge, and the
        // phase's departure from that is the drift the coefficients follow.
        {
            int const segments = tile.segment - anchor.segment;

            tile.base_x = anchor.base_x + segments * chain.seg_x;
            tile.drift = static_cast<float>(segments) * drift_per_segment;
        }

        since_build += 1;

        return tile;
    };

    // whether the workgroup's tile `index` has any target frame to write
    auto has_work = [&] (int index) -> bool
    {
        int const first_segment = carry ? (index > 0 ? 16 * index - 1 : 0) : 15 * (first_tile + index * tile_stride);

        return first_segment * chain.seg_y < call.target_frames;
    };

    // ---- tiles of this stream and channel ------------------------------------------
    Tile tile = place(0);

    if (!ABLATED(1))
    {
        fetch_tile(chain, call, tile.base_x, stream_history, source_channel_stride, channel, fetch);
        store_tile(chain, call, buf, tile.base_x, stream_history, source_channel_stride, channel, fetch);
    }

    for (;;)
    {
        DRB_GLOBAL float * const target_lane = global_pointer<float>(call.target) + channel * target_channel_stride
            + tile.base_y + lane.column * chain.seg_y + 4 * lane.quad;

        STAMP(0);

        // the tile's source rows are in LDS already (see below)

        STAMP(1);

        __syncthreads();   // source tile (and bank) in place

        STAMP(2);

        int const next_index = tile.index + 1;
        bool const more = has_work(next_index);
        Tile next = tile;
        Sinc_Anchor const build_anchor = anchor;   // placing the next tile may move the anchor on

        // The next tile's source travels while this one is filtered (from memory
        // that nobody has touched before, it takes microseconds).  Fetching and
        // storing happen whether or not a tile follows (the last one fetches its own rows
        // again): every load then meets its wait within the iteration, and none
        // looks pending when the loop comes round to load into the same registers.
        if (more) next = place(next_index);

        if (!ABLATED(1))
        {
            fetch_tile(chain, call, next.base_x, stream_history, source_channel_stride, channel, fetch);
        }

        STAMP(10);

        if (tile.rebuild && !ABLATED(4))
        {
            // The coefficient bank of the fractional stage comes to LDS as
            // [tap][row] for the occasion: it may lie where d's planes are,
            // which nobody touches between the barrier above and the next one.
            {
                int const shift = chain.sinc_taps == 32 ? 5 : 4;           // 16 or 32 taps
                int const count = (CUBIC ? 32 : 64) << shift;
                float const * const source = table_at(chain.sinc_table);
                float value [8];

                #pragma unroll
                for (int n = 0; n < 8; n++)
                {
                    int const i = threadIdx.x + n * blockDim.x;

                    value[n] = source[min(i, count - 1)];
                }

                #pragma unroll
                for (int n = 0; n < 8; n++)
                {
                    int const i = threadIdx.x + n * blockDim.x;
                    int const row = i >> shift, tap = i - (row << shift);

                    if (i < count) buf.bank[tap * kBankTapStride + row] = value[n];
                }

                __syncthreads();
            }

            build_sinc_types<SINC_STEPS, CUBIC>(chain, call, build_anchor, buf.bank,
                                                first_chain / chain.periods, chain.types_per_wave, types, fault_flag);

            __syncthreads();   // the bank's place may be written again

            applied_drift = 0.0f;
        }

        STAMP(4);

        // The tail of the last tile's last segment becomes the lookback of this
        // tile's segment 0 in front of the decimator (nobody touches the planes of
        // d between the barrier above and the sinc stage's first store).
        if (DOWN_STEPS > 0 && tile.first_column == 0)
        {
            int const half = chain.seg_s >> 1;

            for (int i = threadIdx.x; i < 2 * chain.look_d; i += blockDim.x)
            {
                float * const plane = i < chain.look_d ? buf.d_odd : buf.d_even;
                int const row = i < chain.look_d ? i : i - chain.look_d;

                plane[cell(row, 0)] = plane[cell(half + row, 15)];
            }

            if (UP_STEPS == 0) __syncthreads();   // no barrier of an up-sampler's before the sinc stage stores
        }


        if (UP_STEPS > 0)
        {
            if (!ABLATED(2))
            {
                float a_up [UP_STEPS ? UP_STEPS : 1];

                load_rows<UP_STEPS ? UP_STEPS : 1, 4>(a_up_lane, a_up);
                stage_up<UP_STEPS ? UP_STEPS : 1>(chain, buf, lane, wave, a_up);
            }

            STAMP(5);

            __syncthreads();   // odd plane of the sinc input complete

            STAMP(6);
        }


        // coefficients follow the phase from the previous tile's to this tile's
        {
            float const step = tile.drift - applied_drift;

            #pragma unroll
            for (int i = 0; i < kMaxTypes; i++)
            {
                #pragma unroll
                for (int s = 0; s < SINC_STEPS; s++)
                {
                    unsigned const pair = types[i].g[s / 2];
                    float const slope = __uint_as_float((s & 1) ? pair & 0xffff0000u : pair << 16);

                    types[i].a[s] = fmaf(step, slope, types[i].a[s]);
                }
            }

            applied_drift = tile.drift;
        }

        if (!ABLATED(4))
        {
            stage_sinc<SINC_STEPS, (UP_STEPS > 0), (DOWN_STEPS > 0), PLAIN>
                (chain, call, tile, lds, buf, lane, target_lane, channel, target_channel_stride, types, first_chain, last_chain);
        }

        STAMP(7);

        __syncthreads();   // both planes of d complete; source plane free again

        STAMP(8);

        // The next tile's rows have long arrived; they go to LDS here rather
        // than at the top of the loop, where waiting for them would also wait
        // for the stores of the down-sampler below.
        // likewise the tail of the up-sampler's odd plane, for the next tile
        // (nobody touches that plane between the barrier above and the next
        // tile's up-sampler)
        if (UP_STEPS > 0 && more && next.first_column == 0)
        {
            for (int i = threadIdx.x; i < chain.look_u; i += blockDim.x) buf.u[cell(i, 0)] = buf.u[cell(chain.seg_x + i, 15)];
        }

        if (!ABLATED(1))
        {
            store_tile(chain, call, buf, next.base_x, stream_history, source_channel_stride, channel, fetch);
        }

        STAMP(3);

        if (DOWN_STEPS > 0 && !ABLATED(8))
        {
            float a_down [DOWN_STEPS ? DOWN_STEPS : 1];

            load_rows<DOWN_STEPS ? DOWN_STEPS : 1, 4>(a_down_lane, a_down);
            stage_down<DOWN_STEPS ? DOWN_STEPS : 1, PLAIN>
                (chain, call, tile, buf, lane, target_lane, channel, target_channel_stride, wave, a_down, tile.first_column);
        }

        STAMP(9);

        if (!more) break;

        tile = next;
    }

#ifdef DRB_FAST_STAMPS
    if ((threadIdx.x & 63) == 0)
    {
        for (int phase = 0; phase < 11; phase++)
        {
            atomicAdd(reinterpret_cast<unsigned long long *>(fault_flag) + 1 + 16 * wave + phase,
                      static_cast<unsigned long long>(spent[phase]));
        }
    }
#endif
}


// ---- a half-band decimator on its own ----------------------------------------------------
// Conversions that are one 2x down-sampler and nothing else (96 -> 48 kHz ...)
// run the same banded product without the fractional stage: a tile is 16
// segments of seg_y target frames; the source goes to LDS as its even and its
// odd samples ([pair][16 segments] each), lookback included, so every segment
// is output -- none is spent on warming up.  Planar float32 sources only.
//   y[f] = 0.5 x[2 f - N] + sum_j h[j] x[2 (f - N + j) + 1]   (drb-audio-converter.c:2033-2057)

constexpr int kPairFetch = 8;   // float4 (two sample pairs) per thread per tile: up to 256 pairs per segment of four waves

struct Pair_Fetch
{
    float4_t v [kPairFetch];
};

// Thread t fetches pairs 2 j, 2 j + 1 of segment (t >> 2) & 15 for j = (t & 3) + 4 (t >> 6) + (threads / 16) i.
__device__ __forceinline__ void fetch_pairs
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call, int first_frame,
        float const * history, long long source_channel_stride, int channel, Pair_Fetch & fetch
    )
{
    int const segment = (threadIdx.x >> 2) & 15, group = (threadIdx.x & 3) + 4 * (threadIdx.x >> 6);
    int const first = first_frame + 2 * segment * chain.seg_y + 4 * group;
    int const pass = 16 * chain.waves;   // frames between a thread's groups
    DRB_GLOBAL float const * const source = global_pointer<float const>(call.source) + channel * source_channel_stride;
    DRB_GLOBAL float const * const past =
        global_pointer<float const>(reinterpret_cast<unsigned long long>(history)) + channel * chain.history + chain.history;

    #pragma unroll
    for (int i = 0; i < kPairFetch; i++)
    {
        int const frame = first + pass * i;
        bool const in_chunk = frame >= 0 && frame + 4 <= call.source_frames;
        bool const in_past = frame + 4 <= 0 && frame >= -chain.history;
        DRB_GLOBAL float const * const from = in_chunk ? source + frame : past + (in_past ? frame : -4);

        fetch.v[i] = *reinterpret_cast<DRB_GLOBAL float4_unaligned const *>(from);
    }
}

__device__ __forceinline__ void store_pairs
    (
        Drb_Fast_Chain const & chain, Drb_Device_Call const & call, Buffers const & buf, int first_frame,
        float const * history, long long source_channel_stride, int channel, Pair_Fetch & fetch
    )
{
    int const segment = (threadIdx.x >> 2) & 15, group = (threadIdx.x & 3) + 4 * (threadIdx.x >> 6);
    int const pairs = chain.seg_y + chain.look_d;
    int const first = first_frame + 2 * segment * chain.seg_y + 4 * group;
    int const pass = 16 * chain.waves;
    int const mine = cell(2 * group, segment);

    #pragma unroll
    for (int i = 0; i < kPairFetch; i++)
    {
        asm volatile ("" : "+v"(fetch.v[i]));   // every group counts as used, whichever rows take it

        int const pair = 2 * group + 8 * chain.waves * i;

        if (pair >= pairs) continue;

        int const frame = first + pass * i;
        bool const in_chunk = frame >= 0 && frame + 4 <= call.source_frames;
        bool const in_past = frame + 4 <= 0 && frame >= -chain.history;
        float4_t & value = fetch.v[i];

        // groups of four frames across a seam or an end: one by one
        if (!in_chunk && !in_past)
        {
            #pragma unroll
            for (int e = 0; e < 4; e++)
            {
                value[e] = source_frame<true>(chain, call, history, source_channel_stride, channel, frame + e);
            }

            asm volatile ("" : "+v"(value));
        }

        int const at = mine + 16 * 8 * chain.waves * i;

        buf.d_even[at] = value[0];
        buf.d_odd[at] = value[1];

        if (pair + 1 < pairs)
        {
            buf.d_even[at + 16] = value[2];
            buf.d_odd[at + 16] = value[3];
        }
    }
}

template <int STEPS, bool PLAIN_OUT>
__global__ __launch_bounds__(kMaxThreads, 3) void drb_decimate_kernel
    (
        Drb_Fast_Chain const chain,
        Drb_Device_Call const * __restrict__ calls, int stream_count, int tile_stride,
        float const * __restrict__ history,
        long long source_channel_stride, long long target_channel_stride,
        int * fault_flag
    )
{
    extern __shared__ __attribute__((aligned(16))) float lds [];

    (void)fault_flag;

    int const pair = blockIdx.x % (stream_count * chain.channels);
    int const first_tile = blockIdx.x / (stream_count * chain.channels);
    int const stream = pair / chain.channels, channel = pair - stream * chain.channels;
    Drb_Device_Call const call = calls[stream];

    if (first_tile * chain.tile_frames >= call.target_frames) return;

    // LDS arrives as the last workgroup left it, and the zero-padded products
    // also read rows that nobody writes: they must hold numbers (0 x NaN = NaN)
    for (int i = threadIdx.x; i < chain.lds_floats; i += blockDim.x) lds[i] = 0.0f;

    __syncthreads();

    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    Lane lane;

    lane.column = threadIdx.x & 15;
    lane.quad = (threadIdx.x & 63) >> 4;
    lane.operand = cell(lane.quad, lane.column);
    lane.result = cell(4 * lane.quad, lane.column);
    lane.planes = cell(2 * lane.quad, lane.column);
    lane.sinc_even = lane.sinc_odd = 0;

    Buffers const buf = { nullptr, nullptr, lds + chain.off_d_odd, lds + chain.off_d_even, nullptr };
    float const * const stream_history = history + static_cast<long long>(stream) * chain.channels * chain.history;

    // taps between 16 zeros on either side: row m of a block multiplies input k by h[k - m]
    float * const taps_down = lds + chain.off_taps;

    for (int i = threadIdx.x; i < chain.down_taps + 32; i += blockDim.x)
    {
        int const j = i - 15;
        taps_down[i] = j >= 0 && j < chain.down_taps ? table_at(chain.down_table)[j] : 0.0f;
    }

    float const * const a_lane = taps_down + 15 + lane.quad - lane.column;

    // the slicer's delay: in front of the filter it shifts the source, behind it the target
    int const delay_front = chain.delay_place == 2 ? call.delay : 0;
    int const delay_behind = chain.delay_place == 1 ? call.delay : 0;

    // source frame of (segment 0, pair 0) of tile `index`
    auto first_frame_of = [&] (int index) -> int
    {
        return 2 * (index * chain.tile_frames - delay_behind - chain.look_d) - delay_front;
    };

    Pair_Fetch fetch;

    fetch_pairs(chain, call, first_frame_of(first_tile), stream_history, source_channel_stride, channel, fetch);

    for (int index = first_tile; ; index += tile_stride)
    {
        Tile tile;

        tile.index = index;
        tile.base_y = index * chain.tile_frames;
        tile.base_x = 0;
        tile.drift = 0.0f;
        tile.rebuild = false;
        tile.inside = tile.base_y + 16 * chain.seg_y <= call.target_frames;

        DRB_GLOBAL float * const target_lane = global_pointer<float>(call.target) + channel * target_channel_stride
            + tile.base_y + lane.column * chain.seg_y + 4 * lane.quad;

        store_pairs(chain, call, buf, first_frame_of(index), stream_history, source_channel_stride, channel, fetch);

        __syncthreads();   // both planes (and the taps) in place

        // the next tile's source travels while this one is filtered (the last
        // tile fetches its own again, so that every load meets its wait)
        bool const more = (index + tile_stride) * chain.tile_frames < call.target_frames;

        fetch_pairs(chain, call, first_frame_of(more ? index + tile_stride : index), stream_history,
                    source_channel_stride, channel, fetch);

        {
            float a [STEPS];

            load_rows<STEPS, 4>(a_lane, a);
            stage_down<STEPS, PLAIN_OUT>(chain, call, tile, buf, lane, target_lane, channel, target_channel_stride, wave, a, 0);
        }

        if (!more) break;

        __syncthreads();   // the planes may be written again
    }
}


// ---- a half-band interpolator on its own --------------------------------------------------
// One 2x up-sampler and nothing else (48 -> 96 kHz ..., or a stage of a longer
// chain run as a launch of its own): the source goes to LDS as for the cascade,
// every segment is output, and a lane that holds the odd outputs of four
// source rows reads the even ones beside them from the source plane and
// stores eight consecutive target frames.  Planar float32 only.
//   y[2 f] = x[f - N / 2],  y[2 f + 1] = 2 sum_j h[j] x[f - N + 1 + j]   (drb-audio-converter.c:1984-2004)
template <int STEPS, int FETCH>
__global__ __launch_bounds__(kMaxThreads, 3) void drb_interpolate_kernel
    (
        Drb_Fast_Chain const chain,
        Drb_Device_Call const * __restrict__ calls, int stream_count, int tile_stride,
        float const * __restrict__ history,
        long long source_channel_stride, long long target_channel_stride,
        int * fault_flag
    )
{
    extern __shared__ __attribute__((aligned(16))) float lds [];

    (void)fault_flag;

    int const pair = blockIdx.x % (stream_count * chain.channels);
    int const first_tile = blockIdx.x / (stream_count * chain.channels);
    int const stream = pair / chain.channels, channel = pair - stream * chain.channels;
    Drb_Device_Call const call = calls[stream];

    // the slicer's delay: in front of the filter it shifts the source, behind it
    // the target; tiles are laid out on the undelayed target, from an even frame
    int const delay_front = chain.delay_place == 2 ? call.delay : 0;
    int const delay_behind = chain.delay_place == 1 ? call.delay : 0;
    int const lead = (delay_behind + 1) & ~1;
    int const frames = call.target_frames + lead - delay_behind;   // undelayed frames to cover, from -lead

    if (first_tile * chain.tile_frames >= frames) return;

    // LDS arrives as the last workgroup left it, and the zero-padded products
    // also read rows that nobody writes: they must hold numbers (0 x NaN = NaN)
    for (int i = threadIdx.x; i < chain.lds_floats; i += blockDim.x) lds[i] = 0.0f;

    __syncthreads();

    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    Lane lane;

    lane.column = threadIdx.x & 15;
    lane.quad = (threadIdx.x & 63) >> 4;
    lane.operand = cell(lane.quad, lane.column);
    lane.result = cell(4 * lane.quad, lane.column);
    lane.planes = cell(2 * lane.quad, lane.column);
    lane.sinc_even = lane.sinc_odd = 0;

    Buffers const buf = { lds + chain.off_x, nullptr, nullptr, nullptr, nullptr };
    float const * const stream_history = history + static_cast<long long>(stream) * chain.channels * chain.history;

    // taps between 16 zeros on either side: row m of a block multiplies input k by 2 h[k - m - 1]
    float * const taps_up = lds + chain.off_taps;

    for (int i = threadIdx.x; i < chain.up_taps + 32; i += blockDim.x)
    {
        int const j = i - 16;
        taps_up[i] = j >= 0 && j < chain.up_taps ? 2.0f * table_at(chain.up_table)[j] : 0.0f;
    }

    float const * const a_lane = taps_up + 15 + lane.quad - lane.column;

    // source frame of (segment 0, row 0) of tile `index`
    auto base_of = [&] (int index) -> int
    {
        return (index * chain.tile_frames - lead) / 2 - delay_front;
    };

    Fetch<FETCH, true> fetch;

    fetch_tile(chain, call, base_of(first_tile), stream_history, source_channel_stride, channel, fetch);

    int const blocks = (chain.seg_x + 15) / 16;
    float const * const origin = buf.x + 16 * (chain.look_x - chain.up_taps) + lane.operand;
    float const * const centre = buf.x + 16 * (chain.look_x - chain.up_taps / 2) + lane.result;

    for (int index = first_tile; ; index += tile_stride)
    {
        store_tile(chain, call, buf, base_of(index), stream_history, source_channel_stride, channel, fetch);

        __syncthreads();   // source plane (and the taps) in place

        bool const more = (index + tile_stride) * chain.tile_frames < frames;

        fetch_tile(chain, call, base_of(more ? index + tile_stride : index), stream_history,
                   source_channel_stride, channel, fetch);

        float a [STEPS];

        load_rows<STEPS, 4>(a_lane, a);

        // target frame (delay included) of (segment 0, row 0)
        int const base_y = index * chain.tile_frames - lead + delay_behind;
        DRB_GLOBAL float * const out = global_pointer<float>(call.target) + channel * target_channel_stride;

        for (int block = wave; block < blocks; block += chain.waves)
        {
            float b [STEPS];

            load_rows<STEPS, 64>(origin + 256 * block, b);

            float4_t const odd = multiply<STEPS>(a, b);
            int const row = 16 * block + 4 * lane.quad;
            int const frame = base_y + 2 * (lane.column * chain.seg_x + row);
            float value [8];

            #pragma unroll
            for (int r = 0; r < 4; r++)
            {
                value[2 * r] = centre[16 * (16 * block + r)];
                value[2 * r + 1] = odd[r];
            }

            if (row + 4 <= chain.seg_x && frame >= 0 && frame + 8 <= call.target_frames)
            {
                *reinterpret_cast<DRB_GLOBAL float4_unaligned *>(out + frame) = float4_unaligned{ value[0], value[1], value[2], value[3] };
                *reinterpret_cast<DRB_GLOBAL float4_unaligned *>(out + frame + 4) = float4_unaligned{ value[4], value[5], value[6], value[7] };
            }
            else
            {
                #pragma unroll
                for (int e = 0; e < 8; e++)
                {
                    if (row + e / 2 < chain.seg_x && frame + e >= 0 && frame + e < call.target_frames) out[frame + e] = value[e];
                }
            }
        }

        if (!more) break;

        __syncthreads();   // the plane may be written again
    }
}

#include "drb_collapsed.inc"
#include "drb_split.inc"

typedef void Fast_Kernel
    (
        Drb_Fast_Chain, Drb_Device_Call const *, int, int, float const *, long long, long long, int *
    );

// The kernel images that exist: (up steps, sinc steps, down steps, groups of 64
// source rows a tile has at most), steps = (taps + 16) / 4 for a half-band,
// chain.ksteps for the fractional stage.
Fast_Kernel * pick_kernel (Drb_Fast_Chain const * chain)
{
    if (chain->kind == 1)
    {
        // one half-band decimator: planar float32 source, up to 256 pairs per segment
        if (chain->sample_type != kSampleF32 || chain->layout != kPlanar) return nullptr;
        if (chain->waves != 4 || chain->seg_y + chain->look_d > 8 * chain->waves * kPairFetch) return nullptr;

        bool const plain_out = chain->target_sample_type == kSampleF32 && chain->target_layout == kPlanar;

#define DRB_DECIMATE_CASE(STEPS) \
          case STEPS: return plain_out ? drb_decimate_kernel<STEPS, true> : drb_decimate_kernel<STEPS, false>;

        switch ((chain->down_taps + 16) / 4)
        {
          DRB_DECIMATE_CASE(6)
          DRB_DECIMATE_CASE(8)
          DRB_DECIMATE_CASE(10)
          DRB_DECIMATE_CASE(12)
          DRB_DECIMATE_CASE(14)
          DRB_DECIMATE_CASE(16)
        }

#undef DRB_DECIMATE_CASE

        return nullptr;
    }

    if (chain->kind == 2)
    {
        // one half-band interpolator: planar float32 on both sides, up to 192 rows per segment
        if (chain->sample_type != kSampleF32 || chain->layout != kPlanar) return nullptr;
        if (chain->target_sample_type != kSampleF32 || chain->target_layout != kPlanar) return nullptr;
        if (chain->waves != 4 || chain->seg_x + chain->look_x > 192) return nullptr;

        switch ((chain->up_taps + 16) / 4)
        {
          case 6: return drb_interpolate_kernel<6, 3>;
          case 8: return drb_interpolate_kernel<8, 3>;
          case 10: return drb_interpolate_kernel<10, 3>;
          case 12: return drb_interpolate_kernel<12, 3>;
          case 14: return drb_interpolate_kernel<14, 3>;
          case 16: return drb_interpolate_kernel<16, 3>;
        }

        return nullptr;
    }

    if (chain->pre_ups > 0) return nullptr;   // a folded chain of interpolators: the split kernel or none

    int const up = chain->has_up ? (chain->up_taps + 16) / 4 : 0;
    int const down = chain->has_down ? (chain->down_taps + 16) / 4 : 0;
    int const key = up * 10000 + chain->ksteps * 100 + down;
    int const rows = chain->seg_x + chain->look_x;
    int const pass = 16 * chain->waves;

    if (chain->waves * 64 > kMaxThreads || chain->waves < 1) return nullptr;

    bool const plain = chain->sample_type == kSampleF32 && chain->layout == kPlanar
                    && chain->target_sample_type == kSampleF32 && chain->target_layout == kPlanar;

#define DRB_FAST_CASE(KEY, UP, SINC, DOWN, FETCH, CUBIC) \
      case KEY: if (rows > pass * FETCH || (chain->sinc_cubic != 0) != CUBIC) return nullptr; \
                return plain ? drb_fast_kernel<UP, SINC, DOWN, FETCH, CUBIC, true> \
                             : drb_fast_kernel<UP, SINC, DOWN, FETCH, CUBIC, false>;

    switch (key)
    {
      DRB_FAST_CASE(141214, 14, 12, 14, 3, false)   // good, rate going up
      DRB_FAST_CASE(141314, 14, 13, 14, 4, false)   // good, rate going down
      DRB_FAST_CASE(120812, 12, 8, 12, 3, true)    // fine
      DRB_FAST_CASE(120912, 12, 9, 12, 3, true)
      DRB_FAST_CASE(161216, 16, 12, 16, 4, true)   // best
      DRB_FAST_CASE(161316, 16, 13, 16, 4, true)
      DRB_FAST_CASE(101414, 10, 14, 14, 4, false) // 64 -> 128 -> 96 -> 48 kHz: the core of 16 -> 48 kHz, good
      DRB_FAST_CASE(161312, 16, 13, 12, 4, true)  // 192 -> 384 -> 352.8 -> 176.4 kHz: the core of 192 -> 44.1 kHz, best
      DRB_FAST_CASE(141310, 14, 13, 10, 4, false) // the same, good
      DRB_FAST_CASE(81414, 8, 14, 14, 4, false)   // the core of 8 -> 48 kHz, good
      DRB_FAST_CASE(141208, 14, 12, 8, 3, false)  // the core of 48 -> 8 kHz, good
      DRB_FAST_CASE(800, 0, 8, 0, 4, false)         // poor
      DRB_FAST_CASE(900, 0, 9, 0, 4, false)
    }

#undef DRB_FAST_CASE

    return nullptr;
}

} // namespace

extern "C" int drb_hip_fast_kernel_exists (Drb_Fast_Chain const * chain)
{
    if (chain->kind == 3) return 0;   // a collapsed geometry before its split form is chosen: no kernel of its own

    return chain->kind == 4 ? pick_split(chain) != nullptr : pick_kernel(chain) != nullptr;
}

extern "C" long long drb_hip_collapse_set_words (Drb_Fast_Chain const * chain)
{
    return split_set_words(chain->blocks, chain->chunks);
}

extern "C" int drb_hip_launch_collapse_build
    (
        Drb_Fast_Chain const * chain, Drb_Collapse_Job const * jobs, int job_count,
        unsigned int * sets, int * fault_flag, void * stream
    )
{
    if (job_count <= 0) return 0;
    if (chain->kind != 4 || chain->blocks < 1 || chain->blocks > drb_split_max_blocks) return hipErrorInvalidValue;

    size_t const dynamic = collapse_build_lds_bytes(chain);

    if (dynamic > kBuildLdsLimit) return hipErrorInvalidValue;

    hipError_t const status = hipFuncSetAttribute
    (
        reinterpret_cast<void const *>(drb_collapse_build_kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(dynamic)
    );

    if (status != hipSuccess) return status;

    drb_collapse_build_kernel<<<dim3(static_cast<unsigned>(job_count), static_cast<unsigned>(chain->blocks), 2), dim3(1024), dynamic, static_cast<hipStream_t>(stream)>>>
    (
        *chain, jobs, sets, drb_hip_collapse_set_words(chain), fault_flag
    );

    return hipGetLastError();
}

// Many streams of a few tiles each, all of as many tiles: as many workgroups as the
// chip holds at a time, each of which goes from stream to stream (a workgroup's
// set-up and the latency of its first loads then come once per workgroup, not
// once per tile or two).  How many such workgroups a launch gets (0: every
// stream and channel gets workgroups of its own; below 0: a HIP call failed),
// and their kernel.
static long long split_walkers (Drb_Fast_Chain const * chain, int stream_count, int max_tiles, int min_tiles, Split_Kernel ** kernel_out)
{
    // (a step of long streams does not so much as look the other kernel up)
    bool const may_walk = min_tiles == max_tiles && (chain->unit_walk > 0 || (chain->unit_walk == 0 && max_tiles <= kSplitWalkTiles));
    Split_Kernel * const walking = may_walk ? pick_split(chain, true) : nullptr;

    if (kernel_out) *kernel_out = walking;
    if (!walking) return 0;

    size_t const bytes = static_cast<size_t>(chain->lds_floats) * sizeof(unsigned);
    int const threads = 64 * (chain->blocks + kSplitLoaders);
    long long const pairs = static_cast<long long>(stream_count) * (chain->layout == kInterleaved && chain->channels == 2 ? 1 : chain->channels);
    long long walkers = chain->unit_walk > 0 ? chain->unit_walk : 0;

    if (walkers == 0)
    {
        // how many of the kernel's workgroups the chip holds: asked once per kernel and size
        struct Held { Split_Kernel * kernel; size_t bytes; int device; long long workgroups; };
        static Held held [8] = {};
        static int next = 0;
        static std::mutex held_lock;
        std::lock_guard<std::mutex> const guard(held_lock);
        int device = 0;

        if (hipGetDevice(&device) != hipSuccess) return -1;

        for (Held const & one : held) if (one.kernel == walking && one.bytes == bytes && one.device == device) walkers = one.workgroups;

        if (walkers == 0)
        {
            int per_cu = 0, cus = 0;

            if (bytes > 48 * 1024
                && hipFuncSetAttribute(reinterpret_cast<void const *>(walking), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)) != hipSuccess) return -1;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<void const *>(walking), threads, bytes) != hipSuccess) return -1;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return -1;

            walkers = static_cast<long long>(per_cu) * cus;
            held[next] = Held{ walking, bytes, device, walkers };
            next = (next + 1) % 8;
        }
    }

    return walkers >= pairs ? 0 : walkers;   // (a workgroup per unit: nothing to walk)
}

extern "C" int drb_hip_collapsed_walkers (Drb_Fast_Chain const * chain, int stream_count, int max_tiles, int min_tiles)
{
    long long const walkers = chain->kind == 4 && pick_split(chain) ? split_walkers(chain, stream_count, max_tiles, min_tiles, nullptr) : 0;

    return walkers > 0 ? static_cast<int>(walkers) : 0;
}

extern "C" int drb_hip_launch_collapsed
    (
        Drb_Fast_Chain const * chain,
        Drb_Device_Call const * calls, int stream_count, int max_tiles, int min_tiles,
        float const * history,
        long long source_channel_stride, long long target_channel_stride,
        Drb_Collapse_Job const * jobs, Drb_Collapse_Spot const * spots, int max_stretches,
        unsigned int const * sets,
        int * fault_flag, void * stream, int * walkers_out
    )
{
    if (walkers_out) *walkers_out = 0;

    if (stream_count <= 0 || max_tiles <= 0) return 0;

    if (chain->kind != 4) return hipErrorInvalidValue;

    Split_Kernel * split = pick_split(chain);

    if (!split || max_stretches < 1) return hipErrorInvalidValue;

    size_t const bytes = static_cast<size_t>(chain->lds_floats) * sizeof(unsigned);
    unsigned const threads = static_cast<unsigned>(64 * (chain->blocks + kSplitLoaders));

    // one workgroup fills a CU; a workgroup walks over the tiles of its stream
    // and channel (or channel pair) with a stride, a few workgroups per CU in all
    long long const pairs = static_cast<long long>(stream_count) * (chain->layout == kInterleaved && chain->channels == 2 ? 1 : chain->channels);
    long long stride = (1024 + pairs - 1) / pairs;

    if (chain->tile_stride > 0) stride = chain->tile_stride;
    if (stride > max_tiles) stride = max_tiles;
    if (stride < 1) stride = 1;
    if (pairs * stride > 0x7fffffffLL) return hipErrorInvalidValue;

    Split_Kernel * walking = nullptr;
    long long const walkers = split_walkers(chain, stream_count, max_tiles, min_tiles, &walking);

    if (walkers < 0) return hipErrorInvalidValue;

    if (bytes > 48 * 1024)
    {
        hipError_t const status = hipFuncSetAttribute(reinterpret_cast<void const *>(walkers > 0 ? walking : split),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));

        if (status != hipSuccess) return status;
    }

    if (walkers_out) *walkers_out = static_cast<int>(walkers);

    if (walkers > 0)
    {
        walking<<<dim3(static_cast<unsigned>(walkers)), dim3(threads), bytes, static_cast<hipStream_t>(stream)>>>
        (
            *chain, calls, stream_count, 1, static_cast<int>(walkers), history,
            source_channel_stride, target_channel_stride, jobs, spots, max_stretches, sets, fault_flag
        );
    }
    else
    {
        split<<<dim3(static_cast<unsigned>(pairs * stride)), dim3(threads), bytes, static_cast<hipStream_t>(stream)>>>
        (
            *chain, calls, stream_count, static_cast<int>(stride), static_cast<int>(pairs), history,
            source_channel_stride, target_channel_stride, jobs, spots, max_stretches, sets, fault_flag
        );
    }

    return hipGetLastError();
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

    if (chain->tile_stride > 0) tile_stride = chain->tile_stride;
    if (tile_stride > max_tiles) tile_stride = max_tiles;
    if (tile_stride < 1) tile_stride = 1;

    long long const blocks = pairs * tile_stride;

    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;

    kernel<<<dim3(static_cast<unsigned>(blocks)), dim3(static_cast<unsigned>(64 * chain->waves)), dynamic, static_cast<hipStream_t>(stream)>>>
    (
        *chain, calls, stream_count, static_cast<int>(tile_stride), history,
        source_channel_stride, target_channel_stride, fault_flag
    );

    return hipGetLastError();
}
