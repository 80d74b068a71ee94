// gfx950 kernels for the resampling cascade, and the thin C ABI the host code
// (plain C) uses to reach HIP.
//
// Execution model.  One workgroup produces one tile of `tile_frames` target
// frames of one stream, all channels.  It first walks the chain backwards to
// find which index range it needs on every edge between stages, then walks
// forwards: load the source range (history + new chunk) into LDS, run every
// stage LDS -> LDS, store the tile.  Nothing but the source chunk, the
// per-stream history and the target tile touches HBM.  Stages are pure
// functions of the frame index (zero input before the start of the stream),
// so the halo a tile recomputes is bit-identical to what its neighbour
// computes and a stream may be cut into tiles and steps anywhere.
//
// The "exact" arithmetic below reproduces the upstream scalar code path bit
// for bit: float32 data, every product and every sum rounded separately, taps
// accumulated in ascending order (drb-audio-converter.c:615-798).

#include <hip/hip_runtime.h>

#include <atomic>
#include <type_traits>
#include <cstdlib>
#include <mutex>

#include "drb_hip.h"

namespace
{

constexpr int kThreads = 512;

enum { kStageUp2x = 1, kStageDown2x = 2, kStageSinc = 3 };
enum { kSampleF32 = 0, kSampleI16 = 1, kSampleI32 = 2 };
enum { kPlanar = 0, kInterleaved = 1 };

// The coefficient tables (data/drb_tables.bin) as bit patterns.
__constant__ unsigned int drb_table_words [] =
{
#include "../../data/drb_tables_words.inc"
};

__device__ inline float const * table_at (int float_offset)
{
    return reinterpret_cast<float const *>(drb_table_words) + float_offset;
}

// One rounding per operation, whatever the command line says about contraction: from
// here to the end of the file no product is fused with a sum (the pragma takes the
// `contract` flag off every operation compiled below it, the helpers' included; the
// Makefile passes -ffp-contract=off as well).  The packed two-channel arithmetic of the
// fractional stage is plain vector code under the same rule.
#pragma clang fp contract(off)

typedef float float2_t __attribute__((ext_vector_type(2)));

__device__ inline float mul (float a, float b) { return a * b; }
__device__ inline float add (float a, float b) { return a + b; }
__device__ inline float mac (float acc, float a, float b) { float const product = a * b; return acc + product; }

__device__ inline int floor_half (int v) { return v >> 1; }

// ---- sample load / store with type conversion ------------------------------

// (the callers' buffers are device memory: pointers that say so become global
// instead of flat loads and stores)
#define DRB_GLOBAL __attribute__((address_space(1)))

template <typename T>
__device__ __forceinline__ DRB_GLOBAL T * global_pointer (void const * address)
{
    return reinterpret_cast<DRB_GLOBAL T *>(reinterpret_cast<unsigned long long>(address));
}

__device__ inline float load_sample (void const * base, long long index, int type)
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

__device__ inline void store_sample (void * base, long long index, int type, float value)
{
    switch (type)
    {
      case kSampleI16:
      {
        float const scaled = rintf(mul(value, 32768.0f));
        int const q = scaled >= 32767.0f ? 32767
                    : scaled <= -32768.0f ? -32768
                    : scaled != scaled ? 0 : static_cast<int>(scaled);
        global_pointer<short>(base)[index] = static_cast<short>(q);
        break;
      }
      case kSampleI32:
      {
        float const scaled = rintf(mul(value, 2147483648.0f));
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

// Source frame `frame` of channel `channel` of one stream, where negative
// frames come from the stream's history and frames before that are silence.
__device__ inline float source_frame
    (
        Drb_Device_Chain const & chain, Drb_Device_Call const & call,
        float const * history, long long channel_stride, int channel, int frame
    )
{
    if (frame < 0)
    {
        return frame >= -chain.history ? history[channel * chain.history + chain.history + frame] : 0.0f;
    }

    if (frame >= call.source_frames) return 0.0f;

    long long const index = chain.layout == kInterleaved
        ? static_cast<long long>(frame) * chain.channels + channel
        : channel * channel_stride + frame;

    return load_sample(reinterpret_cast<void const *>(call.source), index, chain.sample_type);
}

// ---- fractional stage position ---------------------------------------------

// For accumulator value `phase` (drb-audio-converter.c:1617-1648 push,
// 1677-1707 pull): the number of source frames consumed before this output
// (relative to the step) and the sub-sample position in [0, 1).
//   push: m = ceil(phase / Sd) - 1, blend = (phase - m Sd - 1) / Sd
//   pull: m = floor(phase / Sd),    blend = (phase - m Sd) / Sd
__device__ inline void sinc_position
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

// ---- edge buffers -----------------------------------------------------------

// An edge buffer holds, per channel, `width` floats.  Position 0 is the lowest
// index the consumer of the edge reads.  A decimator reads every other sample,
// so the edge in front of one is stored as two planes (even positions, then
// odd positions) to keep its reads contiguous.
__device__ inline void put (float * channel_base, int width, int split, int position, float value)
{
    if (position < 0 || position >= width) return;

    int const at = split ? (position & 1) * (width >> 1) + (position >> 1) : position;

    channel_base[at] = value;
}

struct Ranges
{
    int lo [drb_device_max_stages + 1];  // consumer index stored at position 0, per edge
    int hi [drb_device_max_stages + 1];
    int fault;
};

// ---- half-band 2x up-sampler -----------------------------------------------
// y[2f] = (x[f - N/2] * 0.5) * 2 ; y[2f+1] = 2 * sum_j k[j] x[f - N + 1 + j]
// (drb-audio-converter.c:1984-2004).  Position 0 of the input edge is
// x[first_frame - N + 1].  Each work item produces four consecutive source
// frames' worth of output for one channel from one aligned window in registers.
template <int N>
__device__ __forceinline__ void stage_up2x
    (
        float const * in, float * out, int width, int channels, int out_split,
        int frames, int out_offset, float const * kernel
    )
{
    int const groups = (frames + 3) >> 2;

    for (int item = threadIdx.x; item < groups * channels; item += kThreads)
    {
        int const channel = item / groups, group = item - channel * groups;
        float const * const src = in + channel * width + 4 * group;

        float window [N + 4];

        #pragma unroll
        for (int chunk = 0; chunk < N / 4 + 1; chunk++)
        {
            float4 const v = *reinterpret_cast<float4 const *>(src + 4 * chunk);
            window[4 * chunk + 0] = v.x;
            window[4 * chunk + 1] = v.y;
            window[4 * chunk + 2] = v.z;
            window[4 * chunk + 3] = v.w;
        }

        float * const dst = out + channel * width;

        #pragma unroll
        for (int r = 0; r < 4; r++)
        {
            float acc = 0.0f;

            #pragma unroll
            for (int j = 0; j < N; j++) acc = mac(acc, kernel[j], window[r + j]);

            float const even = mul(mul(window[r + N / 2 - 1], 0.5f), 2.0f);
            float const odd = mul(acc, 2.0f);

            if (4 * group + r < frames)
            {
                int const position = 2 * (4 * group + r) + out_offset;
                put(dst, width, out_split, position, even);
                put(dst, width, out_split, position + 1, odd);
            }
        }
    }
}

// ---- half-band 2x down-sampler ---------------------------------------------
// y[f] = 0.5 x[2f - N] + sum_j k[j] x[2 (f - N + j) + 1]
// (drb-audio-converter.c:2033-2057).  Position 0 of the (two-plane) input edge
// is x[2 first_frame - 2N + 1]; the taps therefore sit in plane 0 and the
// centre sample in plane 1.
template <int N>
__device__ __forceinline__ void stage_down2x
    (
        float const * in, float * out, int width, int channels, int out_split,
        int frames, float const * kernel
    )
{
    int const groups = (frames + 3) >> 2;
    int const plane = width >> 1;
    constexpr int kCentre = N / 2 - 1;

    for (int item = threadIdx.x; item < groups * channels; item += kThreads)
    {
        int const channel = item / groups, group = item - channel * groups;
        float const * const taps = in + channel * width + 4 * group;
        float const * const centre = in + channel * width + plane + 4 * group + (kCentre & ~3);

        float window [N + 4];
        float middle [8];

        #pragma unroll
        for (int chunk = 0; chunk < N / 4 + 1; chunk++)
        {
            float4 const v = *reinterpret_cast<float4 const *>(taps + 4 * chunk);
            window[4 * chunk + 0] = v.x;
            window[4 * chunk + 1] = v.y;
            window[4 * chunk + 2] = v.z;
            window[4 * chunk + 3] = v.w;
        }

        #pragma unroll
        for (int chunk = 0; chunk < 2; chunk++)
        {
            float4 const v = *reinterpret_cast<float4 const *>(centre + 4 * chunk);
            middle[4 * chunk + 0] = v.x;
            middle[4 * chunk + 1] = v.y;
            middle[4 * chunk + 2] = v.z;
            middle[4 * chunk + 3] = v.w;
        }

        float * const dst = out + channel * width;

        #pragma unroll
        for (int r = 0; r < 4; r++)
        {
            float acc = 0.0f;

            #pragma unroll
            for (int j = 0; j < N; j++) acc = mac(acc, kernel[j], window[r + j]);

            float const value = add(mul(middle[r + (kCentre & 3)], 0.5f), acc);

            if (4 * group + r < frames) put(dst, width, out_split, 4 * group + r, value);
        }
    }
}

// ---- windowed-sinc fractional stage -----------------------------------------
// One work item = one output frame, all channels two at a time.  The bank rows
// (64 x TAPS linear, 32 x TAPS cubic) sit in LDS with a one-float skew per row
// so that lanes reading different rows at the same tap hit different banks.
// Row pick and blend: drb-audio-converter.c:881-941; linear blend 827-848;
// Catmull-Rom 850-875.

// A position on the fractional stage's time line: whole input frames q and the rest r of
// the accumulator, 0 <= r < source_delta (accumulator = q source_delta + r).
struct Sinc_At
{
    long long q, r;
};

// the position of accumulator value `phase` (the floor division sinc_position makes)
__device__ inline Sinc_At sinc_at (long long phase, long long sd, double scale)
{
    Sinc_At at;

    at.q = static_cast<long long>(floor(static_cast<double>(phase) * scale));
    at.r = phase - at.q * sd;

    while (at.r < 0) { at.q -= 1; at.r += sd; }
    while (at.r >= sd) { at.q += 1; at.r -= sd; }

    return at;
}

// `by` further on (by.r < source_delta): whole numbers throughout, so a position reached in
// steps is the position the division finds
__device__ __forceinline__ void sinc_advance (Sinc_At & at, Sinc_At const & by, long long sd)
{
    at.q += by.q;
    at.r += by.r;

    if (at.r >= sd) { at.r -= sd; at.q += 1; }
}

// where the output at position `at` reads: its bank row, the position inside the row's
// interval and the first input position (relative to the edge buffer's origin); the
// phase conventions are sinc_position's
template <int TAPS, int CUBIC>
__device__ __forceinline__ void sinc_frame
    (
        Drb_Device_Chain const & chain, Sinc_At const & at, int in_origin,
        int & row, float & sub, int & start
    )
{
    long long q = at.q, r = at.r;
    double blend;

    if (chain.sinc_pull)
    {
        blend = static_cast<double>(r) * chain.scale;
    }
    else
    {
        if (r == 0) { q -= 1; r = chain.source_delta; }
        blend = static_cast<double>(r - 1) * chain.scale;
    }

    double const scaled = blend * (CUBIC ? 29.0 : 63.0);

    row = static_cast<int>(scaled);
    sub = static_cast<float>(scaled - static_cast<double>(row));
    start = static_cast<int>(q) - TAPS - in_origin;
}

// the rows' sums of one frame and channel pair into the frame's two samples
// (PAIR false: the sums' first halves only, a lone channel's)
template <int CUBIC, bool PAIR = true>
__device__ __forceinline__ void sinc_blend (float2_t const * p, float sub, float & ya, float & yb)
{
    constexpr int kRows = CUBIC ? 4 : 2;
    float pa [kRows], pb [kRows];

    #pragma unroll
    for (int r = 0; r < kRows; r++) { pa[r] = p[r][0]; pb[r] = p[r][1]; }

    if (CUBIC)
    {
        // a_i = sum_j M[i][j] p[j], accumulated from zero in order
        float const m [4][4] =
        {
            {  0.0f,  1.0f,  0.0f,  0.0f },
            { -0.5f,  0.0f,  0.5f,  0.0f },
            {  1.0f, -2.5f,  2.0f, -0.5f },
            { -0.5f,  1.5f, -1.5f,  0.5f }
        };
        float ca [4], cb [4];

        #pragma unroll
        for (int i = 0; i < 4; i++)
        {
            float sa = 0.0f, sb = 0.0f;

            #pragma unroll
            for (int j = 0; j < 4; j++)
            {
                sa = mac(sa, m[i][j], pa[j]);
                if (PAIR) sb = mac(sb, m[i][j], pb[j]);
            }

            ca[i] = sa;
            cb[i] = sb;
        }

        ya = add(mul(add(mul(add(mul(ca[3], sub), ca[2]), sub), ca[1]), sub), ca[0]);
        yb = PAIR ? add(mul(add(mul(add(mul(cb[3], sub), cb[2]), sub), cb[1]), sub), cb[0]) : ya;
    }
    else
    {
        float const a0 = mac(mac(0.0f, 1.0f, pa[0]), 0.0f, pa[1]);
        float const a1 = mac(mac(0.0f, -1.0f, pa[0]), 1.0f, pa[1]);
        ya = add(mul(a1, sub), a0);

        if (PAIR)
        {
            float const b0 = mac(mac(0.0f, 1.0f, pb[0]), 0.0f, pb[1]);
            float const b1 = mac(mac(0.0f, -1.0f, pb[0]), 1.0f, pb[1]);

            yb = add(mul(b1, sub), b0);
        }
        else yb = ya;
    }
}

// one output frame (`item` of the stage's outputs, at position `at`), all channels
// (LONE: the workgroup's channel count may be odd, and its last channel then goes alone, in
// scalar registers: a packed instruction with one half idle would issue at half the rate
// for nothing.  LONE 0: the image for even counts; 1: odd counts; 2: one channel a workgroup.)
template <int TAPS, int CUBIC, int LONE>
__device__ __forceinline__ void sinc_one
    (
        Drb_Device_Chain const & chain, Sinc_At const & at,
        float const * in, float * out, int width, int channels, int out_split,
        int item, int in_origin, float const * bank
    )
{
    // (p + x * k below must stay a rounded product and a rounded sum, as upstream's)
    #pragma clang fp contract(off)

    constexpr int kRows = CUBIC ? 4 : 2;
    constexpr int kStride = TAPS + 1;

    int row, start;
    float sub;

    sinc_frame<TAPS, CUBIC>(chain, at, in_origin, row, sub, start);

    float const * const rows = bank + row * kStride;

    if constexpr (!LONE)
    {
        for (int channel = 0; channel < channels; channel += 2)
        {
            bool const pair = channel + 1 < channels;
            float const * const a = in + channel * width + start;
            float const * const b = pair ? a + width : a;

            // Two channels side by side in packed registers: one v_pk_mul_f32 and one
            // v_pk_add_f32 per tap and row, each half rounded as the scalar
            // instructions round, in the same order.
            float2_t p [kRows];

            #pragma unroll
            for (int r = 0; r < kRows; r++) p[r] = float2_t{ 0.0f, 0.0f };

            #pragma unroll
            for (int t = 0; t < TAPS; t++)
            {
                float2_t const x = { a[t], b[t] };

                #pragma unroll
                for (int r = 0; r < kRows; r++)
                {
                    float const k = rows[r * kStride + t];
                    p[r] = p[r] + x * float2_t{ k, k };
                }
            }

            float ya, yb;

            sinc_blend<CUBIC>(p, sub, ya, yb);

            put(out + channel * width, width, out_split, item, ya);
            if (pair) put(out + (channel + 1) * width, width, out_split, item, yb);
        }
    }
    else
    {
        // Two channels side by side in packed registers: one v_pk_mul_f32 and one
        // v_pk_add_f32 per tap and row, each half rounded as the scalar instructions
        // round, in the same order.  A lone channel (the last of an odd count) in scalar
        // registers: a packed instruction with one half idle would issue at half the rate
        // for nothing.
        auto channels_at = [&] (int channel, auto paired) __attribute__((always_inline))
        {
            constexpr bool kPair = decltype(paired)::value;
            typedef typename std::conditional<kPair, float2_t, float>::type sums_t;

            float const * const a = in + channel * width + start;
            float const * const b = a + width;

            sums_t p [kRows];

            #pragma unroll
            for (int r = 0; r < kRows; r++) p[r] = sums_t{};

            #pragma unroll
            for (int t = 0; t < TAPS; t++)
            {
                #pragma unroll
                for (int r = 0; r < kRows; r++)
                {
                    float const k = rows[r * kStride + t];

                    if constexpr (kPair) p[r] = p[r] + float2_t{ a[t], b[t] } * float2_t{ k, k };
                    else p[r] = p[r] + a[t] * k;
                }
            }

            float2_t both [kRows];

            #pragma unroll
            for (int r = 0; r < kRows; r++)
            {
                if constexpr (kPair) both[r] = p[r];
                else both[r] = float2_t{ p[r], p[r] };
            }

            float ya, yb;

            sinc_blend<CUBIC, kPair>(both, sub, ya, yb);

            put(out + channel * width, width, out_split, item, ya);
            if (kPair) put(out + (channel + 1) * width, width, out_split, item, yb);
        };

        if constexpr (LONE == 2)
        {
            channels_at(0, std::false_type{});
        }
        else
        {
            for (int channel = 0; channel < channels; channel += 2)
            {
                if (channel + 1 < channels) channels_at(channel, std::true_type{});
                else channels_at(channel, std::false_type{});
            }
        }
    }
}

// Outputs a whole number of the rate ratio's periods apart sit at (nearly) the same
// fractional position: the same but for the slow drift of the fixed-point phase.  A work
// item takes kShare such frames (item i: outputs i, i + part, ..., `part` a whole number
// of periods), and where all of them pick the same bank row, as they do but for a frame
// in a million, every coefficient is read from LDS once for the kShare frames.  Each
// frame's sums are formed as sinc_one forms them.  What is left past kShare parts, and
// any stage with fewer outputs than kShare periods, goes frame by frame.
// A thread divides once (sinc_at) and steps from output to output in whole numbers.
constexpr int kShare = 2;

template <int TAPS, int CUBIC, int LONE>
__device__ __forceinline__ void stage_sinc
    (
        Drb_Device_Chain const & chain, long long phase_base,
        float const * in, float * out, int width, int channels, int out_split,
        int first_output, int outputs, int in_origin, float const * bank
    )
{
    #pragma clang fp contract(off)

    constexpr int kRows = CUBIC ? 4 : 2;
    constexpr int kStride = TAPS + 1;

    long long const sd = chain.source_delta;
    int const period = chain.sinc_period;
    int const part = period > 0 ? outputs / (kShare * period) * period : 0;

    // a thread's next output but one, and a frame's partner
    Sinc_At const stride = sinc_at(static_cast<long long>(kThreads) * chain.target_delta, sd, chain.scale);
    Sinc_At const apart = sinc_at(static_cast<long long>(part) * chain.target_delta, sd, chain.scale);

    if (static_cast<int>(threadIdx.x) < part)
    {
        Sinc_At at = sinc_at(phase_base + static_cast<long long>(first_output + static_cast<int>(threadIdx.x)) * chain.target_delta, sd, chain.scale);

        for (int item = threadIdx.x; item < part; item += kThreads, sinc_advance(at, stride, sd))
        {
            Sinc_At frame [kShare];
            int row [kShare], start [kShare];
            float sub [kShare];

            frame[0] = at;

            #pragma unroll
            for (int m = 1; m < kShare; m++) { frame[m] = frame[m - 1]; sinc_advance(frame[m], apart, sd); }

            #pragma unroll
            for (int m = 0; m < kShare; m++) sinc_frame<TAPS, CUBIC>(chain, frame[m], in_origin, row[m], sub[m], start[m]);

            // (ablate bit 32, a test's: every item takes the way of frames whose rows differ)
            bool same = !(chain.ablate & 32);

            #pragma unroll
            for (int m = 1; m < kShare; m++) same = same && row[m] == row[0];

            if (!same)
            {
                #pragma unroll
                for (int m = 0; m < kShare; m++)
                {
                    sinc_one<TAPS, CUBIC, LONE>(chain, frame[m], in, out, width, channels, out_split, item + m * part, in_origin, bank);
                }

                continue;
            }

            float const * const rows = bank + row[0] * kStride;

            if constexpr (!LONE)
            {
                for (int channel = 0; channel < channels; channel += 2)
                {
                    bool const pair = channel + 1 < channels;
                    float const * const a = in + channel * width;
                    float const * const b = pair ? a + width : a;

                    float2_t p [kShare][kRows];

                    #pragma unroll
                    for (int m = 0; m < kShare; m++)
                    {
                        #pragma unroll
                        for (int r = 0; r < kRows; r++) p[m][r] = float2_t{ 0.0f, 0.0f };
                    }

                    // (four taps at a time: unrolled further, the compiler starts every read of the
                    // stage at once and runs out of registers)
                    #pragma unroll 4
                    for (int t = 0; t < TAPS; t++)
                    {
                        float k [kRows];

                        #pragma unroll
                        for (int r = 0; r < kRows; r++) k[r] = rows[r * kStride + t];

                        #pragma unroll
                        for (int m = 0; m < kShare; m++)
                        {
                            float2_t const x = { a[start[m] + t], b[start[m] + t] };

                            #pragma unroll
                            for (int r = 0; r < kRows; r++) p[m][r] = p[m][r] + x * float2_t{ k[r], k[r] };
                        }
                    }

                    #pragma unroll
                    for (int m = 0; m < kShare; m++)
                    {
                        float ya, yb;

                        sinc_blend<CUBIC>(p[m], sub[m], ya, yb);

                        put(out + channel * width, width, out_split, item + m * part, ya);
                        if (pair) put(out + (channel + 1) * width, width, out_split, item + m * part, yb);
                    }
                }
            }
            else
            {
                auto channels_at = [&] (int channel, auto paired) __attribute__((always_inline))
                {
                    constexpr bool kPair = decltype(paired)::value;
                    typedef typename std::conditional<kPair, float2_t, float>::type sums_t;

                    float const * const a = in + channel * width;
                    float const * const b = a + width;

                    sums_t p [kShare][kRows];

                    #pragma unroll
                    for (int m = 0; m < kShare; m++)
                    {
                        #pragma unroll
                        for (int r = 0; r < kRows; r++) p[m][r] = sums_t{};
                    }

                    // (four taps at a time: unrolled further, the compiler starts every read of the
                    // stage at once and runs out of registers)
                    #pragma unroll 4
                    for (int t = 0; t < TAPS; t++)
                    {
                        float k [kRows];

                        #pragma unroll
                        for (int r = 0; r < kRows; r++) k[r] = rows[r * kStride + t];

                        #pragma unroll
                        for (int m = 0; m < kShare; m++)
                        {
                            #pragma unroll
                            for (int r = 0; r < kRows; r++)
                            {
                                if constexpr (kPair) p[m][r] = p[m][r] + float2_t{ a[start[m] + t], b[start[m] + t] } * float2_t{ k[r], k[r] };
                                else p[m][r] = p[m][r] + a[start[m] + t] * k[r];
                            }
                        }
                    }

                    #pragma unroll
                    for (int m = 0; m < kShare; m++)
                    {
                        float2_t both [kRows];

                        #pragma unroll
                        for (int r = 0; r < kRows; r++)
                        {
                            if constexpr (kPair) both[r] = p[m][r];
                            else both[r] = float2_t{ p[m][r], p[m][r] };
                        }

                        float ya, yb;

                        sinc_blend<CUBIC, kPair>(both, sub[m], ya, yb);

                        put(out + channel * width, width, out_split, item + m * part, ya);
                        if (kPair) put(out + (channel + 1) * width, width, out_split, item + m * part, yb);
                    }
                };

                if constexpr (LONE == 2)
                {
                    channels_at(0, std::false_type{});
                }
                else
                {
                    for (int channel = 0; channel < channels; channel += 2)
                    {
                        if (channel + 1 < channels) channels_at(channel, std::true_type{});
                        else channels_at(channel, std::false_type{});
                    }
                }
            }
        }
    }

    // the rest frame by frame, the last threads first (the first ones had one item more)
    int const next = kShare * part + kThreads - 1 - static_cast<int>(threadIdx.x);

    if (next < outputs)
    {
        Sinc_At at = sinc_at(phase_base + static_cast<long long>(first_output + next) * chain.target_delta, sd, chain.scale);

        for (int item = next; item < outputs; item += kThreads, sinc_advance(at, stride, sd))
        {
            sinc_one<TAPS, CUBIC, LONE>(chain, at, in, out, width, channels, out_split, item, in_origin, bank);
        }
    }
}

template <int N>
__device__ __forceinline__ void run_half_band
    (
        int kind, float const * in, float * out, int width, int channels, int out_split,
        int frames, int out_offset, float const * kernel
    )
{
    if (kind == kStageUp2x) stage_up2x<N>(in, out, width, channels, out_split, frames, out_offset, kernel);
    else stage_down2x<N>(in, out, width, channels, out_split, frames, kernel);
}

// ---- the cascade kernel -----------------------------------------------------

template <int LONE>
__global__ __launch_bounds__(kThreads, 6) void drb_cascade_kernel
    (
        Drb_Device_Chain const chain,
        Drb_Device_Call const * __restrict__ calls, int stream_count,
        float const * __restrict__ history,
        long long source_channel_stride, long long target_channel_stride,
        int bank_floats, int * fault_flag
    )
{
    // All LDS is one dynamic array (staging buffers, coefficient bank, ranges):
    // a static __shared__ object in front of it would misalign the float4 reads.
    extern __shared__ __attribute__((aligned(16))) float lds [];

    int const stream = blockIdx.x % stream_count;
    int const tile = blockIdx.x / stream_count;
    Drb_Device_Call const call = calls[stream];

    int const first = tile * chain.tile_frames;

    if (first >= call.target_frames) return;

    int const last = min(first + chain.tile_frames, call.target_frames);
    int const stages = chain.stage_count;
    int const width = chain.lds_width;
    // this workgroup's channels: channel_0 .. channel_0 + channels - 1 of the stream's
    int const channel_0 = blockIdx.y * chain.channel_group;
    int const channels = min(chain.channel_group, chain.channels - channel_0);

    // (offsets from the one LDS array, not an array of pointers: the compiler then
    // keeps every access an LDS instruction instead of a flat one)
    int const buffer_floats = chain.channel_group * width;
    float * const bank = lds + 2 * buffer_floats;
    Ranges & ranges = *reinterpret_cast<Ranges *>(bank + bank_floats);

    // -- backwards: index range needed on every edge --------------------------
    if (threadIdx.x == 0)
    {
        ranges.fault = 0;
        ranges.lo[stages] = first;
        ranges.hi[stages] = last;

        for (int k = stages - 1; k >= 0; k--)
        {
            // what stage k must produce, in its own output indexing
            int const shift = chain.delay_edge == k + 1 ? call.delay : 0;
            int const lo = ranges.lo[k + 1] - shift, hi = ranges.hi[k + 1] - shift;
            Drb_Device_Stage const stage = chain.stages[k];

            if (stage.kind == kStageUp2x)
            {
                ranges.lo[k] = floor_half(lo) - stage.taps + 1;
                ranges.hi[k] = floor_half(hi - 1) + 1;
            }
            else if (stage.kind == kStageDown2x)
            {
                ranges.lo[k] = 2 * lo - 2 * stage.taps + 1;
                ranges.hi[k] = 2 * hi - 2;
            }
            else
            {
                int consumed_lo, consumed_hi;
                double unused;

                sinc_position(call.phase + static_cast<long long>(lo) * chain.target_delta,
                              chain.source_delta, chain.scale, chain.sinc_pull, consumed_lo, unused);
                sinc_position(call.phase + static_cast<long long>(hi - 1) * chain.target_delta,
                              chain.source_delta, chain.scale, chain.sinc_pull, consumed_hi, unused);

                ranges.lo[k] = consumed_lo - stage.taps;
                ranges.hi[k] = consumed_hi;
            }
        }

        for (int e = 0; e <= stages; e++)
        {
            // every stage may read up to 15 floats past its range (discarded work items)
            if (ranges.hi[e] - ranges.lo[e] + 16 > width) ranges.fault = 1;
        }

        int const source_shift = chain.delay_edge == 0 ? call.delay : 0;

        if (ranges.lo[0] - source_shift < -chain.history) ranges.fault = 2;
        if (ranges.hi[0] - source_shift > call.source_frames) ranges.fault = 3;

        if (ranges.fault) atomicOr(fault_flag, ranges.fault);
    }

    // the fractional stage's coefficient bank, rows skewed by one float
    if (bank_floats > 0)
    {
        for (int k = 0; k < stages; k++)
        {
            if (chain.stages[k].kind != kStageSinc) continue;

            int const taps = chain.stages[k].taps;
            int const rows = chain.stages[k].cubic ? 32 : 64;
            float const * const source = table_at(chain.stages[k].table);

            for (int i = threadIdx.x; i < rows * taps; i += kThreads)
            {
                bank[(i / taps) * (taps + 1) + i % taps] = source[i];
            }
        }
    }

    __syncthreads();

    if (ranges.fault) return;

    // -- load the source range -------------------------------------------------
    if (!(chain.ablate & 8))
    {
        int const lo = ranges.lo[0], count = ranges.hi[0] - ranges.lo[0];
        int const shift = chain.delay_edge == 0 ? call.delay : 0;
        int const split = stages > 0 && chain.stages[0].kind == kStageDown2x;
        float const * const stream_history = history + static_cast<long long>(stream) * chain.channels * chain.history;

        if (chain.layout == kInterleaved)
        {
            for (int i = threadIdx.x; i < count * channels; i += kThreads)
            {
                int const position = i / channels, channel = i - position * channels;
                float const v = source_frame(chain, call, stream_history, source_channel_stride,
                                             channel_0 + channel, lo + position - shift);
                put(lds + channel * width, width, split, position, v);
            }
        }
        else
        {
            for (int channel = 0; channel < channels; channel++)
            {
                for (int position = threadIdx.x; position < count; position += kThreads)
                {
                    float const v = source_frame(chain, call, stream_history, source_channel_stride,
                                                 channel_0 + channel, lo + position - shift);
                    put(lds + channel * width, width, split, position, v);
                }
            }
        }
    }

    __syncthreads();

    // -- forwards: every stage LDS -> LDS --------------------------------------
    for (int k = 0; k < stages; k++)
    {
        Drb_Device_Stage const stage = chain.stages[k];
        float const * const in = lds + (k & 1) * buffer_floats;
        float * const out = lds + ((k + 1) & 1) * buffer_floats;
        int const shift = chain.delay_edge == k + 1 ? call.delay : 0;
        int const lo = ranges.lo[k + 1] - shift, hi = ranges.hi[k + 1] - shift;
        int const out_split = k + 1 < stages && chain.stages[k + 1].kind == kStageDown2x;
        float const * const kernel = table_at(stage.table);

        if (chain.ablate & (stage.kind == kStageSinc ? 1 : stage.kind == kStageUp2x ? 2 : 4))
        {
        }
        else if (stage.kind == kStageSinc)
        {
            if (stage.taps == 32 && !stage.cubic)
                stage_sinc<32, 0, LONE>(chain, call.phase, in, out, width, channels, out_split, lo, hi - lo, ranges.lo[k], bank);
            else if (stage.taps == 32)
                stage_sinc<32, 1, LONE>(chain, call.phase, in, out, width, channels, out_split, lo, hi - lo, ranges.lo[k], bank);
            else if (!stage.cubic)
                stage_sinc<16, 0, LONE>(chain, call.phase, in, out, width, channels, out_split, lo, hi - lo, ranges.lo[k], bank);
            else
                stage_sinc<16, 1, LONE>(chain, call.phase, in, out, width, channels, out_split, lo, hi - lo, ranges.lo[k], bank);
        }
        else
        {
            // up-sampler: source frames floor(lo/2) .. floor((hi-1)/2); its first
            // output lands at position 2 floor(lo/2) - lo (0 or -1).
            int const frames = stage.kind == kStageUp2x
                ? floor_half(hi - 1) + 1 - floor_half(lo) : hi - lo;
            int const out_offset = stage.kind == kStageUp2x ? 2 * floor_half(lo) - lo : 0;

            switch (stage.taps)
            {
              case 8:  run_half_band<8>(stage.kind, in, out, width, channels, out_split, frames, out_offset, kernel); break;
              case 16: run_half_band<16>(stage.kind, in, out, width, channels, out_split, frames, out_offset, kernel); break;
              case 24: run_half_band<24>(stage.kind, in, out, width, channels, out_split, frames, out_offset, kernel); break;
              case 32: run_half_band<32>(stage.kind, in, out, width, channels, out_split, frames, out_offset, kernel); break;
              case 40: run_half_band<40>(stage.kind, in, out, width, channels, out_split, frames, out_offset, kernel); break;
              default: run_half_band<48>(stage.kind, in, out, width, channels, out_split, frames, out_offset, kernel); break;
            }
        }

        __syncthreads();
    }

    // -- store the tile ----------------------------------------------------------
    if (!(chain.ablate & 16))
    {
        float const * const result = lds + (stages & 1) * buffer_floats;
        int const count = last - first;
        void * const target = reinterpret_cast<void *>(call.target);

        if (chain.layout == kInterleaved)
        {
            for (int i = threadIdx.x; i < count * channels; i += kThreads)
            {
                int const position = i / channels, channel = i - position * channels;
                store_sample(target, static_cast<long long>(first + position) * chain.channels + channel_0 + channel,
                             chain.sample_type, result[channel * width + position]);
            }
        }
        else
        {
            for (int channel = 0; channel < channels; channel++)
            {
                for (int position = threadIdx.x; position < count; position += kThreads)
                {
                    store_sample(target, (channel_0 + channel) * target_channel_stride + first + position,
                                 chain.sample_type, result[channel * width + position]);
                }
            }
        }
    }
}

// New look-back per stream: the last `history` frames of (old history ++ chunk).
__global__ __launch_bounds__(kThreads) void drb_history_kernel
    (
        Drb_Device_Chain const chain,
        Drb_Device_Call const * __restrict__ calls,
        float const * __restrict__ history_in, float * __restrict__ history_out,
        long long source_channel_stride
    )
{
    int const stream = blockIdx.x;
    Drb_Device_Call const call = calls[stream];
    long long const base = static_cast<long long>(stream) * chain.channels * chain.history;

    for (int i = threadIdx.x; i < chain.channels * chain.history; i += kThreads)
    {
        int const channel = i / chain.history, slot = i - channel * chain.history;
        int const frame = call.source_frames - chain.history + slot;

        history_out[base + i] = source_frame(chain, call, history_in + base, source_channel_stride, channel, frame);
    }
}

} // namespace

// ---- static workspace ---------------------------------------------------------------
// Device memory of the library's own for callers that may not allocate (see
// drb_amd_batch_process_dense_static): part of the code object, so every GPU
// the library runs on holds its own copy from the first use to the end of the
// process.  `scratch` as written in host code is only a handle for
// hipGetSymbolAddress; what kernels and copies get is the address that call
// returns, looked up once per device with that device current.
constexpr size_t kScratchBytes = static_cast<size_t>(8) << 20;
constexpr int kScratchDevices = 64;

static __device__ __align__(256) unsigned char scratch [kScratchBytes];

static void * scratch_address [kScratchDevices];
// The last step that used a device's workspace: an event recorded behind it (created
// with the address, so that a step creates nothing), and the stream it ran on.  A
// step on another stream waits for that event before it rewrites the workspace.
static hipEvent_t scratch_event [kScratchDevices];
static void * scratch_stream [kScratchDevices];
static bool scratch_used [kScratchDevices];
static std::mutex scratch_lock;

// ---- C ABI --------------------------------------------------------------------

// ---- a launch that only changes the buffer format ------------------------------------------
// Fast arithmetic runs longer chains as several launches with planar float32 between
// them.  Where the first (last) launch's kernel cannot take the caller's buffer
// format, this one stands in front of (behind) it: every sample through
// load_sample / store_sample, the conversions of the exact path, by way of a tile
// in LDS so that both sides are accessed along their own layout.
constexpr int kConvertFrames = 256, kConvertTiles = 4, kConvertChannels = 8;

__global__ __launch_bounds__(256) void drb_convert_kernel
    (
        Drb_Device_Call const * __restrict__ calls, int channels,
        int source_type, int source_layout, long long source_channel_stride,
        int target_type, int target_layout, long long target_channel_stride
    )
{
    __shared__ float tile [kConvertChannels][kConvertFrames + 1];

    Drb_Device_Call const call = calls[blockIdx.x];   // (streams along x: there may be more of them than y takes)
    int const frames = call.target_frames;
    void const * const source = reinterpret_cast<void const *>(call.source);
    void * const target = reinterpret_cast<void *>(call.target);

    for (int t = 0; t < kConvertTiles; t++)
    {
        int const first = (blockIdx.y * kConvertTiles + t) * kConvertFrames;
        int const count = min(kConvertFrames, frames - first);

        if (count <= 0) return;   // (the same for every thread of the workgroup)

        if (source_layout == kInterleaved)
        {
            for (int i = threadIdx.x; i < count * channels; i += blockDim.x)
            {
                int const frame = i / channels, c = i - frame * channels;

                tile[c][frame] = load_sample(source, static_cast<long long>(first) * channels + i, source_type);
            }
        }
        else
        {
            for (int c = 0; c < channels; c++)
            {
                for (int frame = threadIdx.x; frame < count; frame += blockDim.x)
                {
                    tile[c][frame] = load_sample(source, c * source_channel_stride + first + frame, source_type);
                }
            }
        }

        __syncthreads();

        if (target_layout == kInterleaved)
        {
            for (int i = threadIdx.x; i < count * channels; i += blockDim.x)
            {
                int const frame = i / channels, c = i - frame * channels;

                store_sample(target, static_cast<long long>(first) * channels + i, target_type, tile[c][frame]);
            }
        }
        else
        {
            for (int c = 0; c < channels; c++)
            {
                for (int frame = threadIdx.x; frame < count; frame += blockDim.x)
                {
                    store_sample(target, c * target_channel_stride + first + frame, target_type, tile[c][frame]);
                }
            }
        }

        __syncthreads();
    }
}

extern "C" {

// The calling thread's current device must be `device`.
int drb_hip_static_workspace (int device, void ** pointer, size_t * bytes)
{
    if (device < 0 || device >= kScratchDevices) return hipErrorInvalidDevice;

    std::lock_guard<std::mutex> guard(scratch_lock);

    if (!scratch_address[device])
    {
        void * address = nullptr;
        hipError_t status = hipGetSymbolAddress(&address, HIP_SYMBOL(scratch));

        if (status != hipSuccess) return status;
        if (!address) return hipErrorInvalidSymbol;

        status = hipEventCreateWithFlags(&scratch_event[device], hipEventDisableTiming);

        if (status != hipSuccess) return status;

        scratch_address[device] = address;
    }

    *pointer = scratch_address[device];
    *bytes = kScratchBytes;

    return 0;
}

// Brackets of a step that keeps its tables in the workspace.  Every batch of a
// GPU shares the one workspace: `enter` makes `stream` wait for the last such
// step if that ran on another stream (on the same stream the order of the
// launches already keeps them apart), `leave` marks the end of this one.
// Neither allocates.
int drb_hip_static_workspace_enter (int device, void * stream)
{
    if (device < 0 || device >= kScratchDevices || !scratch_address[device]) return hipErrorInvalidDevice;

    std::lock_guard<std::mutex> guard(scratch_lock);

    if (scratch_used[device] && scratch_stream[device] != stream)
    {
        return hipStreamWaitEvent(static_cast<hipStream_t>(stream), scratch_event[device], 0);
    }

    return 0;
}

int drb_hip_static_workspace_leave (int device, void * stream)
{
    if (device < 0 || device >= kScratchDevices || !scratch_address[device]) return hipErrorInvalidDevice;

    std::lock_guard<std::mutex> guard(scratch_lock);

    scratch_used[device] = true;
    scratch_stream[device] = stream;

    return hipEventRecord(scratch_event[device], static_cast<hipStream_t>(stream));
}

int drb_hip_device_count (int * count) { return hipGetDeviceCount(count); }
int drb_hip_get_device (int * device) { return hipGetDevice(device); }
int drb_hip_set_device (int device) { return hipSetDevice(device); }

int drb_hip_device_is_gfx950 (int device, int * yes, int * lds_bytes_per_block)
{
    hipDeviceProp_t properties;
    hipError_t const status = hipGetDeviceProperties(&properties, device);

    if (status != hipSuccess) return status;

    // gcnArchName looks like "gfx950:sramecc+:xnack-"
    char const * const name = properties.gcnArchName;

    *yes = name[0] == 'g' && name[1] == 'f' && name[2] == 'x'
        && name[3] == '9' && name[4] == '5' && name[5] == '0';
    *lds_bytes_per_block = static_cast<int>(properties.sharedMemPerBlock);

    return 0;
}

// Every call of the library that allocates or frees through HIP (memory, streams,
// events) is counted: the entry point that allocates nothing is tested against
// this count.
static std::atomic<long long> allocation_calls { 0 };

long long drb_hip_allocation_calls (void) { return allocation_calls.load(); }
void drb_hip_count_allocation (void) { allocation_calls.fetch_add(1); }

int drb_hip_alloc (void ** pointer, size_t bytes) { allocation_calls.fetch_add(1); return hipMalloc(pointer, bytes); }
int drb_hip_free (void * pointer) { allocation_calls.fetch_add(1); return hipFree(pointer); }
int drb_hip_host_alloc (void ** pointer, size_t bytes) { allocation_calls.fetch_add(1); return hipHostMalloc(pointer, bytes, hipHostMallocDefault); }
int drb_hip_host_free (void * pointer) { allocation_calls.fetch_add(1); return hipHostFree(pointer); }

// forgets the calling thread's last HIP error (a call that failed on purpose-built grounds, and was made good otherwise)
void drb_hip_clear_error (void)
{
    (void)hipGetLastError();
}

int drb_hip_memset_async (void * pointer, int value, size_t bytes, void * stream)
{
    return hipMemsetAsync(pointer, value, bytes, static_cast<hipStream_t>(stream));
}

int drb_hip_upload_async (void * device, void const * host, size_t bytes, void * stream)
{
    return hipMemcpyAsync(device, host, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream));
}

// A step's tables (descriptors, stretch table, coefficient jobs: some tens of KB in
// pinned host memory) brought over by ONE kernel that reads the host memory
// itself: a DMA copy of that size costs 20-30 us of latency each, this launch 5.
struct Drb_Table_Copies
{
    unsigned const * host [3];
    unsigned * device [3];
    unsigned words [3];
};

__global__ __launch_bounds__(256) void drb_table_kernel (Drb_Table_Copies const copies)
{
    #pragma unroll
    for (int t = 0; t < 3; t++)
    {
        unsigned const * const host = copies.host[t];
        unsigned * const device = copies.device[t];

        for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < copies.words[t]; i += gridDim.x * 256u) device[i] = host[i];
    }
}

int drb_hip_upload_tables (void * const * device, void const * const * pinned_host, size_t const * bytes, int count, void * stream)
{
    Drb_Table_Copies copies = {};
    unsigned most = 0;
    int used = 0;

    for (int t = 0; t < count; t++)
    {
        if (bytes[t] == 0) continue;

        // (whole 32-bit words of pinned memory, three tables a launch; anything else goes the usual way)
        if ((bytes[t] & 3) || bytes[t] > (64u << 20) || used == 3)
        {
            hipError_t const status = hipMemcpyAsync(device[t], pinned_host[t], bytes[t], hipMemcpyHostToDevice, static_cast<hipStream_t>(stream));

            if (status != hipSuccess) return status;

            continue;
        }

        copies.host[used] = static_cast<unsigned const *>(pinned_host[t]);
        copies.device[used] = static_cast<unsigned *>(device[t]);
        copies.words[used] = static_cast<unsigned>(bytes[t] / 4);

        if (copies.words[used] > most) most = copies.words[used];

        used++;
    }

    if (!used) return 0;

    unsigned const blocks = (most + 255u) / 256u < 64u ? (most + 255u) / 256u : 64u;

    drb_table_kernel<<<dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(copies);

    return hipGetLastError();
}

int drb_hip_download_async (void * host, void const * device, size_t bytes, void * stream)
{
    return hipMemcpyAsync(host, device, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream));
}

int drb_hip_stream_synchronize (void * stream) { return hipStreamSynchronize(static_cast<hipStream_t>(stream)); }

int drb_hip_stream_create (void ** stream)
{
    hipStream_t created;
    allocation_calls.fetch_add(1);

    hipError_t const status = hipStreamCreateWithFlags(&created, hipStreamNonBlocking);
    *stream = created;
    return status;
}

int drb_hip_stream_destroy (void * stream) { return hipStreamDestroy(static_cast<hipStream_t>(stream)); }

int drb_hip_stream_wait_event (void * stream, void * event)
{
    return hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0);
}

int drb_hip_event_create (void ** event)
{
    hipEvent_t created;
    allocation_calls.fetch_add(1);

    hipError_t const status = hipEventCreateWithFlags(&created, hipEventDisableTiming);
    *event = created;
    return status;
}

int drb_hip_timing_event_create (void ** event)
{
    hipEvent_t created;
    allocation_calls.fetch_add(1);

    hipError_t const status = hipEventCreate(&created);
    *event = created;
    return status;
}

int drb_hip_event_elapsed_ms (float * ms, void * start, void * stop)
{
    return hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop));
}

int drb_hip_event_destroy (void * event) { return hipEventDestroy(static_cast<hipEvent_t>(event)); }
int drb_hip_event_record (void * event, void * stream) { return hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)); }
int drb_hip_event_synchronize (void * event) { return hipEventSynchronize(static_cast<hipEvent_t>(event)); }
char const * drb_hip_error_text (int code) { return hipGetErrorString(static_cast<hipError_t>(code)); }

int drb_hip_launch_cascade
    (
        Drb_Device_Chain const * chain, int exact,
        Drb_Device_Call const * calls, int stream_count, int max_tiles,
        float const * history,
        long long source_channel_stride, long long target_channel_stride,
        int lds_bytes, int * fault_flag, void * stream
    )
{
    (void)exact;

    if (stream_count <= 0 || max_tiles <= 0) return 0;

    int bank_floats = 0;

    for (int k = 0; k < chain->stage_count; k++)
    {
        if (chain->stages[k].kind == kStageSinc)
        {
            bank_floats = (chain->stages[k].cubic ? 32 : 64) * (chain->stages[k].taps + 1);
        }
    }

    // (DRB_AMD_EXACT_LDS_PAD=bytes: LDS asked for and not used, to time the kernel at fewer workgroups per CU)
    static int const pad = [] { char const * const v = getenv("DRB_AMD_EXACT_LDS_PAD"); return v ? atoi(v) : 0; }();
    size_t const dynamic = static_cast<size_t>(lds_bytes) + bank_floats * sizeof(float) + sizeof(Ranges) + pad;

    // (the attribute is per device and batches may live on several: set on every
    // launch that needs it, which is cheap, rather than remembered per process)
    // the image whose fractional stage takes a lone channel in scalar registers, where a
    // workgroup's channel count can be odd (DRB_AMD_EXACT_LONE=0: the packed image throughout)
    static bool const lone_allowed = [] { char const * const v = getenv("DRB_AMD_EXACT_LONE"); return !(v && v[0] == '0'); }();
    bool const odd = lone_allowed && chain->channel_group >= 1
                  && (chain->channel_group % 2 != 0 || (chain->channels % chain->channel_group) % 2 != 0);
    int const lone = !odd ? 0 : chain->channel_group == 1 ? 2 : 1;

    if (dynamic > 48 * 1024)
    {
        hipError_t const status = hipFuncSetAttribute
        (
            lone == 2 ? reinterpret_cast<void const *>(drb_cascade_kernel<2>)
                      : lone ? reinterpret_cast<void const *>(drb_cascade_kernel<1>) : reinterpret_cast<void const *>(drb_cascade_kernel<0>),
            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(dynamic)
        );

        if (status != hipSuccess) return status;
    }

    long long const blocks = static_cast<long long>(stream_count) * max_tiles;

    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;

    if (chain->channel_group < 1 || chain->channel_group > chain->channels) return hipErrorInvalidValue;

    unsigned const groups = static_cast<unsigned>((chain->channels + chain->channel_group - 1) / chain->channel_group);

    // (DRB_AMD_SINC_SHARE=0: the fractional stage frame by frame throughout; =2: frames taken
    // two at a time, every pair the way of pairs whose bank rows differ, which else one pair
    // in a million takes: for the tests)
    static int const share = [] { char const * const v = getenv("DRB_AMD_SINC_SHARE"); return v ? atoi(v) : 1; }();
    Drb_Device_Chain launched = *chain;

    if (share == 0) launched.sinc_period = 0;

    // (DRB_AMD_EXACT_ABLATE=mask: phases skipped, for timing only)
    static int const ablate = [] { char const * const v = getenv("DRB_AMD_EXACT_ABLATE"); return v ? atoi(v) : 0; }();

    launched.ablate = (ablate & 31) | (share == 2 ? 32 : 0);

    if (lone == 2)
    {
        drb_cascade_kernel<2><<<dim3(static_cast<unsigned>(blocks), groups), dim3(kThreads), dynamic, static_cast<hipStream_t>(stream)>>>
        (
            launched, calls, stream_count, history,
            source_channel_stride, target_channel_stride, bank_floats, fault_flag
        );
    }
    else if (lone)
    {
        drb_cascade_kernel<1><<<dim3(static_cast<unsigned>(blocks), groups), dim3(kThreads), dynamic, static_cast<hipStream_t>(stream)>>>
        (
            launched, calls, stream_count, history,
            source_channel_stride, target_channel_stride, bank_floats, fault_flag
        );
    }
    else
    {
        drb_cascade_kernel<0><<<dim3(static_cast<unsigned>(blocks), groups), dim3(kThreads), dynamic, static_cast<hipStream_t>(stream)>>>
        (
            launched, calls, stream_count, history,
            source_channel_stride, target_channel_stride, bank_floats, fault_flag
        );
    }

    return hipGetLastError();
}

int drb_hip_launch_convert
    (
        Drb_Device_Call const * calls, int stream_count, long long most_frames, int channels,
        int source_type, int source_layout, long long source_channel_stride,
        int target_type, int target_layout, long long target_channel_stride, void * stream
    )
{
    if (stream_count <= 0 || most_frames <= 0) return 0;
    if (channels < 1 || channels > kConvertChannels) return hipErrorInvalidValue;

    long long const each = static_cast<long long>(kConvertFrames) * kConvertTiles;
    long long const blocks = (most_frames + each - 1) / each;

    if (blocks > 65535) return hipErrorInvalidValue;   // (a step has at most 2^24 frames: 16 384 blocks)

    drb_convert_kernel<<<dim3(static_cast<unsigned>(stream_count), static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream)>>>
    (
        calls, channels, source_type, source_layout, source_channel_stride, target_type, target_layout, target_channel_stride
    );

    return hipGetLastError();
}

int drb_hip_launch_history
    (
        Drb_Device_Chain const * chain,
        Drb_Device_Call const * calls, int stream_count,
        float const * history_in, float * history_out,
        long long source_channel_stride, void * stream
    )
{
    if (stream_count <= 0) return 0;

    drb_history_kernel<<<dim3(stream_count), dim3(kThreads), 0, static_cast<hipStream_t>(stream)>>>
    (
        *chain, calls, history_in, history_out, source_channel_stride
    );

    return hipGetLastError();
}

} // extern "C"
