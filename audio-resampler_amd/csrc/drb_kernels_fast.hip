// Matrix-core ("fast") cascade for gfx950: [half-band up2x] -> windowed-sinc
// fractional stage -> [half-band down2x], every filter run as a banded
// (Toeplitz) matrix product on v_mfma_f32_16x16x4_f32.
//
// Layout.  A workgroup owns one channel of one stream and a tile of 16
// segments.  Every buffer in LDS is stored "transposed": element (segment c,
// row t) sits at float index (t + lookback) * 16 + c, so the 16 segments are
// the 16 columns of the MFMA and a row block of 16 consecutive rows times 16
// segments is one 16 x 16 product tile:
//
//     D[row m][segment c] = sum_k A[m][k] * B[k][c],   B[k][c] = in(c, t0 + k)
//
// with A a banded matrix of filter coefficients (constant for the half-bands,
// rebuilt per tile from the exact int64 phase for the fractional stage, where
// row m of a block uses its own blended bank row, shifted by its own window
// position).  Segments are whole periods of the rate ratio apart, so all 16
// share A.  The first `lookback` rows of a segment repeat the tail of the
// previous segment; segment 0 has no previous segment, so it only warms the
// filters up and the tile's output comes from segments 1..15.
//
// Arithmetic is fused multiply-add in an order of the hardware's choosing and
// the fractional stage applies ONE blended kernel per output instead of
// blending two (or four) dot products; results agree with the upstream scalar
// code to about 1e-7 for inputs in [-1, 1] (bound 1e-5 relative / 1e-6
// absolute, checked by the tests).  Upstream formulas:
// drb-audio-converter.c:1984-2004 (up), 2033-2057 (down), 1617-1648 /
// 1677-1707 + 881-941 (fractional stage).

#include <hip/hip_runtime.h>

#include "drb_hip.h"

namespace
{

constexpr int kThreads = 256;
constexpr int kMaxSteps = 16;

enum { kSampleF32 = 0, kSampleI16 = 1, kSampleI32 = 2 };
enum { kPlanar = 0, kInterleaved = 1 };

typedef float float4_t __attribute__((ext_vector_type(4)));

__constant__ unsigned int drb_fast_table_words [] =
{
#include "../../data/drb_tables_words.inc"
};

__device__ inline float const * table_at (int float_offset)
{
    return reinterpret_cast<float const *>(drb_fast_table_words) + float_offset;
}

// Float index of (row, column) in a transposed buffer.  Rows alternate between
// the two halves of the 32 banks; the extra flip on bit 2 of the row spreads a
// wave's result writes (rows 4 apart per lane group) over both halves.
__device__ inline int cell (int row, int column)
{
    return ((row << 4) + column) ^ (((row >> 2) & 1) << 4);
}

__device__ inline float load_sample (void const * base, long long index, int type)
{
    switch (type)
    {
      case kSampleI16:
        return static_cast<float>(static_cast<short const *>(base)[index]) * (1.0f / 32768.0f);
      case kSampleI32:
        return static_cast<float>(static_cast<int const *>(base)[index]) * (1.0f / 2147483648.0f);
      default:
        return static_cast<float const *>(base)[index];
    }
}

__device__ inline void store_sample (void * base, long long index, int type, float value)
{
    switch (type)
    {
      case kSampleI16:
      {
        float const scaled = rintf(value * 32768.0f);
        int const q = scaled >= 32767.0f ? 32767
                    : scaled <= -32768.0f ? -32768
                    : scaled != scaled ? 0 : static_cast<int>(scaled);
        static_cast<short *>(base)[index] = static_cast<short>(q);
        break;
      }
      case kSampleI32:
      {
        float const scaled = rintf(value * 2147483648.0f);
        int const q = scaled >= 2147483648.0f ? 2147483647
                    : scaled <= -2147483648.0f ? (-2147483647 - 1)
                    : scaled != scaled ? 0 : static_cast<int>(scaled);
        static_cast<int *>(base)[index] = q;
        break;
      }
      default:
        static_cast<float *>(base)[index] = value;
    }
}

// Source frame of this workgroup's channel; negative frames come from the
// stream's history, anything older is silence.
__device__ inline float source_frame
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

    long long const index = chain.layout == kInterleaved
        ? static_cast<long long>(frame) * chain.channels + channel
        : channel * channel_stride + frame;

    return load_sample(reinterpret_cast<void const *>(call.source), index, chain.sample_type);
}

// Source frames consumed before the output with accumulator value `phase`, and
// its sub-sample position (drb-audio-converter.c:1617-1648 push, 1677-1707 pull).
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

// One row block: acc[row][segment] = sum over `steps` 4-tap steps.
__device__ inline float4_t row_block (float const * in, int first_row, int steps, float const (& a) [kMaxSteps])
{
    int const lane = threadIdx.x & 63, column = lane & 15, quad = lane >> 4;
    float4_t acc = { 0.0f, 0.0f, 0.0f, 0.0f };

    #pragma unroll
    for (int s = 0; s < kMaxSteps; s++)
    {
        if (s < steps)
        {
            float const b = in[cell(first_row + 4 * s + quad, column)];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b, acc, 0, 0, 0);
        }
    }

    return acc;
}

__global__ __launch_bounds__(kThreads) void drb_fast_kernel
    (
        Drb_Fast_Chain const chain,
        Drb_Device_Call const * __restrict__ calls, int stream_count,
        float const * __restrict__ history,
        long long source_channel_stride, long long target_channel_stride,
        int * fault_flag
    )
{
    extern __shared__ __attribute__((aligned(16))) float lds [];

    int const pair = blockIdx.x % (stream_count * chain.channels);
    int const tile = blockIdx.x / (stream_count * chain.channels);
    int const stream = pair / chain.channels, channel = pair - stream * chain.channels;
    Drb_Device_Call const call = calls[stream];

    int const first = tile * chain.tile_frames;

    if (first >= call.target_frames) return;

    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int const column = lane & 15, quad = lane >> 4;

    float * const buf_x = lds + chain.off_x;
    float * const buf_u = lds + chain.off_u;
    float * const buf_d_odd = lds + chain.off_d_odd;
    float * const buf_d_even = lds + chain.off_d_even;
    float * const bank = lds + chain.off_bank;

    // ---- where the tile sits on every edge (indices relative to the step) ----
    // target index of (segment 0, row 0); segment 1 starts at the tile's first frame
    int const base_y = first - chain.seg_y;
    int const delay_behind = chain.delay_place == 1 ? call.delay : 0;
    int const delay_front = chain.delay_place == 2 ? call.delay : 0;
    int const base_s = (chain.has_down ? 2 * base_y : base_y) - delay_behind;

    // window ends of the middle segment (8) define the phases of all segments
    int middle_consumed;
    double unused;

    sinc_position(call.phase + static_cast<long long>(base_s + 8 * chain.seg_s) * chain.target_delta,
                  chain.source_delta, chain.scale, chain.sinc_pull, middle_consumed, unused);

    // sinc-input index of (segment 0, row 0): the first row's window ends at row
    // -2 or -1 (so that the last row's window still ends inside the segment),
    // chosen so that the producer-side index is even when an up-sampler feeds it
    int base_u = middle_consumed - 8 * chain.seg_u + 2;

    if (chain.has_up && ((base_u - delay_front) & 1)) base_u -= 1;

    int const base_x = chain.has_up ? (base_u - delay_front) / 2 : base_u - delay_front;
    float const * const stream_history = history + static_cast<long long>(stream) * chain.channels * chain.history;

    // ---- coefficient bank of the fractional stage -> LDS, rows skewed by one ----
    {
        int const taps = chain.sinc_taps, rows = chain.sinc_cubic ? 32 : 64;
        float const * const source = table_at(chain.sinc_table);

        for (int i = threadIdx.x; i < rows * taps; i += kThreads)
        {
            bank[(i / taps) * (taps + 1) + i % taps] = source[i];
        }
    }

    // ---- load the source: rows -look_x .. seg_x - 1 of all 16 segments ----------
    {
        int const rows = chain.seg_x + chain.look_x;
        int const segment = threadIdx.x & 15;

        for (int row = threadIdx.x >> 4; row < rows; row += kThreads / 16)
        {
            int const t = row - chain.look_x;
            float const v = source_frame(chain, call, stream_history, source_channel_stride, channel,
                                         base_x + segment * chain.seg_x + t);

            buf_x[cell(row, segment)] = v;

            if (chain.has_up)
            {
                // even sinc-input samples are the source delayed by half the filter:
                // u[2 f] = x[f - N / 2]  ->  row 2 t + N of the same segment
                int const u = 2 * t + chain.up_taps;

                if (u >= -chain.look_u && u < chain.seg_u) buf_u[cell(u + chain.look_u, segment)] = v;
            }
        }
    }

    // ---- half-band coefficients as banded A fragments ---------------------------
    // up:   A[m][k] = 2 h[k - m - 1]   (window starts at row 16 i - N)
    // down: A[m][k] =   h[k - m]
    float a_up [kMaxSteps], a_down [kMaxSteps];

    #pragma unroll
    for (int s = 0; s < kMaxSteps; s++)
    {
        int const k = 4 * s + quad;
        int const ju = k - column - 1, jd = k - column;

        a_up[s] = chain.has_up && ju >= 0 && ju < chain.up_taps ? 2.0f * table_at(chain.up_table)[ju] : 0.0f;
        a_down[s] = chain.has_down && jd >= 0 && jd < chain.down_taps ? table_at(chain.down_table)[jd] : 0.0f;
    }

    __syncthreads();

    // ---- up-sampler: odd sinc-input samples ---------------------------------------
    if (chain.has_up)
    {
        int const steps = (chain.up_taps + 16) / 4;
        int const blocks = (chain.seg_x + 15) / 16;

        for (int block = wave; block < blocks; block += kThreads / 64)
        {
            float4_t const acc = row_block(buf_x, 16 * block - chain.up_taps + chain.look_x, steps, a_up);

            #pragma unroll
            for (int r = 0; r < 4; r++)
            {
                int const t = 16 * block + 4 * quad + r;   // source row
                int const u = 2 * t + 1;                   // sinc-input row

                if (t < chain.seg_x)
                {
                    buf_u[cell(u + chain.look_u, column)] = acc[r];

                    // tail of a segment = lookback of the next one
                    if (u >= chain.seg_u - chain.look_u && column < 15)
                    {
                        buf_u[cell(u - chain.seg_u + chain.look_u, column + 1)] = acc[r];
                    }
                }
            }
        }

        __syncthreads();
    }

    // ---- fractional stage -----------------------------------------------------------
    {
        int const chains = chain.blocks * chain.periods;
        int const share = (chains + 3) / 4;
        int const taps = chain.sinc_taps, stride = taps + 1;
        int built = -1, first_row = 0;
        float a [kMaxSteps];

        for (int q = wave * share; q < min((wave + 1) * share, chains); q++)
        {
            int const block = q / chain.periods, period = q - block * chain.periods;

            if (block != built)
            {
                // this lane's row of the block, at the middle segment: window end
                // (relative to the segment) and blended kernel
                int const row = 16 * block + column;
                int consumed;
                double blend;

                sinc_position(call.phase + static_cast<long long>(base_s + 8 * chain.seg_s + row) * chain.target_delta,
                              chain.source_delta, chain.scale, chain.sinc_pull, consumed, blend);

                int const end = consumed - (base_u + 8 * chain.seg_u);
                int const end_of_first = __builtin_amdgcn_readfirstlane(end);

                first_row = (end_of_first - taps) & ~1;

                int const skew = end - taps - first_row;

                if (column == 15 && skew + taps > 4 * chain.ksteps) atomicOr(fault_flag, 8);

                double const scaled = blend * (chain.sinc_cubic ? 29.0 : 63.0);
                int const bank_row = static_cast<int>(scaled);
                float const sub = static_cast<float>(scaled - static_cast<double>(bank_row));
                float const * const k0 = bank + bank_row * stride;

                float w0, w1, w2 = 0.0f, w3 = 0.0f;

                if (chain.sinc_cubic)
                {
                    // Catmull-Rom weights of the four rows (drb-audio-converter.c:861-874)
                    float const b = sub, b2 = b * b, b3 = b2 * b;

                    w0 = -0.5f * b + b2 - 0.5f * b3;
                    w1 = 1.0f - 2.5f * b2 + 1.5f * b3;
                    w2 = 0.5f * b + 2.0f * b2 - 1.5f * b3;
                    w3 = -0.5f * b2 + 0.5f * b3;
                }
                else
                {
                    w0 = 1.0f - sub;
                    w1 = sub;
                }

                #pragma unroll
                for (int s = 0; s < kMaxSteps; s++)
                {
                    int const t = 4 * s + quad - skew;
                    float value = 0.0f;

                    if (s < chain.ksteps && t >= 0 && t < taps)
                    {
                        value = w0 * k0[t] + w1 * k0[stride + t];

                        if (chain.sinc_cubic) value += w2 * k0[2 * stride + t] + w3 * k0[3 * stride + t];
                    }

                    a[s] = value;
                }

                built = block;
            }

            float4_t const acc = row_block(buf_u, first_row + period * chain.period_in + chain.look_u, chain.ksteps, a);

            #pragma unroll
            for (int r = 0; r < 4; r++)
            {
                int const in_period = 16 * block + 4 * quad + r;
                int const t = period * chain.period_out + in_period;   // row of the sinc output = row of d

                if (in_period >= chain.period_out) continue;

                if (chain.has_down)
                {
                    float * const plane = (t & 1) ? buf_d_odd : buf_d_even;
                    int const g = t >> 1, half = chain.seg_s >> 1;

                    plane[cell(g + chain.look_d, column)] = acc[r];

                    if (g >= half - chain.look_d && column < 15)
                    {
                        plane[cell(g - half + chain.look_d, column + 1)] = acc[r];
                    }
                }
                else
                {
                    int const frame = base_y + column * chain.seg_y + t;

                    if (column >= 1 && frame < call.target_frames)
                    {
                        long long const index = chain.layout == kInterleaved
                            ? static_cast<long long>(frame) * chain.channels + channel
                            : channel * target_channel_stride + frame;

                        store_sample(reinterpret_cast<void *>(call.target), index, chain.sample_type, acc[r]);
                    }
                }
            }
        }
    }

    // ---- down-sampler ------------------------------------------------------------------
    if (chain.has_down)
    {
        __syncthreads();

        int const steps = (chain.down_taps + 16) / 4;
        int const blocks = (chain.seg_y + 15) / 16;

        for (int block = wave; block < blocks; block += kThreads / 64)
        {
            float4_t const acc = row_block(buf_d_odd, 16 * block - chain.down_taps + chain.look_d, steps, a_down);

            #pragma unroll
            for (int r = 0; r < 4; r++)
            {
                int const t = 16 * block + 4 * quad + r;
                int const frame = base_y + column * chain.seg_y + t;

                if (t < chain.seg_y && column >= 1 && frame < call.target_frames)
                {
                    float const centre = buf_d_even[cell(t - chain.down_taps / 2 + chain.look_d, column)];
                    float const value = fmaf(0.5f, centre, acc[r]);
                    long long const index = chain.layout == kInterleaved
                        ? static_cast<long long>(frame) * chain.channels + channel
                        : channel * target_channel_stride + frame;

                    store_sample(reinterpret_cast<void *>(call.target), index, chain.sample_type, value);
                }
            }
        }
    }
}

} // namespace

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

    static size_t raised = 0;

    if (dynamic > 48 * 1024 && dynamic > raised)
    {
        hipError_t const status = hipFuncSetAttribute
        (
            reinterpret_cast<void const *>(drb_fast_kernel),
            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(dynamic)
        );

        if (status != hipSuccess) return status;

        raised = dynamic;
    }

    long long const blocks = static_cast<long long>(stream_count) * chain->channels * max_tiles;

    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;

    drb_fast_kernel<<<dim3(static_cast<unsigned>(blocks)), dim3(kThreads), dynamic, static_cast<hipStream_t>(stream)>>>
    (
        *chain, calls, stream_count, history,
        source_channel_stride, target_channel_stride, fault_flag
    );

    return hipGetLastError();
}
