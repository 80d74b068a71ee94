/*
 * TEST INFRASTRUCTURE ONLY - CPU restatement of the upstream converter's
 * resampling path ("the oracle").  It is the checker the HIP implementation
 * is compared against; the product library never includes, links or calls it.
 *
 * Style: whole-stream, closed-form.  Where the upstream code keeps ring
 * windows and per-call state, this file evaluates every stage as a function
 * of the absolute frame index with x[i] = 0 for i < 0, which is what the
 * upstream zero-initialised windows amount to.  Arithmetic follows the
 * upstream scalar ("No SIMD") branch exactly: float32 data, one rounding per
 * multiply and per add, taps accumulated in ascending order
 * (drb-audio-converter.c:615-798); build with -ffp-contract=off and without
 * any SIMD-enabling -m flag.
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks this file bit-for-bit
 * against tests/golden/ (vectors produced by the upstream converter compiled
 * from its own source, see tests/golden/make_golden.py) and, when oracle/_ref
 * is present, against that build directly.
 */
#include "drb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- coefficient tables (data; see tools/extract_tables.py) -------------- */

static unsigned int const table_words [] =
{
#include "../data/drb_tables_words.inc"
};

static float const * table (int float_offset)
{
    return (float const *)(void const *)table_words + float_offset;
}

/* half-band kernel for order 16..96: N = order / 2 taps */
static float const * half_band_kernel (int order)
{
    switch (order)
    {
      case 16: return table(0);
      case 32: return table(8);
      case 48: return table(24);
      case 64: return table(48);
      case 80: return table(80);
      case 96: return table(120);
    }
    return 0;
}

static float const * sinc_bank (int taps, int cubic)
{
    if (!cubic) return taps == 16 ? table(168) : table(1192);
    return taps == 16 ? table(3240) : table(3752);
}

/* ---- planning ------------------------------------------------------------ */

static int valid_rate (int rate)
{
    /* drb-audio-converter.h:31-46 */
    static int const rates [] =
    {
        8000, 11025, 16000, 22050, 44100, 48000, 60000,
        88200, 96000, 120000, 176400, 192000, 240000
    };

    for (unsigned i = 0; i < sizeof(rates) / sizeof(rates[0]); i++)
    {
        if (rates[i] == rate) return 1;
    }
    return 0;
}

static int valid_block_size (int size)
{
    /* drb-audio-converter.h:49-58 */
    return size == 1 || size == 4 || size == 16 || size == 64
        || size == 256 || size == 1024 || size == 4096;
}

/* number of 2x stages needed before the residual ratio is within a factor of
 * sqrt(2)-ish of unity, evaluated in float exactly as
 * drb-audio-converter.c:2887-2895 / 2912-2920 do */
static int doubling_count (int low_rate, int high_rate)
{
    int factor = 0;

    while ((float)high_rate / (low_rate * (1 << factor))
         > (float)(low_rate * (2 << factor)) / high_rate)
    {
        factor += 1;
    }

    return factor;
}

int drb_oracle_plan
    (
        int source_rate, int target_rate, int channels,
        int block_size, int max_block_count, int direction, int quality,
        Drb_Oracle_Plan * plan
    )
{
    /* validation: drb-audio-converter.c:3264-3364 (the side opposite the
     * user block has block size 1 and max block count 0, always valid) */
    if (!valid_rate(source_rate) || !valid_rate(target_rate)) return 0;
    if (channels < 1 || channels > 8) return 0;
    if (!valid_block_size(block_size)) return 0;
    if (max_block_count < 0 || max_block_count > 8192) return 0;
    if (block_size * max_block_count > 8192) return 0;

    memset(plan, 0, sizeof(*plan));

    plan->source_rate = source_rate;
    plan->target_rate = target_rate;
    plan->channels = channels;
    plan->block_size = block_size;
    plan->max_block_count = max_block_count;
    plan->direction = direction;
    plan->quality = quality;

    /* drb-audio-converter.c:2875-2935.  The 2x structure is the same whether
     * the chain is planned forwards (push) or planned for the opposite
     * conversion and inverted (pull, drb-audio-converter.c:3191-3212). */
    int up = 0, down = 0, fractional = 0;

    if (source_rate < target_rate)
    {
        up = doubling_count(source_rate, target_rate);
        fractional = source_rate * (1 << up) != target_rate;
    }
    else
    {
        down = doubling_count(target_rate, source_rate);
        fractional = target_rate * (1 << down) != source_rate;
    }

    if (fractional && quality > 0)
    {
        up += 1;    /* oversampling, drb-audio-converter.c:2903-2907 */
        down += 1;
    }

    if (up >= 6 || down >= 6) return -1;
    if ((channels > 1 ? 2 : 0) + up + down + (fractional ? 2 : 1) > 10) return -1;
    if (max_block_count == 0) return -1;

    /* filter orders per quality, longest first: drb-audio-converter.c:2955-3017 */
    static int const orders [4][6] =
    {
        { 48, 32, 16, 16, 16, 16 },
        { 64, 48, 32, 16, 16, 16 },
        { 80, 64, 48, 32, 16, 16 },
        { 96, 80, 64, 48, 32, 16 }
    };
    static int const sinc_taps [4] = { 16, 16, 32, 32 };
    static int const sinc_cubic [4] = { 0, 1, 0, 1 };

    int const pull = direction == 1;

    /* slicer: present with a fractional stage, or when the unit-block side is
     * finer than the user block (drb-audio-converter.c:3051-3059).  Its block
     * is the larger of its two edge block sizes (drb-audio-converter.c:2673). */
    int const has_slicer = fractional || block_size > 1;

    /* 2x stages between the slicer and the user's block side */
    int const block_side_2x = pull ? up : down;
    int const unit_side_2x = pull ? down : up;
    int const block_side_bs = block_size << block_side_2x;
    int const unit_side_bs = fractional ? 1 : 1 << unit_side_2x;
    int const slicer_block = block_side_bs > unit_side_bs ? block_side_bs : unit_side_bs;

    /* chain in data-flow order */
    int count = 0, rate = source_rate;

    for (int i = 0; i < up; i++)
    {
        plan->stage_kind[count] = drb_oracle_stage_up2x;
        plan->stage_arg[count] = orders[quality][i];
        plan->stage_in_rate[count] = rate;
        plan->stage_out_rate[count] = rate * 2;
        rate *= 2;
        count++;
    }

    if (has_slicer && pull)
    {
        plan->stage_kind[count] = drb_oracle_stage_slicer;
        plan->stage_arg[count] = slicer_block;
        plan->stage_in_rate[count] = plan->stage_out_rate[count] = rate;
        count++;
    }

    if (fractional)
    {
        int const out_rate = target_rate << down;

        plan->stage_kind[count] = drb_oracle_stage_sinc;
        plan->stage_arg[count] = sinc_taps[quality] + 100 * sinc_cubic[quality];
        plan->stage_in_rate[count] = rate;
        plan->stage_out_rate[count] = out_rate;
        rate = out_rate;
        count++;
    }

    if (has_slicer && !pull)
    {
        plan->stage_kind[count] = drb_oracle_stage_slicer;
        plan->stage_arg[count] = slicer_block;
        plan->stage_in_rate[count] = plan->stage_out_rate[count] = rate;
        count++;
    }

    for (int i = down - 1; i >= 0; i--)
    {
        plan->stage_kind[count] = drb_oracle_stage_down2x;
        plan->stage_arg[count] = orders[quality][i];
        plan->stage_in_rate[count] = rate;
        plan->stage_out_rate[count] = rate / 2;
        rate /= 2;
        count++;
    }

    plan->stage_count = count;

    /* throttling = frames per driver sub-block on the unit-block side
     * (drb-audio-converter.c:3697): walk the max block count from the user's
     * block side to the other end through the slicer and fractional stage
     * restraints (drb-audio-converter.c:2715-2723, 1843-1853). */
    long blocks = max_block_count;

    if (has_slicer)
    {
        blocks = blocks * block_side_bs / unit_side_bs;
    }

    if (fractional)
    {
        /* rates of the fractional stage as seen from the planning direction:
         * its "target" is the slicer side */
        long const near_rate = pull ? (long)source_rate << up : (long)target_rate << down;
        long const far_rate = pull ? (long)target_rate << down : (long)source_rate << up;

        blocks = blocks * far_rate / near_rate;
        blocks = blocks / (1 << unit_side_2x);
    }

    plan->throttling = (int)blocks;

    if (plan->throttling == 0) return -1;

    /* a 2x stage with no slicer in front of it sees whatever frame count the
     * user passes; odd counts then drop or invent a frame upstream */
    plan->consistent = has_slicer || block_side_2x == 0;

    return 1;
}

/* ---- frame counts -------------------------------------------------------- */

typedef __int128 wide;

static long long sinc_delta (int rate)
{
    return (long long)(0x1.0p48 / rate); /* drb-audio-converter.c:1763-1764 */
}

/* target frames emitted once `consumed` source frames were pushed, starting
 * from phase = source_delta (drb-audio-converter.c:1516-1519, 1765) */
static long sinc_push_count (long long sd, long long td, long consumed)
{
    wide const count = ((wide)td - sd + (wide)sd * consumed) / td;
    return (long)count;
}

/* source frames consumed once `produced` target frames were pulled
 * (drb-audio-converter.c:1535-1538) */
static long sinc_pull_count (long long sd, long long td, long produced)
{
    wide const count = ((wide)sd - td + (wide)td * produced) / sd;
    return (long)count;
}

static long slicer_count (int block, long frames)
{
    /* pre-filled to half a block: drb-audio-converter.c:2495-2499, 2687 */
    return (block / 2 + frames) / block * block;
}

long drb_oracle_push_count (Drb_Oracle_Plan const * plan, long frames)
{
    for (int i = 0; i < plan->stage_count; i++)
    {
        switch (plan->stage_kind[i])
        {
          case drb_oracle_stage_up2x: frames *= 2; break;
          case drb_oracle_stage_down2x: frames /= 2; break;
          case drb_oracle_stage_slicer: frames = slicer_count(plan->stage_arg[i], frames); break;
          case drb_oracle_stage_sinc:
            frames = sinc_push_count
            (
                sinc_delta(plan->stage_in_rate[i]),
                sinc_delta(plan->stage_out_rate[i]), frames
            );
            break;
        }
    }
    return frames;
}

long drb_oracle_pull_count (Drb_Oracle_Plan const * plan, long frames)
{
    for (int i = plan->stage_count - 1; i >= 0; i--)
    {
        switch (plan->stage_kind[i])
        {
          case drb_oracle_stage_up2x: frames /= 2; break;
          case drb_oracle_stage_down2x: frames *= 2; break;
          case drb_oracle_stage_slicer: frames = slicer_count(plan->stage_arg[i], frames); break;
          case drb_oracle_stage_sinc:
            frames = sinc_pull_count
            (
                sinc_delta(plan->stage_in_rate[i]),
                sinc_delta(plan->stage_out_rate[i]), frames
            );
            break;
        }
    }
    return frames;
}

int drb_oracle_callback_counts
    (
        Drb_Oracle_Plan const * plan, int const * chunks, int chunk_count,
        int * counts, int capacity
    )
{
    /* the driver cuts every process() call into sub-blocks of at most
     * `throttling` frames and invokes the callback once per sub-block
     * (drb-audio-converter.c:3511-3515, 3573 / 3601-3605, 3636) */
    long total = 0;
    int callbacks = 0;

    for (int chunk = 0; chunk < chunk_count; chunk++)
    {
        for (int offset = 0; offset < chunks[chunk]; )
        {
            int const left = chunks[chunk] - offset;
            int const step = left < plan->throttling ? left : plan->throttling;

            long const before = plan->direction
                ? drb_oracle_pull_count(plan, total)
                : drb_oracle_push_count(plan, total);

            total += step;

            long const after = plan->direction
                ? drb_oracle_pull_count(plan, total)
                : drb_oracle_push_count(plan, total);

            if (callbacks < capacity) counts[callbacks] = (int)(after - before);

            callbacks++;
            offset += step;
        }
    }

    return callbacks;
}

/* ---- stages -------------------------------------------------------------- */

static inline float at (float const * x, long length, long index)
{
    return index >= 0 && index < length ? x[index] : 0.0f;
}

/* strictly sequential dot product: drb-audio-converter.c:615-798 */
static float dot (float const * kernel, float const * window, int taps)
{
    float acc = 0.0f;

    for (int j = 0; j < taps; j++)
    {
        acc += kernel[j] * window[j];
    }

    return acc;
}

/* y[2f] = (x[f - N/2] * 0.5) * 2, y[2f+1] = 2 * sum_j k[j] x[f - N + 1 + j]
 * (drb-audio-converter.c:1984-2004) */
static void up2x (int order, float const * x, long frames, float * y)
{
    int const taps = order / 2;
    float const * const kernel = half_band_kernel(order);
    float window [48];

    for (long f = 0; f < frames; f++)
    {
        for (int j = 0; j < taps; j++)
        {
            window[j] = at(x, frames, f - taps + 1 + j);
        }

        float const acc_0 = at(x, frames, f - taps / 2) * 0.5f;
        float const acc_1 = dot(kernel, window, taps);

        y[2 * f + 0] = acc_0 * 2.0f;
        y[2 * f + 1] = acc_1 * 2.0f;
    }
}

/* y[f] = 0.5 x[2f - N] + sum_j k[j] x[2 (f - N + j) + 1]
 * (drb-audio-converter.c:2033-2057; the two windows are read before the
 * frame's own pair of samples is written) */
static void down2x (int order, float const * x, long in_frames, long frames, float * y)
{
    int const taps = order / 2;
    float const * const kernel = half_band_kernel(order);
    float window [48];

    for (long f = 0; f < frames; f++)
    {
        for (int j = 0; j < taps; j++)
        {
            window[j] = at(x, in_frames, 2 * (f - taps + j) + 1);
        }

        float const acc_0 = at(x, in_frames, 2 * f - taps) * 0.5f;
        float const acc_1 = dot(kernel, window, taps);

        y[f] = acc_0 + acc_1;
    }
}

/* drb-audio-converter.c:827-848 */
static float blend_linear (float blend, float const points [2])
{
    static float const mat [2][2] = { { +1.0f, 0.0f }, { -1.0f, +1.0f } };

    float const a_0 = dot(mat[0], points, 2);
    float const a_1 = dot(mat[1], points, 2);

    return a_1 * blend + a_0;
}

/* Catmull-Rom, Horner in float: drb-audio-converter.c:850-875 */
static float blend_cubic (float blend, float const points [4])
{
    static float const mat [4][4] =
    {
        {  0.0f,  1.0f,  0.0f,  0.0f },
        { -0.5f,  0.0f,  0.5f,  0.0f },
        {  1.0f, -2.5f,  2.0f, -0.5f },
        { -0.5f,  1.5f, -1.5f,  0.5f }
    };

    float const a_0 = dot(mat[0], points, 4);
    float const a_1 = dot(mat[1], points, 4);
    float const a_2 = dot(mat[2], points, 4);
    float const a_3 = dot(mat[3], points, 4);

    return (((a_3 * blend + a_2) * blend) + a_1) * blend + a_0;
}

/* One output of the fractional stage from the `taps` newest consumed source
 * frames (oldest first) and the sub-sample position `blend` in [0, 1):
 * drb-audio-converter.c:881-941.  Rows are consecutive entries of the bank;
 * the row index and the remainder are formed in double, the remainder is then
 * narrowed to float. */
static float sinc_point (int taps, int cubic, double blend, float const * window)
{
    float const * const bank = sinc_bank(taps, cubic);
    double const scale = cubic ? 29.0 : 63.0;
    int const index = (int)(blend * scale);
    double const sub_blend = blend * scale - (double)index;
    float points [4];

    for (int r = 0; r < (cubic ? 4 : 2); r++)
    {
        points[r] = dot(bank + (index + r) * taps, window, taps);
    }

    return cubic ? blend_cubic((float)sub_blend, points)
                 : blend_linear((float)sub_blend, points);
}

/* Fractional stage over a whole stream.  With P_n = Sd + n Td:
 *   push (drb-audio-converter.c:1617-1648): output n is taken after
 *        m = ceil(P_n / Sd) - 1 source frames, at blend (P_n - m Sd - 1) / Sd
 *   pull (drb-audio-converter.c:1677-1707): m = floor(P_n / Sd),
 *        blend (P_n - m Sd) / Sd
 * and the window holds source frames m - taps .. m - 1. */
static void sinc_stage
    (
        int pull, int taps, int cubic, int in_rate, int out_rate,
        float const * x, long in_frames, long frames, float * y
    )
{
    long long const sd = sinc_delta(in_rate);
    long long const td = sinc_delta(out_rate);
    double const scale = 1.0 / sd; /* drb-audio-converter.c:1767 */
    float window [32];

    for (long n = 0; n < frames; n++)
    {
        wide const phase = (wide)sd + (wide)n * td;
        long const m = pull ? (long)(phase / sd) : (long)((phase + sd - 1) / sd) - 1;
        long long const rest = (long long)(phase - (wide)m * sd);
        double const blend = pull ? rest * scale : (rest - 1) * scale;

        for (int t = 0; t < taps; t++)
        {
            window[t] = at(x, in_frames, m - taps + t);
        }

        y[n] = sinc_point(taps, cubic, blend, window);
    }
}

/* Push: y[j] = x[j - B/2] (drb-audio-converter.c:2502-2575, 2684-2688).  Pull: the
 * block starts at index B/2 and hands out its B - B/2 remaining frames of
 * silence before it fetches anything (drb-audio-converter.c:2592-2606), so
 * y[j] = x[j - (B - B/2)]: the same for every even B, one frame later for B = 1. */
static void slicer_stage (int pull, int block, float const * x, long in_frames, long frames, float * y)
{
    int const delay = pull ? block - block / 2 : block / 2;

    for (long j = 0; j < frames; j++)
    {
        y[j] = at(x, in_frames, j - delay);
    }
}

/* ---- whole-stream drivers ------------------------------------------------ */

/* Runs one channel through the chain given the frame count on every edge. */
static int run_channel
    (
        Drb_Oracle_Plan const * plan, long const * edge,
        float const * in, float * out
    )
{
    float * current = malloc(sizeof(float) * (size_t)(edge[0] > 0 ? edge[0] : 1));

    if (!current) return 0;

    memcpy(current, in, sizeof(float) * (size_t)edge[0]);

    for (int i = 0; i < plan->stage_count; i++)
    {
        long const n_in = edge[i], n_out = edge[i + 1];
        float * const next = malloc(sizeof(float) * (size_t)(n_out > 0 ? n_out : 1));

        if (!next) { free(current); return 0; }

        switch (plan->stage_kind[i])
        {
          case drb_oracle_stage_up2x:
            up2x(plan->stage_arg[i], current, n_out / 2, next);
            break;
          case drb_oracle_stage_down2x:
            down2x(plan->stage_arg[i], current, n_in, n_out, next);
            break;
          case drb_oracle_stage_slicer:
            slicer_stage(plan->direction, plan->stage_arg[i], current, n_in, n_out, next);
            break;
          case drb_oracle_stage_sinc:
            sinc_stage
            (
                plan->direction, plan->stage_arg[i] % 100, plan->stage_arg[i] >= 100,
                plan->stage_in_rate[i], plan->stage_out_rate[i],
                current, n_in, n_out, next
            );
            break;
        }

        free(current);
        current = next;
    }

    memcpy(out, current, sizeof(float) * (size_t)edge[plan->stage_count]);
    free(current);

    return 1;
}

long drb_oracle_push
    (
        Drb_Oracle_Plan const * plan,
        float const * in, long source_frames, long in_stride,
        float * out, long out_capacity, long out_stride
    )
{
    long edge [drb_oracle_max_stages + 1];

    edge[0] = source_frames;

    for (int i = 0; i < plan->stage_count; i++)
    {
        Drb_Oracle_Plan single = *plan;

        single.stage_count = 1;
        single.stage_kind[0] = plan->stage_kind[i];
        single.stage_arg[0] = plan->stage_arg[i];
        single.stage_in_rate[0] = plan->stage_in_rate[i];
        single.stage_out_rate[0] = plan->stage_out_rate[i];

        edge[i + 1] = drb_oracle_push_count(&single, edge[i]);
    }

    long const produced = edge[plan->stage_count];

    if (produced > out_capacity) return -2;

    for (int channel = 0; channel < plan->channels; channel++)
    {
        if (!run_channel(plan, edge, in + channel * in_stride, out + channel * out_stride))
        {
            return -3;
        }
    }

    return produced;
}

long drb_oracle_pull
    (
        Drb_Oracle_Plan const * plan,
        float const * in, long source_frames, long in_stride,
        float * out, long target_frames, long out_stride
    )
{
    long edge [drb_oracle_max_stages + 1];

    edge[plan->stage_count] = target_frames;

    for (int i = plan->stage_count - 1; i >= 0; i--)
    {
        Drb_Oracle_Plan single = *plan;

        single.stage_count = 1;
        single.stage_kind[0] = plan->stage_kind[i];
        single.stage_arg[0] = plan->stage_arg[i];
        single.stage_in_rate[0] = plan->stage_in_rate[i];
        single.stage_out_rate[0] = plan->stage_out_rate[i];

        edge[i] = drb_oracle_pull_count(&single, edge[i + 1]);
    }

    long const consumed = edge[0];
    float * const padded = calloc((size_t)(consumed > 0 ? consumed : 1), sizeof(float));

    if (!padded) return -3;

    for (int channel = 0; channel < plan->channels; channel++)
    {
        long const have = source_frames < consumed ? source_frames : consumed;

        memset(padded, 0, sizeof(float) * (size_t)consumed);
        memcpy(padded, in + channel * in_stride, sizeof(float) * (size_t)(have > 0 ? have : 0));

        if (!run_channel(plan, edge, padded, out + channel * out_stride))
        {
            free(padded);
            return -3;
        }
    }

    free(padded);

    return consumed;
}

/* ---- integer sample conventions ------------------------------------------ */

void drb_oracle_i16_to_f32 (short const * in, float * out, long count)
{
    for (long i = 0; i < count; i++) out[i] = (float)in[i] * (1.0f / 32768.0f);
}

void drb_oracle_i32_to_f32 (int const * in, float * out, long count)
{
    for (long i = 0; i < count; i++) out[i] = (float)in[i] * (1.0f / 2147483648.0f);
}

void drb_oracle_f32_to_i16 (float const * in, short * out, long count)
{
    for (long i = 0; i < count; i++)
    {
        float const scaled = rintf(in[i] * 32768.0f);

        out[i] = scaled >= 32767.0f ? 32767
               : scaled <= -32768.0f ? -32768
               : scaled != scaled ? 0 : (short)scaled;
    }
}

void drb_oracle_f32_to_i32 (float const * in, int * out, long count)
{
    for (long i = 0; i < count; i++)
    {
        float const scaled = rintf(in[i] * 2147483648.0f);

        out[i] = scaled >= 2147483648.0f ? 2147483647
               : scaled <= -2147483648.0f ? (-2147483647 - 1)
               : scaled != scaled ? 0 : (int)scaled;
    }
}
