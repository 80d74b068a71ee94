/*
 * Host-side planner.  It must resolve every conversion to exactly the chain
 * the upstream library builds (drb-audio-converter.c:2863-3234), because the
 * chain fixes which filters run, the slicer's delay and the block-quantised
 * frame counts.  The implementation is a small constraint solver over the
 * edges between passes: every pass kind contributes one "settle" rule that
 * fills in unknown (zero) fields of its two edges from the known ones.
 */
#include "drb_plan.h"
#include "drb_hip.h"

#include <stdlib.h>
#include <string.h>

enum { layout_interleaved = 1, layout_planar = 2 };

/* ---- validation (drb-audio-converter.c:3264-3364) ------------------------ */

static int one_of (int value, int const * set, int count)
{
    for (int i = 0; i < count; i++)
    {
        if (set[i] == value) return 1;
    }
    return 0;
}

static int edge_is_valid (Drb_Edge const * edge)
{
    static int const rates [] =
    {
        8000, 11025, 16000, 22050, 44100, 48000, 60000,
        88200, 96000, 120000, 176400, 192000, 240000
    };
    static int const blocks [] = { 1, 4, 16, 64, 256, 1024, 4096 };

    return one_of(edge->rate, rates, 13)
        && edge->channels >= 1 && edge->channels <= 8
        && one_of(edge->block, blocks, 7)
        && edge->max_blocks >= 0 && edge->max_blocks <= 8192
        && edge->block * edge->max_blocks <= 8192;
}

/* ---- which passes (drb-audio-converter.c:2875-3077) ---------------------- */

/* Smallest number of doublings of `low` after which the remaining ratio to
 * `high` is closer to unity than one more doubling would leave it; the
 * comparison is done in float, as upstream does (2887-2895, 2912-2920). */
static int doublings (int low, int high)
{
    int count = 0;

    while ((float)high / (low * (1 << count)) > (float)(low * (2 << count)) / high)
    {
        count++;
    }

    return count;
}

static int const half_band_orders [4][6] =
{
    /* poor */ { 48, 32, 16, 16, 16, 16 },
    /* fine */ { 64, 48, 32, 16, 16, 16 },
    /* good */ { 80, 64, 48, 32, 16, 16 },
    /* best */ { 96, 80, 64, 48, 32, 16 }
};

static int const sinc_modes [4] = { 16, 116, 32, 132 }; /* taps + 100 * cubic */

/* Lists the passes converting `from` to `to` (planning direction).  Returns
 * the pass count, or -1 if upstream's tables would be overrun. */
static int list_passes (Drb_Edge const * from, Drb_Edge const * to, int quality, Drb_Pass * passes)
{
    int up = 0, down = 0, fractional = 0;

    if (from->rate < to->rate)
    {
        up = doublings(from->rate, to->rate);
        fractional = from->rate * (1 << up) != to->rate;
    }
    else
    {
        down = doublings(to->rate, from->rate);
        fractional = to->rate * (1 << down) != from->rate;
    }

    if (fractional && quality > 0) /* oversample around the fractional stage */
    {
        up++;
        down++;
    }

    int const multichannel = from->channels > 1;
    int const slicer = fractional || from->block < to->block;

    if (up >= 6 || down >= 6) return -1;
    if (2 * multichannel + up + down + fractional + slicer > drb_max_passes) return -1;

    int count = 0;

    if (multichannel) passes[count++] = (Drb_Pass){ drb_pass_interleave, 0 };

    /* longest filter at the lowest rate, on both sides */
    for (int i = 0; i < up; i++)
    {
        passes[count++] = (Drb_Pass){ drb_pass_up2x, half_band_orders[quality][i] };
    }

    if (fractional) passes[count++] = (Drb_Pass){ drb_pass_sinc, sinc_modes[quality] };
    if (slicer) passes[count++] = (Drb_Pass){ drb_pass_slicer, 0 };

    for (int i = down - 1; i >= 0; i--)
    {
        passes[count++] = (Drb_Pass){ drb_pass_down2x, half_band_orders[quality][i] };
    }

    if (multichannel) passes[count++] = (Drb_Pass){ drb_pass_deinterleave, 0 };

    return count;
}

/* ---- edge formats (drb-audio-converter.c:3079-3157 and the per-pass
 *      restrain functions 1413-1467, 1817-1893, 2366-2440, 2701-2753) ------- */

static void agree (int * a, int * b)
{
    if (*a == 0 && *b > 0) *a = *b;
    else if (*a > 0 && *b == 0) *b = *a;
}

static void scale_by_two (int * small, int * large)
{
    if (*small == 0 && *large > 0) *small = *large / 2;
    if (*small > 0 && *large == 0) *large = *small * 2;
}

static void settle (Drb_Pass pass, Drb_Edge * a, Drb_Edge * b)
{
    switch (pass.kind)
    {
      case drb_pass_interleave:
      case drb_pass_deinterleave:
        if (a->layout == 0) a->layout = pass.kind == drb_pass_interleave ? layout_planar : layout_interleaved;
        if (b->layout == 0) b->layout = pass.kind == drb_pass_interleave ? layout_interleaved : layout_planar;
        agree(&a->rate, &b->rate);
        agree(&a->channels, &b->channels);
        agree(&a->block, &b->block);
        agree(&a->max_blocks, &b->max_blocks);
        break;

      case drb_pass_up2x:
        scale_by_two(&a->rate, &b->rate);
        scale_by_two(&a->block, &b->block);
        agree(&a->channels, &b->channels);
        agree(&a->layout, &b->layout);
        agree(&a->max_blocks, &b->max_blocks);
        break;

      case drb_pass_down2x:
        scale_by_two(&b->rate, &a->rate);
        scale_by_two(&b->block, &a->block);
        agree(&a->channels, &b->channels);
        agree(&a->layout, &b->layout);
        agree(&a->max_blocks, &b->max_blocks);
        break;

      case drb_pass_sinc:
        if (b->block == 0) b->block = 1;

        if (a->rate > 0 && b->rate > 0 && a->block > 0 && b->block > 0)
        {
            if (a->max_blocks == 0 && b->max_blocks > 0)
            {
                /* most source frames that can never yield more target frames */
                long frames = (long)b->max_blocks * b->block;
                frames = frames * a->rate / b->rate;
                a->max_blocks = (int)(frames / a->block);
            }
            else if (a->max_blocks > 0 && b->max_blocks == 0)
            {
                long frames = (long)a->max_blocks * a->block;
                frames = (frames * b->rate + a->rate - 1) / a->rate;
                b->max_blocks = (int)(frames / b->block);
            }
        }

        agree(&a->channels, &b->channels);
        agree(&a->layout, &b->layout);
        break;

      case drb_pass_slicer:
        if (a->block > 0 && b->block > 0)
        {
            if (a->max_blocks == 0 && b->max_blocks > 0)
            {
                a->max_blocks = (int)((long)b->max_blocks * b->block / a->block);
            }
            else if (a->max_blocks > 0 && b->max_blocks == 0)
            {
                b->max_blocks = (int)(((long)a->max_blocks * a->block + b->block - 1) / b->block);
            }
        }

        agree(&a->rate, &b->rate);
        agree(&a->channels, &b->channels);
        agree(&a->layout, &b->layout);
        break;
    }
}

static void settle_edges (Drb_Plan * plan, Drb_Edge const * from, Drb_Edge const * to)
{
    int const n = plan->pass_count;

    memset(plan->edges, 0, sizeof(plan->edges));

    plan->edges[0] = *from;
    plan->edges[n] = *to; /* with no passes the far side wins, as upstream */

    if (plan->edges[0].channels == 1) plan->edges[0].layout = layout_interleaved;
    if (plan->edges[n].channels == 1) plan->edges[n].layout = layout_interleaved;

    for (int round = 0; round < 2; round++)
    {
        for (int i = 0; i < n; i++) settle(plan->passes[i], &plan->edges[i], &plan->edges[i + 1]);
        for (int i = n - 1; i >= 0; i--) settle(plan->passes[i], &plan->edges[i], &plan->edges[i + 1]);
    }

    if (n == 0)
    {
        plan->edges[0].max_blocks =
            from->max_blocks > to->max_blocks ? from->max_blocks : to->max_blocks;
    }
}

/* ---- sizes upstream reports ---------------------------------------------- */

static long align64 (long size) { return (size + 63) & ~63L; }

/* Bytes upstream's bump allocator hands out for a converter on an LP64 target
 * (drb-audio-converter.c:53-88, 3690-3753): a 128-byte header plus 40 bytes per
 * pass; per up-sampler a 16-byte header, one pointer and one double-length
 * ring per channel (2206-2254); per down-sampler the same with two rings per
 * channel (2295-2343); per fractional stage a 64-byte header, one pointer and
 * one ring per channel (1743-1793); per slicer a 64-byte header plus one block
 * (2676-2680).  Every allocation is rounded up to 64 bytes. */
static long upstream_converter_size (Drb_Plan const * plan)
{
    long const ch = plan->channels;
    long size = align64(128 + 40L * plan->pass_count);

    for (int i = 0; i < plan->pass_count; i++)
    {
        Drb_Pass const pass = plan->passes[i];
        long const ring = align64((pass.arg % 100) / (pass.kind == drb_pass_sinc ? 1 : 2) * 2 * 4L);

        switch (pass.kind)
        {
          case drb_pass_up2x:
            size += align64(16 + ch * 8) + ch * ring;
            break;
          case drb_pass_down2x:
            size += align64(16 + ch * 2 * 8) + ch * 2 * ring;
            break;
          case drb_pass_sinc:
            size += align64(64 + ch * 8) + ch * ring;
            break;
          case drb_pass_slicer:
          {
            long const a = plan->edges[i].block, b = plan->edges[i + 1].block;
            size += align64(64 + (a > b ? a : b) * ch * 4);
            break;
          }
        }
    }

    return size;
}

/* ---- resolve ------------------------------------------------------------- */

int drb_plan_resolve
    (
        int source_rate, int target_rate, int channels,
        int block_size, int max_block_count, int direction, int quality,
        Drb_Plan * plan
    )
{
    int const pull = direction == 1;

    /* the user's block structure sits on the target side for push and on the
     * source side for pull (drb-audio-converter.c:3926-3954) */
    Drb_Edge const source =
    {
        source_rate, channels, layout_planar,
        pull ? block_size : 1, pull ? max_block_count : 0
    };
    Drb_Edge const target =
    {
        target_rate, channels, layout_planar,
        pull ? 1 : block_size, pull ? 0 : max_block_count
    };

    if (!edge_is_valid(&source) || !edge_is_valid(&target)) return 0;
    if (quality < 0 || quality > 3) return 0;

    memset(plan, 0, sizeof(*plan));

    plan->source_rate = source_rate;
    plan->target_rate = target_rate;
    plan->channels = channels;
    plan->block_size = block_size;
    plan->max_block_count = max_block_count;
    plan->direction = direction;
    plan->quality = quality;

    /* pull plans the opposite conversion, user side first, then swaps every
     * pass for its inverse (drb-audio-converter.c:3191-3212, 2826-2859) */
    Drb_Edge const * const from = pull ? &target : &source;
    Drb_Edge const * const to = pull ? &source : &target;

    plan->pass_count = list_passes(from, to, quality, plan->passes);

    if (plan->pass_count < 0) return -1;

    settle_edges(plan, from, to);

    if (pull)
    {
        for (int i = 0; i < plan->pass_count; i++)
        {
            static int const inverse [] =
            {
                drb_pass_deinterleave, drb_pass_interleave,
                drb_pass_down2x, drb_pass_up2x, drb_pass_sinc, drb_pass_slicer
            };

            plan->passes[i].kind = inverse[plan->passes[i].kind];
        }
    }

    plan->throttling = plan->edges[0].block * plan->edges[0].max_blocks;

    long floats = 0;

    for (int i = 0; i <= plan->pass_count; i++)
    {
        long const need = (long)plan->edges[i].channels * plan->edges[i].block * plan->edges[i].max_blocks;
        if (need > floats) floats = need;
    }

    plan->buffer_length = align64(floats * 4) / 4;
    plan->converter_size = upstream_converter_size(plan);

    /* DSP stages in data-flow order */
    plan->sinc_stage = -1;

    double macs = 0.0;
    int granule = 1, odd_hazard = 0;

    for (int k = 0; k < plan->pass_count; k++)
    {
        int const i = pull ? plan->pass_count - 1 - k : k;
        Drb_Pass const pass = plan->passes[i];
        Drb_Edge const * const in = pull ? &plan->edges[i + 1] : &plan->edges[i];
        Drb_Edge const * const out = pull ? &plan->edges[i] : &plan->edges[i + 1];
        /* (de)interleaving is fused into the first load / last store: no stage, and
         * with eight stages in front of it no room for one either */
        if (pass.kind != drb_pass_up2x && pass.kind != drb_pass_down2x && pass.kind != drb_pass_sinc && pass.kind != drb_pass_slicer) continue;

        if (plan->stage_count >= drb_max_stages) return -1;

        Drb_Stage * const stage = &plan->stages[plan->stage_count];

        stage->in_rate = in->rate;
        stage->out_rate = out->rate;
        stage->arg = pass.arg;

        switch (pass.kind)
        {
          case drb_pass_up2x:
            stage->kind = drb_stage_up2x;
            macs += (double)(pass.arg / 2) * in->rate / target_rate;
            break;
          case drb_pass_down2x:
            stage->kind = drb_stage_down2x;
            macs += (double)(pass.arg / 2 + 1) * out->rate / target_rate;
            break;
          case drb_pass_sinc:
          {
            /* two (linear) or four (cubic) bank rows plus the blend */
            int const taps = pass.arg % 100;
            int const per_output = pass.arg >= 100 ? 4 * taps + 19 : 2 * taps + (taps == 32 ? 3 : 2);

            stage->kind = drb_stage_sinc;
            plan->sinc_stage = plan->stage_count;
            plan->source_delta = (long long)(0x1.0p48 / in->rate);
            plan->target_delta = (long long)(0x1.0p48 / out->rate);
            macs += (double)per_output * out->rate / target_rate;
            break;
          }
          case drb_pass_slicer:
            stage->kind = drb_stage_delay;
            stage->arg = in->block > out->block ? in->block : out->block;
            plan->slicer_block = stage->arg;
            break;
          default:
            break;
        }

        plan->stage_count++;
    }

    plan->macs_per_output_sample = macs;

    /* A 2x stage must never see an odd frame count.  Frame counts are
     * quantised by the slicer; follow the guaranteed divisor along the chain
     * from the side where the user hands in arbitrary counts. */
    for (int k = 0; k < plan->stage_count; k++)
    {
        Drb_Stage const * const stage = &plan->stages[pull ? plan->stage_count - 1 - k : k];
        int const doubling = stage->kind == (pull ? drb_stage_down2x : drb_stage_up2x);
        int const halving = stage->kind == (pull ? drb_stage_up2x : drb_stage_down2x);

        if (doubling) granule *= 2;
        if (halving) { if (granule % 2) odd_hazard = 1; else granule /= 2; }
        if (stage->kind == drb_stage_sinc) granule = 1;
        if (stage->kind == drb_stage_delay) granule = stage->arg;
    }

    plan->runnable = plan->throttling > 0 && !odd_hazard;

    return 1;
}

/* ---- per-stream frame arithmetic ----------------------------------------- */

void drb_stream_reset (Drb_Plan const * plan, Drb_Stream_State * state)
{
    state->phase = plan->source_delta;            /* drb-audio-converter.c:1765 */
    state->slicer_index = plan->slicer_block / 2; /* drb-audio-converter.c:2687 */
}

int drb_stream_step
    (
        Drb_Plan const * plan, Drb_Stream_State * state, int frames, Drb_Step * step
    )
{
    int const pull = plan->direction == 1;
    long long const sd = plan->source_delta, td = plan->target_delta;
    long long count = frames;

    step->phase = state->phase;
    step->delay = 0;

    /* push walks the chain with the data, pull against it */
    for (int k = 0; k < plan->stage_count; k++)
    {
        Drb_Stage const * const stage = &plan->stages[pull ? plan->stage_count - 1 - k : k];

        switch (stage->kind)
        {
          case drb_stage_up2x:
            if (pull) { if (count & 1) return 0; count /= 2; } else count *= 2;
            break;

          case drb_stage_down2x:
            if (pull) count *= 2; else { if (count & 1) return 0; count /= 2; }
            break;

          case drb_stage_sinc:
            if (pull)
            {
                /* source frames consumed for `count` more outputs
                 * (drb-audio-converter.c:1535-1538, 1677-1707) */
                long long const consumed =
                    count > 0 ? (state->phase - td + td * count) / sd : 0;

                state->phase += td * count - sd * consumed;
                count = consumed;
            }
            else
            {
                /* outputs released by `count` more source frames
                 * (drb-audio-converter.c:1516-1519, 1617-1648) */
                long long const produced =
                    count > 0 ? (td - state->phase + sd * count) / td : 0;

                state->phase += td * produced - sd * count;
                count = produced;
            }
            break;

          case drb_stage_delay:
          {
            /* whole blocks only; the block starts half full of silence
             * (drb-audio-converter.c:2495-2499, 2502-2652).  Seen from the
             * output side of the stage, sample j of this step is input sample
             * j - delay of this step. */
            int const block = stage->arg;
            long long const filled = state->slicer_index + count;

            step->delay = pull ? block - state->slicer_index : state->slicer_index;
            state->slicer_index = (int)(filled % block);
            count = filled / block * block;
            break;
          }
        }
    }

    step->source_frames = pull ? (int)count : frames;
    step->target_frames = pull ? frames : (int)count;

    return 1;
}

/* ---- kernel geometry ------------------------------------------------------ */

static long long floor_div (long long a, long long b)
{
    long long q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) q--;
    return q;
}

/* Lowest input index (relative to the step) stage `stage` reads to produce
 * output index `lo`, for the worst-case phase and delay. */
static long long reach_back (Drb_Plan const * plan, Drb_Stage const * stage, long long lo)
{
    switch (stage->kind)
    {
      case drb_stage_up2x: return floor_div(lo, 2) - stage->arg / 2 + 1;
      case drb_stage_down2x: return 2 * lo - stage->arg + 1;
      case drb_stage_delay: return lo - stage->arg;
      case drb_stage_sinc:
        return floor_div(lo * plan->target_delta, plan->source_delta) - 2 - stage->arg % 100;
    }
    return lo;
}

/* Most input frames stage `stage` touches to produce `width` output frames. */
static long long input_width (Drb_Plan const * plan, Drb_Stage const * stage, long long width)
{
    switch (stage->kind)
    {
      case drb_stage_up2x: return (width + 1) / 2 + stage->arg / 2 + 1;
      case drb_stage_down2x: return 2 * width + stage->arg;
      case drb_stage_delay: return width;
      case drb_stage_sinc:
        return (width * plan->target_delta + plan->source_delta - 1) / plan->source_delta
             + stage->arg % 100 + 3;
    }
    return width;
}

void drb_plan_geometry (Drb_Plan const * plan, int lds_budget_bytes, int channels_apart, Drb_Geometry * geometry)
{
    long long lo = 0;

    for (int k = plan->stage_count - 1; k >= 0; k--)
    {
        lo = reach_back(plan, &plan->stages[k], lo);
    }

    geometry->history_frames = (int)((-lo + 4 + 3) / 4 * 4);
    if (geometry->history_frames < 4) geometry->history_frames = 4;

    /* (2048 down to 16 in steps of 16 target frames, then halves) */
    int tiles [140];
    unsigned tile_count = 0;

    for (int t = 2048; t >= 16; t -= 16) tiles[tile_count++] = t;
    for (int t = 8; t >= 4; t /= 2) tiles[tile_count++] = t;

    {
        /* (diagnostic: this tile or none, tools/exact_tiles.sh) */
        char const * const forced = getenv("DRB_AMD_EXACT_TILE");

        if (forced && atoi(forced) > 0)
        {
            tile_count = 0;
            tiles[tile_count++] = atoi(forced);
            lds_budget_bytes = drb_lds_hard_limit;
        }
    }

    /* what the kernel keeps in LDS beside the two edge buffers: the fractional stage's
     * bank, rows skewed by one float, and the index ranges */
    int beside = 128;

    for (int k = 0; k < plan->stage_count; k++)
    {
        if (plan->stages[k].kind == drb_stage_sinc)
        {
            beside += (plan->stages[k].arg >= 100 ? 32 : 64) * (plan->stages[k].arg % 100 + 1) * 4;
        }
    }

    /* The largest tile within the budget (a third of a CU's LDS, less the allocation's
     * granule: measured, three workgroups per CU run 1.2 to 1.4 times faster than two on
     * a tile a fifth larger), all channels in one workgroup.  Chains that decimate many
     * channels a long way have none: their channels are dealt out to several
     * workgroups (half of them each, a quarter, one), and failing that a
     * workgroup takes its whole LDS. */
    for (int pass = 0; pass < 2; pass++)
    {
        int const limit = pass == 0 ? lds_budget_bytes : drb_lds_hard_limit;

        for (int share = 1; share <= 8; share *= 2)
        {
            int const group = (plan->channels + share - 1) / share;

            if (pass == 0 && channels_apart && group > 1) continue;

            for (unsigned t = 0; t < tile_count; t++)
            {
                if (pass == 0 && tiles[t] < 16) break;   /* (tiles that small are mostly look-back) */

                long long width = tiles[t], widest = tiles[t];

                for (int k = plan->stage_count - 1; k >= 0; k--)
                {
                    width = input_width(plan, &plan->stages[k], width);
                    if (width > widest) widest = width;
                }

                /* stages read whole 4-float groups, up to 15 floats past their range;
                 * two-plane edges split the width in halves of whole groups */
                widest = (widest + 16 + 7) / 8 * 8;

                geometry->tile_frames = tiles[t];
                geometry->lds_width = (int)widest;
                geometry->channel_group = group;
                geometry->lds_bytes = (int)(2 * widest * group * 4);

                if (geometry->lds_bytes + (pass == 0 ? beside : 0) <= limit) return;
            }

            if (group == 1) break;
        }
    }
}

double drb_plan_callback_latency (Drb_Plan const * plan, Drb_Stream_State const * state)
{
    double latency = 0.0;

    for (int i = 0; i < plan->pass_count; i++)
    {
        Drb_Pass const * const pass = &plan->passes[i];
        int const near = plan->edges[i].rate, far = plan->edges[i + 1].rate;

        switch (pass->kind)
        {
          case drb_pass_up2x:
          case drb_pass_down2x:
            latency += (double)(pass->arg / 2) / (double)(near > far ? near : far);
            break;

          case drb_pass_slicer:
          {
            int const a = plan->edges[i].block, b = plan->edges[i + 1].block;
            latency += ((a > b ? a : b) / 2.0) / (double)near;
            break;
          }

          case drb_pass_sinc:
          {
            double const scale = 1.0 / (double)plan->source_delta;
            double const own = (double)(pass->arg % 100) / 2.0 + 1.0;

            latency = plan->direction == 1 ? (double)state->phase * scale - own
                                           : (double)(state->phase - 1) * scale - own;
            break;
          }

          default:
            break;
        }
    }

    return latency;
}

/* ---- matrix-core geometry -------------------------------------------------- */

static long long gcd_ll (long long a, long long b)
{
    while (b) { long long const t = a % b; a = b; b = t; }
    return a;
}

static int half_band_table (int order)
{
    switch (order)
    {
      case 16: return 0;
      case 32: return 8;
      case 48: return 24;
      case 64: return 48;
      case 80: return 80;
      default: return 120;
    }
}

int drb_plan_fast_geometry (Drb_Plan const * plan, int lds_budget_bytes, Drb_Fast_Chain * chain)
{
    memset(chain, 0, sizeof(*chain));

    /* shape: at most one 2x stage on either side of a fractional stage */
    int up = -1, down = -1, sinc = -1, delay = -1;
    int pre [3], pre_count = 0;   /* up-samplers in front of the one next to the sinc stage, in plan order */

    for (int k = 0; k < plan->stage_count; k++)
    {
        switch (plan->stages[k].kind)
        {
          case drb_stage_up2x:
            if (sinc >= 0) return 0;
            if (up >= 0) { if (pre_count == 3) return 0; pre[pre_count++] = up; }
            up = k;
            break;
          case drb_stage_down2x: if (down >= 0) return 0; down = k; break;
          case drb_stage_sinc: sinc = k; break;
          case drb_stage_delay: delay = k; break;
        }
    }

    if (sinc >= 0 && down >= 0 && down < sinc) return 0;
    if (sinc < 0 && pre_count > 0) return 0;
    if (pre_count > 0 && delay >= 0 && delay < up) return 0;   /* (a slicer between the interpolators: not folded) */

    if (sinc < 0 && up >= 0)
    {
        /* one half-band interpolator on its own (a slicer may stand on either side) */
        if (down >= 0) return 0;
        if (delay >= 0 && delay != up - 1 && delay != up + 1) return 0;
        if (plan->stage_count != (delay >= 0 ? 2 : 1)) return 0;

        chain->kind = 2;
        chain->channels = plan->channels;
        chain->has_up = 1;
        chain->up_taps = plan->stages[up].arg / 2;
        chain->up_table = half_band_table(plan->stages[up].arg);
        chain->delay_place = delay < 0 ? 0 : delay < up ? 2 : 1;
        chain->seg_x = 128;   /* 8 row blocks: two per wave */
        chain->seg_y = 2 * chain->seg_x;
        chain->look_x = chain->up_taps;
        chain->waves = 4;
        chain->off_x = 0;
        chain->off_taps = (chain->seg_x + chain->look_x) * 16;
        chain->lds_floats = chain->off_taps + chain->up_taps + 32 + 32;
        chain->tile_frames = 16 * chain->seg_y;
        chain->usable = chain->lds_floats * 4 <= lds_budget_bytes;

        return 0;
    }

    if (sinc < 0)
    {
        /* one half-band decimator on its own (a slicer may stand on either side) */
        if (up >= 0 || down < 0) return 0;
        if (delay >= 0 && delay != down - 1 && delay != down + 1) return 0;
        if (plan->stage_count != (delay >= 0 ? 2 : 1)) return 0;

        chain->kind = 1;
        chain->channels = plan->channels;
        chain->has_down = 1;
        chain->down_taps = plan->stages[down].arg / 2;
        chain->down_table = half_band_table(plan->stages[down].arg);
        chain->delay_place = delay < 0 ? 0 : delay < down ? 2 : 1;
        chain->seg_y = 192;   /* 12 row blocks: three per wave */
        chain->seg_s = 2 * chain->seg_y;
        chain->look_d = chain->down_taps;
        chain->waves = 4;
        chain->off_d_odd = 0;
        chain->off_d_even = (chain->seg_y + chain->look_d) * 16;
        chain->off_taps = 2 * chain->off_d_even;
        chain->lds_floats = chain->off_taps + chain->down_taps + 32 + 32;
        chain->tile_frames = 16 * chain->seg_y;
        chain->usable = chain->lds_floats * 4 <= lds_budget_bytes;

        return 0;
    }
    if (delay >= 0 && delay != sinc + 1 && delay != sinc - 1) return 0;

    Drb_Stage const * const stage = &plan->stages[sinc];
    long long const g = gcd_ll(stage->in_rate, stage->out_rate);
    long long const ratio_in = stage->in_rate / g, ratio_out = stage->out_rate / g;

    /* period: a multiple of the exact rate ratio with at least 128 target frames */
    int multiple = 1;

    while (ratio_out * multiple < 128) multiple++;

    int const period_out = (int)(ratio_out * multiple), period_in = (int)(ratio_in * multiple);

    /* segment: whole periods, about 320 outputs of the sinc stage (160 where it
     * stands alone and outputs are target frames), halvable where a 2x stage follows */
    int const aim = up >= 0 || down >= 0 ? 320 : 160;
    int periods = (aim + period_out / 2) / period_out;

    if (periods < 1) periods = 1;

    while ((up >= 0 && (periods * period_in) % 2) || (down >= 0 && (periods * period_out) % 2)) periods++;

    int const taps = stage->arg % 100;

    chain->channels = plan->channels;
    chain->has_up = up >= 0;
    chain->up_taps = up >= 0 ? plan->stages[up].arg / 2 : 0;
    chain->up_table = up >= 0 ? half_band_table(plan->stages[up].arg) : 0;
    /* Further interpolators in front: only the collapsed form takes them in (the
     * geometry below is that of [up] sinc [down], its basis). */
    chain->pre_ups = pre_count;

    for (int i = 0; i < pre_count; i++)
    {
        Drb_Stage const * const outer = &plan->stages[pre[pre_count - 1 - i]];

        chain->pre_taps[i] = outer->arg / 2;
        chain->pre_table[i] = half_band_table(outer->arg);
    }

    chain->has_down = down >= 0;
    chain->down_taps = down >= 0 ? plan->stages[down].arg / 2 : 0;
    chain->down_table = down >= 0 ? half_band_table(plan->stages[down].arg) : 0;
    chain->sinc_taps = taps;
    chain->sinc_cubic = stage->arg >= 100;
    chain->sinc_table = stage->arg == 16 ? 168 : stage->arg == 32 ? 1192 : stage->arg == 116 ? 3240 : 3752;
    chain->sinc_pull = plan->direction == 1;
    chain->delay_place = delay < 0 ? 0 : delay == sinc + 1 ? 1 : 2;
    chain->period_in = period_in;
    chain->period_out = period_out;
    chain->periods = periods;
    chain->blocks = (period_out + 15) / 16;
    chain->source_delta = plan->source_delta;
    chain->target_delta = plan->target_delta;
    chain->scale = 1.0 / (double)plan->source_delta;

    /* a 16-row block spans ceil(15 in/out) + 1 source frames of window ends,
     * plus the taps, plus one for the even alignment of the window start */
    int const spread = (int)((15 * ratio_in + ratio_out - 1) / ratio_out) + 1;

    chain->ksteps = (spread + taps + 1 + 3) / 4;

    if (chain->ksteps > 16) return 0;

    chain->seg_u = periods * period_in;
    chain->seg_s = periods * period_out;
    chain->seg_x = up >= 0 ? chain->seg_u / 2 : chain->seg_u;
    chain->seg_y = down >= 0 ? chain->seg_s / 2 : chain->seg_s;
    /* Window ends of a segment's rows run from -2 to seg_u - 1, so the sinc
     * stage reaches back taps + 4 of its input rows.  Behind an up-sampler its
     * input lives in two planes indexed by source row: the source itself
     * (even samples, u[2 f] = x[f - N / 2]) and the up-sampler's output (odd
     * samples); look_u then counts rows of the odd plane. */
    chain->look_d = down >= 0 ? chain->down_taps : 0;

    if (up >= 0)
    {
        int const reach = (taps + 4) / 2 + 2;
        int const even_reach = reach + chain->up_taps / 2;

        chain->look_u = (reach + 1) & ~1;
        chain->look_x = ((chain->up_taps > even_reach ? chain->up_taps : even_reach) + 1) & ~1;
    }
    else
    {
        chain->look_u = taps + 4;
        chain->look_x = chain->look_u;
    }

    /* the fixed-point accumulator is not exactly periodic: all 16 segments use
     * the phases of the middle one, which is off by at most this much (in
     * source frames of the sinc stage) at the outermost segments */
    double const exact = (double)plan->source_delta * (double)ratio_in / (double)ratio_out;
    double const drift = ((double)plan->target_delta - exact) / (double)plan->source_delta;
    double const worst = 7.5 * chain->seg_s * (drift < 0 ? -drift : drift);

    /* beyond the accuracy budget the kernel adds a first-order term per segment
     * (a second product with the coefficients' derivatives); that one is good
     * for far more than any of the 13 x 13 rate pairs needs */
    if (worst > 2.0e-6) return 0;

    chain->column_drift = worst > 1.0e-7 ? (float)(chain->seg_s * drift) : 0.0f;

    /* LDS map.  Every buffer is [row][16 segments]; a row is 16 floats, so
     * consecutive rows alternate between the two halves of the 32 banks.  In
     * the sinc stage one half of a wave's lanes reads the source plane and the
     * other half the odd plane, at rows of equal parity: the odd plane is
     * placed an odd number of rows away so that the two never share banks. */
    int offset = 0;

    chain->off_x = offset;
    offset += (chain->seg_x + chain->look_x) * 16;

    if (up >= 0)
    {
        /* rows read together: source row t - N / 2 + look_x and odd row t + look_u */
        if (((chain->up_taps / 2 + chain->look_x + chain->look_u + offset / 16) & 1) == 0) offset += 16;

        chain->off_u = offset;
        offset += (chain->seg_x + chain->look_u) * 16;
    }
    else
    {
        chain->off_u = chain->off_x;
    }

    if (down >= 0)
    {
        chain->off_d_odd = offset;
        offset += (chain->seg_s / 2 + chain->look_d) * 16;
        chain->off_d_even = offset;
        offset += (chain->seg_s / 2 + chain->look_d) * 16;
    }

    /* The sinc bank, [tap][row] with 65 floats between taps, is only read while
     * coefficients are rebuilt, when the planes of d are idle: it shares them. */
    {
        int const bank = taps * 65 + 8;

        if (down >= 0 && bank <= offset - chain->off_d_odd)
        {
            chain->off_bank = chain->off_d_odd;
        }
        else
        {
            chain->off_bank = offset;
            offset += bank;
        }
    }

    /* half-band taps between 16 zeros on either side (what a banded 16-row
     * block multiplies by depends on tap index minus row only) */
    chain->off_taps = offset;
    offset += (chain->up_taps + 32) + (chain->down_taps + 32);
    chain->lds_floats = offset + 32;

    if (chain->lds_floats * 4 > lds_budget_bytes) return 0;

    /* Waves per workgroup.  Every stage deals whole 16-row blocks to the waves
     * and ends in a barrier, so a stage takes as long as the wave with most
     * blocks.  Five waves would deal the usual 10 + 20 + 10 blocks out evenly,
     * but measured slower on MI355X: a CU has four SIMDs, the fifth wave doubles
     * up on one of them and fewer workgroups fit.  Row blocks of the sinc stage
     * go to the waves in runs, and a wave keeps coefficients for at most three
     * block types. */
    {
        int const chains = chain->blocks * chain->periods;
        int const up_blocks = up >= 0 ? (chain->seg_x + 15) / 16 : 0;
        int const down_blocks = down >= 0 ? (chain->seg_y + 15) / 16 : 0;
        int const up_steps = up >= 0 ? (chain->up_taps + 16) / 4 : 0;
        int const down_steps = down >= 0 ? (chain->down_taps + 16) / 4 : 0;
        long best = 0;

        chain->waves = 0;

        for (int waves = 4; waves <= 4; waves++)
        {
            int const share = (chains + waves - 1) / waves;
            int types = 0;

            for (int w = 0; w < waves; w++)
            {
                int const first = w * share, last = first + share < chains ? first + share : chains;

                if (first < last)
                {
                    int const span = (last - 1) / chain->periods - first / chain->periods + 1;
                    if (span > types) types = span;
                }
            }

            if (types > 3) continue;
            if ((chain->seg_x + chain->look_x + 16 * waves - 1) / (16 * waves) > 4) continue;

            long const cost = (long)((up_blocks + waves - 1) / waves) * up_steps
                            + (long)share * chain->ksteps
                            + (long)((down_blocks + waves - 1) / waves) * down_steps;

            if (chain->waves == 0 || cost < best)
            {
                chain->waves = waves;
                chain->types_per_wave = types;
                best = cost;
            }
        }

        if (chain->waves == 0) return 0;
    }

    chain->tile_frames = 15 * chain->seg_y;
    chain->usable = 1;

    /* warm-up segment plus its own lookback, seen from the first target frame */
    int const reach = chain->seg_x + chain->look_x + (up >= 0 ? chain->up_taps : 0) + taps + 8;

    return (reach + 3) / 4 * 4;
}

/* ---- collapsed chains -------------------------------------------------------- */

/* Source frames (of the sinc stage) consumed before sinc output `nu`, counted
 * from the anchor, whose phase is `rest` past a whole source frame
 * (drb-audio-converter.c:1617-1648 push, 1677-1707 pull). */
static long long collapse_window_end (Drb_Fast_Chain const * chain, long long rest, long long nu)
{
    long long const phase = rest + nu * chain->target_delta;
    long long q = floor_div(phase, chain->source_delta);

    if (!chain->sinc_pull && phase - q * chain->source_delta == 0) q -= 1;

    return q;
}

/* First and one-past-last source frame (relative to the anchor's own) that the
 * 16 target frames of every row block depend on. */
static void collapse_extents (Drb_Fast_Chain const * chain, long long rest, int parity, int * lo, int * hi)
{
    for (int b = 0; b < chain->blocks; b++)
    {
        int const rho0 = 16 * b, rho1 = 16 * b + 15 < chain->seg_y ? 16 * b + 15 : chain->seg_y - 1;
        long long const nu_lo = chain->has_down ? 2 * (rho0 - chain->down_taps) + 1 : rho0;
        long long const nu_hi = chain->has_down ? 2 * rho1 - 1 : rho1;
        long long const e_first = parity + collapse_window_end(chain, rest, nu_lo) - chain->sinc_taps;
        long long const e_last = parity + collapse_window_end(chain, rest, nu_hi) - 1;

        long long first = chain->has_up ? floor_div(e_first, 2) - chain->up_taps + 1 : e_first;
        long long last = chain->has_up ? floor_div(e_last - 1, 2) : e_last;

        for (int i = 0; i < chain->pre_ups; i++)
        {
            first = floor_div(first, 2) - chain->pre_taps[i] + 1;
            last = floor_div(last - 1, 2);
        }

        lo[b] = (int)first;
        hi[b] = (int)last + 1;
    }
}

int drb_plan_collapse_geometry (Drb_Fast_Chain * chain, int lds_budget_bytes, int drifting_columns)
{
    if (!chain->usable || chain->kind != 0) return 0;
    if (chain->column_drift != 0.0f && !drifting_columns) return 0;

    Drb_Fast_Chain c = *chain;

    /* segment: whole periods of the exact ratio, 128 .. 192 target frames */
    long long const g = gcd_ll(c.period_in, c.period_out);
    int const ratio_in = (int)(c.period_in / g), ratio_out = (int)(c.period_out / g);
    int m = 0;

    int const ups = (c.has_up ? 1 : 0) + c.pre_ups;

    /* Among the candidates the one whose last row block is fullest (a row block
     * costs a wave whether 16 or 1 of its target frames exist), at most ten
     * blocks where that is possible (the split kernel's limit). */
    double best = 0.0;

    for (int k = 1; k <= 4096; k++)
    {
        int const seg_s = ratio_out * k, seg_u = ratio_in * k;
        int const seg_y = c.has_down ? seg_s / 2 : seg_s;

        if ((c.has_down && (seg_s & 1)) || (seg_u & ((1 << ups) - 1))) continue;
        if (seg_y > 192) break;
        /* (a folded chain of interpolators has short source segments: only odd ones,
         * whose plane is a straight copy of the source) */
        if (c.pre_ups > 0 && !((seg_u >> ups) & 1)) continue;
        if (seg_y < 128) continue;
        /* (an even source segment makes a padded plane, which takes segments of 128 frames
         * and more: a shorter even one would be chosen for nothing) */
        if (!((seg_u >> ups) & 1) && (seg_u >> ups) < 128) continue;

        int const blocks = (seg_y + 15) / 16;
        double const fill = (double)seg_y / (16.0 * blocks) * (blocks <= drb_split_max_blocks ? 1.0 : 0.01);

        if (fill > best + 1e-9) { best = fill; m = k; }
    }

    if (!m) return 0;

    c.kind = 3;
    c.periods = m;
    c.period_in = ratio_in;
    c.period_out = ratio_out;
    c.seg_s = ratio_out * m;
    c.seg_u = ratio_in * m;
    c.seg_y = c.has_down ? c.seg_s / 2 : c.seg_s;
    c.seg_x = c.seg_u >> ups;
    c.blocks = (c.seg_y + 15) / 16;

    if (c.blocks > drb_collapse_max_blocks) return 0;

    /* widest a row block's window gets for any phase: between its first and its
     * last sinc output the stage consumes floor(d nu ratio) or one more frame */
    long long const nu_block = c.has_down ? 2 * 15 - 1 - (1 - 2 * c.down_taps) : 15;
    long long const nu_all = c.has_down ? 2 * (c.seg_y - 1) - 1 - (1 - 2 * c.down_taps) : c.seg_y - 1;
    long long const span_block = floor_div(nu_block * c.target_delta, c.source_delta) + 1 + c.sinc_taps - 1;
    long long const span_all = floor_div(nu_all * c.target_delta, c.source_delta) + 1 + c.sinc_taps - 1;
    int width = c.has_up ? (int)(span_block >> 1) + c.up_taps : (int)span_block + 1;
    int rows = c.has_up ? (int)(span_all >> 1) + c.up_taps : (int)span_all + 1;

    /* (L frames behind an up-sampler of N taps come from at most L / 2 + N in front of it) */
    for (int i = 0; i < c.pre_ups; i++)
    {
        width = (width >> 1) + c.pre_taps[i];
        rows = (rows >> 1) + c.pre_taps[i];
    }

    c.ksteps = (width + 3) / 4 | 1;   /* kernel images exist for odd counts */
    c.window = width;
    c.look_x = (rows - c.seg_x + 3) / 4 * 4;
    if (c.look_x < 4) c.look_x = 4;
    c.look_u = c.look_d = 0;

    c.waves = c.blocks % 5 == 0 || c.blocks == 9 ? 5 : 4;
    c.types_per_wave = (c.blocks + c.waves - 1) / c.waves;   /* row blocks per wave */

    /* one plane of source rows, [row][16 segments], and a few rows behind it that
     * only zero coefficients ever meet */
    c.off_x = 0;
    c.lds_floats = (c.seg_x + c.look_x + 8) * 16;
    c.tile_frames = 16 * c.seg_y;

    /* Planar float32 sources: a plane in LDS is a straight copy of the stretch of
     * the source that the 16 segments of a tile need, from a 128-byte line on
     * (up to 31 frames in front), in 10, 12 or 16 wave-wide loads of 256 frames
     * (the kernel images there are); three planes take turns.  plane_rows: rows
     * of one segment; plane_stride: floats of a plane. */
    c.plane_rows = (c.seg_x + c.look_x + 8 + 3) / 4 * 4;
    c.plane_stride = 15 * c.seg_x + c.plane_rows + 31 <= 10 * 256 ? 10 * 256
                   : 15 * c.seg_x + c.plane_rows + 31 <= 12 * 256 ? 12 * 256 : 16 * 256;

    if (15 * c.seg_x + c.plane_rows + 31 <= c.plane_stride && 3 * c.plane_stride > c.lds_floats) c.lds_floats = 3 * c.plane_stride;

    if (c.lds_floats * 4 > lds_budget_bytes) return 0;

    *chain = c;

    return 1;
}

/* The collapsed form on the half-precision matrix instruction (kind 4, see
 * drb_hip.h): the same segments and row blocks; a window is 3 or 4 steps of 32
 * source frames, one wave per row block. */
int drb_plan_split_geometry (Drb_Fast_Chain * chain, int lds_budget_bytes)
{
    if (chain->kind != 3 || chain->window < 1) return 0;

    Drb_Fast_Chain c = *chain;

    c.chunks = (c.window + 31) / 32;

    if (c.chunks < 3) c.chunks = 3;
    if (c.chunks > 4 || c.blocks > drb_split_max_blocks) return 0;

    c.kind = 4;
    c.ksteps = 8 * c.chunks;          /* a window is 4 * ksteps rows, as for kind 3 */
    c.waves = c.blocks;
    c.types_per_wave = 1;
    c.column_drift = 0.0f;            /* (every segment gets its own phase anyway) */

    /* a window may not start behind the row that leaves it whole inside the segment's rows */
    while (c.seg_x + c.look_x + 8 - 4 * c.ksteps < 0) c.look_x += 4;

    c.plane_rows = (c.seg_x + c.look_x + 8 + 3) / 4 * 4;


    int const reach = 15 * c.seg_x + c.plane_rows + 31;

    c.plane_stride = reach <= 10 * 256 ? 10 * 256 : reach <= 12 * 256 ? 12 * 256 : 0;

    /* (short source segments of a folded chain of interpolators: fewer loads a tile) */
    if (c.pre_ups > 0) c.plane_stride = reach <= 4 * 256 ? 4 * 256 : reach <= 6 * 256 ? 6 * 256 : c.plane_stride;

    if (!c.plane_stride) return 0;

    /* The 16 columns of an operand read must fall into 16 different LDS banks: they
     * do where the segments lie an odd number of words apart.  A straight copy of
     * the source has them seg_x apart; for even seg_x the plane has a word of
     * padding in front of every segment but the first, seg_x + 1 words from
     * segment to segment.  A window that runs across the padding has a zero
     * coefficient there and one frame less: it may cross one padding only. */
    if (c.seg_x & 1)
    {
        c.apart_stride = 0;
        c.plane_words = c.plane_stride;
    }
    else
    {
        if (c.plane_rows > 2 * c.seg_x || c.window > 32 * c.chunks - 1) return 0;

        /* (every frame of a tile's loads is written: 36 words in front of segment 0's
         * row 0 for the up to 31 frames the loads begin in front of it, a word of
         * padding for each of the up to 20 segments they reach, see drb_split.inc) */
        c.apart_stride = c.seg_x + 1;
        c.plane_words = (36 + c.plane_stride + 20 + 3) / 4 * 4;
    }

    /* The images of a tile's results: a lane writes one target frame of four
     * segments, the four groups of lanes 4 segments apart.  With the segments
     * 4 (mod 8) words apart, two groups share a bank at most (which a 32-bit
     * store does not notice).  Segments of whole 16-byte runs only: a lane of the
     * waves that store a tile reads four consecutive frames as one run. */
    c.image_seg = c.seg_y;

    if ((c.seg_y & 3) == 0) while ((c.image_seg & 7) != 4) c.image_seg += 4;

    /* planes [3][channels of a workgroup], then two images [channels] of a tile's
     * target frames; a workgroup takes both channels of an interleaved stereo
     * stream, else one channel */
    int const together = c.layout == 1 && c.channels == 2 ? 2 : 1;

    c.off_x = 0;
    c.lds_floats = drb_split_planes * together * c.plane_words + 2 * together * 16 * c.image_seg;

    if (c.lds_floats * 4 > lds_budget_bytes) return 0;

    *chain = c;

    return 1;
}

void drb_plan_collapse_key
    (
        Drb_Fast_Chain const * chain, long long phase, int delay, long long anchor_segment,
        long long * rest, int * parity, long long * frame
    )
{
    int const delay_behind = chain->delay_place == 1 ? delay : 0;
    int const delay_front = chain->delay_place == 2 ? delay : 0;
    long long const first = anchor_segment * chain->seg_y;
    long long const nu = (chain->has_down ? 2 * first : first) - delay_behind;
    long long const at = phase + nu * chain->target_delta;
    long long const q = floor_div(at, chain->source_delta);
    long long const mu = q - delay_front;

    *rest = at - q * chain->source_delta;
    /* (behind n up-samplers: where in a group of 2^n sinc-input frames the window ends) */
    int const ups = (chain->has_up ? 1 : 0) + chain->pre_ups;

    *parity = (int)(mu & ((1LL << ups) - 1));
    *frame = floor_div(mu, 1LL << ups);
}

int drb_plan_collapse_job (Drb_Fast_Chain const * chain, long long rest, int parity, Drb_Collapse_Job * job)
{
    int lo [drb_collapse_max_blocks] = { 0 }, hi [drb_collapse_max_blocks] = { 0 };

    memset(job, 0, sizeof(*job));

    job->rest = rest;
    job->parity = parity;

    collapse_extents(chain, rest, parity, lo, hi);

    job->origin = hi[chain->blocks - 1] - chain->seg_x;   /* row look_x of the plane */

    for (int b = 0; b < chain->blocks; b++)
    {
        /* (a window that would run past the plane starts earlier instead: the
         * last block of a segment may have few rows and a short span) */
        int const last_start = chain->seg_x + chain->look_x + 8 - 4 * chain->ksteps;
        int const start = lo[b] - job->origin + chain->look_x;

        job->first_row[b] = start < last_start ? start : last_start;

        /* (a padded plane's window gives one slot to the padding) */
        if (job->first_row[b] < 0 || hi[b] - lo[b] > 4 * chain->ksteps - (chain->apart_stride ? 1 : 0)) return 0;
    }

    return 1;
}
