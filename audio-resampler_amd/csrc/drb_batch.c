/*
 * The batched many-stream entry point: host scheduler + per-stream state.
 *
 * Per stream the host keeps what the upstream converter keeps in its passes
 * that is not sample data - the fractional stage's accumulator and the
 * slicer's fill level (drb-audio-converter.c:1494-1505, 2476-2484) - and the
 * device keeps the sample data: the last `history` source frames, which is
 * everything the filters still need from the past.  A step computes, on the
 * host, how many frames every stream consumes and produces (closed forms,
 * drb_plan.c), sends one small descriptor per stream to the device and
 * launches two kernels: the cascade and the history update.
 */
#include "drb_audio_converter_amd.h"
#include "drb_hip.h"
#include "drb_plan.h"

#include <stdarg.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* host allocations are counted with the device's (drb_hip_allocation_calls) */
static void * counted_calloc (size_t count, size_t size) { drb_hip_count_allocation(); return calloc(count, size); }
static void * counted_malloc (size_t size) { drb_hip_count_allocation(); return malloc(size); }
static void counted_free (void * pointer) { if (pointer) drb_hip_count_allocation(); free(pointer); }

enum { ring_slots = 4, max_step_frames = 1 << 24, fault_words = 128, host_slots = 3 };
enum { fault_sample_range = 0x100 };   /* bit of the kernels' fault word: see fetch_fault */

enum { max_launches = drb_max_stages };

static size_t const collapse_set_bytes_limit = (size_t)1 << 30;   /* of coefficient sets at a time */

/* Where the distinct phases of a step need more coefficient sets than that (or
 * than the static workspace holds), the step's streams are dealt with in groups,
 * one after the other on the step's HIP stream: the sets of a group are built,
 * its streams filtered, then the next group's sets take their place. */
enum { max_groups = 256 };

typedef struct Collapse_Group
{
    int first_stream, stream_count;
    int first_job, job_count;         /* in the step's job table; a stretch's job index counts from first_job */
}
Collapse_Group;
/* pinned, per ring slot, from creation on: the stretch table and jobs of any step of a few
 * streams, of a modest step of many (a batch of one stream, which is what the
 * drop-in converters are, does not pin a megabyte it can never fill) */
static size_t static_host_bytes_for (size_t streams)
{
    size_t const wanted = ((size_t)16 << 10) + streams * ((size_t)4 << 10);
    size_t const most = (size_t)256 << 10;

    return wanted < most ? wanted : most;
}

typedef struct Fast_Launch
{
    Drb_Fast_Chain fast;              /* geometry of this launch                                  */
    Drb_Fast_Chain staged;            /* where `fast` is a collapsed form: the three stages kept apart,
                                         for steps with too many different phases for coefficient sets */
    int has_staged;
    Drb_Device_Chain chain;           /* for the history kernel of launches behind the first      */
    int ups_before, downs_behind;     /* 2x stages of the plan in front of / behind this launch   */
    int halves;                       /* 1: a decimator on its own (target = source / 2)          */
    int doubles;                      /* 1: an interpolator on its own                            */
    int has_sinc;
    int converts;                     /* 1: changes the buffer format only (the caller's <-> planar float32), frame for frame:
                                         fast.sample_type / layout -> fast.target_sample_type / target_layout */
    double rate_ratio;                /* this launch's target rate over the chain's               */
    float * history [2];              /* device, launches behind the first                        */
    float * output;                   /* device, planar float32 [stream][channel][output_frames]  */
    long long output_frames;
}
Fast_Launch;

struct DrbAmd_Batch
{
    DrbAmd_Batch_Config config;
    Drb_Plan plan;
    Drb_Geometry geometry;
    Drb_Device_Chain chain;
    /* matrix-core path: one launch for [up] sinc [down] and for a single 2x
     * stage, or one per group of stages for longer chains, with planar float32
     * buffers of the batch's own between them */
    Fast_Launch launches [max_launches];
    int launch_count;                 /* 0: fast arithmetic runs the cascade kernel */
    int device;
    int sample_bytes;

    Drb_Stream_State * states;        /* host, one per stream                  */
    Drb_Stream_State * trial_states;  /* host scratch for two-phase commit     */
    Drb_Step * steps;                 /* host scratch                          */
    void const ** dense_source;       /* host scratch for the dense entry      */
    void ** dense_target;
    int * dense_frames;
    int * dense_capacity;
    int * dense_produced;
    int * feed_counts;                /* host scratch for the host-fed entry   */

    float * history [2];              /* device, [stream][channel][history]    */
    int history_current;

    Drb_Device_Call * calls_device;
    Drb_Device_Call * calls_host [ring_slots]; /* pinned                       */
    void * slot_done [ring_slots];             /* events                       */
    int slot_pending [ring_slots];
    int slot;

    int * fault_device;
    int * fault_host;                 /* pinned                                */
    /* One block of pinned memory per batch holds the fault word, the ring's step
     * descriptors and its tables (a process has some 65 000 memory mappings, and every
     * pinned allocation is one: nine of them per batch ended at 7 271 batches). */
    void * pinned_block;
    int collapse_host_own [ring_slots];   /* 1: collapse_host[i] outgrew the block and is an allocation of its own */

    /* host-fed entry: three slices of streams in flight */
    void * feed_streams [3];          /* copy in, convert, copy out            */
    void * feed_source [host_slots], * feed_target [host_slots];   /* device   */
    size_t feed_source_bytes, feed_target_bytes;                   /* per slot */
    void * feed_in [host_slots], * feed_done [host_slots], * feed_out [host_slots]; /* events */

    /* collapsed launch (the one with the sinc stage, if its geometry allows):
     * coefficient sets for the distinct phases of a step, and which one every
     * stretch of every stream uses */
    void * pending_device [3]; void const * pending_host [3]; size_t pending_bytes [3];   /* tables of the step at hand still to be uploaded */
    Drb_Collapse_Job * collapse_jobs;          /* device                       */
    Drb_Collapse_Spot * collapse_spots;        /* device                       */
    unsigned int * collapse_sets;              /* device                       */
    size_t collapse_jobs_room, collapse_spots_room, collapse_sets_room;   /* elements / words */
    void * collapse_host [ring_slots];         /* pinned: spots, then jobs     */
    size_t collapse_host_bytes [ring_slots];
    int * collapse_hash;                       /* host scratch                 */
    size_t collapse_hash_room;
    Collapse_Group groups [max_groups];        /* of the step at hand          */
    int group_count;
    int step_walkers;       /* the most recent collapsed launch: workgroups that went from stream to stream (0: none) */
    size_t set_bytes_limit;                    /* collapse_set_bytes_limit, or DRB_AMD_SET_BYTES_LIMIT (diagnostic) */
    int wish_apart;                            /* DRB_AMD_FAST_APART (diagnostic): 1 always the staged kernels, 0 never, -1 unset */

    int timing;                       /* bracket the cascade kernel with events */
    void * kernel_start, * kernel_stop;
    void * step_start, * step_stop;   /* ... and everything a step runs on the device */
    int kernel_timed;

    void * last_stream;               /* stream of the step enqueued last, and whether there was one */
    int last_slot;

    /* the library's static device workspace on this batch's GPU, and whether the
     * call in progress must live within it (drb_amd_batch_process_dense_static) */
    void * static_base;
    size_t static_bytes;
    int no_alloc;
};

/* ---- where the host's time goes (DRB_AMD_HOST_PROFILE=1 prints at destroy) ---- */

#include <time.h>

static double host_spent [6];
static long host_steps;

static double host_now (void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec * 1e6 + (double)t.tv_nsec * 1e-3;
}

#define HOST_MARK(section) do { double const now_ = host_now(); host_spent[section] += now_ - host_mark; host_mark = now_; } while (0)

/* ---- errors ---------------------------------------------------------------- */

static _Thread_local char last_error [256];

static DrbAmd_Status fail (DrbAmd_Status status, char const * format, ...)
{
    va_list arguments;
    va_start(arguments, format);
    vsnprintf(last_error, sizeof(last_error), format, arguments);
    va_end(arguments);
    return status;
}

static DrbAmd_Status fail_hip (char const * what, int code)
{
    return fail(drb_amd_error_device, "%s: %s", what, drb_hip_error_text(code));
}

char const * drb_amd_last_error (void) { return last_error; }

static DrbAmd_Status fetch_fault (DrbAmd_Batch * batch, void * hip_stream);

/* ---- plan -> device chain --------------------------------------------------- */

static int table_offset (int kind, int arg)
{
    /* float offsets into data/drb_tables.bin */
    if (kind == drb_stage_sinc)
    {
        switch (arg)
        {
          case 16: return 168;
          case 32: return 1192;
          case 116: return 3240;
          default: return 3752;
        }
    }

    switch (arg)
    {
      case 16: return 0;
      case 32: return 8;
      case 48: return 24;
      case 64: return 48;
      case 80: return 80;
      default: return 120;
    }
}

static void build_chain (DrbAmd_Batch_Config const * config, Drb_Plan const * plan,
                         Drb_Geometry const * geometry, Drb_Device_Chain * chain)
{
    memset(chain, 0, sizeof(*chain));

    chain->channels = plan->channels;
    chain->sinc_pull = plan->direction == 1;
    chain->delay_edge = -1;
    chain->history = geometry->history_frames;
    chain->tile_frames = geometry->tile_frames;
    chain->lds_width = geometry->lds_width;
    chain->channel_group = geometry->channel_group > 0 ? geometry->channel_group : plan->channels;
    chain->sample_type = config->sample_type;
    chain->layout = config->layout;
    chain->source_delta = plan->source_delta > 0 ? plan->source_delta : 1;
    chain->target_delta = plan->target_delta > 0 ? plan->target_delta : 1;
    chain->scale = 1.0 / (double)chain->source_delta; /* drb-audio-converter.c:1767 */

    /* the fractional stage's rates back from the deltas (2^48 / rate, rounded down), and
     * the period of their ratio in output frames */
    if (plan->source_delta > 0 && plan->target_delta > 0)
    {
        long long const in_rate = llround(0x1.0p48 / (double)plan->source_delta);
        long long const out_rate = llround(0x1.0p48 / (double)plan->target_delta);
        long long a = in_rate, b = out_rate;

        while (b > 0) { long long const r = a % b; a = b; b = r; }

        long long const period = a > 0 ? out_rate / a : 0;

        chain->sinc_period = period >= 1 && period <= 2048 ? (int)period : 0;
    }

    for (int k = 0; k < plan->stage_count; k++)
    {
        Drb_Stage const * const stage = &plan->stages[k];

        if (stage->kind == drb_stage_delay)
        {
            chain->delay_edge = chain->stage_count;
            continue;
        }

        Drb_Device_Stage * const out = &chain->stages[chain->stage_count++];

        out->kind = stage->kind;
        out->taps = stage->kind == drb_stage_sinc ? stage->arg % 100 : stage->arg / 2;
        out->cubic = stage->kind == drb_stage_sinc && stage->arg >= 100;
        out->table = table_offset(stage->kind, stage->arg);
    }
}

/* one channel per workgroup of the exact kernel: chains without a fractional stage (its
 * position and coefficients are what channels share) in planar buffers (a workgroup that
 * takes one channel of interleaved frames reads every frame for a fraction of it) */
static int channels_apart (DrbAmd_Batch_Config const * config, Drb_Plan const * plan)
{
    if (config->layout != drb_amd_layout_planar || plan->channels < 2) return 0;

    for (int k = 0; k < plan->stage_count; k++)
    {
        if (plan->stages[k].kind == drb_stage_sinc) return 0;
    }

    return 1;
}

enum { lds_budget = 53500, fast_lds_budget = 78 * 1024, split_lds_budget = 160 * 1024 };

static DrbAmd_Status resolve (DrbAmd_Batch_Config const * config, Drb_Plan * plan)
{
    int const status = drb_plan_resolve
    (
        config->source_sampling_rate, config->target_sampling_rate, config->channel_count,
        config->block_size, config->max_block_count, config->direction, config->quality, plan
    );

    if (status == 0) return fail(drb_amd_error_invalid_conversion, "conversion rejected by validation");
    if (status < 0) return fail(drb_amd_error_unsupported, "conversion needs more 2x stages than the filter tables hold");
    if (!plan->runnable)
    {
        return fail(drb_amd_error_unsupported,
                    plan->throttling == 0 ? "max_block_count leaves no room for a single sub-block"
                                          : "a 2x stage would see odd frame counts (block_size 1 without a slicer)");
    }
    {
        Drb_Geometry geometry;

        drb_plan_geometry(plan, lds_budget, channels_apart(config, plan), &geometry);

        if (geometry.lds_bytes > drb_lds_hard_limit)
        {
            return fail(drb_amd_error_unsupported, "the chain's look-back times %d channels does not fit a workgroup's LDS", plan->channels);
        }
    }
    if ((unsigned)config->sample_type > drb_amd_sample_i32) return fail(drb_amd_error_bad_argument, "unknown sample type");
    if ((unsigned)config->layout > drb_amd_layout_interleaved) return fail(drb_amd_error_bad_argument, "unknown layout");
    if ((unsigned)config->arithmetic > drb_amd_arithmetic_fast) return fail(drb_amd_error_bad_argument, "unknown arithmetic mode");

    return drb_amd_ok;
}


/* ---- plan -> launches of the matrix-core path -------------------------------- */

/* Cuts the plan's stages into launches: every 2x up-sampler but the one next to
 * the sinc stage runs on its own, then [up] [slicer] sinc [slicer] [down] in one
 * launch, then every further down-sampler on its own; chains without a sinc
 * stage are 2x stages only, one launch each (the slicer with the last
 * up-sampler or the first down-sampler).  Returns the number of launches, 0 if
 * some launch has no kernel (fast arithmetic then runs the cascade kernel). */
static int make_launches (DrbAmd_Batch_Config const * config, Drb_Plan const * plan, int const * first, int const * last, int count, Fast_Launch * launches);
static int partition_launches
    (
        DrbAmd_Batch_Config const * config, Drb_Plan const * plan, int sinc,
        int const * first, int const * last, int count, Fast_Launch * launches
    );

/* why the last build_launches on this thread found no matrix-core launches ("" where it found them) */
static _Thread_local char launches_why [96];

static int no_launches (char const * format, ...)
{
    va_list arguments;

    va_start(arguments, format);
    vsnprintf(launches_why, sizeof(launches_why), format, arguments);
    va_end(arguments);

    return 0;
}

static int build_launches (DrbAmd_Batch_Config const * config, Drb_Plan const * plan, Fast_Launch * launches)
{
    int sinc = -1, count = 0;

    /* One step of a 32-bit sample is 2^-31: finer than float32 resolves anywhere
     * above 2^-7 of full scale, so "within one step of upstream" means the same
     * float32 bit for bit.  Only the exact kernels deliver that: fast arithmetic
     * on int32 buffers runs them. */
    launches_why[0] = 0;

    if (config->sample_type == drb_amd_sample_i32) return no_launches("int32 samples: one step of theirs is a float32's last bit");
    if (config->arithmetic != drb_amd_arithmetic_fast) return no_launches("exact arithmetic asked for");

    int first [max_launches], last [max_launches];

    for (int k = 0; k < plan->stage_count; k++) if (plan->stages[k].kind == drb_stage_sinc) sinc = k;

    if (sinc >= 0)
    {
        int lo = sinc, hi = sinc;

        while (lo > 0 && plan->stages[lo - 1].kind == drb_stage_delay) lo--;
        if (lo > 0 && plan->stages[lo - 1].kind == drb_stage_up2x) lo--;
        while (hi + 1 < plan->stage_count && plan->stages[hi + 1].kind == drb_stage_delay) hi++;
        if (hi + 1 < plan->stage_count && plan->stages[hi + 1].kind == drb_stage_down2x) hi++;

        for (int k = 0; k < lo; k++) { first[count] = last[count] = k; count++; }
        first[count] = lo; last[count] = hi; count++;
        for (int k = hi + 1; k < plan->stage_count; k++) { first[count] = last[count] = k; count++; }
    }
    else
    {
        for (int k = 0; k < plan->stage_count; k++)
        {
            if (plan->stages[k].kind == drb_stage_delay)
            {
                /* with the launch in front of it, or, at the head of the chain, behind it */
                if (count > 0) last[count - 1] = k;
                else { first[count] = last[count] = k; count++; }
            }
            else if (count > 0 && plan->stages[last[count - 1]].kind == drb_stage_delay && first[count - 1] == last[count - 1])
            {
                last[count - 1] = k;
            }
            else
            {
                first[count] = last[count] = k; count++;
            }
        }
    }

    if (count < 1 || count > max_launches) return no_launches("%d launches", count);

    int const made = partition_launches(config, plan, sinc, first, last, count, launches);

    if (made > 0) return made;

    /* The caller's buffer format is what no kernel of the first or the last launch takes
     * (interleaved buffers of more than two channels, 16-bit samples where a launch
     * keeps planar float32 on its other side, ...): the launches that planar float32
     * gets, with one in front that turns the caller's source into planar float32
     * and one behind that turns the result into the caller's format.  Two more
     * passes over the samples, and still several times faster than the exact
     * kernels. */
    int const plain = config->sample_type == drb_amd_sample_f32 && (config->layout == drb_amd_layout_planar || config->channel_count == 1);

    if (plain || config->channel_count > 8) return 0;

    /* Three passes over the samples instead of one pay where the one is expensive.  Measured
     * on interleaved buffers of 3 to 8 channels, kernel time against the exact kernel's: chains
     * of 40, 52 and 82 flop per target sample (one or two interpolators, one decimator)
     * 41 %, 47 % and 5 % slower this way; of 214 and 262 flop (two decimators) 16 % and 24 %
     * faster; of 442 flop and more (a fractional stage among its 2x stages) three to four
     * times as fast. */
    double const flops = 2.0 * plan->macs_per_output_sample;

    if (flops < 150.0)
    {
        return no_launches("two more passes for the buffer format would not pay at %.0f flop per sample", flops);
    }

    char why [sizeof(launches_why)];

    memcpy(why, launches_why, sizeof(why));

    DrbAmd_Batch_Config between = *config;

    between.sample_type = drb_amd_sample_f32;
    between.layout = drb_amd_layout_planar;

    Fast_Launch * const inner = counted_calloc(max_launches, sizeof(Fast_Launch));

    if (!inner) return no_launches("out of host memory");

    int const inside = partition_launches(&between, plan, sinc, first, last, count, inner);

    if (inside <= 0 || inside + 2 > max_launches)
    {
        counted_free(inner);
        memcpy(launches_why, why, sizeof(why));   /* (what kept the caller's own format out says more) */

        return 0;
    }

    memset(&launches[0], 0, sizeof(launches[0]));
    memset(&launches[inside + 1], 0, sizeof(launches[0]));
    memcpy(&launches[1], inner, (size_t)inside * sizeof(Fast_Launch));
    counted_free(inner);

    Fast_Launch * const in = &launches[0], * const out = &launches[inside + 1];

    in->converts = out->converts = 1;
    in->rate_ratio = (double)plan->source_rate / (double)plan->target_rate;
    out->rate_ratio = 1.0;
    in->fast.usable = out->fast.usable = 1;
    in->fast.channels = out->fast.channels = config->channel_count;
    in->fast.sample_type = (int)config->sample_type;
    in->fast.layout = (int)config->layout;
    in->fast.target_sample_type = (int)drb_amd_sample_f32;
    in->fast.target_layout = (int)drb_amd_layout_planar;
    out->fast.sample_type = (int)drb_amd_sample_f32;
    out->fast.layout = (int)drb_amd_layout_planar;
    out->fast.target_sample_type = (int)config->sample_type;
    out->fast.target_layout = (int)config->layout;
    launches_why[0] = 0;

    return inside + 2;
}

/* The launches of the partition first[i] .. last[i] of the plan's stages, or, where
 * that serves, of the partition with the interpolators in front of the sinc stage's
 * launch folded into it.  0: some launch has no kernel (launches_why says which). */
static int partition_launches
    (
        DrbAmd_Batch_Config const * config, Drb_Plan const * plan, int sinc,
        int const * first, int const * last, int count, Fast_Launch * launches
    )
{
    /* Interpolators in front of the sinc stage's launch: folded into its filter where
     * the split kernel takes the longer chain (the rates in between then never
     * reach memory), else launches of their own. */
    if (sinc >= 0)
    {
        int core = -1, lone_ups = 1;

        for (int i = 0; i < count && core < 0; i++)
        {
            if (first[i] <= sinc && sinc <= last[i]) core = i;
            else if (first[i] != last[i] || plan->stages[first[i]].kind != drb_stage_up2x) lone_ups = 0;
        }

        char const * const wish = getenv("DRB_AMD_FAST_FOLD");   /* diagnostic: 0 keeps them apart */

        if (core > 0 && core <= 3 && lone_ups && !(wish && atoi(wish) == 0))
        {
            int folded_first [max_launches], folded_last [max_launches], folded = 0;

            folded_first[0] = 0; folded_last[0] = last[core]; folded = 1;

            for (int i = core + 1; i < count; i++) { folded_first[folded] = first[i]; folded_last[folded] = last[i]; folded++; }

            int const made = make_launches(config, plan, folded_first, folded_last, folded, launches);

            if (made > 0) return made;
        }
    }

    launches_why[0] = 0;

    return make_launches(config, plan, first, last, count, launches);
}

/* The launches of a partition of the plan's stages: stages first[i] .. last[i] each. */
static int make_launches (DrbAmd_Batch_Config const * config, Drb_Plan const * plan, int const * first, int const * last, int count, Fast_Launch * launches)
{
    for (int i = 0; i < count; i++)
    {
        Fast_Launch * const launch = &launches[i];
        Drb_Plan part = *plan;
        Drb_Geometry geometry;

        memset(launch, 0, sizeof(*launch));

        part.stage_count = last[i] - first[i] + 1;
        memmove(part.stages, plan->stages + first[i], sizeof(Drb_Stage) * (size_t)part.stage_count);

        launch->rate_ratio = (double)plan->stages[last[i]].out_rate / (double)plan->stages[plan->stage_count - 1].out_rate;

        for (int k = 0; k < first[i]; k++) if (plan->stages[k].kind == drb_stage_up2x) launch->ups_before++;
        for (int k = last[i] + 1; k < plan->stage_count; k++) if (plan->stages[k].kind == drb_stage_down2x) launch->downs_behind++;

        for (int k = 0; k < part.stage_count; k++)
        {
            if (part.stages[k].kind == drb_stage_sinc) launch->has_sinc = 1;
        }

        if (!launch->has_sinc)
        {
            for (int k = 0; k < part.stage_count; k++)
            {
                if (part.stages[k].kind == drb_stage_up2x) launch->doubles = 1;
                if (part.stages[k].kind == drb_stage_down2x) launch->halves = 1;
            }

            if (!launch->doubles && !launch->halves) return no_launches("launch %d of %d: a slicer alone", i + 1, count);   /* re-blocking, no filter */
        }

        drb_plan_geometry(&part, lds_budget, 0, &geometry);
        drb_plan_fast_geometry(&part, fast_lds_budget, &launch->fast);

        /* buffers between launches are planar float32 */
        DrbAmd_Batch_Config between = *config;

        between.sample_type = drb_amd_sample_f32;
        between.layout = drb_amd_layout_planar;

        build_chain(i == 0 ? config : &between, &part, &geometry, &launch->chain);

        launch->fast.sample_type = i == 0 ? (int)config->sample_type : (int)drb_amd_sample_f32;
        launch->fast.layout = i == 0 ? (int)config->layout : (int)drb_amd_layout_planar;
        launch->fast.target_sample_type = i == count - 1 ? (int)config->sample_type : (int)drb_amd_sample_f32;
        launch->fast.target_layout = i == count - 1 ? (int)config->layout : (int)drb_amd_layout_planar;
        launch->fast.history = geometry.history_frames;

        /* the staged kernels exist for some filter lengths only, and for no folded chain */
        int const folded = launch->fast.pre_ups > 0;
        int const staged_ok = !folded && drb_hip_fast_kernel_exists(&launch->fast);

        if (!launch->fast.usable)
        {
            return no_launches("launch %d of %d (%s): no tile geometry (segment %d -> %d frames, lookback %d)", i + 1, count,
                               launch->has_sinc ? "sinc" : launch->doubles ? "up2x" : "down2x",
                               launch->fast.seg_x, launch->fast.seg_y, launch->fast.look_x);
        }

        /* [up] sinc [down] as one filter on the half-precision matrix instruction
         * where that form applies and its kernel fits (DRB_AMD_FAST_SPLIT=0, a
         * diagnostic, keeps the three stages everywhere) */
        char const * const wish = getenv("DRB_AMD_FAST_SPLIT");

        /* (a sinc stage on its own has nothing to fold, but the split kernel is the faster one
         * for it as well: the headline at poor quality 1.56 against 1.93 ms per 10 s step;
         * DRB_AMD_FAST_LONE=0, a diagnostic, keeps its staged kernel where that one exists) */
        char const * const lone_wish = getenv("DRB_AMD_FAST_LONE");
        int const lone = !launch->fast.has_up && !launch->fast.has_down;
        int const skip_lone = lone && staged_ok && lone_wish && atoi(lone_wish) == 0;

        if (launch->has_sinc && !skip_lone && !(wish && atoi(wish) == 0))
        {
            Drb_Fast_Chain split = launch->fast;

            launch->staged = launch->fast;
            launch->has_staged = staged_ok;

            if (drb_plan_collapse_geometry(&split, split_lds_budget, 1)
                && drb_plan_split_geometry(&split, split_lds_budget) && drb_hip_fast_kernel_exists(&split))
            {
                launch->fast = split;
            }
        }

        if (launch->fast.kind != 4 && !staged_ok)
        {
            return no_launches("launch %d of %d (%s): no kernel image (up %d, sinc %d steps, down %d taps; segment %d -> %d)", i + 1, count,
                               launch->has_sinc ? "sinc" : launch->doubles ? "up2x" : "down2x",
                               launch->fast.has_up ? launch->fast.up_taps : 0, launch->fast.ksteps,
                               launch->fast.has_down ? launch->fast.down_taps : 0, launch->fast.seg_x, launch->fast.seg_y);
        }
    }

    return count;
}

/* what the launches hand to the matrix core per target sample of the whole chain;
 * *half_part: the share of it that runs on the half-precision instruction */
static double issued_flops (Fast_Launch const * launches, int count, double * half_part)
{
    double total = 0.0;

    *half_part = 0.0;

    for (int i = 0; i < count; i++)
    {
        Drb_Fast_Chain const * const fast = &launches[i].fast;

        if (launches[i].converts) continue;

        /* 16 x 16 x 4 products of 2048 flop each, per tile of one channel */
        long const up_blocks = fast->has_up ? (fast->seg_x + 15) / 16 : 0;
        long const down_blocks = fast->has_down ? (fast->seg_y + 15) / 16 : 0;
        if (fast->kind == 4)
        {
            /* four 16 x 16 x 32 products of 16384 flop per row block and step */
            double const flops = 4.0 * 16384.0 * (double)fast->blocks * fast->chunks / (double)fast->tile_frames * launches[i].rate_ratio;

            total += flops;
            *half_part += flops;
            continue;
        }

        long const products = up_blocks * ((fast->up_taps + 16) / 4)
                            + (long)fast->blocks * fast->periods * fast->ksteps * (fast->column_drift != 0.0f ? 2 : 1)
                            + down_blocks * ((fast->down_taps + 16) / 4);

        total += 2048.0 * (double)products / (double)fast->tile_frames * launches[i].rate_ratio;
    }

    return total;
}

DrbAmd_Status drb_amd_plan_info (DrbAmd_Batch_Config const * config, DrbAmd_Plan_Info * info)
{
    Drb_Plan plan;
    Drb_Geometry geometry;

    if (!config || !info) return fail(drb_amd_error_bad_argument, "null argument");

    DrbAmd_Status const status = resolve(config, &plan);

    if (status != drb_amd_ok) return status;

    drb_plan_geometry(&plan, lds_budget, channels_apart(config, &plan), &geometry);

    memset(info, 0, sizeof(*info));

    info->stage_count = plan.stage_count;

    for (int k = 0; k < plan.stage_count; k++)
    {
        info->stage_kind[k] = plan.stages[k].kind;
        info->stage_arg[k] = plan.stages[k].arg;
    }

    int const sample_bytes = config->sample_type == drb_amd_sample_i16 ? 2 : 4;

    info->throttling = plan.throttling;
    info->buffer_length = plan.buffer_length;
    info->converter_size = plan.converter_size;
    info->history_frames = geometry.history_frames;
    info->tile_frames = geometry.tile_frames;
    info->flops_per_output_sample = 2.0 * plan.macs_per_output_sample;
    info->bytes_per_output_sample =
        sample_bytes * ((double)plan.source_rate / plan.target_rate) + sample_bytes;

    Fast_Launch * const launches = counted_calloc(max_launches, sizeof(Fast_Launch));

    if (!launches) return fail(drb_amd_error_out_of_memory, "out of host memory");

    int const count = build_launches(config, &plan, launches);

    info->fast_path = count > 0;
    info->fast_launches = count;
    info->fast_tile_frames = 0;

    for (int i = 0; i < count; i++) if (!launches[i].converts) info->fast_tile_frames = launches[i].fast.tile_frames;   /* (of the last launch that filters) */

    info->fast_issued_flops_per_output_sample = issued_flops(launches, count, &info->fast_issued_half_flops_per_output_sample);

    for (int i = 0; i < count; i++) if (launches[i].fast.kind >= 3) info->fast_collapsed = launches[i].fast.kind - 2;

    /* the kernel that does most of the work */
    {
        char const * name = "drb_cascade_kernel";

        for (int i = 0, seen = 0; i < count; i++)
        {
            int const kind = launches[i].fast.kind;

            if (launches[i].converts) continue;

            if (!seen++ || launches[i].has_sinc)
            {
                name = kind == 4 ? "drb_split_kernel" : kind == 2 ? "drb_interpolate_kernel"
                     : kind == 1 ? "drb_decimate_kernel" : "drb_fast_kernel";
            }
        }

        snprintf(info->kernel_name, sizeof(info->kernel_name), "%s", name);
    }

    snprintf(info->fast_why_not, sizeof(info->fast_why_not), "%s", count > 0 ? "" : launches_why);

    counted_free(launches);

    return drb_amd_ok;
}

int drb_amd_plan_describe
    (
        DrbAmd_Batch_Config const * config,
        int passes [20], int edges [55], int * throttling, long * buffer_length
    )
{
    Drb_Plan plan;

    if (!config || drb_plan_resolve(config->source_sampling_rate, config->target_sampling_rate,
                                    config->channel_count, config->block_size, config->max_block_count,
                                    config->direction, config->quality, &plan) != 1)
    {
        return -1;
    }

    for (int i = 0; i < plan.pass_count; i++)
    {
        passes[2 * i] = plan.passes[i].kind;
        passes[2 * i + 1] = plan.passes[i].arg;
    }

    for (int i = 0; i <= plan.pass_count; i++)
    {
        edges[5 * i + 0] = plan.edges[i].rate;
        edges[5 * i + 1] = plan.edges[i].channels;
        edges[5 * i + 2] = plan.edges[i].layout;
        edges[5 * i + 3] = plan.edges[i].block;
        edges[5 * i + 4] = plan.edges[i].max_blocks;
    }

    *throttling = plan.throttling;
    *buffer_length = plan.buffer_length;

    return plan.pass_count;
}

/* ---- create / destroy ------------------------------------------------------- */

static void release (DrbAmd_Batch * batch)
{
    if (!batch) return;

    for (int i = 0; i < ring_slots; i++)
    {
        if (batch->slot_done[i])
        {
            if (batch->slot_pending[i]) drb_hip_event_synchronize(batch->slot_done[i]);
            drb_hip_event_destroy(batch->slot_done[i]);
        }
        batch->calls_host[i] = 0;   /* (in the batch's pinned block) */
    }

    for (int i = 0; i < host_slots; i++)
    {
        if (batch->feed_source[i]) drb_hip_free(batch->feed_source[i]);
        if (batch->feed_target[i]) drb_hip_free(batch->feed_target[i]);
        if (batch->feed_in[i]) drb_hip_event_destroy(batch->feed_in[i]);
        if (batch->feed_done[i]) drb_hip_event_destroy(batch->feed_done[i]);
        if (batch->feed_out[i]) drb_hip_event_destroy(batch->feed_out[i]);
    }

    for (int i = 0; i < 3; i++) if (batch->feed_streams[i]) drb_hip_stream_destroy(batch->feed_streams[i]);

    for (int i = 0; i < max_launches; i++)
    {
        if (batch->launches[i].history[0]) drb_hip_free(batch->launches[i].history[0]);
        if (batch->launches[i].history[1]) drb_hip_free(batch->launches[i].history[1]);
        if (batch->launches[i].output) drb_hip_free(batch->launches[i].output);
    }

    for (int i = 0; i < ring_slots; i++) if (batch->collapse_host[i] && batch->collapse_host_own[i]) drb_hip_host_free(batch->collapse_host[i]);

    if (batch->collapse_jobs) drb_hip_free(batch->collapse_jobs);
    if (batch->collapse_spots) drb_hip_free(batch->collapse_spots);
    if (batch->collapse_sets) drb_hip_free(batch->collapse_sets);
    counted_free(batch->collapse_hash);

    if (batch->kernel_start) drb_hip_event_destroy(batch->kernel_start);
    if (batch->kernel_stop) drb_hip_event_destroy(batch->kernel_stop);
    if (batch->step_start) drb_hip_event_destroy(batch->step_start);
    if (batch->step_stop) drb_hip_event_destroy(batch->step_stop);
    if (batch->history[0]) drb_hip_free(batch->history[0]);
    if (batch->history[1]) drb_hip_free(batch->history[1]);
    if (batch->calls_device) drb_hip_free(batch->calls_device);
    if (batch->fault_device) drb_hip_free(batch->fault_device);
    if (batch->pinned_block) drb_hip_host_free(batch->pinned_block);

    counted_free(batch->states);
    counted_free(batch->trial_states);
    counted_free(batch->steps);
    counted_free(batch->dense_source);
    counted_free(batch->dense_target);
    counted_free(batch->dense_frames);
    counted_free(batch->dense_capacity);
    counted_free(batch->dense_produced);
    counted_free(batch->feed_counts);
    counted_free(batch);
}

DrbAmd_Status drb_amd_batch_create (DrbAmd_Batch_Config const * config, DrbAmd_Batch ** out)
{
    if (!config || !out) return fail(drb_amd_error_bad_argument, "null argument");

    *out = 0;

    if (config->stream_count < 1) return fail(drb_amd_error_bad_argument, "stream_count must be positive");

    Drb_Plan plan;
    DrbAmd_Status status = resolve(config, &plan);

    if (status != drb_amd_ok) return status;

    /* the device: fail loudly, there is no host fallback */
    int device_count = 0, device = config->device, is_gfx950 = 0, lds_per_block = 0, code;

    if ((code = drb_hip_device_count(&device_count)) != 0 || device_count < 1)
    {
        return fail(drb_amd_error_no_device, "no HIP device visible (%s)", code ? drb_hip_error_text(code) : "count is 0");
    }

    if (device < 0 && (code = drb_hip_get_device(&device)) != 0) return fail_hip("hipGetDevice", code);
    if (device >= device_count) return fail(drb_amd_error_no_device, "device %d of %d does not exist", device, device_count);
    if ((code = drb_hip_device_is_gfx950(device, &is_gfx950, &lds_per_block)) != 0) return fail_hip("hipGetDeviceProperties", code);
    if (!is_gfx950) return fail(drb_amd_error_no_device, "device %d is not gfx950; this library carries gfx950 code only", device);
    if ((code = drb_hip_set_device(device)) != 0) return fail_hip("hipSetDevice", code);

    DrbAmd_Batch * const batch = counted_calloc(1, sizeof(*batch));

    if (!batch) return fail(drb_amd_error_out_of_memory, "out of host memory");

    batch->config = *config;
    batch->plan = plan;
    batch->device = device;
    batch->last_slot = -1;
    batch->sample_bytes = config->sample_type == drb_amd_sample_i16 ? 2 : 4;

    drb_plan_geometry(&plan, lds_budget, channels_apart(config, &plan), &batch->geometry);
    build_chain(config, &plan, &batch->geometry, &batch->chain);

    batch->launch_count = build_launches(config, &plan, batch->launches);

    /* the first launch reads the history the batch keeps for the whole chain */
    if (batch->launch_count > 0) batch->launches[0].fast.history = batch->launches[0].staged.history = batch->geometry.history_frames;

    /* timing diagnostics only: skip phases of the fast kernel (results are wrong) */
    {
        char const * const limit = getenv("DRB_AMD_SET_BYTES_LIMIT"), * const wish = getenv("DRB_AMD_FAST_APART");

        batch->set_bytes_limit = limit && atoll(limit) > 0 ? (size_t)atoll(limit) : collapse_set_bytes_limit;
        batch->wish_apart = wish ? atoi(wish) : -1;
        batch->group_count = 1;
    }

    char const * const ablate = getenv("DRB_AMD_FAST_ABLATE");

    char const * const stride = getenv("DRB_AMD_FAST_TILE_STRIDE");
    char const * const no_carry = getenv("DRB_AMD_FAST_NO_CARRY");
    char const * const unit_walk = getenv("DRB_AMD_UNIT_WALK");

    for (int i = 0; i < batch->launch_count; i++)
    {
        batch->launches[i].fast.ablate = ablate ? atoi(ablate) : 0;
        batch->launches[i].fast.tile_stride = stride ? atoi(stride) : 0;
        batch->launches[i].fast.no_carry = no_carry ? atoi(no_carry) : 0;
        batch->launches[i].fast.unit_walk = unit_walk ? atoi(unit_walk) : 0;
        batch->launches[i].fast.rebuild_tiles = getenv("DRB_AMD_FAST_REBUILD_TILES") ? atoi(getenv("DRB_AMD_FAST_REBUILD_TILES")) : 0;

        /* How many tiles one coefficient set of the split kernel serves follows from how
         * fast the fixed-point phase leaves the exact rate ratio, in source frames per
         * segment: 44.1 <-> 48 kHz (1.7e-8, 1.9e-8) and everything slower keeps its error
         * up to 192 tiles (tools/stretch_probe.py) and gets 128; 176.4 <-> 192 kHz
         * (4.5e-8, 4.8e-8) loses a third of its margin from 64 to 96 and gets 32. */
        if (batch->launches[i].fast.kind == 4 && batch->launches[i].fast.rebuild_tiles <= 0)
        {
            Drb_Fast_Chain const * const fast = &batch->launches[i].fast;
            double const drift = fabs((double)((long long)fast->seg_s * fast->target_delta - (long long)fast->seg_u * fast->source_delta) * fast->scale);

            /* (chains with interpolators folded in keep 64: their ragged benchmark batch, 16 -> 48 kHz
             * clips of 2 to 6 s, was measured 5 % slower with 128) */
            batch->launches[i].fast.rebuild_tiles = drift <= 2.0e-8 && fast->pre_ups == 0 ? 128 : drift >= 3.0e-8 ? 32 : 0;
        }
        batch->launches[i].staged.ablate = batch->launches[i].fast.ablate;
        batch->launches[i].staged.tile_stride = batch->launches[i].fast.tile_stride;
        batch->launches[i].staged.no_carry = batch->launches[i].fast.no_carry;
    }

    size_t const n = (size_t)config->stream_count;

    batch->states = counted_calloc(n, sizeof(Drb_Stream_State));
    batch->trial_states = counted_calloc(n, sizeof(Drb_Stream_State));
    batch->steps = counted_calloc(n, sizeof(Drb_Step));
    batch->dense_source = counted_calloc(n, sizeof(void *));
    batch->dense_target = counted_calloc(n, sizeof(void *));
    batch->dense_frames = counted_calloc(n, sizeof(int));
    batch->dense_capacity = counted_calloc(n, sizeof(int));
    batch->dense_produced = counted_calloc(n, sizeof(int));
    batch->feed_counts = counted_calloc(n, sizeof(int));

    if (!batch->states || !batch->trial_states || !batch->steps || !batch->dense_source
        || !batch->dense_target || !batch->dense_frames || !batch->dense_capacity || !batch->dense_produced
        || !batch->feed_counts)
    {
        release(batch);
        return fail(drb_amd_error_out_of_memory, "out of host memory");
    }

    size_t const history_bytes = n * plan.channels * batch->geometry.history_frames * sizeof(float);

    code = drb_hip_alloc((void **)&batch->history[0], history_bytes);
    if (!code) code = drb_hip_alloc((void **)&batch->history[1], history_bytes);
    if (!code) code = drb_hip_alloc((void **)&batch->calls_device, n * max_launches * sizeof(Drb_Device_Call));
    if (!code) code = drb_hip_alloc((void **)&batch->fault_device, fault_words * sizeof(long long));

    /* The buffers between a chain's launches, for steps of up to 16 384 frames on the
     * caller's side (or what fits half a gigabyte): they grow with the steps that come,
     * but a call that may not allocate finds these from the first step on. */
    if (!code && batch->launch_count > 1)
    {
        enum { modest_step = 16384 };
        size_t const budget = ((size_t)512 << 20) / (size_t)(batch->launch_count - 1);
        int const last_filter = batch->launch_count - 1 - (batch->launches[batch->launch_count - 1].converts ? 1 : 0);
        Drb_Stream_State trial;
        Drb_Step step;

        drb_stream_reset(&plan, &trial);

        if (drb_stream_step(&plan, &trial, modest_step, &step))
        {
            long long frames = step.source_frames;

            for (int l = 0; l + 1 < batch->launch_count && !code; l++)
            {
                Fast_Launch * const launch = &batch->launches[l];

                /* (the rule of drb_amd_batch_process for the frames between launches) */
                frames = l >= last_filter ? step.target_frames
                       : launch->converts ? frames
                       : launch->doubles ? 2 * frames
                       : (long long)step.target_frames << launch->downs_behind;

                long long room = (frames + frames / 4 + 255) / 256 * 256;
                size_t const each = n * plan.channels * sizeof(float);

                if ((size_t)room * each > budget) room = (long long)(budget / each) / 256 * 256;
                if (room < 1024) continue;   /* (very many streams: left to the first step) */

                code = drb_hip_alloc((void **)&launch->output, (size_t)room * each);
                if (!code) launch->output_frames = room;
            }
        }
    }

    /* pinned: the fault word, then per ring slot the step descriptors, then per ring slot
     * the tables every step of a modest size fits (a call that may not allocate
     * finds them there) */
    size_t const calls_bytes = (n * max_launches * sizeof(Drb_Device_Call) + 255) / 256 * 256;
    size_t const tables_bytes = static_host_bytes_for(n);

    if (!code) code = drb_hip_host_alloc(&batch->pinned_block, 256 + ring_slots * (calls_bytes + tables_bytes));

    if (!code)
    {
        char * const block = batch->pinned_block;

        batch->fault_host = (int *)block;

        for (int i = 0; i < ring_slots; i++)
        {
            batch->calls_host[i] = (Drb_Device_Call *)(block + 256 + (size_t)i * calls_bytes);
            batch->collapse_host[i] = block + 256 + ring_slots * calls_bytes + (size_t)i * tables_bytes;
            batch->collapse_host_bytes[i] = tables_bytes;
            batch->collapse_host_own[i] = 0;
        }
    }

    for (int i = 0; i < ring_slots && !code; i++) code = drb_hip_event_create(&batch->slot_done[i]);

    for (int i = 1; i < batch->launch_count && !code; i++)
    {
        if (batch->launches[i].converts) continue;   /* (frame for frame: nothing to remember) */

        size_t const bytes = n * plan.channels * (size_t)batch->launches[i].chain.history * sizeof(float);

        code = drb_hip_alloc((void **)&batch->launches[i].history[0], bytes);
        if (!code) code = drb_hip_alloc((void **)&batch->launches[i].history[1], bytes);
    }

    if (!code)
    {
        size_t hash_size = 64;

        while (hash_size < 16 * n + 2) hash_size *= 2;

        batch->collapse_hash = counted_malloc(hash_size * sizeof(int));
        batch->collapse_hash_room = batch->collapse_hash ? hash_size : 0;
    }

    /* the static workspace's address on this GPU (looked up once per device); a
     * failure here only concerns drb_amd_batch_process_dense_static, which reports it */
    if (!code && drb_hip_static_workspace(device, &batch->static_base, &batch->static_bytes) != 0)
    {
        batch->static_base = 0;
        batch->static_bytes = 0;
    }

    if (!code) code = drb_hip_memset_async(batch->fault_device, 0, fault_words * sizeof(long long), 0);

    if (code)
    {
        release(batch);
        return fail(drb_amd_error_out_of_memory, "device allocation failed: %s", drb_hip_error_text(code));
    }

    *batch->fault_host = 0;

    status = drb_amd_batch_reset(batch, 0, config->stream_count);

    if (status != drb_amd_ok)
    {
        release(batch);
        return status;
    }

    *out = batch;

    return drb_amd_ok;
}

void drb_amd_batch_destroy (DrbAmd_Batch * batch)
{
    if (!batch) return;

    if (getenv("DRB_AMD_HOST_PROFILE") && host_steps > 0)
    {
        fprintf(stderr, "host us per step: frame arithmetic %.1f, descriptors %.1f, coefficient jobs + uploads %.1f, launches %.1f, waiting for a step slot %.1f (%ld steps)\n",
                host_spent[0] / host_steps, host_spent[1] / host_steps, host_spent[2] / host_steps, host_spent[3] / host_steps, host_spent[4] / host_steps, host_steps);
    }

    drb_hip_set_device(batch->device);
    release(batch);
}

DrbAmd_Status drb_amd_batch_reset (DrbAmd_Batch * batch, int first, int count)
{
    if (!batch || first < 0 || count < 0 || first + count > batch->config.stream_count)
    {
        return fail(drb_amd_error_bad_argument, "bad stream range");
    }

    int code = drb_hip_set_device(batch->device);

    /* Earlier steps may still be reading the histories, and later ones must find them
     * cleared: the clearing goes on the stream of the batch's last step, behind it.  A
     * step on another stream waits for that step's event, which is recorded again behind
     * the clearing.  Nothing waits on the host (a pull converter that starts over before
     * every step would otherwise never have two steps in flight).  A batch that has
     * nothing in flight clears on the null stream and waits for that. */
    int pending = 0;

    for (int i = 0; i < ring_slots; i++) pending |= batch->slot_pending[i];

    int ordered = pending && batch->last_slot >= 0 && batch->slot_pending[batch->last_slot];

    /* (twice at most: a caller may have waited for its stream itself and destroyed it
     * since; the clearing then fails there and is done the other way) */
    for (int attempt = 0; attempt < 2; attempt++)
    {
        void * const hip_stream = ordered ? batch->last_stream : 0;

        if (!ordered)
        {
            for (int i = 0; i < ring_slots && !code; i++)
            {
                if (batch->slot_pending[i])
                {
                    code = drb_hip_event_synchronize(batch->slot_done[i]);
                    batch->slot_pending[i] = 0;
                }
            }
        }

        size_t const per_stream = (size_t)batch->plan.channels * batch->geometry.history_frames * sizeof(float);

        for (int i = 0; i < 2 && !code; i++)
        {
            code = drb_hip_memset_async((char *)batch->history[i] + first * per_stream, 0, count * per_stream, hip_stream);
        }

        for (int l = 1; l < batch->launch_count && !code; l++)
        {
            if (batch->launches[l].converts) continue;

            size_t const each = (size_t)batch->plan.channels * batch->launches[l].chain.history * sizeof(float);

            for (int i = 0; i < 2 && !code; i++)
            {
                code = drb_hip_memset_async((char *)batch->launches[l].history[i] + first * each, 0, count * each, hip_stream);
            }
        }

        if (!code && ordered) code = drb_hip_event_record(batch->slot_done[batch->last_slot], hip_stream);
        if (!code && !ordered) code = drb_hip_stream_synchronize(0);

        if (!code || !ordered) break;

        drb_hip_clear_error();
        ordered = 0;
        code = 0;
    }

    if (code) return fail_hip("history reset", code);

    for (int s = first; s < first + count; s++) drb_stream_reset(&batch->plan, &batch->states[s]);

    return drb_amd_ok;
}

/* ---- stepping ---------------------------------------------------------------- */

/* for the drop-in wrapper (not in the public header): the timestamp offset
 * upstream would hand to the callback for a sub-block starting now */
double drb_batch_callback_latency (DrbAmd_Batch const * batch, int stream)
{
    return drb_plan_callback_latency(&batch->plan, &batch->states[stream]);
}

DrbAmd_Status drb_amd_batch_query (DrbAmd_Batch const * batch, int const * frames, int * counts)
{
    if (!batch || !frames || !counts) return fail(drb_amd_error_bad_argument, "null argument");

    int const pull = batch->plan.direction == 1;

    for (int s = 0; s < batch->config.stream_count; s++)
    {
        Drb_Stream_State state = batch->states[s];
        Drb_Step step;

        if (frames[s] < 0 || frames[s] > max_step_frames) return fail(drb_amd_error_bad_argument, "stream %d: frame count %d out of range", s, frames[s]);
        if (!drb_stream_step(&batch->plan, &state, frames[s], &step)) return fail(drb_amd_error_bad_argument, "stream %d: odd frame count reaches a 2x stage", s);

        counts[s] = pull ? step.source_frames : step.target_frames;
    }

    return drb_amd_ok;
}

/* ---- collapsed launch: which coefficient sets a step needs ------------------- */

enum { code_no_room = -77 };   /* not a HIP code: a buffer would have to grow in a call that may not allocate */

static int collapse_grow_device (void ** pointer, size_t * room, size_t wanted, size_t element, void * hip_stream, int may_grow)
{
    if (wanted <= *room) return 0;
    if (!may_grow) return code_no_room;

    int code = drb_hip_stream_synchronize(hip_stream);   /* earlier steps may still read the old one */

    if (code) return code;
    if (*pointer) drb_hip_free(*pointer);

    *pointer = 0;
    *room = 0;
    wanted += wanted / 2 + 16;

    if ((code = drb_hip_alloc(pointer, wanted * element)) != 0) return code;

    *room = wanted;

    return 0;
}

/* Lays the step of every stream out in stretches of tiles, finds the distinct
 * phases among their anchors (streams that run in step share them), and sends
 * the jobs and the per-stretch table to the device.  The kernel lays the
 * stretches out by the same rule (drb_collapsed.inc). */
static DrbAmd_Status collapse_prepare
    (
        DrbAmd_Batch * batch, Fast_Launch const * launch, Drb_Device_Call const * calls, int slot,
        void * hip_stream, int * max_stretches_out, int * job_count_out, int * too_many_out
    )
{
    *too_many_out = 0;

    Drb_Fast_Chain const * const fast = &launch->fast;
    int const n = batch->config.stream_count;
    int const wanted = fast->rebuild_tiles > 0 ? fast->rebuild_tiles : drb_split_stretch_tiles;
    int max_stretches = 1, code;
    size_t total = 0;

    for (int s = 0; s < n; s++)
    {
        int const tiles = (calls[s].target_frames + fast->tile_frames - 1) / fast->tile_frames;
        int const stretches = (tiles + wanted - 1) / wanted;

        if (stretches > max_stretches) max_stretches = stretches;
        total += (size_t)stretches;
    }

    size_t const spots_bytes = (size_t)n * max_stretches * sizeof(Drb_Collapse_Spot);
    size_t const host_bytes = spots_bytes + (total + 1) * sizeof(Drb_Collapse_Job);

    if (host_bytes > batch->collapse_host_bytes[slot])
    {
        if (batch->no_alloc) return fail(drb_amd_error_capacity, "the step's coefficient tables do not fit the pinned memory the batch was created with");

        /* (the slot's share of the batch's pinned block is left where it is) */
        if (batch->collapse_host[slot] && batch->collapse_host_own[slot]) drb_hip_host_free(batch->collapse_host[slot]);

        batch->collapse_host[slot] = 0;
        batch->collapse_host_bytes[slot] = 0;
        batch->collapse_host_own[slot] = 0;

        if ((code = drb_hip_host_alloc(&batch->collapse_host[slot], host_bytes + host_bytes / 2)) != 0)
        {
            return fail(drb_amd_error_out_of_memory, "pinned memory for the coefficient jobs: %s", drb_hip_error_text(code));
        }

        batch->collapse_host_bytes[slot] = host_bytes + host_bytes / 2;
        batch->collapse_host_own[slot] = 1;
    }

    size_t hash_size = 64;

    while (hash_size < 2 * total + 2) hash_size *= 2;

    if (hash_size > batch->collapse_hash_room)
    {
        if (batch->no_alloc) return fail(drb_amd_error_capacity, "too many stretches for the tables the batch was created with");

        counted_free(batch->collapse_hash);
        batch->collapse_hash = counted_malloc(hash_size * sizeof(int));
        batch->collapse_hash_room = batch->collapse_hash ? hash_size : 0;

        if (!batch->collapse_hash) return fail(drb_amd_error_out_of_memory, "out of host memory");
    }

    memset(batch->collapse_hash, 0, hash_size * sizeof(int));

    Drb_Collapse_Spot * const spots = batch->collapse_host[slot];
    Drb_Collapse_Job * const jobs = (Drb_Collapse_Job *)((char *)batch->collapse_host[slot] + spots_bytes);
    int job_count = 0, most_jobs = 0;
    size_t const set_bytes = (size_t)drb_hip_collapse_set_words(fast) * sizeof(unsigned int);
    size_t const limit = batch->no_alloc ? batch->collapse_sets_room * sizeof(unsigned int) : batch->set_bytes_limit;
    Collapse_Group * group = &batch->groups[0];

    batch->group_count = 1;
    group->first_stream = 0;
    group->first_job = 0;

    for (int s = 0; s < n; s++)
    {
        Drb_Device_Call const * const call = &calls[s];
        Drb_Collapse_Spot * const row = spots + (size_t)s * max_stretches;
        int const tiles = (call->target_frames + fast->tile_frames - 1) / fast->tile_frames;
        int const stretches = (tiles + wanted - 1) / wanted;
        int const every = wanted;   /* a fixed length: streams of one phase share sets whatever their lengths */
        int const jobs_before = job_count;

        /* streams that run in step: the same stretches */
        if (s > group->first_stream && call->phase == call[-1].phase && call->delay == call[-1].delay
            && call->target_frames == call[-1].target_frames)
        {
            memcpy(row, row - max_stretches, (size_t)max_stretches * sizeof(Drb_Collapse_Spot));
            continue;
        }

        for (int j = 0; j < max_stretches; j++)
        {
            if (j >= stretches) { row[j].job = 0; row[j].base_x = 0; continue; }

            long long rest, frame;
            int parity;

            drb_plan_collapse_key(fast, call->phase, call->delay, 16LL * j * every + 8LL * every, &rest, &parity, &frame);

            unsigned long long const mixed = ((unsigned long long)rest * 2 + (unsigned)parity) * 0x9E3779B97F4A7C15ULL;
            size_t at = (size_t)(mixed >> 20) & (hash_size - 1);
            int found = -1;

            while (batch->collapse_hash[at])
            {
                Drb_Collapse_Job const * const other = &jobs[batch->collapse_hash[at] - 1];

                if (other->rest == rest && other->parity == parity) { found = batch->collapse_hash[at] - 1; break; }

                at = (at + 1) & (hash_size - 1);
            }

            if (found < 0)
            {
                /* A set per phase and stretch: with very many phases apart (thousands of
                 * streams, each cut differently) the sets would outweigh the samples.
                 * Such a step runs the three stages apart, which need no sets; a chain
                 * that has no such kernels goes on with this stream in a group of its own. */
                if (((size_t)(job_count - group->first_job) + 2) * set_bytes > limit)
                {
                    if (launch->has_staged)
                    {
                        *too_many_out = 1;
                        return drb_amd_ok;
                    }

                    if (s == group->first_stream || batch->group_count == max_groups)
                    {
                        return fail(drb_amd_error_capacity, "stream %d: the coefficient sets of this step do not fit %s", s,
                                    batch->no_alloc ? "the static workspace" : "their memory");
                    }

                    /* the group ends in front of this stream, which starts the next one over */
                    job_count = jobs_before;
                    group->stream_count = s - group->first_stream;
                    group->job_count = job_count - group->first_job;
                    if (group->job_count > most_jobs) most_jobs = group->job_count;

                    group = &batch->groups[batch->group_count++];
                    group->first_stream = s;
                    group->first_job = job_count;
                    memset(batch->collapse_hash, 0, hash_size * sizeof(int));
                    s -= 1;
                    goto next_stream;
                }

                if (!drb_plan_collapse_job(fast, rest, parity, &jobs[job_count]))
                {
                    return fail(drb_amd_error_device, "stream %d: filter windows do not fit the kernel's geometry", s);
                }

                found = job_count++;
                batch->collapse_hash[at] = found + 1;
            }

            long long const base = frame + jobs[found].origin;

            if (base < -0x40000000LL || base > 0x40000000LL) return fail(drb_amd_error_bad_argument, "stream %d: step too long", s);

            row[j].job = found - group->first_job;
            row[j].base_x = (int)base;
        }

      next_stream: ;
    }

    group->stream_count = n - group->first_stream;
    group->job_count = job_count - group->first_job;
    if (group->job_count > most_jobs) most_jobs = group->job_count;

    code = collapse_grow_device((void **)&batch->collapse_spots, &batch->collapse_spots_room, (size_t)n * max_stretches, sizeof(Drb_Collapse_Spot), hip_stream, !batch->no_alloc);
    if (!code) code = collapse_grow_device((void **)&batch->collapse_jobs, &batch->collapse_jobs_room, (size_t)job_count + 1, sizeof(Drb_Collapse_Job), hip_stream, !batch->no_alloc);
    if (!code) code = collapse_grow_device((void **)&batch->collapse_sets, &batch->collapse_sets_room,
                                           ((size_t)most_jobs + 1) * (size_t)drb_hip_collapse_set_words(fast), sizeof(unsigned int), hip_stream, !batch->no_alloc);

    if (code == code_no_room) return fail(drb_amd_error_capacity, "the step's coefficient tables do not fit the static workspace");
    if (code) return fail(drb_amd_error_out_of_memory, "coefficient sets: %s", drb_hip_error_text(code));

    /* (they travel with the step's descriptors, in one launch: drb_amd_batch_process) */
    batch->pending_device[0] = batch->collapse_spots; batch->pending_host[0] = spots; batch->pending_bytes[0] = spots_bytes;
    batch->pending_device[1] = batch->collapse_jobs; batch->pending_host[1] = jobs; batch->pending_bytes[1] = (size_t)job_count * sizeof(Drb_Collapse_Job);

    *max_stretches_out = max_stretches;
    *job_count_out = job_count;

    return drb_amd_ok;
}

DrbAmd_Status drb_amd_batch_process
    (
        DrbAmd_Batch * batch,
        void const * const * source, long long source_channel_stride,
        int const * source_frames,
        void * const * target, long long target_channel_stride,
        int const * target_frames,
        int * produced,
        void * hip_stream
    )
{
    if (!batch || !source || !source_frames || !target || !target_frames)
    {
        return fail(drb_amd_error_bad_argument, "null argument");
    }

    int const n = batch->config.stream_count;
    int const pull = batch->plan.direction == 1;
    int most_target = 0, code;
    double host_mark = host_now();

    /* phase 1: frame arithmetic on a copy of the state, so that a rejected
     * call leaves the batch untouched */
    for (int s = 0; s < n; s++)
    {
        int const frames = pull ? target_frames[s] : source_frames[s];
        Drb_Step * const step = &batch->steps[s];

        batch->trial_states[s] = batch->states[s];

        if (frames < 0 || frames > max_step_frames) return fail(drb_amd_error_bad_argument, "stream %d: frame count %d out of range", s, frames);
        if (!drb_stream_step(&batch->plan, &batch->trial_states[s], frames, step)) return fail(drb_amd_error_bad_argument, "stream %d: odd frame count reaches a 2x stage", s);

        if (pull && step->source_frames > source_frames[s]) return fail(drb_amd_error_capacity, "stream %d: %d source frames needed, %d supplied", s, step->source_frames, source_frames[s]);
        if (!pull && step->target_frames > target_frames[s]) return fail(drb_amd_error_capacity, "stream %d: %d target frames produced, room for %d", s, step->target_frames, target_frames[s]);
        if ((step->source_frames > 0 && !source[s]) || (step->target_frames > 0 && !target[s])) return fail(drb_amd_error_bad_argument, "stream %d: null buffer", s);

        if (step->target_frames > most_target) most_target = step->target_frames;
    }

    HOST_MARK(0);

    if ((code = drb_hip_set_device(batch->device)) != 0) return fail_hip("hipSetDevice", code);

    /* phase 2: descriptors -> device, launch */
    int const slot = batch->slot;

    HOST_MARK(1);

    if (batch->slot_pending[slot])
    {
        if ((code = drb_hip_event_synchronize(batch->slot_done[slot])) != 0) return fail_hip("hipEventSynchronize", code);
        batch->slot_pending[slot] = 0;
    }

    HOST_MARK(4);   /* waiting for one of the ring's step slots: the device's time, not the host's */

    Drb_Device_Call * const calls = batch->calls_host[slot];
    int const fast = batch->config.arithmetic == drb_amd_arithmetic_fast && batch->launch_count > 0;
    int const launches = fast ? batch->launch_count : 1;
    long long most [max_launches];   /* most frames any stream gets out of each launch */

    /* frames between the launches: an interpolator on its own doubles what it
     * gets; everything from the sinc stage on is counted back from the target */
    long long least [max_launches];  /* and fewest */
    /* the last launch that filters makes the step's target frames (a slicer behind it may hold
     * frames back or hand out held ones: not "twice what came in"); a launch behind it only
     * changes their format */
    int const last_filter = launches - 1 - (fast && launches > 1 && batch->launches[launches - 1].converts ? 1 : 0);

    for (int l = 0; l < launches; l++) { most[l] = 0; least[l] = 0x7fffffffffffffffLL; }

    for (int s = 0; s < n; s++)
    {
        long long frames = batch->steps[s].source_frames;

        for (int l = 0; l < launches; l++)
        {
            Fast_Launch const * const launch = &batch->launches[l];

            frames = l >= last_filter ? batch->steps[s].target_frames
                   : launch->converts ? frames
                   : launch->doubles ? 2 * frames
                   : (long long)batch->steps[s].target_frames << launch->downs_behind;

            if (frames > most[l]) most[l] = frames;
            if (frames < least[l]) least[l] = frames;
        }
    }

    /* the batch's buffers between launches grow when a step needs more */
    for (int l = 0; l + 1 < launches; l++)
    {
        Fast_Launch * const launch = &batch->launches[l];

        if (most[l] > launch->output_frames)
        {
            if (batch->no_alloc) return fail(drb_amd_error_capacity, "a buffer between this chain's launches would have to grow");

            long long const frames = (most[l] + most[l] / 4 + 255) / 256 * 256;

            if ((code = drb_hip_stream_synchronize(hip_stream)) != 0) return fail_hip("hipStreamSynchronize", code);
            if (launch->output) drb_hip_free(launch->output);

            launch->output = 0;
            launch->output_frames = 0;

            if ((code = drb_hip_alloc((void **)&launch->output, (size_t)n * batch->plan.channels * (size_t)frames * sizeof(float))) != 0)
            {
                return fail(drb_amd_error_out_of_memory, "buffer between launches: %s", drb_hip_error_text(code));
            }

            launch->output_frames = frames;
        }
    }

    for (int s = 0; s < n; s++)
    {
        Drb_Step const * const step = &batch->steps[s];
        long long frames = step->source_frames;

        for (int l = 0; l < launches; l++)
        {
            Fast_Launch const * const launch = &batch->launches[l];
            Fast_Launch const * const before = l > 0 ? &batch->launches[l - 1] : 0;
            Drb_Device_Call * const call = &calls[(size_t)l * n + s];
            long long const out = l >= last_filter ? step->target_frames
                                : launch->converts ? frames
                                : launch->doubles ? 2 * frames
                                : (long long)step->target_frames << launch->downs_behind;

            call->source = before ? (unsigned long long)(size_t)(before->output + (size_t)s * batch->plan.channels * before->output_frames)
                                  : (unsigned long long)(size_t)source[s];
            call->target = l == launches - 1 ? (unsigned long long)(size_t)target[s]
                         : (unsigned long long)(size_t)(launch->output + (size_t)s * batch->plan.channels * launch->output_frames);
            call->phase = step->phase;
            call->source_frames = (int)frames;
            call->target_frames = (int)out;
            call->delay = step->delay;
            call->reserved = 0;

            frames = out;
        }
    }

    int const current = batch->history_current;
    float * const history_in = batch->history[current];
    float * const history_out = batch->history[current ^ 1];
    int const tile = batch->geometry.tile_frames;

    int collapse_stretches = 1, collapse_job_count = 0;

    batch->pending_bytes[0] = batch->pending_bytes[1] = 0;   /* (set by collapse_prepare where a step needs them) */
    batch->step_walkers = 0;
    int apart [max_launches];   /* 1: this step runs the launch's three stages apart */

    HOST_MARK(1);

    for (int l = 0; l < launches; l++)
    {
        apart[l] = 0;

        if (fast && batch->launches[l].fast.kind >= 3)
        {
            if (batch->wish_apart == 1 && batch->launches[l].has_staged) { apart[l] = 1; continue; }

            /* A step of a tile or two per stream (calls of a few thousand frames) is all
             * set-up: the staged kernels need no coefficient sets built and uploaded
             * first, and their small workgroups overlap on a CU where the split
             * kernel's one workgroup per CU starts and drains alone.  Measured on
             * 4096-frame calls of 1024 streams: 0.092 against 0.141 ms of device time
             * per call.  Not so where the streams run in step, every one gets as many
             * tiles and there are more of them than the chip holds workgroups: the
             * split kernel's workgroups then go from stream to stream, a few sets
             * serve them all, and a call takes 0.048 ms. */
            if (batch->launches[l].has_staged && batch->wish_apart != 0)
            {
                Drb_Device_Call const * const row = calls + (size_t)l * n;
                int const each = batch->launches[l].fast.tile_frames;
                long long most_frames = 0, least_frames = 0x7fffffff;
                int phases = 1;

                for (int s = 0; s < n; s++)
                {
                    if (row[s].target_frames > most_frames) most_frames = row[s].target_frames;
                    if (row[s].target_frames < least_frames) least_frames = row[s].target_frames;
                    if (s > 0 && (row[s].phase != row[s - 1].phase || row[s].delay != row[s - 1].delay)) phases += 1;
                }

                if (most_frames <= 2LL * each)
                {
                    int const in_step = phases <= 4 || phases <= n / 64;

                    if (!in_step || !drb_hip_collapsed_walkers(&batch->launches[l].fast, n, (int)((most_frames + each - 1) / each), (int)((least_frames + each - 1) / each)))
                    {
                        apart[l] = 1;
                        continue;
                    }
                }
            }

            DrbAmd_Status const status = collapse_prepare(batch, &batch->launches[l], calls + (size_t)l * n, slot, hip_stream,
                                                          &collapse_stretches, &collapse_job_count, &apart[l]);

            if (status != drb_amd_ok) return status;
        }
    }

    HOST_MARK(2);

    /* A batch has one set of descriptors, histories and coefficient sets: a step on
     * another stream than the last one waits for that one first. */
    code = 0;

    if (batch->last_slot >= 0 && batch->last_stream != hip_stream && batch->slot_pending[batch->last_slot])
    {
        code = drb_hip_stream_wait_event(hip_stream, batch->slot_done[batch->last_slot]);
    }

    if (!code && batch->timing) code = drb_hip_event_record(batch->step_start, hip_stream);
    if (!code)
    {
        batch->pending_device[2] = batch->calls_device; batch->pending_host[2] = calls;
        batch->pending_bytes[2] = (size_t)n * launches * sizeof(Drb_Device_Call);

        code = drb_hip_upload_tables(batch->pending_device, batch->pending_host, batch->pending_bytes, 3, hip_stream);
    }

    /* coefficient sets of the step: they depend on the streams' phases only, not
     * on their samples (built in front of the timed region, which is the filter
     * kernels'; on a second stream, beside the last step's filter kernel, this
     * was measured far slower: every cross-stream wait cost most of a millisecond) */
    for (int l = 0; l < launches && !code; l++)
    {
        if (fast && batch->launches[l].fast.kind >= 3 && !apart[l] && batch->group_count == 1)
        {
            code = drb_hip_launch_collapse_build(&batch->launches[l].fast, batch->collapse_jobs, collapse_job_count,
                                                 batch->collapse_sets, batch->fault_device, hip_stream);
        }
    }

    if (!code && batch->timing) code = drb_hip_event_record(batch->kernel_start, hip_stream);

    if (fast)
    {
        for (int l = 0; l < launches && !code; l++)
        {
            Fast_Launch const * const launch = &batch->launches[l];
            Drb_Fast_Chain const * const shape = apart[l] ? &launch->staged : &launch->fast;
            Drb_Device_Call const * const launch_calls = batch->calls_device + (size_t)l * n;
            long long const source_stride = l == 0 ? source_channel_stride : batch->launches[l - 1].output_frames;
            long long const target_stride = l == launches - 1 ? target_channel_stride : launch->output_frames;
            int const fast_tile = shape->tile_frames;

            if (launch->converts)
            {
                code = drb_hip_launch_convert
                (
                    launch_calls, n, most[l], batch->plan.channels,
                    shape->sample_type, shape->layout, source_stride,
                    shape->target_sample_type, shape->target_layout, target_stride, hip_stream
                );
            }
            else if (shape->kind >= 3)
            {
                /* (several groups: a group's sets are built where the last group's were, behind its filter kernel) */
                for (int g = 0; g < batch->group_count && !code; g++)
                {
                    Collapse_Group const * const group = &batch->groups[g];
                    float const * const past = (l == 0 ? history_in : launch->history[current])
                                             + (size_t)group->first_stream * shape->channels * shape->history;

                    if (batch->group_count > 1)
                    {
                        code = drb_hip_launch_collapse_build(shape, batch->collapse_jobs + group->first_job, group->job_count,
                                                             batch->collapse_sets, batch->fault_device, hip_stream);
                    }

                    if (!code) code = drb_hip_launch_collapsed
                    (
                        shape, launch_calls + group->first_stream, group->stream_count,
                        (int)((most[l] + fast_tile - 1) / fast_tile), (int)((least[l] + fast_tile - 1) / fast_tile),
                        past, source_stride, target_stride,
                        batch->collapse_jobs + group->first_job,
                        batch->collapse_spots + (size_t)group->first_stream * collapse_stretches, collapse_stretches,
                        batch->collapse_sets, batch->fault_device, hip_stream, &batch->step_walkers
                    );
                }
            }
            else code = drb_hip_launch_fast
            (
                shape, launch_calls, n, (int)((most[l] + fast_tile - 1) / fast_tile),
                l == 0 ? history_in : launch->history[current], source_stride, target_stride,
                batch->fault_device, hip_stream
            );

            /* what this launch read becomes the history of its next step */
            if (!code && l > 0 && !launch->converts)
            {
                code = drb_hip_launch_history
                (
                    &launch->chain, launch_calls, n, launch->history[current], launch->history[current ^ 1],
                    source_stride, hip_stream
                );
            }
        }
    }
    else if (!code)
    {
        code = drb_hip_launch_cascade
        (
            &batch->chain, 1,
            batch->calls_device, n, (most_target + tile - 1) / tile, history_in,
            source_channel_stride, target_channel_stride,
            batch->geometry.lds_bytes, batch->fault_device, hip_stream
        );
    }

    if (!code && batch->timing)
    {
        code = drb_hip_event_record(batch->kernel_stop, hip_stream);
        batch->kernel_timed = 1;
    }

    if (!code)
    {
        code = drb_hip_launch_history
        (
            &batch->chain, batch->calls_device, n, history_in, history_out,
            source_channel_stride, hip_stream
        );
    }

    if (!code && batch->timing) code = drb_hip_event_record(batch->step_stop, hip_stream);
    if (!code) code = drb_hip_event_record(batch->slot_done[slot], hip_stream);

    if (code) return fail_hip("launch", code);

    HOST_MARK(3);
    host_steps++;

    batch->slot_pending[slot] = 1;
    batch->last_slot = slot;
    batch->last_stream = hip_stream;
    batch->slot = (slot + 1) % ring_slots;
    batch->history_current ^= 1;

    for (int s = 0; s < n; s++)
    {
        batch->states[s] = batch->trial_states[s];
        if (produced) produced[s] = pull ? batch->steps[s].source_frames : batch->steps[s].target_frames;
    }

    return drb_amd_ok;
}

DrbAmd_Status drb_amd_batch_process_dense
    (
        DrbAmd_Batch * batch,
        void const * source_base, long long source_stream_stride, long long source_channel_stride,
        void * target_base, long long target_stream_stride, long long target_channel_stride,
        int frames, int capacity,
        int * produced_each,
        void * hip_stream
    )
{
    if (!batch) return fail(drb_amd_error_bad_argument, "null argument");

    int const n = batch->config.stream_count;
    int const pull = batch->plan.direction == 1;

    for (int s = 0; s < n; s++)
    {
        batch->dense_source[s] = (char const *)source_base + (long long)s * source_stream_stride * batch->sample_bytes;
        batch->dense_target[s] = (char *)target_base + (long long)s * target_stream_stride * batch->sample_bytes;
        batch->dense_frames[s] = frames;
        batch->dense_capacity[s] = capacity;
    }

    return drb_amd_batch_process
    (
        batch,
        batch->dense_source, source_channel_stride, pull ? batch->dense_capacity : batch->dense_frames,
        batch->dense_target, target_channel_stride, pull ? batch->dense_frames : batch->dense_capacity,
        produced_each ? produced_each : batch->dense_produced, hip_stream
    );
}

/* ---- the same without allocating --------------------------------------------- */

/* How the static workspace is cut up: step descriptors, then the coefficient
 * tables of a collapsed launch (stretch table, jobs, sets). */
enum { static_calls_bytes = 1 << 20, static_spots_bytes = 1 << 20, static_jobs_bytes = 256 << 10 };

DrbAmd_Status drb_amd_batch_process_dense_static
    (
        DrbAmd_Batch * batch,
        void const * source_base, long long source_stream_stride, long long source_channel_stride,
        void * target_base, long long target_stream_stride, long long target_channel_stride,
        int frames, int capacity,
        int * produced_each,
        void * hip_stream
    )
{
    if (!batch) return fail(drb_amd_error_bad_argument, "null argument");

    size_t const fixed = static_calls_bytes + static_spots_bytes + static_jobs_bytes;

    if (!batch->static_base || batch->static_bytes <= fixed) return fail(drb_amd_error_device, "the static workspace is not available");

    size_t const launches = batch->config.arithmetic == drb_amd_arithmetic_fast && batch->launch_count > 0 ? (size_t)batch->launch_count : 1;

    if ((size_t)batch->config.stream_count * launches * sizeof(Drb_Device_Call) > static_calls_bytes)
    {
        return fail(drb_amd_error_capacity, "%d streams x %d launches: more step descriptors than the static workspace holds (%d)",
                    batch->config.stream_count, (int)launches, (int)(static_calls_bytes / sizeof(Drb_Device_Call)));
    }

    /* this call's tables live in the workspace instead of the batch's own buffers */
    Drb_Device_Call * const own_calls = batch->calls_device;
    Drb_Collapse_Spot * const own_spots = batch->collapse_spots;
    Drb_Collapse_Job * const own_jobs = batch->collapse_jobs;
    unsigned int * const own_sets = batch->collapse_sets;
    size_t const own_spots_room = batch->collapse_spots_room, own_jobs_room = batch->collapse_jobs_room, own_sets_room = batch->collapse_sets_room;
    char * const base = batch->static_base;

    batch->calls_device = (Drb_Device_Call *)base;
    batch->collapse_spots = (Drb_Collapse_Spot *)(base + static_calls_bytes);
    batch->collapse_spots_room = static_spots_bytes / sizeof(Drb_Collapse_Spot);
    batch->collapse_jobs = (Drb_Collapse_Job *)(base + static_calls_bytes + static_spots_bytes);
    batch->collapse_jobs_room = static_jobs_bytes / sizeof(Drb_Collapse_Job);
    batch->collapse_sets = (unsigned int *)(base + fixed);
    batch->collapse_sets_room = (batch->static_bytes - fixed) / sizeof(unsigned int);
    batch->no_alloc = 1;

    /* another batch's static step on another stream of this GPU may still be reading the workspace */
    int code = drb_hip_set_device(batch->device);

    if (!code) code = drb_hip_static_workspace_enter(batch->device, hip_stream);

    DrbAmd_Status status = code ? fail_hip("static workspace", code) : drb_amd_batch_process_dense
    (
        batch, source_base, source_stream_stride, source_channel_stride,
        target_base, target_stream_stride, target_channel_stride, frames, capacity, produced_each, hip_stream
    );

    if (!code && (code = drb_hip_static_workspace_leave(batch->device, hip_stream)) != 0 && status == drb_amd_ok)
    {
        status = fail_hip("static workspace", code);
    }

    batch->no_alloc = 0;
    batch->calls_device = own_calls;
    batch->collapse_spots = own_spots;
    batch->collapse_jobs = own_jobs;
    batch->collapse_sets = own_sets;
    batch->collapse_spots_room = own_spots_room;
    batch->collapse_jobs_room = own_jobs_room;
    batch->collapse_sets_room = own_sets_room;

    return status;
}

/* ---- host-fed: copy in, convert and copy out in overlapping slices ---------- */

static int feed_prepare (DrbAmd_Batch * batch, size_t source_bytes, size_t target_bytes)
{
    int code = 0;

    for (int i = 0; i < 3 && !code; i++)
    {
        if (!batch->feed_streams[i]) code = drb_hip_stream_create(&batch->feed_streams[i]);
    }

    for (int i = 0; i < host_slots && !code; i++)
    {
        if (!batch->feed_in[i]) code = drb_hip_event_create(&batch->feed_in[i]);
        if (!code && !batch->feed_done[i]) code = drb_hip_event_create(&batch->feed_done[i]);
        if (!code && !batch->feed_out[i]) code = drb_hip_event_create(&batch->feed_out[i]);
    }

    if (!code && source_bytes > batch->feed_source_bytes)
    {
        for (int i = 0; i < host_slots && !code; i++)
        {
            if (batch->feed_source[i]) drb_hip_free(batch->feed_source[i]);
            batch->feed_source[i] = 0;
            code = drb_hip_alloc(&batch->feed_source[i], source_bytes);
        }

        batch->feed_source_bytes = code ? 0 : source_bytes;
    }

    if (!code && target_bytes > batch->feed_target_bytes)
    {
        for (int i = 0; i < host_slots && !code; i++)
        {
            if (batch->feed_target[i]) drb_hip_free(batch->feed_target[i]);
            batch->feed_target[i] = 0;
            code = drb_hip_alloc(&batch->feed_target[i], target_bytes);
        }

        batch->feed_target_bytes = code ? 0 : target_bytes;
    }

    return code;
}

DrbAmd_Status drb_amd_batch_process_host_dense
    (
        DrbAmd_Batch * batch,
        void const * source_base, long long source_stream_stride, long long source_channel_stride,
        void * target_base, long long target_stream_stride, long long target_channel_stride,
        int frames, int capacity,
        int * produced_each
    )
{
    if (!batch || !source_base || !target_base || source_stream_stride <= 0 || target_stream_stride <= 0)
    {
        return fail(drb_amd_error_bad_argument, "bad argument");
    }

    int const n = batch->config.stream_count;
    int const pull = batch->plan.direction == 1;
    size_t const source_stream_bytes = (size_t)source_stream_stride * batch->sample_bytes;
    size_t const target_stream_bytes = (size_t)target_stream_stride * batch->sample_bytes;

    /* slices of about 256 MB (smaller ones leave the link partly idle: 64 MB
     * slices reach 54 GB/s where 256 MB ones reach 74), at least four of
     * them so that the copies of one overlap the conversion of another */
    char const * const tuned = getenv("DRB_AMD_HOST_SLICE_MB");   /* diagnostic */
    long long const slice_bytes = (long long)(tuned && atoi(tuned) > 0 ? atoi(tuned) : 256) << 20;
    long long per_slice = slice_bytes / (long long)(source_stream_bytes + target_stream_bytes);

    if (per_slice > (n + 3) / 4) per_slice = (n + 3) / 4;
    if (per_slice < 1) per_slice = 1;

    int code = drb_hip_set_device(batch->device);

    if (!code) code = feed_prepare(batch, (size_t)per_slice * source_stream_bytes, (size_t)per_slice * target_stream_bytes);
    if (code) return fail_hip("host feed set-up", code);

    /* Two streams: copies in on one, conversion and copies out on the other.  (With the
     * copies out on a stream of their own they ran at 10 GB/s instead of 45 once the
     * caller's buffers were a few GB: 3.0 instead of 6.7 Gsamples/s for 1024 streams
     * of 10 s.) */
    void * const in = batch->feed_streams[0], * const work = batch->feed_streams[1], * const out = work;
    int * const produced = produced_each ? produced_each : batch->dense_produced;
    DrbAmd_Status status = drb_amd_ok;
    int slice = 0;

    for (int first = 0; first < n && status == drb_amd_ok; first += (int)per_slice, slice++)
    {
        int const count = first + per_slice <= n ? (int)per_slice : n - first;
        int const slot = slice % host_slots;

        /* the slot's buffers are free once the slice that used them is back on the host */
        if (slice >= host_slots) code = drb_hip_stream_wait_event(in, batch->feed_out[slot]);

        if (!code) code = drb_hip_upload_async(batch->feed_source[slot], (char const *)source_base + (size_t)first * source_stream_bytes,
                                               (size_t)count * source_stream_bytes, in);
        if (!code) code = drb_hip_event_record(batch->feed_in[slot], in);
        if (!code) code = drb_hip_stream_wait_event(work, batch->feed_in[slot]);
        if (code) { status = fail_hip("host feed upload", code); break; }

        for (int s = 0; s < n; s++)
        {
            int const mine = s >= first && s < first + count;

            batch->dense_source[s] = mine ? (char const *)batch->feed_source[slot] + (size_t)(s - first) * source_stream_bytes : 0;
            batch->dense_target[s] = mine ? (char *)batch->feed_target[slot] + (size_t)(s - first) * target_stream_bytes : 0;
            batch->dense_frames[s] = mine ? frames : 0;
            batch->dense_capacity[s] = mine ? capacity : 0;
        }

        /* (the produced counts of the streams outside the slice come back as 0) */
        status = drb_amd_batch_process
        (
            batch,
            batch->dense_source, source_channel_stride, pull ? batch->dense_capacity : batch->dense_frames,
            batch->dense_target, target_channel_stride, pull ? batch->dense_frames : batch->dense_capacity,
            batch->feed_counts, work
        );

        if (status != drb_amd_ok) break;

        for (int s = first; s < first + count; s++) produced[s] = batch->feed_counts[s];

        code = drb_hip_event_record(batch->feed_done[slot], work);

        if (!code) code = drb_hip_stream_wait_event(out, batch->feed_done[slot]);
        if (!code) code = drb_hip_download_async((char *)target_base + (size_t)first * target_stream_bytes, batch->feed_target[slot],
                                                 (size_t)count * target_stream_bytes, out);
        if (!code) code = drb_hip_event_record(batch->feed_out[slot], out);
        if (code) { status = fail_hip("host feed download", code); break; }
    }

    /* whatever happened, nothing may still be using the caller's memory */
    int const drained_in = drb_hip_stream_synchronize(in);
    int const drained_work = drb_hip_stream_synchronize(work);
    int const drained_out = drb_hip_stream_synchronize(out);

    if (status != drb_amd_ok) return status;
    if (drained_in || drained_work || drained_out) return fail_hip("host feed", drained_in ? drained_in : drained_work ? drained_work : drained_out);

    return fetch_fault(batch, work);
}

/* Waits for `hip_stream` and brings the kernels' fault word over. */
static DrbAmd_Status fetch_fault (DrbAmd_Batch * batch, void * hip_stream)
{
    int code = drb_hip_download_async(batch->fault_host, batch->fault_device, sizeof(int), hip_stream);

    if (!code) code = drb_hip_stream_synchronize(hip_stream);
    if (code) return fail_hip("hipStreamSynchronize", code);

    int const fault = *batch->fault_host;

    /* A sample that fast arithmetic's pairs of halves cannot carry (beyond +-65519,
     * infinite or not a number) is the caller's data, not a broken kernel: the
     * bit is reported once and cleared, the batch stays usable (the stream that
     * held the sample has it in its history until enough frames have followed;
     * drb_amd_batch_reset clears that too). */
    if (fault & fault_sample_range)
    {
        *batch->fault_host = fault & ~fault_sample_range;

        code = drb_hip_upload_async(batch->fault_device, batch->fault_host, sizeof(int), hip_stream);

        if (!code) code = drb_hip_stream_synchronize(hip_stream);
        if (code) return fail_hip("hipStreamSynchronize", code);

        if (!(fault & ~fault_sample_range)) return fail(drb_amd_error_bad_argument, "sample outside fast arithmetic's range (beyond +-65519, infinite or not a number)");
    }

    if (fault & ~fault_sample_range) return fail(drb_amd_error_device, "kernel range check failed (code %d)", fault & ~fault_sample_range);

    return drb_amd_ok;
}

DrbAmd_Status drb_amd_batch_synchronize (DrbAmd_Batch * batch, void * hip_stream)
{
    if (!batch) return fail(drb_amd_error_bad_argument, "null argument");

    int code = drb_hip_set_device(batch->device);

    if (code) return fail_hip("hipSetDevice", code);

    return fetch_fault(batch, hip_stream);
}

DrbAmd_Status drb_amd_batch_time_kernels (DrbAmd_Batch * batch, int enable)
{
    if (!batch) return fail(drb_amd_error_bad_argument, "null argument");

    if (enable && !batch->kernel_start)
    {
        int code = drb_hip_set_device(batch->device);

        if (!code) code = drb_hip_timing_event_create(&batch->kernel_start);
        if (!code) code = drb_hip_timing_event_create(&batch->kernel_stop);
        if (!code) code = drb_hip_timing_event_create(&batch->step_start);
        if (!code) code = drb_hip_timing_event_create(&batch->step_stop);
        if (code) return fail_hip("hipEventCreate", code);
    }

    batch->timing = enable != 0;
    batch->kernel_timed = 0;

    return drb_amd_ok;
}

DrbAmd_Status drb_amd_batch_kernel_ms (DrbAmd_Batch * batch, float * milliseconds)
{
    if (!batch || !milliseconds) return fail(drb_amd_error_bad_argument, "null argument");
    if (!batch->kernel_timed) return fail(drb_amd_error_bad_argument, "no timed kernel launch yet");

    int code = drb_hip_event_synchronize(batch->kernel_stop);

    if (!code) code = drb_hip_event_elapsed_ms(milliseconds, batch->kernel_start, batch->kernel_stop);
    if (code) return fail_hip("hipEventElapsedTime", code);

    return drb_amd_ok;
}

DrbAmd_Status drb_amd_batch_step_ms (DrbAmd_Batch * batch, float * milliseconds)
{
    if (!batch || !milliseconds) return fail(drb_amd_error_bad_argument, "null argument");
    if (!batch->kernel_timed) return fail(drb_amd_error_bad_argument, "no timed step yet");

    int code = drb_hip_event_synchronize(batch->step_stop);

    if (!code) code = drb_hip_event_elapsed_ms(milliseconds, batch->step_start, batch->step_stop);
    if (code) return fail_hip("hipEventElapsedTime", code);

    return drb_amd_ok;
}

long long drb_amd_allocation_calls (void)
{
    return drb_hip_allocation_calls();
}

int drb_amd_batch_step_groups (DrbAmd_Batch const * batch)
{
    return batch ? batch->group_count : 0;
}

int drb_amd_batch_step_walkers (DrbAmd_Batch const * batch)
{
    return batch ? batch->step_walkers : 0;
}

long drb_amd_host_times (double microseconds [5], int start_over)
{
    long const steps = host_steps;

    for (int i = 0; i < 5; i++)
    {
        if (microseconds) microseconds[i] = host_spent[i];
        if (start_over) host_spent[i] = 0.0;
    }

    if (start_over) host_steps = 0;

    return steps;
}

#ifdef DRB_FAST_STAMPS
/* Diagnostic build only: the cycle counters the stamped fast kernel sums up
 * behind the fault word (see drb_kernels_fast.hip). */
DrbAmd_Status drb_amd_debug_counters (DrbAmd_Batch * batch, unsigned long long * counters, int count)
{
    if (!batch || count > fault_words - 1) return fail(drb_amd_error_bad_argument, "bad argument");

    int code = drb_hip_set_device(batch->device);

    if (!code) code = drb_hip_stream_synchronize(0);
    if (!code) code = drb_hip_download_async(counters, (char *)batch->fault_device + 8, (size_t)count * 8, 0);
    if (!code) code = drb_hip_stream_synchronize(0);
    if (!code) code = drb_hip_memset_async((char *)batch->fault_device + 8, 0, (size_t)count * 8, 0);

    return code ? fail_hip("debug counters", code) : drb_amd_ok;
}
#endif
