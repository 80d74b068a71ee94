/* Host-side planning: which passes run for a conversion, the frame-count
 * arithmetic of every stage, and the sizes the GPU kernels need. */
#ifndef DRB_PLAN_H
#define DRB_PLAN_H

#ifdef __cplusplus
extern "C" {
#endif

enum
{
    drb_pass_interleave = 0,
    drb_pass_deinterleave = 1,
    drb_pass_up2x = 2,      /* arg = half-band order           */
    drb_pass_down2x = 3,    /* arg = half-band order           */
    drb_pass_sinc = 4,      /* arg = taps + 100 * cubic        */
    drb_pass_slicer = 5
};

enum
{
    drb_stage_up2x = 1,
    drb_stage_down2x = 2,
    drb_stage_sinc = 3,
    drb_stage_delay = 4     /* the slicer, seen as a pure delay; arg = block B */
};

enum { drb_max_passes = 10, drb_max_stages = 8 };

typedef struct Drb_Edge
{
    int rate, channels, layout, block, max_blocks;
}
Drb_Edge;

typedef struct Drb_Pass { int kind, arg; } Drb_Pass;

typedef struct Drb_Stage { int kind, arg, in_rate, out_rate; } Drb_Stage;

typedef struct Drb_Plan
{
    int source_rate, target_rate, channels, block_size, max_block_count;
    int direction, quality;

    /* planning order, as upstream lists it: for pull, index 0 is the user's
     * (target) side and data flows from the last pass to the first */
    int pass_count;
    Drb_Pass passes [drb_max_passes];
    Drb_Edge edges [drb_max_passes + 1];

    int throttling;          /* user-side frames per driver sub-block  */
    long buffer_length;      /* floats per work buffer                 */
    long converter_size;     /* bytes upstream's converter block takes */

    /* DSP stages in data-flow order ((de)interleaving is not a stage) */
    int stage_count;
    Drb_Stage stages [drb_max_stages];
    int slicer_block;        /* 0 = no slicer                          */
    int sinc_stage;          /* index into stages, -1 = none           */
    long long source_delta;  /* 2^48 / sinc input rate                 */
    long long target_delta;  /* 2^48 / sinc output rate                */
    int runnable;            /* 0: see drb_plan_resolve                */
    double macs_per_output_sample;
}
Drb_Plan;

/* 1 = resolved; 0 = rejected by validation; -1 = needs six or more 2x stages
 * or more than ten passes (outside upstream's tables).  plan->runnable is 0
 * when the driver could not run it: zero throttling, or a 2x stage exposed to
 * odd frame counts. */
int drb_plan_resolve
    (
        int source_rate, int target_rate, int channels,
        int block_size, int max_block_count, int direction, int quality,
        Drb_Plan * plan
    );

/* ---- per-stream frame arithmetic ---------------------------------------- */

typedef struct Drb_Stream_State
{
    long long phase;        /* fractional stage accumulator              */
    int slicer_index;       /* frames held in the slicer's block         */
}
Drb_Stream_State;

/* What the device needs to know about one stream for one step. */
typedef struct Drb_Step
{
    int source_frames;      /* consumed                                   */
    int target_frames;      /* produced                                   */
    long long phase;        /* accumulator at the start of the step       */
    int delay;              /* frames the delay stage shifts by this step */
}
Drb_Step;

void drb_stream_reset (Drb_Plan const * plan, Drb_Stream_State * state);

/* Advances `state` by one step of `frames` user-side frames (push: source
 * frames offered, pull: target frames wanted) and reports the step.  Returns
 * 0 if a 2x stage would see an odd frame count. */
int drb_stream_step
    (
        Drb_Plan const * plan, Drb_Stream_State * state, int frames, Drb_Step * step
    );

/* What upstream adds to the timestamp it hands to the data callback for a
 * sub-block that starts in `state` (drb-audio-converter.c:3513-3530, 3576,
 * 3603-3620, 3639): every pass's frame-count function adds its latency in
 * planning order -- half-bands order / 2 frames at their higher rate in seconds
 * (1948, 1962, 2214-2224, 2303-2313), the slicer half a block in seconds (2497,
 * 2684) -- except that the sinc stage OVERWRITES the sum with its position in
 * source frames minus taps / 2 + 1 (1521, 1540, 1751-1766).  The mix of
 * seconds and frames is upstream's (its author's TODO at 2497); it is
 * replicated, not repaired, so that callbacks see the same numbers. */
double drb_plan_callback_latency (Drb_Plan const * plan, Drb_Stream_State const * state);

/* ---- kernel geometry ----------------------------------------------------- */

typedef struct Drb_Geometry
{
    int history_frames;     /* source frames of look-back kept per stream      */
    int tile_frames;        /* target frames per workgroup                      */
    int lds_width;          /* floats per channel per staging buffer            */
    int lds_bytes;          /* dynamic LDS per workgroup                        */
    int channel_group;      /* channels per workgroup                           */
}
Drb_Geometry;

/* what a workgroup of the cascade kernel may take of the CU's 160 KB for its two
 * staging buffers (the coefficient bank and the index ranges come on top) */
enum { drb_lds_hard_limit = 148 * 1024 };

/* geometry->lds_bytes is above drb_lds_hard_limit where no tile fits: such a plan does not run */
/* channels_apart: one channel per workgroup where a tile fits that way (chains of half-band
 * stages only, whose channels have nothing in common, in planar buffers: the tile grows
 * and less look-back is recomputed; measured 6 % on 8 channels 96 -> 48 kHz) */
void drb_plan_geometry (Drb_Plan const * plan, int lds_budget_bytes, int channels_apart, Drb_Geometry * geometry);

struct Drb_Fast_Chain;

/* Fills the matrix-core geometry for plans of the shape [up2x] sinc [down2x]
 * whose fixed-point phase drift over a tile stays below the accuracy budget;
 * chain->usable is 0 otherwise.  Returns the source frames of history the
 * fast kernel wants (0 if unusable). */
int drb_plan_fast_geometry (Drb_Plan const * plan, int lds_budget_bytes, struct Drb_Fast_Chain * chain);

/* Turns a usable [up] sinc [down] geometry into the collapsed form (kind 3: one
 * banded product per segment, see drb_hip.h); returns 0 and leaves the chain
 * alone where that form does not apply.  drifting_columns: also where the
 * phase differs noticeably from segment to segment (kind 4 gives every segment
 * its own; kind 3 does not). */
int drb_plan_collapse_geometry (struct Drb_Fast_Chain * chain, int lds_budget_bytes, int drifting_columns);

/* Turns a collapsed geometry (kind 3) into the split form (kind 4: the same
 * filter on the half-precision matrix instruction, see drb_hip.h); returns 0
 * and leaves the chain alone where no kernel of that form fits. */
int drb_plan_split_geometry (struct Drb_Fast_Chain * chain, int lds_budget_bytes);

/* A stretch of a stream's step, whose anchor segment (seg_y target frames each,
 * counted from the step's first) is `anchor_segment`: what its coefficients
 * depend on (rest, parity) and the source frame they are counted from. */
void drb_plan_collapse_key
    (
        struct Drb_Fast_Chain const * chain, long long phase, int delay, long long anchor_segment,
        long long * rest, int * parity, long long * frame
    );

/* The job that builds the coefficient set of (rest, parity): where every row
 * block's window starts.  0 if the windows do not fit the geometry (never, for
 * the bounds the geometry was made from). */
struct Drb_Collapse_Job;
int drb_plan_collapse_job (struct Drb_Fast_Chain const * chain, long long rest, int parity, struct Drb_Collapse_Job * job);

#ifdef __cplusplus
}
#endif

#endif
