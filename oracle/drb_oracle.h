/*
 * TEST INFRASTRUCTURE ONLY - a CPU restatement of the upstream converter's
 * resampling path, used as the checker for the HIP implementation.  The
 * product library never includes, links or calls anything in this directory.
 */
#ifndef DRB_ORACLE_H
#define DRB_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum
{
    drb_oracle_stage_up2x = 1,   /* arg = half-band order 16..96            */
    drb_oracle_stage_down2x = 2, /* arg = half-band order 16..96            */
    drb_oracle_stage_sinc = 3,   /* arg = taps (16|32) + 100 if cubic       */
    drb_oracle_stage_slicer = 4  /* arg = block length B in frames          */
};

enum { drb_oracle_max_stages = 8 };

typedef struct Drb_Oracle_Plan
{
    int source_rate, target_rate, channels, block_size, max_block_count;
    int direction;                 /* 0 push, 1 pull                          */
    int quality;                   /* 0 poor, 1 fine, 2 good, 3 best          */
    int stage_count;               /* DSP stages in data-flow order           */
    int stage_kind [drb_oracle_max_stages];
    int stage_arg [drb_oracle_max_stages];
    int stage_in_rate [drb_oracle_max_stages];
    int stage_out_rate [drb_oracle_max_stages];
    int throttling;                /* user-side frames per driver sub-block   */
    int consistent;                /* 0: a 2x stage can see odd frame counts  */
}
Drb_Oracle_Plan;

/* 1 = planned, 0 = rejected by validation, -1 = outside the supported range
 * (six or more 2x stages, or a zero max_block_count). */
int drb_oracle_plan
    (
        int source_rate, int target_rate, int channels,
        int block_size, int max_block_count, int direction, int quality,
        Drb_Oracle_Plan * plan
    );

/* Cumulative frame counts: target frames emitted after `source_total` pushed
 * source frames / source frames requested after `target_total` pulled frames. */
long drb_oracle_push_count (Drb_Oracle_Plan const * plan, long source_total);
long drb_oracle_pull_count (Drb_Oracle_Plan const * plan, long target_total);

/* Frame counts handed to the data callback for a schedule of process() calls
 * (one count per throttled sub-block).  Returns the number of callbacks. */
int drb_oracle_callback_counts
    (
        Drb_Oracle_Plan const * plan, int const * chunks, int chunk_count,
        int * counts, int capacity
    );

/* Whole-stream conversion, planar float32 [channels][frames].
 * push: returns target frames written (== drb_oracle_push_count), -2 if `out`
 *       is too small.
 * pull: writes exactly target_frames frames; returns the source frames
 *       consumed (== drb_oracle_pull_count); producer data past `source_frames`
 *       reads as silence. */
long drb_oracle_push
    (
        Drb_Oracle_Plan const * plan,
        float const * in, long source_frames, long in_stride,
        float * out, long out_capacity, long out_stride
    );

long drb_oracle_pull
    (
        Drb_Oracle_Plan const * plan,
        float const * in, long source_frames, long in_stride,
        float * out, long target_frames, long out_stride
    );

/* Integer sample conventions of the batched entry point (the upstream library
 * is float-only): int16 <-> float scales by 2^15, int32 by 2^31; float -> int
 * rounds to nearest even and saturates. */
void drb_oracle_i16_to_f32 (short const * in, float * out, long count);
void drb_oracle_i32_to_f32 (int const * in, float * out, long count);
void drb_oracle_f32_to_i16 (float const * in, short * out, long count);
void drb_oracle_f32_to_i32 (float const * in, int * out, long count);

#ifdef __cplusplus
}
#endif

#endif
