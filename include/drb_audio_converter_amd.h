/*
 * MI355X (gfx950) implementation of the drb-audio-converter resampling path.
 *
 * Part 1 re-declares the upstream public interface, name for name and value
 * for value, so that code written against drb-audio-converter.h links against
 * this library unchanged.  Part 2 is the batched many-stream entry point the
 * upstream library does not have.  Everything is plain C ABI: pointers, sizes
 * and integers; no HIP or torch types appear in any signature (a HIP stream is
 * passed as void *).
 */
#ifndef DRB_AUDIO_CONVERTER_AMD_H
#define DRB_AUDIO_CONVERTER_AMD_H

#include <stdbool.h>

#if defined(__cplusplus)
extern "C" {
#endif

/* ======================================================================== */
/* Part 1 - drop-in for drb-audio-converter.h                               */
/* ======================================================================== */

/* replaces drb-audio-converter.h:24-28 */
static struct { short major, minor, patch; }
const drb_audio_converter_version = { 0, 1, 0 };

/* replaces drb-audio-converter.h:31-46 */
enum
{
    drb_audio_converter_sampling_rate_8000 = 8000,
    drb_audio_converter_sampling_rate_11025 = 11025,
    drb_audio_converter_sampling_rate_16000 = 16000,
    drb_audio_converter_sampling_rate_22050 = 22050,
    drb_audio_converter_sampling_rate_44100 = 44100,
    drb_audio_converter_sampling_rate_48000 = 48000,
    drb_audio_converter_sampling_rate_60000 = 60000,
    drb_audio_converter_sampling_rate_88200 = 88200,
    drb_audio_converter_sampling_rate_96000 = 96000,
    drb_audio_converter_sampling_rate_120000 = 120000,
    drb_audio_converter_sampling_rate_176400 = 176400,
    drb_audio_converter_sampling_rate_192000 = 192000,
    drb_audio_converter_sampling_rate_240000 = 240000
};

/* replaces drb-audio-converter.h:49-58 */
enum
{
    drb_audio_converter_block_size_1 = 1,
    drb_audio_converter_block_size_4 = 4,
    drb_audio_converter_block_size_16 = 16,
    drb_audio_converter_block_size_64 = 64,
    drb_audio_converter_block_size_256 = 256,
    drb_audio_converter_block_size_1024 = 1024,
    drb_audio_converter_block_size_4096 = 4096
};

/* replaces drb-audio-converter.h:61-65 */
enum { min_channel_count = 1, max_channel_count = 8 };

/* replaces drb-audio-converter.h:68-73 */
typedef enum DrB_Audio_Converter_Direction
{
    drb_audio_converter_direction_push,
    drb_audio_converter_direction_pull
}
DrB_Audio_Converter_Direction;

/* replaces drb-audio-converter.h:76-83 */
typedef enum DrB_Audio_Converter_Quality
{
    drb_audio_converter_quality_poor,
    drb_audio_converter_quality_fine,
    drb_audio_converter_quality_good,
    drb_audio_converter_quality_best
}
DrB_Audio_Converter_Quality;

/* replaces drb-audio-converter.h:87-91: one planar float buffer per channel */
typedef struct DrB_Audio_Converter_Buffer { float * samples; } DrB_Audio_Converter_Buffer;

/* replaces drb-audio-converter.h:96-115.  Push: called once per throttled
 * sub-block with converted frames (frame_count is a multiple of block_size,
 * possibly 0); the buffers point into the caller's work memory and are valid
 * only during the call.  Pull: called to FILL the buffers with frame_count
 * source frames. */
typedef struct DrB_Audio_Converter_Data_Callback
{
    void (* process)
        (
            void * state, double timestamp,
            DrB_Audio_Converter_Buffer const buffers [], int frame_count
        );
    void * state;
}
DrB_Audio_Converter_Data_Callback;

typedef struct DrB_Audio_Converter DrB_Audio_Converter; /* drb-audio-converter.h:118 */

/* replaces drb-audio-converter.h:132-143 (drb-audio-converter.c:3834-3911).
 * Returns false exactly when upstream validation rejects the request; the
 * alignment (64) and size are the upstream values. */
extern bool drb_audio_converter_alignment_and_size
    (
        int source_sampling_rate, int target_sampling_rate, int channel_count,
        int block_size, int max_block_count,
        DrB_Audio_Converter_Direction direction,
        DrB_Audio_Converter_Quality resampling_quality,
        long * alignment, long * size
    );

/* replaces drb-audio-converter.h:158-169 (drb-audio-converter.c:3913-3989).
 * The returned pointer equals `memory`.  NULL when validation rejects the
 * request, when no gfx950 device / kernel image is usable, or for the few
 * requests this implementation does not run (see drb_amd_last_error). The
 * converter's device-side state lives outside `memory`; it is released by
 * drb_amd_converter_release or at process exit. */
extern DrB_Audio_Converter * drb_audio_converter_construct
    (
        void * memory,
        int source_sampling_rate, int target_sampling_rate, int channel_count,
        int block_size, int max_block_count,
        DrB_Audio_Converter_Direction direction,
        DrB_Audio_Converter_Quality resampling_quality,
        DrB_Audio_Converter_Data_Callback data_callback
    );

/* replaces drb-audio-converter.h:177-182 (drb-audio-converter.c:3991-4000) */
extern void drb_audio_converter_work_memory_alignment_and_size
    (
        DrB_Audio_Converter * converter, long * alignment, long * size
    );

/* replaces drb-audio-converter.h:194-201 (drb-audio-converter.c:4002-4012,
 * driver loops 3493-3671): same sub-block splitting, same callback frame
 * counts, same sample stream; the filtering runs on the GPU. */
extern void drb_audio_converter_process
    (
        DrB_Audio_Converter * converter, void * work_memory, double timestamp,
        DrB_Audio_Converter_Buffer const buffers [], int frame_count
    );

/* No upstream counterpart (upstream has no destructor because it owns no
 * resources): frees the device-side state of a converter built by
 * drb_audio_converter_construct.  Safe to skip; everything still registered is
 * freed at process exit. */
extern void drb_amd_converter_release (DrB_Audio_Converter * converter);

/* ======================================================================== */
/* Part 2 - batched many-stream entry point (no upstream counterpart)       */
/* ======================================================================== */

/* Semantics: a batch behaves as `stream_count` independent upstream converters
 * of one shared configuration, each fed its own data.  Stream s of the batch
 * produces exactly the sample stream (and frame counts) upstream converter s
 * would, whatever the call pattern.  Sample buffers are DEVICE memory. */

typedef enum DrbAmd_Sample_Type
{
    drb_amd_sample_f32 = 0,  /* upstream's type */
    drb_amd_sample_i16 = 1,  /* x = v / 2^15 on load; round-to-nearest-even, saturating on store */
    drb_amd_sample_i32 = 2   /* x = (float)v / 2^31 on load; same store rule */
}
DrbAmd_Sample_Type;

typedef enum DrbAmd_Layout
{
    drb_amd_layout_planar = 0,      /* [channel][frame], channels `channel_stride` elements apart */
    drb_amd_layout_interleaved = 1  /* [frame][channel] */
}
DrbAmd_Layout;

typedef enum DrbAmd_Arithmetic
{
    /* every multiply and add rounded separately, taps in upstream order:
     * float results are bit-identical to the upstream scalar build */
    drb_amd_arithmetic_exact = 0,
    /* Matrix-core kernels, free summation order.  float32 buffers: within
     * 1e-6 + 1e-5 |y| of upstream's y for samples in [-1, 1] (the same relative
     * bound for larger samples, up to +-65504: they travel as pairs of halves).
     * int16 buffers: within one step of upstream's integer.  int32 buffers:
     * one step of 2^-31 is finer than float32 resolves, so only upstream's own
     * float32, bit for bit, is within it; such batches run the exact kernels
     * whatever this field says (DrbAmd_Plan_Info.fast_path reads 0). */
    drb_amd_arithmetic_fast = 1
}
DrbAmd_Arithmetic;

typedef enum DrbAmd_Status
{
    drb_amd_ok = 0,
    drb_amd_error_invalid_conversion = 1, /* upstream validation rejects it */
    drb_amd_error_unsupported = 2,        /* valid upstream, not run here */
    drb_amd_error_no_device = 3,          /* no gfx950 GPU / kernel image */
    drb_amd_error_out_of_memory = 4,
    drb_amd_error_bad_argument = 5,
    drb_amd_error_capacity = 6,           /* a target buffer is too small */
    drb_amd_error_device = 7              /* a HIP call failed */
}
DrbAmd_Status;

typedef struct DrbAmd_Batch_Config
{
    int source_sampling_rate;
    int target_sampling_rate;
    int channel_count;
    int block_size;             /* as in drb_audio_converter_construct */
    int max_block_count;
    DrB_Audio_Converter_Direction direction;
    DrB_Audio_Converter_Quality quality;
    DrbAmd_Sample_Type sample_type;   /* of both source and target buffers */
    DrbAmd_Layout layout;             /* of both source and target buffers */
    DrbAmd_Arithmetic arithmetic;
    int stream_count;
    int device;                 /* HIP device ordinal, -1 = current device */
}
DrbAmd_Batch_Config;

typedef struct DrbAmd_Batch DrbAmd_Batch;

extern DrbAmd_Status drb_amd_batch_create (DrbAmd_Batch_Config const * config, DrbAmd_Batch ** batch);
extern void drb_amd_batch_destroy (DrbAmd_Batch * batch);

/* Returns every stream (or streams [first, first + count)) to its
 * just-constructed state: zero history, initial phase.  Does not wait for the
 * device: while steps of the batch are in flight the histories are cleared on
 * the HIP stream of the most recent step, behind it, and the next step, on
 * whatever stream, comes behind that. */
extern DrbAmd_Status drb_amd_batch_reset (DrbAmd_Batch * batch, int first, int count);

/* Frame counts the next process call would yield, without changing state.
 * push: frames[s] = source frames offered -> counts[s] = target frames produced.
 * pull: frames[s] = target frames wanted  -> counts[s] = source frames needed. */
extern DrbAmd_Status drb_amd_batch_query
    (
        DrbAmd_Batch const * batch, int const * frames, int * counts
    );

/* One conversion step for every stream of the batch.
 *
 *  source / target   host arrays [stream_count] of device pointers to the
 *                    first sample of each stream's chunk
 *  *_channel_stride  planar layout: elements between consecutive channels of
 *                    one stream; ignored for interleaved buffers
 *  push: source_frames[s] frames are consumed; target_frames[s] is the capacity
 *        of stream s's target buffer; produced[s] receives the frames written
 *        (always a multiple of block_size, possibly 0).
 *  pull: target_frames[s] frames are written; source_frames[s] is the number of
 *        frames available at source[s], which must be at least what
 *        drb_amd_batch_query reports; produced[s] receives the source frames
 *        consumed.
 *
 * Work is enqueued on `hip_stream` (a hipStream_t, NULL = the default stream)
 * and the call returns without waiting for it.  Per-stream filter history and
 * phase carry over to the next call.  At most 2^24 frames per stream per call.
 * Steps of one batch run in the order of the calls: a call on another stream
 * than the one before makes its stream wait for that step.  Calls on one batch
 * must not overlap in time (one thread at a time per batch). */
extern DrbAmd_Status drb_amd_batch_process
    (
        DrbAmd_Batch * batch,
        void const * const * source, long long source_channel_stride,
        int const * source_frames,
        void * const * target, long long target_channel_stride,
        int const * target_frames,
        int * produced,
        void * hip_stream
    );

/* Dense form of drb_amd_batch_process for uniform batches: stream s reads at
 * source_base + s * source_stream_stride elements and writes at target_base +
 * s * target_stream_stride elements; every stream gets the same frame count
 * (`frames`; push: source frames, pull: target frames) and `capacity` bounds
 * the other side (push: room in each target buffer, pull: source frames
 * available).  The per-stream count of the other side is written to
 * produced_each when it is not NULL. */
extern DrbAmd_Status drb_amd_batch_process_dense
    (
        DrbAmd_Batch * batch,
        void const * source_base, long long source_stream_stride, long long source_channel_stride,
        void * target_base, long long target_stream_stride, long long target_channel_stride,
        int frames, int capacity,
        int * produced_each,
        void * hip_stream
    );

/* drb_amd_batch_process_dense for callers that may not allocate: tight serving
 * loops, harnesses that forbid hipMalloc, code that runs while another thread
 * has a HIP graph capture open on the device.  It has no counterpart upstream
 * (whose converter owns no resources at all).  The same step, with the same
 * results; but the device tables a step needs beyond the batch's per-stream
 * state (step descriptors and, for the collapsed matrix-core kernels, the
 * coefficient tables of the step) live in a static 8 MiB workspace inside the
 * library's code object instead of buffers that grow on demand, and the call
 * makes no hipMalloc / hipFree / hipHostMalloc and no host allocation either.
 * The call itself is NOT capturable into a HIP graph: phases, delays and frame
 * counts of the step are worked out on the host and advance with every call (a
 * replayed graph would run the recorded step's tables again), and the call
 * waits on an event of the batch's ring of descriptor slots.
 * Limits, reported as drb_amd_error_capacity:
 * stream_count x launches per step <= 26214 (1 MiB of step descriptors);
 * stream_count x stretches of tiles (32 to 128 tiles each, by rate pair) <= 131072; a chain of several launches
 * must have run a step at least this long through drb_amd_batch_process before
 * (its buffers between launches do not grow here).  Where the streams' phases
 * differ so much that their coefficient sets exceed the workspace (about 47
 * sets), the step runs in groups of streams whose sets fit, one after the other.
 * Every batch on a GPU shares that GPU's workspace: a call on another HIP stream
 * than the last static call's (of whichever batch) waits for that call's work
 * first; calls from DIFFERENT threads on batches of one GPU must not overlap in
 * time (calls on one batch never may, as for drb_amd_batch_process).  The
 * workspace keeps the last call's contents and is rewritten by every call
 * before anything reads it. */
extern DrbAmd_Status drb_amd_batch_process_dense_static
    (
        DrbAmd_Batch * batch,
        void const * source_base, long long source_stream_stride, long long source_channel_stride,
        void * target_base, long long target_stream_stride, long long target_channel_stride,
        int frames, int capacity,
        int * produced_each,
        void * hip_stream
    );

/* The same with source and target in HOST memory (pinned memory copies at the
 * link's speed; pageable memory works, slowly): streams are handed to the
 * device in slices of about 256 MB, and one slice's copy in runs while another
 * is converted and copied back (two internal HIP streams, three slices in
 * flight, staging buffers owned by the batch and grown on demand).  Blocks
 * until every target is back in host memory.  No counterpart upstream, whose caller's thread does
 * the filtering; it is what a host-side data loader calls. */
extern DrbAmd_Status drb_amd_batch_process_host_dense
    (
        DrbAmd_Batch * batch,
        void const * source_base, long long source_stream_stride, long long source_channel_stride,
        void * target_base, long long target_stream_stride, long long target_channel_stride,
        int frames, int capacity,
        int * produced_each
    );

/* Waits for the work enqueued on `hip_stream` and reports any error the
 * kernels flagged since the batch was created (the exact kernels check every
 * index range they derive; the matrix-core kernels read silence where a
 * window reaches behind the history the batch keeps, and flag nothing). */
extern DrbAmd_Status drb_amd_batch_synchronize (DrbAmd_Batch * batch, void * hip_stream);

/* Measurement hook: when enabled, every process call brackets its cascade
 * kernel with HIP events on the launch stream; drb_amd_batch_kernel_ms waits
 * for the most recent bracket and returns its duration in milliseconds. */
extern DrbAmd_Status drb_amd_batch_time_kernels (DrbAmd_Batch * batch, int enable);
extern DrbAmd_Status drb_amd_batch_kernel_ms (DrbAmd_Batch * batch, float * milliseconds);
/* The same for everything the most recent step ran on the device: descriptor
 * upload, coefficient sets, filter kernels, history update. */
extern DrbAmd_Status drb_amd_batch_step_ms (DrbAmd_Batch * batch, float * milliseconds);

/* What the host has spent inside drb_amd_batch_process* since the process began (or
 * since the last call with start_over set), in microseconds, over all batches
 * (a diagnostic: not kept per thread): [0] frame arithmetic, [1] preparing the
 * step, [2] coefficient jobs and descriptors, [3] launches, [4] WAITING for one
 * of a batch's four step slots to come free, which is the device's time: a
 * host that enqueues steps faster than the device runs them spends its time
 * there.  Returns the number of steps. */
extern long drb_amd_host_times (double microseconds [5], int start_over);

/* How often this library has allocated or freed anything so far, on the host or
 * through HIP (device and pinned memory, streams, events): what a caller that
 * may not allocate checks drb_amd_batch_process_dense_static against. */
extern long long drb_amd_allocation_calls (void);

/* In how many groups of streams the most recent step ran its collapsed
 * matrix-core launch: 1, unless the distinct phases of the step's streams
 * needed more coefficient sets than their memory holds at a time (1 GiB, or
 * the static workspace), in which case the groups ran one after the other. */
extern int drb_amd_batch_step_groups (DrbAmd_Batch const * batch);

/* How many workgroups of the most recent step's collapsed launch went on from
 * stream to stream (0: every stream and channel had workgroups of its own,
 * or the step ran other kernels).  Steps of a few thousand frames for each of
 * many streams run that way, where every stream gets as many tiles. */
extern int drb_amd_batch_step_walkers (DrbAmd_Batch const * batch);

/* Planner facts for a configuration (host only, no device needed): the DSP
 * stages in data-flow order and the sizes the kernels derive from them. */
typedef struct DrbAmd_Plan_Info
{
    int stage_count;
    int stage_kind [8];     /* 1 up2x, 2 down2x, 3 sinc, 4 slicer (delay)          */
    int stage_arg [8];      /* filter order | taps + 100 * cubic | block length    */
    int throttling;         /* user-side frames per driver sub-block               */
    long buffer_length;     /* floats per work buffer (upstream value)             */
    long converter_size;    /* bytes, upstream value                               */
    int history_frames;     /* source frames of history kept per stream on device  */
    int tile_frames;        /* target frames per workgroup tile                    */
    double flops_per_output_sample;  /* upstream cascade as written, 2 x MACs      */
    double bytes_per_output_sample;  /* source + target bytes per target sample     */
    int fast_path;          /* 1: fast arithmetic runs the matrix-core kernel,
                               0: it runs the same kernel as exact arithmetic      */
    int fast_tile_frames;   /* target frames per workgroup tile of that kernel      */
    int fast_launches;      /* kernel launches of that path per step (longer chains
                               run their outer 2x stages as launches of their own) */
    double fast_issued_flops_per_output_sample;   /* what that kernel hands to the
                               matrix core per target sample (zero padding of the
                               banded products and the warm-up segment included;
                               one blended kernel per output instead of upstream's
                               two or four dot products)                          */
    int fast_collapsed;     /* 2: [up2x ...] sinc [down2x] run as ONE time-varying
                               filter from source to target frames (its coefficients
                               are the product of the stages', interpolators in
                               front of the fractional stage's launch folded in
                               where their ratio allows), on the half-precision
                               matrix instruction, every float32 carried as two
                               halves; 0: the stages one after the other (1, the
                               same filter in float32, is no longer produced)     */
    double fast_issued_half_flops_per_output_sample;  /* the part of
                               fast_issued_flops_per_output_sample that runs on the
                               half-precision instruction (16 times the float32
                               instruction's rate)                                 */
    char kernel_name [48];  /* the kernel that does most of the work of a step in
                               the configuration's arithmetic mode                 */
    char fast_why_not [96]; /* fast_path 0: what kept the configuration from the
                               matrix-core kernels (which launch of its chain, and
                               what that launch lacks); "" where fast_path is 1   */
}
DrbAmd_Plan_Info;

extern DrbAmd_Status drb_amd_plan_info (DrbAmd_Batch_Config const * config, DrbAmd_Plan_Info * info);

/* The resolved pass list in upstream's planning order (for pull, entry 0 is the
 * user's side) with the format of every edge, as upstream's planner resolves
 * them (drb-audio-converter.c:3159-3214).  passes: pass_count pairs {kind, arg}
 * with kind 0 interleaver, 1 deinterleaver, 2 up2x, 3 down2x, 4 sinc, 5 slicer;
 * edges: (pass_count + 1) rows {rate, channels, layout, block size, max block
 * count}.  Works for every request upstream validation accepts, also those
 * this library declines to run; returns their pass count, or -1. */
extern int drb_amd_plan_describe
    (
        DrbAmd_Batch_Config const * config,
        int passes [20], int edges [55], int * throttling, long * buffer_length
    );

/* Text of the most recent error on the calling thread ("" if none). */
extern char const * drb_amd_last_error (void);

#if defined(__cplusplus)
}
#endif

#endif
