/* The thin C ABI between the host code (plain C) and the HIP translation unit.
 * Host files include only this header, never a HIP header. */
#ifndef DRB_HIP_H
#define DRB_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { drb_device_max_stages = 8 };

typedef struct Drb_Device_Stage
{
    int kind;    /* drb_stage_up2x / drb_stage_down2x / drb_stage_sinc          */
    int taps;    /* half-band: order / 2; fractional stage: 16 or 32            */
    int table;   /* float offset of the coefficients in the table blob          */
    int cubic;   /* fractional stage: 1 = four-row Catmull-Rom blend            */
}
Drb_Device_Stage;

/* One conversion chain, passed to the kernels by value. */
typedef struct Drb_Device_Chain
{
    int stage_count;         /* filtering stages only                            */
    int channels;
    int sinc_pull;           /* phase convention of the fractional stage         */
    int delay_edge;          /* edge the slicer delay sits on (0 = in front of
                                stage 0 ... stage_count = behind the last), -1   */
    int history;             /* source frames of look-back kept per stream       */
    int tile_frames;         /* target frames per workgroup                      */
    int lds_width;           /* floats per channel per staging buffer            */
    int sample_type;         /* DrbAmd_Sample_Type                               */
    int layout;              /* DrbAmd_Layout                                    */
    int channel_group;       /* channels a workgroup takes through the chain     */
    long long source_delta;  /* 2^48 / fractional stage input rate               */
    long long target_delta;  /* 2^48 / fractional stage output rate              */
    double scale;            /* 1.0 / source_delta                               */
    int sinc_period;         /* fractional stage outputs after which its position
                                recurs (but for the fixed-point drift); 0: unknown */
    int ablate;              /* diagnostic: bit mask of phases to skip (timing only):
                                1 fractional stage, 2 up-samplers, 4 down-samplers,
                                8 source loads, 16 target stores; 32 (tests): the
                                fractional stage's pairs of frames all take the way
                                of pairs that differ in their bank row            */
    Drb_Device_Stage stages [drb_device_max_stages];
}
Drb_Device_Chain;

/* One stream's share of one step. */
typedef struct Drb_Device_Call
{
    unsigned long long source;   /* device address of the chunk's first sample */
    unsigned long long target;
    long long phase;             /* accumulator at the start of the step       */
    int source_frames;
    int target_frames;
    int delay;                   /* slicer delay in frames for this step       */
    int reserved;
}
Drb_Device_Call;

/* Geometry of the matrix-core ("fast") cascade: [up2x] -> sinc -> [down2x].
 * A workgroup owns one channel of one stream and a tile of 16 equal segments
 * per buffer; segment c of every buffer covers the same stretch of time, and
 * segments are whole numbers of periods of the rate ratio apart, so that all
 * 16 see the same fractional phases.  Segment 0 is warm-up (its lookback is
 * not available), segments 1..15 produce the tile's output. */
typedef struct Drb_Fast_Chain
{
    int usable;
    int ablate;                             /* diagnostic: bit mask of phases to skip (timing only) */
    int tile_stride;                        /* diagnostic: tiles between a workgroup's tiles (0: chosen by the launcher) */
    int unit_walk;                          /* split kernel: workgroups that go from stream to stream where every stream has as many tiles
                                               (0: chosen by the launcher, -1: never, else: so many) */
    int rebuild_tiles;                      /* tiles of a stream between exact rebuilds of the sinc coefficients (0: the kernel's default) */
    int no_carry;                           /* diagnostic: every tile warms up on its own segment 0 */
    int channels, sample_type, layout, history;   /* sample_type, layout: of the source */
    int target_sample_type, target_layout;
    int has_up, up_taps, up_table;          /* half-band in front of the sinc stage   */
    int pre_ups, pre_taps [3], pre_table [3]; /* collapsed chains: further half-band up-samplers in front of that one,
                                               the nearest first (a chain of interpolators folded into the one filter) */
    int has_down, down_taps, down_table;    /* half-band behind it                    */
    int sinc_taps, sinc_cubic, sinc_table, sinc_pull;
    int delay_place;                        /* 0 none, 1 behind the sinc stage, 2 in front of it */
    int period_in, period_out;              /* source / target frames of the sinc stage per period */
    int periods;                            /* periods per segment                    */
    int blocks;                             /* 16-row blocks per period               */
    int ksteps;                             /* 4-tap steps of a sinc row block        */
    int seg_x, seg_u, seg_s, seg_y;         /* segment lengths: source, sinc in, sinc out, target */
    int look_x, look_u, look_d;             /* lookback rows kept in front of a segment:
                                               source, sinc input (odd plane when an
                                               up-sampler feeds it), planes of d      */
    int off_x, off_u, off_d_odd, off_d_even; /* LDS float offsets                      */
    int off_bank;                           /* sinc bank while coefficients are rebuilt: shares the planes of d where a down-sampler exists */
    int off_taps;                           /* half-band taps, zero-padded: up, then down */
    int kind;                               /* 0: [up] sinc [down]; 1: one half-band decimator on its own; 2: one interpolator;
                                               3: [up] sinc [down] collapsed into one time-varying filter (see Drb_Collapse_Job),
                                                  a geometry only, on the way to
                                               4: that filter with every float32 carried as two halves (see below) */
    int waves;                              /* waves per workgroup: 4 or 5, whichever deals the row blocks out more evenly */
    float column_drift;                     /* sub-sample position of segment c minus segment c - 1 (0: too small to matter) */
    int types_per_wave;                     /* row-block types one wave keeps coefficients for */
    int plane_stride, plane_rows;           /* kinds 3, 4: source frames a tile is loaded as (whole 256-frame loads),
                                               and rows one segment needs of them */
    int window;                             /* kinds 3, 4: source frames a row block's 16 target frames depend on, at most */
    int chunks;                             /* kind 4: 32-frame steps of a row block's window (3 or 4)                      */
    int plane_words;                        /* kind 4: words of one channel's source plane in LDS                           */
    int apart_stride;                       /* kind 4: 0: a plane is a straight copy of the source (odd segment lengths);
                                               else seg_x + 1: a word of padding in front of every segment but the first,
                                               so that the segments start an odd number of words apart                      */
    int image_seg;                          /* kind 4: words from one segment's target frames to the next's in the
                                               images of a tile's results in LDS (seg_y, or a few more so that the
                                               four groups of a wave's lanes write to different banks)                */
    int lds_floats;
    int tile_frames;                        /* target frames per tile = 15 * seg_y    */
    long long source_delta, target_delta;
    double scale;
}
Drb_Fast_Chain;

/* Collapsed chains (kinds 3, 4).  [up2x] -> sinc -> [down2x] is one linear map from
 * source frames to target frames whose rows repeat every period of the rate
 * ratio (up to the drift of the fixed-point phase), so a segment of seg_y
 * target frames is ONE banded product: `blocks` row blocks of 16 target frames,
 * each over a window of 4 * ksteps source frames.  The coefficients depend on
 * the phase only; a set is built on the device for every distinct phase the
 * host finds among the streams of a step ("job"), for an anchor segment in the
 * middle of a stretch of tiles, and every tile of the stretch uses it,
 * corrected to first order for the drift since the anchor. */
enum { drb_collapse_max_blocks = 12 };

/* Split chains (kind 4): the collapsed filter on v_mfma_f32_16x16x32_f16, which
 * runs at 16 times the rate of the float32 matrix instruction.  Every float32
 * value v, source sample or coefficient, travels as two halves h = half(v) and
 * l = half((v - h) * 2048), and a product c x is accumulated in float32 as
 * ch xh + (ch xl + cl xh) / 2048; the dropped term cl xl is below 2^-22 |c x|.
 * A window is `chunks` steps of 32 source frames; a workgroup is one wave per
 * row block plus six that load, and takes up to two channels of a stream
 * through a tile (they share the phase, and with it the coefficients).  The drift of the fixed-point phase is not
 * folded into the coefficients: a fourth product with the coefficients'
 * derivatives is scaled per tile AND per segment by the distance from the
 * stretch's anchor, so a set serves a stretch of Drb_Fast_Chain.rebuild_tiles
 * tiles (drb_split_stretch_tiles where that is 0) and every segment gets its
 * own phase. */
enum { drb_split_max_blocks = 10, drb_split_stretch_tiles = 64, drb_split_loaders = 6, drb_split_planes = 3 };

typedef struct Drb_Collapse_Job
{
    long long rest;      /* phase of the anchor segment's first sinc output minus whole source frames: [0, source_delta) */
    int parity;          /* behind an up-sampler: whether that output's window ends on an odd sinc-input frame           */
    int origin;          /* source frame, relative to the anchor's, that row look_x of the plane holds                    */
    int first_row [drb_collapse_max_blocks];   /* LDS row (lookback included) where each row block's window starts    */
}
Drb_Collapse_Job;

/* One stretch of one stream: which coefficient set, and the source frame that
 * row `look_x` of the anchor segment's column holds. */
typedef struct Drb_Collapse_Spot
{
    int job;
    int base_x;
}
Drb_Collapse_Spot;

/* All functions return 0 on success, else a HIP error code. */
int drb_hip_device_count (int * count);
int drb_hip_get_device (int * device);
int drb_hip_set_device (int device);
int drb_hip_device_is_gfx950 (int device, int * yes, int * lds_bytes_per_block);
/* The library's static device workspace on `device` (which must be the calling
 * thread's current device): its address there and its size.  Allocates nothing. */
int drb_hip_static_workspace (int device, void ** pointer, size_t * bytes);
/* Brackets of a step that keeps its tables there: every batch of a GPU shares the
 * workspace, so a step on another stream than the last one's waits for that one
 * (an event that exists since the look-up: no allocation here either). */
int drb_hip_static_workspace_enter (int device, void * stream);
int drb_hip_static_workspace_leave (int device, void * stream);
/* How often the library has allocated or freed so far: device or pinned memory,
 * streams and events through the functions below, host memory through
 * drb_hip_count_allocation (which the host files call beside malloc and free). */
long long drb_hip_allocation_calls (void);
void drb_hip_count_allocation (void);
int drb_hip_alloc (void ** pointer, size_t bytes);
int drb_hip_free (void * pointer);
int drb_hip_host_alloc (void ** pointer, size_t bytes);
int drb_hip_host_free (void * pointer);
void drb_hip_clear_error (void);
int drb_hip_memset_async (void * pointer, int value, size_t bytes, void * stream);
int drb_hip_upload_async (void * device, void const * host, size_t bytes, void * stream);
/* the same for up to three small tables in pinned host memory (hipHostMalloc), fetched by one kernel instead of DMA engines */
int drb_hip_upload_tables (void * const * device, void const * const * pinned_host, size_t const * bytes, int count, void * stream);
int drb_hip_download_async (void * host, void const * device, size_t bytes, void * stream);
int drb_hip_stream_synchronize (void * stream);
int drb_hip_stream_create (void ** stream);
int drb_hip_stream_destroy (void * stream);
int drb_hip_stream_wait_event (void * stream, void * event);
int drb_hip_event_create (void ** event);
int drb_hip_timing_event_create (void ** event);
int drb_hip_event_elapsed_ms (float * ms, void * start, void * stop);
int drb_hip_event_destroy (void * event);
int drb_hip_event_record (void * event, void * stream);
int drb_hip_event_synchronize (void * event);
char const * drb_hip_error_text (int code);

/* Filters one step of every stream: grid = stream_count x max_tiles. */
int drb_hip_launch_cascade
    (
        Drb_Device_Chain const * chain, int exact,
        Drb_Device_Call const * calls, int stream_count, int max_tiles,
        float const * history,
        long long source_channel_stride, long long target_channel_stride,
        int lds_bytes, int * fault_flag, void * stream
    );

/* 1 if a kernel image was built for this chain's filter lengths. */
int drb_hip_fast_kernel_exists (Drb_Fast_Chain const * chain);

/* Matrix-core cascade: grid = stream_count x channels x max_tiles. */
int drb_hip_launch_fast
    (
        Drb_Fast_Chain const * chain,
        Drb_Device_Call const * calls, int stream_count, int max_tiles,
        float const * history,
        long long source_channel_stride, long long target_channel_stride,
        int * fault_flag, void * stream
    );

/* Split chains: coefficient sets for `job_count` jobs into `sets`
 * (drb_hip_collapse_set_words(chain) 32-bit words each), then the filter
 * itself: spots[stream * max_stretches + stretch]. */
long long drb_hip_collapse_set_words (Drb_Fast_Chain const * chain);
int drb_hip_launch_collapse_build
    (
        Drb_Fast_Chain const * chain, Drb_Collapse_Job const * jobs, int job_count,
        unsigned int * sets, int * fault_flag, void * stream
    );
/* how many workgroups of such a launch would go from stream to stream (0: none) */
int drb_hip_collapsed_walkers (Drb_Fast_Chain const * chain, int stream_count, int max_tiles, int min_tiles);
int drb_hip_launch_collapsed
    (
        Drb_Fast_Chain const * chain,
        Drb_Device_Call const * calls, int stream_count, int max_tiles, int min_tiles,
        float const * history,
        long long source_channel_stride, long long target_channel_stride,
        Drb_Collapse_Job const * jobs, Drb_Collapse_Spot const * spots, int max_stretches,
        unsigned int const * sets,
        int * fault_flag, void * stream, int * walkers_out   /* (may be null) how many workgroups went from stream to stream, 0: none */
    );

/* Writes every stream's new look-back (the last `history` source frames seen
 * so far) into history_out. */
/* a launch that only changes the buffer format of every stream's step (calls: its source and target, frames = target_frames) */
int drb_hip_launch_convert
    (
        Drb_Device_Call const * calls, int stream_count, long long most_frames, int channels,
        int source_type, int source_layout, long long source_channel_stride,
        int target_type, int target_layout, long long target_channel_stride, void * stream
    );
int drb_hip_launch_history
    (
        Drb_Device_Chain const * chain,
        Drb_Device_Call const * calls, int stream_count,
        float const * history_in, float * history_out,
        long long source_channel_stride, void * stream
    );

#ifdef __cplusplus
}
#endif

#endif
