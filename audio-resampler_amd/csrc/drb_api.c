/*
 * Drop-in for the four functions of drb-audio-converter.h, on top of the
 * batched GPU path with a single stream.  The driver loops follow
 * drb-audio-converter.c:3493-3671: each process() call is cut into sub-blocks
 * of at most `throttling` user-side frames and the data callback runs once per
 * sub-block with the frame counts upstream would hand it.  Filtering always
 * runs on the GPU; there is no host fallback.
 */
#include "drb_audio_converter_amd.h"
#include "drb_hip.h"
#include "drb_plan.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* host allocations are counted with the device's (drb_hip_allocation_calls) */
static void * counted_calloc (size_t count, size_t size) { drb_hip_count_allocation(); return calloc(count, size); }
static void counted_free (void * pointer) { if (pointer) drb_hip_count_allocation(); free(pointer); }

typedef struct Converter_Body
{
    struct Converter_Body * next;
    void * owner;                 /* the caller's converter memory              */
    Drb_Plan plan;
    DrB_Audio_Converter_Data_Callback callback;
    DrbAmd_Batch * batch;
    float * device_source;        /* [channel][buffer_length / channels]        */
    float * device_target;
    long channel_capacity;        /* frames per channel in the device buffers   */
    int warned;                   /* a sample outside fast arithmetic's range was reported */
}
Converter_Body;

/* What lives in the caller's converter memory (upstream's block is at least
 * 128 bytes, drb-audio-converter.c:3465-3474, 3690-3694). */
typedef struct Converter_Header
{
    unsigned long long magic;
    Converter_Body * body;
}
Converter_Header;

static unsigned long long const header_magic = 0x44724241756d6431ULL;

static pthread_mutex_t registry_lock = PTHREAD_MUTEX_INITIALIZER;
static Converter_Body * registry = 0;
static int exit_hook_installed = 0;

static void free_body (Converter_Body * body)
{
    if (body->device_source) drb_hip_free(body->device_source);
    if (body->device_target) drb_hip_free(body->device_target);
    if (body->batch) drb_amd_batch_destroy(body->batch);
    counted_free(body);
}

static void release_all (void)
{
    pthread_mutex_lock(&registry_lock);

    while (registry)
    {
        Converter_Body * const body = registry;
        registry = body->next;
        free_body(body);
    }

    pthread_mutex_unlock(&registry_lock);
}

static void unregister_owner (void * owner)
{
    pthread_mutex_lock(&registry_lock);

    for (Converter_Body ** link = &registry; *link; )
    {
        if ((*link)->owner == owner)
        {
            Converter_Body * const body = *link;
            *link = body->next;
            free_body(body);
        }
        else
        {
            link = &(*link)->next;
        }
    }

    pthread_mutex_unlock(&registry_lock);
}

bool drb_audio_converter_alignment_and_size
    (
        int source_sampling_rate, int target_sampling_rate, int channel_count,
        int block_size, int max_block_count,
        DrB_Audio_Converter_Direction direction,
        DrB_Audio_Converter_Quality resampling_quality,
        long * alignment, long * size
    )
{
    Drb_Plan plan;

    if (drb_plan_resolve(source_sampling_rate, target_sampling_rate, channel_count, block_size,
                         max_block_count, direction, resampling_quality, &plan) != 1)
    {
        return false;
    }

    *alignment = 64;
    *size = plan.converter_size;

    return true;
}

DrB_Audio_Converter * drb_audio_converter_construct
    (
        void * memory,
        int source_sampling_rate, int target_sampling_rate, int channel_count,
        int block_size, int max_block_count,
        DrB_Audio_Converter_Direction direction,
        DrB_Audio_Converter_Quality resampling_quality,
        DrB_Audio_Converter_Data_Callback data_callback
    )
{
    if (!memory) return 0;

    /* a converter constructed earlier in the same memory is replaced */
    unregister_owner(memory);

    Converter_Body * const body = counted_calloc(1, sizeof(*body));

    if (!body) return 0;

    if (drb_plan_resolve(source_sampling_rate, target_sampling_rate, channel_count, block_size,
                         max_block_count, direction, resampling_quality, &body->plan) != 1)
    {
        counted_free(body);
        return 0;
    }

    char const * const mode = getenv("DRB_AMD_ARITHMETIC");

    DrbAmd_Batch_Config const config =
    {
        .source_sampling_rate = source_sampling_rate,
        .target_sampling_rate = target_sampling_rate,
        .channel_count = channel_count,
        .block_size = block_size,
        .max_block_count = max_block_count,
        .direction = direction,
        .quality = resampling_quality,
        .sample_type = drb_amd_sample_f32,
        .layout = drb_amd_layout_planar,
        .arithmetic = mode && !strcmp(mode, "fast") ? drb_amd_arithmetic_fast : drb_amd_arithmetic_exact,
        .stream_count = 1,
        .device = -1
    };

    if (drb_amd_batch_create(&config, &body->batch) != drb_amd_ok)
    {
        fprintf(stderr, "drb_audio_converter_construct: %s\n", drb_amd_last_error());
        counted_free(body);
        return 0;
    }

    /* the work buffer length bounds the frames on every edge of the chain */
    body->channel_capacity = body->plan.buffer_length / channel_count + 1;

    size_t const bytes = (size_t)body->channel_capacity * channel_count * sizeof(float);

    if (drb_hip_alloc((void **)&body->device_source, bytes) || drb_hip_alloc((void **)&body->device_target, bytes))
    {
        fprintf(stderr, "drb_audio_converter_construct: device allocation failed\n");
        free_body(body);
        return 0;
    }

    body->callback = data_callback;
    body->owner = memory;

    pthread_mutex_lock(&registry_lock);

    body->next = registry;
    registry = body;

    if (!exit_hook_installed)
    {
        atexit(release_all);
        exit_hook_installed = 1;
    }

    pthread_mutex_unlock(&registry_lock);

    Converter_Header * const header = memory;

    header->magic = header_magic;
    header->body = body;

    return memory;
}

static Converter_Body * body_of (DrB_Audio_Converter * converter)
{
    Converter_Header * const header = (Converter_Header *)converter;

    return header && header->magic == header_magic ? header->body : 0;
}

void drb_amd_converter_release (DrB_Audio_Converter * converter)
{
    if (body_of(converter))
    {
        ((Converter_Header *)converter)->magic = 0;
        unregister_owner(converter);
    }
}

void drb_audio_converter_work_memory_alignment_and_size
    (
        DrB_Audio_Converter * converter, long * alignment, long * size
    )
{
    Converter_Body const * const body = body_of(converter);

    *alignment = 64;
    *size = body ? body->plan.buffer_length * 2 * (long)sizeof(float) : 0;
}

double drb_batch_callback_latency (DrbAmd_Batch const * batch, int stream);   /* drb_batch.c */

static void die (char const * what)
{
    /* process() cannot report failure (drb-audio-converter.h:194-201) */
    fprintf(stderr, "drb_audio_converter_process: %s: %s\n", what, drb_amd_last_error());
    abort();
}

/* Waits for the sub-block.  Only fast arithmetic (DRB_AMD_ARITHMETIC=fast) can
 * meet a sample it cannot carry; upstream converts any float, and process()
 * has no way to refuse one, so that is said once on stderr and the stream goes on
 * (its frames around the sample are not numbers). */
static void settle (Converter_Body * body)
{
    DrbAmd_Status const status = drb_amd_batch_synchronize(body->batch, 0);

    if (status == drb_amd_error_bad_argument)
    {
        if (!body->warned) fprintf(stderr, "drb_audio_converter_process: %s\n", drb_amd_last_error());

        body->warned = 1;
    }
    else if (status != drb_amd_ok) die("synchronize");
}

void drb_audio_converter_process
    (
        DrB_Audio_Converter * converter, void * work_memory, double timestamp,
        DrB_Audio_Converter_Buffer const buffers [], int frame_count
    )
{
    Converter_Body * const body = body_of(converter);

    if (!body) die("not a converter built by this library");

    int const channels = body->plan.channels;
    int const pull = body->plan.direction == 1;
    long const capacity = body->channel_capacity;
    float * const work = work_memory;
    DrB_Audio_Converter_Buffer views [max_channel_count];

    for (int offset = 0, step = 0; offset < frame_count; offset += step)
    {
        int const left = frame_count - offset;
        int other = 0;

        /* the callback's timestamp carries upstream's latency figure for the
         * state this sub-block starts in (drb-audio-converter.c:3576, 3639) */
        double const stamp = timestamp + drb_batch_callback_latency(body->batch, 0);

        step = left < body->plan.throttling ? left : body->plan.throttling;

        if (drb_amd_batch_query(body->batch, &step, &other) != drb_amd_ok) die("frame count");
        if (step > capacity || other > capacity) die("sub-block exceeds the work buffer");

        void const * source = body->device_source;
        void * target = body->device_target;
        int const cap = (int)capacity;
        int produced = 0;

        if (pull)
        {
            /* the producer fills planar views into the work memory
             * (drb-audio-converter.c:3628-3642) */
            for (int c = 0; c < channels; c++) views[c].samples = work + (long)c * other;

            body->callback.process(body->callback.state, stamp, views, other);

            for (int c = 0; c < channels; c++)
            {
                if (other > 0 && drb_hip_upload_async(body->device_source + c * capacity, views[c].samples, (size_t)other * sizeof(float), 0)) die("upload");
            }

            if (drb_amd_batch_process(body->batch, &source, capacity, &other, &target, capacity, &step, &produced, 0) != drb_amd_ok) die("process");

            for (int c = 0; c < channels; c++)
            {
                if (drb_hip_download_async(buffers[c].samples + offset, body->device_target + c * capacity, (size_t)step * sizeof(float), 0)) die("download");
            }

            settle(body);
        }
        else
        {
            for (int c = 0; c < channels; c++)
            {
                if (drb_hip_upload_async(body->device_source + c * capacity, buffers[c].samples + offset, (size_t)step * sizeof(float), 0)) die("upload");
            }

            if (drb_amd_batch_process(body->batch, &source, capacity, &step, &target, capacity, &cap, &produced, 0) != drb_amd_ok) die("process");

            /* the consumer sees planar views into the work memory
             * (drb-audio-converter.c:3565-3579) */
            for (int c = 0; c < channels; c++)
            {
                views[c].samples = work + (long)c * produced;

                if (produced > 0 && drb_hip_download_async(views[c].samples, body->device_target + c * capacity, (size_t)produced * sizeof(float), 0)) die("download");
            }

            settle(body);

            body->callback.process(body->callback.state, stamp, views, produced);
        }
    }
}
