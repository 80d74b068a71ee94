/*
 * TEST INFRASTRUCTURE ONLY.
 *
 * Drives ANY implementation of the four-function converter API
 * (drb-audio-converter.h:132-201) over whole buffers: it constructs a
 * converter in caller-owned memory, feeds it a schedule of process() calls and
 * records everything observable - output samples, the frame_count of every
 * callback invocation and the converter / work-memory sizes.  The same file is
 * compiled twice: against the upstream source (oracle/_ref, the checker) and
 * against this repository's drop-in library (the thing being checked), so the
 * parity tests compare like with like.
 *
 * HARNESS_API_HEADER selects the public header to compile against.
 */
#include HARNESS_API_HEADER

/* Upstream has no destructor: a converter ends with the caller's memory block.  A
 * library that keeps state of its own per converter names its release function
 * here (-DHARNESS_RELEASE=...), the one call a maintainer adds when switching. */
#ifndef HARNESS_RELEASE
#define HARNESS_RELEASE(converter) ((void)(converter))
#endif

#include <stdlib.h>
#include <string.h>

typedef struct Harness_Sink
{
    float * out;          /* planar [channels][capacity] */
    long capacity;
    long written;
    int channels;
    int * counts;         /* callback frame_count log */
    int counts_capacity;
    int counts_written;
    double * stamps;      /* callback timestamp log (may be NULL) */
    int overflow;
}
Harness_Sink;

static void harness_consume
    (
        void * state, double timestamp,
        DrB_Audio_Converter_Buffer const buffers [], int frame_count
    )
{
    Harness_Sink * const sink = state;

    if (sink->counts_written < sink->counts_capacity)
    {
        sink->counts[sink->counts_written] = frame_count;

        if (sink->stamps)
        {
            sink->stamps[sink->counts_written] = timestamp;
        }
    }

    sink->counts_written++;

    if (sink->written + frame_count > sink->capacity)
    {
        sink->overflow = 1;
        return;
    }

    for (int channel = 0; channel < sink->channels; channel++)
    {
        memcpy
        (
            sink->out + channel * sink->capacity + sink->written,
            buffers[channel].samples,
            (size_t)frame_count * sizeof(float)
        );
    }

    sink->written += frame_count;
}

typedef struct Harness_Source
{
    float const * in;     /* planar [channels][length] */
    long length;
    long read;
    int channels;
    int * counts;
    int counts_capacity;
    int counts_written;
    double * stamps;
    int underflow;
}
Harness_Source;

static void harness_produce
    (
        void * state, double timestamp,
        DrB_Audio_Converter_Buffer const buffers [], int frame_count
    )
{
    Harness_Source * const source = state;

    if (source->counts_written < source->counts_capacity)
    {
        source->counts[source->counts_written] = frame_count;

        if (source->stamps)
        {
            source->stamps[source->counts_written] = timestamp;
        }
    }

    source->counts_written++;

    for (int channel = 0; channel < source->channels; channel++)
    {
        for (int frame = 0; frame < frame_count; frame++)
        {
            long const at = source->read + frame;

            /* past the end of the supplied data the producer yields silence */
            buffers[channel].samples[frame] =
                at < source->length ? source->in[channel * source->length + at] : 0.0f;
        }
    }

    if (source->read + frame_count > source->length)
    {
        source->underflow = 1;
    }

    source->read += frame_count;
}

/* Returns converter size in bytes (0 = rejected by validation); also reports
 * the work-memory size of a constructed converter. */
extern long harness_sizes
    (
        int source_rate, int target_rate, int channels,
        int block_size, int max_block_count, int direction, int quality,
        long * work_size
    )
{
    long alignment = 0, size = 0;

    if (!drb_audio_converter_alignment_and_size
        (
            source_rate, target_rate, channels, block_size, max_block_count,
            (DrB_Audio_Converter_Direction)direction,
            (DrB_Audio_Converter_Quality)quality, &alignment, &size
        ))
    {
        return 0;
    }

    void * const memory = aligned_alloc(64, (size_t)((size + 63) & ~63L));

    DrB_Audio_Converter_Data_Callback const callback = { 0 };

    DrB_Audio_Converter * const converter = drb_audio_converter_construct
    (
        memory, source_rate, target_rate, channels, block_size, max_block_count,
        (DrB_Audio_Converter_Direction)direction,
        (DrB_Audio_Converter_Quality)quality, callback
    );

    if (converter && work_size)
    {
        long work_alignment = 0;

        drb_audio_converter_work_memory_alignment_and_size
        (
            converter, &work_alignment, work_size
        );
    }

    if (converter) HARNESS_RELEASE(converter);
    free(memory);

    return converter ? size : 0;
}

/* Push: `in` is planar [channels][length]; process() is called once per entry
 * of `chunks` (frame counts summing to <= length).  Output goes to planar
 * `out` [channels][out_capacity].  Returns total output frames, or -1 if the
 * request was rejected, -2 on output overflow. */
extern long harness_push
    (
        int source_rate, int target_rate, int channels,
        int block_size, int max_block_count, int quality,
        float const * in, long length,
        int const * chunks, int chunk_count,
        float * out, long out_capacity,
        int * callback_counts, int callback_capacity, int * callback_total,
        double * callback_stamps, double timestamp
    )
{
    long alignment = 0, size = 0;

    if (!drb_audio_converter_alignment_and_size
        (
            source_rate, target_rate, channels, block_size, max_block_count,
            drb_audio_converter_direction_push,
            (DrB_Audio_Converter_Quality)quality, &alignment, &size
        ))
    {
        return -1;
    }

    Harness_Sink sink =
    {
        .out = out, .capacity = out_capacity, .channels = channels,
        .counts = callback_counts, .counts_capacity = callback_capacity,
        .stamps = callback_stamps
    };

    void * const memory = aligned_alloc(64, (size_t)((size + 63) & ~63L));

    DrB_Audio_Converter_Data_Callback const callback =
    {
        .process = harness_consume, .state = &sink
    };

    DrB_Audio_Converter * const converter = drb_audio_converter_construct
    (
        memory, source_rate, target_rate, channels, block_size, max_block_count,
        drb_audio_converter_direction_push,
        (DrB_Audio_Converter_Quality)quality, callback
    );

    if (!converter)
    {
        free(memory);
        return -1;
    }

    long work_alignment = 0, work_size = 0;

    drb_audio_converter_work_memory_alignment_and_size
    (
        converter, &work_alignment, &work_size
    );

    void * const work = aligned_alloc(64, (size_t)((work_size + 63) & ~63L));

    long offset = 0;

    for (int index = 0; index < chunk_count; index++)
    {
        DrB_Audio_Converter_Buffer buffers [8];

        for (int channel = 0; channel < channels; channel++)
        {
            buffers[channel].samples = (float *)in + channel * length + offset;
        }

        drb_audio_converter_process
        (
            converter, work, timestamp, buffers, chunks[index]
        );

        offset += chunks[index];
    }

    HARNESS_RELEASE(converter);
    free(work);
    free(memory);

    if (callback_total)
    {
        *callback_total = sink.counts_written;
    }

    return sink.overflow ? -2 : sink.written;
}

/* Pull: the producer serves planar `in` [channels][length] in order (silence
 * past its end); process() is called once per entry of `chunks` (target frame
 * counts) and fills planar `out` [channels][sum(chunks)].  Returns the number
 * of source frames the converter asked the producer for, or -1 if rejected. */
extern long harness_pull
    (
        int source_rate, int target_rate, int channels,
        int block_size, int max_block_count, int quality,
        float const * in, long length,
        int const * chunks, int chunk_count,
        float * out, long out_capacity,
        int * callback_counts, int callback_capacity, int * callback_total,
        double * callback_stamps, double timestamp
    )
{
    long alignment = 0, size = 0;

    if (!drb_audio_converter_alignment_and_size
        (
            source_rate, target_rate, channels, block_size, max_block_count,
            drb_audio_converter_direction_pull,
            (DrB_Audio_Converter_Quality)quality, &alignment, &size
        ))
    {
        return -1;
    }

    Harness_Source source =
    {
        .in = in, .length = length, .channels = channels,
        .counts = callback_counts, .counts_capacity = callback_capacity,
        .stamps = callback_stamps
    };

    void * const memory = aligned_alloc(64, (size_t)((size + 63) & ~63L));

    DrB_Audio_Converter_Data_Callback const callback =
    {
        .process = harness_produce, .state = &source
    };

    DrB_Audio_Converter * const converter = drb_audio_converter_construct
    (
        memory, source_rate, target_rate, channels, block_size, max_block_count,
        drb_audio_converter_direction_pull,
        (DrB_Audio_Converter_Quality)quality, callback
    );

    if (!converter)
    {
        free(memory);
        return -1;
    }

    long work_alignment = 0, work_size = 0;

    drb_audio_converter_work_memory_alignment_and_size
    (
        converter, &work_alignment, &work_size
    );

    void * const work = aligned_alloc(64, (size_t)((work_size + 63) & ~63L));

    long offset = 0;

    for (int index = 0; index < chunk_count; index++)
    {
        DrB_Audio_Converter_Buffer buffers [8];

        if (offset + chunks[index] > out_capacity)
        {
            break;
        }

        for (int channel = 0; channel < channels; channel++)
        {
            buffers[channel].samples = out + channel * out_capacity + offset;
        }

        drb_audio_converter_process
        (
            converter, work, timestamp, buffers, chunks[index]
        );

        offset += chunks[index];
    }

    HARNESS_RELEASE(converter);
    free(work);
    free(memory);

    if (callback_total)
    {
        *callback_total = source.counts_written;
    }

    return source.read;
}
