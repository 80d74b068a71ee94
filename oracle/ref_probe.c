/*
 * TEST INFRASTRUCTURE ONLY - never linked into the product library.
 *
 * Builds the upstream converter straight from its own source file, where it
 * lies (REF_SOURCE, normally /root/reference/drb-audio-converter.c), by
 * textual inclusion, so that this translation unit can also reach the
 * converter's file-static planner and coefficient tables.  Nothing of the
 * upstream text is copied into this repository; the output (a shared object)
 * goes to oracle/_ref/ which is git-ignored.
 *
 * Build flags must stay scalar: no -msse4.1 / -march=native (the upstream SSE
 * branch does an aligned load on a 4-byte aligned address and crashes), no
 * -ffast-math, no -DDEBUG (an over-strict assert aborts 16 k -> 48 k pull).
 */
#ifndef REF_SOURCE
#  error "pass -DREF_SOURCE='\"/path/to/drb-audio-converter.c\"'"
#endif

#include REF_SOURCE

/* ---- coefficient tables ------------------------------------------------- */

/* ids 0..5 = half-band orders 16..96, 6 = sinc_16_linear, 7 = sinc_32_linear,
 * 8 = sinc_16_cubic, 9 = sinc_32_cubic (the only reachable sinc banks,
 * drb-audio-converter.c:3011-3017). */
extern int ref_probe_table (int id, float * dst, int * rows, int * cols)
{
    float const * src = 0;
    int r = 0, c = 0;

    switch (id)
    {
      case 0: src = kernel_2x_fir_16; r = 1; c = 8; break;
      case 1: src = kernel_2x_fir_32; r = 1; c = 16; break;
      case 2: src = kernel_2x_fir_48; r = 1; c = 24; break;
      case 3: src = kernel_2x_fir_64; r = 1; c = 32; break;
      case 4: src = kernel_2x_fir_80; r = 1; c = 40; break;
      case 5: src = kernel_2x_fir_96; r = 1; c = 48; break;
      case 6: src = &kernels_sinc_16_linear[0][0]; r = 64; c = 16; break;
      case 7: src = &kernels_sinc_32_linear[0][0]; r = 64; c = 32; break;
      case 8: src = &kernels_sinc_16_cubic[0][0]; r = 32; c = 16; break;
      case 9: src = &kernels_sinc_32_cubic[0][0]; r = 32; c = 32; break;
      default: return 0;
    }

    *rows = r;
    *cols = c;

    if (dst)
    {
        memcpy(dst, src, (size_t)r * c * sizeof(float));
    }

    return 1;
}

/* ---- planner known answers ---------------------------------------------- */

/* Fills: passes[10] (pass tags, planning order), formats[11][5]
 * (rate, channels, layout, block size, max block count per edge), and the
 * derived converter integers.  Returns -2 when validation rejects the request,
 * -1 when the request would index past the six-entry filter-order tables or
 * the ten-entry pass list (drb-audio-converter.c:3031-3032, 3074), else the
 * pass count (which may be 0). */
extern int ref_probe_plan
    (
        int source_rate, int target_rate, int channels,
        int block_size, int max_block_count,
        int direction, int quality,
        int * passes, int * formats,
        int * throttling, long * buffer_length
    )
{
    int const pull = direction == drb_audio_converter_direction_pull;

    Format const source_format =
    {
        .sampling_rate = source_rate, .channel_count = channels,
        .layout = layout_deinterleaved,
        .block_size = pull ? block_size : 1,
        .max_block_count = pull ? max_block_count : 0
    };

    Format const target_format =
    {
        .sampling_rate = target_rate, .channel_count = channels,
        .layout = layout_deinterleaved,
        .block_size = pull ? 1 : block_size,
        .max_block_count = pull ? 0 : max_block_count
    };

    if (!validate_conversion(&source_format, &target_format))
    {
        return -2;
    }

    Sampling_Rate_Conversion src_conv = { 0 };

    resolve_sampling_rate_conversion
    (
        pull ? target_rate : source_rate,
        pull ? source_rate : target_rate,
        quality > drb_audio_converter_quality_poor,
        &src_conv
    );

    int const planned_pass_count =
        (channels > 1 ? 2 : 0)
      + src_conv.up_2x_pass_count + src_conv.down_2x_pass_count
      + (src_conv.needs_fractional_resampling ? 2 : 1);

    if (src_conv.up_2x_pass_count >= 6 || src_conv.down_2x_pass_count >= 6
        || planned_pass_count > max_pass_count)
    {
        return -1;
    }

    DrB_Audio_Conversion conversion = { 0 };

    resolve_conversion
    (
        &source_format, &target_format,
        (DrB_Audio_Converter_Direction)direction,
        (DrB_Audio_Converter_Quality)quality,
        &conversion
    );

    for (int index = 0; index < conversion.pass_count; index++)
    {
        passes[index] = conversion.passes[index];
    }

    for (int index = 0; index < conversion.format_count; index++)
    {
        formats[index * 5 + 0] = conversion.formats[index].sampling_rate;
        formats[index * 5 + 1] = conversion.formats[index].channel_count;
        formats[index * 5 + 2] = conversion.formats[index].layout;
        formats[index * 5 + 3] = conversion.formats[index].block_size;
        formats[index * 5 + 4] = conversion.formats[index].max_block_count;
    }

    *throttling =
        conversion.formats[0].block_size * conversion.formats[0].max_block_count;

    *buffer_length = cache_align
    (
        required_buffer_size(conversion.formats, conversion.format_count)
    ) / sizeof(float);

    return conversion.pass_count;
}

/* Fixed-point constants of the fractional stage for a given rate pair
 * (drb-audio-converter.c:1763-1764). */
extern void ref_probe_deltas (int source_rate, int target_rate, long long * sd, long long * td)
{
    *sd = 0x1.0p48 / source_rate;
    *td = 0x1.0p48 / target_rate;
}
