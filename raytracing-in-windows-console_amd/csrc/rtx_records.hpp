// rtx_records.hpp -- the pixel record encoders (RayTracing.cu:206-252, 287-332, 370-472, 507-609, 645-751)
// and the compact pixel word they are built from, shared by the trace kernels, rtx_expand_words (rtx_kernels.hip) and the
// Minimize kernels that read pixel words (rtx_post.hip).
#pragma once

#include "rtx_kernels.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtx {

// Three decimal digits of i = 0..255, NUL padded (RayTracing.cu:212-229): d0 | d1 << 8 | d2 << 16.  Computed, not
// loaded: every workgroup fills its LDS table from this at start-up, and a dozen integer instructions are cheaper
// than a round trip to memory at the head of the kernel.
__host__ __device__ constexpr uint32_t digits_word(uint32_t i)
{
    const uint32_t h = i / 100u, r = i - 100u * h, t = r / 10u, u = r - 10u * t;
    return (i >= 100u ? 48u + h : 0u) | ((i >= 10u ? 48u + t : 0u) << 8) | ((48u + u) << 16);
}
static_assert(digits_word(0) == 0x300000u && digits_word(7) == 0x370000u && digits_word(42) == 0x323400u &&
              digits_word(100) == 0x303031u && digits_word(255) == 0x353532u, "digit encoder");

// What a visible pixel's record is made of: the colour bytes (r, g, b; or the xterm-256 index in c0 for the
// 8-bit modes) and the glyph.  RTX_RENDER_COMPACT stores exactly these 4 bytes per pixel and rtx_expand
// builds the record from them later (on the GPU that assembles the frame), so both go through record_words.
struct Fields {
    uint32_t c0, c1, c2, glyph;
};

// The record (5 dwords RGB, 3 dwords 8-bit) of one pixel, App. B of SURVEY.md byte for byte.
template <int MODE>
__device__ __forceinline__ void record_words(bool visible, const Fields& f, const uint32_t* s_digits, uint32_t* w)
{
    constexpr bool kRgb = (MODE == RTX_K_RGB_ASCII || MODE == RTX_K_RGB_PIXEL || MODE == RTX_K_RGB_NORMALS);
    const uint32_t ESC_BR = 0x1bu | (0x5bu << 8); // ESC [
    if (kRgb) {
        if (visible) {
            const uint32_t dr = s_digits[f.c0], dg = s_digits[f.c1], db = s_digits[f.c2];
            const uint32_t kind = (MODE == RTX_K_RGB_ASCII) ? '3' : '4';
            w[0] = ESC_BR | (kind << 16) | ((uint32_t)'8' << 24);
            w[1] = (uint32_t)';' | ((uint32_t)'2' << 8) | ((uint32_t)';' << 16) | ((dr & 255u) << 24);
            w[2] = (dr >> 8) | ((uint32_t)';' << 16) | ((dg & 255u) << 24);
            w[3] = (dg >> 8) | ((uint32_t)';' << 16) | ((db & 255u) << 24);
            w[4] = (db >> 8) | ((uint32_t)'m' << 16) | (f.glyph << 24);
        } else {
            // ESC [ 4 8 ; 2 ; \0 \0 0 ; \0 \0 0 ; \0 \0 0 m ' '
            w[0] = ESC_BR | ((uint32_t)'4' << 16) | ((uint32_t)'8' << 24);
            w[1] = (uint32_t)';' | ((uint32_t)'2' << 8) | ((uint32_t)';' << 16);
            w[2] = ((uint32_t)'0' << 8) | ((uint32_t)';' << 16);
            w[3] = ((uint32_t)'0' << 8) | ((uint32_t)';' << 16);
            w[4] = ((uint32_t)'0' << 8) | ((uint32_t)'m' << 16) | ((uint32_t)' ' << 24);
        }
    } else {
        if (visible) {
            const uint32_t d = s_digits[f.c0];
            const uint32_t kind = (MODE == RTX_K_BIT_ASCII) ? '3' : '4';
            w[0] = ESC_BR | (kind << 16) | ((uint32_t)'8' << 24);
            w[1] = (uint32_t)';' | ((uint32_t)'5' << 8) | ((uint32_t)';' << 16) | ((d & 255u) << 24);
            w[2] = (d >> 8) | ((uint32_t)'m' << 16) | (f.glyph << 24);
        } else {
            // ESC [ 4 8 ; 5 ; \0 1 6 m ' '
            w[0] = ESC_BR | ((uint32_t)'4' << 16) | ((uint32_t)'8' << 24);
            w[1] = (uint32_t)';' | ((uint32_t)'5' << 8) | ((uint32_t)';' << 16);
            w[2] = (uint32_t)'1' | ((uint32_t)'6' << 8) | ((uint32_t)'m' << 16) | ((uint32_t)' ' << 24);
        }
    }
}

// Compact pixel word (RTX_RENDER_COMPACT): c0 | c1<<8 | c2<<16 | glyph<<24 for a visible pixel (the glyph byte
// is never 0), 0 for a pixel beyond the far plane or without a hit, 0xffffffff for the untraced column W-1.
constexpr uint32_t kCompactMiss = 0u, kCompactNewline = 0xffffffffu;

} // namespace rtx
