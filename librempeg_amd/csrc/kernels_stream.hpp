// Streaming kernels: every pixel is independent, every wave instruction moves contiguous KiBs (C3a, C5).
#pragma once
#include "kernels_generic.hpp"
#include "wave_util.hpp"

namespace swsk {

// ------------------------------------------------------------------------------------------
// C5: planar float RGB -> planar 4:4:4 YUV with identity filters in both directions
// (planar_rgbf32_to_y/uv input.c:1300-1334 -> hScale16To15/19_c with 1 tap -> lum/chrRange*Jpeg(16)_c ->
//  yuv2plane1_{8,10,16}_c).  Every pixel is independent: lane = 4 pixels (3 x 16-byte float loads, 3 stores),
// all three output planes come from ONE pass over the input (the generic path re-reads and re-quantises the
// three float planes once per output plane).  grid.x over 4-pixel groups of a frame, grid.z = frame.
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) sws_k_f32rgb_to_yuv444_unity(SwsFrameSet fs, SwsDevParams p)
{
    constexpr int PX = 8;                                       // pixels per lane: 2 x 16-byte loads per plane, one 16-byte store per plane
    const int groups = (p.srcW + PX - 1) / PX;
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= (int64_t)groups * p.srcH) return;
    const int y = (int)(item / groups), x = PX * (int)(item % groups);
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int n = min(PX, p.srcW - x);
    float g[PX], b[PX], r[PX];
    const uint8_t *pg = f.src[0] + (int64_t)y * f.srcStride[0] + 4 * x, *pb = f.src[1] + (int64_t)y * f.srcStride[1] + 4 * x,
                  *pr = f.src[2] + (int64_t)y * f.srcStride[2] + 4 * x;
    if (n == PX) {
#pragma unroll
        for (int h = 0; h < PX / 4; h++) {
            const f32x4 vg = *(const SWS_GLOBAL f32x4 *)(pg + 16 * h), vb = *(const SWS_GLOBAL f32x4 *)(pb + 16 * h), vr = *(const SWS_GLOBAL f32x4 *)(pr + 16 * h);
#pragma unroll
            for (int k = 0; k < 4; k++) { g[4 * h + k] = vg[k]; b[4 * h + k] = vb[k]; r[4 * h + k] = vr[k]; }
        }
    } else {
        for (int k = 0; k < PX; k++) {
            const bool in = k < n;
            g[k] = in ? ((const float *)pg)[k] : 0.f; b[k] = in ? ((const float *)pb)[k] : 0.f; r[k] = in ? ((const float *)pr)[k] : 0.f;
        }
    }
    const int32_t *t = p.rgb2yuv;
    int out[3][PX];
#pragma unroll
    for (int k = 0; k < PX; k++) {
        const int gi = f32_to_u16(g[k]), bi = f32_to_u16(b[k]), ri = f32_to_u16(r[k]);
        // 16-bit samples x 15-bit coefficients: 24-bit multiplies, 32-bit wrap-around sums like the C code
        int c[3];
        c[0] = (int)((unsigned)(mad24(t[0], ri, mad24(t[1], gi, __mul24(t[2], bi))) + (int)(0x2001u << 14))) >> 15;
        c[1] = (int)((unsigned)(mad24(t[3], ri, mad24(t[4], gi, __mul24(t[5], bi))) + (int)(0x10001u << 14))) >> 15;
        c[2] = (int)((unsigned)(mad24(t[6], ri, mad24(t[7], gi, __mul24(t[8], bi))) + (int)(0x10001u << 14))) >> 15;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            int v = min((int)(((uint16_t)c[q] * 16384u) >> p.hshift), p.hclip);   // 1-tap hscale of the u16 line
            if (!p.wide) v = (int16_t)v;
            out[q][k] = range_sample(p, v, q != 0);
        }
    }
    // vertical 1-tap writers (yuv2plane1_*): planes Y, U, V
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int pl = q == 0 ? 0 : q == 1 ? 1 : 2;  // yuv444p*: U = plane 1, V = plane 2
        uint8_t *d = f.dst[pl] + (int64_t)y * f.dstStride[pl];
        if (p.dstKind == DSTK_PLANAR16 || p.dstKind == DSTK_PLANARN) {
            uint32_t o[PX];
            if (p.dstKind == DSTK_PLANAR16) {
#pragma unroll
                for (int k = 0; k < PX; k++) o[k] = (uint32_t)clip_u16((out[q][k] + 4) >> 3);
            } else {
                const int shift = 15 - p.dst_bits;
#pragma unroll
                for (int k = 0; k < PX; k++) o[k] = (uint32_t)clip_uintp2((out[q][k] + (1 << (shift - 1))) >> shift, p.dst_bits);
            }
            if (n == PX) {
                u32x4 v = { o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16) };
                gstore16_nt(d + 2 * x, v);
            } else for (int k = 0; k < n; k++) ((uint16_t *)d)[x + k] = (uint16_t)o[k];
        } else {
            const int off = q == 2 ? 3 : 0;
            for (int k = 0; k < n; k++) d[x + k] = (uint8_t)clip_u8_shr(out[q][k] + dither8(p.should_dither, y, x + k + off), 7);
        }
    }
}

// Plane copies between a picture whose planes are not 16-byte aligned and its aligned working copy (launch_plan_le stages such pictures when the
// context runs helper passes, which read and write 16-byte granules): src[k] -> dst[k] for every plane present.  IN: the destination is the aligned
// side (byte loads, dword stores into rows padded to 256 bytes); otherwise the source is (dword loads, byte stores of the visible bytes only).
struct StageExtents { int32_t row_bytes[4], rows[4]; };
constexpr int STAGE_RPW = 8;
template <bool IN>
__global__ void __launch_bounds__(256) sws_k_stage_planes(SwsFrameSet fs, StageExtents e)
{
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int y0 = blockIdx.y * STAGE_RPW;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int rb = U(e.row_bytes[k]), rows = U(e.rows[k]);
        if (!f.src[k] || !f.dst[k] || x >= rb) continue;
        const int nb = min(4, rb - x);
        for (int y = y0; y < min(rows, y0 + STAGE_RPW); y++) {
            const uint8_t *s = f.src[k] + (int64_t)y * f.srcStride[k] + x;
            uint8_t *d = f.dst[k] + (int64_t)y * f.dstStride[k] + x;
            if (IN) {
                uint32_t v = 0;
                for (int b = 0; b < nb; b++) v |= (uint32_t)s[b] << (8 * b);
                *(uint32_t *)d = v;
            } else {
                const uint32_t v = *(const uint32_t *)s;
                for (int b = 0; b < nb; b++) d[b] = (uint8_t)(v >> (8 * b));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Full-chroma packed RGB epilogue (SWS_FULL_CHR_H_INT: RGB -> RGB scaling, 4:4:4 sources, odd widths, the user's full_chroma_int): the strip kernels
// leave the vertical sums of Y, U and V -- all three at the destination size -- as int32 planes (DSTK_RAW32); this pass is
// yuv2rgb_full_X_c_template's tail and yuv2rgb_write_full for the 24 / 32 bpp targets (output.c:2163-2207, :2005-2070): Y = (sum + (1 << 9)) >> 10,
// U / V = (sum + (1 << 9) - (128 << 19)) >> 10, the 30-bit matrix with wrap-around 32-bit arithmetic, clip, >> 22.  Lane = four pixels (three 16-byte
// loads), a wave walks down FULLCHR_RPW rows with the next row's loads in flight; 12- or 16-byte stores.
// ------------------------------------------------------------------------------------------
constexpr int FULLCHR_RPW = 4;
// SRCM of the two epilogues below: 0 = the strip kernels' int32 sums; 1 / 2 = the source planes themselves, 8-bit / 9 .. 15-bit samples -- a 4:4:4 planar
// source at the same size has four identity filters, and one-tap banks turn a sample into the sum (hScale8To15_c / hScale16To15_c of the tap 1 << 14, then the
// tap 1 << 12): (s << 7) << 12, min((s << 14) >> (depth - 1), 32767) << 12.  No strip launch, no working picture (dev_prepare_on: fullchr_direct).
template <int SRCM>
__device__ __forceinline__ void fullchr_fetch4(const uint8_t *plane, int64_t stride, int y, int x, int npx, int sh, int (&out)[4])
{
    if constexpr (SRCM == 0) {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        const i32x4 v = *(const SWS_GLOBAL i32x4 *)(plane + y * stride + 4 * (int64_t)x);   // (working planes: padded to whole 16-byte groups)
        out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
    } else if constexpr (SRCM == 1) {
        const uint8_t *q = plane + y * stride + x;
        uint32_t w = 0;
        if (npx == 4) w = *(const SWS_GLOBAL uint32_t *)q;
        else for (int k = 0; k < npx; k++) w |= (uint32_t)q[k] << (8 * k);      // (the caller's planes: nothing is read past the row)
#pragma unroll
        for (int k = 0; k < 4; k++) out[k] = (int)(((w >> (8 * k)) & 0xFFu) << 19);
    } else {
        const uint16_t *q = (const uint16_t *)(plane + y * stride) + x;
        uint32_t w[2] = { 0, 0 };
        if (npx == 4) { const u32x2 t = *(const SWS_GLOBAL u32x2 *)q; w[0] = t[0]; w[1] = t[1]; }
        else for (int k = 0; k < npx; k++) w[k >> 1] |= (uint32_t)q[k] << (16 * (k & 1));
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t sv = (w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
            out[k] = (int)(min((sv << 14) >> sh, 32767u) << 12);
        }
    }
}


// The rounding constant of the full-chroma packed writers for output row y: 1 << 9 in yuv2rgb_full_X_c_template (output.c:2130-2180) and, where it matters, in
// yuv2rgb_full_1_c_template's one-row form; none in the rows packed_vscale (vscale.c:135-157) gives to yuv2rgb_full_2_c_template (two taps each that sum to 4096:
// bilinear up-scaling, output.c:2225-2252) and to yuv2rgb_full_1_c_template's chroma blend (one luma tap, two chroma taps that sum to 4096, output.c:2289-2306;
// its luma 4 buf0 and its one-row chroma are exact either way).  The alpha sums keep their 1 << 18 in every form.  Round 5; wave-uniform, a few scalar loads per row.
__device__ __forceinline__ unsigned fullchr_row_rnd(const SwsDevParams &p, int y)
{
    const int lfs = U(p.vLumFs), cfs = U(p.vChrFs);
    if (cfs != 2 || lfs > 2) return 1u << 9;      // (the planner keeps such rows away from this kernel when the option no_short_forms is set: dev_plan*.hip)
    const int16_t *cf = p.vChrF + 2 * (int64_t)y;
    const unsigned c0 = (uint16_t)cf[0], c1 = (uint16_t)cf[1];
    if (c0 + c1 != 4096u || c1 > 4096u) return 1u << 9;
    if (lfs == 1) return 0;
    const int16_t *lf = p.vLumF + 2 * (int64_t)y;
    const unsigned l0 = (uint16_t)lf[0], l1 = (uint16_t)lf[1];
    return (l0 + l1 == 4096u && l1 <= 4096u) ? 0u : 1u << 9;
}

// ALPHA: a fourth sum plane (the alpha plane through the luma filters): A = (sum + (1 << 18)) >> 19, clipped the way the writer does (output.c:2193-2201)
template <int BPP, bool ALPHA, int SRCM = 0>
__global__ void __launch_bounds__(256) sws_k_fullchr_rgb(SwsFrameSet fs, SwsDevParams p)
{
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int lane = threadIdx.x & 63;
    const int cx = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = U(p.dstW), H = U(p.dstH);
    if (cx * 256 >= W) return;
    const int x = (cx * 64 + lane) * 4;
    const bool in = x < W;
    const int npx = min(4, W - x);
    const int y0 = blockIdx.y * FULLCHR_RPW, y1 = min(H, y0 + FULLCHR_RPW);
    const uint8_t *pY = f.src[0], *pU = f.src[1], *pV = f.src[2], *pA = ALPHA ? f.src[3] : nullptr;
    const int64_t sY = f.srcStride[0], sU = f.srcStride[1], sV = f.srcStride[2], sA = ALPHA ? f.srcStride[3] : 0;
    const int ssh = U(p.src_depth) - 1;                       // (SRCM == 2: hScale16To15_c's shift)
    const SwsLutParams &L = p.lut;
    const int y_offset = U(L.y_offset), y_coeff = U(L.y_coeff), v2r = U(L.v2r), v2g = U(L.v2g), u2g = U(L.u2g), u2b = U(L.u2b);
    const int r_pos = U(L.r_pos), g_pos = U(L.g_pos), b_pos = U(L.b_pos), a_pos = U(L.a_pos);
    int nY[4] = { 0, 0, 0, 0 }, nU[4] = { 0, 0, 0, 0 }, nV[4] = { 0, 0, 0, 0 }, nA[4] = { 0, 0, 0, 0 };
    auto fetch = [&](int y) {
        if constexpr (SRCM == 3) {      // a gray source (round 5): the luma sums, and the chroma sums of the reference's constant chroma lines -- what sws_k_gray_chroma would have
            fullchr_fetch4<0>(pY, sY, y, x, npx, ssh, nY);      // written into two planes of sums for this kernel to read back: (1 << 14) x the row's vertical chroma taps
            const int fsz = U(p.vChrFs);
            int tsum = 0;
            if (fsz == 1 && U(p.vLumFs) == 1) tsum = 4096;
            else for (int j = 0; j < fsz; j++) tsum += p.vChrF[(int64_t)y * fsz + j];
            const int acc = (int)((uint32_t)(1 << 14) * (uint32_t)tsum);
#pragma unroll
            for (int k = 0; k < 4; k++) nU[k] = nV[k] = acc;
        } else {
            fullchr_fetch4<SRCM>(pY, sY, y, x, npx, ssh, nY); fullchr_fetch4<SRCM>(pU, sU, y, x, npx, ssh, nU); fullchr_fetch4<SRCM>(pV, sV, y, x, npx, ssh, nV);
            if constexpr (ALPHA) fullchr_fetch4<SRCM>(pA, sA, y, x, npx, ssh, nA);
        }
    };
    if (in && y0 < y1) fetch(y0);
    for (int y = y0; y < y1; y++) {
        int vY[4], vU[4], vV[4], vA[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { vY[k] = nY[k]; vU[k] = nU[k]; vV[k] = nV[k]; vA[k] = nA[k]; }
        if (in && y + 1 < y1) fetch(y + 1);
        if (!in) continue;
        uint32_t px[4];
        const unsigned rq = fullchr_row_rnd(p, y);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int Y = (int)((unsigned)vY[k] + rq) >> 10;
            const int Uc = (int)((unsigned)vU[k] + rq - (unsigned)(128 << 19)) >> 10, Vc = (int)((unsigned)vV[k] + rq - (unsigned)(128 << 19)) >> 10;
            Y -= y_offset;
            Y = (int)((unsigned)Y * (unsigned)y_coeff);
            Y = (int)((unsigned)Y + (1u << 21));
            int R = (int)((unsigned)Y + (unsigned)Vc * (unsigned)v2r);
            int G = (int)((unsigned)Y + (unsigned)Vc * (unsigned)v2g + (unsigned)Uc * (unsigned)u2g);
            int B = (int)((unsigned)Y + (unsigned)Uc * (unsigned)u2b);
            if ((R | G | B) & 0xC0000000) { R = clip_uintp2(R, 30); G = clip_uintp2(G, 30); B = clip_uintp2(B, 30); }
            px[k] = ((uint32_t)(R >> 22) << (8 * r_pos)) | ((uint32_t)(G >> 22) << (8 * g_pos)) | ((uint32_t)(B >> 22) << (8 * b_pos));
            if (BPP == 4) {
                int A = 255;
                if constexpr (ALPHA) { A = (int)((unsigned)vA[k] + (1u << 18)) >> 19; if (A & 0x100) A = clip_u8(A); }
                px[k] |= ((uint32_t)A & 0xFFu) << (8 * a_pos);
            }
        }
        uint8_t *d = f.dst[0] + (int64_t)y * f.dstStride[0] + (int64_t)BPP * x;
        if (BPP == 4) {
            if (npx == 4) { const u32x4 o = { px[0], px[1], px[2], px[3] }; *(SWS_GLOBAL u32x4 *)d = o; }
            else for (int k = 0; k < npx; k++) ((uint32_t *)d)[k] = px[k];
        } else {
            if (npx == 4) {
                ((uint32_t *)d)[0] = (px[0] & 0xFFFFFFu) | (px[1] << 24);
                ((uint32_t *)d)[1] = ((px[1] >> 8) & 0xFFFFu) | (px[2] << 16);
                ((uint32_t *)d)[2] = ((px[2] >> 16) & 0xFFu) | (px[3] << 8);
            } else for (int k = 0; k < npx; k++) { d[3 * k] = (uint8_t)px[k]; d[3 * k + 1] = (uint8_t)(px[k] >> 8); d[3 * k + 2] = (uint8_t)(px[k] >> 16); }
        }
    }
}

// The same epilogue for planar RGB destinations of 8 .. 14 bits (gbrp, gbrap, gbrp10le ... and the msb-aligned twins): yuv2gbrp_full_X_c
// (output.c:2342-2421) -- any_vscale always takes the X form for them, and full chroma is forced (utils.c:1270-1286).  Planes G, B, R (, A);
// rounding 1 << (SH - 1) and >> SH with SH = 22 + 8 - depth; alpha: (1 << 18) + sum, clipped to 27 bits, >> (SH - 3).
template <bool WIDE, bool ALPHA, int SRCM = 0>
__global__ void __launch_bounds__(256) sws_k_fullchr_gbrp(SwsFrameSet fs, SwsDevParams p)
{
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int lane = threadIdx.x & 63;
    const int cx = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = U(p.dstW), H = U(p.dstH);
    if (cx * 256 >= W) return;
    const int x = (cx * 64 + lane) * 4;
    const bool in = x < W;
    const int npx = min(4, W - x);
    const int y0 = blockIdx.y * FULLCHR_RPW, y1 = min(H, y0 + FULLCHR_RPW);
    const uint8_t *pY = f.src[0], *pU = f.src[1], *pV = f.src[2], *pA = ALPHA ? f.src[3] : nullptr;
    const int64_t sY = f.srcStride[0], sU = f.srcStride[1], sV = f.srcStride[2], sA = ALPHA ? f.srcStride[3] : 0;
    const int ssh = U(p.src_depth) - 1;                       // (SRCM == 2: hScale16To15_c's shift)
    const SwsLutParams &L = p.lut;
    const int y_offset = U(L.y_offset), y_coeff = U(L.y_coeff), v2r = U(L.v2r), v2g = U(L.v2g), u2g = U(L.u2g), u2b = U(L.u2b);
    const int SH = 22 + 8 - U(p.dst_bits), dsh = U(p.dst_shift);
    int nY[4] = { 0, 0, 0, 0 }, nU[4] = { 0, 0, 0, 0 }, nV[4] = { 0, 0, 0, 0 }, nA[4] = { 0, 0, 0, 0 };
    auto fetch = [&](int y) {
        fullchr_fetch4<SRCM>(pY, sY, y, x, npx, ssh, nY); fullchr_fetch4<SRCM>(pU, sU, y, x, npx, ssh, nU); fullchr_fetch4<SRCM>(pV, sV, y, x, npx, ssh, nV);
        if constexpr (ALPHA) fullchr_fetch4<SRCM>(pA, sA, y, x, npx, ssh, nA);
    };
    if (in && y0 < y1) fetch(y0);
    for (int y = y0; y < y1; y++) {
        int vY[4], vU[4], vV[4], vA[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { vY[k] = nY[k]; vU[k] = nU[k]; vV[k] = nV[k]; vA[k] = nA[k]; }
        if (in && y + 1 < y1) fetch(y + 1);
        if (!in) continue;
        uint32_t g[4], b[4], r[4], a[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int Y = (int)((unsigned)vY[k] + (1u << 9)) >> 10;
            const int Uc = (int)((unsigned)vU[k] + (unsigned)((1 << 9) - (128 << 19))) >> 10, Vc = (int)((unsigned)vV[k] + (unsigned)((1 << 9) - (128 << 19))) >> 10;
            Y -= y_offset;
            Y = (int)((unsigned)Y * (unsigned)y_coeff);
            Y = (int)((unsigned)Y + (1u << (SH - 1)));
            int R = (int)((unsigned)Y + (unsigned)Vc * (unsigned)v2r);
            int G = (int)((unsigned)Y + (unsigned)Vc * (unsigned)v2g + (unsigned)Uc * (unsigned)u2g);
            int B = (int)((unsigned)Y + (unsigned)Uc * (unsigned)u2b);
            if ((R | G | B) & 0xC0000000) { R = clip_uintp2(R, 30); G = clip_uintp2(G, 30); B = clip_uintp2(B, 30); }
            g[k] = (uint32_t)(G >> SH); b[k] = (uint32_t)(B >> SH); r[k] = (uint32_t)(R >> SH);
            if constexpr (ALPHA) {
                int A = (int)((unsigned)vA[k] + (1u << 18));
                if (A & 0xF8000000) A = clip_uintp2(A, 27);
                a[k] = (uint32_t)(A >> (SH - 3));
            }
            if constexpr (WIDE) {
                g[k] = (g[k] << dsh) & 0xFFFFu; b[k] = (b[k] << dsh) & 0xFFFFu; r[k] = (r[k] << dsh) & 0xFFFFu;
                if constexpr (ALPHA) a[k] &= 0xFFFFu;
            } else {
                g[k] &= 0xFFu; b[k] &= 0xFFu; r[k] &= 0xFFu;
                if constexpr (ALPHA) a[k] &= 0xFFu;
            }
        }
        auto put = [&](int plane, const uint32_t (&v)[4]) {
            uint8_t *d = f.dst[plane] + (int64_t)y * f.dstStride[plane] + (int64_t)(WIDE ? 2 : 1) * x;
            if constexpr (WIDE) {
                if (npx == 4) { const u32x2 o = { v[0] | v[1] << 16, v[2] | v[3] << 16 }; *(SWS_GLOBAL u32x2 *)d = o; }
                else for (int k = 0; k < npx; k++) ((uint16_t *)d)[k] = (uint16_t)v[k];
            } else {
                if (npx == 4) *(SWS_GLOBAL uint32_t *)d = v[0] | v[1] << 8 | v[2] << 16 | v[3] << 24;
                else for (int k = 0; k < npx; k++) d[k] = (uint8_t)v[k];
            }
        };
        put(0, g); put(1, b); put(2, r);
        if constexpr (ALPHA) put(3, a);
    }
}

// The epilogue for planar RGB of 16 bits and float32 (gbrp16le, gbrpf32le; round 5) behind the 19-bit strip kernel's int32 sums:
// yuv2gbrp16_full_X_c / yuv2gbrpf32_full_X_c (output.c:2467-2605) from "Y >>= 14" on -- Y = (Σ - 0x40000000 >> 14) + 0x10000, U / V = Σ - (128 << 23) >> 14,
// the 13-bit matrix in 32-bit wrap-around arithmetic, then ((Y + R) >> 14) + (1 << 15) clipped to 16 bits: the 16-bit writer adds Y and R in 64 bits,
// the float writer in 32 (as the reference's two routines do), and the float is float_mult * (float)v with float_mult = 1.0f / 65535.0f.
// Lane = 4 pixels (three 16-byte loads of sums; three 8- or 16-byte stores), a wave walks down FULLCHR_RPW rows with the next row's loads in flight.
template <bool F32>
__global__ void __launch_bounds__(256) sws_k_fullchr_gbrp16(SwsFrameSet fs, SwsDevParams p)
{
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int lane = threadIdx.x & 63;
    const int cx = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = U(p.dstW), H = U(p.dstH);
    if (cx * 256 >= W) return;
    const int x = (cx * 64 + lane) * 4;
    const bool in = x < W;
    const int npx = min(4, W - x);
    const int y0 = blockIdx.y * FULLCHR_RPW, y1 = min(H, y0 + FULLCHR_RPW);
    const uint8_t *pY = f.src[0], *pU = f.src[1], *pV = f.src[2];
    const int64_t sY = f.srcStride[0], sU = f.srcStride[1], sV = f.srcStride[2];
    const SwsLutParams &L = p.lut;
    const int y_offset = U(L.y_offset), y_coeff = U(L.y_coeff), v2r = U(L.v2r), v2g = U(L.v2g), u2g = U(L.u2g), u2b = U(L.u2b);
    int nY[4] = { 0, 0, 0, 0 }, nU[4] = { 0, 0, 0, 0 }, nV[4] = { 0, 0, 0, 0 };
    auto fetch = [&](int y) { fullchr_fetch4<0>(pY, sY, y, x, npx, 0, nY); fullchr_fetch4<0>(pU, sU, y, x, npx, 0, nU); fullchr_fetch4<0>(pV, sV, y, x, npx, 0, nV); };
    if (in && y0 < y1) fetch(y0);
    for (int y = y0; y < y1; y++) {
        int vY[4], vU[4], vV[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { vY[k] = nY[k]; vU[k] = nU[k]; vV[k] = nV[k]; }
        if (in && y + 1 < y1) fetch(y + 1);
        if (!in) continue;
        uint32_t g[4], b[4], r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int Y = ((int)((unsigned)vY[k] - 0x40000000u) >> 14) + 0x10000;
            const int Uc = (int)((unsigned)vU[k] - (unsigned)(128 << 23)) >> 14, Vc = (int)((unsigned)vV[k] - (unsigned)(128 << 23)) >> 14;
            Y -= y_offset;
            Y = (int)((unsigned)Y * (unsigned)y_coeff);
            Y = (int)((unsigned)Y + (unsigned)((1 << 13) - (1 << 29)));
            const int R = (int)((unsigned)Vc * (unsigned)v2r);
            const int G = (int)((unsigned)Vc * (unsigned)v2g + (unsigned)Uc * (unsigned)u2g);
            const int B = (int)((unsigned)Uc * (unsigned)u2b);
            if constexpr (F32) {
                r[k] = (uint32_t)clip_uintp2(((int)((unsigned)Y + (unsigned)R) >> 14) + (1 << 15), 16);
                g[k] = (uint32_t)clip_uintp2(((int)((unsigned)Y + (unsigned)G) >> 14) + (1 << 15), 16);
                b[k] = (uint32_t)clip_uintp2(((int)((unsigned)Y + (unsigned)B) >> 14) + (1 << 15), 16);
            } else {
                r[k] = (uint32_t)clip_uintp2((int)(((int64_t)Y + R) >> 14) + (1 << 15), 16);
                g[k] = (uint32_t)clip_uintp2((int)(((int64_t)Y + G) >> 14) + (1 << 15), 16);
                b[k] = (uint32_t)clip_uintp2((int)(((int64_t)Y + B) >> 14) + (1 << 15), 16);
            }
        }
        auto put = [&](int plane, const uint32_t (&v)[4]) {
            uint8_t *d = f.dst[plane] + (int64_t)y * f.dstStride[plane] + (int64_t)(F32 ? 4 : 2) * x;
            if constexpr (F32) {
                const float float_mult = 1.0f / 65535.0f;
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; k++) o[k] = __fmul_rn(float_mult, (float)(int)v[k]);
                if (npx == 4) { const u32x4 w = { __float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3]) }; *(SWS_GLOBAL u32x4 *)d = w; }
                else for (int k = 0; k < npx; k++) ((float *)d)[k] = o[k];
            } else {
                if (npx == 4) { const u32x2 o = { v[0] | v[1] << 16, v[2] | v[3] << 16 }; *(SWS_GLOBAL u32x2 *)d = o; }
                else for (int k = 0; k < npx; k++) ((uint16_t *)d)[k] = (uint16_t)v[k];
            }
        };
        put(0, g); put(1, b); put(2, r);
    }
}

// The chroma planes of a gray source scaled into a planar / semi-planar YUV picture (round 5; the luma plane is the strip kernels' luma launch).
// The reference gives such a conversion chroma LINES that hold the line buffers' initial value (ff_init_desc_no_chr: fill_ones, slice.c:190-208 -- 1 << 14,
// or 1 << 18 for the 19-bit lines; never h-scaled, never range converted) and runs the chroma writers over them with the real vertical bank: a sample is
// writer(C * sum of row cy's taps), which only the dither pattern varies along a row.  Writers as in strip_body / strip_body_wide: yuv2planeX_8_c / the N-bit
// and 16-bit forms / yuv2nv12cX_c / yuv2p01xcX_c / yuv2nv12cX_16_c (output.c:149-217, :327-357, :468-589); a one-tap bank enters the planar forms as 4096
// (yuv2plane1_*: the coefficient is not looked at), the semi-planar chroma writers multiply by its value.  One thread per chroma column.
__global__ void __launch_bounds__(256) sws_k_gray_chroma(SwsFrameSet fs, SwsDevParams p)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, cy = blockIdx.y;
    const int cW = U(p.chrDstW), cH = U(p.chrDstH);
    if (x >= cW || cy >= cH) return;
    const SwsFramePtrs f = frame_copy(fs, blockIdx.z);
    const int kind = U(p.dstKind), fsz = U(p.vChrFs), bits = U(p.dst_bits), osh = U(p.dst_shift);
    const bool semi = kind == DSTK_NV12 || kind == DSTK_P010 || kind == DSTK_P016;
    const bool rawk = kind == DSTK_RAW32;     // (a gray source into 24 / 32 bpp RGB through the LUT epilogue: the chroma SUMS, int32, for sws_k_lut_rgb; the packed writers' one-tap form
                                              //  -- both banks one tap -- is the X arithmetic with the tap 4096 like the planar one, dev_plan*.hip raw_one_one)
    int tsum = 0;
    if (fsz == 1 && !semi && (!rawk || U(p.vLumFs) == 1)) tsum = 4096;
    else for (int j = 0; j < fsz; j++) tsum += p.vChrF[(int64_t)cy * fsz + j];
    const uint32_t acc = (uint32_t)(p.wide ? 1 << 18 : 1 << 14) * (uint32_t)tsum;     // (32-bit wrap-around like the writers' own sums)
    uint32_t u, v;
    if (rawk) {
        const int up = U(p.u_plane_dst), vp = U(p.v_plane_dst);
        ((uint32_t *)(f.dst[up] + (int64_t)cy * f.dstStride[up]))[x] = acc;
        ((uint32_t *)(f.dst[vp] + (int64_t)cy * f.dstStride[vp]))[x] = acc;
        return;
    }
    if (p.wide) {
        const int val = (int)(acc + (uint32_t)((1 << 14) - 0x40000000));
        u = v = (uint32_t)(0x8000 + min(max(val >> 15, -32768), 32767));
    } else if (kind == DSTK_PLANAR8 || kind == DSTK_NV12) {
        u = (uint32_t)clip_u8_shr((dither8(p.should_dither, cy, x) << 12) + (int)acc, 19);
        v = (uint32_t)clip_u8_shr((dither8(p.should_dither, cy, x + 3) << 12) + (int)acc, 19);
    } else {
        const int shift = 11 + 16 - bits;
        u = v = (uint32_t)(clip_uintp2(((1 << (shift - 1)) + (int)acc) >> shift, bits) << osh);
    }
    if (semi) {
        if (p.uv_swap_dst) { const uint32_t t = u; u = v; v = t; }
        uint8_t *d = f.dst[1] + (int64_t)cy * f.dstStride[1];
        if (kind == DSTK_NV12) ((uint16_t *)d)[x] = (uint16_t)(u | v << 8);
        else ((uint32_t *)d)[x] = u | v << 16;
    } else {
        const int up = U(p.u_plane_dst), vp = U(p.v_plane_dst);
        uint8_t *du = f.dst[up] + (int64_t)cy * f.dstStride[up], *dv = f.dst[vp] + (int64_t)cy * f.dstStride[vp];
        if (kind == DSTK_PLANAR8) { du[x] = (uint8_t)u; dv[x] = (uint8_t)v; }
        else { ((uint16_t *)du)[x] = (uint16_t)u; ((uint16_t *)dv)[x] = (uint16_t)v; }
    }
}

// The same for 16-byte aligned planes: one thread = 16 bytes of a row of each chroma plane (16 / 8 / 4 samples of 8 / 16 / 32 bits; 8 / 4 pairs of a semi-planar row).
// Only the 8-bit writers' dither varies along a row (period 8: output.c:420-431, :468-482); the others store one value.  The thread behind the last whole group
// of a row stores its samples one by one.
__global__ void __launch_bounds__(256) sws_k_gray_chroma_vec(SwsFrameSet fs, SwsDevParams p)
{
    const int cy = blockIdx.y;
    const int cW = U(p.chrDstW);
    const SwsFramePtrs f = frame_copy(fs, blockIdx.z);
    const int kind = U(p.dstKind), fsz = U(p.vChrFs), bits = U(p.dst_bits), osh = U(p.dst_shift);
    const bool semi = kind == DSTK_NV12 || kind == DSTK_P010 || kind == DSTK_P016, rawk = kind == DSTK_RAW32;
    const bool b8 = kind == DSTK_PLANAR8 || kind == DSTK_NV12;
    const int sbytes = rawk ? 4 : b8 ? 1 : 2;                                  // bytes per stored sample
    const int spt = 16 / (sbytes * (semi ? 2 : 1));                            // chroma columns per thread
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * spt;
    if (x0 >= cW) return;
    int tsum = 0;
    if (fsz == 1 && !semi && (!rawk || U(p.vLumFs) == 1)) tsum = 4096;
    else for (int j = 0; j < fsz; j++) tsum += p.vChrF[(int64_t)cy * fsz + j];
    const uint32_t acc = (uint32_t)(p.wide ? 1 << 18 : 1 << 14) * (uint32_t)tsum;
    uint32_t cu, cv;                                                           // the value of the writers without dither
    if (rawk) cu = cv = acc;
    else if (p.wide) { const int val = (int)(acc + (uint32_t)((1 << 14) - 0x40000000)); cu = cv = (uint32_t)(0x8000 + min(max(val >> 15, -32768), 32767)); }
    else if (b8) cu = cv = 0;
    else { const int shift = 11 + 16 - bits; cu = cv = (uint32_t)(clip_uintp2(((1 << (shift - 1)) + (int)acc) >> shift, bits) << osh); }
    auto u8 = [&](int x) -> uint32_t { return (uint32_t)clip_u8_shr((dither8(p.should_dither, cy, x) << 12) + (int)acc, 19); };
    auto v8 = [&](int x) -> uint32_t { return (uint32_t)clip_u8_shr((dither8(p.should_dither, cy, x + 3) << 12) + (int)acc, 19); };
    const bool sw = semi && p.uv_swap_dst;
    const int up = semi ? 1 : U(p.u_plane_dst), vp = semi ? 1 : U(p.v_plane_dst);
    uint8_t *du = f.dst[up] + (int64_t)cy * f.dstStride[up], *dv = f.dst[vp] + (int64_t)cy * f.dstStride[vp];
    if (x0 + spt > cW) {                                                       // the ragged end of a row
        for (int x = x0; x < cW; x++) {
            uint32_t u = b8 ? u8(x) : cu, v = b8 ? v8(x) : cv;
            if (sw) { const uint32_t t = u; u = v; v = t; }
            if (semi) { if (b8) ((uint16_t *)du)[x] = (uint16_t)(u | v << 8); else ((uint32_t *)du)[x] = u | v << 16; }
            else if (sbytes == 1) { du[x] = (uint8_t)u; dv[x] = (uint8_t)v; }
            else if (sbytes == 2) { ((uint16_t *)du)[x] = (uint16_t)u; ((uint16_t *)dv)[x] = (uint16_t)v; }
            else { ((uint32_t *)du)[x] = u; ((uint32_t *)dv)[x] = v; }
        }
        return;
    }
    u32x4 ou, ov;
    if (b8) {                    // x0 is a multiple of 8: the dither pattern of columns 0 .. 7, twice (planar) or interleaved (semi-planar)
        uint32_t pu[2] = { 0, 0 }, pv[2] = { 0, 0 };
#pragma unroll
        for (int k = 0; k < 8; k++) { pu[k >> 2] |= u8(k) << (8 * (k & 3)); pv[k >> 2] |= v8(k) << (8 * (k & 3)); }
        if (semi) {
            const uint32_t *a = sw ? pv : pu, *b = sw ? pu : pv;
            ou = u32x4{ __builtin_amdgcn_perm(b[0], a[0], 0x05010400u), __builtin_amdgcn_perm(b[0], a[0], 0x07030602u),
                        __builtin_amdgcn_perm(b[1], a[1], 0x05010400u), __builtin_amdgcn_perm(b[1], a[1], 0x07030602u) };
            ov = ou;
        } else { ou = u32x4{ pu[0], pu[1], pu[0], pu[1] }; ov = u32x4{ pv[0], pv[1], pv[0], pv[1] }; }
    } else if (sbytes == 2) {
        const uint32_t a = sw ? cv : cu, b = sw ? cu : cv;
        const uint32_t wu = semi ? (a | b << 16) : (cu | cu << 16), wv = semi ? wu : (cv | cv << 16);
        ou = u32x4{ wu, wu, wu, wu }; ov = u32x4{ wv, wv, wv, wv };
    } else { ou = u32x4{ cu, cu, cu, cu }; ov = u32x4{ cv, cv, cv, cv }; }
    const int boff = x0 * sbytes * (semi ? 2 : 1);
    *(u32x4 *)(du + boff) = ou;
    if (!semi) *(u32x4 *)(dv + boff) = ov;
}

// Planar / semi-planar 8-bit YUV into packed 8-bit 4:2:2 at the same size with a vertical chroma step only (yuv420p / nv12 -> yuyv422 / uyvy422 / yvyu422: a decoder's
// picture for a playout card or a V4L2 sink; round 5).  The mixed plan + join (plane pass, chroma strip launch into a planar 4:2:2 working picture, interleave) moved 62 MB
// per 4K frame for 29 MB of pictures; this is the same arithmetic in one pass: luma bytes as they are (identity filters: yuv2422_X_c's (4096 (y << 7) + (1 << 18)) >> 19),
// chroma through hScale8To15_c's one tap (<< 7) and the vertical bank, (sum + (1 << 18)) >> 19 clipped (output.c:880-917; one tap on both sides: yuv2422_1_c, the
// sample itself) -- computed as (sum of tap x byte + (1 << 11)) >> 12, the same number, so that two tap rows go through one v_dot2_i32_i16 ({byte of row j, byte of row
// j + 1} assembled by one v_perm).  One thread = 4 chroma columns = 8 pixels of MJ422_ROWS consecutive rows: every wave instruction reads or writes one contiguous run
// (8 luma bytes, 4 + 4 (NV: 8 interleaved) chroma bytes per tap row -- neighbouring output rows share them: L2 hits -- and 16 bytes out per lane).
constexpr int MJ422_ROWS = 8;
template <bool NV>
__global__ void __launch_bounds__(256) sws_k_mixed_join422(SwsFrameSet fs, SwsDevParams p, int uyvy, int groups)
{
    const int gx = blockIdx.x * 64 + (threadIdx.x & 63);
    if (gx >= groups) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int H = U(p.dstH), cH = U(p.chrSrcH), fsz = U(p.vChrFs);
    const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * MJ422_ROWS, y1 = min(H, y0 + MJ422_ROWS);
    const bool vfirst = U(p.u_plane_dst) == 2;                                  // (yvyu422: the planner swapped the planes of the working picture)
    const bool swapc = NV ? ((U(p.uv_swap_src) != 0) != vfirst) : vfirst;       // the FIRST chroma byte of a group comes from the second source component
    const int pu = NV ? 1 : U(p.u_plane_src), pv = NV ? 1 : U(p.v_plane_src);
    const uint8_t *s0 = pu == 1 ? f.src[1] : f.src[2], *s1 = pv == 1 ? f.src[1] : f.src[2];
    const int64_t st0 = pu == 1 ? f.srcStride[1] : f.srcStride[2], st1 = pv == 1 ? f.srcStride[1] : f.srcStride[2];
    auto chroma_row = [&](int r, uint32_t &cu, uint32_t &cv) {
        r = min(max(r, 0), cH - 1);
        if constexpr (NV) {
            const u32x2 a = *((const SWS_GLOBAL u32x2 *)(s0 + (int64_t)r * st0) + gx);     // {u0 v0 u1 v1} {u2 v2 u3 v3}
            cu = __builtin_amdgcn_perm(a[1], a[0], 0x06040200u); cv = __builtin_amdgcn_perm(a[1], a[0], 0x07050301u);
        } else {
            cu = *((const SWS_GLOBAL uint32_t *)(s0 + (int64_t)r * st0) + gx); cv = *((const SWS_GLOBAL uint32_t *)(s1 + (int64_t)r * st1) + gx);
        }
    };
    for (int y = y0; y < y1; y++) {
        const u32x2 yw = *((const SWS_GLOBAL u32x2 *)(f.src[0] + (int64_t)y * f.srcStride[0]) + gx);
        const int pos = p.vChrPos[y];
        int au[4], av[4];
#pragma unroll
        for (int k = 0; k < 4; k++) au[k] = av[k] = 1 << 11;
        for (int j = 0; j < fsz; j += 2) {
            const int t0 = fsz == 1 ? 4096 : (int)p.vChrF[(int64_t)y * fsz + j], t1 = j + 1 < fsz ? (int)p.vChrF[(int64_t)y * fsz + j + 1] : 0;
            const uint32_t tp = ((uint32_t)t0 & 0xFFFFu) | (uint32_t)t1 << 16;
            uint32_t ua, va, ub, vb;
            chroma_row(pos + j, ua, va);
            chroma_row(pos + (j + 1 < fsz ? j + 1 : j), ub, vb);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t sel = 0x0C000C00u | (uint32_t)k | (uint32_t)(4 + k) << 16;           // {byte k of row j, 0, byte k of row j + 1, 0}
                au[k] = sdot2(__builtin_amdgcn_perm(ub, ua, sel), tp, au[k]);
                av[k] = sdot2(__builtin_amdgcn_perm(vb, va, sel), tp, av[k]);
            }
        }
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t c0 = (uint32_t)clip_u8_shr(au[k], 12), c1 = (uint32_t)clip_u8_shr(av[k], 12);
            if (swapc) { const uint32_t t = c0; c0 = c1; c1 = t; }
            const uint32_t cc = c0 | c1 << 16;
            const uint32_t m = 2 * (k & 1);                                    // the column's luma pair: bytes m, m + 1 of its dword
            const uint32_t sy = m | 4u << 8 | (m + 1) << 16 | 6u << 24, su = 4u | m << 8 | 6u << 16 | (m + 1) << 24;     // y0 c0 y1 c1 / c0 y0 c1 y1
            o[k] = __builtin_amdgcn_perm(cc, yw[k >> 1], uyvy ? su : sy);
        }
        *((SWS_GLOBAL u32x4 *)(f.dst[0] + (int64_t)y * f.dstStride[0]) + gx) = o;
    }
}

// The LUT writers behind the strip kernels' raw sums (dev_prepare_on: fullchr_on == 3): 24 / 32 bpp RGB destinations WITHOUT full chroma whose filters are
// too long for sws_k_strip_rgb (ratios of 4:1 and more: thumbnails for display or inference).  Y sums at the destination size, U / V sums at half the
// width; yuv2rgb_X_c_template (output.c:1795-1850): every sum + (1 << 18) >> 19, then the table look-ups in their closed form (lut_pair, kernels_striprgb.hpp).
// Lane = two pixel pairs, a wave walks down FULLCHR_RPW rows.
template <int BPP>
__global__ void __launch_bounds__(256) sws_k_lut_rgb(SwsFrameSet fs, SwsDevParams p)
{
    __shared__ __attribute__((aligned(16))) u32x2 lds_tab[2][256];
    build_lut_tabs(p.lut, lds_tab[0], lds_tab[1], (int)threadIdx.x);
    __syncthreads();
    const LutTabs T = { (const uint8_t *)lds_tab[0], (const uint8_t *)lds_tab[1] };
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int lane = threadIdx.x & 63;
    const int cx = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = U(p.dstW), H = U(p.dstH);
    const int x = (cx * 64 + lane) * 4;
    if (x >= W) return;
    const int npx = min(4, W - x);                            // 2 or 4: the planner takes even widths only
    const bool swap_rb = BPP == 4 ? p.lut.swap_rb32 != 0 : p.lut.rgb_order != 0;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    const int y0 = blockIdx.y * FULLCHR_RPW, y1 = min(H, y0 + FULLCHR_RPW);
    for (int y = y0; y < y1; y++) {
        const i32x4 vY = *(const SWS_GLOBAL i32x4 *)(f.src[0] + (int64_t)y * f.srcStride[0] + 4 * (int64_t)x);     // (planes padded to whole 16-byte groups)
        const i32x2 vU = *(const SWS_GLOBAL i32x2 *)(f.src[1] + (int64_t)y * f.srcStride[1] + 2 * (int64_t)x);
        const i32x2 vV = *(const SWS_GLOBAL i32x2 *)(f.src[2] + (int64_t)y * f.srcStride[2] + 2 * (int64_t)x);
        uint32_t w[2][2];
        // (rows packed_vscale gives to yuv2rgb_2_c_template -- two taps each that sum to 4096 -- have no rounding constant, output.c:1853-1895; the blend of yuv2rgb_1 rounds like X)
        unsigned rq = 1u << 18;
        if (U(p.vLumFs) == 2 && U(p.vChrFs) == 2) {
            const int16_t *lf = p.vLumF + 2 * (int64_t)y, *cf = p.vChrF + 2 * (int64_t)y;
            const unsigned l0 = (uint16_t)lf[0], l1 = (uint16_t)lf[1], c0 = (uint16_t)cf[0], c1 = (uint16_t)cf[1];
            if (l0 + l1 == 4096u && l1 <= 4096u && c0 + c1 == 4096u && c1 <= 4096u) rq = 0;
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int Y1 = (int)((unsigned)vY[2 * k] + rq) >> 19, Y2 = (int)((unsigned)vY[2 * k + 1] + rq) >> 19;
            const int Uc = (int)((unsigned)vU[k] + rq) >> 19, Vc = (int)((unsigned)vV[k] + rq) >> 19;
            lut_pair<BPP>(p.lut, T, swap_rb, Y1, Y2, Uc, Vc, w[k]);
        }
        uint8_t *d = f.dst[0] + (int64_t)y * f.dstStride[0] + (int64_t)BPP * x;
        if (BPP == 4) {
            if (npx == 4) { const u32x4 o = { w[0][0], w[0][1], w[1][0], w[1][1] }; *(SWS_GLOBAL u32x4 *)d = o; }
            else { const u32x2 o = { w[0][0], w[0][1] }; *(SWS_GLOBAL u32x2 *)d = o; }
        } else {
            ((uint32_t *)d)[0] = w[0][0];
            if (npx == 4) { ((uint32_t *)d)[1] = (w[0][1] & 0xFFFFu) | (w[1][0] << 16); ((uint32_t *)d)[2] = (w[1][0] >> 16) | (w[1][1] << 16); }
            else ((uint16_t *)d)[2] = (uint16_t)w[0][1];
        }
    }
}

// The alpha bytes of a 32 bpp destination written by the LUT writers' strip kernel (sws_k_strip_rgb stores 255): yuva420p -> bgra and the like
// (needAlpha without full chroma).  The A samples went through one more luma launch with the raw writer (int32 sums); this pass is the alpha part of
// yuv2rgb_X_c_template (output.c:1820-1835): A = (sum + (1 << 18)) >> 19 per pixel, and both pixels of a pair are clipped when either has bit 8 set.
// Lane = four pixels: one 16-byte load of sums, a 16-byte read-modify-write of the destination.
__global__ void __launch_bounds__(256) sws_k_alpha_merge32(SwsFrameSet fs, SwsDevParams p)
{
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int lane = threadIdx.x & 63;
    const int cx = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = U(p.dstW), H = U(p.dstH);
    const int x = (cx * 64 + lane) * 4;
    if (x >= W) return;
    const int npx = min(4, W - x);
    const int sh = 8 * U(p.lut.a_pos);
    const uint32_t keep = ~(0xFFu << sh);
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const int y0 = blockIdx.y * FULLCHR_RPW, y1 = min(H, y0 + FULLCHR_RPW);
    for (int y = y0; y < y1; y++) {
        const i32x4 v = *(const SWS_GLOBAL i32x4 *)(f.src[0] + (int64_t)y * f.srcStride[0] + 4 * (int64_t)x);   // (the working plane is padded to whole 16-byte groups)
        int A[4];
#pragma unroll
        for (int k = 0; k < 4; k++) A[k] = (int)((unsigned)v[k] + (1u << 18)) >> 19;
#pragma unroll
        for (int k = 0; k < 4; k += 2)
            if ((A[k] | A[k + 1]) & 0x100) { A[k] = clip_u8(A[k]); A[k + 1] = clip_u8(A[k + 1]); }
        uint32_t *d = (uint32_t *)(f.dst[0] + (int64_t)y * f.dstStride[0] + 4 * (int64_t)x);
        if (npx == 4) {
            u32x4 o = *(const SWS_GLOBAL u32x4 *)d;
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = (o[k] & keep) | (((uint32_t)A[k] & 0xFFu) << sh);
            *(SWS_GLOBAL u32x4 *)d = o;
        } else {
            // (an odd width never gets here -- full chroma is forced for it -- but a width of 4 n + 2 does: the pair rule still sees whole pairs)
            for (int k = 0; k < npx; k++) d[k] = (d[k] & keep) | (((uint32_t)A[k] & 0xFFu) << sh);
        }
    }
}

// ------------------------------------------------------------------------------------------
// C3a: planarToP01xWrapper (swscale_unscaled.c:273-322) for aligned 16-bit sources, streaming form.
// grid.y = luma rows then chroma rows; a lane moves CH x 16 bytes spaced one wave apart, so every load/store
// instruction of a wave covers 1 KiB of contiguous memory; plane pointers live in SGPRs; stores are non-temporal.
// Luma: out = in << shiftY.  Chroma: 4 U + 4 V samples -> 4 interleaved pairs (16 bytes).
// ------------------------------------------------------------------------------------------
template <int CH>
__global__ void __launch_bounds__(256) sws_k_p01x_stream(SwsFrameSet fs, SwsDevParams p, int y0, int nrows)
{
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = blockIdx.y;
    const int chr_rows = (nrows + 1) >> 1;
    auto sh2 = [](uint32_t w, int sh) { return (uint32_t)(uint16_t)((w & 0xFFFF) << sh) | ((uint32_t)(uint16_t)((w >> 16) << sh) << 16); };
    if (r < nrows) {
        const int y = y0 + r;
        const uint8_t *srow = f.src[0] + (int64_t)y * f.srcStride[0];
        uint8_t *drow = f.dst[0] + (int64_t)y * f.dstStride[0];
        const int row_bytes = 2 * p.srcW, sh = p.shiftY;
        u32x4 v[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const int off = ((wave * CH + k) * 64 + lane) * 16;
            if (off + 16 <= row_bytes) v[k] = gload16(srow + off);
            else if (off < row_bytes) v[k] = gload16_partial(srow + off, row_bytes - off);
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const int off = ((wave * CH + k) * 64 + lane) * 16;
            if (off >= row_bytes) continue;
            const u32x4 o = { sh2(v[k][0], sh), sh2(v[k][1], sh), sh2(v[k][2], sh), sh2(v[k][3], sh) };
            if (off + 16 <= row_bytes) gstore16_nt(drow + off, o); else gstore_partial(drow + off, o, row_bytes - off);
        }
    } else if (r < nrows + chr_rows) {
        const int cr = (y0 >> 1) + (r - nrows);
        const int cw = p.srcW / 2;                             // the reference converts srcW/2 chroma samples (:311)
        const uint8_t *su = f.src[1] + (int64_t)cr * f.srcStride[1], *sv = f.src[2] + (int64_t)cr * f.srcStride[2];
        uint8_t *drow = f.dst[1] + (int64_t)cr * f.dstStride[1];
        const int in_bytes = 2 * cw, shu = p.shiftU, shv = p.shiftV;
        u32x2 u[CH], w[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const int off = ((wave * CH + k) * 64 + lane) * 8;   // bytes into the U / V row: 4 samples
            if (off + 8 <= in_bytes) { u[k] = gload8(su + off); w[k] = gload8(sv + off); }
            else if (off < in_bytes) {
                const u32x4 tu = gload16_partial(su + off, in_bytes - off), tv = gload16_partial(sv + off, in_bytes - off);
                u[k][0] = tu[0]; u[k][1] = tu[1]; w[k][0] = tv[0]; w[k][1] = tv[1];
            }
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const int off = ((wave * CH + k) * 64 + lane) * 8;
            if (off >= in_bytes) continue;
            auto il = [&](uint32_t a, uint32_t b) { return (uint32_t)(uint16_t)(a << shu) | ((uint32_t)(uint16_t)(b << shv) << 16); };
            const u32x4 o = { il(u[k][0] & 0xFFFF, w[k][0] & 0xFFFF), il(u[k][0] >> 16, w[k][0] >> 16),
                              il(u[k][1] & 0xFFFF, w[k][1] & 0xFFFF), il(u[k][1] >> 16, w[k][1] >> 16) };
            if (off + 8 <= in_bytes) gstore16_nt(drow + 2 * off, o); else gstore_partial(drow + 2 * off, o, 2 * (in_bytes - off));
        }
    }
}

} // namespace swsk
