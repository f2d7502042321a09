// Packed 24 / 32 bpp RGB sources into 8-bit planar / semi-planar YUV of the same size (the capture -> encoder conversion):
// rgb24ToY_c / rgb24ToUV_half_c (input.c:1068-1172), rgb16_32ToY / UV_half_c_template with the 32-bit rows (:264-393), the identity
// hScale16To15_c (swscale.c:106-131, sh = 9 for RGB sources), yuv2plane1_8_c for luma (output.c:468-493) and yuv2planeX_8_c /
// yuv2nv12cX_c for the vertically scaled chroma (:438-466, :495-528).
//
// Each lane owns four luma columns (= two chroma columns of the "half" chroma readers) and walks down a band of rows: one 12- or
// 16-byte load per source row, four luma bytes stored as one dword.  The two {U, V} pairs of the row go into a 16-deep (8-deep for short filters) ring of
// lane-private LDS slots - LDS as a register file that can be indexed by a run-time row number - from which the vertical chroma filter
// of an output row is read once its last source row has passed.  No barriers: a lane only ever reads what it wrote.
#pragma once
#include "kernels_common.hpp"
#include "wave_util.hpp"

namespace swsk {

struct RgbSrcGeom { int32_t band_rows, bands; const SwsRgbSrcRow *rows; };

// One row entry through the scalar data cache (s_load_dwordx*): loads from the constant address space, because a plain load inside a loop
// that also stores to global memory is not provably unclobbered and becomes a vector load with a full memory latency in the dependence chain.
__device__ __forceinline__ SwsRgbSrcRow load_rgbsrc_row(const SwsRgbSrcRow *rows, int idx)
{
    typedef const uint32_t __attribute__((address_space(4))) *cptr;
    cptr q = (cptr)(uintptr_t)(rows + idx);
    SwsRgbSrcRow e;
    e.first = (int)q[0]; e.last = (int)q[1];
#pragma unroll
    for (int k = 0; k < 8; k++) e.vt[k] = q[4 + k];
    return e;
}

template <int BPP, bool NV, int RING>
__global__ void __launch_bounds__(256) sws_k_rgbsrc_unity(SwsFrameSet fs, SwsDevParams p, RgbSrcGeom g)
{
    __shared__ uint2 ring[RING][256];     // RING = 8 when the vertical chroma filter has at most 8 taps: 16 KB, six waves per SIMD
    const int tid = threadIdx.x, t = blockIdx.x * 256 + tid;
    const int W = U(p.dstW), H = U(p.dstH), x0 = 4 * t;
    if (x0 >= W) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);   // (wave-uniform: the row addressing stays on the scalar unit)
    const int vs = U(p.chrDstVSub), cW = U(p.chrDstW), cH = U(p.chrDstH), cSH = U(p.chrSrcH);
    const int y0 = blockIdx.y * U(g.band_rows), y1 = min(H, y0 + U(g.band_rows));
    int cy = y0 >> vs;
    const int cy1 = min(cH, (y1 + (1 << vs) - 1) >> vs);
    const int vfs = U(p.vChrFs);
    const SwsRgbSrcRow *rows = U(g.rows);
    const int hshift = U(p.hshift), hclip = U(p.hclip);
    auto clampc = [&](int r) { return min(max(r, 0), cSH - 1); };
    SwsRgbSrcRow e = load_rgbsrc_row(rows, min(cy, cH - 1));
    int clast = cy < cy1 ? clampc(e.last) : -1;
    const int cfirst = cy < cy1 ? clampc(e.first) : 0x7fffffff;
    const int rlo = min(y0, cfirst), rhi = cy < cy1 ? max(y1 - 1, clampc(load_rgbsrc_row(rows, cy1 - 1).last)) : y1 - 1;

    // per-byte coefficients: rgb24 has r / b at byte 0 / 2 or 2 / 0; the 32-bit rows have every component at any of the four bytes.  Packed for
    // v_dot2_i32_i16 against a pixel split into {byte 0, byte 2} and {byte 1, byte 3} halves (every coefficient fits int16: host check)
    const Rgb2YuvRow ty = rgb2yuv_row(p.rgb2yuv, 0), tu = rgb2yuv_row(p.rgb2yuv, 3), tv = rgb2yuv_row(p.rgb2yuv, 6);
    // (BPP == 0: planar 8-bit G, B, R planes -- gbrp, gbrap without its alpha: planar_rgb_to_y / gbr24pToUV_half_c (input.c:1174-1186, :414-432) are
    //  rgb24ToY_c / rgb24ToUV_half_c with the bytes fetched from three planes; a pixel is assembled as {R, B} / {G} halves, i.e. r, g, b at bytes 0, 1, 2)
    const int rp = BPP == 0 ? 0 : U(p.src_r_pos), gp = BPP == 4 ? U(p.src_g_pos) : 1, bp = BPP == 0 ? 2 : U(p.src_b_pos);
    auto coef = [&](const Rgb2YuvRow &w, int k) { return (uint32_t)(uint16_t)(k == rp ? w.r : k == gp ? w.g : k == bp ? w.b : 0); };
    const uint32_t cyA = coef(ty, 0) | coef(ty, 2) << 16, cyB = coef(ty, 1) | coef(ty, 3) << 16;
    const uint32_t cuA = coef(tu, 0) | coef(tu, 2) << 16, cuB = coef(tu, 1) | coef(tu, 3) << 16;
    const uint32_t cvA = coef(tv, 0) | coef(tv, 2) << 16, cvB = coef(tv, 1) | coef(tv, 3) << 16;

    const bool full = x0 + 4 <= W;                     // (the last group of a ragged width goes byte by byte)
    const int ncol = min(2, cW - 2 * t);               // chroma columns of this lane
    const int uplane = U(p.u_plane_dst), vplane = U(p.v_plane_dst);
    const uint8_t *s0 = f.src[0], *s1 = f.src[1], *s2 = f.src[2];
    const int64_t sst = f.srcStride[0], sst1 = f.srcStride[1], sst2 = f.srcStride[2];
    // (the next row's load is issued before this row's arithmetic and stores: the compiler cannot move a load above a store that may alias)
    uint32_t nx[4] = {};
    auto fetch = [&](int r) {
        const uint8_t *row = s0 + (int64_t)r * sst + (int64_t)(BPP ? BPP : 1) * x0;
        if (BPP == 0) { nx[0] = *(const uint32_t *)row; nx[1] = *(const uint32_t *)(s1 + (int64_t)r * sst1 + x0); nx[2] = *(const uint32_t *)(s2 + (int64_t)r * sst2 + x0); }
        else if (BPP == 3) { nx[0] = ((const uint32_t *)row)[0]; nx[1] = ((const uint32_t *)row)[1]; nx[2] = ((const uint32_t *)row)[2]; }
        else { const uint4 q = *(const uint4 *)row; nx[0] = q.x; nx[1] = q.y; nx[2] = q.z; nx[3] = q.w; }
    };
    if (full) fetch(rlo);
    for (int r = rlo; r <= rhi; r++) {
        // ---- the row's four pixels (and, for a ragged width, the partner of the last odd pixel: the half readers read it too) ----
        uint32_t lo[4], hi[4];     // per pixel: {byte 0, byte 2} and {byte 1, byte 3 (0 for 24 bpp)} as 16-bit halves
        const uint8_t *row = s0 + (int64_t)r * sst + (int64_t)(BPP ? BPP : 1) * x0;
        if (full) {
            const uint32_t d[4] = { nx[0], nx[1], nx[2], nx[3] };
            if (r < rhi) fetch(r + 1);
            if (BPP == 0) {   // d[0] = four G, d[1] = four B, d[2] = four R
                lo[0] = __builtin_amdgcn_perm(d[1], d[2], 0x0c040c00u); hi[0] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c00u);
                lo[1] = __builtin_amdgcn_perm(d[1], d[2], 0x0c050c01u); hi[1] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c01u);
                lo[2] = __builtin_amdgcn_perm(d[1], d[2], 0x0c060c02u); hi[2] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c02u);
                lo[3] = __builtin_amdgcn_perm(d[1], d[2], 0x0c070c03u); hi[3] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c03u);
            } else if (BPP == 3) {
                lo[0] = __builtin_amdgcn_perm(d[0], d[0], 0x0c020c00u); hi[0] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c01u);
                lo[1] = __builtin_amdgcn_perm(d[1], d[0], 0x0c050c03u); hi[1] = __builtin_amdgcn_perm(d[1], d[0], 0x0c0c0c04u);
                lo[2] = __builtin_amdgcn_perm(d[2], d[1], 0x0c040c02u); hi[2] = __builtin_amdgcn_perm(d[2], d[1], 0x0c0c0c03u);
                lo[3] = __builtin_amdgcn_perm(d[2], d[2], 0x0c030c01u); hi[3] = __builtin_amdgcn_perm(d[2], d[2], 0x0c0c0c02u);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) { lo[k] = d[k] & 0x00FF00FFu; hi[k] = (d[k] >> 8) & 0x00FF00FFu; }
            }
        } else {
            const int npx = min(4, ((W - x0) + 1) & ~1);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                lo[k] = hi[k] = 0;
                if (k < npx) {
                    if (BPP == 0) {
                        lo[k] = (s2 + (int64_t)r * sst2 + x0)[k] | (uint32_t)(s1 + (int64_t)r * sst1 + x0)[k] << 16;
                        hi[k] = row[k];
                    } else {
                        lo[k] = row[BPP * k] | (uint32_t)row[BPP * k + 2] << 16;
                        hi[k] = row[BPP * k + 1] | (BPP == 4 ? (uint32_t)row[BPP * k + 3] << 16 : 0u);
                    }
                }
            }
        }
        // ---- luma: reader -> identity hscale -> yuv2plane1_8_c ----
        if (r >= y0 && r < y1) {
            uint32_t out = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int S = sdot2(lo[k], cyA, sdot2_first_s(hi[k], cyB));
                int yr;
                if (BPP != 4) yr = (uint16_t)((S + (32 << 14) + (1 << 8)) >> 9);
                else yr = (uint16_t)((((unsigned)S << 8) + ((32u << 22) + (1u << 16))) >> 17);
                const int y15 = (int16_t)min((yr * 16384) >> hshift, hclip);
                out |= (uint32_t)clip_u8_shr(y15 + 64, 7) << (8 * k);   // (8-bit sources: no dither pattern, swscale.c:292-293)
            }
            uint8_t *drow = f.dst[0] + (int64_t)r * f.dstStride[0] + x0;
            if (full) *(uint32_t *)drow = out;
            else for (int k = 0; k < W - x0; k++) drow[k] = (uint8_t)(out >> (8 * k));
        }
        // ---- chroma: the half readers on the two pixel pairs (byte sums stay inside their 16-bit halves) -> identity hscale -> ring ----
        if (r >= cfirst && r < cSH) {
            uint32_t e2[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint32_t al = lo[2 * k] + lo[2 * k + 1], ah = hi[2 * k] + hi[2 * k + 1];
                const int Su = sdot2(al, cuA, sdot2_first_s(ah, cuB)), Sv = sdot2(al, cvA, sdot2_first_s(ah, cvB));
                int ur, vr;
                if (BPP != 4) {
                    ur = (uint16_t)((Su + (256 << 15) + (1 << 9)) >> 10);
                    vr = (uint16_t)((Sv + (256 << 15) + (1 << 9)) >> 10);
                } else {
                    const unsigned rnd = (256u << 23) + (1u << 17);
                    ur = (uint16_t)((((unsigned)Su << 8) + rnd) >> 18);
                    vr = (uint16_t)((((unsigned)Sv << 8) + rnd) >> 18);
                }
                const int u15 = (int16_t)min((ur * 16384) >> hshift, hclip), v15 = (int16_t)min((vr * 16384) >> hshift, hclip);
                e2[k] = (uint32_t)(uint16_t)u15 | (uint32_t)(uint16_t)v15 << 16;
            }
            ring[r & (RING - 1)][tid] = make_uint2(e2[0], e2[1]);
        }
        // ---- every chroma output row whose last source row this was ----
        while (cy < cy1 && clast <= r) {
            const bool x_form = NV || vfs > 1;
            int au[2], av[2];
            au[0] = au[1] = av[0] = av[1] = x_form ? 64 << 12 : 64;
#pragma unroll
            for (int jp = 0; jp < RING / 2; jp++)
                if (2 * jp < vfs) {   // (a tap past an odd filter length is zero: its slot may hold anything)
                    const uint2 q0 = ring[clampc(e.first + 2 * jp) & (RING - 1)][tid], q1 = ring[clampc(e.first + 2 * jp + 1) & (RING - 1)][tid];
                    const int w0 = (int16_t)(e.vt[jp] & 0xFFFF), w1 = (int16_t)(e.vt[jp] >> 16);
                    au[0] += (int)(unsigned)((int)(int16_t)(q0.x & 0xFFFF) * w0) + (int)(unsigned)((int)(int16_t)(q1.x & 0xFFFF) * w1);
                    av[0] += (int)(unsigned)((int)(int16_t)(q0.x >> 16) * w0) + (int)(unsigned)((int)(int16_t)(q1.x >> 16) * w1);
                    au[1] += (int)(unsigned)((int)(int16_t)(q0.y & 0xFFFF) * w0) + (int)(unsigned)((int)(int16_t)(q1.y & 0xFFFF) * w1);
                    av[1] += (int)(unsigned)((int)(int16_t)(q0.y >> 16) * w0) + (int)(unsigned)((int)(int16_t)(q1.y >> 16) * w1);
                }
            const int sh = x_form ? 19 : 7;
            const uint32_t u0 = clip_u8_shr(au[0], sh), u1 = clip_u8_shr(au[1], sh), v0 = clip_u8_shr(av[0], sh), v1 = clip_u8_shr(av[1], sh);
            if (NV) {
                uint8_t *d = f.dst[1] + (int64_t)cy * f.dstStride[1] + 4 * t;
                const int sw = U(p.uv_swap_dst);
                const uint32_t p0 = sw ? (v0 | u0 << 8) : (u0 | v0 << 8), p1 = sw ? (v1 | u1 << 8) : (u1 | v1 << 8);
                if (ncol == 2) *(uint32_t *)d = p0 | p1 << 16;
                else *(uint16_t *)d = (uint16_t)p0;
            } else {
                uint8_t *du = pick4(f.dst, uplane) + (int64_t)cy * pick4(f.dstStride, uplane) + 2 * t;
                uint8_t *dv = pick4(f.dst, vplane) + (int64_t)cy * pick4(f.dstStride, vplane) + 2 * t;
                if (ncol == 2) { *(uint16_t *)du = (uint16_t)(u0 | u1 << 8); *(uint16_t *)dv = (uint16_t)(v0 | v1 << 8); }
                else { du[0] = (uint8_t)u0; dv[0] = (uint8_t)v0; }
            }
            cy++;
            if (cy < cy1) { e = load_rgbsrc_row(rows, cy); clast = clampc(e.last); }
            else clast = 0x7fffffff;
        }
    }
}

// Reader pre-pass for SCALED packed 24 / 32 bpp RGB sources (capture -> smaller / larger planar YUV): the input stage of the scaler on
// whole rows -- rgb24ToY_c / rgb24ToUV_half_c (input.c:1068-1172), rgb16_32ToY / UV_half_c_template with the 32-bit rows (:264-393) -- written
// as the 16-bit planes hScale16To15_c reads (formatConvBuffer of the line ring, swscale.c:69-97): Y[srcH][srcW], U / V[srcH][srcW / 2].
// The marching strip kernel then takes these planes like a planar 16-bit source (k_strip.hip launch_rgbread_strip).  Lane = four pixels of a
// row (12- or 16-byte load; 8 + 4 + 4 bytes stored), a wave walks down RGBREAD_RPW rows with the next row's load in flight.
struct RgbReadLayout { uint8_t *base; int64_t frame_bytes, offU, offV, offA; int32_t strideY, strideC, a_pos; };   // a_pos < 0: no alpha plane (else rgbaToA_c / abgrToA_c, input.c:454-472)
constexpr int RGBREAD_RPW = 8;

// HALF: the "half" chroma readers (chroma planes of srcW / 2 columns: every shape whose chroma destination is at most half as wide as the source,
// utils.c:1419-1425); otherwise rgb24ToUV_c / the full-width 32-bit rows (input.c:1096-1124, :310-334): chroma planes of srcW columns
template <int BPP, bool HALF>
__global__ void __launch_bounds__(256) sws_k_rgb_read16(SwsFrameSet fs, SwsDevParams p, RgbReadLayout lay)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int W = U(p.srcW), H = U(p.srcH), x0 = 4 * t;
    if (x0 >= W) return;                               // (W even; a width of 4 k + 2 writes its last group whole: the working planes hold it)
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y0 = blockIdx.y * RGBREAD_RPW, y1 = min(H, y0 + RGBREAD_RPW);
    const Rgb2YuvRow ty = rgb2yuv_row(p.rgb2yuv, 0), tu = rgb2yuv_row(p.rgb2yuv, 3), tv = rgb2yuv_row(p.rgb2yuv, 6);
    // (BPP == 0: planar 8-bit G, B, R planes: planar_rgb_to_y / _uv / gbr24pToUV_half_c (input.c:1174-1211, :414-432) are the rgb24 readers with the
    //  bytes fetched from three planes; a pixel is assembled as {R, B} / {G} halves, i.e. r, g, b at bytes 0, 1, 2)
    const int rp = BPP == 0 ? 0 : U(p.src_r_pos), gp = BPP == 4 ? U(p.src_g_pos) : 1, bp = BPP == 0 ? 2 : U(p.src_b_pos);
    auto coef = [&](const Rgb2YuvRow &w, int k) { return (uint32_t)(uint16_t)(k == rp ? w.r : k == gp ? w.g : k == bp ? w.b : 0); };
    const uint32_t cyA = coef(ty, 0) | coef(ty, 2) << 16, cyB = coef(ty, 1) | coef(ty, 3) << 16;
    const uint32_t cuA = coef(tu, 0) | coef(tu, 2) << 16, cuB = coef(tu, 1) | coef(tu, 3) << 16;
    const uint32_t cvA = coef(tv, 0) | coef(tv, 2) << 16, cvB = coef(tv, 1) | coef(tv, 3) << 16;
    const uint8_t *s0 = f.src[0], *s1 = f.src[1], *s2 = f.src[2];
    const int64_t sst = f.srcStride[0], sst1 = f.srcStride[1], sst2 = f.srcStride[2];
    uint8_t *fb = U(lay.base) + (int64_t)blockIdx.z * U(lay.frame_bytes);
    uint8_t *dY = fb + 2 * (int64_t)x0, *dU = fb + U(lay.offU) + (int64_t)x0 * (HALF ? 1 : 2), *dV = fb + U(lay.offV) + (int64_t)x0 * (HALF ? 1 : 2);
    const int64_t dsY = U(lay.strideY), dsC = U(lay.strideC);
    uint32_t nx[4] = {};
    const bool tail = x0 + 4 > W;                      // (a width of 4 k + 2: the last group holds two pixels -- nothing is read behind the row's last pixel)
    auto fetch = [&](int r) {
        const uint8_t *row = s0 + (int64_t)r * sst + (int64_t)(BPP ? BPP : 1) * x0;
        if (tail) {
            if (BPP == 0) { nx[0] = *(const uint16_t *)row; nx[1] = *(const uint16_t *)(s1 + (int64_t)r * sst1 + x0); nx[2] = *(const uint16_t *)(s2 + (int64_t)r * sst2 + x0); }
            else if (BPP == 3) { nx[0] = ((const uint32_t *)row)[0]; nx[1] = ((const uint16_t *)row)[2]; nx[2] = 0; }
            else { nx[0] = ((const uint32_t *)row)[0]; nx[1] = ((const uint32_t *)row)[1]; nx[2] = nx[3] = 0; }
            return;
        }
        if (BPP == 0) { nx[0] = *(const uint32_t *)row; nx[1] = *(const uint32_t *)(s1 + (int64_t)r * sst1 + x0); nx[2] = *(const uint32_t *)(s2 + (int64_t)r * sst2 + x0); }
        else if (BPP == 3) { nx[0] = ((const uint32_t *)row)[0]; nx[1] = ((const uint32_t *)row)[1]; nx[2] = ((const uint32_t *)row)[2]; }
        else { const uint4 q = *(const uint4 *)row; nx[0] = q.x; nx[1] = q.y; nx[2] = q.z; nx[3] = q.w; }
    };
    if (y0 < y1) fetch(y0);
    for (int r = y0; r < y1; r++) {
        uint32_t lo[4], hi[4];     // per pixel: {byte 0, byte 2} and {byte 1, byte 3 (0 for 24 bpp)} as 16-bit halves
        const uint32_t d[4] = { nx[0], nx[1], nx[2], nx[3] };
        if (r + 1 < y1) fetch(r + 1);
        if (BPP == 0) {   // d[0] = four G, d[1] = four B, d[2] = four R
            lo[0] = __builtin_amdgcn_perm(d[1], d[2], 0x0c040c00u); hi[0] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c00u);
            lo[1] = __builtin_amdgcn_perm(d[1], d[2], 0x0c050c01u); hi[1] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c01u);
            lo[2] = __builtin_amdgcn_perm(d[1], d[2], 0x0c060c02u); hi[2] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c02u);
            lo[3] = __builtin_amdgcn_perm(d[1], d[2], 0x0c070c03u); hi[3] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c03u);
        } else if (BPP == 3) {
            lo[0] = __builtin_amdgcn_perm(d[0], d[0], 0x0c020c00u); hi[0] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c01u);
            lo[1] = __builtin_amdgcn_perm(d[1], d[0], 0x0c050c03u); hi[1] = __builtin_amdgcn_perm(d[1], d[0], 0x0c0c0c04u);
            lo[2] = __builtin_amdgcn_perm(d[2], d[1], 0x0c040c02u); hi[2] = __builtin_amdgcn_perm(d[2], d[1], 0x0c0c0c03u);
            lo[3] = __builtin_amdgcn_perm(d[2], d[2], 0x0c030c01u); hi[3] = __builtin_amdgcn_perm(d[2], d[2], 0x0c0c0c02u);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) { lo[k] = d[k] & 0x00FF00FFu; hi[k] = (d[k] >> 8) & 0x00FF00FFu; }
        }
        uint32_t yv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int S = sdot2(lo[k], cyA, sdot2_first_s(hi[k], cyB));
            if (BPP != 4) yv[k] = (uint16_t)((S + (32 << 14) + (1 << 8)) >> 9);
            else yv[k] = (uint16_t)((((unsigned)S << 8) + ((32u << 22) + (1u << 16))) >> 17);
        }
        const u32x2 oy = { yv[0] | yv[1] << 16, yv[2] | yv[3] << 16 };
        *(u32x2 *)(dY + (int64_t)r * dsY) = oy;
        if (BPP == 4 && U(lay.a_pos) >= 0) {     // the alpha bytes as the 14-bit samples the luma scaler reads: a << 6 | a >> 2
            const int sh8 = 8 * (U(lay.a_pos) & 3);
            const uint32_t opaque = (U(lay.a_pos) & 8) ? 0xFFu : 0u;     // (an rgb0-style source feeding a real alpha channel: its X byte counts as 255, swscale.c:1106-1124)
            uint32_t av[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint32_t a = ((d[k] >> sh8) & 0xFFu) | opaque; av[k] = (a << 6) | (a >> 2); }
            const u32x2 oa = { av[0] | av[1] << 16, av[2] | av[3] << 16 };
            *(u32x2 *)(fb + U(lay.offA) + 2 * (int64_t)x0 + (int64_t)r * dsY) = oa;
        }
        if constexpr (HALF) {
            uint32_t uv[2], vv[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint32_t al = lo[2 * k] + lo[2 * k + 1], ah = hi[2 * k] + hi[2 * k + 1];
                const int Su = sdot2(al, cuA, sdot2_first_s(ah, cuB)), Sv = sdot2(al, cvA, sdot2_first_s(ah, cvB));
                if (BPP != 4) {
                    uv[k] = (uint16_t)((Su + (256 << 15) + (1 << 9)) >> 10);
                    vv[k] = (uint16_t)((Sv + (256 << 15) + (1 << 9)) >> 10);
                } else {
                    const unsigned rnd = (256u << 23) + (1u << 17);
                    uv[k] = (uint16_t)((((unsigned)Su << 8) + rnd) >> 18);
                    vv[k] = (uint16_t)((((unsigned)Sv << 8) + rnd) >> 18);
                }
            }
            *(uint32_t *)(dU + (int64_t)r * dsC) = uv[0] | uv[1] << 16;
            *(uint32_t *)(dV + (int64_t)r * dsC) = vv[0] | vv[1] << 16;
        } else {
            uint32_t uv[4], vv[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int Su = sdot2(lo[k], cuA, sdot2_first_s(hi[k], cuB)), Sv = sdot2(lo[k], cvA, sdot2_first_s(hi[k], cvB));
                if (BPP != 4) {
                    uv[k] = (uint16_t)((Su + (256 << 14) + (1 << 8)) >> 9);
                    vv[k] = (uint16_t)((Sv + (256 << 14) + (1 << 8)) >> 9);
                } else {
                    uv[k] = (uint16_t)((((unsigned)Su << 8) + ((256u << 22) + (1u << 16))) >> 17);
                    vv[k] = (uint16_t)((((unsigned)Sv << 8) + ((256u << 22) + (1u << 16))) >> 17);
                }
            }
            const u32x2 ou = { uv[0] | uv[1] << 16, uv[2] | uv[3] << 16 }, ov = { vv[0] | vv[1] << 16, vv[2] | vv[3] << 16 };
            *(u32x2 *)(dU + (int64_t)r * dsC) = ou;
            *(u32x2 *)(dV + (int64_t)r * dsC) = ov;
        }
    }
}

// Packed 24 / 32 bpp RGB and planar 8-bit GBR into planar 8-bit 4:4:4 YUV of the same size: every filter is the identity, every pixel
// independent.  rgb24ToY_c / rgb24ToUV_c (input.c:1068-1124), rgb16_32ToY / UV_c_template with the 32-bit rows (:264-334), planar_rgb_to_y /
// _uv (:1174-1211), hScale16To15_c with one tap, yuv2plane1_8_c (8-bit source: the constant 64, swscale.c:385-387).  Lane = four pixels, a
// wave walks down a band of rows with the next row's load in flight; three dword stores per row.
template <int BPP>
__global__ void __launch_bounds__(256) sws_k_rgb_yuv444_unity(SwsFrameSet fs, SwsDevParams p, int band_rows)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int W = U(p.dstW), H = U(p.dstH), x0 = 4 * t;
    if (x0 >= W) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y0 = blockIdx.y * U(band_rows), y1 = min(H, y0 + U(band_rows));
    const int hshift = U(p.hshift), hclip = U(p.hclip);
    const Rgb2YuvRow ty = rgb2yuv_row(p.rgb2yuv, 0), tu = rgb2yuv_row(p.rgb2yuv, 3), tv = rgb2yuv_row(p.rgb2yuv, 6);
    const int rp = BPP == 0 ? 0 : U(p.src_r_pos), gp = BPP == 4 ? U(p.src_g_pos) : 1, bp = BPP == 0 ? 2 : U(p.src_b_pos);
    auto coef = [&](const Rgb2YuvRow &w, int k) { return (uint32_t)(uint16_t)(k == rp ? w.r : k == gp ? w.g : k == bp ? w.b : 0); };
    const uint32_t cyA = coef(ty, 0) | coef(ty, 2) << 16, cyB = coef(ty, 1) | coef(ty, 3) << 16;
    const uint32_t cuA = coef(tu, 0) | coef(tu, 2) << 16, cuB = coef(tu, 1) | coef(tu, 3) << 16;
    const uint32_t cvA = coef(tv, 0) | coef(tv, 2) << 16, cvB = coef(tv, 1) | coef(tv, 3) << 16;
    const bool full = x0 + 4 <= W;
    const int uplane = U(p.u_plane_dst), vplane = U(p.v_plane_dst);
    const uint8_t *s0 = f.src[0], *s1 = f.src[1], *s2 = f.src[2];
    const int64_t sst = f.srcStride[0], sst1 = f.srcStride[1], sst2 = f.srcStride[2];
    uint32_t nx[4] = {};
    auto fetch = [&](int r) {
        const uint8_t *row = s0 + (int64_t)r * sst + (int64_t)(BPP ? BPP : 1) * x0;
        if (BPP == 0) { nx[0] = *(const uint32_t *)row; nx[1] = *(const uint32_t *)(s1 + (int64_t)r * sst1 + x0); nx[2] = *(const uint32_t *)(s2 + (int64_t)r * sst2 + x0); }
        else if (BPP == 3) { nx[0] = ((const uint32_t *)row)[0]; nx[1] = ((const uint32_t *)row)[1]; nx[2] = ((const uint32_t *)row)[2]; }
        else { const uint4 q = *(const uint4 *)row; nx[0] = q.x; nx[1] = q.y; nx[2] = q.z; nx[3] = q.w; }
    };
    if (full && y0 < y1) fetch(y0);
    for (int r = y0; r < y1; r++) {
        uint32_t lo[4], hi[4];     // per pixel: {byte 0, byte 2} and {byte 1, byte 3 (0 for 24 bpp)} as 16-bit halves
        const uint8_t *row = s0 + (int64_t)r * sst + (int64_t)(BPP ? BPP : 1) * x0;
        if (full) {
            const uint32_t d[4] = { nx[0], nx[1], nx[2], nx[3] };
            if (r + 1 < y1) fetch(r + 1);
            if (BPP == 0) {
                lo[0] = __builtin_amdgcn_perm(d[1], d[2], 0x0c040c00u); hi[0] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c00u);
                lo[1] = __builtin_amdgcn_perm(d[1], d[2], 0x0c050c01u); hi[1] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c01u);
                lo[2] = __builtin_amdgcn_perm(d[1], d[2], 0x0c060c02u); hi[2] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c02u);
                lo[3] = __builtin_amdgcn_perm(d[1], d[2], 0x0c070c03u); hi[3] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c03u);
            } else if (BPP == 3) {
                lo[0] = __builtin_amdgcn_perm(d[0], d[0], 0x0c020c00u); hi[0] = __builtin_amdgcn_perm(d[0], d[0], 0x0c0c0c01u);
                lo[1] = __builtin_amdgcn_perm(d[1], d[0], 0x0c050c03u); hi[1] = __builtin_amdgcn_perm(d[1], d[0], 0x0c0c0c04u);
                lo[2] = __builtin_amdgcn_perm(d[2], d[1], 0x0c040c02u); hi[2] = __builtin_amdgcn_perm(d[2], d[1], 0x0c0c0c03u);
                lo[3] = __builtin_amdgcn_perm(d[2], d[2], 0x0c030c01u); hi[3] = __builtin_amdgcn_perm(d[2], d[2], 0x0c0c0c02u);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) { lo[k] = d[k] & 0x00FF00FFu; hi[k] = (d[k] >> 8) & 0x00FF00FFu; }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                lo[k] = hi[k] = 0;
                if (k < W - x0) {
                    if (BPP == 0) { lo[k] = (s2 + (int64_t)r * sst2 + x0)[k] | (uint32_t)(s1 + (int64_t)r * sst1 + x0)[k] << 16; hi[k] = row[k]; }
                    else { lo[k] = row[BPP * k] | (uint32_t)row[BPP * k + 2] << 16; hi[k] = row[BPP * k + 1] | (BPP == 4 ? (uint32_t)row[BPP * k + 3] << 16 : 0u); }
                }
            }
        }
        uint32_t oy = 0, ou = 0, ov = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int Sy = sdot2(lo[k], cyA, sdot2_first_s(hi[k], cyB)), Su = sdot2(lo[k], cuA, sdot2_first_s(hi[k], cuB)), Sv = sdot2(lo[k], cvA, sdot2_first_s(hi[k], cvB));
            int yr, ur, vr;
            if (BPP != 4) {
                yr = (uint16_t)((Sy + (32 << 14) + (1 << 8)) >> 9);
                ur = (uint16_t)((Su + (256 << 14) + (1 << 8)) >> 9);
                vr = (uint16_t)((Sv + (256 << 14) + (1 << 8)) >> 9);
            } else {
                yr = (uint16_t)((((unsigned)Sy << 8) + ((32u << 22) + (1u << 16))) >> 17);
                ur = (uint16_t)((((unsigned)Su << 8) + ((256u << 22) + (1u << 16))) >> 17);
                vr = (uint16_t)((((unsigned)Sv << 8) + ((256u << 22) + (1u << 16))) >> 17);
            }
            const int y15 = (int16_t)min((yr * 16384) >> hshift, hclip), u15 = (int16_t)min((ur * 16384) >> hshift, hclip), v15 = (int16_t)min((vr * 16384) >> hshift, hclip);
            oy |= (uint32_t)clip_u8_shr(y15 + 64, 7) << (8 * k);
            ou |= (uint32_t)clip_u8_shr(u15 + 64, 7) << (8 * k);
            ov |= (uint32_t)clip_u8_shr(v15 + 64, 7) << (8 * k);
        }
        uint8_t *dy = f.dst[0] + (int64_t)r * f.dstStride[0] + x0;
        uint8_t *du = pick4(f.dst, uplane) + (int64_t)r * pick4(f.dstStride, uplane) + x0, *dv = pick4(f.dst, vplane) + (int64_t)r * pick4(f.dstStride, vplane) + x0;
        if (full) { *(uint32_t *)dy = oy; *(uint32_t *)du = ou; *(uint32_t *)dv = ov; }
        else for (int k = 0; k < W - x0; k++) { dy[k] = (uint8_t)(oy >> (8 * k)); du[k] = (uint8_t)(ou >> (8 * k)); dv[k] = (uint8_t)(ov >> (8 * k)); }
    }
}

} // namespace swsk
