// Specialised, vectorised kernels for the shapes the BASELINE configs take:
//   sws_k_yuv2rgb_unscaled   C2a  yuv420p/yuv422p -> rgb24/bgr24/rgb32 family, LUT converter
//   sws_k_rgb_fused_unity    C2b/C4  8-bit planar or nv12 -> packed RGB through the polyphase chain with
//                            identity horizontal filters (N-tap vertical chroma)
//   sws_k_p01x_unscaled      C3a  yuv420p10le/16le -> p010le shift + interleave
//   sws_k_planar_misc        other unscaled planar converters (scalar per element)
// Every lane handles 8 horizontally adjacent pixels so that a wave64 load/store instruction moves
// 512 B .. 2 KiB of contiguous memory.  All arithmetic is the generic kernels' arithmetic.
#pragma once
#include "kernels_generic.hpp"

namespace swsk {

// final per-pixel LUT stage for 8 pixels (4 chroma pairs); writes 24 or 32 bytes at d
template <int BPP, bool VEC>
__device__ __forceinline__ void emit8(const SwsLutParams &L, uint8_t *d, const int (&Y)[8], const int (&U)[4], const int (&V)[4])
{
    if constexpr (BPP == 4) {
        uint32_t px[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const ChromaIdx c = lut_chroma(L, U[k], V[k]);
            px[2 * k] = lut_rgb32(L, c, Y[2 * k]);
            px[2 * k + 1] = lut_rgb32(L, c, Y[2 * k + 1]);
        }
        if constexpr (VEC) {
            uint4 *o = (uint4 *)d;
            o[0] = make_uint4(px[0], px[1], px[2], px[3]);
            o[1] = make_uint4(px[4], px[5], px[6], px[7]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) { d[4 * k] = px[k]; d[4 * k + 1] = px[k] >> 8; d[4 * k + 2] = px[k] >> 16; d[4 * k + 3] = px[k] >> 24; }
        }
    } else {
        // 24 channel values kept as ints and packed with shifts (a byte array here made hipcc 7.2 emit a
        // wrong SDWA byte merge for some third bytes: caught by the parity tests)
        int v[24];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const ChromaIdx c = lut_chroma(L, U[k], V[k]);
            const int k0 = L.rgb_order ? c.b : c.r, k2 = L.rgb_order ? c.r : c.b;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int y = Y[2 * k + h];
                v[6 * k + 3 * h + 0] = lut_luma(L, k0 + y);
                v[6 * k + 3 * h + 1] = lut_luma(L, c.g + y);
                v[6 * k + 3 * h + 2] = lut_luma(L, k2 + y);
            }
        }
        if constexpr (VEC) {
            uint32_t w[6];
#pragma unroll
            for (int k = 0; k < 6; k++)
                w[k] = (uint32_t)v[4 * k] | ((uint32_t)v[4 * k + 1] << 8) | ((uint32_t)v[4 * k + 2] << 16) | ((uint32_t)v[4 * k + 3] << 24);
            uint2 *o = (uint2 *)d;
            o[0] = make_uint2(w[0], w[1]); o[1] = make_uint2(w[2], w[3]); o[2] = make_uint2(w[4], w[5]);
        } else {
#pragma unroll
            for (int k = 0; k < 24; k++) d[k] = (uint8_t)v[k];
        }
    }
}

// one pixel pair, scalar (row tails / unaligned images)
template <int BPP>
__device__ __forceinline__ void emit_pair(const SwsLutParams &L, uint8_t *drow, int i, int Y1, int Y2, int U, int V)
{
    const ChromaIdx c = lut_chroma(L, U, V);
    if constexpr (BPP == 4) {
        const uint32_t a = lut_rgb32(L, c, Y1), b = lut_rgb32(L, c, Y2);
        uint8_t *d = drow + 8 * i;
        d[0] = a; d[1] = a >> 8; d[2] = a >> 16; d[3] = a >> 24;
        d[4] = b; d[5] = b >> 8; d[6] = b >> 16; d[7] = b >> 24;
    } else {
        uint8_t *d = drow + 6 * i;
        const int k0 = L.rgb_order ? c.b : c.r, k2 = L.rgb_order ? c.r : c.b;
        d[0] = (uint8_t)lut_luma(L, k0 + Y1); d[1] = (uint8_t)lut_luma(L, c.g + Y1); d[2] = (uint8_t)lut_luma(L, k2 + Y1);
        d[3] = (uint8_t)lut_luma(L, k0 + Y2); d[4] = (uint8_t)lut_luma(L, c.g + Y2); d[5] = (uint8_t)lut_luma(L, k2 + Y2);
    }
}

template <bool VEC> __device__ __forceinline__ void load8(const uint8_t *s, int (&o)[8])
{
    if constexpr (VEC) {
        const uint2 v = *(const uint2 *)s;
#pragma unroll
        for (int k = 0; k < 4; k++) { o[k] = (v.x >> (8 * k)) & 0xFF; o[4 + k] = (v.y >> (8 * k)) & 0xFF; }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) o[k] = s[k];
    }
}
template <bool VEC> __device__ __forceinline__ void load4(const uint8_t *s, int (&o)[4])
{
    if constexpr (VEC) {
        const uint32_t v = *(const uint32_t *)s;
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = (v >> (8 * k)) & 0xFF;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = s[k];
    }
}

// ------------------------------------------------------------------------------------------
// C2a: unscaled yuv420p / yuv422p -> packed RGB (YUV420FUNC/YUV422FUNC + PUTRGB24/PUTBGR24/PUTRGB,
// libswscale/yuv2rgb.c:68-559).  Chroma sample i feeds pixels 2i, 2i+1 of both rows of the row pair
// (4:2:0) or of its own row (4:2:2); npairs = pairs the reference's 8/4/2-pixel blocks cover.
// Work item = 8 pixels x 2 rows.  grid.x over items of one frame, grid.z = frame.
// ------------------------------------------------------------------------------------------
template <int BPP, bool VEC>
__global__ void __launch_bounds__(256) sws_k_yuv2rgb_unscaled(SwsFrameSet fs, SwsDevParams p, int is422, int npairs,
                                                              int y0, int nrowpairs)
{
    const int blocks_per_row = (npairs + 3) >> 2;           // 8-pixel blocks per row
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= (int64_t)blocks_per_row * nrowpairs) return;
    const int rp = (int)(item / blocks_per_row), cb = (int)(item % blocks_per_row);
    const SwsFramePtrs &f = frame_of(fs, blockIdx.z);
    const SwsLutParams &L = p.lut;
    const int yrow = y0 + 2 * rp;                            // first row of the pair (absolute image row)
    const int crow0 = is422 ? yrow : (yrow >> 1);
    const uint8_t *py = f.src[0] + (int64_t)yrow * f.srcStride[0] + 8 * cb;
    const uint8_t *pu = f.src[1] + (int64_t)crow0 * f.srcStride[1] + 4 * cb;
    const uint8_t *pv = f.src[2] + (int64_t)crow0 * f.srcStride[2] + 4 * cb;
    uint8_t *d0 = f.dst[0] + (int64_t)yrow * f.dstStride[0];
    if (4 * cb + 4 <= npairs) {
        int Y[8], U[4], V[4];
        load4<VEC>(pu, U); load4<VEC>(pv, V);
        load8<VEC>(py, Y);
        emit8<BPP, VEC>(L, d0 + (int64_t)8 * BPP * cb, Y, U, V);
        if (is422) { load4<VEC>(pu + f.srcStride[1], U); load4<VEC>(pv + f.srcStride[2], V); }
        load8<VEC>(py + f.srcStride[0], Y);
        emit8<BPP, VEC>(L, d0 + f.dstStride[0] + (int64_t)8 * BPP * cb, Y, U, V);
    } else {
        for (int i = 4 * cb; i < npairs; i++)
            for (int l = 0; l < 2; l++) {
                const uint8_t *yy = f.src[0] + (int64_t)(yrow + l) * f.srcStride[0];
                const int cr = is422 ? yrow + l : (yrow >> 1);
                const int U1 = f.src[1][(int64_t)cr * f.srcStride[1] + i], V1 = f.src[2][(int64_t)cr * f.srcStride[2] + i];
                emit_pair<BPP>(L, d0 + (int64_t)l * f.dstStride[0], i, yy[2 * i], yy[2 * i + 1], U1, V1);
            }
    }
}

// ------------------------------------------------------------------------------------------
// C2b / C4: polyphase chain with identity horizontal filters, 8-bit planar (SRCK_PLANAR8) or
// semi-planar (SRCK_NV12) source, packed RGB LUT writer (chroma shared by pixel pairs).
// = hScale8To15_c with 1 tap (src << 7) + packed_vscale + yuv2rgb_{1,2,X}_c_template.
// Work item = 8 pixels of one output row.  grid.x over items of one frame, grid.z = frame.
// ------------------------------------------------------------------------------------------
template <int BPP, bool NV, bool VEC>
__global__ void __launch_bounds__(256) sws_k_rgb_fused_unity(SwsFrameSet fs, SwsDevParams p)
{
    const int npairs = (p.dstW + 1) >> 1;
    const int blocks_per_row = (npairs + 3) >> 2;
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= (int64_t)blocks_per_row * p.dstH) return;
    const int y = (int)(item / blocks_per_row), cb = (int)(item % blocks_per_row);
    const SwsFramePtrs &f = frame_of(fs, blockIdx.z);
    const SwsLutParams &L = p.lut;
    const int cy = y >> p.chrDstVSub;
    const int lfs = p.vLumFs, cfs = p.vChrFs;
    const int16_t *lf = p.vLumF + y * lfs, *cf = p.vChrF + cy * cfs;
    const int firstL = max(1 - lfs, p.vLumPos[y]), firstC = max(1 - cfs, p.vChrPos[cy]);
    const int lH = p.srcH - 1, cH = p.chrSrcH - 1;
    uint8_t *drow = f.dst[0] + (int64_t)y * f.dstStride[0];

    int mode = 0, ua = 0, ya = 0;
    if (lfs == 1 && cfs == 1) mode = 1;
    else if (lfs == 1 && cfs == 2 && (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) { mode = 1; ua = (uint16_t)cf[1]; }
    else if (lfs == 2 && cfs == 2 && (uint16_t)lf[1] + (uint16_t)lf[0] == 4096 && (uint16_t)lf[1] <= 4096U &&
             (uint16_t)cf[1] + (uint16_t)cf[0] == 4096 && (uint16_t)cf[1] <= 4096U) { mode = 2; ya = (uint16_t)lf[1]; ua = (uint16_t)cf[1]; }

    const bool full_block = 4 * cb + 4 <= npairs && 8 * cb + 8 <= p.dstW;
    const int up = p.u_plane_src, vp = p.v_plane_src;
    auto lum_row = [&](int j) { return f.src[0] + (int64_t)min(firstL + j, lH) * f.srcStride[0]; };
    // chroma loaders: 4 U and 4 V samples at chroma columns 4cb..4cb+3 of chroma row r
    auto chr4 = [&](int r, int (&u)[4], int (&v)[4]) {
        if constexpr (NV) {
            const uint8_t *s = f.src[1] + (int64_t)r * f.srcStride[1] + 8 * cb;
            int t[8];
            load8<VEC>(s, t);
#pragma unroll
            for (int k = 0; k < 4; k++) { u[k] = t[2 * k + p.uv_swap_src]; v[k] = t[2 * k + 1 - p.uv_swap_src]; }
        } else {
            load4<VEC>(f.src[up] + (int64_t)r * f.srcStride[up] + 4 * cb, u);
            load4<VEC>(f.src[vp] + (int64_t)r * f.srcStride[vp] + 4 * cb, v);
        }
    };
    if (full_block) {
        int Y[8], U[4], V[4], t8[8], tu[4], tv[4];
        if (mode == 0) {
#pragma unroll
            for (int k = 0; k < 8; k++) Y[k] = 1 << 18;
#pragma unroll
            for (int k = 0; k < 4; k++) U[k] = V[k] = 1 << 18;
            for (int j = 0; j < lfs; j++) {
                load8<VEC>(lum_row(j) + 8 * cb, t8);
                const unsigned w = (unsigned)(int)lf[j];
#pragma unroll
                for (int k = 0; k < 8; k++) Y[k] += (int)((unsigned)(t8[k] << 7) * w);
            }
            for (int j = 0; j < cfs; j++) {
                chr4(min(firstC + j, cH), tu, tv);
                const unsigned w = (unsigned)(int)cf[j];
#pragma unroll
                for (int k = 0; k < 4; k++) { U[k] += (int)((unsigned)(tu[k] << 7) * w); V[k] += (int)((unsigned)(tv[k] << 7) * w); }
            }
#pragma unroll
            for (int k = 0; k < 8; k++) Y[k] >>= 19;
#pragma unroll
            for (int k = 0; k < 4; k++) { U[k] >>= 19; V[k] >>= 19; }
        } else if (mode == 2) {
            int a8[8], au[4], av[4];
            const int ya1 = 4096 - ya, ua1 = 4096 - ua;
            load8<VEC>(lum_row(0) + 8 * cb, t8); load8<VEC>(lum_row(1) + 8 * cb, a8);
#pragma unroll
            for (int k = 0; k < 8; k++) Y[k] = ((t8[k] << 7) * ya1 + (a8[k] << 7) * ya) >> 19;
            chr4(min(firstC, cH), tu, tv); chr4(min(firstC + 1, cH), au, av);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                U[k] = ((tu[k] << 7) * ua1 + (au[k] << 7) * ua) >> 19;
                V[k] = ((tv[k] << 7) * ua1 + (av[k] << 7) * ua) >> 19;
            }
        } else {
            load8<VEC>(lum_row(0) + 8 * cb, t8);
#pragma unroll
            for (int k = 0; k < 8; k++) Y[k] = ((t8[k] << 7) + 64) >> 7;
            chr4(min(firstC, cH), tu, tv);
            if (ua == 0) {
#pragma unroll
                for (int k = 0; k < 4; k++) { U[k] = ((tu[k] << 7) + 64) >> 7; V[k] = ((tv[k] << 7) + 64) >> 7; }
            } else {
                int au[4], av[4];
                const int ua1 = 4096 - ua;
                chr4(min(firstC + 1, cH), au, av);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    U[k] = ((tu[k] << 7) * ua1 + (au[k] << 7) * ua + (128 << 11)) >> 19;
                    V[k] = ((tv[k] << 7) * ua1 + (av[k] << 7) * ua + (128 << 11)) >> 19;
                }
            }
        }
        emit8<BPP, VEC>(L, drow + (int64_t)8 * BPP * cb, Y, U, V);
    } else {
        // row tail: scalar pairs through the generic routine (identical arithmetic)
        const DirectSampler<SwsDevParams> smp{&p, &f};
        for (int i = 4 * cb; i < npairs; i++) rgb_write_unit(p, smp, f, i, y);
    }
}

// ------------------------------------------------------------------------------------------
// C3a: planarToP01xWrapper (swscale_unscaled.c:273-322): dst = src << shift per component, U/V
// interleaved.  Work item = 8 luma samples of a luma row, or 4 U + 4 V samples of a chroma row.
// grid.y: [0, H) luma rows then [H, H + chrH) chroma rows.  SRC8 = planar8ToP01xleWrapper (:333-372).
// ------------------------------------------------------------------------------------------
template <bool SRC8, bool VEC>
__global__ void __launch_bounds__(256) sws_k_p01x_unscaled(SwsFrameSet fs, SwsDevParams p, int y0, int nrows)
{
    const SwsFramePtrs &f = frame_of(fs, blockIdx.z);
    const int x8 = blockIdx.x * blockDim.x + threadIdx.x;
    const int chr_rows = (nrows + 1) >> 1;  // rows y with !(y & 1), y relative to the slice (y0 is even)
    const int r = blockIdx.y;
    if (r < nrows) {
        const int y = y0 + r;
        const int x = 8 * x8;
        if (x >= p.srcW) return;
        uint16_t *d = (uint16_t *)(f.dst[0] + (int64_t)y * f.dstStride[0]) + x;
        const int n = min(8, p.srcW - x);
        if constexpr (SRC8) {
            const uint8_t *s = f.src[0] + (int64_t)y * f.srcStride[0] + x;
            if (VEC && n == 8) {
                int t[8]; load8<true>(s, t);
                uint4 o = make_uint4((t[0] << 8) | (t[1] << 24), (t[2] << 8) | (t[3] << 24), (t[4] << 8) | (t[5] << 24), (t[6] << 8) | (t[7] << 24));
                *(uint4 *)d = o;
            } else for (int k = 0; k < n; k++) d[k] = (uint16_t)(s[k] << 8);
        } else {
            const uint16_t *s = (const uint16_t *)(f.src[0] + (int64_t)y * f.srcStride[0]) + x;
            const int sh = p.shiftY;
            if (VEC && n == 8) {
                const uint4 v = *(const uint4 *)s;
                auto sh2 = [sh](uint32_t w) { return (uint32_t)(uint16_t)((w & 0xFFFF) << sh) | ((uint32_t)(uint16_t)((w >> 16) << sh) << 16); };
                *(uint4 *)d = make_uint4(sh2(v.x), sh2(v.y), sh2(v.z), sh2(v.w));
            } else for (int k = 0; k < n; k++) d[k] = (uint16_t)(s[k] << sh);
        }
    } else if (r < nrows + chr_rows) {
        const int cr = (y0 >> 1) + (r - nrows);
        const int cw = p.srcW / 2;   // the reference converts srcW/2 chroma samples (:311)
        const int x = 4 * x8;
        if (x >= cw) return;
        const int n = min(4, cw - x);
        uint16_t *d = (uint16_t *)(f.dst[1] + (int64_t)cr * f.dstStride[1]) + 2 * x;
        if constexpr (SRC8) {
            const uint8_t *su = f.src[1] + (int64_t)cr * f.srcStride[1] + x, *sv = f.src[2] + (int64_t)cr * f.srcStride[2] + x;
            if (VEC && n == 4) {
                int u[4], v[4]; load4<true>(su, u); load4<true>(sv, v);
                *(uint4 *)d = make_uint4((u[0] << 8) | (v[0] << 24), (u[1] << 8) | (v[1] << 24), (u[2] << 8) | (v[2] << 24), (u[3] << 8) | (v[3] << 24));
            } else for (int k = 0; k < n; k++) { d[2 * k] = (uint16_t)(su[k] << 8); d[2 * k + 1] = (uint16_t)(sv[k] << 8); }
        } else {
            const uint16_t *su = (const uint16_t *)(f.src[1] + (int64_t)cr * f.srcStride[1]) + x;
            const uint16_t *sv = (const uint16_t *)(f.src[2] + (int64_t)cr * f.srcStride[2]) + x;
            const int su_sh = p.shiftU, sv_sh = p.shiftV;
            if (VEC && n == 4) {
                const uint2 u = *(const uint2 *)su, v = *(const uint2 *)sv;
                auto il = [&](uint32_t a, uint32_t b) { return (uint32_t)(uint16_t)(a << su_sh) | ((uint32_t)(uint16_t)(b << sv_sh) << 16); };
                *(uint4 *)d = make_uint4(il(u.x & 0xFFFF, v.x & 0xFFFF), il(u.x >> 16, v.x >> 16), il(u.y & 0xFFFF, v.y & 0xFFFF), il(u.y >> 16, v.y >> 16));
            } else for (int k = 0; k < n; k++) { d[2 * k] = (uint16_t)(su[k] << su_sh); d[2 * k + 1] = (uint16_t)(sv[k] << sv_sh); }
        }
    }
}

} // namespace swsk
