// Wave-tiled kernels for the packed-RGB writers (the north-star shapes C2a / C2b / C4).
//
// One wavefront = one output-row tile of 64 lanes x 16 pixels = 1024 pixels.  Each lane converts 16
// horizontally adjacent pixels (one 16-byte luma load per row, 8-byte chroma loads); the wave then
// transposes its 3 KiB (rgb24) / 4 KiB (rgb32) of output through LDS so that every global store
// instruction writes one contiguous, 16-byte-per-lane, 1 KiB run of the destination row.
// Plane pointers / strides live in SGPRs (FrameRegs), loads and stores use the global address space,
// the output is written non-temporally.  Arithmetic is the generic kernels' arithmetic.
#pragma once
#include "wave_util.hpp"

namespace swsk {

// EXP template arguments select profiling experiments (results are WRONG): they exist only in -DSWS_HIP_PROFILING builds
#ifdef SWS_HIP_PROFILING
#define SWS_EXP(n) (EXP == (n))
#else
#define SWS_EXP(n) false
#endif

// chroma part of the LUT for the 8 pixel pairs of a lane
struct Chroma8 { int r[8], g[8], b[8]; };
template <bool SWAP_RB>
__device__ __forceinline__ void chroma8(const SwsLutParams &L, const int (&U)[8], const int (&V)[8], Chroma8 &c)
{
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const ChromaIdx q = lut_chroma(L, U[k], V[k]);
        c.r[k] = SWAP_RB ? q.b : q.r; c.g[k] = q.g; c.b[k] = SWAP_RB ? q.r : q.b;  // "r" = first byte of the pixel
    }
}

// LUT stage for 16 pixels (8 chroma pairs) -> NDW = 4*BPP packed dwords in w[].
// channel value = clip_u8((yb0r + (idx + Y) * cy) >> 16)   (lut_luma); the clamp/shift/pack is pack4_u8_shr16.
template <int BPP>
__device__ __forceinline__ void lut16(const SwsLutParams &L, const int (&Y)[16], const Chroma8 &c, uint32_t (&w)[4 * BPP])
{
    if constexpr (BPP == 4) {
        const int ta = 255 << 16;
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int y = Y[2 * k + h];
                // canonical byte order first,g,third,alpha; L.perm32 moves the bytes to the format's positions
                const uint32_t px = pack4_u8_shr16(mad24(c.r[k] + y, L.cy, L.yb0r), mad24(c.g[k] + y, L.cy, L.yb0r),
                                                   mad24(c.b[k] + y, L.cy, L.yb0r), ta);
                w[2 * k + h] = __builtin_amdgcn_perm(px, px, L.perm32);
            }
    } else {
        int t[48];
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int y = Y[2 * k + h];
                t[6 * k + 3 * h + 0] = mad24(c.r[k] + y, L.cy, L.yb0r);
                t[6 * k + 3 * h + 1] = mad24(c.g[k] + y, L.cy, L.yb0r);
                t[6 * k + 3 * h + 2] = mad24(c.b[k] + y, L.cy, L.yb0r);
            }
#pragma unroll
        for (int k = 0; k < 12; k++) w[k] = pack4_u8_shr16(t[4 * k], t[4 * k + 1], t[4 * k + 2], t[4 * k + 3]);
    }
}

// Transpose the wave's packed pixels through its private LDS region and store them as contiguous 16-byte chunks.
// seg: global address of the first byte of this wave's 1024-pixel segment (16-byte aligned), seg_bytes: valid bytes.
// LDS layout: lane l owns dwords [l*LS, l*LS + NDW); LS = 20 for 32 bpp (bank-conflict padding), 12 for 24 bpp.
template <int BPP, bool XPOSE = true, bool NT = true>
__device__ __forceinline__ void wave_store16(uint8_t *seg, int seg_bytes, const uint32_t (&w)[4 * BPP], uint32_t *lds, int lane)
{
    constexpr int NDW = 4 * BPP, LS = BPP == 4 ? 20 : 12, NCH = NDW / 4;
    if constexpr (!XPOSE) { // experiment: each lane stores its own 16-pixel run (lane stride 48/64 bytes)
#pragma unroll
        for (int k = 0; k < NCH; k++) {
            const int off = lane * NDW * 4 + 16 * k;
            u32x4 t = { w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3] };
            if (off + 16 <= seg_bytes) { if constexpr (NT) gstore16_nt(seg + off, t); else *(SWS_GLOBAL u32x4 *)(seg + off) = t; }
            else if (off < seg_bytes) gstore_partial(seg + off, t, seg_bytes - off);
        }
        return;
    }
    u32x4 *lw = (u32x4 *)(lds + lane * LS);
#pragma unroll
    for (int k = 0; k < NCH; k++) { u32x4 t = { w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3] }; lw[k] = t; }
    // same-wave LDS hand-off: DS instructions of one wave execute in order; only stop compiler reordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        const int c = j * 64 + lane;              // 16-byte chunk index inside the segment
        const int off = 16 * c;
        const u32x4 v = *(const u32x4 *)(lds + (c / NCH) * LS + (c % NCH) * 4);
        if (off + 16 <= seg_bytes) { if constexpr (NT) gstore16_nt(seg + off, v); else *(SWS_GLOBAL u32x4 *)(seg + off) = v; }
        else if (off < seg_bytes) gstore_partial(seg + off, v, seg_bytes - off);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();              // the next row reuses the LDS region
}

// ------------------------------------------------------------------------------------------
// C2a: unscaled yuv420p / yuv422p -> packed RGB (yuv2rgb.c:68-559).  Wave = 1024 pixels x 2 rows.
// grid.x over (row pair, 1024-pixel segment) in units of waves, 4 waves per block; grid.z = frame.
// Requires 16-byte aligned planes/strides (else the host picks sws_k_yuv2rgb_unscaled).
// ------------------------------------------------------------------------------------------
template <int BPP, bool SWAP_RB, bool XPOSE = true, bool NT = true, int EXP = 0>
__global__ void __launch_bounds__(256) sws_k_yuv2rgb_unscaled_wave(SwsFrameSet fs, SwsDevParams p, int is422, int npairs,
                                                                   int y0, int nrowpairs)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[4 * 64 * (BPP == 4 ? 20 : 12)];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index in the block: scalar, so that rows,
                                                                        // segments, taps and branches below are wave-uniform (SGPR)
    uint32_t *lds = lds_all + wib * 64 * (BPP == 4 ? 20 : 12);
    const int npix = 2 * npairs;                               // pixels the reference's block structure covers
    const int segs = (npix + 1023) >> 10;                      // waves per row
    const int64_t wid = (int64_t)blockIdx.x * 4 + wib;
    if (wid >= (int64_t)segs * nrowpairs) return;              // whole wave exits together
    const int rp = (int)(wid / segs), seg = (int)(wid % segs);
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const SwsLutParams &L = p.lut;
    const int yrow = y0 + 2 * rp, crow = is422 ? yrow : (yrow >> 1);
    const int x = seg * 1024 + lane * 16;                      // first pixel of this lane
    const int seg_bytes = min(1024, npix - seg * 1024) * BPP;
    const int nvalid = npix - x;                               // pixels of this lane inside the row (may be <= 0)
    int Y[16], U[8], V[8];
    uint32_t w[4 * BPP];
    Chroma8 c;
#pragma unroll
    for (int l = 0; l < 2; l++) {
        if (l == 0 || is422) {   // 4:2:0: both rows of the pair share the chroma row and its LUT indices
            const int cr = crow + (is422 ? l : 0);
            unpack8(load8_or_tail(f.src[1] + (int64_t)cr * f.srcStride[1] + (x >> 1), nvalid >> 1), U);
            unpack8(load8_or_tail(f.src[2] + (int64_t)cr * f.srcStride[2] + (x >> 1), nvalid >> 1), V);
            chroma8<SWAP_RB>(L, U, V, c);
        }
        unpack16(load16_or_tail(f.src[0] + (int64_t)(yrow + l) * f.srcStride[0] + x, nvalid), Y);
        if constexpr (SWS_EXP(1)) { // experiment: memory-only floor (no LUT arithmetic)
#pragma unroll
            for (int k = 0; k < 4 * BPP; k++) w[k] = (uint32_t)(Y[k & 15] + U[k & 7] * 256 + V[(k + 3) & 7] * 65536);
        } else lut16<BPP>(L, Y, c, w);
        uint8_t *segp = f.dst[0] + (int64_t)(yrow + l) * f.dstStride[0] + (int64_t)seg * 1024 * BPP;
        wave_store16<BPP, XPOSE, NT>(segp, seg_bytes, w, lds, lane);
    }
}

// ------------------------------------------------------------------------------------------
// Second generation of the kernel above (same decomposition, same results, about half the vector instructions):
//  * vertical filters run on v_dot2_i32_i16: two vertically adjacent source rows are byte-interleaved into (lo | hi << 16)
//    pairs with one v_perm_b32 per sample and multiplied by a scalar tap pair.  The sums are formed on the raw bytes:
//    ((1 << 18) + sum((u << 7) * w)) >> 19 == (2048 + sum(u * w)) >> 12 exactly (both sides floor the same multiple of 128).
//  * the accumulated chroma is shifted, clamped to u8 and packed by v_ashr_pk_u8_i32 (the index tables of yuv2rgb.c have
//    head-room entries equal to the clamped ones, so clamping is what the table does);
//  * per pixel pair the chroma side of the LUT is folded into three addends A_c = (base_c + idx_c(u, v)) * cy + yb0r, so that a
//    pixel costs one multiply-add per channel: t_c = Y * cy + A_c, out_c = clip_u8(t_c >> 16);
//  * 32 bpp: the alpha byte position is a template parameter (operand order of the pack) instead of a v_perm per pixel;
//  * segments that lie completely inside the row take a store path without per-lane bounds checks.
// ------------------------------------------------------------------------------------------
template <int BPP, bool FULL, bool NT = true>
__device__ __forceinline__ void wave_store16_v2(uint8_t *seg, int seg_bytes, const uint32_t (&w)[4 * BPP], uint32_t *lds, int lane)
{
    constexpr int NDW = 4 * BPP, LS = BPP == 4 ? 20 : 12, NCH = NDW / 4;
    u32x4 *lw = (u32x4 *)(lds + lane * LS);
#pragma unroll
    for (int k = 0; k < NCH; k++) { u32x4 t = { w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3] }; lw[k] = t; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        const int c = j * 64 + lane;
        const int off = 16 * c;
        const u32x4 v = *(const u32x4 *)(lds + (c / NCH) * LS + (c % NCH) * 4);
        if constexpr (FULL) gstore16_nt(seg + off, v);
        else {
            if (off + 16 <= seg_bytes) gstore16_nt(seg + off, v);
            else if (off < seg_bytes) gstore_partial(seg + off, v, seg_bytes - off);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// uvp[m >> 1] holds the clamped bytes {U_m, V_m, U_m+1, V_m+1}; Y[16] raw luma values -> packed pixels
template <int BPP, bool SWAP_RB, bool AFIRST>
__device__ __forceinline__ void lut16_v2(const SwsLutParams &L, const int (&Y)[16], const uint32_t (&uvp)[4], uint32_t (&w)[4 * BPP])
{
    int A0[8], A1[8], A2[8];   // addends of the first, green and third channel per pixel pair
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const uint32_t d = uvp[m >> 1];
        const int cu = (int)((d >> (16 * (m & 1))) & 0xFF), cv = (int)((d >> (16 * (m & 1) + 8)) & 0xFF);
        const int ir = L.base_r + (__mul24(cv, L.crv) >> 16);
        const int ig = L.base_g + (__mul24(cu, L.cgu) >> 16) + (__mul24(cv, L.cgv) >> 16);
        const int ib = L.base_b + (__mul24(cu, L.cbu) >> 16);
        const int ar = mad24(ir, L.cy, L.yb0r), ab = mad24(ib, L.cy, L.yb0r);
        A0[m] = SWAP_RB ? ab : ar; A1[m] = mad24(ig, L.cy, L.yb0r); A2[m] = SWAP_RB ? ar : ab;
    }
    if constexpr (BPP == 4) {
        const int ta = 255 << 16;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int m = k >> 1, y = Y[k];
            const int t0 = mad24(y, L.cy, A0[m]), t1 = mad24(y, L.cy, A1[m]), t2 = mad24(y, L.cy, A2[m]);
            w[k] = AFIRST ? pack4_u8_shr16(ta, t0, t1, t2) : pack4_u8_shr16(t0, t1, t2, ta);
        }
    } else {
        int t[48];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int m = k >> 1, y = Y[k];
            t[3 * k + 0] = mad24(y, L.cy, A0[m]); t[3 * k + 1] = mad24(y, L.cy, A1[m]); t[3 * k + 2] = mad24(y, L.cy, A2[m]);
        }
#pragma unroll
        for (int k = 0; k < 12; k++) w[k] = pack4_u8_shr16(t[4 * k], t[4 * k + 1], t[4 * k + 2], t[4 * k + 3]);
    }
}

template <int BPP, bool SWAP_RB, bool NV, bool AFIRST, int ROWS, int NCR, bool FULL>
__device__ __forceinline__ void rgb_fused_unity_v2_body(const FrameRegs &f, const SwsDevParams &p, int rg, int seg, uint32_t *lds, int lane)
{
    const SwsLutParams &L = p.lut;
    const int npix = p.dstW;
    const int x = seg * 1024 + lane * 16;
    const int seg_bytes = min(1024, npix - seg * 1024) * BPP;
    const int nvalid = FULL ? 16 : npix - x;
    const int lfs = p.vLumFs, cfs = p.vChrFs;
    const int lH = p.srcH - 1, cH = p.chrSrcH - 1;
    const int yb = rg * ROWS;
    const int nrows = min(ROWS, p.dstH - yb);

    // per output row: chroma window start (scalar); taps one per lane, lane l holds tap l - NCR (v_readlane broadcasts them)
    int firstC[ROWS], wv[ROWS];
    int cmin = 0x7fffffff, cmax = -1;
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        const int y = min(yb + r, p.dstH - 1);
        const int cy = y >> p.chrDstVSub;
        firstC[r] = max(1 - cfs, p.vChrPos[cy]);
        const int tl = lane - NCR;
        wv[r] = (tl >= 0 && tl < cfs) ? (int)p.vChrF[cy * cfs + tl] : 0;
        cmin = min(cmin, firstC[r]); cmax = max(cmax, firstC[r] + cfs - 1);
    }
    const bool u1 = p.u_plane_src == 1;
    const uint8_t *ub = u1 ? f.src[1] : f.src[2], *vb = u1 ? f.src[2] : f.src[1];
    const int us = u1 ? f.srcStride[1] : f.srcStride[2], vs = u1 ? f.srcStride[2] : f.srcStride[1];
    u32x4 craw[NCR];
#pragma unroll
    for (int i = 0; i < NCR; i++) {
        const int cr = cmin + i;
        // straight-line code on purpose (wave-uniform branches around the accumulation made the compiler copy the 32 accumulators
        // per path): rows beyond the window are re-reads of its last row and meet zero taps; the host picks the smallest NCR
        const int srow = min(max(min(cr, cmax), 0), cH);
        if constexpr (NV) craw[i] = load16_or_tail(f.src[1] + (int64_t)srow * f.srcStride[1] + x, nvalid);
        else {
            const u32x2 a = load8_or_tail(ub + (int64_t)srow * us + (x >> 1), nvalid >> 1);
            const u32x2 b = load8_or_tail(vb + (int64_t)srow * vs + (x >> 1), nvalid >> 1);
            u32x4 t = { a[0], a[1], b[0], b[1] };
            craw[i] = t;
        }
    }
    // luma rows of the identity case are independent of the chroma work: issue them now
    const bool lum_unity = lfs == 1 && p.vLumF[yb * lfs] == 4096 && p.vLumF[min(yb + ROWS - 1, p.dstH - 1) * lfs] == 4096;
    u32x4 yraw[ROWS];
    if (lum_unity) {
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int y = min(yb + r, p.dstH - 1);
            yraw[r] = load16_or_tail(f.src[0] + (int64_t)min(max(p.vLumPos[y], 0), lH) * f.srcStride[0] + x, nvalid);
        }
    }
    int acc[ROWS][16];
#pragma unroll
    for (int r = 0; r < ROWS; r++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc[r][k] = 2048;
#pragma unroll
    for (int ip = 0; ip < NCR / 2; ip++) {
        const int cr = cmin + 2 * ip;
        uint32_t P[16];
        interleave_rows(craw[2 * ip], craw[2 * ip + 1], P);
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int j0 = cr - firstC[r] + NCR;                                   // in [1, 2 * NCR)
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane(wv[r], j0) & 0xFFFFu;
            const uint32_t w1 = (uint32_t)__builtin_amdgcn_readlane(wv[r], j0 + 1);
            const uint32_t wp = w0 | (w1 << 16);
#pragma unroll
            for (int k = 0; k < 16; k++) acc[r][k] = sdot2(P[k], wp, acc[r][k]);
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        if (r >= nrows) break;
        const int y = yb + r;
        int Y[16];
        if (lum_unity) unpack16(yraw[r], Y);
        else {
            const int16_t *lf = p.vLumF + y * lfs;
            const int firstL = max(1 - lfs, p.vLumPos[y]);
#pragma unroll
            for (int k = 0; k < 16; k++) Y[k] = 2048;
            for (int j = 0; j < lfs; j += 2) {
                const int r0 = min(max(firstL + j, 0), lH), r1 = min(max(firstL + j + 1, 0), lH);
                const u32x4 a = load16_or_tail(f.src[0] + (int64_t)r0 * f.srcStride[0] + x, nvalid);
                const u32x4 b = load16_or_tail(f.src[0] + (int64_t)r1 * f.srcStride[0] + x, nvalid);
                const uint32_t wp = ((uint32_t)(int)lf[j] & 0xFFFFu) | (j + 1 < lfs ? (uint32_t)(int)lf[j + 1] << 16 : 0u);
                uint32_t P[16];
                interleave_rows(a, b, P);
#pragma unroll
                for (int k = 0; k < 16; k++) Y[k] = sdot2(P[k], wp, Y[k]);
            }
#pragma unroll
            for (int k = 0; k < 16; k++) Y[k] >>= 12;
        }
        // chroma: shift, clamp, pack {U_m, V_m, U_m+1, V_m+1}
        uint32_t uvp[4];
        if constexpr (NV) {
            if (p.uv_swap_src) {
#pragma unroll
                for (int q = 0; q < 4; q++) uvp[q] = pack4_u8_shr12(acc[r][4 * q + 1], acc[r][4 * q], acc[r][4 * q + 3], acc[r][4 * q + 2]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++) uvp[q] = pack4_u8_shr12(acc[r][4 * q], acc[r][4 * q + 1], acc[r][4 * q + 2], acc[r][4 * q + 3]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) uvp[q] = pack4_u8_shr12(acc[r][2 * q], acc[r][8 + 2 * q], acc[r][2 * q + 1], acc[r][8 + 2 * q + 1]);
        }
        uint32_t w[4 * BPP];
        lut16_v2<BPP, SWAP_RB, AFIRST>(L, Y, uvp, w);
        uint8_t *segp = f.dst[0] + (int64_t)y * f.dstStride[0] + (int64_t)seg * 1024 * BPP;
        wave_store16_v2<BPP, FULL>(segp, seg_bytes, w, lds, lane);
    }
}

template <int BPP, bool SWAP_RB, bool NV, bool AFIRST, int ROWS, int NCR>
__global__ void __launch_bounds__(256) sws_k_rgb_fused_unity_wave2(SwsFrameSet fs, SwsDevParams p)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[4 * 64 * (BPP == 4 ? 20 : 12)];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *lds = lds_all + wib * 64 * (BPP == 4 ? 20 : 12);
    const int segs = (p.dstW + 1023) >> 10;
    const int rgroups = (p.dstH + ROWS - 1) / ROWS;
    const int per_xcd = gridDim.x >> 3;                          // XCD-aware block order, see sws_k_rgb_fused_unity_wave
    const int lb = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int64_t wid = (int64_t)lb * 4 + wib;
    if (wid >= (int64_t)segs * rgroups) return;
    const int rg = (int)(wid / segs), seg = (int)(wid % segs);
    const FrameRegs f = load_frame(fs, blockIdx.z);
    if (seg * 1024 + 1024 <= p.dstW) rgb_fused_unity_v2_body<BPP, SWAP_RB, NV, AFIRST, ROWS, NCR, true>(f, p, rg, seg, lds, lane);
    else rgb_fused_unity_v2_body<BPP, SWAP_RB, NV, AFIRST, ROWS, NCR, false>(f, p, rg, seg, lds, lane);
}


} // namespace swsk
