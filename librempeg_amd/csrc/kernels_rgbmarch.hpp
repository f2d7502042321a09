// Marching packed-RGB kernel for the identity-horizontal / identity-vertical-luma chain (C2b, C4: same-size conversions whose
// chroma is up-sampled vertically): hScale8To15_c with one tap + packed_vscale + yuv2rgb24_X_c / yuv2rgbx32_X_c
// (+ nv12ToUV_c), swscale.c:128-142, vscale.c:109-171, output.c:1788-1840.
//
// A wave owns a 1024-pixel column strip (lane = 16 adjacent pixels) and walks down a band of output-row pairs:
//  * the chroma source rows live in a ring of NCR register rows; a step loads only the row that enters the window, the loads of
//    step g+1 (chroma row, two luma rows) and the plan entry of step g+2 are issued before step g is computed;
//  * everything a step needs from the filter banks is a 64-byte host-built plan entry fetched with scalar loads;
//  * vertical chroma on v_dot2_i32_i16 over byte-interleaved row pairs (v_perm_b32).  The sums are formed on the raw bytes:
//    ((1 << 18) + sum((u << 7) * w)) >> 19 == (2048 + sum(u * w)) >> 12 exactly (both sides floor the same multiple of 128).
//    With NCR = 5 (4 taps at 2x up-sampling: five rows per row pair) the sixth slot of the three pairs carries the rounding
//    constant (sample 1 x tap 2048, host plan), so the accumulators are never initialised;
//  * shift + clamp + pack by v_ashr_pk_u8_i32 (the index tables of yuv2rgb.c have head-room entries equal to the clamped ones, so
//    clamping is what the table does); per pixel pair the chroma side of the LUT is folded into three addends
//    A_c = (base_c + idx_c(u, v)) * cy + yb0r, a pixel then costs one multiply-add per channel: out_c = clip_u8((Y * cy + A_c) >> 16);
//  * the step is computed in two halves of 8 pixels per lane, one after the other (scheduling barriers keep them apart): the live
//    set stays under 128 VGPRs without spills, i.e. 4 waves per SIMD instead of the 3 the one-piece form got;
//  * the packed pixels of a row go through a wave-private LDS transpose so that every store instruction writes one contiguous KiB;
//  * memory goes through buffer descriptors: constant per-lane offsets, scalar row offsets, reads past the end of a row stay
//    inside the plane's descriptor, stores past the end of a row are dropped by the per-row destination descriptor.  vmcnt counts
//    loads and stores together and the compiler has to assume they retire out of order: the ring advances (= the wait for the
//    prefetch) BEFORE the step's stores are issued, and only buffer (never flat) stores are used.
#pragma once
#include "wave_util.hpp"

namespace swsk {

#ifdef SWS_HIP_PROFILING
#define SWS_MEXP(n) (EXP == (n))
#else
#define SWS_MEXP(n) false
#endif

// The chroma side of the LUT as two 256-entry tables in LDS, built once per workgroup (the reference's table_rV / table_gU /
// table_gV / table_bU, yuv2rgb.c:901-961, with the luma ramp's slope and offset folded in):
//   tabV[v] = { (base_r + ((v * crv) >> 16)) * cy + yb0r,  (base_g + ((v * cgv) >> 16)) * cy + yb0r }
//   tabU[u] = { (base_b + ((u * cbu) >> 16)) * cy + yb0r,  ((u * cgu) >> 16) * cy }
// so that a pixel pair costs two 8-byte gathers and one add instead of twelve vector instructions: A_r = tabV[v].x,
// A_b = tabU[u].x, A_g = tabV[v].y + tabU[u].y.
struct LutTabs { const uint8_t *v, *u; };
__device__ __forceinline__ void build_lut_tabs(const SwsLutParams &L, u32x2 *tv, u32x2 *tu, int t)
{
    if (t < 256) {
        const int ir = L.base_r + (__mul24(t, L.crv) >> 16), gv = L.base_g + (__mul24(t, L.cgv) >> 16);
        const int ib = L.base_b + (__mul24(t, L.cbu) >> 16), gu = __mul24(t, L.cgu) >> 16;
        u32x2 ev = { (uint32_t)mad24(ir, L.cy, L.yb0r), (uint32_t)mad24(gv, L.cy, L.yb0r) };
        u32x2 eu = { (uint32_t)mad24(ib, L.cy, L.yb0r), (uint32_t)__mul24(gu, L.cy) };
        tv[t] = ev; tu[t] = eu;
    }
}

// 8 pixels (two luma dwords) with their 4 chroma pairs (uv0 = clamped bytes {U0, V0, U1, V1}, uv1 = {U2, V2, U3, V3}) -> 2 * BPP dwords
template <int BPP, bool SWAP_RB, bool AFIRST>
__device__ __forceinline__ void lut8(const SwsLutParams &L, const LutTabs &T, uint32_t y0, uint32_t y1, uint32_t uv0, uint32_t uv1, uint32_t (&w)[2 * BPP])
{
    int A0[4], A1[4], A2[4];   // addends of the first, green and third channel per pixel pair
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const uint32_t d = m < 2 ? uv0 : uv1;
        const uint32_t ou = ((d >> (16 * (m & 1))) & 0xFFu) << 3, ov = ((d >> (16 * (m & 1) + 8)) & 0xFFu) << 3;
        const u32x2 ev = *(const u32x2 *)(T.v + ov), eu = *(const u32x2 *)(T.u + ou);
        A0[m] = (int)(SWAP_RB ? eu[0] : ev[0]); A1[m] = (int)(ev[1] + eu[1]); A2[m] = (int)(SWAP_RB ? ev[0] : eu[0]);
    }
    int t[24];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int m = k >> 1, y = (int)(((k < 4 ? y0 : y1) >> (8 * (k & 3))) & 0xFF);
        t[3 * k + 0] = mad24(y, L.cy, A0[m]); t[3 * k + 1] = mad24(y, L.cy, A1[m]); t[3 * k + 2] = mad24(y, L.cy, A2[m]);
    }
    if constexpr (BPP == 4) {
        const int ta = 255 << 16;
#pragma unroll
        for (int k = 0; k < 8; k++)
            w[k] = AFIRST ? pack4_u8_shr16(ta, t[3 * k], t[3 * k + 1], t[3 * k + 2]) : pack4_u8_shr16(t[3 * k], t[3 * k + 1], t[3 * k + 2], ta);
    } else {
#pragma unroll
        for (int k = 0; k < 6; k++) w[k] = pack4_u8_shr16(t[4 * k], t[4 * k + 1], t[4 * k + 2], t[4 * k + 3]);
    }
}

#ifndef SWS_MARCH_ATTR
#define SWS_MARCH_ATTR
#endif
template <int BPP, bool SWAP_RB, bool NV, bool AFIRST, int NCR, int EXP = 0>
__global__ void __launch_bounds__(256) SWS_MARCH_ATTR sws_k_rgb_march(SwsFrameSet fs, SwsDevParams p, const SwsRgbGroupPlan *__restrict__ plan,
                                                                     int ngroups, int bands, int band_groups)
{
    constexpr int LS = BPP == 4 ? 20 : 12, NDW = 4 * BPP, NCH = NDW / 4;
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[4 * 2 * 64 * LS];   // per wave: one region per output row of a step
    __shared__ __attribute__((aligned(16))) u32x2 lds_tab[2][256];
    build_lut_tabs(p.lut, lds_tab[0], lds_tab[1], (int)threadIdx.x);
    __syncthreads();
    const LutTabs T = { (const uint8_t *)lds_tab[0], (const uint8_t *)lds_tab[1] };
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *lds = lds_all + wib * 2 * 64 * LS;
    const int segs = (p.dstW + 1023) >> 10;
    const int wid = blockIdx.x * 4 + wib;                        // the 4 waves of a block: adjacent segments of one band
    if (wid >= segs * bands) return;
    const int band = wid / segs, seg = wid % segs;
    const int g0 = band * band_groups, g1 = min(ngroups, g0 + band_groups);
    if (g0 >= g1) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const SwsLutParams &L = p.lut;
    const int x = seg * 1024 + lane * 16;
    const int seg_bytes = min(1024, p.dstW - seg * 1024) * BPP;
    const int cH = p.chrSrcH - 1;
    const bool u1 = p.u_plane_src == 1;
    // descriptors: whole planes for the sources (strides are positive and planes < 2 GiB: checked on the host)
    // (whole rows including their padding: a dword that is only partially inside the visible row must not be cut off)
    const sws_rsrc_t ry = make_rsrc(f.src[0], (uint32_t)f.srcStride[0] * (uint32_t)p.srcH);
    const sws_rsrc_t ru = make_rsrc(NV ? f.src[1] : (u1 ? f.src[1] : f.src[2]),
                                    (uint32_t)(NV ? f.srcStride[1] : (u1 ? f.srcStride[1] : f.srcStride[2])) * (uint32_t)p.chrSrcH);
    const sws_rsrc_t rv = make_rsrc(NV ? f.src[1] : (u1 ? f.src[2] : f.src[1]),
                                    (uint32_t)(NV ? f.srcStride[1] : (u1 ? f.srcStride[2] : f.srcStride[1])) * (uint32_t)p.chrSrcH);
    const int us = NV ? f.srcStride[1] : (u1 ? f.srcStride[1] : f.srcStride[2]), vs = NV ? f.srcStride[1] : (u1 ? f.srcStride[2] : f.srcStride[1]);
    const int ys = f.srcStride[0];
    const int cvoff = NV ? x : (x >> 1);
    auto load_crow = [&](int cr) -> u32x4 {
        const int srow = min(max(cr, 0), cH);
        if constexpr (SWS_MEXP(5)) {   // experiment: non-temporal loads
            if constexpr (NV) return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, cvoff, srow * us, 2));
            else {
                const u32x2 a = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(ru, cvoff, srow * us, 2));
                const u32x2 b = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rv, cvoff, srow * vs, 2));
                u32x4 t = { a[0], a[1], b[0], b[1] };
                return t;
            }
        }
        if constexpr (NV) return bload16(ru, cvoff, srow * us);
        else {
            const u32x2 a = bload8(ru, cvoff, srow * us), b = bload8(rv, cvoff, srow * vs);
            u32x4 t = { a[0], a[1], b[0], b[1] };
            return t;
        }
    };
    auto load_yrow = [&](int row) -> u32x4 {
        if constexpr (SWS_MEXP(5)) return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, x, row * ys, 2));
        return bload16(ry, x, row * ys);
    };

    constexpr bool BIAS = (NCR & 1) != 0;
    constexpr int NPAIR = (NCR + 1) / 2;
    const uint32_t ones = 0x01010101u;
    SwsRgbGroupPlan e = plan[g0], en = plan[min(g0 + 1, g1 - 1)];
    u32x4 craw[NCR];
#pragma unroll
    for (int i = 0; i < NCR; i++) craw[i] = load_crow(e.cbase + i);
    u32x4 yr0 = load_yrow(e.ylum0), yr1 = load_yrow(e.ylum1);
    // finish the initial fill here: otherwise the waits for it inside the loop (counted from the newest request) also drain the
    // previous step's stores in every later iteration
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)

    for (int g = g0; g < g1; g++) {
        const SwsRgbGroupPlan en2 = plan[min(g + 2, g1 - 1)];
        const int delta = en.cbase - e.cbase;                     // 0 or 1 (checked on the host)
        // prefetch of the next step (the last step of a band re-reads its own rows: harmless)
        const u32x4 n0 = load_crow(en.cbase + NCR - 1);
        const u32x4 ny0 = load_yrow(en.ylum0), ny1 = load_yrow(en.ylum1);
        const bool row1 = 2 * g + 1 < p.dstH;                     // (odd heights: the last pair has one row)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            // source dwords of this half in a ring row: planar {U 4h..4h+3, V 4h..4h+3} = dwords h and 2 + h; nv12 {U,V,U,V} x 2 = dwords 2h, 2h + 1
            constexpr int dummy = 0; (void)dummy;
            const int da = NV ? 2 * h : h, db = NV ? 2 * h + 1 : 2 + h;
            int acc[2][8];
            if constexpr (!BIAS) {
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int k = 0; k < 8; k++) acc[r][k] = 2048;
            }
#pragma unroll
            for (int ip = 0; ip < NPAIR; ip++) {
                const bool second = 2 * ip + 1 < NCR;
                const u32x4 &ra = craw[2 * ip], &rb = craw[second ? 2 * ip + 1 : 2 * ip];
                uint32_t P[8];
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const uint32_t a = q ? ra[db] : ra[da], b = second ? (q ? rb[db] : rb[da]) : ones;
                    P[4 * q + 0] = __builtin_amdgcn_perm(b, a, 0x0c040c00u);
                    P[4 * q + 1] = __builtin_amdgcn_perm(b, a, 0x0c050c01u);
                    P[4 * q + 2] = __builtin_amdgcn_perm(b, a, 0x0c060c02u);
                    P[4 * q + 3] = __builtin_amdgcn_perm(b, a, 0x0c070c03u);
                }
                if constexpr (SWS_MEXP(2)) {       // experiment: no vertical filter
#pragma unroll
                    for (int k = 0; k < 8; k++) { acc[0][k] = (int)P[k] + (ip ? acc[0][k] : 0); acc[1][k] = (int)P[k] + (ip ? acc[1][k] : 0); }
                    continue;
                }
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int k = 0; k < 8; k++) acc[r][k] = (BIAS && ip == 0) ? sdot2_first_s(P[k], e.wp[r][ip]) : sdot2(P[k], e.wp[r][ip], acc[r][k]);
            }
#pragma unroll
            for (int r = 0; r < 2; r++) {
                // chroma: shift, clamp, pack {U_m, V_m, U_m+1, V_m+1}
                uint32_t uv[2];
                if constexpr (NV) {
                    if (p.uv_swap_src) {
#pragma unroll
                        for (int q = 0; q < 2; q++) uv[q] = pack4_u8_shr12(acc[r][4 * q + 1], acc[r][4 * q], acc[r][4 * q + 3], acc[r][4 * q + 2]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 2; q++) uv[q] = pack4_u8_shr12(acc[r][4 * q], acc[r][4 * q + 1], acc[r][4 * q + 2], acc[r][4 * q + 3]);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 2; q++) uv[q] = pack4_u8_shr12(acc[r][2 * q], acc[r][4 + 2 * q], acc[r][2 * q + 1], acc[r][4 + 2 * q + 1]);
                }
                const u32x4 &yr = r ? yr1 : yr0;
                uint32_t w[2 * BPP];
                if constexpr (SWS_MEXP(1)) {       // experiment: memory-only floor (no LUT stage)
#pragma unroll
                    for (int k = 0; k < 2 * BPP; k++) w[k] = yr[2 * h + (k & 1)] + uv[k & 1];
                } else lut8<BPP, SWAP_RB, AFIRST>(L, T, yr[2 * h], yr[2 * h + 1], uv[0], uv[1], w);
                // park the packed pixels in the wave's LDS region of this row (lane-major); they are stored after the ring advanced
                uint32_t *lw = lds + r * 64 * LS + lane * LS + h * 2 * BPP;
                if constexpr (BPP == 4) {
                    u32x4 t0 = { w[0], w[1], w[2], w[3] }, t1 = { w[4], w[5], w[6], w[7] };
                    ((u32x4 *)lw)[0] = t0; ((u32x4 *)lw)[1] = t1;
                } else {
#pragma unroll
                    for (int k = 0; k < 3; k++) { u32x2 t = { w[2 * k], w[2 * k + 1] }; ((u32x2 *)lw)[k] = t; }
                }
            }
            __builtin_amdgcn_sched_barrier(0);                    // keep the halves apart: the second half reuses the first one's registers
        }
        // advance the ring BEFORE this step's stores are issued (see the header)
        if (delta == 1) {
#pragma unroll
            for (int i = 0; i + 1 < NCR; i++) craw[i] = craw[i + 1];
            craw[NCR - 1] = n0;
        }
        yr0 = ny0; yr1 = ny1;
        // pin the hand-over here: without it the register copies (and with them the wait for the prefetch) sink below the stores
        asm volatile("" : "+v"(yr0), "+v"(yr1), "+v"(craw[NCR - 1]) :: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // transposed read-back: every store instruction writes one contiguous KiB; the row's descriptor drops what lies beyond it
#pragma unroll
        for (int r = 0; r < 2; r++) {
            if (r == 1 && !row1) break;
            const int y = 2 * g + r;
            uint8_t *segp = f.dst[0] + (int64_t)y * f.dstStride[0] + (int64_t)seg * 1024 * BPP;
            const sws_rsrc_t rd = make_rsrc(segp, SWS_MEXP(3) ? (p.dstW == 12345 ? 16u : 0u) : (uint32_t)seg_bytes);   // a dword that straddles the end is dropped as a whole
            const uint32_t *lr = lds + r * 64 * LS;
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                const int c = j * 64 + lane;
                const u32x4 v = *(const u32x4 *)(lr + (c / NCH) * LS + (c % NCH) * 4);
                if constexpr (SWS_MEXP(4)) __builtin_amdgcn_raw_buffer_store_b128(v, rd, 16 * c, 0, 0);   // experiment: plain stores
                else __builtin_amdgcn_raw_buffer_store_b128(v, rd, 16 * c, 0, 2 /* nt */);
            }
            if constexpr (BPP == 3) {
                if (seg_bytes & 2) {      // 24 bpp rows end on a multiple of 6 bytes: a last half dword
                    const int b = seg_bytes - 2;
                    // (a buffer store, not a flat one: flat stores complete out of order and would turn every wait into vmcnt(0))
                    if (lane == 0) __builtin_amdgcn_raw_buffer_store_b16(*(const uint16_t *)((const uint8_t *)lr + (b / (NDW * 4)) * LS * 4 + b % (NDW * 4)), rd, b, 0, 0);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();              // the next step reuses the LDS region
        e = en; en = en2;
    }
}

} // namespace swsk
