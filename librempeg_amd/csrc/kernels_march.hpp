// Wave-marching fused h+v polyphase kernel (planar / semi-planar YUV outputs, planar <= 15-bit sources).
//
// ONE WAVEFRONT owns a strip of 128 output columns (2 per lane) and marches down a band of output rows,
// exactly like the reference's line ring buffer (libswscale/slice.c, ff_swscale swscale.c:412-534) but with the
// ring in LDS and no workgroup barrier anywhere:
//   * when the vertical window of the next output row needs source rows that are not h-scaled yet, the wave
//     stages the next PAIR of source rows (16-byte global loads, prefetched one step ahead) in its private LDS
//     area, runs the horizontal filter on dword-packed sample pairs (v_dot2c_i32_i16, taps host-padded to a
//     4-sample alignment so that every LDS read is an aligned ds_read_b64) and stores {even row, odd row} dwords
//     into a ring of RING row pairs;
//   * the vertical filter reads the ring with one ds_read_b64 per tap pair (2 columns) and the writer stores
//     2 outputs per lane (256 B / 128 B contiguous per wave).
// Each source row is h-scaled once per band (halo = vertical filter size per band), nothing but source pixels
// and output pixels ever crosses HBM.  Arithmetic is the generic kernels'.
#pragma once
#include "kernels_tile.hpp"

namespace swsk {

constexpr int MARCH_RING = 9;   // row pairs kept (>= max vertical tap pairs, 8)

template <int NPH>
__device__ __forceinline__ void march_hscale2(const uint32_t *s0, const uint32_t *s1, const uint32_t (&t)[8], int &a, int &b)
{
    a = 0; b = 0;
#pragma unroll
    for (int k = 0; k < NPH; k += 2) {       // aligned ds_read_b64: two tap pairs per read
        const u32x2 p0 = *(const u32x2 *)(s0 + k), p1 = *(const u32x2 *)(s1 + k);
        a = dot2(p0[0], t[k], a); a = dot2(p0[1], t[k + 1], a);
        b = dot2(p1[0], t[k], b); b = dot2(p1[1], t[k + 1], b);
    }
}

template <int NP, int NCOMP>
__device__ __forceinline__ void march_vscale(const uint32_t *ring, int slot, int lane, const uint32_t *vt, int (&acc)[NCOMP][2])
{
#pragma unroll
    for (int k = 0; k < NP; k++) {
        const uint32_t w = vt[k];
#pragma unroll
        for (int c = 0; c < NCOMP; c++) {
            const u32x2 hv = *(const u32x2 *)(ring + (c * MARCH_RING + slot) * 128 + 2 * lane);
            acc[c][0] = dot2(hv[0], w, acc[c][0]); acc[c][1] = dot2(hv[1], w, acc[c][1]);
        }
        slot = slot + 1 == MARCH_RING ? 0 : slot + 1;
    }
}

template <bool SRC16, int NCOMP>
__global__ void __launch_bounds__(256) sws_k_march_dot2(SwsFrameSet fs, SwsDevParams p, SwsMarchGeom g)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool chroma = g.chroma != 0;
    const int W = chroma ? p.chrDstW : p.dstW, H = chroma ? p.chrDstH : p.dstH, sH = chroma ? p.chrSrcH : p.srcH;
    const int wid = blockIdx.x * 4 + wib;
    if (wid >= g.strips * g.bands) return;                     // whole wave
    const int strip = wid % g.strips, band = wid / g.strips;
    const int compsel = blockIdx.y;                            // planar chroma: 0 = U, 1 = V (NCOMP == 1)
    const FrameRegs f = load_frame(fs, blockIdx.z);
    constexpr int SPC = SRC16 ? 8 : 16;
    const int srow_dw = g.NCmax >> 1;
    // per-wave LDS: [2 buffers][2 rows][NCmax/2] staged sample pairs, then NCOMP rings [RING][128] of row-pair dwords
    const int wave_dw = NCOMP * (4 * srow_dw) + NCOMP * MARCH_RING * 128;
    uint32_t *base = (uint32_t *)smem + (size_t)wib * wave_dw;
    uint32_t *Sbuf = base;                                     // comp c, buffer b, row r: Sbuf + ((c*2 + b)*2 + r) * srow_dw
    uint32_t *ring = base + NCOMP * 4 * srow_dw;               // comp c, slot s: ring + (c*RING + s)*128

    const int x0 = strip * 128, y0 = band * g.BAND, y1 = min(H, y0 + g.BAND);
    const int cs = g.colStart[strip], ncp = g.colCount[strip];
    const int chunks = ncp / SPC;                              // 16-byte chunks per source row window
    const int32_t *hpos = chroma ? p.hChrPos : p.hLumPos, *vpos = chroma ? p.vChrPos : p.vLumPos;
    const int xa = x0 + 2 * lane;
    const bool va = xa < W, vb = xa + 1 < W;
    // horizontal taps of this lane's two columns (pairs, 4-sample aligned rows) stay in VGPRs for the whole band
    uint32_t ta[8], tb[8];
    int spa = 0, spb = 0;
    {
        const int xaa = min(xa, W - 1), xbb = min(xa + 1, W - 1);
        spa = ((hpos[xaa] & ~3) - cs) >> 1; spb = ((hpos[xbb] & ~3) - cs) >> 1;
        const uint32_t *pa = (const uint32_t *)(g.hT4 + (int64_t)xaa * g.hfs4), *pb = (const uint32_t *)(g.hT4 + (int64_t)xbb * g.hfs4);
#pragma unroll
        for (int k = 0; k < 8; k++) { ta[k] = 2 * k < g.hfs4 ? pa[k] : 0u; tb[k] = 2 * k < g.hfs4 ? pb[k] : 0u; }
    }
    const int nph = g.hfs4 >> 1, npv = g.vfs2 >> 1;

    // source plane(s)
    const uint8_t *sb[NCOMP]; int sst[NCOMP];
#pragma unroll
    for (int c = 0; c < NCOMP; c++) {
        if (!chroma) { sb[c] = f.src[0]; sst[c] = f.srcStride[0]; }
        else {
            const int comp = NCOMP == 2 ? c : compsel;         // 0 = U, 1 = V
            const bool first = (comp == 0) == (p.u_plane_src == 1);
            sb[c] = first ? f.src[1] : f.src[2]; sst[c] = first ? f.srcStride[1] : f.srcStride[2];
        }
    }
    // prefetch machinery: lane L loads chunk L (and L + 64) of rows (2q, 2q+1) of every component.  Everything that does
    // not depend on the row pair (per-lane byte offset, validity, full-16-byte flag, LDS destination) is computed once.
    constexpr int CPL = 2;                                     // chunk slots per lane per row (chunks <= 128)
    int coff[CPL]; bool cval[CPL], cfull[NCOMP][CPL]; int ldsoff[CPL];
#pragma unroll
    for (int s = 0; s < CPL; s++) {
        const int ch = lane + 64 * s;
        cval[s] = ch < chunks;
        coff[s] = (cs + ch * SPC) * (SRC16 ? 2 : 1);
        ldsoff[s] = ch * (SPC / 2);
#pragma unroll
        for (int c = 0; c < NCOMP; c++) { const int ast = sst[c] < 0 ? -sst[c] : sst[c]; cfull[c][s] = coff[s] + 16 <= ast; }
    }
    u32x4 pre[NCOMP][2][CPL];
    auto issue = [&](int pair) {
#pragma unroll
        for (int c = 0; c < NCOMP; c++) {
            const int r0 = min(2 * pair, sH - 1), r1 = min(2 * pair + 1, sH - 1);
            const uint8_t *p0 = sb[c] + (int64_t)r0 * sst[c], *p1 = sb[c] + (int64_t)r1 * sst[c];   // scalar row bases
#pragma unroll
            for (int s = 0; s < CPL; s++) {
                if (cval[s]) {
                    if (cfull[c][s]) { pre[c][0][s] = gload16(p0 + coff[s]); pre[c][1][s] = gload16(p1 + coff[s]); }
                    else {
                        const int ast = sst[c] < 0 ? -sst[c] : sst[c];
                        pre[c][0][s] = gload16_partial(p0 + coff[s], max(0, ast - coff[s]));
                        pre[c][1][s] = gload16_partial(p1 + coff[s], max(0, ast - coff[s]));
                    }
                }
            }
        }
    };
    auto commit = [&](int buf) {                               // prefetched registers -> LDS sample pairs
#pragma unroll
        for (int c = 0; c < NCOMP; c++)
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int s = 0; s < CPL; s++) {
                    if (cval[s]) {
                        uint32_t *dst = Sbuf + ((c * 2 + buf) * 2 + r) * srow_dw + ldsoff[s];
                        const u32x4 v = pre[c][r][s];
                        if constexpr (SRC16) *(u32x4 *)dst = v;
                        else {
                            u32x4 lo, hi;
#pragma unroll
                            for (int q = 0; q < 2; q++) {
                                lo[2 * q] = (v[q] & 0xFF) | ((v[q] & 0xFF00) << 8); lo[2 * q + 1] = ((v[q] >> 16) & 0xFF) | ((v[q] >> 8) & 0xFF0000);
                                hi[2 * q] = (v[q + 2] & 0xFF) | ((v[q + 2] & 0xFF00) << 8); hi[2 * q + 1] = ((v[q + 2] >> 16) & 0xFF) | ((v[q + 2] >> 8) & 0xFF0000);
                            }
                            *(u32x4 *)dst = lo; *(u32x4 *)(dst + 4) = hi;
                        }
                    }
                }
    };
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    int next_pair = (vpos[y0] & ~1) >> 1;                      // next source row pair to h-scale
    int next_slot = next_pair % MARCH_RING;                    // its ring slot (kept incrementally: no division in the loop)
    int fetched = next_pair;                                   // pair whose loads are in flight in pre[]
    int buf = 0;
    issue(fetched);
    const int bits = p.dst_bits;
    for (int y = y0; y < y1; y++) {
        const int need_first = (vpos[y] & ~1) >> 1, need_last = need_first + npv - 1;
        if (next_pair < need_first) {                          // window jumped (strong down-scaling): skip unused pairs
            next_pair = need_first; next_slot = next_pair % MARCH_RING;
            if (fetched != next_pair) { fetched = next_pair; issue(fetched); }
        }
        while (next_pair <= need_last) {
            if (fetched != next_pair) { fetched = next_pair; issue(fetched); }
            commit(buf);
            fetched = next_pair + 1;
            issue(fetched);                                    // prefetch the following pair while this one is filtered
            wave_sync();
            const int slot = next_slot;
#pragma unroll
            for (int c = 0; c < NCOMP; c++) {
                const uint32_t *s0 = Sbuf + ((c * 2 + buf) * 2 + 0) * srow_dw, *s1 = s0 + srow_dw;
                int a0, b0, a1, b1;
                switch (nph) {
#define SWS_MH(NP) case NP: march_hscale2<NP>(s0 + spa, s1 + spa, ta, a0, b0); march_hscale2<NP>(s0 + spb, s1 + spb, tb, a1, b1); break;
                SWS_MH(2) SWS_MH(4) SWS_MH(6) SWS_MH(8)
#undef SWS_MH
                default: a0 = b0 = a1 = b1 = 0; break;
                }
                const bool isc = chroma;
                const int va0 = range_sample(p, (int16_t)min(a0 >> p.hshift, p.hclip), isc), vb0 = range_sample(p, (int16_t)min(b0 >> p.hshift, p.hclip), isc);
                const int va1 = range_sample(p, (int16_t)min(a1 >> p.hshift, p.hclip), isc), vb1 = range_sample(p, (int16_t)min(b1 >> p.hshift, p.hclip), isc);
                u32x2 hv = { (uint32_t)(uint16_t)va0 | ((uint32_t)(uint16_t)vb0 << 16), (uint32_t)(uint16_t)va1 | ((uint32_t)(uint16_t)vb1 << 16) };
                *(u32x2 *)(ring + (c * MARCH_RING + slot) * 128 + 2 * lane) = hv;
            }
            buf ^= 1;
            next_pair++; next_slot = next_slot + 1 == MARCH_RING ? 0 : next_slot + 1;
        }
        wave_sync();
        // vertical filter for row y: 2 columns per lane
        const uint32_t *vt = (const uint32_t *)(g.vT2 + (int64_t)y * g.vfs2);
        int acc[NCOMP][2];
#pragma unroll
        for (int c = 0; c < NCOMP; c++) { acc[c][0] = 0; acc[c][1] = 0; }
        {   // slot of pair need_first: next_slot is the slot of pair next_pair (> need_last >= need_first, distance < RING)
            int slot = next_slot - (next_pair - need_first);
            if (slot < 0) slot += MARCH_RING;
            switch (npv) {
#define SWS_MV(NP) case NP: march_vscale<NP, NCOMP>(ring, slot, lane, vt, acc); break;
            SWS_MV(1) SWS_MV(2) SWS_MV(3) SWS_MV(4) SWS_MV(5) SWS_MV(6) SWS_MV(7) SWS_MV(8)
#undef SWS_MV
            }
        }
        // writers ("X" forms, filter size >= 2): output.c:468-483, :344-357, :554-589, :495-528
        if (!chroma || NCOMP == 1) {
            const int pl = !chroma ? 0 : (compsel == 0 ? p.u_plane_dst : p.v_plane_dst);
            uint8_t *drow = (pl == 0 ? f.dst[0] : pl == 1 ? f.dst[1] : f.dst[2]) +
                            (int64_t)y * (pl == 0 ? f.dstStride[0] : pl == 1 ? f.dstStride[1] : f.dstStride[2]);
            if (p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_NV12) {
                const int off = (chroma && compsel == 1) ? 3 : 0;
                const int o0 = clip_u8_shr((dither8(p.should_dither, y, xa + off) << 12) + acc[0][0], 19);
                const int o1 = clip_u8_shr((dither8(p.should_dither, y, xa + 1 + off) << 12) + acc[0][1], 19);
                if (vb) *(uint16_t *)(drow + xa) = (uint16_t)(o0 | (o1 << 8));
                else if (va) drow[xa] = (uint8_t)o0;
            } else {
                const int shift = 11 + 16 - bits, osh = p.dst_shift;   // p010-style and msb planar formats keep the samples in the high bits
                const uint32_t o0 = (uint32_t)clip_uintp2(((1 << (shift - 1)) + acc[0][0]) >> shift, bits) << osh;
                const uint32_t o1 = (uint32_t)clip_uintp2(((1 << (shift - 1)) + acc[0][1]) >> shift, bits) << osh;
                uint16_t *d16 = (uint16_t *)drow + xa;
                if (vb) *(uint32_t *)d16 = o0 | (o1 << 16);
                else if (va) d16[0] = (uint16_t)o0;
            }
        } else if constexpr (NCOMP == 2) {                     // semi-planar chroma: U,V interleaved
            uint8_t *drow = f.dst[1] + (int64_t)y * f.dstStride[1];
            if (p.dstKind == DSTK_NV12) {
                uint8_t o[4];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int u = clip_u8_shr((dither8(p.should_dither, y, xa + e) << 12) + acc[0][e], 19);
                    const int v = clip_u8_shr((dither8(p.should_dither, y, xa + e + 3) << 12) + acc[1][e], 19);
                    o[2 * e + p.uv_swap_dst] = (uint8_t)u; o[2 * e + 1 - p.uv_swap_dst] = (uint8_t)v;
                }
                if (vb) *(uint32_t *)(drow + 2 * xa) = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
                else if (va) { drow[2 * xa] = o[0]; drow[2 * xa + 1] = o[1]; }
            } else {
                const int shift = 11 + 16 - bits;
                uint32_t o[2];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const uint32_t u = (uint32_t)clip_uintp2(((1 << (shift - 1)) + acc[0][e]) >> shift, bits) << p.dst_shift;
                    const uint32_t v = (uint32_t)clip_uintp2(((1 << (shift - 1)) + acc[1][e]) >> shift, bits) << p.dst_shift;
                    o[e] = u | (v << 16);
                }
                uint32_t *d32 = (uint32_t *)((uint16_t *)drow + 2 * xa);
                if (vb) { u32x2 v = { o[0], o[1] }; *(u32x2 *)d32 = v; }
                else if (va) d32[0] = o[0];
            }
        }
    }
}

} // namespace swsk
