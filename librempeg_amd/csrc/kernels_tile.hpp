// Fused horizontal + vertical polyphase kernel for planar / semi-planar YUV outputs (C1, C3b shapes):
// the CPU code's line ring buffer (libswscale/slice.c) becomes an LDS tile.
//
// One workgroup produces a TW x TH tile of one component group (luma, or U+V):
//   phase 1  stage the source window the tile needs (reader applied: input.c) in LDS as 16-bit samples
//   phase 2  horizontal stage for every needed source row (hScale*_c + range conversion) -> LDS, 15/19-bit
//   phase 3  vertical stage + output writer straight from LDS (output.c planar / nv12 / p010 writers)
// No intermediate ever touches HBM.  Window origins/sizes per tile row/column are precomputed on the host
// from the filter banks (SwsTileGeom); arithmetic is the generic kernels' (same device functions).
#pragma once
#include <type_traits>
#include "kernels_generic.hpp"
#include "wave_util.hpp"

namespace swsk {

#ifndef SWS_DBG
#ifdef SWS_HIP_PROFILING
#define SWS_DBG(g, bit) ((g).debug & (bit))
#else
#define SWS_DBG(g, bit) false
#endif
#endif

template <typename HT> struct LdsSampler {
    const HT *h[3]; int r0, x0, tw;
    __device__ __forceinline__ int get(int comp, int row, int x) const { return h[comp][(row - r0) * tw + (x - x0)]; }
};

// (SK >= 0: the instantiation for one source kind -- phase 1 is a loop around the reader, kernels_generic.hpp kind_view)
template <typename HT, bool CHROMA, int SK = -1>
__global__ void __launch_bounds__(256) sws_k_tile_planar(SwsFrameSet fs, SwsDevParams pa, SwsTileGeom g)
{
    const auto &p = kind_view<SK, -1>(pa);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tx = blockIdx.x, ty = blockIdx.y, tid = threadIdx.x;
    const SwsFramePtrs &f = frame_of(fs, blockIdx.z);
    const int W = CHROMA ? p.chrDstW : p.dstW, H = CHROMA ? p.chrDstH : p.dstH;
    const int sW = CHROMA ? p.chrSrcW : p.srcW, sH = CHROMA ? p.chrSrcH : p.srcH;
    const int x0 = tx * g.TW, y0 = ty * g.TH;
    const int tw = min(g.TW, W - x0), th = min(g.TH, H - y0);
    const int r0 = g.rowStart[ty], nr = g.rowCount[ty], c0 = g.colStart[tx], nc = g.colCount[tx];
    const int16_t *hf = CHROMA ? p.hChrF : p.hLumF; const int32_t *hpos = CHROMA ? p.hChrPos : p.hLumPos;
    const int hfs = CHROMA ? p.hChrFs : p.hLumFs;
    constexpr int NCOMP = CHROMA ? 2 : 1;
    uint16_t *S = (uint16_t *)smem;                                        // [NRmax][NCmax]
    HT *Hbase = (HT *)(smem + (((size_t)g.NRmax * g.NCmax * 2 + 15) & ~(size_t)15)); // NCOMP x [NRmax][TW]
    const int hplane = g.NRmax * g.TW;

#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++) {
        const int comp = CHROMA ? 1 + ci : 0;
        // phase 1: source window -> LDS (reader fused; coordinates clamped to the plane, so taps that the filter
        // folded onto the border (utils.c:519-560) never read outside)
        // (four samples per thread and turn, their loads issued together; the reader form -- component, half-width chroma -- decided outside the loop)
        auto stage = [&](const auto &q) {
            const int tot = nr * nc;
            for (int i0 = tid; i0 < tot; i0 += 4 * 256) {
                int v[4], o[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = min(i0 + 256 * u, tot - 1);
                    const int r = i / nc, cc = i - r * nc;
                    o[u] = r * g.NCmax + cc;
                    v[u] = read_sample(q, f, comp, min(r0 + r, sH - 1), min(c0 + cc, sW - 1));
                }
#pragma unroll
                for (int u = 0; u < 4; u++) if (i0 + 256 * u < tot) S[o[u]] = (uint16_t)v[u];
            }
        };
        if (p.chr_half) stage(chr_half_view<1>(p)); else stage(chr_half_view<0>(p));
        __syncthreads();
        // phase 2: horizontal stage; thread = one output column, marching down the window rows
        HT *Hc = Hbase + ci * hplane;
        const int xl = tid % g.TW, rstep = 256 / g.TW;
        if (xl < tw) {
            const int x = x0 + xl;
            const int sp = hpos[x] - c0;
            const int16_t *taps = hf + (int64_t)x * hfs;
            for (int r = tid / g.TW; r < nr; r += rstep) {
                const uint16_t *srow = S + r * g.NCmax + sp;
                int val = 0;
                for (int j = 0; j < hfs; j++) val = mad24((int)srow[j], (int)taps[j], val);
                int v = min(val >> p.hshift, p.hclip);
                if (!p.wide) v = (int16_t)v;
                Hc[r * g.TW + xl] = (HT)range_sample(p, v, CHROMA);
            }
        }
        __syncthreads();
    }
    // phase 3: vertical stage + writer
    LdsSampler<HT> smp;
    smp.h[0] = Hbase; smp.h[1] = Hbase; smp.h[2] = Hbase + hplane; smp.r0 = r0; smp.x0 = x0; smp.tw = g.TW;
    const bool nvdst = p.dstKind == DSTK_NV12 || p.dstKind == DSTK_P010;
    for (int i = tid; i < tw * th; i += 256) {
        const int yl = i / tw, xl = i - yl * tw;
        if (!CHROMA) planar_write_one(p, smp, f, 0, x0 + xl, y0 + yl);
        else if (nvdst) nv_chroma_write_one(p, smp, f, x0 + xl, y0 + yl);
        else { planar_write_one(p, smp, f, 1, x0 + xl, y0 + yl); planar_write_one(p, smp, f, 2, x0 + xl, y0 + yl); }
    }
}

} // namespace swsk

namespace swsk {

// ------------------------------------------------------------------------------------------
// Optimised fused h+v tile kernel: planar 8-bit or <= 15-bit LE sources, 15-bit (int16) intermediates,
// vertical filter size >= 2.  Same three phases as sws_k_tile_planar, but
//   * the source window is staged with 16-byte loads (no per-sample reader),
//   * both filter stages run on v_dot2c_i32_i16: two taps per instruction on dword-packed sample pairs.
//     The host pads every tap row so that it starts on an even sample / even row (a leading zero tap when the
//     filter position is odd; SwsTileGeom.hT2/vT2), window origins are even, so pairs are always aligned;
//   * the h-scaled tile is stored as row PAIRS ({row 2k, row 2k+1} per dword) so the vertical stage reads one
//     dword per tap pair; a lane produces 4 adjacent outputs from ds_read_b128 and stores them in one go.
// Taps that fall outside the window multiply by zero (integer arithmetic: 0 * garbage == 0).
// ------------------------------------------------------------------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}

// horizontal stage for one output column over `npairs` row pairs (rows 2q, 2q+1), NP tap pairs per output.
// The kernel is instruction-issue bound, so the epilogue is as short as it gets: with the 15-bit clip (hclip == 32767)
// min(v >> sh, 32767) followed by the int16 store IS v_cvt_pk_i16_i32's signed saturation (the negative side cannot
// overflow: samples are unsigned and the taps' negative lobes are a fraction of their sum), which also packs the row
// pair into the dword the vertical stage reads.  Range conversion (rare) takes the general per-sample path.
typedef short s16x2v __attribute__((ext_vector_type(2)));
template <int NP, bool CHROMA, bool RANGE>
__device__ __forceinline__ void hscale_pairs_impl(const SwsDevParams &p, const uint32_t *S, int srow_dw, const uint32_t (&t)[NP],
                                                  uint32_t *Hcol, int tw_dw, int q0, int npairs, int qstep)
{
    const int sh = p.hshift;
    for (int q = q0; q < npairs; q += qstep) {
        const uint32_t *s0 = S + (2 * q) * srow_dw, *s1 = s0 + srow_dw;
        int a = 0, b = 0;
#pragma unroll
        for (int k = 0; k < NP; k++) { a = dot2(s0[k], t[k], a); b = dot2(s1[k], t[k], b); }
        if constexpr (RANGE) {
            int va = min(a >> sh, p.hclip), vb = min(b >> sh, p.hclip);
            va = range_sample(p, (int16_t)va, CHROMA); vb = range_sample(p, (int16_t)vb, CHROMA);
            Hcol[q * tw_dw] = (uint32_t)(uint16_t)va | ((uint32_t)(uint16_t)vb << 16);
        } else {
            Hcol[q * tw_dw] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(a >> sh, b >> sh));
        }
    }
}
template <int NP, bool CHROMA>
__device__ __forceinline__ void hscale_pairs(const SwsDevParams &p, const uint32_t *S, int srow_dw, const uint32_t *tp,
                                             uint32_t *Hcol, int tw_dw, int q0, int npairs, int qstep)
{
    uint32_t t[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) t[k] = tp[k];
    if (p.range_active || p.hclip != 32767) hscale_pairs_impl<NP, CHROMA, true>(p, S, srow_dw, t, Hcol, tw_dw, q0, npairs, qstep);
    else hscale_pairs_impl<NP, CHROMA, false>(p, S, srow_dw, t, Hcol, tw_dw, q0, npairs, qstep);
}

// vertical stage for 4 adjacent columns, NP tap pairs
template <int NP, int NCOMP>
__device__ __forceinline__ void vscale_pairs4(const uint32_t *Hp, int hplane, int tw_dw, const uint32_t *vt, int (&acc)[NCOMP][4])
{
#pragma unroll
    for (int k = 0; k < NP; k++) {
        const uint32_t w = vt[k];
#pragma unroll
        for (int ci = 0; ci < NCOMP; ci++) {
            const u32x4 hv = *(const u32x4 *)(Hp + ci * hplane + k * tw_dw);
#pragma unroll
            for (int e = 0; e < 4; e++) acc[ci][e] = dot2(hv[e], w, acc[ci][e]);
        }
    }
}

template <bool SRC16, bool CHROMA, int NT>
__global__ void __launch_bounds__(NT) sws_k_tile_dot2(SwsFrameSet fs, SwsDevParams p, SwsTileGeom g)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tx = blockIdx.x, ty = blockIdx.y, tid = threadIdx.x;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int W = CHROMA ? p.chrDstW : p.dstW, H = CHROMA ? p.chrDstH : p.dstH;
    const int x0 = tx * g.TW, y0 = ty * g.TH;
    const int tw = min(g.TW, W - x0), th = min(g.TH, H - y0);
    const int rs = g.rowStart[ty], nrp = g.rowCount[ty];     // even start row, even row count
    const int cs = g.colStart[tx], ncp = g.colCount[tx];     // window start aligned to a 16-byte source chunk
    const int32_t *hpos = CHROMA ? p.hChrPos : p.hLumPos;
    constexpr int NCOMP = CHROMA ? 2 : 1;
    constexpr int SPC = SRC16 ? 8 : 16;                      // samples per 16-byte source chunk
    uint32_t *S = (uint32_t *)smem;                          // [NRmax][NCmax/2] dwords = sample pairs
    const int srow_dw = g.NCmax >> 1;
    uint32_t *Hp = (uint32_t *)(smem + (size_t)g.NRmax * g.NCmax * 2);   // NCOMP x [NRmax/2][TW] row-pair dwords
    const int hplane = (g.NRmax >> 1) * g.TW;
    const int sH = CHROMA ? p.chrSrcH : p.srcH;

#pragma unroll
    for (int ci = 0; ci < NCOMP; ci++) {
        const bool second = CHROMA && ci == 1;
        const bool u1 = p.u_plane_src == 1;
        const uint8_t *sb = !CHROMA ? f.src[0] : ((ci == 0) == u1 ? f.src[1] : f.src[2]);
        const int sst = !CHROMA ? f.srcStride[0] : ((ci == 0) == u1 ? f.srcStride[1] : f.srcStride[2]);
        if (second) __syncthreads();                          // phase 2 of the previous component still reads S
        // ---- phase 1: 16-byte chunks of the source window -> LDS as u16 pairs ----
        // The kernel is instruction-issue bound (about 1000 wave instructions per tile, < 20 % of them dot2), so the row
        // is wave-uniform (scalar address math), a lane owns a chunk column, and 4 rows are in flight per wave.
        {
            const int chunks = ncp / SPC;
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
            const int64_t asst = sst < 0 ? -(int64_t)sst : (int64_t)sst;
            const int64_t csb = (int64_t)cs * (SRC16 ? 2 : 1);
            auto put = [&](int r, int ch, u32x4 v) {
                uint32_t *dst = S + r * srow_dw + ch * (SPC / 2);
                if constexpr (SRC16) { *(u32x4 *)dst = v; }
                else {
                    u32x4 lo, hi;                                  // bytes -> u16 pairs: v_perm_b32 with a zero byte source
                    lo[0] = __builtin_amdgcn_perm(0, v[0], 0x0c010c00u); lo[1] = __builtin_amdgcn_perm(0, v[0], 0x0c030c02u);
                    lo[2] = __builtin_amdgcn_perm(0, v[1], 0x0c010c00u); lo[3] = __builtin_amdgcn_perm(0, v[1], 0x0c030c02u);
                    hi[0] = __builtin_amdgcn_perm(0, v[2], 0x0c010c00u); hi[1] = __builtin_amdgcn_perm(0, v[2], 0x0c030c02u);
                    hi[2] = __builtin_amdgcn_perm(0, v[3], 0x0c010c00u); hi[3] = __builtin_amdgcn_perm(0, v[3], 0x0c030c02u);
                    *(u32x4 *)dst = lo; *(u32x4 *)(dst + 4) = hi;
                }
            };
            if (SWS_DBG(g, 4)) {} else
            if (chunks >= 32) {   // wide windows: lane = chunk column, wave-uniform rows (scalar address math), 4 rows in flight
                for (int ch = lane; ch < chunks; ch += 64) {
                    const int64_t boff = csb + (int64_t)ch * 16;
                    const bool full = boff + 16 <= asst;
                    const int nvalid = (int)max((int64_t)0, asst - boff);
                    for (int r0 = wave; r0 < nrp; r0 += 4 * (NT / 64)) {   // 4 rows of this wave in flight
                        u32x4 v[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int r = r0 + (NT / 64) * k;
                            if (r < nrp) {
                                const uint8_t *src = sb + (int64_t)min(rs + r, sH - 1) * sst + boff;
                                v[k] = full ? gload16(src) : gload16_partial(src, nvalid);
                            }
                        }
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int r = r0 + (NT / 64) * k;
                            if (r < nrp) put(r, ch, v[k]);
                        }
                    }
                }
            } else {              // narrow windows (few chunks per row): flat index over (row, chunk) keeps all lanes busy
                for (int i = tid; i < nrp * chunks; i += NT) {
                    const int r = i / chunks, ch = i - r * chunks;
                    const int64_t boff = csb + (int64_t)ch * 16;
                    const uint8_t *src = sb + (int64_t)min(rs + r, sH - 1) * sst + boff;
                    put(r, ch, boff + 16 <= asst ? gload16(src) : gload16_partial(src, (int)max((int64_t)0, asst - boff)));
                }
            }
        }
        __syncthreads();
        // ---- phase 2: horizontal stage on sample pairs; thread = output column, marching down the window in ROW PAIRS
        //      (one dword store per pair: {even row, odd row}) ----
        {
            const int xl = tid & (g.TW - 1), half = tid / g.TW;    // TW == 128: two interleaved pair phases
            if (xl < tw && !SWS_DBG(g, 1)) {
                const int x = x0 + xl;
                const int spd = ((hpos[x] & ~1) - cs) >> 1;          // dword index of the first (even-aligned) pair
                const uint32_t *tp = (const uint32_t *)(g.hT2 + (int64_t)x * g.hfs2);
                uint32_t *Hc = Hp + ci * hplane;
                switch (g.hfs2 >> 1) {
#define SWS_HCASE(NP) case NP: hscale_pairs<NP, CHROMA>(p, S + spd, srow_dw, tp, Hc + xl, g.TW, half, nrp >> 1, NT / g.TW); break;
                SWS_HCASE(1) SWS_HCASE(2) SWS_HCASE(3) SWS_HCASE(4) SWS_HCASE(5) SWS_HCASE(6) SWS_HCASE(7) SWS_HCASE(8)
#undef SWS_HCASE
                }
            }
        }
    }
    __syncthreads();
    // ---- phase 3: vertical stage on row pairs + writer; lane = 4 adjacent outputs of one row ----
    const int16_t *vT2 = g.vT2;
    const int32_t *vpos = CHROMA ? p.vChrPos : p.vLumPos;
    const int q4 = (tw + 3) >> 2;
    const int bits = p.dst_bits;
    const int q4s = 31 - __builtin_clz((unsigned)(g.TW >> 2));      // log2(TW / 4): full-width tiles index with shifts
    for (int i = tid; i < (SWS_DBG(g, 2) ? 0 : q4 * th); i += NT) {
        const int yl = tw == g.TW ? i >> q4s : i / q4, xl = 4 * (i - yl * q4);
        const int y = y0 + yl, x = x0 + xl;
        const int n = min(4, tw - xl);
        const int rpd = ((vpos[y] & ~1) - rs) >> 1;              // first row pair
        const uint32_t *vt = (const uint32_t *)(vT2 + (int64_t)y * g.vfs2);
        int acc[NCOMP][4];
#pragma unroll
        for (int ci = 0; ci < NCOMP; ci++)
#pragma unroll
            for (int k = 0; k < 4; k++) acc[ci][k] = 0;
        switch (g.vfs2 >> 1) {
#define SWS_VCASE(NP) case NP: vscale_pairs4<NP, NCOMP>(Hp + rpd * g.TW + xl, hplane, g.TW, vt, acc); break;
        SWS_VCASE(1) SWS_VCASE(2) SWS_VCASE(3) SWS_VCASE(4) SWS_VCASE(5) SWS_VCASE(6) SWS_VCASE(7) SWS_VCASE(8)
#undef SWS_VCASE
        }
        // writers, "X" forms (filter size >= 2): output.c:468-483 (8 bit), :344-357 (9..14 bit), :554-569 / :571-589 (P01x),
        // :495-528 (nv12 chroma)
        if (!CHROMA || (p.dstKind != DSTK_NV12 && p.dstKind != DSTK_P010)) {
#pragma unroll
            for (int ci = 0; ci < NCOMP; ci++) {
                const int pl = !CHROMA ? 0 : ci == 0 ? p.u_plane_dst : p.v_plane_dst;
                uint8_t *drow = (pl == 0 ? f.dst[0] : pl == 1 ? f.dst[1] : f.dst[2]) +
                                (int64_t)y * (pl == 0 ? f.dstStride[0] : pl == 1 ? f.dstStride[1] : f.dstStride[2]);
                if (p.dstKind == DSTK_PLANAR8 || p.dstKind == DSTK_NV12) {
                    const int off = (CHROMA && ci == 1) ? 3 : 0;
                    uint32_t o = 0;
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        o |= (uint32_t)clip_u8_shr((dither8(p.should_dither, y, x + e + off) << 12) + acc[ci][e], 19) << (8 * e);
                    if (n == 4 && !((uintptr_t)(drow + x) & 3)) *(uint32_t *)(drow + x) = o;
                    else for (int e = 0; e < n; e++) drow[x + e] = (uint8_t)(o >> (8 * e));
                } else { // DSTK_PLANARN, or the luma plane of P010
                    const int shift = 11 + 16 - bits, osh = p.dst_shift;   // p010-style and msb planar formats keep the samples in the high bits
                    uint16_t o[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) o[e] = (uint16_t)(clip_uintp2(((1 << (shift - 1)) + acc[ci][e]) >> shift, bits) << osh);
                    uint16_t *d16 = (uint16_t *)drow + x;
                    if (n == 4 && !((uintptr_t)d16 & 7)) { u32x2 v = { (uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16) }; *(u32x2 *)d16 = v; }
                    else for (int e = 0; e < n; e++) d16[e] = o[e];
                }
            }
        } else if constexpr (CHROMA) {
            uint8_t *drow = f.dst[1] + (int64_t)y * f.dstStride[1];
            if (p.dstKind == DSTK_NV12) {
                uint8_t o[8];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int u = clip_u8_shr((dither8(p.should_dither, y, x + e) << 12) + acc[0][e], 19);
                    const int v = clip_u8_shr((dither8(p.should_dither, y, x + e + 3) << 12) + acc[1][e], 19);
                    o[2 * e + p.uv_swap_dst] = (uint8_t)u; o[2 * e + 1 - p.uv_swap_dst] = (uint8_t)v;
                }
                for (int e = 0; e < 2 * n; e++) drow[2 * x + e] = o[e];
            } else { // P010 chroma
                const int shift = 11 + 16 - bits;
                uint16_t *d16 = (uint16_t *)drow + 2 * x;
                uint16_t o[8];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    o[2 * e] = (uint16_t)(clip_uintp2(((1 << (shift - 1)) + acc[0][e]) >> shift, bits) << p.dst_shift);
                    o[2 * e + 1] = (uint16_t)(clip_uintp2(((1 << (shift - 1)) + acc[1][e]) >> shift, bits) << p.dst_shift);
                }
                if (n == 4 && !((uintptr_t)d16 & 15)) {
                    u32x4 v = { (uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16),
                                (uint32_t)o[4] | ((uint32_t)o[5] << 16), (uint32_t)o[6] | ((uint32_t)o[7] << 16) };
                    *(u32x4 *)d16 = v;
                } else for (int e = 0; e < 2 * n; e++) d16[e] = o[e];
            }
        }
    }
}

} // namespace swsk
