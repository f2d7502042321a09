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
#include "kernels_generic.hpp"

namespace swsk {

template <typename HT> struct LdsSampler {
    const HT *h[3]; int r0, x0, tw;
    __device__ __forceinline__ int get(int comp, int row, int x) const { return h[comp][(row - r0) * tw + (x - x0)]; }
};

template <typename HT, bool CHROMA>
__global__ void __launch_bounds__(256) sws_k_tile_planar(SwsFrameSet fs, SwsDevParams p, SwsTileGeom g)
{
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
        for (int i = tid; i < nr * nc; i += 256) {
            const int r = i / nc, cc = i - r * nc;
            S[r * g.NCmax + cc] = (uint16_t)read_sample(p, f, comp, min(r0 + r, sH - 1), min(c0 + cc, sW - 1));
        }
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
