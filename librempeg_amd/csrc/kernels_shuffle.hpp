// Packed RGB byte shuffles and packed copies: rgbToRgbWrapper / findRgbConvFn (swscale_unscaled.c:1843-2060, with the
// C converters of rgb2rgb.c / rgb2rgb_template.c) and packedCopyWrapper (:2138-2157) for the 8-bit 24/32 bpp formats.
// Pure HBM streaming: one thread moves 4 pixels (12 or 16 bytes in, 12 or 16 bytes out) with v_perm_b32.
#pragma once
#include "kernels_common.hpp"

namespace swsk {

struct ShufflePlan {
    uint32_t sel[4];     // v_perm_b32 selector of pixel i of a 4-pixel group: destination byte j <- byte sel[j] of the
                         // source dword pair that holds the pixel (0x0d.. = constant 0xff, 0x0c = 0)
    int32_t src_step, dst_step;         // 3 or 4 bytes per pixel
    int32_t spos[4], dpos[4];           // byte offsets of R,G,B,A in a source / destination pixel (A: -1 = none)
    int32_t opaque;                     // write 255 to destination alpha even though the source has an alpha byte
};

struct Tri { uint32_t a, b, c; };

__device__ __forceinline__ void shuffle_px_bytes(const ShufflePlan &sp, const uint8_t *s, uint8_t *d)
{
#pragma unroll
    for (int k = 0; k < 3; k++) d[sp.dpos[k]] = s[sp.spos[k]];
    if (sp.dpos[3] >= 0) d[sp.dpos[3]] = (sp.spos[3] >= 0 && !sp.opaque) ? s[sp.spos[3]] : 255;
}

template <bool S3, bool D3>
__global__ void __launch_bounds__(256) sws_k_rgb_shuffle(SwsFrameSet fs, ShufflePlan sp, int w, int sliceY)
{
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= w) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y = blockIdx.y;
    const uint8_t *srow = f.src[0] + (int64_t)y * f.srcStride[0];
    uint8_t *drow = f.dst[0] + (int64_t)(sliceY + y) * f.dstStride[0];
    constexpr int SS = S3 ? 3 : 4, DS = D3 ? 3 : 4;
    const uint8_t *s = srow + x0 * SS;
    uint8_t *d = drow + x0 * DS;
    const bool aligned = ((((uintptr_t)srow) & (S3 ? 3 : 15)) | (((uintptr_t)drow) & (D3 ? 3 : 15))) == 0; // wave-uniform
    if (!aligned || x0 + 4 > w) {
        const int n = min(4, w - x0);
        for (int i = 0; i < n; i++) shuffle_px_bytes(sp, s + i * SS, d + i * DS);
        return;
    }
    uint32_t px[4];
    if (S3) {
        const Tri t = *reinterpret_cast<const Tri *>(s);
        px[0] = __builtin_amdgcn_perm(t.a, t.a, sp.sel[0]);
        px[1] = __builtin_amdgcn_perm(t.b, t.a, sp.sel[1]);
        px[2] = __builtin_amdgcn_perm(t.c, t.b, sp.sel[2]);
        px[3] = __builtin_amdgcn_perm(t.c, t.c, sp.sel[3]);
    } else {
        const uint4 q = *reinterpret_cast<const uint4 *>(s);
        px[0] = __builtin_amdgcn_perm(q.x, q.x, sp.sel[0]);
        px[1] = __builtin_amdgcn_perm(q.y, q.y, sp.sel[1]);
        px[2] = __builtin_amdgcn_perm(q.z, q.z, sp.sel[2]);
        px[3] = __builtin_amdgcn_perm(q.w, q.w, sp.sel[3]);
    }
    if (D3) {
        Tri o;
        o.a = __builtin_amdgcn_perm(px[1], px[0], 0x04020100u);
        o.b = __builtin_amdgcn_perm(px[2], px[1], 0x05040201u);
        o.c = __builtin_amdgcn_perm(px[3], px[2], 0x06050402u);
        *reinterpret_cast<Tri *>(d) = o;
    } else {
        *reinterpret_cast<uint4 *>(d) = make_uint4(px[0], px[1], px[2], px[3]);
    }
}

// packedCopyWrapper: visible bytes of every row; `alpha_pos` >= 0 forces that byte of every 4-byte pixel to 255
// (the rgb0 -> rgba "scratch copy" of swscale.c:1106-1124 folded into the copy)
__global__ void __launch_bounds__(256) sws_k_packed_copy(SwsFrameSet fs, int row_bytes, int sliceY, int alpha_pos)
{
    const int b0 = (blockIdx.x * 256 + threadIdx.x) * 16;
    if (b0 >= row_bytes) return;
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int y = blockIdx.y;
    const uint8_t *srow = f.src[0] + (int64_t)y * f.srcStride[0];
    uint8_t *drow = f.dst[0] + (int64_t)(sliceY + y) * f.dstStride[0];
    const bool aligned = ((((uintptr_t)srow) | ((uintptr_t)drow)) & 15) == 0;
    if (aligned && b0 + 16 <= row_bytes) {
        uint4 q = *reinterpret_cast<const uint4 *>(srow + b0);
        if (alpha_pos >= 0) {
            const uint32_t m = 0xffu << (8 * alpha_pos);
            q.x |= m; q.y |= m; q.z |= m; q.w |= m;
        }
        *reinterpret_cast<uint4 *>(drow + b0) = q;
        return;
    }
    const int n = min(16, row_bytes - b0);
    for (int i = 0; i < n; i++) {
        const int b = b0 + i;
        drow[b] = (alpha_pos >= 0 && (b & 3) == alpha_pos) ? 255 : srow[b];
    }
}

} // namespace swsk
