// Streaming form of the pure layout / depth converters (planarCopyWrapper incl. DITHER_COPY, planarToNv12 / nv12ToPlanar and the nv24
// twins, yuyv / uyvy <-> planar): the decoder -> filter -> encoder format changes.  Every sample is independent and the work is a
// handful of shifts per 16 bytes, so the only thing that matters is how the bytes move: a lane owns 16-byte chunks spaced one wave apart
// (every load / store instruction of a wave covers a contiguous KiB), plane pointers and strides sit in SGPRs (load_frame), a wave walks down
// eight rows of its chunk column with the loads of four rows issued before the first store, stores are non-temporal.  The element-per-thread kernels of kernels_misc.hpp / kernels_shuffle.hpp
// keep the pictures whose pointers or strides are not 16-byte aligned and the rare converters (yvu9, nv24 -> yuv420p).
// Arithmetic: the same expressions as those kernels (cited there), per element.
#pragma once
#include "kernels_common.hpp"
#include "wave_util.hpp"

namespace swsk {

// dithers[8][8][8] of swscale_unscaled.c:39-112 (DITHER_COPY): [src_depth - dst_depth - 1][row & 7][column & 7], eight bytes per row
__device__ __constant__ const uint8_t k_layout_dithers[8][8][8] __attribute__((aligned(8))) = {
{ {0,1,0,1,0,1,0,1},{1,0,1,0,1,0,1,0},{0,1,0,1,0,1,0,1},{1,0,1,0,1,0,1,0},{0,1,0,1,0,1,0,1},{1,0,1,0,1,0,1,0},{0,1,0,1,0,1,0,1},{1,0,1,0,1,0,1,0} },
{ {1,2,1,2,1,2,1,2},{3,0,3,0,3,0,3,0},{1,2,1,2,1,2,1,2},{3,0,3,0,3,0,3,0},{1,2,1,2,1,2,1,2},{3,0,3,0,3,0,3,0},{1,2,1,2,1,2,1,2},{3,0,3,0,3,0,3,0} },
{ {2,4,3,5,2,4,3,5},{6,0,7,1,6,0,7,1},{3,5,2,4,3,5,2,4},{7,1,6,0,7,1,6,0},{2,4,3,5,2,4,3,5},{6,0,7,1,6,0,7,1},{3,5,2,4,3,5,2,4},{7,1,6,0,7,1,6,0} },
{ {4,8,7,11,4,8,7,11},{12,0,15,3,12,0,15,3},{6,10,5,9,6,10,5,9},{14,2,13,1,14,2,13,1},{4,8,7,11,4,8,7,11},{12,0,15,3,12,0,15,3},{6,10,5,9,6,10,5,9},{14,2,13,1,14,2,13,1} },
{ {9,17,15,23,8,16,14,22},{25,1,31,7,24,0,30,6},{13,21,11,19,12,20,10,18},{29,5,27,3,28,4,26,2},{8,16,14,22,9,17,15,23},{24,0,30,6,25,1,31,7},{12,20,10,18,13,21,11,19},{28,4,26,2,29,5,27,3} },
{ {18,34,30,46,17,33,29,45},{50,2,62,14,49,1,61,13},{26,42,22,38,25,41,21,37},{58,10,54,6,57,9,53,5},{16,32,28,44,19,35,31,47},{48,0,60,12,51,3,63,15},{24,40,20,36,27,43,23,39},{56,8,52,4,59,11,55,7} },
{ {18,34,30,46,17,33,29,45},{50,2,62,14,49,1,61,13},{26,42,22,38,25,41,21,37},{58,10,54,6,57,9,53,5},{16,32,28,44,19,35,31,47},{48,0,60,12,51,3,63,15},{24,40,20,36,27,43,23,39},{56,8,52,4,59,11,55,7} },
{ {36,68,60,92,34,66,58,90},{100,4,124,28,98,2,122,26},{52,84,44,76,50,82,42,74},{116,20,108,12,114,18,106,10},{32,64,56,88,38,70,62,94},{96,0,120,24,102,6,126,30},{48,80,40,72,54,86,46,78},{112,16,104,8,118,22,110,14} },
};

// two 16-bit samples per dword through the packed 16-bit ALU (v_pk_add_u16, v_pk_lshlrev_b16, v_pk_min_u16 ...): half the instructions of the
// per-sample forms.  Every use below keeps its intermediate values inside 16 bits (the host checks the source depth), so the packed results are
// the reference's int arithmetic.
typedef unsigned short pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk16 pk_of(uint32_t w) { return __builtin_bit_cast(pk16, w); }
__device__ __forceinline__ uint32_t pk_bits(pk16 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ pk16 pk_splat(int v) { const pk16 r = { (unsigned short)v, (unsigned short)v }; return r; }
__device__ __forceinline__ pk16 pk_min(pk16 a, pk16 b) { return __builtin_elementwise_min(a, b); }
// bytes {2k, 2k + 1} of a dword as two zero-extended 16-bit values
__device__ __forceinline__ pk16 pk_bytes_lo(uint32_t w) { return pk_of(__builtin_amdgcn_perm(0, w, 0x0c010c00u)); }
__device__ __forceinline__ pk16 pk_bytes_hi(uint32_t w) { return pk_of(__builtin_amdgcn_perm(0, w, 0x0c030c02u)); }

enum { LOP_COPY = 0, LOP_FILL, LOP_IL, LOP_DIL, LOP_8TO16, LOP_16TO8, LOP_16TO16, LOP_P422_SPLIT, LOP_P422_SPLIT420, LOP_P422_JOIN, LOP_P1_16TO8, LOP_P1_16TO16, LOP_DIL16, LOP_P1_8TO8R };

// one class of rows: `rows` rows starting at source row ys / destination row yd of the planes named below
struct LayoutJob {
    int32_t op, rows, ys, yd;
    int32_t sa, sb, da, db;        // source planes A, B (B: second input of an interleave), destination planes A, B
    int32_t n;                     // bytes of the job's widest row side that carry data (what the 16-byte chunks are counted over)
    int32_t a0, a1, a2, a3, a4;    // op parameters
};
struct LayoutPlan { int32_t njobs; LayoutJob job[6]; };

__device__ __forceinline__ void lstore16(uint8_t *d, u32x4 v, int nvalid) { if (nvalid >= 16) gstore16_nt(d, v); else gstore_partial(d, v, nvalid); }
__device__ __forceinline__ void lstore8(uint8_t *d, u32x2 v, int nvalid)
{
    if (nvalid >= 8) __builtin_nontemporal_store(v, (SWS_GLOBAL u32x2 *)d);
    else { const u32x4 t = { v[0], v[1], 0, 0 }; gstore_partial(d, t, nvalid); }
}

// RPW rows per wave: a wave owns one 1 KiB chunk column of a job (64 lanes x 16 bytes of the wider side) and walks down RPW rows of it, RU rows
// at a time: the loads of RU rows are issued before the first store, the per-wave set-up (frame descriptor, job look-up) is paid once per
// RPW KiB instead of once per KiB.  blockIdx.y counts groups of RPW rows over all jobs (LayoutJob::rows rounded up per job).
constexpr int LAYOUT_RPW = 8;

template <int RU>
__global__ void __launch_bounds__(256) sws_k_layout_stream(SwsFrameSet fs, SwsDevParams p, LayoutPlan plan)
{
    const FrameRegs f = load_frame(fs, blockIdx.z);
    const int lane = threadIdx.x & 63;
    const int cx = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // chunk column of this wave
    int g = blockIdx.y, ji = 0;
    while (ji < plan.njobs && g >= (U(plan.job[ji].rows) + LAYOUT_RPW - 1) / LAYOUT_RPW) { g -= (U(plan.job[ji].rows) + LAYOUT_RPW - 1) / LAYOUT_RPW; ji++; }
    if (ji >= plan.njobs) return;
    // (scalar copies of the job: a run-time index into the by-value argument would put it into scratch memory)
    int op = 0, rows = 0, ys = 0, yd = 0, sa = 0, sb = 0, da = 0, db = 0, n = 0, a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
    for (int k = 0; k < 6; k++)
        if (k == ji) {
            const LayoutJob &J = plan.job[k];
            op = U(J.op); rows = U(J.rows); ys = U(J.ys); yd = U(J.yd); sa = U(J.sa); sb = U(J.sb); da = U(J.da); db = U(J.db); n = U(J.n);
            a0 = U(J.a0); a1 = U(J.a1); a2 = U(J.a2); a3 = U(J.a3); a4 = U(J.a4);
        }
    // chunks are counted in 16 bytes of n, or in 8 (the ops whose other side is twice as wide)
    const int unit = (op == LOP_IL || op == LOP_8TO16 || op == LOP_P422_JOIN) ? 8 : 16;
    if (cx * 64 * unit >= n) return;
    const int off = (cx * 64 + lane) * unit;          // this lane's byte offset into the n-byte side of a row
    const bool in = off < n;
    const int r0 = g * LAYOUT_RPW, r1 = min(rows, r0 + LAYOUT_RPW);
    const int64_t ssA = pick4(f.srcStride, sa), ssB = pick4(f.srcStride, sb), dsA = pick4(f.dstStride, da), dsB = pick4(f.dstStride, db);
    const uint8_t *sbaseA = pick4(f.src, sa) + (int64_t)ys * ssA, *sbaseB = pick4(f.src, sb) + (int64_t)ys * ssB;
    uint8_t *dbaseA = pick4(f.dst, da) + (int64_t)yd * dsA, *dbaseB = pick4(f.dst, db) + (int64_t)yd * dsB;

    switch (op) {
    case LOP_COPY: {   // n bytes of a row as they are
        for (int r = r0; r < r1; r += RU) {
            u32x4 v[RU];
#pragma unroll
            for (int i = 0; i < RU; i++) if (in && r + i < r1) v[i] = load16_or_tail(sbaseA + (r + i) * ssA + off, n - off);
#pragma unroll
            for (int i = 0; i < RU; i++) if (in && r + i < r1) lstore16(dbaseA + (r + i) * dsA + off, v[i], n - off);
        }
        break;
    }
    case LOP_FILL: {   // fillPlane / fillPlane16: a0 = the 32-bit pattern
        const u32x4 v = { (uint32_t)a0, (uint32_t)a0, (uint32_t)a0, (uint32_t)a0 };
        for (int r = r0; r < r1; r++) if (in) lstore16(dbaseA + r * dsA + off, v, n - off);
        break;
    }
    case LOP_IL: {     // planarToNv12 / Nv24: n bytes of plane A and of plane B -> 2n interleaved bytes (A first)
        for (int r = r0; r < r1; r += RU) {
            u32x2 a[RU], b[RU];
#pragma unroll
            for (int i = 0; i < RU; i++) if (in && r + i < r1) { a[i] = load8_or_tail(sbaseA + (r + i) * ssA + off, n - off); b[i] = load8_or_tail(sbaseB + (r + i) * ssB + off, n - off); }
#pragma unroll
            for (int i = 0; i < RU; i++) {
                if (!(in && r + i < r1)) continue;
                const u32x4 o = { __builtin_amdgcn_perm(b[i][0], a[i][0], 0x05010400u), __builtin_amdgcn_perm(b[i][0], a[i][0], 0x07030602u),
                                  __builtin_amdgcn_perm(b[i][1], a[i][1], 0x05010400u), __builtin_amdgcn_perm(b[i][1], a[i][1], 0x07030602u) };
                lstore16(dbaseA + (r + i) * dsA + 2 * off, o, 2 * (n - off));
            }
        }
        break;
    }
    case LOP_DIL: {    // nv12ToPlanar / nv24ToPlanar: n interleaved bytes -> n/2 to plane A (even bytes), n/2 to plane B (odd bytes)
        for (int r = r0; r < r1; r += RU) {
            u32x4 v[RU];
#pragma unroll
            for (int i = 0; i < RU; i++) if (in && r + i < r1) v[i] = load16_or_tail(sbaseA + (r + i) * ssA + off, n - off);
#pragma unroll
            for (int i = 0; i < RU; i++) {
                if (!(in && r + i < r1)) continue;
                const u32x2 ea = { __builtin_amdgcn_perm(v[i][1], v[i][0], 0x06040200u), __builtin_amdgcn_perm(v[i][3], v[i][2], 0x06040200u) };
                const u32x2 eb = { __builtin_amdgcn_perm(v[i][1], v[i][0], 0x07050301u), __builtin_amdgcn_perm(v[i][3], v[i][2], 0x07050301u) };
                lstore8(dbaseA + (r + i) * dsA + (off >> 1), ea, (n - off + 1) >> 1); lstore8(dbaseB + (r + i) * dsB + (off >> 1), eb, (n - off) >> 1);
            }
        }
        break;
    }
    case LOP_DIL16: {  // p010ToUV-style split (input.c:950-1008): n bytes of interleaved 16-bit pairs -> n/2 bytes to plane A (first words), n/2 to plane B, every word >> a0
        for (int r = r0; r < r1; r += RU) {
            u32x4 v[RU];
#pragma unroll
            for (int i = 0; i < RU; i++) if (in && r + i < r1) v[i] = load16_or_tail(sbaseA + (r + i) * ssA + off, n - off);
#pragma unroll
            for (int i = 0; i < RU; i++) {
                if (!(in && r + i < r1)) continue;
                const pk16 sh = pk_splat(a0);
                const u32x2 ea = { pk_bits(pk_of(__builtin_amdgcn_perm(v[i][1], v[i][0], 0x05040100u)) >> sh), pk_bits(pk_of(__builtin_amdgcn_perm(v[i][3], v[i][2], 0x05040100u)) >> sh) };
                const u32x2 eb = { pk_bits(pk_of(__builtin_amdgcn_perm(v[i][1], v[i][0], 0x07060302u)) >> sh), pk_bits(pk_of(__builtin_amdgcn_perm(v[i][3], v[i][2], 0x07060302u)) >> sh) };
                lstore8(dbaseA + (r + i) * dsA + (off >> 1), ea, (n - off) >> 1); lstore8(dbaseB + (r + i) * dsB + (off >> 1), eb, (n - off) >> 1);
            }
        }
        break;
    }
    case LOP_8TO16: {  // planarCopy 8 -> 9..16 bit (swscale_unscaled.c:2266-2284): a0 = left shift, a1 = right shift of the replicated bits (32: none), a2 = dst_shift
        for (int r = r0; r < r1; r += RU) {
            u32x2 s8[RU];
#pragma unroll
            for (int i = 0; i < RU; i++) if (in && r + i < r1) s8[i] = load8_or_tail(sbaseA + (r + i) * ssA + off, n - off);
#pragma unroll
            for (int i = 0; i < RU; i++) {
                if (!(in && r + i < r1)) continue;
                uint32_t o[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const pk16 e = (q & 1) ? pk_bytes_hi(s8[i][q >> 1]) : pk_bytes_lo(s8[i][q >> 1]);
                    pk16 v = e << pk_splat(a0);
                    if (a1 < 32) v |= e >> pk_splat(a1);
                    if (a2) v = v << pk_splat(a2);
                    o[q] = pk_bits(v);
                }
                const u32x4 ov = { o[0], o[1], o[2], o[3] };
                lstore16(dbaseA + (r + i) * dsA + 2 * off, ov, 2 * (n - off));
            }
        }
        break;
    }
    case LOP_16TO8:    // DITHER_COPY to 8 bit (:2159-2218): n = source bytes of the row
    case LOP_16TO16: { // DITHER_COPY to 9..15 bit, or the widening copy N -> M (:2285-2331)
        // a0 & 15 = mode: 0 dither off, 1 dither + shiftonly, 2 dither full range, 3 widening shiftonly, 4 widening with bit replication
        // a1 = shift (|sd - dd|), a2 = src_shift, a3 = dst_shift, a4 = dd | body_end << 8 (elements), widening: a4 = 2 * sd - dd
        const int mode = a0 & 15, shift = a1, ss = a2, dsh = a3, dd = a4 & 0xFF, body_end = a4 >> 8;
        const bool packed = (a0 & 16) != 0;   // (the host sets it for sources of at most 15 bits)
        const bool body = (off >> 1) < body_end;
        for (int r = r0; r < r1; r += RU) {
            u32x4 s16[RU];
#pragma unroll
            for (int i = 0; i < RU; i++) if (in && r + i < r1) s16[i] = load16_or_tail(sbaseA + (r + i) * ssA + off, n - off);
#pragma unroll
            for (int i = 0; i < RU; i++) {
                if (!(in && r + i < r1)) continue;
                // the dither row of this picture row: eight bytes, the same for every lane (x & 7 = the element's place in its group of eight)
                uint32_t dlo = 0, dhi = 0;
                if (mode == 1 || mode == 2) {
                    const uint32_t *dr = (const uint32_t *)k_layout_dithers[shift - 1][(r + i) & 7];
                    dlo = U(dr[0]); dhi = U(dr[1]);
                }
                if (packed) {   // source samples of at most 15 bits: sample + dither stays inside 16 bits
                    const pk16 dt[4] = { pk_bytes_lo(dlo), pk_bytes_hi(dlo), pk_bytes_lo(dhi), pk_bytes_hi(dhi) };
                    uint32_t o[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const pk16 raw = pk_of(s16[i][q]);
                        const pk16 sv = body ? raw >> pk_splat(ss) : raw;
                        pk16 v;
                        if (mode == 0) { const pk16 t = (sv + pk_splat(1 << (shift - 1))) >> pk_splat(shift); v = (t - (t >> pk_splat(dd))) << pk_splat(dsh); }
                        else if (mode == 1) { const pk16 t = (sv + dt[q]) >> pk_splat(shift); v = (t - (t >> pk_splat(dd))) << pk_splat(dsh); }
                        else if (mode == 2) { v = (sv - (sv >> pk_splat(dd)) + dt[q]) >> pk_splat(shift); if (body) v = v << pk_splat(dsh); }
                        else if (mode == 3) { v = ((raw >> pk_splat(ss)) << pk_splat(shift)) << pk_splat(dsh); }
                        else { const pk16 t = raw >> pk_splat(ss); v = ((t << pk_splat(shift)) | (t >> pk_splat(a4))) << pk_splat(dsh); }
                        o[q] = pk_bits(v);
                    }
                    if (op == LOP_16TO8) {
                        const u32x2 ob = { __builtin_amdgcn_perm(o[1], o[0], 0x06040200u), __builtin_amdgcn_perm(o[3], o[2], 0x06040200u) };
                        lstore8(dbaseA + (r + i) * dsA + (off >> 1), ob, (n - off) >> 1);
                    } else {
                        const u32x4 ow = { o[0], o[1], o[2], o[3] };
                        lstore16(dbaseA + (r + i) * dsA + off, ow, n - off);
                    }
                    continue;
                }
                uint32_t e[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const uint32_t sv = (s16[i][q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
                    const uint32_t dth = ((q < 4 ? dlo : dhi) >> (8 * (q & 3))) & 0xFFu;
                    uint32_t tmp, v;
                    if (mode == 0) { const uint32_t bias = 1u << (shift - 1); tmp = ((body ? sv >> ss : sv) + bias) >> shift; v = (tmp - (tmp >> dd)) << dsh; }
                    else if (mode == 1) { tmp = ((body ? sv >> ss : sv) + dth) >> shift; v = (tmp - (tmp >> dd)) << dsh; }
                    else if (mode == 2) { tmp = body ? sv >> ss : sv; v = (tmp - (tmp >> dd) + dth) >> shift; if (body) v <<= dsh; }
                    else if (mode == 3) { v = ((sv >> ss) << shift) << dsh; }
                    else { const uint32_t t = sv >> ss; v = ((t << shift) | (t >> (a4))) << dsh; }
                    e[q] = v;
                }
                if (op == LOP_16TO8) {
                    const u32x2 o = { (e[0] & 0xFF) | ((e[1] & 0xFF) << 8) | ((e[2] & 0xFF) << 16) | (e[3] << 24),
                                      (e[4] & 0xFF) | ((e[5] & 0xFF) << 8) | ((e[6] & 0xFF) << 16) | (e[7] << 24) };
                    lstore8(dbaseA + (r + i) * dsA + (off >> 1), o, (n - off) >> 1);
                } else {
                    const u32x4 o = { (e[0] & 0xFFFF) | (e[1] << 16), (e[2] & 0xFFFF) | (e[3] << 16), (e[4] & 0xFFFF) | (e[5] << 16), (e[6] & 0xFFFF) | (e[7] << 16) };
                    lstore16(dbaseA + (r + i) * dsA + off, o, n - off);
                }
            }
        }
        break;
    }
    case LOP_P1_8TO8R: {   // (round 5) the scaler chain with identity filters and a RANGE CONVERSION on an 8-bit plane into an 8-bit plane: hScale8To15_c with the single tap
        // 1 << 14 (s * 128), lumRange{To,From}Jpeg_c ((v * coeff + offset) >> 14, the ToJpeg clip: swscale.c:163-209), yuv2plane1_8_c with the constant dither 64 of 8-bit
        // sources ((v + 64) >> 7, clipped).  a0 = 128 * coeff (23 bits), a1 = offset, a2 = 32767 or INT_MAX
        for (int r = r0; r < r1; r += RU) {
            u32x4 v[RU];
#pragma unroll
            for (int i = 0; i < RU; i++) if (in && r + i < r1) v[i] = load16_or_tail(sbaseA + (r + i) * ssA + off, n - off);
#pragma unroll
            for (int i = 0; i < RU; i++) {
                if (!(in && r + i < r1)) continue;
                u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t e[4];
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const int sv = (int)((v[i][q] >> (8 * b)) & 0xFFu);
                        const int rv = (int)(int16_t)min(mad24(sv, a0, a1) >> 14, a2);
                        e[b] = (uint32_t)clip_u8_shr(rv + 64, 7);     // (clamp, then shift: hipcc 7.2 turns sat_u8(x >> n) pairs into v_ashr_pk_u8_i32 with stale upper bits)
                    }
                    o[q] = e[0] | e[1] << 8 | e[2] << 16 | e[3] << 24;
                }
                lstore16(dbaseA + (r + i) * dsA + off, o, n - off);
            }
        }
        break;
    }
    case LOP_P1_16TO8:     // the scaler chain with identity filters on a plane of 9..16-bit samples: hScale16To15_c with the single tap 1 << 14
    case LOP_P1_16TO16: {  // (swscale.c:99-125), then yuv2plane1_8_c with the row's ff_dither_8x8_128 line (output.c:485-493, swscale.c:519-522) or
        // yuv2plane1_10 / 12 / 14 (output.c:327-341).  a0 = src_shift, a1 = hscale shift, a2 = destination bits, a3 = dst_shift; rows are absolute
        for (int r = r0; r < r1; r += RU) {
            u32x4 s16[RU];
#pragma unroll
            for (int i = 0; i < RU; i++) if (in && r + i < r1) s16[i] = load16_or_tail(sbaseA + (r + i) * ssA + off, n - off);
#pragma unroll
            for (int i = 0; i < RU; i++) {
                if (!(in && r + i < r1)) continue;
                // (the row's dither line: eight bytes, the same for every lane)
                const uint32_t *dr = (const uint32_t *)k_dither_8x8_128[(yd + r + i) & 7];
                const uint32_t dlo = a4 ? 0x40404040u : U(dr[0]), dhi = a4 ? 0x40404040u : U(dr[1]);   // (a4: no ordered dither -- the planar working picture behind a packed 4:2:2 destination)
                const pk16 dt[4] = { pk_bytes_lo(dlo), pk_bytes_hi(dlo), pk_bytes_lo(dhi), pk_bytes_hi(dhi) };
                // min((sv << 14) >> a1, 32767) inside 16 bits: a sample of 2^(15 - L) or more saturates (L = 14 - a1), below that sv << L fits
                const int L = 14 - a1;
                uint32_t o[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const pk16 sv = pk_of(s16[i][q]) >> pk_splat(a0);
                    const pk16 hv = L >= 0 ? pk_min(pk_min(sv, pk_splat(1 << (15 - L))) << pk_splat(L), pk_splat(32767)) : pk_min(sv >> pk_splat(-L), pk_splat(32767));
                    pk16 e;
                    if (op == LOP_P1_16TO8) e = pk_min((hv + dt[q]) >> pk_splat(7), pk_splat(255));
                    else { const int sh = 15 - a2; e = pk_min((hv + pk_splat(1 << (sh - 1))) >> pk_splat(sh), pk_splat((1 << a2) - 1)) << pk_splat(a3); }
                    o[q] = pk_bits(e);
                }
                if (op == LOP_P1_16TO8) {
                    const u32x2 ob = { __builtin_amdgcn_perm(o[1], o[0], 0x06040200u), __builtin_amdgcn_perm(o[3], o[2], 0x06040200u) };
                    lstore8(dbaseA + (r + i) * dsA + (off >> 1), ob, (n - off) >> 1);
                } else {
                    const u32x4 ow = { o[0], o[1], o[2], o[3] };
                    lstore16(dbaseA + (r + i) * dsA + off, ow, n - off);
                }
            }
        }
        break;
    }
    case LOP_P422_SPLIT: {   // yuyvtoyuv422_c / uyvytoyuv422_c (rgb2rgb_template.c:751-825): n = packed bytes that carry whole pairs; a0 = 1 for uyvy, a1 = swap U / V (yvyu)
        // a2 = luma samples of the row (an odd width: the last pair's second luma byte is not stored)
        const uint32_t ysel = a0 ? 0x07050301u : 0x06040200u;
        // chroma bytes of a dword pair: yuyv: U at 1, 5; V at 3, 7.  uyvy: U at 0, 4; V at 2, 6
        const uint32_t usel = a0 ? 0x0c0c0400u : 0x0c0c0501u, vsel = a0 ? 0x0c0c0602u : 0x0c0c0703u;
        const int upl = a1 ? 2 : 1, vpl = 3 - upl;
        const int64_t dsU = pick4(f.dstStride, upl), dsV = pick4(f.dstStride, vpl);
        uint8_t *ubase = pick4(f.dst, upl) + (int64_t)yd * dsU, *vbase = pick4(f.dst, vpl) + (int64_t)yd * dsV;
        for (int r = r0; r < r1; r += RU) {
            u32x4 v[RU];
#pragma unroll
            for (int i = 0; i < RU; i++) if (in && r + i < r1) v[i] = load16_or_tail(sbaseA + (r + i) * ssA + off, n - off);
#pragma unroll
            for (int i = 0; i < RU; i++) {
                if (!(in && r + i < r1)) continue;
                const u32x2 yv = { __builtin_amdgcn_perm(v[i][1], v[i][0], ysel), __builtin_amdgcn_perm(v[i][3], v[i][2], ysel) };
                const uint32_t u = __builtin_amdgcn_perm(v[i][1], v[i][0], usel) | (__builtin_amdgcn_perm(v[i][3], v[i][2], usel) << 16);
                const uint32_t w = __builtin_amdgcn_perm(v[i][1], v[i][0], vsel) | (__builtin_amdgcn_perm(v[i][3], v[i][2], vsel) << 16);
                lstore8(dbaseA + (r + i) * dsA + (off >> 1), yv, min(8, a2 - (off >> 1)));
                const int nc = (n - off) >> 2;   // chroma samples behind this chunk
                uint8_t *pu = ubase + (r + i) * dsU + (off >> 2), *pv = vbase + (r + i) * dsV + (off >> 2);
                if (nc >= 4) { *(uint32_t *)pu = u; *(uint32_t *)pv = w; }
                else for (int b = 0; b < nc; b++) { pu[b] = (uint8_t)(u >> (8 * b)); pv[b] = (uint8_t)(w >> (8 * b)); }
            }
        }
        break;
    }
    case LOP_P422_SPLIT420: {   // yuyvtoyuv420_c / uyvytoyuv420_c: a ROW PAIR per job row (luma of both rows, chroma = truncating mean of the two);
        // a3 = rows of the slice (an odd count: the last row has luma only); ys / yd count luma rows; chroma planes 1 / 2
        const uint32_t ysel = a0 ? 0x07050301u : 0x06040200u;
        const uint32_t usel = a0 ? 0x0c0c0400u : 0x0c0c0501u, vsel = a0 ? 0x0c0c0602u : 0x0c0c0703u;
        const int upl = a1 ? 2 : 1, vpl = 3 - upl;
        const int64_t dsU = pick4(f.dstStride, upl), dsV = pick4(f.dstStride, vpl);
        uint8_t *ubase = pick4(f.dst, upl) + (int64_t)(yd >> 1) * dsU, *vbase = pick4(f.dst, vpl) + (int64_t)(yd >> 1) * dsV;
        constexpr int PU = RU > 2 ? 2 : RU;   // row pairs in flight
        for (int r = r0; r < r1; r += PU) {
            u32x4 v0[PU], v1[PU];
#pragma unroll
            for (int i = 0; i < PU; i++) {
                if (!(in && r + i < r1)) continue;
                const int y0 = 2 * (r + i);
                v0[i] = load16_or_tail(sbaseA + y0 * ssA + off, n - off);
                v1[i] = (y0 + 1 < a3) ? load16_or_tail(sbaseA + (y0 + 1) * ssA + off, n - off) : v0[i];
            }
#pragma unroll
            for (int i = 0; i < PU; i++) {
                if (!(in && r + i < r1)) continue;
                const int y0 = 2 * (r + i);
                const bool two = y0 + 1 < a3;
                const u32x2 ya = { __builtin_amdgcn_perm(v0[i][1], v0[i][0], ysel), __builtin_amdgcn_perm(v0[i][3], v0[i][2], ysel) };
                const u32x2 yb = { __builtin_amdgcn_perm(v1[i][1], v1[i][0], ysel), __builtin_amdgcn_perm(v1[i][3], v1[i][2], ysel) };
                const int nl = min(8, a2 - (off >> 1));
                lstore8(dbaseA + y0 * dsA + (off >> 1), ya, nl);
                if (two) {
                    lstore8(dbaseA + (y0 + 1) * dsA + (off >> 1), yb, nl);
                    // per-byte truncating mean (a + b) >> 1 = (a & b) + ((a ^ b) >> 1) on the chroma bytes
                    uint32_t m[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) m[q] = (v0[i][q] & v1[i][q]) + (((v0[i][q] ^ v1[i][q]) >> 1) & 0x7F7F7F7Fu);
                    const uint32_t u = __builtin_amdgcn_perm(m[1], m[0], usel) | (__builtin_amdgcn_perm(m[3], m[2], usel) << 16);
                    const uint32_t w = __builtin_amdgcn_perm(m[1], m[0], vsel) | (__builtin_amdgcn_perm(m[3], m[2], vsel) << 16);
                    const int nc = (n - off) >> 2;
                    uint8_t *pu = ubase + (r + i) * dsU + (off >> 2), *pv = vbase + (r + i) * dsV + (off >> 2);
                    if (nc >= 4) { *(uint32_t *)pu = u; *(uint32_t *)pv = w; }
                    else for (int b = 0; b < nc; b++) { pu[b] = (uint8_t)(u >> (8 * b)); pv[b] = (uint8_t)(w >> (8 * b)); }
                }
            }
        }
        break;
    }
    case LOP_P422_JOIN: {   // yuvPlanartoyuy2_c / yuvPlanartouyvy_c (rgb2rgb_template.c:379-470): n = luma bytes that are in whole pairs; a0 = 1 for uyvy;
        // a1 = luma rows per chroma row (1: 4:2:2 source, 2: 4:2:0), a2 = first chroma row
        const int64_t ssU = f.srcStride[1], ssV = f.srcStride[2];
        for (int r = r0; r < r1; r += RU) {
            u32x2 yv[RU]; uint32_t u[RU], w[RU];
#pragma unroll
            for (int i = 0; i < RU; i++) {
                if (!(in && r + i < r1)) continue;
                const int cr = a2 + (a1 == 2 ? ((r + i) >> 1) : (r + i));
                const uint8_t *su = f.src[1] + cr * ssU + (off >> 1), *sv = f.src[2] + cr * ssV + (off >> 1);
                yv[i] = load8_or_tail(sbaseA + (r + i) * ssA + off, n - off);
                if (n - off >= 8) { u[i] = *(const uint32_t *)su; w[i] = *(const uint32_t *)sv; }
                else { u[i] = w[i] = 0; for (int b = 0; b < ((n - off) >> 1); b++) { u[i] |= (uint32_t)su[b] << (8 * b); w[i] |= (uint32_t)sv[b] << (8 * b); } }
            }
#pragma unroll
            for (int i = 0; i < RU; i++) {
                if (!(in && r + i < r1)) continue;
                // chroma pairs: c01 = U0 V0 U1 V1, c23 = U2 V2 U3 V3
                const uint32_t c01 = __builtin_amdgcn_perm(w[i], u[i], 0x05010400u), c23 = __builtin_amdgcn_perm(w[i], u[i], 0x07030602u);
                u32x4 o;
                if (!a0) {   // Y0 U Y1 V
                    o[0] = __builtin_amdgcn_perm(c01, yv[i][0], 0x05010400u); o[1] = __builtin_amdgcn_perm(c01, yv[i][0], 0x07030602u);
                    o[2] = __builtin_amdgcn_perm(c23, yv[i][1], 0x05010400u); o[3] = __builtin_amdgcn_perm(c23, yv[i][1], 0x07030602u);
                } else {     // U Y0 V Y1
                    o[0] = __builtin_amdgcn_perm(yv[i][0], c01, 0x05010400u); o[1] = __builtin_amdgcn_perm(yv[i][0], c01, 0x07030602u);
                    o[2] = __builtin_amdgcn_perm(yv[i][1], c23, 0x05010400u); o[3] = __builtin_amdgcn_perm(yv[i][1], c23, 0x07030602u);
                }
                lstore16(dbaseA + (r + i) * dsA + 2 * off, o, 2 * (n - off));
            }
        }
        break;
    }
    }
}

} // namespace swsk
