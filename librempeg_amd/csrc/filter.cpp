// Polyphase tap generation on the host: produces exactly the int16 taps / int32 start
// positions the reference's initFilter() produces (libswscale/utils.c:197-612), because the
// GPU kernels consume the tables and the output must be bit-identical.  The GPU never
// computes taps.  Fixed-point steps are kept in int64 as in the reference; Lanczos / Gauss /
// sinc / spline weights go through libm in double, like the reference.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "swsint.hpp"

namespace swship {

namespace {

inline int64_t iabs64(int64_t v) { return v < 0 ? -v : v; }
inline int64_t rdiv(int64_t a, int64_t b) { return (a >= 0 ? a + (b >> 1) : a - (b >> 1)) / b; } // ROUNDED_DIV
inline int floor_log2(unsigned v) { int n = 0; v |= 1; while (v >>= 1) ++n; return n; }

double spline_piece(double a, double b, double c, double d, double dist) // utils.c:152-166
{
    while (dist > 1.0) {
        const double nb = b + 2.0 * c + 3.0 * d, nc = c + 3.0 * d, nd = -b - 3.0 * c - 6.0 * d;
        a = 0.0; b = nb; c = nc; d = nd; dist -= 1.0;
        // the reference recurses with (0, b+2c+3d, c+3d, -b-3c-6d); note the new d uses the OLD b,c,d
    }
    return ((d * dist + c) * dist + b) * dist + a;
}

// weight of one tap at distance d (Q30, already scaled for down-scaling) for the "general" scalers
struct TapWeight {
    int scaler; const double *param; int64_t fone; int xInc;
    int64_t operator()(int64_t d) const
    {
        const double fd = d * (1.0 / (1 << 30));
        switch (scaler) {
        case SWS_BICUBIC: { // utils.c:312-332
            const int64_t B = (int64_t)((param[0] != SWS_PARAM_DEFAULT ? param[0] : 0) * (1 << 24));
            const int64_t C = (int64_t)((param[1] != SWS_PARAM_DEFAULT ? param[1] : 0.6) * (1 << 24));
            int64_t w = 0;
            if (d < (1LL << 31)) {
                const int64_t dd = (d * d) >> 30, ddd = (dd * d) >> 30;
                if (d < (1LL << 30))
                    w = (12 * (1 << 24) - 9 * B - 6 * C) * ddd + (-18 * (1 << 24) + 12 * B + 6 * C) * dd +
                        (6 * (1 << 24) - 2 * B) * (1 << 30);
                else
                    w = (-B - 6 * C) * ddd + (6 * B + 30 * C) * dd + (-12 * B - 48 * C) * d +
                        (8 * B + 24 * C) * (1 << 30);
            }
            return w / ((1LL << 54) / fone);
        }
        case SWS_BILINEAR: { // :366-370
            int64_t w = (1 << 30) - d;
            if (w < 0) w = 0;
            return w * (fone >> 30);
        }
        case SWS_LANCZOS: { // :360-365
            const double p = param[0] != SWS_PARAM_DEFAULT ? param[0] : 3.0;
            int64_t w = (int64_t)((d ? std::sin(fd * M_PI) * std::sin(fd * M_PI / p) / (fd * fd * M_PI * M_PI / p) : 1.0) * fone);
            if (fd > p) w = 0;
            return w;
        }
        case SWS_AREA: { // :346-354 (down-scaling form)
            const int64_t d2 = d - (1 << 29);
            int64_t w;
            if (d2 * xInc < -(1LL << (29 + 16))) w = 1LL << (30 + 16);
            else if (d2 * xInc < (1LL << (29 + 16))) w = -d2 * xInc + (1LL << (29 + 16));
            else w = 0;
            return w * (fone >> (30 + 16));
        }
        case SWS_GAUSS: { const double p = param[0] != SWS_PARAM_DEFAULT ? param[0] : 3.0;
                          return (int64_t)(std::exp2(-p * fd * fd) * fone); }
        case SWS_SINC: return (int64_t)((d ? std::sin(fd * M_PI) / (fd * M_PI) : 1.0) * fone);
        case SWS_SPLINE: { const double p = -2.196152422706632;
                           return (int64_t)(spline_piece(1.0, 0.0, p, -p - 1.0, fd) * fone); }
        case SWS_X: { const double A = param[0] != SWS_PARAM_DEFAULT ? param[0] : 1.0;
                      double c = fd < 1.0 ? std::cos(fd * M_PI) : -1.0;
                      c = c < 0.0 ? -std::pow(-c, A) : std::pow(c, A);
                      return (int64_t)((c * 0.5 + 0.5) * fone); }
        }
        return 0;
    }
};

int support_factor(int scaler, const double *param) // utils.c:183-195, :278-279
{
    switch (scaler) {
    case SWS_AREA: return 1;
    case SWS_BICUBIC: return 4;
    case SWS_BILINEAR: return 2;
    case SWS_GAUSS: return 8;
    case SWS_SINC: return 20;
    case SWS_SPLINE: return 20;
    case SWS_X: return 8;
    case SWS_LANCZOS: return param[0] != SWS_PARAM_DEFAULT ? (int)std::ceil(2 * param[0]) : 6;
    }
    return -1;
}

} // namespace

int build_filter_bank(FilterBank &out, int xInc, int srcW, int dstW, int filterAlign, int one,
                      int scaler, int flags, const double param[2], int srcPos, int dstPos, const FilterVec &vec)
{
    const int64_t fone = 1LL << (54 - std::min(floor_log2((unsigned)(srcW / dstW)), 8));
    std::vector<int64_t> raw;  // dstW x rawSize, 54-ish bit fixed point
    std::vector<int32_t> pos((size_t)dstW + 3);
    int rawSize;

    // ---- stage 1: raw weights ----
    if (std::abs(xInc - 0x10000) < 10 && srcPos == dstPos) {           // unity (:219-228)
        rawSize = 1;
        raw.assign(dstW, fone);
        for (int i = 0; i < dstW; i++) pos[i] = i;
    } else if (scaler == SWS_POINT) {                                    // :229-243
        rawSize = 1;
        raw.assign(dstW, fone);
        int64_t x = ((dstPos * (int64_t)xInc) >> 8) - ((srcPos * 0x8000LL) >> 7);
        for (int i = 0; i < dstW; i++, x += xInc) pos[i] = (int)((x + (1 << 15)) >> 16);
    } else if ((xInc <= (1 << 16) && scaler == SWS_AREA) || scaler == SWS_FAST_BILINEAR) { // :244-267
        rawSize = 2;
        raw.resize((size_t)dstW * 2);
        int64_t x = ((dstPos * (int64_t)xInc) >> 8) - ((srcPos * 0x8000LL) >> 7);
        for (int i = 0; i < dstW; i++, x += xInc) {
            int xx = (int)((x - (1 << 15) + (1 << 15)) >> 16);
            pos[i] = xx;
            for (int j = 0; j < 2; j++, xx++) {
                int64_t w = fone - iabs64((int64_t)xx * (1 << 16) - x) * (fone >> 16);
                raw[(size_t)i * 2 + j] = w < 0 ? 0 : w;
            }
        }
    } else {                                                             // general (:268-383)
        const int sf = support_factor(scaler, param);
        if (sf <= 0 || sf > 50) return FILTER_ERR;
        rawSize = xInc <= (1 << 16) ? 1 + sf : 1 + (sf * srcW + dstW - 1) / dstW;
        rawSize = std::max(std::min(rawSize, srcW - 2), 1);
        raw.resize((size_t)dstW * rawSize);
        const TapWeight weight{scaler, param, fone, xInc};
        int64_t x = ((dstPos * (int64_t)xInc) >> 7) - ((srcPos * 0x10000LL) >> 7);
        for (int i = 0; i < dstW; i++, x += 2LL * xInc) {
            int xx = (int)((x - (rawSize - 2) * (1LL << 16)) / (1 << 17));
            pos[i] = xx;
            for (int j = 0; j < rawSize; j++, xx++) {
                int64_t d = iabs64((int64_t)xx * (1 << 17) - x) << 13;
                if (xInc > (1 << 16)) d = d * dstW / srcW;
                raw[(size_t)i * rawSize + j] = weight(d);
            }
        }
    }

    // ---- stage 1b: SwsFilter vectors (:385-415).  The source vector is convolved into every row (double products
    // accumulated into the int64 taps, as the C expression does); the destination vector only widens the rows
    // ("FIXME dstFilter" in the reference); positions are re-centred. ----
    if ((vec.coeff && vec.length > 0) || vec.dst_length > 0) {
        int size2 = rawSize;
        if (vec.coeff) size2 += vec.length - 1;
        if (vec.dst_length) size2 += vec.dst_length - 1;
        std::vector<int64_t> raw2((size_t)dstW * size2, 0);
        for (int i = 0; i < dstW; i++) {
            if (vec.coeff) {
                for (int k = 0; k < vec.length; k++)
                    for (int j = 0; j < rawSize; j++) {
                        int64_t &t = raw2[(size_t)i * size2 + k + j];
                        t = (int64_t)((double)t + vec.coeff[k] * (double)raw[(size_t)i * rawSize + j]);
                    }
            } else {
                for (int j = 0; j < rawSize; j++) raw2[(size_t)i * size2 + j] = raw[(size_t)i * rawSize + j];
            }
            pos[i] += (rawSize - 1) / 2 - (size2 - 1) / 2;
        }
        raw.swap(raw2);
        rawSize = size2;
    }

    // ---- stage 2: trim near-zero taps (:417-457).  Leading taps are shifted out while the row
    // stays monotone in position; trailing near-zeros only shrink the common size. ----
    int needed = 0;
    const double cutoff = SWS_MAX_REDUCE_CUTOFF * (double)fone;
    for (int i = dstW - 1; i >= 0; i--) {
        int64_t *row = &raw[(size_t)i * rawSize];
        int64_t acc = 0;
        for (int j = 0; j < rawSize; j++) {
            acc += iabs64(row[0]);
            if ((double)acc > cutoff) break;
            if (i < dstW - 1 && pos[i] >= pos[i + 1]) break;
            std::memmove(row, row + 1, sizeof(int64_t) * (rawSize - 1));
            row[rawSize - 1] = 0;
            pos[i]++;
        }
        int keep = rawSize;
        acc = 0;
        for (int j = rawSize - 1; j > 0; j--) {
            acc += iabs64(row[j]);
            if ((double)acc > cutoff) break;
            keep--;
        }
        needed = std::max(needed, keep);
    }
    const int size = (needed + (filterAlign - 1)) & ~(filterAlign - 1);
    // extreme ratios are handled by a two-step cascade in the reference (:492-496; APCK_SIZE == 16)
    if (size >= 256 * 16 / 16) return FILTER_USE_CASCADE;

    std::vector<int64_t> f((size_t)dstW * size);
    for (int i = 0; i < dstW; i++)
        for (int j = 0; j < size; j++) {
            int64_t v = j < rawSize ? raw[(size_t)i * rawSize + j] : 0;
            if ((flags & SWS_BITEXACT) && j >= needed) v = 0;            // :512-513
            f[(size_t)i * size + j] = v;
        }

    // ---- stage 3: fold taps that fall outside [0, srcW) onto the edge samples (:519-560) ----
    for (int i = 0; i < dstW; i++) {
        int64_t *row = &f[(size_t)i * size];
        if (pos[i] < 0) {
            for (int j = 1; j < size; j++) {
                const int left = std::max(j + pos[i], 0);
                row[left] += row[j];
                row[j] = 0;
            }
            pos[i] = 0;
        }
        if (pos[i] + size > srcW) {
            const int shift = pos[i] + std::min(size - srcW, 0);
            int64_t acc = 0;
            for (int j = size - 1; j >= 0; j--)
                if (pos[i] + j >= srcW) { acc += row[j]; row[j] = 0; }
            for (int j = size - 1; j >= 0; j--)
                row[j] = j < shift ? 0 : row[j - shift];
            pos[i] -= shift;
            row[srcW - 1 - pos[i]] += acc;
        }
    }

    // ---- stage 4: normalise each row to `one` with running error feedback (:568-588) ----
    out.size = size;
    out.count = dstW;
    out.taps.assign((size_t)(dstW + 3) * size, 0);
    for (int i = 0; i < dstW; i++) {
        const int64_t *row = &f[(size_t)i * size];
        int64_t sum = 0, err = 0;
        for (int j = 0; j < size; j++) sum += row[j];
        sum = (sum + one / 2) / one;
        if (!sum) sum = 1;
        for (int j = 0; j < size; j++) {
            const int64_t v = row[j] + err;
            const int q = (int)rdiv(v, sum);
            out.taps[(size_t)i * size + j] = (int16_t)q;
            err = v - q * sum;
        }
    }
    // the reference replicates the last row/position 3 times for SIMD over-read (:590-599)
    for (int k = 0; k < 3; k++) {
        pos[dstW + k] = pos[dstW - 1];
        for (int j = 0; j < size; j++)
            out.taps[(size_t)(dstW + k) * size + j] = out.taps[(size_t)(dstW - 1) * size + j];
    }
    out.pos = std::move(pos);
    return FILTER_OK;
}

} // namespace swship
