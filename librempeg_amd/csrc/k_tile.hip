// Translation unit of the LDS-tile kernels (C1 and the narrow pictures of the same family; 16-bit outputs).
#include "generic_kinds.hpp"
#include "kernels_tile.hpp"

namespace swship {

int launch_tile_dot2(const LaunchCtx &L)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const SwsFramePtrs *frames = L.frames; const int n = L.n, sliceY = L.sliceY, sliceH = L.sliceH; const bool vec = L.vec;
    const dim3 blk(256);
    (void)c; (void)d; (void)frames; (void)vec; (void)sliceY; (void)sliceH;
            const dim3 gl(d->dotL.tilesX, d->dotL.tilesY, n), gc(d->dotC.tilesX, d->dotC.tilesY, n);
            const int NT = c->tune.tile_threads;
            const bool s16 = p.srcKind == SRCK_PLANAR16;
#define LAUNCH_T(N) do { const dim3 b(N); \
    if (s16) { hipLaunchKernelGGL((swsk::sws_k_tile_dot2<true, false, N>), gl, b, d->dotL.lds_bytes, st, fs, p, d->dotL); \
               hipLaunchKernelGGL((swsk::sws_k_tile_dot2<true, true, N>), gc, b, d->dotC.lds_bytes, st, fs, p, d->dotC); } \
    else     { hipLaunchKernelGGL((swsk::sws_k_tile_dot2<false, false, N>), gl, b, d->dotL.lds_bytes, st, fs, p, d->dotL); \
               hipLaunchKernelGGL((swsk::sws_k_tile_dot2<false, true, N>), gc, b, d->dotC.lds_bytes, st, fs, p, d->dotC); } } while (0)
            if (NT == 512) LAUNCH_T(512); else if (NT == 1024) LAUNCH_T(1024); else LAUNCH_T(256);
#undef LAUNCH_T
    return 0;
}

int launch_tile(const LaunchCtx &L)
{
    SwsInternal *c = L.c; DeviceState *d = L.d; const SwsDevParams &p = *L.p; hipStream_t st = L.st; const SwsFrameSet &fs = L.fs;
    const SwsFramePtrs *frames = L.frames; const int n = L.n, sliceY = L.sliceY, sliceH = L.sliceH; const bool vec = L.vec;
    const dim3 blk(256);
    (void)c; (void)d; (void)frames; (void)vec; (void)sliceY; (void)sliceH;
            const dim3 gl(d->tileL.tilesX, d->tileL.tilesY, n), gc(d->tileC.tilesX, d->tileC.tilesY, n);
            const GenericKindFns *ks = c->tune.no_generic_kinds ? nullptr : generic_kind_fns(p.srcKind);   // the instantiations for this source kind (k_generic_kinds.hip)
            if (ks) {
                hipLaunchKernelGGL(ks->tile[p.wide ? 1 : 0][0], gl, blk, d->tileL.lds_bytes, st, fs, p, d->tileL);
                hipLaunchKernelGGL(ks->tile[p.wide ? 1 : 0][1], gc, blk, d->tileC.lds_bytes, st, fs, p, d->tileC);
            } else if (p.wide) {
                hipLaunchKernelGGL((swsk::sws_k_tile_planar<int32_t, false>), gl, blk, d->tileL.lds_bytes, st, fs, p, d->tileL);
                hipLaunchKernelGGL((swsk::sws_k_tile_planar<int32_t, true>), gc, blk, d->tileC.lds_bytes, st, fs, p, d->tileC);
            } else {
                hipLaunchKernelGGL((swsk::sws_k_tile_planar<int16_t, false>), gl, blk, d->tileL.lds_bytes, st, fs, p, d->tileL);
                hipLaunchKernelGGL((swsk::sws_k_tile_planar<int16_t, true>), gc, blk, d->tileC.lds_bytes, st, fs, p, d->tileC);
            }
    return 0;
}

} // namespace swship
