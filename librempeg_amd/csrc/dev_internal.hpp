// Declarations shared by the four host-side translation units of the HIP layer (dev_state.hip, dev_plan.hip, dev_exec.hip, dev_api.hip).
#pragma once
#include <string>

#include "devstate.hpp"

namespace swship {
// ---- dev_state.hip ----
int ensure_dev(SwsInternal *c);
DeviceState *dev_state_for(SwsInternal *c, int device);      // the home state, or a peer state created on first use (sws_scale_frames() sharding)
bool poison_enabled();                                       // SWS_HIP_DEBUG & 16
int poison(SwsInternal *c, void *buf, size_t bytes);
bool guards_enabled();                                       // SWS_HIP_DEBUG & 64
void guard_forget(void *p);
int guard_arm(SwsInternal *c, void *p, size_t bytes);
int guards_check(SwsInternal *c, hipStream_t st);
uint64_t fnv1a64(const void *p, size_t n);
int table_alloc(SwsInternal *c, DeviceState *d, void **buf, size_t *cap, size_t need);
int table_put(SwsInternal *c, DeviceState *d, void *dst, const void *src, size_t bytes);
int dev_check_state(SwsInternal *c, DeviceState *d, std::string &out);
// bit 63 of TableRecord::serial: the block was read back and compared with its upload at the first launch of the plan that wrote it (verify_tables_once)
constexpr uint64_t TAB_VERIFIED = 1ull << 63;
int verify_tables_once(SwsInternal *c, DeviceState *d);
int ptr_device(const void *p);                               // HIP device that owns a pointer, -1 for host memory
bool is_device_ptr(const void *p);
// ---- dev_rccl.hip ----
bool rccl_tables_wanted(const SwsInternal *c);
int rccl_deliver_tables(SwsInternal *c, DeviceState *home, const std::vector<DeviceState *> &peers);
// ---- dev_plan.hip ----
int dev_prepare_on(SwsInternal *c, DeviceState *d);
int dev_plan_digest(SwsInternal *c, uint64_t out[3]);
int src_kind_of(int f);
int dst_kind_of(int f);
// ---- dev_exec.hip ----
int plane_geometry(int format, int w, int h, int plane, int *row_bytes, int *rows);
int rows_of_slice(int format, int plane, int sliceY, int sliceH, int *y0, int *rows);
int image_layout(int format, int w, int h, int align, int linesize[4], size_t offset[4], size_t *total);
void plane_extent(const PixDesc *d, int w, int h, int k, int *rows, int *row_bytes, int *vsub);
int launch_plan(SwsInternal *c, DeviceState *d, const SwsFramePtrs *frames, int n, int sliceY, int sliceH);
int run_single(SwsInternal *c, DeviceState *d, const uint8_t *const src[4], const int srcStride[4], int sliceY, int sliceH,
               uint8_t *const dst[4], const int dstStride[4], bool casc_flip0 = false);
} // namespace swship
