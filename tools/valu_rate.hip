// Throughput probe: cycles per wave-instruction and SIMD for the integer VALU ops the swscale kernels lean on (gfx950).
// Every kernel runs ITER iterations of 16 independent instructions of one kind in each of WAVES_PER_SIMD waves on every SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITER 4096
#define DEF_KERNEL(NAME, ASM, ...) \
__global__ void __launch_bounds__(256) k_##NAME(uint32_t *out, uint32_t seed) { \
    uint32_t a[16], b = seed + threadIdx.x, c = seed * 3 + 1; \
    for (int i = 0; i < 16; i++) a[i] = seed + i * 7 + threadIdx.x; \
    for (int it = 0; it < ITER; it++) { \
        _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) __VA_ARGS__); \
    } \
    uint32_t s = 0; for (int i = 0; i < 16; i++) s += a[i]; \
    out[blockIdx.x * 256 + threadIdx.x] = s; }

DEF_KERNEL(add_u32, "v_add_u32 %0, %0, %1")
DEF_KERNEL(mad_i32_i24, "v_mad_i32_i24 %0, %1, %2, %0")
DEF_KERNEL(mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
DEF_KERNEL(mul_i32_i24, "v_mul_i32_i24 %0, %0, %1")
DEF_KERNEL(mul_i32_i24_sdwa, "v_mul_i32_i24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
DEF_KERNEL(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL(dot2_i32_i16, "v_dot2_i32_i16 %0, %1, %2, %0")
DEF_KERNEL(dot2c_i32_i16, "v_dot2c_i32_i16 %0, %1, %2")
DEF_KERNEL(dot4_i32_i8, "v_dot4_i32_i8 %0, %1, %2, %0")
DEF_KERNEL(perm_b32, "v_perm_b32 %0, %0, %1, %2")
DEF_KERNEL(ashr_pk_u8_i32, "v_ashr_pk_u8_i32 %0, %1, %2, 16")
DEF_KERNEL(bfe_u32, "v_bfe_u32 %0, %0, 8, 8")
DEF_KERNEL(lshl_sdwa, "v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
DEF_KERNEL(add3_u32, "v_add3_u32 %0, %0, %1, %2")
DEF_KERNEL(pk_add_u16, "v_pk_add_u16 %0, %0, %1")
DEF_KERNEL(pk_mad_i16, "v_pk_mad_i16 %0, %1, %2, %0")
DEF_KERNEL(pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
DEF_KERNEL(cvt_pk_i16_i32, "v_cvt_pk_i16_i32 %0, %0, %1")
DEF_KERNEL(mov_b32, "v_mov_b32 %0, %1")
DEF_KERNEL(fma_f32, "v_fma_f32 %0, %1, %2, %0")
DEF_KERNEL(mad_i32_i16, "v_mad_i32_i16 %0, %1, %2, %0")
DEF_KERNEL(min_i32, "v_min_i32 %0, %0, %1")
DEF_KERNEL(med3_i32, "v_med3_i32 %0, %0, %1, %2")
DEF_KERNEL(lshrrev, "v_lshrrev_b32 %0, 3, %0")
DEF_KERNEL(and_or, "v_and_or_b32 %0, %0, %1, %2")
DEF_KERNEL(mad_u64_u32_skip, "v_add_u32 %0, %0, %2")

struct Probe { const char *name; void (*k)(uint32_t *, uint32_t); };
#define P(N) { #N, k_##N }

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint32_t *out; hipMalloc(&out, (size_t)cus * 16 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    Probe probes[] = { P(add_u32), P(mad_i32_i24), P(mad_u32_u24), P(mul_i32_i24), P(mul_i32_i24_sdwa), P(mul_lo_u32), P(dot2_i32_i16), P(dot2c_i32_i16),
                       P(dot4_i32_i8), P(perm_b32), P(ashr_pk_u8_i32), P(bfe_u32), P(lshl_sdwa), P(add3_u32), P(pk_add_u16), P(pk_mad_i16), P(pk_mul_lo_u16),
                       P(cvt_pk_i16_i32), P(mov_b32), P(fma_f32), P(mad_i32_i16), P(min_i32), P(med3_i32), P(lshrrev), P(and_or) };
    printf("%d CUs, clock %d kHz\n", cus, prop.clockRate);
    for (int wps : { 1, 2, 4 }) {          // waves per SIMD: blocks of 4 waves, one per SIMD -> wps blocks per CU
        printf("---- %d wave(s) per SIMD ----\n", wps);
        for (auto &p : probes) {
            const int blocks = cus * wps;
            hipLaunchKernelGGL(p.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(p.k, dim3(blocks), dim3(256), 0, 0, out, 2u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per SIMD: wps waves x ITER x 16 instructions
            const double instr = (double)wps * ITER * 16;
            const double cyc = ms * 1e-3 * 2.4e9;      // at the 2.4 GHz peak clock: an upper bound of the real cycle count
            printf("%-22s %8.3f ms  %6.2f cycles/instr/SIMD (at 2.4 GHz)\n", p.name, ms, cyc / instr);
        }
    }
    return 0;
}
