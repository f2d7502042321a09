// Hardware probe for the semantics of gfx950 pack/saturate instructions (run on the MI355X box):
//   hipcc --offload-arch=gfx950 -O2 tools/isa_probe.hip -o /tmp/isa_probe && /tmp/isa_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned *out, int a, int b)
{
    unsigned d0 = 0xAABBCCDDu, d1 = 0xAABBCCDDu, d2 = 0xAABBCCDDu, d3 = 0xAABBCCDDu;
    int va = a + (int)threadIdx.x * 0, vb = b + (int)threadIdx.x * 0;
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 16" : "+v"(d0) : "v"(va), "v"(vb));
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 16 op_sel:[0,0,0,1]" : "+v"(d1) : "v"(va), "v"(vb));
    unsigned pk = ((unsigned)(unsigned short)(short)(a >> 16)) | ((unsigned)(unsigned short)(short)(b >> 16) << 16);
    asm volatile("v_sat_pk_u8_i16 %0, %1" : "+v"(d2) : "v"(pk));
    asm volatile("v_ashr_pk_i8_i32 %0, %1, %2, 16" : "+v"(d3) : "v"(va), "v"(vb));
    if (threadIdx.x == 0) { out[0] = d0; out[1] = d1; out[2] = d2; out[3] = d3; }
}
int main()
{
    unsigned *d, h[4];
    hipMalloc(&d, 16);
    int cases[][2] = { { 300 << 16, -5 * 65536 }, { 17 << 16, 200 << 16 }, { (255 << 16) + 65535, 256 << 16 }, { -1, 65536 } };
    for (auto &c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c[0], c[1]);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("a=%d(>>16=%d) b=%d(>>16=%d): ashr_pk_u8=%08x  op_sel_hi=%08x  sat_pk_u8_i16=%08x  ashr_pk_i8=%08x\n", c[0], c[0] >> 16, c[1], c[1] >> 16, h[0], h[1], h[2], h[3]);
    }
    return 0;
}
