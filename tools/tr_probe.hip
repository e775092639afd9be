// probe of ds_read_b64_tr_b16: LDS filled with element ids; every lane of a 16-lane group supplies its own address
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, i16 = lane & 15;
    // rows of 64 elements (128 B); group g reads rows 4g..4g+3 (row = i16 >> 2), columns (i16 & 3) * 4 .. +3 of column block 16*g
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)lds + ((g * 4 + (i16 >> 2)) * 64 + g * 16 + (i16 & 3) * 4) * 2;
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    out[lane * 4 + 0] = v[0] & 0xFFFF; out[lane * 4 + 1] = v[0] >> 16; out[lane * 4 + 2] = v[1] & 0xFFFF; out[lane * 4 + 3] = v[1] >> 16;
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, c = l & 15;
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            const int e = h[l * 4 + j], row = e / 64, col = e % 64;
            printf(" (r%d,c%d)", row, col);
            if (row != g * 4 + j || col != g * 16 + c) ++bad;            // expected: element j = row 4g + j, column 16g + (lane & 15)
        }
        printf("\n");
    }
    printf("tr_probe: %s (%d mismatches against lane l <- rows 4g..4g+3 of column 16g + (l & 15))\n", bad ? "UNEXPECTED" : "as expected", bad);
    return 0;
}
