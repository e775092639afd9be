// Shared device-side helpers for the gfx950 (CDNA4 / MI355X) EEND kernels.
// wave = 64 lanes, MFMA 16x16x32 (f16 linears) and 32x32x16 (bf16 attention).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define EEND_OK 0
#define EEND_EINVAL (-1)
#define EEND_ELAUNCH (-2)

#define DEV __device__ __forceinline__

// fp16 saturating round-to-nearest conversion (keeps +-inf out of the f16 MFMA operands)
DEV _Float16 to_f16_sat(float x) {
    x = __builtin_fminf(__builtin_fmaxf(x, -65504.0f), 65504.0f);
    return (_Float16)x;
}

// Byte offset of 16-byte chunk `c` (0..7) of row `row` in a [rows][128 B] LDS tile.
// XOR swizzle with (row>>1)&7: any 16 distinct consecutive-ish rows reading the
// same logical chunk hit 16 distinct 16-B slots of the 256-B bank row, so the
// ds_read_b128 of an MFMA fragment (16x16x32: lane = row, 32x32x16: lane&31 =
// row) is conflict-free; ds_write_b128 of a contiguous row is conflict-free too.
DEV int swz128(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }

// Lanes of one wave exchanging data through LDS: the hardware runs a wave's LDS operations in order, so no wait is needed, but
// the COMPILER reasons per thread -- without this it may reuse an earlier load of the same address (found the hard way: a
// staging tile read back after other lanes rewrote it).  Generates no instructions.
DEV void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Sums over lane ^ 16 / lane ^ 32 without an LDS round trip (ds_bpermute costs an exposed ~150-cycle wait in one-wave-per-SIMD
// kernels): gfx950's permlane swaps exchange the odd 16-lane rows of one register with the even rows of another (16) or the
// upper half of one with the lower half of another (32); with both registers holding v, their sum is the all-reduce.  By hand:
// this hipcc maps both results of __builtin_amdgcn_permlane16_swap to the same register.  (s_nop: VALU write -> permlane read.)
DEV float wave_xor16_add(float v) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
DEV float wave_xor32_add(float v) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
DEV float wave_g_allreduce_add(float v) { return wave_xor32_add(wave_xor16_add(v)); }      // over the four 16-lane rows

DEV float wave_xor_add(float v, int m) { return v + __shfl_xor(v, m, 64); }
DEV float wave_xor_max(float v, int m) { return __builtin_fmaxf(v, __shfl_xor(v, m, 64)); }

// All-reduce (sum) inside each aligned group of 16 lanes with DPP moves
// (quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_half_mirror, row_mirror).
DEV float row16_allreduce_add(float v) {
    int t;
    t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false);
    v += __builtin_bit_cast(float, t);
    t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false);
    v += __builtin_bit_cast(float, t);
    t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false);
    v += __builtin_bit_cast(float, t);
    t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false);
    v += __builtin_bit_cast(float, t);
    return v;
}

// XCD-aware bijective remap of a 1-D block id: hardware places block b on XCD
// b % 8; give each XCD one contiguous range of logical tile ids so tiles that
// share an activation panel are served by the same (non-coherent) 4 MiB L2.
DEV int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// ---- dropout (training step): counter-based, stateless.  keep(a, b) is a pure function of (seed, a, b), so the
// backward kernels regenerate the mask of any forward site instead of storing it.  a = row-like index, b = column-like.
// hash of a * golden + b (drop_mix below); keep probability 1 - thresh24 / 2^24.
struct DropSpec {
    unsigned seed;        // per-site seed (host: step seed mixed with the site id)
    unsigned thresh24;    // round(p * 2^24); 0 = dropout off
    float scale;          // 1 / (1 - p)
};
// Round 6: two rounds of xor-shift + 24-bit multiply (v_mul_u32_u24: full rate; the 32-bit multiplies of the murmur3 finaliser used until
// round 5 are quarter rate and were half of a mask bit's 18 issue slots in the VALU-bound training kernels -- 10 now).  The compare
// reads the top 24 bits of the second product; keep rate, row / column spread, lag correlations and 2 x 2 block counts measure like the
// murmur masks' (numbers in the test checker's restatement of this function).
DEV unsigned drop_mix(unsigned x) {                  // x = (a * golden + b) ^ seed
    x ^= x >> 16;
    x = __umul24(x, 0x6B2F4Du);                        // v_mul_u32_u24: (x & 0xFFFFFF) * C, low 32 bits (the explicit mask cost a v_and per use)
    x ^= x >> 13;
    return __umul24(x, 0x9E3779u);
}
DEV bool drop_keep(const DropSpec d, unsigned a, unsigned b) {
    return drop_mix((a * 0x9E3779B1u + b) ^ d.seed) >= (d.thresh24 << 8);     // (h >> 8) >= thresh24
}
DEV float drop_apply(const DropSpec d, float v, unsigned a, unsigned b) {
    return d.thresh24 == 0 ? v : (drop_keep(d, a, b) ? v * d.scale : 0.f);
}

// ---- host-side launch helpers: per-device, thread-safe caches (function-local `static bool attr_done` / `static int ncu` were
// neither -- harmless with one process per GPU, wrong the day one process drives two devices or two host threads race on first use).
#include <atomic>
struct EendOncePerDevice {                       // bit d = "done on device d" (devices >= 32 simply redo the call every time)
    std::atomic<unsigned> mask{0};
};
inline int eend_current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess ? dev : 0;
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device)
inline bool eend_set_dynamic_lds(EendOncePerDevice& once, const void* kern, int bytes) {
    const int dev = eend_current_device();
    if (dev >= 0 && dev < 32 && (once.mask.load(std::memory_order_acquire) >> dev & 1u)) return true;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    if (dev >= 0 && dev < 32) once.mask.fetch_or(1u << dev, std::memory_order_release);
    return true;
}
// compute units of the current device (cached per device)
inline int eend_cu_count() {
    static std::atomic<int> cus[32];
    const int dev = eend_current_device();
    int n = (dev >= 0 && dev < 32) ? cus[dev].load(std::memory_order_relaxed) : 0;
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    if (dev >= 0 && dev < 32) cus[dev].store(n, std::memory_order_relaxed);
    return n;
}
