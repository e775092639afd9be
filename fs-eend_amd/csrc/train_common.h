// Device helpers shared by the training-step kernels (backward pass, optimiser).
#pragma once
#include "common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Swizzle of a [rows][128 B] LDS tile that is WRITTEN by the 8x8 register transposes below (a lane writes the
// 8 rows 8*fc .. 8*fc+7 at one 16-byte chunk) and READ as MFMA fragments (16 consecutive rows, one logical chunk).
// key = (row>>1) ^ (row>>4): over 16 consecutive rows (row>>4 fixed) the key still takes 8 distinct values twice
// -> fragment ds_read_b128 stays conflict-free like swz128; over rows 8*fc + e (fc = 0..15) it takes 8 values
// twice as well -> the transposed writes are 2-way instead of 8-way conflicted.
DEV int swzT(int row, int c) { return row * 128 + ((c ^ (((row >> 1) ^ (row >> 4)) & 7)) << 4); }

// 8x8 transpose of 16-bit elements held in registers: in[r] = row r (8 elements, element e in the low/high half of
// dword e>>1); out[e] = column e as a row: dword j = (in[2j][e], in[2j+1][e]).
DEV void transpose8x8_b16(const u32x4 (&in)[8], u32x4 (&out)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned a = in[2 * j][e >> 1], b = in[2 * j + 1][e >> 1];
            out[e][j] = (e & 1) ? ((a >> 16) | (b & 0xFFFF0000u)) : ((a & 0xFFFFu) | (b << 16));
        }
    }
}

// 8 packed f16 -> 8 packed bf16 (round to nearest even through f32).
DEV u32x4 f16x8_to_bf16x8(u32x4 v) {
    u32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        const unsigned x = v[j];       // (bit_cast straight from the vector-element lvalue reads element 0 for every j: hipcc 7.2)
        const h2 h = __builtin_bit_cast(h2, x);
        b2 o;
        o[0] = (__bf16)(float)h[0];
        o[1] = (__bf16)(float)h[1];
        r[j] = __builtin_bit_cast(unsigned, o);
    }
    return r;
}

DEV float bf16_lo(unsigned v) { return __builtin_bit_cast(float, v << 16); }
DEV float bf16_hi(unsigned v) { return __builtin_bit_cast(float, v & 0xFFFF0000u); }
DEV unsigned pack_bf16(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 o;
    o[0] = (__bf16)lo;
    o[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, o);
}
DEV float f16_lo(unsigned v) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    return (float)__builtin_bit_cast(h2, v)[0];
}
DEV float f16_hi(unsigned v) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    return (float)__builtin_bit_cast(h2, v)[1];
}

DEV float wave_sum(float v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}
