// Weight gradients of the training step:  dW[n][k] = sum_m dY[m][n] * X[m][k]   ("TN" GEMM: both operands are
// stored token-major, the contraction runs over the token axis m).
//
// MFMA fragments want the contraction index contiguous per lane, the operands have it strided.  Round 6: the
// transposition is done by the LDS read itself.  The token rows go from HBM / L2 straight into LDS by LDS-DMA
// (buffer_load ... lds, 16 bytes per lane, no staging registers, no VALU), row-major [32 tokens][BT features] per operand and
// stage, and the fragments are fetched with ds_read_b64_tr_b16: a 16-lane group reads a [4 tokens][16 features] block and every
// lane receives the 4 tokens of ITS feature -- two reads make the 8-token k-group of a v_mfma_f32_32x32x16_bf16 operand
// (lane = feature, lane / 32 = k-group).  Which tokens form a k-group is irrelevant as long as both operands agree.
// (Rounds 2 - 5 transposed 8 x 8 blocks in registers on the way to LDS: 32 byte-permutes + 8 ds_write_b128 per thread and step,
// one 64-token register set in flight: ~30 GB/s per CU = 0.2 of the MFMA peak on the FFN shapes; the f16 -> bf16 conversion of
// a saved activation sat in the same staging path.)
//
//   * LDS image: row r (token) of an operand is BT * 2 bytes; its 64-byte chunk c is stored at chunk c ^ (r & 3), so the four
//     token rows a half-wave reads (4 rows x 64 bytes) fall into four distinct quarters of the 256-byte bank row: conflict-free
//     by the (address / 4) % 64 rule.  The swizzle costs nothing: LDS-DMA writes lane-linear, the SOURCE address of a lane is
//     permuted instead (still whole 512- / 256-byte rows per instruction).
//   * 4 stages of 32 tokens (128 KB at BT = 256): three stages = 96 KB per CU in flight, one barrier per stage with a counted
//     vmcnt (every wave issues exactly four DMA instructions per stage; rows beyond the split are out of bounds of the buffer
//     resource and arrive as zeros, so the count never changes).
//   * dY is bf16 (gradients need the exponent range).  X is a saved forward activation: f16 or bf16; f16 fragments are
//     converted behind the read (v_cvt_f32_f16 + v_cvt_pk_bf16_f32: 24 VALU per 8 MFMAs, on the 64-feature side of the
//     wave tile).  Accumulation is f32.
//   * workgroup = BT x BT outputs (BT = 256: 8 waves, 128 accumulator registers each, wave tile 64 k x 128 n; BT = 128: 4 waves,
//     64 x 64).  The FFN shapes are bound by operand delivery (128 flop per byte from L2 at BT = 256; HBM floor of
//     [196608, 2048, 256]: 0.9 GB = 113 us) -- not by the MFMA pipe.
//   * the token range is split over `nsplit` workgroups per output tile; every workgroup writes its partial tile in ACCUMULATOR
//     ORDER (whole 1-KB lines per store instruction) and wgrad_reduce_tr_kernel sums the partials in a fixed order
//     (deterministic: no atomics), scales, and scatters to the f32 gradient with the destination's own row stride.
//   * bias gradient on the side (column sums of dY): one more MFMA per k-step against a fragment of ones, by the wave whose
//     k-group index equals the n-fragment index, in the workgroups of k-tile 0 only (every dY element is seen there once).
//
// Conv1d weight gradient (FS model :30,:40): the same kernel with the X rows of k-tile (tap, c_in block) read at
// frame t + tap - pad of the same sequence, zero outside [0, ilen) -- an out-of-bounds source offset per lane.
#include "train_common.h"
#include "kernels.h"
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) char lds_char;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int TR_TS = 32;           // tokens per stage
#ifndef EEND_WG_NST
#define EEND_WG_NST 4
#endif
constexpr int TR_NST = EEND_WG_NST; // stages in the ring
// timing-only study builds (tools/ab_one_source.sh; results are garbage): EEND_WG_NOREAD = no fragment reads, EEND_WG_NODMA = no LDS-DMA

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// 4 tokens of this lane's feature (see the header): the lane supplies the LDS address of ITS 8 bytes of the [4][16] block.
// By hand: behind the builtin (__builtin_amdgcn_ds_read_tr16_b64_*) hipcc 7.2 waits for vmcnt(0) -- every LDS-DMA piece in
// flight -- before the first read of a stage (it does not for plain LDS loads); the price is that lgkmcnt is counted by hand too
// (tr_wait ties the registers it releases, so that their consumers cannot be scheduled in front of the wait).
template <int OFF>
DEV u32x2 tr_read(unsigned addr) {
    u32x2 v;
#ifdef EEND_WG_NOREAD
    asm volatile("v_mov_b32 %0, %1" : "=v"(v[0]) : "v"(addr));
    v[1] = v[0];
#else
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
#endif
    return v;
}
template <int N>
DEV void tr_wait(u32x2 (&a)[2][2]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]) : "n"(N));
}
template <int N>
DEV void tr_wait(u32x2 (&a)[4][2]) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]),
                 "+v"(a[3][0]), "+v"(a[3][1]) : "n"(N));
}
template <int N>
DEV void tr_wait(u32x2 (&a)[2]) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(a[1]) : "n"(N));
}
// ... and an accumulator of the previous k-step's last MFMA: keeps the wait of k-step 1 behind the MFMAs of k-step 0
// (volatile asm statements keep their order among themselves only; hipcc had hoisted all waits in front of the first MFMA)
template <int N>
DEV void tr_wait_after(u32x2 (&a)[2][2], f32x16& c) {
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(c) : "n"(N));
}

template <bool F16>
DEV bf16x8 tr_frag(const u32x2 (&r)[2]) {
    if constexpr (F16) {
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        const f32x4v fl = __builtin_convertvector(__builtin_bit_cast(f16x4, r[0]), f32x4v);
        const f32x4v fh = __builtin_convertvector(__builtin_bit_cast(f16x4, r[1]), f32x4v);
        const bf16x4 bl = __builtin_convertvector(fl, bf16x4), bh = __builtin_convertvector(fh, bf16x4);
        return __builtin_shufflevector(bl, bh, 0, 1, 2, 3, 4, 5, 6, 7);
    } else {
        return __builtin_shufflevector(__builtin_bit_cast(bf16x4, r[0]), __builtin_bit_cast(bf16x4, r[1]), 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

// uniform 32-bit load through the scalar cache (a kernel that stores gets vector loads + vmcnt(0) for p.ilens[seq] otherwise)
DEV int sload_i32(const int* base, int idx) {
    int v;
    asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(base), "s"(idx * 4) : "memory");
    return v;
}

template <bool B_F16, bool CONV, bool BIAS, int BT>
__global__ __launch_bounds__(BT * 2)
void wgrad_tr_kernel(const WgradParams p) {
    constexpr int ROWB = BT * 2;                                      // bytes of a token row of one operand
    constexpr int OPB = TR_TS * ROWB;                                 // one operand of one stage
    constexpr int STB = 2 * OPB;                                      // one stage: dY rows, then X rows
    constexpr int NJ = BT / 64;                                       // 32-wide n fragments per wave (wave tile 64 k x BT / 2 n)
    constexpr int RPI = 1024 / ROWB;                                  // token rows per DMA instruction (2 / 4)
    constexpr int LPR = 64 / RPI;                                     // lanes per row (32 / 16)
    constexpr int NDA = BT / 16;                                      // DMA instructions per operand and stage (16 / 8)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave >> 1, wn = wave & 1;
    const int ntk = p.K / BT, ntn = p.N / BT;
    // block -> (output tile, token split).  XCD-aware when the split count allows it: workgroup b runs on XCD b % 8
    // (hardware round-robin), and ALL output tiles of a token split go to one XCD, so that the split's dY / X rows are
    // fetched from HBM once and the re-reads by the other tiles are hits in that XCD's L2.
    int tile, split;
    if ((p.nsplit & 7) == 0) {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        split = (i / (ntk * ntn)) * 8 + xcd;
        tile = i % (ntk * ntn);
    } else {
        tile = blockIdx.x % (ntk * ntn);
        split = blockIdx.x / (ntk * ntn);
    }
    const int n0 = (tile / ntk) * BT, k0 = (tile % ntk) * BT;
    const long m_begin = (long)split * p.m_per_split;
    long m_end = m_begin + p.m_per_split;
    if (m_end > p.M) m_end = p.M;
    const int m_len = m_end > m_begin ? (int)(m_end - m_begin) : 0;
    const int nsteps = (m_len + TR_TS - 1) / TR_TS;

    // ---- LDS-DMA role of this wave: instructions d = wave * 4 + e of the BT / 8 per stage; the first NDA fill dY, the rest X
    const bool isB = wave * 4 >= NDA;                                 // (wave-uniform)
    const int dl0 = wave * 4 - (isB ? NDA : 0);                       // first instruction of this wave inside its operand
    const int rl = lane / LPR, p16 = lane % LPR;                      // row inside the instruction, 16-byte position inside the row
    int conv_shift = 0, conv_c0 = 0;
    if constexpr (CONV) {
        const int tap = k0 / p.conv_cin;
        conv_c0 = k0 - tap * p.conv_cin;
        conv_shift = tap - p.conv_pad;
    }
    const bool convB = CONV && isB;
    const int ld = isB ? p.ldb : p.lda;
    const bool blocked = !CONV && (isB ? p.b_blocked : p.a_blocked) != 0;      // (wave-uniform)
    __amdgpu_buffer_rsrc_t rs;
    if (convB) {
        // absolute rows (the shifted rows of the first / last stage of a split lie outside the split)
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)((unsigned)((size_t)p.M * p.ldb * 2)), 0x00020000);
    } else if (blocked) {
        // [M/16][ld/32][16][32] operand (ffn_train_stream.hip): the 16-row groups are as large as 16 row-major rows, so the split's base and
        // the stage step are the row-major ones; whole groups are in bounds (rows of the last group beyond M hold finite values written
        // by the producer and meet zero rows of the other, row-major, operand)
        const char* base = (isB ? (const char*)p.B : (const char*)p.A) + (size_t)m_begin * ld * 2;
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)((unsigned)((size_t)((m_len + 15) & ~15) * ld * 2)), 0x00020000);
    } else {
        const char* base = isB ? (const char*)p.B + ((size_t)m_begin * p.ldb + k0) * 2 : (const char*)p.A + ((size_t)m_begin * p.lda + n0) * 2;
        // rows >= m_len are out of bounds -> zeros (the last row's reach, n0 * 2 + BT * 2 <= ld * 2, stays inside)
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)((unsigned)((size_t)m_len * ld * 2)), 0x00020000);
    }
    unsigned voff[4];                                                 // byte offset of this lane's 16 bytes, per instruction, stage 0
    int vrow[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int r = (dl0 + e) * RPI + rl;                           // token row inside the stage
        const int q16 = (((p16 >> 2) ^ (r & 3)) << 2) | (p16 & 3);   // the 16-byte chunk that lives at position p16 of row r
        vrow[e] = r;
        if (blocked)       // feature col0 + 8 q16 of token r: block (r >> 4, feature >> 5), row r & 15, 16-byte piece q16 & 3
            voff[e] = (unsigned)((r >> 4) * (16 * ld * 2) + (((isB ? k0 : n0) >> 5) + (q16 >> 2)) * 1024 + (r & 15) * 64 + (q16 & 3) * 16);
        else
            voff[e] = convB ? (unsigned)(conv_c0 * 2 + q16 * 16) : (unsigned)(r * ld * 2 + q16 * 16);
    }
    const unsigned stage_step = (unsigned)(TR_TS * ld * 2);
    // CONV: sequence / frame of the next stage to be requested, and that sequence's length (loaded one stage ahead)
    int cseq = 0, ct0 = 0, clen = 0;
    if constexpr (CONV) {
        cseq = (int)(m_begin / p.Tp);
        ct0 = (int)(m_begin - (long)cseq * p.Tp);
        clen = sload_i32(p.ilens, cseq < (int)((p.M + p.Tp - 1) / p.Tp) ? cseq : 0);
    }
    int issued = 0;                                                   // stages requested so far
    auto issue = [&](int buf) __attribute__((always_inline)) {
#ifdef EEND_WG_NODMA
        ++issued;
        return;
#endif
        char* dst = smem + buf * STB + (isB ? OPB : 0) + dl0 * 1024;
        if (convB) {
            const bool live = issued < nsteps;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ts = ct0 + vrow[e] + conv_shift;
                const bool ok = live && ts >= 0 && ts < clen;
                const unsigned off = ok ? (unsigned)(((long)cseq * p.Tp + ts) * p.ldb * 2) + voff[e] : 0xFFFFFF00u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_char*)(dst + e * 1024), 16, off, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_char*)(dst + e * 1024), 16, voff[e], 0, 0, 0);
                voff[e] += stage_step;
            }
        }
        if constexpr (CONV) {
            ct0 += TR_TS;
            if (ct0 >= p.Tp) {
                ct0 = 0;
                ++cseq;
                if ((long)cseq * p.Tp < p.M) clen = sload_i32(p.ilens, cseq);
            }
        }
        ++issued;
    };

    // ---- fragment addresses of this lane (stage 0, k-step 0): token rows (g >> 1) * 8 + (i16 >> 2) (+ 4 for the second read),
    // features fb + (g & 1) * 16 + (i16 & 3) * 4 .. + 3 of the 64-byte chunk c = fb / 32, stored at chunk c ^ (row & 3)
    const int g = lane >> 4, i16 = lane & 15, x2 = i16 >> 2;
    const int lrow = ((g >> 1) * 8 + x2) * ROWB + (g & 1) * 32 + (i16 & 3) * 8;
    auto chunk_off = [&](int c) __attribute__((always_inline)) { return ((c & ~3) | ((c & 3) ^ x2)) << 6; };
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
    unsigned addrB[2], addrA[NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i) addrB[i] = lds0 + OPB + lrow + chunk_off(wk * 2 + i);
#pragma unroll
    for (int j = 0; j < NJ; ++j) addrA[j] = lds0 + lrow + chunk_off(wn * NJ + j);
    const bool do_bias = BIAS && (tile % ntk) == 0;
    const unsigned addrAb = lds0 + lrow + chunk_off(wn * NJ + (wk & (NJ - 1)));   // the n fragment whose column sums this wave carries

    f32x16 acc[2][NJ], accb;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    // ---- main loop, software-pipelined ACROSS the stage barrier: the fragments of a k-step are requested one k-step ahead
    // (two register sets, 48 registers), and the barrier that publishes stage st + 1 sits between the two MFMA groups of stage st,
    // so that no wave meets it with an empty MFMA queue and the first fragments of the next stage travel under this stage's
    // second group.  (First version: barrier at the stage top, all reads of the stage behind it -- the MFMA pipe was 58 % busy
    // even with the DMA compiled out: 163 of 198 us at [196608, 2048, 256].)
    constexpr int RPK = 2 * (2 + NJ + (BIAS ? 1 : 0));                 // fragment reads per k-step
    u32x2 rb[2][2][2], ra[2][NJ][2], rs1[2][2];
    auto request = [&](auto SET, unsigned sb) __attribute__((always_inline)) {   // k-step SET of the stage at LDS offset sb -> register set SET
        constexpr int ks = decltype(SET)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) { rb[ks][i][0] = tr_read<ks * 16 * ROWB>(addrB[i] + sb); rb[ks][i][1] = tr_read<(ks * 16 + 4) * ROWB>(addrB[i] + sb); }
#pragma unroll
        for (int j = 0; j < NJ; ++j) { ra[ks][j][0] = tr_read<ks * 16 * ROWB>(addrA[j] + sb); ra[ks][j][1] = tr_read<(ks * 16 + 4) * ROWB>(addrA[j] + sb); }
        if constexpr (BIAS) { rs1[ks][0] = tr_read<ks * 16 * ROWB>(addrAb + sb); rs1[ks][1] = tr_read<(ks * 16 + 4) * ROWB>(addrAb + sb); }
    };
    auto consume = [&](auto SET) __attribute__((always_inline)) {     // wait for register set SET (the younger set may stay in flight), 8 MFMAs
        constexpr int ks = decltype(SET)::value;
        tr_wait_after<RPK>(rb[ks], acc[1][NJ - 1]);                    // (tied to the previous group's last accumulator: keeps the order)
        tr_wait<RPK>(ra[ks]);
        if constexpr (BIAS) tr_wait<RPK>(rs1[ks]);
        bf16x8 bfr[2], afr[NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i) bfr[i] = tr_frag<B_F16>(rb[ks][i]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) afr[j] = tr_frag<false>(ra[ks][j]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[i], afr[j], acc[i][j], 0, 0, 0);
        if constexpr (BIAS) {
            if (do_bias) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, tr_frag<false>(rs1[ks]), accb, 0, 0, 0);
        }
    };
    typedef std::integral_constant<int, 0> K0;
    typedef std::integral_constant<int, 1> K1;
#pragma unroll
    for (int s = 0; s < TR_NST - 1; ++s) issue(s);
    __builtin_amdgcn_s_waitcnt(0x0F70 | (4 * (TR_NST - 2)));           // this wave's pieces of stage 0 (two younger stages in flight)
    __builtin_amdgcn_s_barrier();
    request(K0{}, 0u);
    int buf = 0;
    for (int st = 0; st < nsteps; ++st) {
        const unsigned sb = buf * STB;
        const int nb = buf + 1 == TR_NST ? 0 : buf + 1;
        request(K1{}, sb);
        consume(K0{});
        __builtin_amdgcn_sched_barrier(0);
        // this wave's pieces of stage st + 1 have landed (one younger stage may stay in flight); behind the barrier every wave's
        // have, and every wave has its last fragments of stage st - 1 in registers: that buffer takes stage st + 3
        __builtin_amdgcn_s_waitcnt(0x0F70 | (4 * (TR_NST - 3)));
        __builtin_amdgcn_s_barrier();
        issue(buf == 0 ? TR_NST - 1 : buf - 1);
        request(K0{}, (unsigned)(nb * STB));                           // (beyond the last stage: zero rows, never consumed)
        consume(K1{});
        __builtin_amdgcn_sched_barrier(0);
        buf = nb;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the trailing request

    // ---- partial tile in accumulator order: [split][tile][wave][i][j][register quad][lane] as f32x4
    // acc[i][j][r]: k = k0 + wk*64 + i*32 + 8*(r>>2) + (lane>>5)*4 + (r&3),  n = n0 + wn*(BT/2) + j*32 + (lane&31)
    float* __restrict__ out = p.partial + (size_t)split * p.N * p.K + (size_t)tile * (BT * BT) + (size_t)wave * (2 * NJ * 16 * 64);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 v = f32x4{acc[i][j][r4 * 4], acc[i][j][r4 * 4 + 1], acc[i][j][r4 * 4 + 2], acc[i][j][r4 * 4 + 3]};
                *(f32x4*)(out + (((i * NJ + j) * 4 + r4) * 64 + lane) * 4) = v;
            }
    if constexpr (BIAS) {
        // every row of accb is the column sum: register 0 of lanes 0..31 (row 0) carries n = lane
        if (do_bias && wk < NJ && lane < 32)
            p.bias_partial[(size_t)split * p.N + n0 + wn * (BT / 2) + wk * 32 + lane] = accb[0];
    }
}

// out[n][k] (row stride ld_out, k < K_out) = scale * sum_s partial[s][...]   (+ out if accumulate), partial in the accumulator order
// of wgrad_tr_kernel<BT>: one thread per f32x4, consecutive threads read consecutive 16 bytes of every split's partial.
template <int BT>
__global__ __launch_bounds__(256)
void wgrad_reduce_tr_kernel(const float* __restrict__ partial, long split_stride, int nsplit, int N, int K, int K_out,
                            float* __restrict__ out, int ld_out, float scale, int accumulate, int group_rows, long group_gap) {
    constexpr int NJ = BT / 64;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;            // f32x4 slot
    if (idx >= (long)N * K / 4) return;
    const int ntk = K / BT;
    const int tile = (int)(idx / (BT * BT / 4));
    int rem = (int)(idx - (long)tile * (BT * BT / 4));
    const int wave = rem / (2 * NJ * 4 * 64);
    rem -= wave * (2 * NJ * 4 * 64);
    const int q = rem >> 6, lane = rem & 63;
    const int i = q / (NJ * 4), j = (q >> 2) % NJ, r4 = q & 3;
    const int wk = wave >> 1, wn = wave & 1;
    const int n = (tile / ntk) * BT + wn * (BT / 2) + j * 32 + (lane & 31);
    const int k = (tile % ntk) * BT + wk * 64 + i * 32 + 8 * r4 + (lane >> 5) * 4;
    const f32x4* src = (const f32x4*)partial + idx;
    const long ss4 = split_stride / 4;
    f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int s = 0;
    for (; s + 4 <= nsplit; s += 4) {
        s0 += src[(size_t)(s + 0) * ss4];
        s1 += src[(size_t)(s + 1) * ss4];
        s2 += src[(size_t)(s + 2) * ss4];
        s3 += src[(size_t)(s + 3) * ss4];
    }
    for (; s < nsplit; ++s) s0 += src[(size_t)s * ss4];
    const f32x4 v = ((s0 + s1) + (s2 + s3)) * scale;
    // (group_rows > 0: the rows belong to several equally spaced destination tensors -- the four projections of a retention module in the
    // flat gradient buffer: group g = n / group_rows starts group_gap floats further than contiguous rows would)
    float* o = out + (size_t)n * ld_out + k + (group_rows > 0 ? (long)(n / group_rows) * group_gap : 0L);
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (k + e < K_out) o[e] = accumulate ? o[e] + v[e] : v[e];
}

// out[n][k] (row stride ld_out, k < K_out) = scale * sum_s partial[s][n][k]   (+ out if accumulate)
__global__ __launch_bounds__(256)
void wgrad_reduce_kernel(const float* __restrict__ partial, long split_stride, int nsplit, int N, int K, int K_out,
                         float* __restrict__ out, int ld_out, float scale, int accumulate) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * K_out) return;
    const int n = (int)(idx / K_out), k = (int)(idx - (long)n * K_out);
    const float* src = partial + (size_t)n * K + k;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int s = 0;
    for (; s + 4 <= nsplit; s += 4) {
        s0 += src[(size_t)(s + 0) * split_stride];
        s1 += src[(size_t)(s + 1) * split_stride];
        s2 += src[(size_t)(s + 2) * split_stride];
        s3 += src[(size_t)(s + 3) * split_stride];
    }
    for (; s < nsplit; ++s) s0 += src[(size_t)s * split_stride];
    float v = ((s0 + s1) + (s2 + s3)) * scale;
    float* o = out + (size_t)n * ld_out + k;
    if (accumulate) v += *o;
    *o = v;
}

// Few outputs, many partials (LayerNorm / bias gradients: <= a few thousand outputs, up to 1024 partials): 8 lane
// groups walk the partials in parallel (coalesced over 32 consecutive outputs) and are combined in a fixed order.
__global__ __launch_bounds__(256)
void wgrad_reduce_small_kernel(const float* __restrict__ partial, long split_stride, int nsplit, int N, int K, int K_out,
                               float* __restrict__ out, int ld_out, float scale, int accumulate) {
    __shared__ float red[8][32];
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long idx = (long)blockIdx.x * 32 + o;
    const bool live = idx < (long)N * K_out;
    float s = 0.f;
    if (live) {
        const int n = (int)(idx / K_out), k = (int)(idx - (long)n * K_out);
        const float* src = partial + (size_t)n * K + k;
        // eight loads in flight per thread (the walk is latency-bound: 20 us per call with two); fixed combination order
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int sp = g;
        for (; sp + 56 < nsplit; sp += 64) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += src[(size_t)(sp + 8 * j) * split_stride];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (sp + 8 * j < nsplit) a[j] += src[(size_t)(sp + 8 * j) * split_stride];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    red[g][o] = s;
    __syncthreads();
    if (g == 0 && live) {
        const int n = (int)(idx / K_out), k = (int)(idx - (long)n * K_out);
        float v = (((red[0][o] + red[1][o]) + (red[2][o] + red[3][o])) + ((red[4][o] + red[5][o]) + (red[6][o] + red[7][o]))) * scale;
        float* dst = out + (size_t)n * ld_out + k;
        if (accumulate) v += *dst;
        *dst = v;
    }
}

// Up to three vectors from one partial matrix [nsplit][ncols] (row stride split_stride): columns [j * W, (j + 1) * W) -> outs j
// (LayerNorm backward: d gamma | d beta | d bias of one pass over the rows -- one launch instead of three; a null
// destination skips its columns).  Same walk and combination order as wgrad_reduce_small_kernel.
__global__ __launch_bounds__(256)
void wgrad_reduce_multi_kernel(const float* __restrict__ partial, long split_stride, int nsplit, int ncols, int W,
                               float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ out2) {
    __shared__ float red[8][32];
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + o;
    const int seg = col / W;
    float* dst = seg == 0 ? out0 : (seg == 1 ? out1 : out2);
    const bool live = col < ncols && dst != nullptr;
    float s = 0.f;
    if (live) {
        const float* src = partial + col;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int sp = g;
        for (; sp + 56 < nsplit; sp += 64) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += src[(size_t)(sp + 8 * j) * split_stride];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (sp + 8 * j < nsplit) a[j] += src[(size_t)(sp + 8 * j) * split_stride];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    red[g][o] = s;
    __syncthreads();
    if (g == 0 && live)
        dst[col - seg * W] = ((red[0][o] + red[1][o]) + (red[2][o] + red[3][o])) + ((red[4][o] + red[5][o]) + (red[6][o] + red[7][o]));
}

// Column sums of a 2-byte-float matrix [M][N] (bias gradients): partial[s][n] = sum over the split's rows.
// A block covers 256 columns (32 threads x 16-byte loads) in 8 row phases; N % 8 == 0.
template <bool IS_BF16>
__global__ __launch_bounds__(256)
void colsum_partial_kernel(const unsigned short* __restrict__ Y, int ld /* in 2-byte elements */, long M, int N, long rows_per_split,
                           float* __restrict__ partial) {
    __shared__ float red[8][256];
    const int cg = threadIdx.x & 31, ph = threadIdx.x >> 5;
    const int col = blockIdx.x * 256 + cg * 8;
    const long r0 = (long)blockIdx.y * rows_per_split;
    long r1 = r0 + rows_per_split;
    if (r1 > M) r1 = M;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        for (long r = r0 + ph; r < r1; r += 8) {
            const u32x4 v = *(const u32x4*)(Y + r * ld + col);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned x = v[j];
                if (IS_BF16) { a[2 * j] += bf16_lo(x); a[2 * j + 1] += bf16_hi(x); }
                else { a[2 * j] += f16_lo(x); a[2 * j + 1] += f16_hi(x); }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ph][cg * 8 + e] = a[e];
    __syncthreads();
    const int c = threadIdx.x;
    if (blockIdx.x * 256 + c < N)
        partial[(size_t)blockIdx.y * N + blockIdx.x * 256 + c] =
            ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) + ((red[4][c] + red[5][c]) + (red[6][c] + red[7][c]));
}

// cnn.weight gradient back to the parameter's own layout: g[co][ci][tap] = tmp[co][tap*cin + ci]
__global__ __launch_bounds__(256)
void conv_wgrad_unpermute_kernel(const float* __restrict__ tmp, float* __restrict__ g, int cout, int cin, int ktaps) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)cout * cin * ktaps) return;
    const int tap = (int)(idx % ktaps);
    const long r = idx / ktaps;
    const int ci = (int)(r % cin), co = (int)(r / cin);
    g[idx] = tmp[((size_t)co * ktaps + tap) * cin + ci];
}

}  // namespace

int eend_launch_wgrad(const WgradParams& p, hipStream_t stream) {
    if (!p.A || !p.B || !p.partial || p.M <= 0 || p.N <= 0 || p.K <= 0) return EEND_EINVAL;
    const int bt = p.tile == 256 ? 256 : 128;
    if ((p.N % bt) || (p.K % bt) || (p.lda & 7) || (p.ldb & 7) || p.nsplit <= 0 || p.m_per_split <= 0 || (p.m_per_split % 64))
        return EEND_EINVAL;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15)) return EEND_EINVAL;
    if ((p.a_blocked && (p.conv || (p.lda & 31) || p.N > p.lda)) || (p.b_blocked && (p.conv || (p.ldb & 31) || p.K > p.ldb))) return EEND_EINVAL;
    if (p.conv && (!p.ilens || p.conv_cin <= 0 || (p.conv_cin % bt) || p.Tp <= 0 || (p.Tp % 64))) return EEND_EINVAL;
    // 32-bit byte offsets inside a split (inside the whole X tensor for the Conv1d form), four stages of run-ahead included
    const size_t reach_a = ((size_t)p.m_per_split + 256) * p.lda * 2, reach_b = ((size_t)(p.conv ? p.M : p.m_per_split) + 256) * p.ldb * 2;
    if (reach_a >= 0xFFFF0000ull || reach_b >= 0xFFFF0000ull) return EEND_EINVAL;
    const int smem = TR_NST * 2 * TR_TS * bt * 2;
    const dim3 grid((unsigned)((p.N / bt) * (p.K / bt) * p.nsplit));
#define WG_LAUNCH(F16, CV, BS, BT)                                                                                      \
    do {                                                                                                                \
        static EendOncePerDevice attr_once;                                                                             \
        if (!eend_set_dynamic_lds(attr_once, (const void*)wgrad_tr_kernel<F16, CV, BS, BT>, smem)) return EEND_ELAUNCH;  \
        hipLaunchKernelGGL((wgrad_tr_kernel<F16, CV, BS, BT>), grid, dim3(BT * 2), smem, stream, p);                    \
    } while (0)
#define WG_PICK(BT)                                                                                                     \
    do {                                                                                                                \
        if (p.conv) { if (p.b_is_f16) WG_LAUNCH(true, true, false, BT); else WG_LAUNCH(false, true, false, BT); }       \
        else if (p.bias_partial) { if (p.b_is_f16) WG_LAUNCH(true, false, true, BT); else WG_LAUNCH(false, false, true, BT); } \
        else { if (p.b_is_f16) WG_LAUNCH(true, false, false, BT); else WG_LAUNCH(false, false, false, BT); }            \
    } while (0)
    if (p.bias_partial && p.conv) return EEND_EINVAL;
    if (bt == 256) WG_PICK(256); else WG_PICK(128);
#undef WG_PICK
#undef WG_LAUNCH
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

// the partial tiles of eend_launch_wgrad (accumulator order of its `tile`) -> out[n][k]
int eend_launch_wgrad_reduce_tiles(const float* partial, int tile, int nsplit, int N, int K, int K_out, float* out, int ld_out,
                                   float scale, int accumulate, hipStream_t stream, int group_rows, long group_gap) {
    const int bt = tile == 256 ? 256 : 128;
    if (!partial || !out || nsplit <= 0 || N <= 0 || K <= 0 || (N % bt) || (K % bt) || K_out <= 0 || K_out > K || ld_out < K_out)
        return EEND_EINVAL;
    const long n4 = (long)N * K / 4;
    const dim3 grid((unsigned)((n4 + 255) / 256));
    if (bt == 256)
        hipLaunchKernelGGL(wgrad_reduce_tr_kernel<256>, grid, dim3(256), 0, stream, partial, (long)N * K, nsplit, N, K, K_out, out, ld_out, scale, accumulate, group_rows, group_gap);
    else
        hipLaunchKernelGGL(wgrad_reduce_tr_kernel<128>, grid, dim3(256), 0, stream, partial, (long)N * K, nsplit, N, K, K_out, out, ld_out, scale, accumulate, group_rows, group_gap);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_wgrad_reduce(const float* partial, long split_stride, int nsplit, int N, int K, int K_out, float* out,
                             int ld_out, float scale, int accumulate, hipStream_t stream) {
    if (!partial || !out || nsplit <= 0 || N <= 0 || K <= 0 || K_out <= 0 || K_out > K || ld_out < K_out) return EEND_EINVAL;
    const long n = (long)N * K_out;
    if (n <= 16384 && nsplit >= 16)
        hipLaunchKernelGGL(wgrad_reduce_small_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, stream, partial, split_stride,
                           nsplit, N, K, K_out, out, ld_out, scale, accumulate);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, partial, split_stride,
                           nsplit, N, K, K_out, out, ld_out, scale, accumulate);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_wgrad_reduce_multi(const float* partial, long split_stride, int nsplit, int W, float* out0, float* out1, float* out2,
                                   hipStream_t stream) {
    if (!partial || nsplit <= 0 || W <= 0 || (W & 31) || !out0) return EEND_EINVAL;
    const int ncols = 3 * W;
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)(ncols / 32)), dim3(256), 0, stream, partial, split_stride, nsplit, ncols, W,
                       out0, out1, out2);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_colsum_partial(const void* Y, int ld, long M, int N, int is_bf16, int nsplit, float* partial,
                               hipStream_t stream) {
    if (!Y || !partial || M <= 0 || N <= 0 || (N & 7) || (ld & 7) || nsplit <= 0 || nsplit > 65535) return EEND_EINVAL;
    const long rps = (M + nsplit - 1) / nsplit;
    const dim3 grid((N + 255) / 256, nsplit);
    if (is_bf16)
        hipLaunchKernelGGL(colsum_partial_kernel<true>, grid, dim3(256), 0, stream, (const unsigned short*)Y, ld, M, N, rps, partial);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<false>, grid, dim3(256), 0, stream, (const unsigned short*)Y, ld, M, N, rps, partial);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

int eend_launch_conv_wgrad_unpermute(const float* tmp, float* g, int cout, int cin, int ktaps, hipStream_t stream) {
    if (!tmp || !g || cout <= 0 || cin <= 0 || ktaps <= 0) return EEND_EINVAL;
    const long n = (long)cout * cin * ktaps;
    hipLaunchKernelGGL(conv_wgrad_unpermute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tmp, g, cout, cin, ktaps);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}
