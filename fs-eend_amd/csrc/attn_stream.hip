// In-projection + causal multi-head attention of one (sequence, head) per persistent workgroup
// (nn.TransformerEncoderLayer.self_attn, FS model :147; self_attn1 of the fusion layers, merge_tfm_encoder.py:379-385):
//   a wave owns the 64 TOKENS whose queries it will run in the flash loop (query blocks w and 15 - w) and keeps their X rows in
//   registers as MFMA operand fragments (128 VGPRs, read from global memory once); the head's 96 KB of weights, pre-packed per
//   layer in fragment order (eend_inproj_attn_pack_f16), are requested by LDS-DMA in one go at the start of the item into space
//   that is free at that time -- four 16-KB items into the V^T tile region (written only by the last two projection items, after
//   a barrier behind their last reader), two into the 32-KB staging region -- and every 1-KB fragment read feeds four MFMAs.
//   768 KB of LDS reads and six barriers per item.  Q never leaves the wave: packed bf16 in 32 registers until its pass, then
//   through the wave's 4-KB staging tile into the flash loop's operand layout.  K and V live in LDS only: no HBM / L2 round trip.
// (Round 3's form of the operator, attn_fused.hip -- weight slices in registers split by FEATURE across the waves, X streamed through
//  an LDS tile every wave read completely, Q through an L2 scratch -- was LDS-read-bound, 19 us of a 32-us decoder item; removed in
//  round 6 together with its A/B switch.)
// Tp = 64 m <= 512 (eight waves x two query blocks of the 16 slots; round 6: shorter windows leave slots empty, see tokbase); longer chunk
// lengths take eend_inproj_heads_bf16 + eend_attn_causal_bf16.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int KB = 64;
constexpr int TILE = KB * 128;            // one [64][64] bf16 tile
constexpr int NW = 8;
constexpr int OSTG = 32 * 128;            // per-wave staging: 32 rows x 128 B (Q in, O out)
constexpr int TP = 512;
constexpr int NT = TP / KB;               // 8 key tiles
constexpr int WITEM = 16384;              // one weight item: 16 fragments of 1 KB (2 feature fragments x 8 k-steps)
constexpr int NITEM = 6;                  // Q (features 0-31, 32-63), K, K, V, V
constexpr int L_K = 0, L_V = NT * TILE, L_X = 2 * NT * TILE;
constexpr int SMEM = L_X + NW * OSTG;     // 160 KB

typedef __attribute__((address_space(3))) char lds_char;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

DEV int swap23(int r) { return (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1); }

DEV u32x2 pack_bf16x4(const f32x4 v) {
    bf16x4 o;
    o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
    return __builtin_bit_cast(u32x2, o);
}

// weight packing, one thread per 16 bytes: [head h][item n][fragment p = ks*2 + hf][lane (f = l & 15, g = l >> 4)][8]
//   = W_in[t*256 + h*64 + ((n & 1)*2 + hf)*16 + f][ks*32 + g*8 + e],  t = n >> 1 (0 q, 1 k, 2 v)
__global__ void inproj_attn_pack_kernel(const _Float16* __restrict__ W, _Float16* __restrict__ out) {
    const int total = 4 * NITEM * 1024;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int h = t / (NITEM * 1024), r = t - h * (NITEM * 1024), n = r >> 10, w = r & 1023;
        const int pfrag = w >> 6, l = w & 63, f = l & 15, g = l >> 4, ks = pfrag >> 1, hf = pfrag & 1;
        const _Float16* src = W + (size_t)((n >> 1) * 256 + h * 64 + ((n & 1) * 2 + hf) * 16 + f) * 256 + ks * 32 + g * 8;
        _Float16* dst = out + (size_t)t * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = src[e];
    }
}

// Perf-study build (-DEEND_AS_TRACE, tools/attn_stream_trace.py): s_memtime stamps of wave 0 of every workgroup, first 6 items
#ifdef EEND_AS_TRACE
__device__ unsigned long long g_as_trace[256 * 6 * 12];
#define AS_STAMP(k) do { ts[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AS_STAMP(k) do {} while (0)
#endif

// LONG (round 6, windows of more than 512 frames): an item is a (sequence, head, query group, key group) of the pair table in the
// parameters -- groups of 512 frames, the last one shorter.  A diagonal item (key group = query group) is the item of the short form on
// that group's rows and writes its rows of O; an off-diagonal item projects Q from the query group's rows, then K / V from the key
// group's (the wave's X fragments are loaded a second time between the Q and the K items), runs the same flash passes with the causal
// limit shifted by the distance of the groups and writes its normalised partial result into a scratch slot.  Every item also leaves the
// log2 of its softmax denominators; attn_long_combine_kernel weighs the partial rows of a query group with them.
// TRAIN (round 6, windows up to 512 frames): the training forward of the same operator -- what eend_inproj_heads_train_bf16 +
// eend_attn_causal_lse_bf16 did in two launches with Q / K / V^T through HBM in between.  The key bias is applied (the saved K is the
// reference's K), Q / K / V also leave as bf16 head rows [seq][H][Tp][64] for the hand-written backward (V a second time in the other
// MFMA orientation: a lane then owns 4 features of a token), the probabilities are dropped by the counter hash of eend_dropout (element
// (a, b) = ((seq H + head) Tp + query, key); the row sums stay un-dropped) and the log2-domain log-sum-exp of every row is kept.
// FULL: Tp = 512 known at compile time (the block map is (wave, 15 - wave) and no pass is skipped: the round-4 kernel, register for register)
template <bool LONG, bool TRAIN = false, bool FULL = false>
__global__ __launch_bounds__(512)
void inproj_attn_stream_kernel(const InprojAttnParams p) {
    static_assert(!(LONG && TRAIN), "the training form covers windows up to 512 frames");
    static_assert(!(LONG && FULL), "groups of a long window have their own lengths");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem + L_K;                           // [8][64 keys][128 B]
    char* Vs = smem + L_V;                           // [8][64 d][128 B]; before that: weight items 0..3
    char* Xs = smem + L_X;                           // weight items 4, 5; afterwards the 8 x 4 KB Q / O staging

    // The thread index is laundered per item (and per phase) so that everything derived from it -- some hundred LDS addresses of
    // the K / V^T / staging writes -- is recomputed where it is used instead of being hoisted out of the item loop and spilled.
    int tid = threadIdx.x;
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int frow = lane & 15, fkg = lane >> 4;
    const int nsh = p.nseq * 4;
    const int nitems = LONG ? nsh * p.npairs : nsh;
    const bool xcd_map = (p.nseq & 7) == 0 && ((gridDim.x & 7) == 0 || (int)gridDim.x >= nitems);
    auto item_of = [&](int L, int& seq_, int& h_) __attribute__((always_inline)) {
        if (xcd_map) {
            const int xcd = L & 7, slot = L >> 3;
            seq_ = (slot >> 2) * 8 + xcd; h_ = slot & 3;
        } else {
            seq_ = L >> 2; h_ = L & 3;
        }
    };
    // the wave's tokens: fragments 0, 1 = query block b1, fragments 2, 3 = query block b2.  Tp = 512: (wave, 15 - wave).  Round 6, shorter
    // padded lengths (Tp = 64 m < 512, nblk = Tp / 32 real blocks): the first nblk / 2 waves take the causally balanced pairs
    // (w, nblk - 1 - w) of real blocks, the others two each of the 16 - nblk block slots beyond Tp -- their rows read as zeros (buffer
    // bounds), their K / V rows are finite and masked (key >= kv_len), their flash passes are skipped -- so that every slot of the LDS
    // tiles is written exactly once as before
    int nblk = FULL ? 16 : p.Tp >> 5;
    int b1 = FULL ? wave : (wave < (nblk >> 1) ? wave : nblk + 2 * (wave - (nblk >> 1)));
    int b2 = FULL ? 15 - wave : (wave < (nblk >> 1) ? nblk - 1 - wave : b1 + 1);
    // item geometry (LONG: set per item): first row and row count of the query / key group, causal limit and valid keys in key-group terms
    int qoff = 0, koff = 0, Tq = p.Tp, Tk = p.Tp, dl = p.mask_delay, kvl = p.kv_len, pair = 0;
    bool offd = false;
    auto tokbase = [&](int jt) __attribute__((always_inline)) { return jt < 2 ? 32 * b1 + 16 * jt : 32 * b2 + 16 * (jt - 2); };
    char* Ow = Xs + wave * OSTG;
    auto relaunder = [&]() __attribute__((always_inline)) {
        asm volatile("" : "+v"(tid));
        lane = tid & 63; frow = lane & 15; fkg = lane >> 4;
    };

    // X rows of the wave's 64 tokens, row-major as requested (each load instruction covers two 512-byte rows; the fragment-shaped
    // request, 16 rows x 64 bytes per instruction, kept the address unit busy for 8.7 us per item in the s_memtime trace).  They are
    // requested at the top of an item BEFORE the barrier that ends the previous one: a wave that has finished its two flash passes
    // has its rows in flight while it waits for the others (and nothing is carried in registers around the loop).
    u32x4 xr[4][8];
    auto request_x = [&](int seq_, auto KEYS) __attribute__((always_inline)) {
        constexpr bool keys = decltype(KEYS)::value;          // the key group's rows of an off-diagonal item: 64 consecutive tokens per wave
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)((const _Float16*)p.X + ((size_t)seq_ * p.Tp + (keys ? koff : qoff)) * p.ldx), 0,
                                                                            (keys ? Tk : Tq) * p.ldx * 2, 0x00020000);      // rows beyond the group: zeros
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const int off = ((keys ? 64 * wave + 16 * jt : tokbase(jt)) + (lane >> 5)) * p.ldx * 2 + (lane & 31) * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) xr[jt][i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off + i * 2 * p.ldx * 2, 0, 0);
        }
    };

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    int seq, h;
    if constexpr (LONG) {
        pair = item / nsh;
        item_of(item - pair * nsh, seq, h);
        const int qg = p.pq[pair], kg = p.pk[pair];
        qoff = qg * TP; koff = kg * TP;
        Tq = p.Tp - qoff < TP ? p.Tp - qoff : TP;
        Tk = p.Tp - koff < TP ? p.Tp - koff : TP;
        offd = qg != kg;
        dl = p.mask_delay + (qoff - koff);
        kvl = p.kv_len - koff < Tk ? p.kv_len - koff : Tk;
        nblk = Tq >> 5;
        b1 = wave < (nblk >> 1) ? wave : nblk + 2 * (wave - (nblk >> 1));
        b2 = wave < (nblk >> 1) ? nblk - 1 - wave : b1 + 1;
    } else {
        item_of(item, seq, h);
    }
    relaunder();
    request_x(seq, std::false_type{});
    // every wave is done with K / V^T / its staging tile of the previous item (not __syncthreads: its fence would wait for
    // the loads just issued)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef EEND_AS_TRACE
    unsigned long long ts[12];
    const int tix = (item - (int)blockIdx.x) / (int)gridDim.x;
#endif
    AS_STAMP(0);

    // ================================================================== phase 1: K, V^T -> LDS, Q -> registers
    u32x2 qpk[4][4];                                 // [feature fragment][token fragment]: bf16 q[tok][ff*16 + fkg*4 .. +3]
    {
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.W + (size_t)h * NITEM * WITEM), 0,
                                                                            NITEM * WITEM, 0x00020000);
        // biases before the weight requests (VMEM returns in order: a bias load behind them would wait for all 96 KB).  The key
        // bias is dropped: q . b_k is the same for every key of a query and cancels in the softmax.
        f32x4 bq4[4];
        float bv1[4];
        f32x4 bk4[TRAIN ? 4 : 1], bv4[TRAIN ? 4 : 1];
#pragma unroll
        for (int ff = 0; ff < 4; ++ff) {
            const float4 t4 = *(const float4*)(p.bias + h * 64 + ff * 16 + fkg * 4);
            bq4[ff] = f32x4{t4.x, t4.y, t4.z, t4.w};
            bv1[ff] = p.bias[512 + h * 64 + ff * 16 + frow];
            if constexpr (TRAIN) {
                const float4 k4 = *(const float4*)(p.bias + 256 + h * 64 + ff * 16 + fkg * 4);
                const float4 v4 = *(const float4*)(p.bias + 512 + h * 64 + ff * 16 + fkg * 4);
                bk4[ff] = f32x4{k4.x, k4.y, k4.z, k4.w};
                bv4[ff] = f32x4{v4.x, v4.y, v4.z, v4.w};
            }
        }
        // the head's six weight items: this wave moves pieces 2 wave, 2 wave + 1 of each
#pragma unroll
        for (int n = 0; n < NITEM; ++n) {
            char* dst = n < 4 ? Vs + n * WITEM : Xs + (n - 4) * WITEM;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_char*)(dst + (wave * 2 + i) * 1024), 16, lane * 16,
                                                         n * WITEM + (wave * 2 + i) * 1024, 0, 0);
        }
        // the X rows are turned into operand fragments x[jt][ks] = X[tok][ks*32 + fkg*8 .. +8] through a wave-private 8-KB tile in the
        // (still unused) K region: 16 rows x 32 chunks of 16 bytes, chunk index XORed with the row so that both the row-major
        // writes and the fragment reads are conflict-free.  No barrier: a wave only touches its own tile.
        f16x8 x[4][8];
        auto rows_to_fragments = [&]() __attribute__((always_inline)) {
            char* xt = Ks + wave * 8192;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = 2 * i + (lane >> 5);
                    *(u32x4*)(xt + r * 512 + (((lane & 31) ^ r) << 4)) = xr[jt][i];
                }
                wave_lds_sync();
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) x[jt][ks] = __builtin_bit_cast(f16x8, *(const u32x4*)(xt + frow * 512 + (((ks * 4 + fkg) ^ frow) << 4)));
                wave_lds_sync();
            }
        };
        rows_to_fragments();
        sfor<NITEM>([&](auto N) __attribute__((always_inline)) {
            constexpr int n = decltype(N)::value, kind = n >> 1, ffb = (n & 1) * 2;      // kind 0 q, 1 k, 2 v
            if constexpr (LONG && n == 2) {
                // off-diagonal item: Q is done with the query rows; the key group's rows take their place (the K region is still this
                // wave's to use: the first K rows are written behind the barrier below)
                if (offd) {
                    relaunder();
                    request_x(seq, std::true_type{});
                    rows_to_fragments();
                }
            }
            // this wave's pieces of item n have landed (its X rows are older): the younger requests are 2 (5 - n) pieces
            __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * (NITEM - 1 - n)));
            __builtin_amdgcn_s_barrier();
            AS_STAMP(1 + n);
            relaunder();
            const char* wi = smem + lane * 16 + (n < 4 ? L_V + n * WITEM : L_X + (n - 4) * WITEM);
            f32x4 acc[2][4];
            f32x4 accv[TRAIN && kind == 2 ? 2 : 1][4];            // TRAIN: V a second time with the token as the lane's column (head rows for the backward)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (kind == 2) b4 = f32x4{bv1[ffb + hf], bv1[ffb + hf], bv1[ffb + hf], bv1[ffb + hf]};      // V^T: the feature is the lane's column
                else if constexpr (kind == 0) b4 = bq4[ffb + hf];
                else if constexpr (TRAIN) b4 = bk4[ffb + hf];
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    acc[hf][jt] = b4;
                    if constexpr (TRAIN && kind == 2) accv[hf][jt] = bv4[ffb + hf];
                }
            }
            // 16 fragments, 4 MFMAs each; fragment reads run PD ahead in a rotation of NB registers, pinned per fragment pair
            // (left to itself the scheduler hoists every read of the item to its top and the register file overflows)
            constexpr int NB = 8, PD = 4;
            f16x8 wf[NB];
            sfor<PD>([&](auto Q) __attribute__((always_inline)) { wf[decltype(Q)::value] = *(const f16x8*)(wi + decltype(Q)::value * 1024); });
            sfor<8>([&](auto KS) __attribute__((always_inline)) {
                constexpr int ks = decltype(KS)::value;
                sfor<2>([&](auto HF) __attribute__((always_inline)) {
                    constexpr int hf = decltype(HF)::value, pi = ks * 2 + hf;
                    const f16x8 w = wf[pi % NB];
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) {
                        if constexpr (kind == 2) acc[hf][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x[jt][ks], w, acc[hf][jt], 0, 0, 0);    // rows = key
                        else acc[hf][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x[jt][ks], acc[hf][jt], 0, 0, 0);                        // rows = d
                        if constexpr (TRAIN && kind == 2) accv[hf][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x[jt][ks], accv[hf][jt], 0, 0, 0);
                    }
                    if constexpr (pi + PD < 16) wf[(pi + PD) % NB] = *(const f16x8*)(wi + (pi + PD) * 1024);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            // lane holds rows fkg*4 .. +3 of column frow of each 16 x 16 block
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    const u32x2 v = pack_bf16x4(acc[hf][jt]);
                    if constexpr (TRAIN) {
                        // bf16 head rows for the backward: 4 features of a token per lane (block slots beyond Tp hold no frames)
                        const int tok = tokbase(jt) + frow;
                        if (tok < p.Tp) {
                            __bf16* dst = (__bf16*)(kind == 0 ? p.Qh : kind == 1 ? p.Kh : p.Vh) + (((size_t)(seq * 4 + h) * p.Tp + tok) * 64 + (ffb + hf) * 16 + fkg * 4);
                            if constexpr (kind == 2) *(u32x2*)dst = pack_bf16x4(accv[hf][jt]);
                            else *(u32x2*)dst = v;
                        }
                    }
                    if constexpr (kind == 0) {
                        qpk[ffb + hf][jt] = v;
                    } else if constexpr (kind == 1) {
                        const int key = (LONG && offd ? 64 * wave + 16 * jt : tokbase(jt)) + frow, d = (ffb + hf) * 16 + fkg * 4;
                        *(u32x2*)(Ks + (key >> 6) * TILE + swz128(key & 63, d >> 3) + (d & 7) * 2) = v;
                    } else {
                        const int d = (ffb + hf) * 16 + frow, key = (LONG && offd ? 64 * wave + 16 * jt : tokbase(jt)) + fkg * 4;
                        *(u32x2*)(Vs + (key >> 6) * TILE + swz128(d, (key & 63) >> 3) + (key & 7) * 2) = v;
                    }
                }
        });
        AS_STAMP(7);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");    // K, V^T complete in LDS
        AS_STAMP(8);
    }

    // ================================================================== phase 2: the flash loop (attn_full.hip, LAZY)
    relaunder();
    const int lq = lane & 31, hi = lane >> 5;
    const int krow = swap23(lq);

    bf16x8 qf[4];
    f32x16 oT[2];
    f32x16 mneg;
    float l_run;
    int qw0, q;

    auto begin_pass = [&](int qb, auto JT0) __attribute__((always_inline)) {
        constexpr int jt0 = decltype(JT0)::value;
        qw0 = qb * 32;
        q = qw0 + lq;
        // Q of the pass's 32 queries: registers -> the wave's staging tile ([query][64 d] bf16 rows) -> operand layout
#pragma unroll
        for (int jl = 0; jl < 2; ++jl)
#pragma unroll
            for (int ff = 0; ff < 4; ++ff) {
                const int row = jl * 16 + frow;
                *(u32x2*)(Ow + row * 128 + (((ff * 2 + (fkg >> 1)) ^ (row & 7)) << 4) + (fkg & 1) * 8) = qpk[ff][jt0 + jl];
            }
        wave_lds_sync();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = __builtin_bit_cast(bf16x8, *(const u32x4*)(Ow + lq * 128 + (((ks * 2 + hi) ^ (lq & 7)) << 4)));
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < 16; ++i) { oT[0][i] = 0.f; oT[1][i] = 0.f; mneg[i] = 0.f; }
        l_run = 0.f;
    };
    auto tile = [&](int j) __attribute__((always_inline)) {
        const int key0 = j * KB;
        const char* kb_ = Ks + j * TILE;
        const char* vb_ = Vs + j * TILE;
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(kb_ + swz128(kb * 32 + krow, ks * 2 + hi));
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? mneg : s[kb], 0, 0, 0);
            }
        const int wlim = qw0 + dl < kvl - 1 ? qw0 + dl : kvl - 1;
        if (key0 + KB - 1 > wlim) {
            const int lim = q + dl < kvl - 1 ? q + dl : kvl - 1;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = key0 + kb * 32 + (i & 7) + 8 * hi + 16 * (i >> 3);
                    if (key > lim) s[kb][i] = -INFINITY;
                }
        }
        float tmax = s[0][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) tmax = __builtin_fmaxf(tmax, s[0][i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) tmax = __builtin_fmaxf(tmax, s[1][i]);
        tmax = wave_xor_max(tmax, 32);
        // the reference only moves when a row outgrows it by 2^8 (or, on the first tile, sits far below it)
        const bool move = tmax > 8.0f || (j == 0 && tmax < -8.0f);
        if (__builtin_amdgcn_ballot_w64(move) != 0) {
            float d = j == 0 ? tmax : __builtin_fmaxf(tmax, 0.f);
            d = d == -INFINITY ? 0.f : d;
            const float alpha = __builtin_amdgcn_exp2f(-d);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                oT[0][i] *= alpha; oT[1][i] *= alpha;
                s[0][i] -= d; s[1][i] -= d;
                mneg[i] -= d;
            }
        }
        float lsum0 = 0.f, lsum1 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s[0][i] = __builtin_amdgcn_exp2f(s[0][i]);
            s[1][i] = __builtin_amdgcn_exp2f(s[1][i]);
            lsum0 += s[0][i];
            lsum1 += s[1][i];
        }
        l_run += lsum0 + lsum1;
        if constexpr (TRAIN) {                         // dropout of the probabilities (the row sum stays un-dropped)
            if (p.drop.thresh24) {
                const unsigned da = (unsigned)((seq * 4 + h) * p.Tp + q);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        s[kb][i] = drop_apply(p.drop, s[kb][i], da, (unsigned)(key0 + kb * 32 + (i & 7) + 8 * hi + 16 * (i >> 3)));
            }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 pf;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) pf[jj] = (__bf16)s[kb][kk * 8 + jj];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 vf = *(const bf16x8*)(vb_ + swz128(db * 32 + lq, kb * 4 + kk * 2 + hi));
                    oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oT[db], 0, 0, 0);
                }
            }
    };
    auto run_pass = [&](int qb, auto JT0) __attribute__((always_inline)) {
        begin_pass(qb, JT0);
        int last_key = qw0 + 31 + dl;
        last_key = last_key < kvl - 1 ? last_key : kvl - 1;
        const int jend = last_key < 0 ? 0 : last_key / KB + 1;
        for (int j = 0; j < jend; ++j) tile(j);
        // O[q][d] = O^T / l: stage the wave's 32 x 64 f16 tile, then 128-byte rows to HBM
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        // (LONG: a row with no key of this key group in reach contributes nothing)
        const float inv = LONG ? (l_tot > 0.f ? 1.0f / l_tot : 0.f) : 1.0f / l_tot;
        if constexpr (LONG) {
            if (hi == 0) p.lse[((size_t)(pair * p.nseq + seq) * 4 + h) * TP + q] = l_tot > 0.f ? __builtin_amdgcn_logf(l_tot) - mneg[0] : -1e30f;
        }
        if constexpr (TRAIN) {                         // log2-domain log-sum-exp of the row, for the backward
            if (hi == 0) p.lse[(size_t)(seq * 4 + h) * p.Tp + q] = __builtin_amdgcn_logf(l_tot) - mneg[0];
        }
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f16x4 o;
                o[0] = to_f16_sat(oT[db][g * 4 + 0] * inv);
                o[1] = to_f16_sat(oT[db][g * 4 + 1] * inv);
                o[2] = to_f16_sat(oT[db][g * 4 + 2] * inv);
                o[3] = to_f16_sat(oT[db][g * 4 + 3] * inv);
                *(f16x4*)(Ow + lq * 128 + (((db * 4 + g) ^ (lq & 7)) << 4) + hi * 8) = o;
            }
        wave_lds_sync();
        _Float16* __restrict__ Og = (_Float16*)p.O + ((size_t)seq * p.Tp + qoff + qw0) * p.ldo + h * 64;
        int ldo = p.ldo;
        if constexpr (LONG) {
            if (offd) { Og = (_Float16*)p.Opart + ((size_t)(p.pslot[pair] * p.nseq + seq) * TP + qw0) * 256 + h * 64; ldo = 256; }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3), ch = lane & 7;
            const f16x8 v = __builtin_bit_cast(f16x8, *(const u32x4*)(Ow + row * 128 + ((ch ^ (row & 7)) << 4)));
            *(f16x8*)(Og + (size_t)row * ldo + ch * 8) = v;
        }
        wave_lds_sync();
    };
    if (FULL || b2 < nblk) run_pass(b2, std::integral_constant<int, 2>{});      // (wave-uniform: a pass has no workgroup barrier)
    AS_STAMP(9);
    if (FULL || b1 < nblk) run_pass(b1, std::integral_constant<int, 0>{});
    AS_STAMP(10);
    AS_STAMP(11);
#ifdef EEND_AS_TRACE
    if (tix < 6 && threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) g_as_trace[((size_t)blockIdx.x * 6 + tix) * 12 + k] = ts[k];
    }
#endif                                 // every wave is done with K / V^T / its staging tile before the next item
    }
}

// O rows of the query groups that have off-diagonal items: O = sum_i 2^(lse_i - lse) O_i over the group's items (the diagonal item's rows are
// in O itself).  Half a wave per row: lane -> (head, 8 features).
__global__ __launch_bounds__(256)
void attn_long_combine_kernel(const InprojAttnParams p) {
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5);           // frame within the sequence
    const int seq = blockIdx.y;
    if (row >= p.Tp) return;
    const int qg = row / TP, ql = row - qg * TP, l32 = threadIdx.x & 31, h = l32 >> 3, d8 = l32 & 7;
    int idx[8], n = 0, pd = -1;
    for (int i = 0; i < p.npairs; ++i)
        if (p.pq[i] == qg) {
            if (p.pk[i] == qg) pd = i;
            else if (n < 8) idx[n++] = i;
        }
    if (n == 0 || pd < 0) return;
    _Float16* o = (_Float16*)p.O + ((size_t)seq * p.Tp + row) * p.ldo + h * 64 + d8 * 8;
    const float ld = p.lse[((size_t)(pd * p.nseq + seq) * 4 + h) * TP + ql];
    float m = ld, lk[8];
    for (int i = 0; i < n; ++i) {
        lk[i] = p.lse[((size_t)(idx[i] * p.nseq + seq) * 4 + h) * TP + ql];
        m = __builtin_fmaxf(m, lk[i]);
    }
    float wd = __builtin_amdgcn_exp2f(ld - m), wsum = wd;
    const f16x8 od = *(const f16x8*)o;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = wd * (float)od[e];
    for (int i = 0; i < n; ++i) {
        const float w = __builtin_amdgcn_exp2f(lk[i] - m);
        wsum += w;
        const f16x8 v = *(const f16x8*)((const _Float16*)p.Opart + ((size_t)(p.pslot[idx[i]] * p.nseq + seq) * TP + ql) * 256 + h * 64 + d8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += w * (float)v[e];
    }
    const float inv = 1.0f / wsum;
    f16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = to_f16_sat(acc[e] * inv);
    *(f16x8*)o = r;
}

}  // namespace

#ifdef EEND_AS_TRACE
extern "C" int eend_debug_attn_stream_trace(void* dst, void* stream) {
    return hipMemcpyFromSymbolAsync(dst, HIP_SYMBOL(g_as_trace), sizeof(g_as_trace), 0, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? 0 : -2;
}
#endif

long eend_inproj_attn_packed_nelems() { return 4L * NITEM * WITEM / 2; }

int eend_launch_inproj_attn_pack(const void* W, void* out, hipStream_t stream) {
    if (!W || !out) return EEND_EINVAL;
    hipLaunchKernelGGL(inproj_attn_pack_kernel, dim3(96), dim3(256), 0, stream, (const _Float16*)W, (_Float16*)out);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

// p.W = the packed weights (eend_launch_inproj_attn_pack)
int eend_launch_inproj_attn_stream(const InprojAttnParams& p, hipStream_t stream) {
    if (p.Tp <= 0 || p.Tp > TP || (p.Tp & 63) || (p.ldo & 7) || (p.ldx & 7) || p.H != 4 || p.nseq <= 0 || !p.X || !p.W || !p.bias || !p.O) return EEND_EINVAL;
    static EendOncePerDevice attr_once;
    static EendOncePerDevice attr_once_full;
    if (!eend_set_dynamic_lds(attr_once, (const void*)inproj_attn_stream_kernel<false, false, false>, SMEM)) return EEND_ELAUNCH;
    if (!eend_set_dynamic_lds(attr_once_full, (const void*)inproj_attn_stream_kernel<false, false, true>, SMEM)) return EEND_ELAUNCH;
    int n_cu = eend_cu_count() & ~31;                // multiple of 32: a persistent workgroup keeps its head (and its XCD)
    if (n_cu <= 0) n_cu = 32;
    const int nitems = p.nseq * 4;
    InprojAttnParams q = p;
    q.npairs = 0; q.Opart = nullptr; q.lse = nullptr;
    if (p.Tp == TP) hipLaunchKernelGGL((inproj_attn_stream_kernel<false, false, true>), dim3(nitems < n_cu ? nitems : n_cu), dim3(512), SMEM, stream, q);
    else hipLaunchKernelGGL((inproj_attn_stream_kernel<false, false, false>), dim3(nitems < n_cu ? nitems : n_cu), dim3(512), SMEM, stream, q);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

// training forward (Tp = 64 m <= 512): p.Qh / p.Kh / p.Vh bf16 head rows and p.lse [nseq][4][Tp] out, p.drop on the probabilities
int eend_launch_inproj_attn_train(const InprojAttnParams& p, hipStream_t stream) {
    if (p.Tp <= 0 || p.Tp > TP || (p.Tp & 63) || (p.ldo & 7) || (p.ldx & 7) || p.H != 4 || p.nseq <= 0 || !p.X || !p.W || !p.bias || !p.O || !p.Qh || !p.Kh ||
        !p.Vh || !p.lse || p.kv_len < 1 || (long)p.nseq * 4 * p.Tp >= (1L << 31))
        return EEND_EINVAL;
    static EendOncePerDevice attr_once;
    static EendOncePerDevice attr_once_full;
    if (!eend_set_dynamic_lds(attr_once, (const void*)inproj_attn_stream_kernel<false, true, false>, SMEM)) return EEND_ELAUNCH;
    if (!eend_set_dynamic_lds(attr_once_full, (const void*)inproj_attn_stream_kernel<false, true, true>, SMEM)) return EEND_ELAUNCH;
    int n_cu = eend_cu_count() & ~31;
    if (n_cu <= 0) n_cu = 32;
    const int nitems = p.nseq * 4;
    InprojAttnParams q = p;
    q.npairs = 0; q.Opart = nullptr;
    if (p.Tp == TP) hipLaunchKernelGGL((inproj_attn_stream_kernel<false, true, true>), dim3(nitems < n_cu ? nitems : n_cu), dim3(512), SMEM, stream, q);
    else hipLaunchKernelGGL((inproj_attn_stream_kernel<false, true, false>), dim3(nitems < n_cu ? nitems : n_cu), dim3(512), SMEM, stream, q);
    return hipGetLastError() == hipSuccess ? EEND_OK : EEND_ELAUNCH;
}

// Windows of more than 512 frames (Tp = 64 m <= 512 * EEND_ATTN_LONG_MAX_GROUPS): the (query group, key group) items a mask reaches, the
// off-diagonal ones first (they are the long ones).  Returns the number of items, 0 if the shape is not covered; *noff = scratch slots.
static int attn_long_pairs(int Tp, int mask_delay, int kv_len, unsigned char* pq, unsigned char* pk, unsigned char* pslot, int* noff) {
    if (Tp <= TP || (Tp & 63) || kv_len < 1 || kv_len > Tp || mask_delay < 0) return 0;
    const int ng = (Tp + TP - 1) / TP;
    int n = 0, slots = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int qg = 0; qg < ng; ++qg)
            for (int kg = 0; kg < ng; ++kg) {
                if ((pass == 0) == (qg == kg)) continue;
                if (qg != kg) {
                    const long qlast = (long)qg * TP + (Tp - qg * TP < TP ? Tp - qg * TP : TP) - 1;
                    if ((long)kg * TP >= kv_len || (long)kg * TP > qlast + mask_delay) continue;
                    int per_q = 0;
                    for (int i = 0; i < n; ++i) per_q += pq[i] == qg;
                    if (per_q >= 8) return 0;                       // the combine kernel's list
                }
                if (n >= EEND_ATTN_LONG_MAX_PAIRS) return 0;
                pq[n] = (unsigned char)qg; pk[n] = (unsigned char)kg; pslot[n] = (unsigned char)(qg != kg ? slots++ : 0);
                ++n;
            }
    *noff = slots;
    return n;
}

int eend_inproj_attn_long_scratch(int nseq, int Tp, int mask_delay, int kv_len, long* part_elems, long* lse_elems) {
    unsigned char a[EEND_ATTN_LONG_MAX_PAIRS], b[EEND_ATTN_LONG_MAX_PAIRS], c[EEND_ATTN_LONG_MAX_PAIRS];
    int noff = 0;
    const int n = attn_long_pairs(Tp, mask_delay, kv_len, a, b, c, &noff);
    if (n == 0 || nseq <= 0) return EEND_EINVAL;
    *part_elems = (long)noff * nseq * TP * 256;
    *lse_elems = (long)n * nseq * 4 * TP;
    return EEND_OK;
}

int eend_launch_inproj_attn_long(const InprojAttnParams& p0, hipStream_t stream) {
    InprojAttnParams p = p0;
    if ((p.ldo & 7) || (p.ldx & 7) || p.H != 4 || p.nseq <= 0 || !p.X || !p.W || !p.bias || !p.O || !p.lse) return EEND_EINVAL;
    if (p.mask_delay > (1 << 24)) p.mask_delay = 1 << 24;
    int noff = 0;
    p.npairs = attn_long_pairs(p.Tp, p.mask_delay, p.kv_len, p.pq, p.pk, p.pslot, &noff);
    if (p.npairs == 0 || (noff > 0 && !p.Opart)) return EEND_EINVAL;
    static EendOncePerDevice attr_once;
    if (!eend_set_dynamic_lds(attr_once, (const void*)inproj_attn_stream_kernel<true, false>, SMEM)) return EEND_ELAUNCH;
    int n_cu = eend_cu_count() & ~31;
    if (n_cu <= 0) n_cu = 32;
    const long nitems = (long)p.nseq * 4 * p.npairs;
    if (nitems > (1L << 30)) return EEND_EINVAL;
    hipLaunchKernelGGL((inproj_attn_stream_kernel<true, false>), dim3(nitems < n_cu ? (int)nitems : n_cu), dim3(512), SMEM, stream, p);
    if (hipGetLastError() != hipSuccess) return EEND_ELAUNCH;
    if (noff > 0) {
        hipLaunchKernelGGL(attn_long_combine_kernel, dim3((p.Tp + 7) / 8, p.nseq), dim3(256), 0, stream, p);
        if (hipGetLastError() != hipSuccess) return EEND_ELAUNCH;
    }
    return EEND_OK;
}
