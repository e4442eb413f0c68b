// Workgroup-level pieces shared by the dense and the factored FieldConv kernels: the LDS slab
// hand-off between the per-wavefront gather phase and the MFMA contraction, the contraction
// itself, and the fixed-order combination of the k-partials.
#pragma once
#include <stdlib.h>
#include "fc_common.hpp"
#include "fc_kernels.hpp"

namespace fc {

// Contraction geometry of one pass: out[M x 16 vertices] = Wpk[M x K] * slab[16 vertices x K].
//   forward : M = O (output channels),  K = R*I
//   backward: M = I (input channels),   K = R*O
//
// Two arithmetic modes for the contraction (same results to fp32 rounding level, see mma_slab_split):
//   fp32   v_mfma_f32_16x16x4_f32 on fp32 operands; k blocks of 16, slab = 2 planes of floats
//   split  v_mfma_f32_16x16x32_f16 on operands split into two halves (hi + lo); k blocks of 32,
//          slab = one row per vertex holding the four planes (re_hi, re_lo, im_hi, im_lo) interleaved
//          in 16-byte fragments: [k / 8][plane][k % 8] halves, so that a plane is an immediate offset.
//   half   the same kernels with the lo halves dropped (planes re_hi, im_hi): a reduced-precision
//          mode (relative error ~2^-11 per operand), selected with FC_MFMA=f16 and reported separately.
// The contraction index is k = r * KI + c (ring r, channel c).
struct MmaGeom {
    int MP;     // ceil16(M)
    int KI;     // channel stride inside k: the channel count (fp32) or ceil8 of it (split)
    int KP;     // k entries per row: ceil16(R*KI) (fp32) or ceil32(R*KI) (split)
    int KS;     // LDS slab row stride: floats per plane row, KP + 8 (fp32); halves per vertex row, 4*KP + 8 (split)
    int NMT;    // MP / 16 output tiles
    int NKP;    // k partitions (wavefronts per output tile)
    int KST;    // k blocks: KP / 16 (fp32) or KP / 32 (split)
    int split;  // halves per operand: 0 fp32 mode, 2 split mode, 1 reduced-precision half mode
};

__host__ __device__ inline MmaGeom make_mma_geom(int M, int R, int channels, int halves = 0) {
    MmaGeom g;
    const bool split = halves != 0;
    const int kblock = split ? 32 : 16;
    g.split = halves;
    g.MP = round_up(M, 16);
    g.KI = split ? round_up(channels, 8) : channels;
    g.KP = round_up(R * g.KI, kblock);
    g.KS = split ? 2 * halves * g.KP + 8 : slab_stride(g.KP);      // all: 16-byte fragment reads of 16 rows hit distinct banks
    g.NMT = g.MP / 16;
    g.KST = g.KP / kblock;
    g.NKP = kWaves / g.NMT;
    if (g.NKP > g.KST) g.NKP = g.KST;
    if (g.NKP < 1) g.NKP = 1;
    return g;
}

// Floats in the packed filter image of one contraction (F frequencies, M rows, K entries per row).
//   fp32 : [F][2 planes re,im][MP][KP] floats
//   split: [MP] inverse row scales (floats), then [F][4 planes re_hi,re_lo,im_hi,im_lo][KP/32 k blocks][MP][32] halves
//   half : the same with the 2 planes re_hi, im_hi
__host__ __device__ inline size_t packed_image_floats(int M, int R, int channels, int F, int halves) {
    const MmaGeom g = make_mma_geom(M, R, channels, halves);
    return halves ? (size_t)g.MP + (size_t)F * halves * g.MP * g.KP : (size_t)F * 2 * g.MP * g.KP;
}

// Floats of one LDS slab buffer (16 vertices).
__host__ __device__ inline int slab_floats(const MmaGeom& g) { return g.split ? kTile * g.KS / 2 : 2 * kTile * g.KS; }

// Halves per operand of a launch: 2 (split mode, the default: fc_dims::mode = FC_MFMA_SPLIT_F16), 0 (FC_MFMA_F32: fp32 MFMA),
// 1 (FC_MFMA_F16: reduced precision).  The mode travels with every call in its dims: the library keeps no arithmetic state.
inline int halves_of(const fc_dims* d) { return d->mode == FC_MFMA_F32 ? 0 : d->mode == FC_MFMA_F16 ? 1 : 2; }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// The packed filter is read through a buffer descriptor: the wave-uniform part of every address
// (frequency, plane, k block) travels in the scalar offset and the per-lane part in ONE 32-bit
// VGPR, instead of a 64-bit VGPR pointer per plane that hipcc would otherwise keep (and spill)
// across the whole kernel.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buffer_load16(rsrc_t r, int voffset_bytes, int soffset_bytes) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voffset_bytes, soffset_bytes, 0));
}
typedef __attribute__((address_space(3))) const u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) _Float16 lds_f16;
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;        // ds_read_b64_tr_b16 operand
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const u32x2 lds_u32x2;

// One slab (one angular frequency) through the MFMA pipe for this wavefront's (output tile mt,
// k partition kp): acc (re, im) += W[mt*16.., k] * slab[vertex, k], complex via four real products
// on two accumulators.  wimg / f_bytes: descriptor of the packed image and byte offset of this
// frequency's two planes [2][MP][KP] floats.  Filter fragments come straight from L2 as 16 rows x 64 B
// per instruction and are double-buffered in registers; slab fragments are conflict-free float4 LDS reads.
__device__ __forceinline__ void mma_slab(rsrc_t wimg, int f_bytes, const float* sre, const float* sim, const MmaGeom& g,
                                         int mt, int kp, int lane, f32x4& acc_re, f32x4& acc_im) {
    const int fr = lane & 15, fq = lane >> 4;
    const int wplane = g.MP * g.KP * 4;                 // bytes
    const int wv = ((mt * 16 + fr) * g.KP + 4 * fq) * 4;
    const float* bre = sre + fr * g.KS + 4 * fq;
    const float* bim = sim + fr * g.KS + 4 * fq;
    auto ldw = [&](int plane, int kb) { return __builtin_bit_cast(float4, buffer_load16(wimg, wv, f_bytes + plane * wplane + 64 * kb)); };
    float4 wr = ldw(0, kp), wi = ldw(1, kp);
    for (int kb = kp; kb < g.KST; kb += g.NKP) {
        const bool more = kb + g.NKP < g.KST;               // (wave-uniform: the last block requests nothing -- a clamped re-read
        float4 wr_n = wr, wi_n = wi;                        //  of it was a third of this wavefront's fragment loads)
        if (more) { wr_n = ldw(0, kb + g.NKP); wi_n = ldw(1, kb + g.NKP); }
        const float4 br = *reinterpret_cast<const float4*>(bre + 16 * kb);
        const float4 bi = *reinterpret_cast<const float4*>(bim + 16 * kb);
        // re += Wre*Sre - Wim*Sim ; im += Wim*Sre + Wre*Sim
        acc_re = mfma16(wr.x, br.x, acc_re); acc_im = mfma16(wi.x, br.x, acc_im);
        acc_re = mfma16(-wi.x, bi.x, acc_re); acc_im = mfma16(wr.x, bi.x, acc_im);
        acc_re = mfma16(wr.y, br.y, acc_re); acc_im = mfma16(wi.y, br.y, acc_im);
        acc_re = mfma16(-wi.y, bi.y, acc_re); acc_im = mfma16(wr.y, bi.y, acc_im);
        acc_re = mfma16(wr.z, br.z, acc_re); acc_im = mfma16(wi.z, br.z, acc_im);
        acc_re = mfma16(-wi.z, bi.z, acc_re); acc_im = mfma16(wr.z, bi.z, acc_im);
        acc_re = mfma16(wr.w, br.w, acc_re); acc_im = mfma16(wi.w, br.w, acc_im);
        acc_re = mfma16(-wi.w, bi.w, acc_re); acc_im = mfma16(wr.w, bi.w, acc_im);
        wr = wr_n;
        wi = wi_n;
    }
}

// ---- split mode ------------------------------------------------------------------------------
// A fp32 value v is carried as two halves, hi = half(v*s), lo = half(v*s - hi), with s a power of two
// that brings the largest magnitude of the row (one vertex of the slab / one output row of the filter)
// to [2^13, 2^14): hi + lo reproduces v*s to 2^-22 relative to the row maximum whatever the half
// denormal behaviour of the matrix pipe.  The product (hi_w + lo_w)(hi_s + lo_s) is accumulated in
// fp32 as hi*hi + hi*lo + lo*hi (the dropped lo*lo term is 2^-22 relative): three
// v_mfma_f32_16x16x32_f16 per real product, 16 cycles each for 32 k, against eight 32-cycle
// v_mfma_f32_16x16x4_f32 -- 5.3x less matrix-pipe time at fp32-level accuracy.  The scales are exact
// (powers of two) and are divided out in the epilogue.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma32h(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// Power-of-two scale (and its inverse) that maps magnitude m into [2^13, 2^14); 1 for m = 0 / non-finite.
__host__ __device__ inline void split_scale(float m, float& scale, float& inv) {
    union { float f; uint32_t u; } bits;
    bits.f = m;
    const int e = (int)((bits.u >> 23) & 0xff);        // m in [2^(e-127), 2^(e-126))
    int se = 267 - e;                                   // biased exponent of 2^(13 - (e - 127))
    if (e == 0 || e == 255) se = 127;
    if (se > 253) se = 253;
    if (se < 1) se = 1;
    bits.u = (uint32_t)se << 23;
    scale = bits.f;
    bits.u = (uint32_t)(254 - se) << 23;
    inv = bits.f;
}

__device__ __forceinline__ void split_halves(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}
// The same for a complex value scaled by s: hi = (re_hi, im_hi), lo = (re_lo, im_lo).  In vector form hipcc
// emits v_cvt_pk_f16_f32 (round to nearest even) and folds the scale into the residual's packed FMA.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// The residual comes from v_fma_mixlo / mixhi_f16: lo = half(v*s - hi) with the half operand read in place and one rounding -- two
// instructions instead of two conversions back to fp32, a packed FMA and a packed conversion (six -> four vector instructions per
// complex value; bit-identical, since v*s is exact for a power-of-two s and v*s - hi is exact in fp32).
__device__ __forceinline__ void split_halves2(f32x2 v, float s, f16x2& hi, f16x2& lo) {
    const f32x2 vs = v * f32x2{s, s};
    hi = __builtin_convertvector(vs, f16x2);
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(v.x), "v"(s), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(v.y), "v"(s), "v"(hi));
    lo = __builtin_bit_cast(f16x2, l);
}
// Writing one ring of a slab row (lane = channel c, element k = r*KI + c).  Neighbouring lanes exchange
// their packed halves over DPP so that every lane stores two full dwords with ONE ds_write2_b32 instead of
// four 2-byte stores (the LDS instruction rate of the 16 wavefronts bounds this phase):
//   even lane: (re_hi[c], re_hi[c+1]) -> plane 0, (re_lo[c], re_lo[c+1]) -> plane 1
//   odd lane : (im_hi[c-1], im_hi[c]) -> plane 2, (im_lo[c-1], im_lo[c]) -> plane 3
// split_pair_offset: dword offset of the lane's first store inside the row (ring 0); the second is 4 dwords on.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
// In half mode (halves = 1) the row holds the planes (re_hi, im_hi): even lanes store the re pair, odd
// lanes the im pair, one dword each.  A ring advances the offset by halves * KI dwords.
__device__ __forceinline__ int split_pair_offset(int c, int halves) {
    return 8 * halves * (c >> 3) + 4 * halves * (c & 1) + ((c & 7) >> 1);
}
__device__ __forceinline__ void split_pair_store(lds_u32* row, int dword_offset, f16x2 hi, f16x2 lo, int lane, int halves) {
    const uint32_t h = __builtin_bit_cast(uint32_t, hi), l = __builtin_bit_cast(uint32_t, lo);
    const uint32_t hp = (uint32_t)__builtin_amdgcn_mov_dpp((int)h, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]: lane ^ 1
    const uint32_t sel = (lane & 1) ? 0x03020706u : 0x05040100u;    // odd: (partner.hi16, own.hi16); even: (own.lo16, partner.lo16)
    row[dword_offset] = __builtin_amdgcn_perm(hp, h, sel);
    if (halves == 2) {
        const uint32_t lp = (uint32_t)__builtin_amdgcn_mov_dpp((int)l, 0xB1, 0xF, 0xF, true);
        row[dword_offset + 4] = __builtin_amdgcn_perm(lp, l, sel);
    }
}

// One slab through the matrix pipe in split mode.  wimg: descriptor of the packed image, f_bytes: byte
// offset of this frequency's four planes [4][MP][KP] halves in it; sp: the slab, [16][KS] halves with
// the planes interleaved per 8-k fragment (LDS).  Lane l holds A[row l&15][k = 8(l>>4)+j] and
// B[k = 8(l>>4)+j][col l&15], j = 0..7, as one 16-byte fragment.
__device__ __forceinline__ void mma_slab_split(rsrc_t wimg, int f_bytes, const lds_f16* sp, const MmaGeom& g, int mt, int kp,
                                               int lane, f32x4& acc_re, f32x4& acc_im) {
    const int fr = lane & 15, fq = lane >> 4;
    // split images are k-block major, [plane][k block][MP rows][32 halves]: the 16 rows x 64 B a wavefront loads per
    // plane and k block are one contiguous KiB (eight full cache lines) instead of sixteen half lines
    const int wplane = g.MP * g.KP * 2;                 // bytes
    const int wv = ((mt * 16 + fr) * 32 + 8 * fq) * 2;
    const int wkb = g.MP * 64;                          // bytes per k block
    const u32x4 sign = {0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
    if (g.split == 1) {
        // reduced precision: planes (re_hi, im_hi) only, one MFMA per real product
        const lds_f16* s1 = sp + fr * g.KS + 16 * fq;   // fragment (k block 4*kb + fq, plane p) at + 64*kb + 8*p halves
        auto ldw1 = [&](int plane, int kb) { return buffer_load16(wimg, wv, f_bytes + plane * wplane + wkb * kb); };
        auto lds1 = [&](int plane, int kb) { return *reinterpret_cast<lds_u32x4*>(s1 + 64 * kb + 8 * plane); };
        u32x4 wr = ldw1(0, kp), wi = ldw1(1, kp);
        for (int kb = kp; kb < g.KST; kb += g.NKP) {
            u32x4 n_wr = wr, n_wi = wi;
            if (kb + g.NKP < g.KST) { n_wr = ldw1(0, kb + g.NKP); n_wi = ldw1(1, kb + g.NKP); }
            const u32x4 sr = lds1(0, kb);
            u32x4 si = lds1(1, kb);
            acc_re = mfma32h(wr, sr, acc_re); acc_im = mfma32h(wi, sr, acc_im);
            acc_im = mfma32h(wr, si, acc_im);
            si ^= sign;
            acc_re = mfma32h(wi, si, acc_re);
            wr = n_wr; wi = n_wi;
        }
        return;
    }
    const lds_f16* s0 = sp + fr * g.KS + 32 * fq;       // fragment (k block 4*kb + fq, plane p) at + 128*kb + 8*p halves
    auto ldw = [&](int plane, int kb) { return buffer_load16(wimg, wv, f_bytes + plane * wplane + wkb * kb); };
    auto lds = [&](int plane, int kb) { return *reinterpret_cast<lds_u32x4*>(s0 + 128 * kb + 8 * plane); };
    u32x4 wrh = ldw(0, kp), wrl = ldw(1, kp), wih = ldw(2, kp), wil = ldw(3, kp);
    for (int kb = kp; kb < g.KST; kb += g.NKP) {
        // the next block's fragments (wave-uniform branch: the last block requests nothing -- clamped re-reads of it were 60 of the 168
        // fragment loads of a slab, and a load costs the CU's memory path by the instruction)
        u32x4 n_wrh = wrh, n_wrl = wrl, n_wih = wih, n_wil = wil;
        if (kb + g.NKP < g.KST) { n_wrh = ldw(0, kb + g.NKP); n_wrl = ldw(1, kb + g.NKP); n_wih = ldw(2, kb + g.NKP); n_wil = ldw(3, kb + g.NKP); }
        // re += Wre*Sre - Wim*Sim ; im += Wim*Sre + Wre*Sim, each product = lo*hi + hi*lo + hi*hi.
        // The real-part fragments of the slab are consumed before the imaginary ones are read, and those
        // are negated in place, to keep the live register set small next to the gather accumulators.
        {
            const u32x4 srh = lds(0, kb), srl = lds(1, kb);
            acc_re = mfma32h(wrl, srh, acc_re); acc_im = mfma32h(wil, srh, acc_im);
            acc_re = mfma32h(wrh, srl, acc_re); acc_im = mfma32h(wih, srl, acc_im);
            acc_re = mfma32h(wrh, srh, acc_re); acc_im = mfma32h(wih, srh, acc_im);
        }
        {
            u32x4 sih = lds(2, kb), sil = lds(3, kb);
            acc_im = mfma32h(wrl, sih, acc_im);
            acc_im = mfma32h(wrh, sil, acc_im);
            acc_im = mfma32h(wrh, sih, acc_im);
            sih ^= sign;
            sil ^= sign;
            acc_re = mfma32h(wil, sih, acc_re);
            acc_re = mfma32h(wih, sil, acc_re);
            acc_re = mfma32h(wih, sih, acc_re);
        }
        wrh = n_wrh; wrl = n_wrl; wih = n_wih; wil = n_wil;
    }
}

// k-partials in LDS: [kp][row m][vertex v] complex, row stride kPartStride floats.  The accumulator tile
// (D layout: column = vertex = lane&15, row = 4*(lane>>4)+j) is written with one 8-byte store per
// lane and row: the 16 lanes of a row cover 32 consecutive floats (all banks once) and the +2 pad
// keeps the four lane groups of an instruction on different banks.
constexpr int kPartStride = 2 * kTile + 2;

__host__ __device__ constexpr int partial_floats(int NKP, int MP) { return NKP * MP * kPartStride; }

__device__ __forceinline__ void store_partial(float* part, const MmaGeom& g, int mt, int kp, int lane, const f32x4& acc_re,
                                              const f32x4& acc_im) {
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = mt * 16 + 4 * fq + j;
        *reinterpret_cast<float2*>(part + (size_t)(kp * g.MP + m) * kPartStride + 2 * fr) = make_float2(acc_re[j], acc_im[j]);
    }
}

// Fixed-order sum over the k partitions of entry (vertex v, row m).  The partition counts of the 48- and 64-channel layers (5 and 4) are
// unrolled -- all reads in flight, the same order of additions -- behind a wave-uniform dispatch.
template <int NKP>
__device__ __forceinline__ float2 sum_partials_n(const float* part, const MmaGeom& g, int v, int m) {
    float2 p[NKP];
#pragma unroll
    for (int q = 0; q < NKP; ++q) p[q] = *reinterpret_cast<const float2*>(part + (size_t)(q * g.MP + m) * kPartStride + 2 * v);
    float re = 0.f, im = 0.f;
#pragma unroll
    for (int q = 0; q < NKP; ++q) {
        re += p[q].x;
        im += p[q].y;
    }
    return make_float2(re, im);
}
__device__ __forceinline__ float2 sum_partials(const float* part, const MmaGeom& g, int v, int m) {
    if (g.NKP == 5) return sum_partials_n<5>(part, g, v, m);
    if (g.NKP == 4) return sum_partials_n<4>(part, g, v, m);
    if (g.NKP == 2) return sum_partials_n<2>(part, g, v, m);       // (the ring-major forward kernel's 8-wavefront workgroups)
    float re = 0.f, im = 0.f;
    for (int q = 0; q < g.NKP; ++q) {
        const float2 p = *reinterpret_cast<const float2*>(part + (size_t)(q * g.MP + m) * kPartStride + 2 * v);
        re += p.x;
        im += p.y;
    }
    return make_float2(re, im);
}

// ---- forward epilogue (include/fieldconv_hip.h: fc_epilogue) ----
struct FwdEpi {
    const float2* addend;
    const float* bias;
    float2* activated;
};
inline FwdEpi make_epi(const fc_epilogue* e) {
    FwdEpi r;
    r.addend = e ? reinterpret_cast<const float2*>(e->addend) : nullptr;
    r.bias = e ? e->modrelu_bias : nullptr;
    r.activated = e ? reinterpret_cast<float2*>(e->activated) : nullptr;
    return r;
}
// s = convolution output of entry idx (channel o): adds the residual, returns the pre-activation (what `y` receives) and
// writes modReLU of it -- the arithmetic of tangent_nonlin_fwd_kernel (fc_pointwise.hip), bit for bit
__device__ __forceinline__ float2 apply_epilogue(float2 s, size_t idx, int o, const FwdEpi& e) {
    if (e.addend) {
        const float2 a = e.addend[idx];
        s.x += a.x;
        s.y += a.y;
    }
    if (e.bias) {
        float2 act = s;
        if (!is_origin(s)) {
            const float r = sqrtf(s.x * s.x + s.y * s.y);
            const float f = fmaxf(r + e.bias[o], 0.f);
            const float k = f / r;
            act = make_float2(s.x * k, s.y * k);
        }
        e.activated[idx] = act;
    }
    return s;
}

// ---- edge split for small meshes (forward and backward data kernels) ----
// dst[idx] = sum over parts of part[p][idx], in part order (float4 per thread; the parts are part_stride complex numbers
// apart, a multiple of 2)
static __global__ void fc_sum_parts_kernel(const f32x4* __restrict__ part, f32x4* __restrict__ dst, size_t count4, size_t stride4, int parts,
                                    size_t tail_floats) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count4) return;
    f32x4 s = part[idx];
    for (int p = 1; p < parts; ++p) s += part[(size_t)p * stride4 + idx];
    if (idx + 1 < count4 || tail_floats == 0) dst[idx] = s;
    else {
        float* d = reinterpret_cast<float*>(dst + idx);
        for (size_t k = 0; k < tail_floats; ++k) d[k] = s[k];
    }
}


// the same for the forward pass with an epilogue: one complex number per thread (channel o = idx % O)
static __global__ void fc_sum_parts_epilogue_kernel(const float2* __restrict__ part, float2* __restrict__ dst, size_t count, size_t stride,
                                                    int parts, int O, const FwdEpi e) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    float2 s = part[idx];
    for (int p = 1; p < parts; ++p) {
        const float2 v = part[(size_t)p * stride + idx];
        s.x += v.x;
        s.y += v.y;
    }
    dst[idx] = apply_epilogue(s, idx, (int)(idx % (size_t)O), e);
}

// complex numbers between the partial outputs of the parts: `count` rounded up to a multiple of 2 (16-byte rows)
inline size_t part_stride(size_t count) { return ((count + 1) / 2) * 2; }

// dst (count complex numbers) = fixed-order sum of `parts` partial arrays
inline int sum_parts(const float* part, float* dst, size_t count, size_t stride, int parts, hipStream_t stream) {
    const size_t floats = count * 2, count4 = (floats + 3) / 4;
    hipLaunchKernelGGL(fc_sum_parts_kernel, dim3((unsigned)((count4 + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const f32x4*>(part), reinterpret_cast<f32x4*>(dst), count4, stride / 2, parts, floats % 4);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

inline int sum_parts_epilogue(const float* part, float* dst, size_t count, size_t stride, int parts, int O, const FwdEpi& e,
                              hipStream_t stream) {
    if (!e.addend && !e.bias) return sum_parts(part, dst, count, stride, parts, stream);
    hipLaunchKernelGGL(fc_sum_parts_epilogue_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float2*>(part), reinterpret_cast<float2*>(dst), count, stride, parts, O, e);
    return hipGetLastError() == hipSuccess ? FC_OK : FC_ERR_LAUNCH;
}

// Work items of a persistent tile kernel (no edge split): [0, nv_full) are whole 16-vertex tiles; the tiles of the last,
// partly filled round follow as HALF tiles -- 8 vertices on wavefronts 0..7, the other wavefronts (and rows 8..15 of the
// slabs) idle -- so that every workgroup gets a share of that round (313 tiles on 256 workgroups, a 5 000-vertex mesh: one
// round and a round of half tiles instead of two rounds).  FC_HALF_TILES=0 switches them off.
struct TileItems {
    int nv_full, nv_total;
};
inline TileItems tile_items(int ntiles, int grid, int parts_log2) {
    static const bool on = !(dev_env("FC_HALF_TILES") && atoi(dev_env("FC_HALF_TILES")) == 0);
    TileItems t;
    t.nv_full = t.nv_total = ntiles << parts_log2;
    if (!on || parts_log2 != 0 || grid <= 0) return t;
    const int rem = ntiles % grid;
    if (ntiles > grid && rem > 0 && 2 * rem <= grid) {
        t.nv_full = ntiles - rem;
        t.nv_total = t.nv_full + 2 * rem;
    }
    return t;
}
// vertex of row `row` (0..15) of work item `item`, or N for a row without a vertex; parts_log2 != 0: items are
// (tile << parts_log2) + part
__host__ __device__ inline int item_vertex(int item, int row, int nv_full, int parts_log2, int N) {
    if (item < nv_full) return (item >> parts_log2) * kTile + row;
    const int h = item - nv_full;
    return row < 8 ? ((nv_full >> parts_log2) + (h >> 1)) * kTile + (h & 1) * 8 + row : N;
}

// How many workgroups share a tile of 16 vertices (log2): only when the tiles alone leave CUs idle and every part still
// gets a few edges per vertex.  FC_EDGE_PARTS_MAX (development) caps it.
inline int edge_parts_log2(const fc_dims* d) {
    if (d->N <= 0) return 0;
    const int ntiles = (d->N + kTile - 1) / kTile;
    const long deg = (long)d->E / d->N;
    static const int cap = [] { const char* e = dev_env("FC_EDGE_PARTS_MAX"); return e ? atoi(e) : 3; }();       // read once per process
    int pl = 0;
    while (pl < cap && (ntiles << (pl + 1)) <= num_cus() && (deg >> (pl + 1)) >= 8) ++pl;
    return pl;
}

}  // namespace fc
